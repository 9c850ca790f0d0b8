#!/bin/bash
# Round-1 profile collection on the GPU box (run from the repo root through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the bench command with ONE window (hipGraph replay, 15 frames in flight): with two
#      windows kernels of the two streams overlap and their individual durations are no longer per-kernel rooflines
#   2. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC slots) of an eager 3-step run
# Outputs land in gpurun_out/prof_r1c/; profiles/summarize.py turns them into the committed summaries.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r1d
mkdir -p $O
FIF=${FIF:-15}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- \
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --no-calibration --no-miou --windows 1 --frames-in-flight $FIF > $O/trace.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-roofline --no-calibration --no-miou --no-graph --windows 1 --frames-in-flight $FIF > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- \
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency-mode --no-roofline --no-calibration --no-miou --no-graph --windows 1 --frames-in-flight $FIF > $O/write.log 2>&1
# 3. the default bench command (two windows on two streams) for the record
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t --output-format csv -- \
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --no-calibration --no-miou --no-roofline --frames-in-flight $FIF > $O/trace2.log 2>&1
ls $O/*
