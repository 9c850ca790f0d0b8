#!/bin/bash
# Round-3 profile collection (same passes as collect_r2.sh) on the GPU box (run from the repo root through gpurun; every rocprofv3 run under its own timeout):
#   1. rocprofv3 --kernel-trace --stats of the bench command with ONE window (HIP-graph replay, 15 frames in flight): with two
#      windows the kernels of the two streams overlap and a kernel's duration is no longer its own
#   2. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC slots) of an eager 2-step run, one window
#   3. the default bench command (two windows on two streams) for the record
#   4. two --pmc passes (SQ counters; TCC + GRBM counters) over the conv kernel alone at 15 frames, shipped tiles (28 / 48 / 58 / 66 / 67 and the weights-stationary 91 where it applies)
# Outputs land in gpurun_out/prof_r3/; profiles/summarize_r2.py (prefix r3), kernel_roofline.py and pmc_conv.py make the committed summaries.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r3
mkdir -p $O
FIF=${FIF:-15}
Q="--no-cpu-baseline --no-latency-mode --no-calibration --no-miou --no-exact-mode --no-pcie"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- \
  python3 bench.py --steps 20 --warmup 5 $Q --windows 1 --frames-in-flight $FIF > $O/trace.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- \
  python3 bench.py --steps 2 --warmup 1 $Q --no-roofline --no-graph --windows 1 --frames-in-flight $FIF > $O/fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- \
  python3 bench.py --steps 2 --warmup 1 $Q --no-roofline --no-graph --windows 1 --frames-in-flight $FIF > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t --output-format csv -- \
  python3 bench.py --steps 20 --warmup 5 $Q --no-roofline --frames-in-flight $FIF > $O/trace2.log 2>&1
CONV="tests/bench_conv.py --batch $FIF --shapes l3.c1,l3.c2,l3.c3,l4.c1,l4.c2,l4.c3 --tiles 28,48,58,66,67,91 --reps 3"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
  -d $O/pmc_sq -o p --output-format csv -- python3 $CONV --manifest $O/pmc_sq_manifest.json > $O/pmc_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE \
  -d $O/pmc_tcc -o p --output-format csv -- python3 $CONV --manifest $O/pmc_tcc_manifest.json > $O/pmc_tcc.log 2>&1
ls $O/*
