#!/bin/bash
# trace_kernels.sh TAG [LIB.so]: rocprofv3 kernel trace of a short one-window bench with the product library or a variant; prints
# the per-kernel table (profiles/kernel_roofline.py)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$1
mkdir -p $O
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-latency-mode --no-calibration --no-miou --no-exact-mode --no-pcie --windows 1 --frames-in-flight 15"
if [ -n "$2" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 profiles/experiments/with_lib.py $2 bench.py $Q > $O/trace.log 2>&1
else
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 bench.py $Q > $O/trace.log 2>&1
fi
python3 profiles/kernel_roofline.py $O/trace/t_kernel_trace.csv 15 | grep -E "stem_kernel|proto_dist|head_tapsum|All kernels|conv_igemm"
