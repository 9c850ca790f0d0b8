#!/bin/bash
# frames-in-flight x windows sweep of the default bench configuration (HIP-graph replay), same box
#   bash profiles/experiments/sweep_windows.sh "15 2" "15 3" ...
if [ $# -eq 0 ]; then set -- "15 2" "15 3" "12 2" "12 3" "10 3" "20 2" "15 1" "8 4"; fi
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --frames-in-flight $1 --windows $2 --no-cpu-baseline --no-miou --no-calibration --no-roofline --no-latency-mode --no-exact-mode --no-pcie --steps ${STEPS:-30} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fif $1 windows $2:', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" || { echo "fif $1 windows $2: failed"; exit 1; }
done
