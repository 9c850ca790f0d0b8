#!/bin/bash
# kernel trace of the one-frame-per-step configuration (HIP-graph replay): where do the 3.3 ms go?
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r2_1f
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- \
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency-mode --no-calibration --no-miou --no-roofline --windows 1 --frames-in-flight 1 > $O/trace.log 2>&1
python3 - <<'PY'
import csv
from collections import defaultdict
rows = list(csv.DictReader(open("gpurun_out/prof_r2_1f/trace/t_kernel_trace.csv")))
stems = sorted(int(r["Start_Timestamp"]) for r in rows if "stem_kernel" in r["Kernel_Name"])
t0, t1 = stems[-60], stems[-10]
sel = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if t0 <= int(r["Start_Timestamp"]) < t1))
steps = 50
busy = sum(e - s for s, e, _ in sel)
gaps = sum(max(0, sel[i + 1][0] - sel[i][1]) for i in range(len(sel) - 1))
dur = defaultdict(float); cnt = defaultdict(int)
for s, e, n in sel:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    dur[k] += e - s; cnt[k] += 1
print(f"per step: wall {(t1 - t0) / steps / 1e3:.1f} us, kernel-busy {busy / steps / 1e3:.1f} us, gaps between kernels {gaps / steps / 1e3:.1f} us, launches {len(sel) / steps:.0f}")
for k in sorted(dur, key=lambda k: -dur[k]):
    print(f"  {k:36s} {cnt[k] / steps:6.1f} launches  {dur[k] / steps / 1e3:8.1f} us  avg {dur[k] / cnt[k] / 1e3:6.2f} us")
PY
