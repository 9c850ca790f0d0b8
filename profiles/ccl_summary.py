"""Per-kernel mean durations of tests/bench_ccl.py from a rocprofv3 --kernel-trace CSV, grouped by grid (= workload).
    python profiles/ccl_summary.py gpurun_out/ccl_prof/ccl_kernel_trace.csv > profiles/r1d_ccl_kernels.md"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    m = re.search(r"::((?:ccl|inst)_\w+)\(", r["Kernel_Name"])
    if not m:
        continue
    key = (m.group(1), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | blocks (x, y) | launches | mean us |\n|---|---|---|---|")
for (name, gx, gy), v in agg.items():
    v = v[3:] if len(v) > 5 else v           # skip the warm-up launches
    print(f"| `{name}` | {gx} x {gy} | {len(v)} | {sum(v) / len(v):.2f} |")
