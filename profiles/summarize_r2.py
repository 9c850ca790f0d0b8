"""Turn gpurun_out/prof_r2 (profiles/collect_r2.sh) into the committed summaries:
   profiles/r2_bench_kernel_stats.csv / r2_bench_2windows_kernel_stats.csv   rocprofv3 --stats tables of the bench command
   profiles/conv_traffic.json    fabric-side bytes of the igemm launches of ONE pass over one window, from the PMC passes:
                                 (2 x FETCH_SIZE + WRITE_SIZE) x 1024  (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE tallies the
                                 128-byte requests of 16-B/lane streams at 64 B -> x2; WRITE_SIZE is exact for 16-B/lane stores;
                                 Infinity-Cache hits are counted, so this is fabric-side traffic, an upper bound of true HBM bytes).
    python profiles/summarize_r2.py gpurun_out/prof_r2 15 "<provenance text>" [prefix = r2]      (round 3: gpurun_out/prof_r3 ... r3)"""
import csv
import json
import os
import shutil
import sys

src, fif, prov = sys.argv[1], int(sys.argv[2]), sys.argv[3]
pre = sys.argv[4] if len(sys.argv) > 4 else "r2"
here = os.path.dirname(os.path.abspath(__file__))
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(here, f"{pre}_bench_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "trace2", "t_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "trace2", "t_kernel_stats.csv"), os.path.join(here, f"{pre}_bench_2windows_kernel_stats.csv"))


def amode(name):
    """conv_igemm_kernel<BM, BN, WM, WN, S, BK, ILV, EPI, AMODE, MF>: AMODE 2 = the fused-residual-input conv1 launches"""
    if "conv1x1_ws_kernel" in name:
        return 0
    args = [a.strip() for a in name[name.index("<") + 1:name.rindex(">")].split(",")]
    return int(args[8]) if len(args) > 8 else 0


def last_pass(path, counter):
    """the igemm launches of the LAST eager pass (105: 104 trunk convs + the head GEMM), split into the plain-input launches
    (bench.py's `roofline` kernel) and the fused-residual-input conv1 launches (priced against HBM in `roofline_hbm`)"""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and
            ("conv_igemm_kernel" in r["Kernel_Name"] or "conv1x1_ws_kernel" in r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    rows = rows[-L:]
    plain = [r for r in rows if amode(r["Kernel_Name"]) != 2]
    fused = [r for r in rows if amode(r["Kernel_Name"]) == 2]
    return sum(float(r["Counter_Value"]) for r in plain), len(plain), sum(float(r["Counter_Value"]) for r in fused), len(fused)


L = 105                                   # 104 trunk convs + the head GEMM, per pass
f, nf, ff, nff = last_pass(os.path.join(src, "fetch", "p_counter_collection.csv"), "FETCH_SIZE")
w, nw, wf, nwf = last_pass(os.path.join(src, "write", "p_counter_collection.csv"), "WRITE_SIZE")
assert nf == nw and nff == nwf, (nf, nw, nff, nwf)
p = os.path.join(here, "conv_traffic.json")
out = json.load(open(p)) if os.path.exists(p) else {}
out[f"frames_in_flight_{fif}"] = {
    "hbm_bytes_per_step": int((2 * f + w) * 1024), "fetch_size_kb_raw": f, "write_size_kb": w, "launches": nf,
    "kernel": "the plain-input igemm launches of one pass (bench.py's `roofline` kernel)",
    "fused_residual_in": {"hbm_bytes_per_step": int((2 * ff + wf) * 1024), "fetch_size_kb_raw": ff, "write_size_kb": wf, "launches": nff},
    "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 over the launches of the last eager pass (fabric-side bytes, Infinity-Cache hits included)",
    "mode": f"one window of {fif} frames, eager launches, two separate --pmc passes (profiles/collect_{pre}.sh)", "collected": prov,
}
json.dump(out, open(p, "w"), indent=1)
print(json.dumps(out[f"frames_in_flight_{fif}"]))
