"""Per-(shape, tile) counter table of the conv kernel from two rocprofv3 --pmc passes over tests/bench_conv.py
(profiles/collect_r2.sh).  Rows are aligned with the launch order bench_conv.py wrote (--manifest): only conv_igemm_kernel
dispatches are counted, in dispatch order; the LAST launch of every (shape, tile) group is reported.
    python profiles/pmc_conv.py gpurun_out/prof_r2 > profiles/r2_conv_pmc.txt"""
import csv
import json
import os
import sys
from collections import defaultdict

src = sys.argv[1]


def load(pass_dir):
    rows = defaultdict(dict)          # dispatch id -> counter -> value
    names = {}
    path = os.path.join(src, pass_dir, "p_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        if "conv_igemm_kernel" not in r["Kernel_Name"] and "conv1x1_ws_kernel" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    order = sorted(rows)
    return [rows[d] for d in order], [names[d] for d in order]


def groups(man, seq):
    i = 0
    for g in man:
        for l in g["launches"]:
            i += l["n"]
            yield g["shape"], g["B"], l, seq[i - 1]
    assert i == len(seq), (i, len(seq))


sq, _ = load("pmc_sq")
tc, _ = load("pmc_tcc")
man_sq = json.load(open(os.path.join(src, "pmc_sq_manifest.json")))
man_tc = json.load(open(os.path.join(src, "pmc_tcc_manifest.json")))
print("# rocprofv3 --pmc over tests/bench_conv.py --batch 15 (15 frames in flight), the tiles the engine's autotune ships at that size")
print("# pass 1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT")
print("# pass 2: TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE")
print("# us / TFLOP/s: event timings printed by bench_conv.py DURING pass 1, i.e. under the profiler (about half the un-profiled speed;")
print("#   un-profiled per-launch times are in rN_kernel_roofline.md / the bench JSON) -- never compare them with un-profiled timings.")
print("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): SIMD-cycles the matrix pipe was busy (= MFMAs x 32,")
print("#   independent of the tile) over all SIMD-cycles of the launch, both under the profiler's clocks.")
print("# wait_any / wait_inst / active: share of wave-cycles parked (s_waitcnt, barrier) / stalled at issue (MFMA pipe, dependencies) / issuing.")
print("shape tile blocks us TFLOP/s wait_any wait_inst active mfma_busy lds_bank_conflict L2_hit TCC_REQ fabric_read_MB(2 x EA0_RDREQ x 64 B)")
tcmap = {(s, l["tile"]): c for s, b, l, c in groups(man_tc, tc)}
for shape, B, l, c in groups(man_sq, sq):
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    t = tcmap.get((shape, l["tile"]), {})
    hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0, 1.0)
    print(f"{shape} t{l['tile']} {l['blocks']} {l['us']:.1f} {l['tflops']:.0f} {c.get('SQ_WAIT_ANY', 0) / wc:.2f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} {busy:.2f} {c.get('SQ_LDS_BANK_CONFLICT', 0):.0f} "
          f"{hit / max(hit + miss, 1):.3f} {t.get('TCC_REQ_sum', 0):.3e} {2 * t.get('TCC_EA0_RDREQ_sum', 0) * 64 / 1e6:.0f}")
