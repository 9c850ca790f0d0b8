"""Per-kernel roofline table from a rocprofv3 --kernel-trace of bench.py (profiles/collect_r2.sh):
algorithmic bytes / FLOPs of one step (k frames in flight, 1024x512) divided by the kernel family's time per step.
    python profiles/kernel_roofline.py gpurun_out/prof_r2/trace/t_kernel_trace.csv 15 profiles/r2_steady_state_kernels.csv > profiles/r2_kernel_roofline.md
The optional third argument receives the steady-state per-kernel totals (calls and microseconds per step) as CSV, so that the
table can be re-derived from tracked files.
Only steady-state steps are counted (the first 8 stem launches = eager warm-up, tile autotune and graph warm-up are skipped);
the number of steps is the stem kernel's remaining call count (one launch per step)."""
import csv
import sys
from collections import defaultdict

path, k = sys.argv[1], int(sys.argv[2])
HBM, MFMA, VALU = 8000.0, 2500.0, 157.3          # GB/s, bf16 TFLOP/s, fp32 vector TFLOP/s (MI355X nominal)
rows_all = list(csv.DictReader(open(path)))
stems = sorted(int(r["Start_Timestamp"]) for r in rows_all if "stem_kernel" in r["Kernel_Name"] or "stem_mfma_kernel" in r["Kernel_Name"])
SKIP = 8                                          # eager warm-up passes (incl. the tile autotune) + graph warm-up
t_begin = stems[SKIP]
dur = defaultdict(float)
calls = defaultdict(int)
for r in rows_all:
    if int(r["Start_Timestamp"]) < t_begin:
        continue
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    key = n.split("(")[0]
    for plain in ("stem_kernel", "proto_pool_kernel", "proto_dist_kernel", "upsample_argmax_hist_kernel"):      # templated on the storage type since round 2
        if key.startswith(plain + "<"):
            key = plain
    if key.startswith("stem_mfma_kernel"):        # the bf16 path's stem since the end of round 2 (three bf16 MFMA terms)
        key = "stem_kernel"
    if key.startswith("conv1x1_ws_kernel"):       # the weights-stationary 1x1 kernel (tile id 91): same work, same row
        key = "conv_igemm_kernel"
    elif key.startswith("conv_igemm_kernel"):
        # template arguments: <BM, BN, WM, WN, S, BK, ILV, EPI, AMODE>; AMODE 2 = "bn3 + residual + relu" of the previous block on
        # the operand path (diga_conv_igemm_bf16_fused_in), a bandwidth-bound launch by construction
        args = [a.strip() for a in n[n.index("<") + 1:n.rindex(">")].split(",")] if "<" in n else []
        key = "conv_igemm_kernel[fused residual input]" if len(args) >= 9 and args[8] == "2" else "conv_igemm_kernel"
    dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    calls[key] += 1
steps = calls["stem_kernel"]
hw1, hw2, hw3 = 256 * 512, 129 * 257, 65 * 129
L = [(3, 64, hw2), (4, 128, hw3), (23, 256, hw3), (3, 512, hw3)]          # (blocks, planes, pixels) per layer
act = sum(nb * px * (2 * pl + 4 * pl) + px * 4 * pl for nb, pl, px in L)     # conv-output elements of the bottlenecks (+ downsample)
conv_out_elems = act + hw1 * 64
conv_in_bytes = 0
for i, (nb, pl, px) in enumerate(L):
    cin0 = 64 if i == 0 else L[i - 1][1] * 4
    pin = hw2 if i < 2 else hw3
    conv_in_bytes += 2 * (pin * cin0 * 2) + (nb - 1) * px * 4 * pl * 2        # conv1 + downsample of block 0, conv1 of the others
    conv_in_bytes += nb * (px * pl * 2) * 2                                   # conv2, conv3 inputs
rows = []


def add(name, kind, amount, peak, unit, note):
    if name not in dur:
        return
    t = dur[name] / steps
    ach = amount * k / t / (1e9 if unit == "GB/s" else 1e12)
    rows.append((name, calls[name] // steps, t * 1e6, kind, ach, unit, ach / peak, note))


fused = "conv_igemm_kernel[fused residual input]" in dur
fz_flops = sum((nb - 2) * 2.0 * px * 4 * pl * pl for nb, pl, px in L) if fused else 0.0       # conv1 of blocks 2..nb-1
fz_bytes = sum((nb - 2) * px * (4 * pl * 2 * 3 + pl * 2) for nb, pl, px in L)
add("conv_igemm_kernel", "mfma", 733.1e9 - fz_flops, MFMA, "TFLOP/s",
    ("79 trunk convs + head GEMM (the 25 fused-input conv1 launches are the next row)" if fused else "104 trunk convs + head GEMM") + "; FLOPs as SURVEY 8d")
if fused:
    add("conv_igemm_kernel[fused residual input]", "hbm", fz_bytes, HBM, "GB/s",
        f"conv1 + the previous block's bn3/residual/relu: read raw3 + x, write x' + raw1; also {fz_flops / 1e9:.1f} GFLOP per frame")
apply1 = sum((1 if fused else nb - 1) * px * 4 * pl * 2 * 3 for nb, pl, px in L)
apply2 = sum(px * 4 * pl * 2 * 3 for nb, pl, px in L)
apply0 = sum(nb * px * pl * 2 * 2 * 2 for nb, pl, px in L)
add("bn_apply_kernel<1>", "hbm", apply1, HBM, "GB/s", "read raw + residual, write block output (bf16)")
add("bn_apply_kernel<2>", "hbm", apply2, HBM, "GB/s", "read raw + raw downsample, write block output")
add("bn_apply_kernel<0>", "hbm", apply0, HBM, "GB/s", "read raw, write activation")
add("bn_finalize_kernel", "latency", sum(nb * ((px + 31) // 32) * 2 * 4 * (2 * pl + 4 * pl) for nb, pl, px in L), HBM, "GB/s",
    "104 launches of <= 32 blocks: launch-latency bound, slab bytes shown")
add("stem_kernel", "hbm", 3 * 512 * 1024 * 4 + hw1 * 64 * 2, HBM, "GB/s",
    "stem_mfma_kernel: 7x7 s2 conv as three bf16 MFMA terms; read the fp32 frame, write the raw bf16 map (also 2.47 GFLOP x 3 terms per frame)")
add("bn_relu_maxpool_kernel", "hbm", hw1 * 64 * 2 + hw2 * 64 * 2, HBM, "GB/s", "read stem output, write pooled")
add("proto_pool_kernel", "hbm", hw3 * 2048 * 2, HBM, "GB/s", "one feature read serves both label sets")
add("proto_dist_kernel", "valu", 3.0 * hw3 * 38 * 2048, VALU, "TFLOP/s", "sub + fma per (pixel, prototype, channel)")
add("head_tapsum_pl_kernel", "l2", hw3 * (18 * 19 * 4 + 19 * 4 + 4), HBM, "GB/s", "18 x 76-byte reads per pixel")
print(f"# Per-kernel roofline, {k} frames in flight, 1024x512, {steps} steps in the trace (rocprofv3 --kernel-trace)\n")
print("| kernel | launches/step | us/step | bound | achieved | frac of nominal peak | algorithmic work counted |")
print("|---|---|---|---|---|---|---|")
for name, c, us, kind, ach, unit, frac, note in rows:
    print(f"| `{name}` | {c} | {us:.0f} | {kind} | {ach:.0f} {unit} | {frac:.3f} | {note} |")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write("kernel,calls_per_step,us_per_step\n")
        for name in sorted(dur, key=lambda n_: -dur[n_]):
            f.write(f"\"{name}\",{calls[name] / steps:.2f},{dur[name] / steps * 1e6:.1f}\n")
tot = sum(dur.values()) / steps
print(f"\nAll kernels: {tot * 1e3:.2f} ms of GPU time per step; nominal peaks: HBM {HBM:.0f} GB/s, bf16 MFMA {MFMA:.0f} TFLOP/s, fp32 vector {VALU:.1f} TFLOP/s.")
