"""Static guard of the hand-counted vmcnt pipeline of the fused-residual-input conv1 (csrc/conv_igemm.hip, the DEEP loop).

That loop keeps two K steps of operand registers in flight with asm loads the compiler does not count; its `s_waitcnt vmcnt(N)`
constants assume that between two barriers a wave issues EXACTLY: B_CH LDS-DMA pieces (next step's weights), NA = 4 + 2 A_CH
register loads (affine vectors, raw and residual chunks of the step after next), A_CH block-input write-back stores -- in that
order and nothing else on the vector-memory counter.  One extra load or store emitted by a future compiler would make a wait
return one operation early: a silently corrupted K step, not a crash.  This test disassembles the gfx950 code object that ships
in conv_igemm.o and checks the emitted instruction stream against those constants.  CPU-only: no GPU needed.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "diga_amd", "csrc", "conv_igemm.o")
LLVM = "/opt/rocm/lib/llvm/bin"

# the tile shapes whose fused-input (AMODE 2) kernel takes the DEEP path: BK = 64 and A_CH = BM / (8 * waves) <= 2
DEEP_TILES = {(64, 64, 2, 2): (2, 2), (64, 128, 2, 2): (2, 4), (128, 256, 2, 4): (2, 4)}      # (BM, BN, WM, WN) -> (A_CH, B_CH)


def _disassemble(tmp_path):
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not (os.path.exists(objdump) and os.path.exists(OBJ)):
        pytest.skip("llvm-objdump or conv_igemm.o not available")
    work = tmp_path / "conv_igemm.o"
    shutil.copy(OBJ, work)                                   # --offloading writes the extracted bundles next to its input
    subprocess.run([objdump, "--offloading", str(work)], check=True, capture_output=True, timeout=120)
    dev = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert len(dev) == 1, os.listdir(tmp_path)
    out = subprocess.run([objdump, "-d", str(tmp_path / dev[0])], check=True, capture_output=True, text=True, timeout=300).stdout
    funcs, cur = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return funcs


def _is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "scratch_", "flat_"))


def _vmcnt(ins):
    m = re.search(r"vmcnt\((\d+)\)", ins)
    return int(m.group(1)) if m else None


def test_deep_fused_input_loop_matches_its_counted_waits(tmp_path):
    funcs = _disassemble(tmp_path)
    checked = 0
    for (bm, bn, wm, wn), (a_ch, b_ch) in DEEP_TILES.items():
        # conv_igemm_kernel<BM, BN, WM, WN, S = 2, BK = 64, ILV = false, EPI = 0, AMODE = 2, MF = 16>
        pat = f"conv_igemm_kernelILi{bm}ELi{bn}ELi{wm}ELi{wn}ELi2ELi64ELb0ELi0ELi2ELi16EEE"
        names = [n for n in funcs if pat in n]
        assert len(names) == 1, (pat, names)
        ins = funcs[names[0]]
        na = 4 + 2 * a_ch
        allowed_waits = {0, a_ch, na, na + a_ch}
        # Anchor on the waits only the DEEP loop has -- vmcnt(NA + A_CH): "the next step's weights have landed; one operand set and
        # this step's write-back stay in flight" -- and walk BACK to the barrier that opened the step: everything the wave put on
        # the vector-memory counter in between must be exactly the step's B_CH pieces, NA loads and A_CH stores, in that order.
        anchors = [i for i, x in enumerate(ins) if x.startswith("s_waitcnt") and _vmcnt(x) == na + a_ch]
        assert len(anchors) >= 2, f"{names[0]}: expected the even and the odd half step, found {len(anchors)} vmcnt({na + a_ch}) waits"
        want = ["global_load_lds_dwordx4"] * b_ch + ["global_load_dwordx4"] * na + ["global_store_dwordx4"] * a_ch
        for a in anchors:
            j = a
            while j > 0 and ins[j - 1].split()[0] != "s_barrier":
                j -= 1
            assert j > 0, (names[0], "no barrier before the wait")
            step = ins[j:a]
            kinds = [x.split()[0] for x in step if _is_vmem(x.split()[0])]
            assert kinds == want, (names[0], "vector-memory instructions of one DEEP step", kinds, want)
            assert sum(x.split()[0].startswith("v_mfma") for x in step) > 0, (names[0], "the step's MFMAs sit between issue and wait")
            # the waits that close the step (wait_b's four cases): nothing but the counted constants
            waits = [_vmcnt(x) for x in step if x.startswith("s_waitcnt") and _vmcnt(x) is not None]
            assert set(waits) <= allowed_waits, (names[0], "vmcnt constants inside the step", waits, sorted(allowed_waits))
        # every vmcnt wait inside the loop's extent is one of the counted constants, and no other kind of vector-memory
        # instruction (a spill, a re-materialised parameter load) sits on the counter there
        lo, hi = min(anchors), max(anchors)
        while lo > 0 and ins[lo - 1].split()[0] != "s_barrier":
            lo -= 1
        for x in ins[lo:hi + 1]:
            mn = x.split()[0]
            if mn.startswith("s_waitcnt") and _vmcnt(x) is not None:
                assert _vmcnt(x) in allowed_waits, (names[0], x)
            if _is_vmem(mn):
                assert mn in ("global_load_lds_dwordx4", "global_load_dwordx4", "global_store_dwordx4"), (names[0], x)
        # prologue: NA loads, B_CH pieces, NA loads, then the first wait leaves exactly the second operand set in flight
        first_barrier = next(i for i, x in enumerate(ins) if x.split()[0] == "s_barrier")
        pro = ins[:first_barrier]
        vm0 = [x.split()[0] for x in pro if _is_vmem(x.split()[0])]
        tail = vm0[-(2 * na + b_ch):]
        assert tail == ["global_load_dwordx4"] * na + ["global_load_lds_dwordx4"] * b_ch + ["global_load_dwordx4"] * na, (names[0], tail)
        w0 = [_vmcnt(x) for x in pro if x.startswith("s_waitcnt") and _vmcnt(x) is not None]
        assert w0 and w0[-1] == na, (names[0], w0)
        checked += 1
    assert checked == len(DEEP_TILES)
