"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/fn_ccl.npz with scipy.ndimage.label (scipy 1.15.3).

The connected-component kernels have no counterpart in the reference, so the golden vectors come
from the third-party anchor SURVEY names: scipy.ndimage.label applied to every per-class mask with the 4- or
8-connectivity structuring element, its labels renumbered in raster order of first appearance (scipy already
numbers them that way; the renumbering makes the fixture independent of that detail).

    python oracle/gen_golden_ccl.py
"""
import os

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
STRUCT = {4: ndimage.generate_binary_structure(2, 1), 8: ndimage.generate_binary_structure(2, 2)}


def blocky_labels(rng, B, h, w, C, cell, noise, ignore):
    """Voronoi-like class regions + salt noise + ignore pixels: many components of very different sizes."""
    gh, gw = -(-h // cell), -(-w // cell)
    coarse = rng.integers(0, C, (B, gh, gw))
    lab = np.repeat(np.repeat(coarse, cell, 1), cell, 2)[:, :h, :w].copy()
    flip = rng.random((B, h, w)) < noise
    lab[flip] = rng.integers(0, C, int(flip.sum()))
    lab[rng.random((B, h, w)) < ignore] = 255
    return lab.astype(np.int32)


def scipy_instances(lab, C, connectivity):
    B, h, w = lab.shape
    inst = np.zeros((B, h, w), np.int32)
    ncomp = np.zeros((B, C), np.int32)
    for b in range(B):
        for c in range(C):
            l, n = ndimage.label(lab[b] == c, structure=STRUCT[connectivity])
            if n:
                first = np.full(n + 1, l.size, np.int64)
                np.minimum.at(first, l.ravel(), np.arange(l.size))
                order = np.argsort(first[1:], kind="stable")             # raster order of the first pixels
                renum = np.zeros(n + 1, np.int32)
                renum[order + 1] = np.arange(1, n + 1)
                inst[b] += renum[l]
            ncomp[b, c] = n
    return inst, ncomp


def main():
    rng = np.random.default_rng(11)
    out = {}
    cases = {
        "tiny": dict(B=2, h=7, w=9, C=3, cell=3, noise=0.2, ignore=0.1),
        "feat": dict(B=2, h=65, w=129, C=19, cell=9, noise=0.08, ignore=0.05),
        "noisy": dict(B=1, h=40, w=67, C=4, cell=5, noise=0.5, ignore=0.2),
        "wide": dict(B=1, h=96, w=300, C=19, cell=24, noise=0.02, ignore=0.03),
    }
    for tag, kw in cases.items():
        lab = blocky_labels(rng, **kw)
        out[f"{tag}_labels"] = lab
        out[f"{tag}_C"] = np.int32(kw["C"])
        for conn in (4, 8):
            inst, ncomp = scipy_instances(lab, kw["C"], conn)
            out[f"{tag}_inst{conn}"] = inst
            out[f"{tag}_ncomp{conn}"] = ncomp
    # hand-made shapes: spiral, checkerboard (8-conn joins the diagonals, 4-conn does not), U shapes that merge late
    h, w = 33, 47
    lab = np.full((1, h, w), 255, np.int32)
    yy, xx = np.mgrid[0:h, 0:w]
    lab[0][(yy + xx) % 2 == 0] = 0
    lab[0][(yy % 4 == 1) & (xx > 2) & (xx < w - 3)] = 1
    lab[0][(xx == w - 4) & (yy % 8 < 5)] = 1
    lab[0][(xx == 3) & (yy % 8 >= 4)] = 1
    lab[0, h - 1, :] = 2
    lab[0, :, 0] = 2
    out["shapes_labels"] = lab
    out["shapes_C"] = np.int32(3)
    for conn in (4, 8):
        inst, ncomp = scipy_instances(lab, 3, conn)
        out[f"shapes_inst{conn}"] = inst
        out[f"shapes_ncomp{conn}"] = ncomp
    np.savez_compressed(os.path.join(GOLD, "fn_ccl.npz"), **out)
    for k, v in out.items():
        if "ncomp" in k:
            print(k, int(v.sum()))


if __name__ == "__main__":
    main()
