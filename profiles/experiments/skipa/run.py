"""Experiment (round 3): how much of a 3x3 conv's K loop is the A operand's fill traffic?  A library built with -DDIGA_EXP_SKIPA
lets 2 of the 3 taps of every tap row fetch the 16-byte zero page instead of their A rows (same number of LDS-DMA instructions,
a third of the A bytes from L2; results are WRONG by construction).  Same-box comparison with the product library:
    python profiles/experiments/skipa/run.py --shapes l3.c2,l4.c2,l2.c2,l1.c2 --tiles 28,48,38,58 --batch 15
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diga_amd import _lib  # noqa: E402

if os.environ.get("DIGA_SKIPA", "1") == "1":
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdiga_hip_skipa.so")
print("library:", _lib.LIB_PATH, flush=True)
sys.argv = [os.path.join(ROOT, "tests", "bench_conv.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
