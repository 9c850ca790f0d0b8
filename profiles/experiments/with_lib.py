"""Run a script of this repo against another build of the library:  python profiles/experiments/with_lib.py LIB.so script.py args...
(experiments only: the product always loads diga_amd/csrc/libdiga_hip.so)"""
import os, runpy, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import diga_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
script = sys.argv[2]
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
runpy.run_path(script, run_name="__main__")
