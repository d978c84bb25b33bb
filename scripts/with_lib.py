"""Run a script (or `-m module`) of this repo against another build of the library -- A/B builds with -D knobs:
    python scripts/with_lib.py <libicp_variant.so> scripts/run_sequence.py --data_dir ...
    python scripts/with_lib.py <libicp_variant.so> -m pytest tests -m gpu -q
The variant is loaded in place of lidar_slam_from_scratch_amd/libicp_mi355x.so for this process only."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lidar_slam_from_scratch_amd import capi
so = os.path.abspath(sys.argv[1])
capi._LIB = capi.load_library(so)
if sys.argv[2] == "-m":
    sys.argv = [sys.argv[3]] + sys.argv[4:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    sys.argv = sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")
