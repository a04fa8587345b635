"""bench.py on a differently built library (timing experiments): CSVAE_EXP_LIB=<path> python tools/bench_variant.py [bench.py flags]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vae_conditional_sampling_amd import _lib  # noqa: E402

if os.environ.get("CSVAE_EXP_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["CSVAE_EXP_LIB"])
import bench  # noqa: E402

bench.main()
