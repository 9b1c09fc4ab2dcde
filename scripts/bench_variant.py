"""Development: run bench.py against another build of the library (same box, same process order A/B).
usage: python scripts/bench_variant.py spatialcore_amd/libvar_X.so [bench.py flags]"""
import sys, runpy
sys.path.insert(0, ".")
from spatialcore_amd import _lib
_lib.LIB_PATH = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
