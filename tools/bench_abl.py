"""bench.py against an ablation build of the library: LAS_ABL_LIB=<path to a liblas_hip.so built with CXXFLAGS_EXTRA=-D...> python tools/bench_abl.py <bench.py flags>.
Alternate it with the product library in ONE gpurun call (box-to-box variance is larger than most effects)."""
import os, sys, runpy
sys.path.insert(0, ".")
import las_pytorch_amd._cabi as c
if os.environ.get("LAS_ABL_LIB"): c.LIB_PATH = os.path.abspath(os.environ["LAS_ABL_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
