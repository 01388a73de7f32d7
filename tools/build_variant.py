"""build_variant.py NAME [-DMACRO ...] -- a library build with extra compiler flags as build_ab/lib_NAME.so, for same-box A/B runs
(tools/ab_bench.sh, tools/ab_libs.sh; load with NVH_LIB=... NVH_ALLOW_STALE=1).  Ablation macros of kernels_synth.hip: NVH_ABL_*."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nvorbis_amd import build
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(build.HERE, "..", "build_ab", "lib_%s.so" % name)
os.makedirs(os.path.dirname(out), exist_ok=True)
print(build._compile_link(build.SOURCES, os.path.abspath(out), extra))
