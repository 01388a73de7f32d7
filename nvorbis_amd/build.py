"""Build script for libnvorbis_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnvorbis_hip.so")
SOURCES = ["nvh_api.hip", "nvh_setup.hip", "nvh_launch.hip", "nvh_ops.hip", "kernels.hip", "kernels_imdct.hip", "kernels_spectrum.hip", "kernels_spectrum2.hip", "kernels_run.hip", "kernels_parse.hip", "host_setup.cpp", "host_parse.cpp", "host_ogg.cpp"]
# -ffp-contract=off: bit-exact parity with the reference needs separately rounded mul/add (no v_fma_f32);
# fp32 denormals are preserved by default (no -fgpu-flush-denormals-to-zero).
# -fno-slp-vectorize: the SLP vectorizer pairs the butterflies of the wavefront IMDCT into v_pk_add_f32 / v_pk_mul_f32;
# the register-pair shuffling around them (inside a 64-VGPR budget) costs more than the packed issue saves:
# k_spectrum_imdct 42.8 -> 39.1 us without it, same bits.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libnvorbis_hip.so cannot be built (there is no CPU fallback)")
    return exe


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for root, _, files in os.walk(CSRC):
        for f in files:
            if os.path.getmtime(os.path.join(root, f)) > t:
                return True
    if os.path.getmtime(os.path.abspath(__file__)) > t:  # the compiler flags live in this file
        return True
    inc = os.path.join(HERE, "..", "include", "nvorbis_hip.h")
    return os.path.getmtime(inc) > t


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


DEBUG_OUT = os.path.join(HERE, "libnvorbis_hip_dbg.so")


def build_debug(verbose=False):
    """The profiling build (-DNVH_DEBUG): the spectrum kernels take a timestamp buffer and a phase mask
    (nvh_debug_set_buffer, NVH_DEBUG_SPECTRUM_MASK; tools/dbg_phase*.py, tools/pmc_phases.sh).  Load it with
    NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so.  The release library has neither the parameters nor the export."""
    cmd = [hipcc()] + FLAGS + ["-DNVH_DEBUG", "-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", DEBUG_OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DEBUG_OUT


if __name__ == "__main__":
    if "--debug" in sys.argv:
        print(build_debug(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(OUT)
