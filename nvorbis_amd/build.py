"""Build script for libnvorbis_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnvorbis_hip.so")
SOURCES = ["nvh_api.hip", "nvh_setup.hip", "nvh_launch.hip", "nvh_ops.hip", "kernels.hip", "kernels_imdct.hip", "kernels_spectrum.hip", "kernels_synth.hip", "kernels_parse.hip", "host_setup.cpp", "host_parse.cpp", "host_ogg.cpp"]
# Kernels that measured slower than the default path and are kept for the record: the run kernel, the
# frame-loop kernel, k_imdct_ola.  They are compiled only into the experiments library (build.py --experiments,
# -DNVH_EXPERIMENTS), never into libnvorbis_hip.so.
EXPERIMENT_SOURCES = ["kernels_spectrum2.hip", "kernels_run.hip"]
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


def source_hash():
    """SHA-256 (first 16 hex digits) over everything the binary is made from: every file under csrc/, the public header,
    the source list and the compiler flags.  It is compiled into the library (-DNVH_SRC_HASH, reported by nvh_version())
    so that a stale binary can be told from a fresh one without trusting file times -- the .so is git-ignored but
    travels to the GPU box with the snapshot."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root, _, names in os.walk(CSRC):
        files += [os.path.join(root, f) for f in names]
    files.append(os.path.join(HERE, "..", "include", "nvorbis_hip.h"))
    for f in sorted(files):
        h.update(os.path.relpath(f, HERE).encode())
        h.update(open(f, "rb").read())
    h.update(repr((SOURCES, FLAGS)).encode())
    return h.hexdigest()[:16]


_HASH_MARK = b"nvh-src-hash="


def embedded_hash(path):
    """The source hash a built library carries (None: no such file, or a build from before the hash existed)."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    k = blob.find(_HASH_MARK)
    return blob[k + len(_HASH_MARK):k + len(_HASH_MARK) + 16].decode("ascii", "replace") if k >= 0 else None


def needs_build(path=None):
    return embedded_hash(path or OUT) != source_hash()


def _hash_flag():
    return ['-DNVH_SRC_HASH="%s"' % source_hash()]


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + _hash_flag() + ["-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


DEBUG_OUT = os.path.join(HERE, "libnvorbis_hip_dbg.so")
EXPERIMENTS_OUT = os.path.join(HERE, "libnvorbis_hip_exp.so")


def build_experiments(verbose=False):
    """The experiments build (-DNVH_EXPERIMENTS): the release library plus the quarantined kernels behind their opt-in
    switches (NVH_RUN=1, NVH_MULTI=1, NVH_FUSED_OLA=1).  Load it with NVH_LIB=nvorbis_amd/libnvorbis_hip_exp.so; the tests
    marked `experiments` do (and skip when it has not been built)."""
    cmd = [hipcc()] + FLAGS + _hash_flag() + ["-DNVH_EXPERIMENTS", "-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES + EXPERIMENT_SOURCES] + ["-o", EXPERIMENTS_OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return EXPERIMENTS_OUT


def build_debug(verbose=False):
    """The profiling build (-DNVH_DEBUG): the spectrum kernels take a timestamp buffer and a phase mask
    (nvh_debug_set_buffer, NVH_DEBUG_SPECTRUM_MASK; tools/dbg_phase*.py, tools/pmc_phases.sh).  Load it with
    NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so.  The release library has neither the parameters nor the export."""
    cmd = [hipcc()] + FLAGS + _hash_flag() + ["-DNVH_DEBUG", "-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", DEBUG_OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return DEBUG_OUT


if __name__ == "__main__":
    if "--debug" in sys.argv:
        print(build_debug(verbose=True))
    elif "--experiments" in sys.argv:
        print(build_experiments(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(OUT)
