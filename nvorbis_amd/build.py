"""Build script for libnvorbis_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libnvorbis_hip.so")
SOURCES = ["nvh_api.hip", "nvh_setup.hip", "nvh_launch.hip", "nvh_ops.hip", "nvh_comm.hip", "kernels.hip", "kernels_imdct.hip", "kernels_spectrum.hip", "kernels_synth.hip", "kernels_parse.hip", "host_setup.cpp", "host_parse.cpp", "host_slab.cpp", "host_ogg.cpp"]
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


_COMPILER_ID = None


def _compiler_id():
    """`hipcc --version`, once: part of every object's cache key."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            _COMPILER_ID = subprocess.check_output([hipcc(), "--version"], stderr=subprocess.STDOUT).decode("utf-8", "replace")
        except (OSError, subprocess.CalledProcessError):
            _COMPILER_ID = "unknown"
    return _COMPILER_ID


def _mtime(path):
    try:
        return os.path.getmtime(path)
    except OSError:
        return 0.0


def source_hash():
    """SHA-256 (first 16 hex digits) over everything the binary is made from: every file under csrc/, the public header,
    the source list and the compiler flags.  It is compiled into the library (-DNVH_SRC_HASH, reported by nvh_version())
    so that a stale binary can be told from a fresh one without trusting file times -- the .so is git-ignored but
    travels to the GPU box with the snapshot."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root, _, names in os.walk(CSRC):  # sources only: editor backups and stray objects do not make a library stale
        files += [os.path.join(root, f) for f in names if f.endswith((".hip", ".cpp", ".h", ".inc"))]
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


def _compile_link(sources, out, extra, verbose=False):
    """Every source to its own object (in parallel, cached under csrc/.obj by a hash of the file, the headers and the flags),
    then one link.  The kernels of a .hip file are reached from the other files through their host-side launch stubs, which
    are ordinary symbols: no relocatable device code is needed."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    hdr = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".h", ".inc")):
            hdr.update(f.encode())
            hdr.update(open(os.path.join(CSRC, f), "rb").read())
    hdr.update(open(os.path.join(HERE, "..", "include", "nvorbis_hip.h"), "rb").read())
    cflags = [f for f in FLAGS if f != "-shared"] + extra
    hdr.update(repr(cflags).encode())
    hdr.update(_compiler_id().encode())  # a ROCm upgrade must not reuse objects of the old compiler
    hash_flag = _hash_flag()

    def one(src):
        path = os.path.join(CSRC, src)
        h = hashlib.sha256(hdr.digest())
        h.update(open(path, "rb").read())
        # the embedded source hash lives in nvh_api.hip only: the other objects do not change with it
        if src == "nvh_api.hip":
            h.update(hash_flag[0].encode())
        obj = os.path.join(objdir, "%s.%s.o" % (src, h.hexdigest()[:16]))
        if not os.path.exists(obj):
            # keep a few older objects of this source (A/B variant builds alternate between flag sets), drop the rest; other
            # builders may be at work in the same directory (the multi-rank tests): a file that is already gone is fine
            mine = sorted((o for o in os.listdir(objdir) if o.startswith(src + ".")),
                          key=lambda o: _mtime(os.path.join(objdir, o)), reverse=True)
            for stale in mine[6:]:
                try:
                    os.remove(os.path.join(objdir, stale))
                except FileNotFoundError:
                    pass
            tmp = "%s.%d.tmp" % (obj, os.getpid())
            cmd = [hipcc()] + cflags + hash_flag + ["-x", "hip", "-c", path, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, obj)  # (atomic: a concurrent builder never links a half-written object)
        return obj

    with ThreadPoolExecutor(max(1, min(len(sources), os.cpu_count() or 1))) as ex:
        objs = list(ex.map(one, sources))
    link_flags = [f for f in FLAGS if f.startswith("--offload-arch=") or f in ("-fPIC", "-shared")]
    tmp_out = "%s.%d.tmp" % (out, os.getpid())
    cmd = [hipcc()] + link_flags + objs + ["-o", tmp_out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp_out, out)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    return _compile_link(SOURCES, OUT, [], verbose)


DEBUG_OUT = os.path.join(HERE, "libnvorbis_hip_dbg.so")

def build_debug(verbose=False):
    """The profiling build (-DNVH_DEBUG): the spectrum kernels take a timestamp buffer and a phase mask
    (nvh_debug_set_buffer, NVH_DEBUG_SPECTRUM_MASK; tools/dbg_phase_synth.py, tools/dbg_phase_parse.py).  Load it with
    NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so.  The release library has neither the parameters nor the export."""
    return _compile_link(SOURCES, DEBUG_OUT, ["-DNVH_DEBUG"], verbose)


if __name__ == "__main__":
    if "--debug" in sys.argv:
        print(build_debug(verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(OUT)
