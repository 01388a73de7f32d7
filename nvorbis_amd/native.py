"""ctypes binding of libnvorbis_hip.so (include/nvorbis_hip.h).

This is the Python stand-in for the C# P/Invoke shim (no .NET in the build image): the same C ABI,
the same call sequence.  There is no fallback: if the shared library is missing it is built with
hipcc, and if that fails the import raises.
"""
import ctypes as C
import os

from . import build as _build

OK = 0
ERR_INVALID_DATA, ERR_ARGUMENT, ERR_RUNTIME, ERR_NOMEM = -1, -2, -3, -4
ERR_NOT_VORBIS, ERR_DEVICE, ERR_UNSUPPORTED, ERR_NO_GPU = -5, -6, -7, -8
PKT_EOS, PKT_RESYNC = 1, 2

_ERRNAMES = {
    ERR_INVALID_DATA: "InvalidDataException", ERR_ARGUMENT: "ArgumentOutOfRangeException",
    ERR_RUNTIME: "runtime fault (IndexOutOfRange/NullReference class)", ERR_NOMEM: "out of memory",
    ERR_NOT_VORBIS: "not a Vorbis stream", ERR_DEVICE: "HIP error", ERR_UNSUPPORTED: "unsupported stream",
    ERR_NO_GPU: "no HIP device (there is no CPU fallback)",
}


class NvhError(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        extra = ""
        if code == ERR_DEVICE:
            extra = " hipError=%d" % lib().nvh_last_hip_error()
        super().__init__("%s failed: %d (%s)%s" % (where, code, _ERRNAMES.get(code, "?"), extra))


# symbol -> (restype, argtypes); also the list tests check against the header
_u8p, _f32p, _i64p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
_vp, _vpp, _ip = C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)
SIGNATURES = {
    "nvh_version": (C.c_char_p, []),
    "nvh_last_hip_error": (C.c_int, []),
    "nvh_device_count": (C.c_int, []),
    "nvh_ctx_create": (C.c_int, [C.c_int, _vpp]),
    "nvh_ctx_destroy": (None, [_vp]),
    "nvh_ctx_set_hip_stream": (C.c_int, [_vp, _vp]),
    "nvh_ctx_synchronize": (C.c_int, [_vp]),
    "nvh_ctx_set_parse_lanes": (C.c_int, [_vp, C.c_int]),
    "nvh_dev_alloc": (C.c_int, [_vp, C.c_size_t, _vpp]),
    "nvh_dev_free": (None, [_vp, _vp]),
    "nvh_dev_upload": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "nvh_dev_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "nvh_measure_copy": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_int, _f32p]),
    "nvh_stream_bitrates": (C.c_int, [_vp, _ip, _ip, _ip]),
    "nvh_mdct_reverse": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int64]),
    "nvh_inverse_couple": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "nvh_mode_decode": (C.c_int, [_vp, _vp, C.c_int, _vp] + [C.POINTER(C.c_int)] * 5),
    "nvh_residue_decode": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(C.c_int)]),
    "nvh_window_apply": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int64]),
    "nvh_overlap_buffers": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "nvh_copy_buffer": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int64, _vp, C.c_int, C.POINTER(C.c_int)]),
    "nvh_stream_mode_info": (C.c_int, [_vp, C.c_int] + [C.POINTER(C.c_int)] * 3),
    "nvh_stream_codebook_info": (C.c_int, [_vp, C.c_int] + [C.POINTER(C.c_int)] * 7),
    "nvh_stream_codebook_tables": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "nvh_stream_position_state": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "nvh_stream_set_position_state": (C.c_int, [_vp, C.c_int, C.c_int64]),
    "nvh_stream_drop_pending": (C.c_int, [_vp]),
    "nvh_stream_packet_sample_count": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "nvh_stream_reset": (C.c_int, [_vp]),
    "nvh_stream_index_packets": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.POINTER(C.c_int64)]),
    "nvh_floor0_apply": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int64, _vp]),
    "nvh_floor1_apply": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int64, _vp]),
    "nvh_stream_floor_info": (C.c_int, [_vp, C.c_int] + [C.POINTER(C.c_int)] * 3),
    "nvh_mdct_tables": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp]),
    "nvh_calc_window": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp]),
    "nvh_calc_overlap": (C.c_int, [C.c_int, C.c_int, C.c_int, _ip, _ip, _ip]),
    "nvh_stream_open": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vpp]),
    "nvh_stream_close": (None, [_vp]),
    "nvh_stream_info": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "nvh_stream_set_clip": (C.c_int, [_vp, C.c_int]),
    "nvh_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "nvh_pinned_free": (None, [_vp]),
    "nvh_stream_set_gpu_parse": (C.c_int, [_vp, C.c_int]),
    "nvh_stream_has_clipped": (C.c_int, [_vp, _ip]),
    "nvh_stream_position": (C.c_int, [_vp, _i64p, _i64p, _ip]),
    "nvh_stream_push_packet": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int]),
    "nvh_stream_push_packets": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "nvh_stream_push_end": (C.c_int, [_vp]),
    "nvh_stream_pending": (C.c_int, [_vp, _ip, _i64p]),
    "nvh_stream_pending_geometry": (C.c_int, [_vp, _vp, C.c_int]),
    "nvh_stream_pending_slabs": (C.c_int, [_vp, _vp, C.c_int64, _i64p, _vp, C.c_int]),
    "nvh_stream_lattice_pool": (C.c_int, [_vp, _vp, C.c_int64, _i64p]),
    "nvh_stream_vq_pool": (C.c_int, [_vp, _vp, C.c_int64, _i64p]),
    "nvh_stream_synth_begin": (C.c_int, [_vp, _vp, C.c_int64, _i64p]),
    "nvh_stream_synth_end": (C.c_int, [_vp, _i64p]),
    "nvh_stream_synth": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i64p]),
    "nvh_stream_parse_errors": (C.c_int, [_vp, _i32p, _i64p, C.c_int, _ip]),
    "nvh_batch_upload": (C.c_int, [_vp, _vpp]),
    "nvh_batch_info": (C.c_int, [_vp, _ip, _ip, _i64p, _i64p]),
    "nvh_batch_stats": (C.c_int, [_vp, _i64p]),
    "nvh_batch_kernels": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "nvh_stream_kernels": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "nvh_batch_synth": (C.c_int, [_vp, _vp, C.c_int64]),
    "nvh_batch_time": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _f32p, _f32p]),
    "nvh_batch_free": (None, [_vp]),
    "nvh_ogg_demux": (C.c_int, [_vp, C.c_size_t, _vp, C.c_int64, _vp, _vp, _vp, C.c_int, _ip, _i64p]),
    "nvh_ogg_demux_forward": (C.c_int, [_vp, C.c_size_t, C.c_int, _vp, C.c_int64, _vp, _vp, _vp, C.c_int, _ip, _i64p, _ip]),
    "nvh_ogg_index_open": (C.c_int, [_vp, C.c_size_t, C.c_int, C.POINTER(_vp)]),
    "nvh_ogg_index_close": (None, [_vp]),
    "nvh_ogg_index_info": (C.c_int, [_vp, _ip, _ip, _ip, _i64p, _ip]),
    "nvh_ogg_index_page": (C.c_int, [_vp, C.c_int, _i64p, _ip, _ip, _ip]),
    "nvh_ogg_seek": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _i64p, _i64p]),
    "nvh_ogg_demux_stream": (C.c_int, [_vp, C.c_size_t, C.c_int, _vp, C.c_int64, _vp, _vp, _vp, C.c_int, _ip, _i64p, _ip]),
    "nvh_ogg_index_packets": (C.c_int, [_vp, C.c_size_t, C.c_int, _vp, C.c_int64, _vp, _vp, _vp, C.c_int, _ip, _i64p, _i64p, _ip]),
    # the corpus gather through RCCL's C API (nvh_comm.hip): what a host without torch.distributed binds to
    "nvh_comm_unique_id": (C.c_int, [_vp]),
    "nvh_comm_create": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "nvh_comm_destroy": (None, [_vp]),
    "nvh_comm_info": (C.c_int, [_vp, _ip, _ip]),
    "nvh_comm_allgather_i64": (C.c_int, [_vp, _i64p, C.c_int, _i64p]),
    "nvh_comm_gather_pcm": (C.c_int, [_vp, _vp, C.c_int64, _vp, _i64p, C.c_int, C.c_int]),
}

_lib = None


def lib_path():
    # NVH_LIB: load another build of the library (development aid: same-box A/B runs, tools/ab_bench.sh)
    return os.environ.get("NVH_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libnvorbis_hip.so")


def lib():
    """Load (building first if needed) the native library.  Raises if it cannot be had."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7).
        # Loading torch first makes the dynamic loader bind our NEEDED libamdhip64.so.7 to that copy;
        # the other order would map a second runtime from /opt/rocm and torch.cuda would see no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = lib_path()
        # A stale binary must never run (the .so is git-ignored but travels to the GPU box with the snapshot): the
        # library carries the hash of the sources it was built from.  The default library is rebuilt when it does not
        # match; an explicitly chosen one (NVH_LIB: debug / experiments / A-B builds) raises instead, unless
        # NVH_ALLOW_STALE=1 says the mismatch is intended (tools/ab_libs.sh compares builds of older commits).
        if os.environ.get("NVH_LIB"):
            if not os.environ.get("NVH_ALLOW_STALE") and _build.embedded_hash(path) != _build.source_hash():
                raise RuntimeError("%s was built from other sources (has %s, tree is %s): rebuild it or set NVH_ALLOW_STALE=1"
                                   % (path, _build.embedded_hash(path), _build.source_hash()))
        elif _build.needs_build(path):
            import sys
            sys.stderr.write("nvorbis_amd: %s is missing or stale (has %s, sources are %s): building\n"
                             % (path, _build.embedded_hash(path), _build.source_hash()))
            _build.build(force=True)
            if _build.needs_build(path):
                raise RuntimeError("libnvorbis_hip.so does not carry the hash of its sources after a rebuild")
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here == ABI drift, fail loudly
            fn.restype = res
            fn.argtypes = args
        ver = handle.nvh_version().decode()
        if not os.environ.get("NVH_ALLOW_STALE") and ("nvh-src-hash=" + _build.source_hash()) not in ver:
            raise RuntimeError("loaded %s reports '%s', sources are %s" % (path, ver, _build.source_hash()))
        _lib = handle
    return _lib


def build_id():
    """'nvorbis_hip <version> (gfx950) nvh-src-hash=<hash>' of the library that is actually loaded."""
    return lib().nvh_version().decode()


def check(rc, where):
    if rc != OK:
        raise NvhError(rc, where)
    return rc
