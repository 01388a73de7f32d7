"""nvorbis_amd -- MI355X (gfx950) back end for NVorbis' per-packet synthesis path.

Native code lives in csrc/ and is built in-tree into libnvorbis_hip.so (see build.py); this package
is the host-side mirror of the reference's reader surface over that library's C ABI.
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that
# share a queue run one after the other: a corpus worker pool (one context = one stream per thread) then has four kernels in
# flight however many workers it starts.  Read once when the runtime initialises (the first HIP call of the process), so it is
# set here, at import, unless the environment already says otherwise.  A host in another language sets it in its environment.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .native import NvhError, lib, lib_path  # noqa: F401
from .reader import Batch, Comm, Context, PacketArray, Stream, StreamDecoder, VorbisReader, demux_ogg, demux_ogg_array, ogg_stream_count  # noqa: F401
