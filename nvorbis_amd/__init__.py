"""nvorbis_amd -- MI355X (gfx950) back end for NVorbis' per-packet synthesis path.

Native code lives in csrc/ and is built in-tree into libnvorbis_hip.so (see build.py); this package
is the host-side mirror of the reference's reader surface over that library's C ABI.
"""
import os as _os


def configure_process(hw_queues=16):
    """Process-wide HIP runtime setting for a corpus worker pool, to be called BEFORE the first HIP call of the process (it is read
    once when the runtime initialises; later calls change nothing): the runtime multiplexes a process's streams onto
    GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels of streams that share a queue run one after the other, so a pool
    of 16 contexts (one HIP stream each) has four kernels in flight however many workers it starts.  It also changes the queueing of
    every other HIP user in the process (torch), therefore opt-in: bench.py and the corpus tools call it (or set the variable
    themselves); a host in another language sets it in its environment.  An existing value wins.  Returns the value in force."""
    return _os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(hw_queues)))


if _os.environ.get("NVH_HW_QUEUES"):  # the same through the environment: NVH_HW_QUEUES=16 python your_script.py
    configure_process(int(_os.environ["NVH_HW_QUEUES"]))

from .native import NvhError, lib, lib_path  # noqa: F401
from .reader import Batch, Comm, Context, PacketArray, Stream, StreamDecoder, VorbisReader, demux_ogg, demux_ogg_array, ogg_stream_count  # noqa: F401
