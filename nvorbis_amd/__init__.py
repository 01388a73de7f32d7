"""nvorbis_amd -- MI355X (gfx950) back end for NVorbis' per-packet synthesis path.

Native code lives in csrc/ and is built in-tree into libnvorbis_hip.so (see build.py); this package
is the host-side mirror of the reference's reader surface over that library's C ABI.
"""
from .native import NvhError, lib, lib_path  # noqa: F401
from .reader import Batch, Comm, Context, PacketArray, Stream, StreamDecoder, VorbisReader, demux_ogg, demux_ogg_array, ogg_stream_count  # noqa: F401
