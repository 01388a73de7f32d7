// NativeMethods.cs -- P/Invoke declarations for libnvorbis_hip.so (include/nvorbis_hip.h).
// Source-only in this repository: the build image has no .NET toolchain.  The Python ctypes binding
// (nvorbis_amd/native.py) declares exactly the same entry points and is what the test suite drives.
// These files are compiled INTO the NVorbis assembly (csharp/README.md): IFactory, the plug-in interfaces, TagData and
// StreamStats are `internal` there.
using System;
using System.Runtime.InteropServices;

namespace NVorbis.Hip
{
    internal static class NativeMethods
    {
        const string Lib = "nvorbis_hip"; // resolves libnvorbis_hip.so

        public const int NVH_OK = 0;
        public const int NVH_ERR_INVALID_DATA = -1, NVH_ERR_ARGUMENT = -2, NVH_ERR_RUNTIME = -3, NVH_ERR_NOMEM = -4;
        public const int NVH_ERR_NOT_VORBIS = -5, NVH_ERR_DEVICE = -6, NVH_ERR_UNSUPPORTED = -7, NVH_ERR_NO_GPU = -8;
        public const int NVH_PKT_EOS = 1, NVH_PKT_RESYNC = 2;

        [DllImport(Lib)] public static extern int nvh_device_count();
        [DllImport(Lib)] public static extern int nvh_last_hip_error();
        [DllImport(Lib)] public static extern int nvh_ctx_create(int device, out IntPtr ctx);
        [DllImport(Lib)] public static extern void nvh_ctx_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern int nvh_ctx_synchronize(IntPtr ctx);
        [DllImport(Lib)] public static extern int nvh_ctx_set_parse_lanes(IntPtr ctx, int lanes);

        [DllImport(Lib)] public static extern int nvh_mdct_reverse(IntPtr ctx, int n, int batch, IntPtr dBuf, long stride);
        [DllImport(Lib)] public static extern int nvh_calc_window(int prevBlock, int block, int nextBlock, [Out] float[] window);
        [DllImport(Lib)] public static extern int nvh_calc_overlap(int prevBlock, int block, int nextBlock, out int start, out int valid, out int total);

        [DllImport(Lib)] public static extern unsafe int nvh_stream_open(IntPtr ctx, byte* id, int idLen, byte* comment, int commentLen,
                                                                       byte* setup, int setupLen, out IntPtr stream);
        [DllImport(Lib)] public static extern void nvh_stream_close(IntPtr stream);
        [DllImport(Lib)] public static extern int nvh_stream_info(IntPtr stream, out int channels, out int sampleRate, out int block0, out int block1);
        [DllImport(Lib)] public static extern int nvh_stream_set_clip(IntPtr stream, int on);
        /// <summary>Parse audio packets on the GPU (kernels_parse.hip); NVH_ERR_UNSUPPORTED (-7) for stream shapes outside its limits.</summary>
        [DllImport(Lib)] public static extern int nvh_stream_set_gpu_parse(IntPtr stream, int on);
        /// <summary>Page-locked host memory: a pinned pcmHost is written by the copy engine directly.</summary>
        [DllImport(Lib)] public static extern int nvh_pinned_alloc(UIntPtr bytes, out IntPtr p);
        [DllImport(Lib)] public static extern void nvh_pinned_free(IntPtr p);
        [DllImport(Lib)] public static extern int nvh_stream_has_clipped(IntPtr stream, out int clipped);
        [DllImport(Lib)] public static extern int nvh_stream_position(IntPtr stream, out long position, out long emitted, out int eos);
        [DllImport(Lib)] public static extern unsafe int nvh_stream_push_packet(IntPtr stream, byte* data, int len, long granule, int flags);
        [DllImport(Lib)] public static extern unsafe int nvh_stream_push_packets(IntPtr stream, byte* bytes, long* offsets, long* granules, byte* flags, int n, int maxPackets, out int consumed);
        [DllImport(Lib)] public static extern int nvh_stream_push_end(IntPtr stream);
        /// <summary>IMode.Decode of one packet: windowed block [channels][block1] before overlap, into device memory.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_mode_decode(IntPtr stream, byte* packet, int len, IntPtr dBlock, out int decoded, out int blockSize, out int start, out int valid, out int total);
        /// <summary>One inverse coupling step (Mapping.cs:150-178) over two device vectors.</summary>
        [DllImport(Lib)] public static extern int nvh_inverse_couple(IntPtr ctx, IntPtr dMagnitude, IntPtr dAngle, int count);
        /// <summary>IFloor.Apply for a Floor1 (Floor1.cs:186-341) on a batch of device vectors; posts is [batch][64] raw Unpack values.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_floor1_apply(IntPtr stream, int floorIndex, int blockSize, int batch, int* posts, int* postCounts, IntPtr dResidue, long stride, int* status);
        /// <summary>IFloor.Apply for a Floor0 (Floor0.cs:152-212) from Data.Amp / Data.Coeff, batched.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_floor0_apply(IntPtr stream, int floorIndex, int blockSize, int batch, float* amps, float* coeffs, int coeffStride, IntPtr dResidue, long stride, int* status);
        /// <summary>IResidue.Decode (Residue0.cs:119-201) from a packet cursor into device planes [channels][block1].</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_residue_decode(IntPtr stream, int residueIndex, byte* packet, int len, int bitOffset, int anyChannelDecodes, int blockSize, IntPtr dBuffer, out int bitsConsumed);
        /// <summary>Mode.Decode's window loop (Mode.cs:160-166).</summary>
        [DllImport(Lib)] public static extern int nvh_window_apply(IntPtr stream, int modeIndex, int prevFlag, int nextFlag, int batch, IntPtr dBuf, long stride);
        /// <summary>StreamDecoder.OverlapBuffers (StreamDecoder.cs:532-541).</summary>
        [DllImport(Lib)] public static extern int nvh_overlap_buffers(IntPtr ctx, IntPtr dPrevious, IntPtr dNext, int prevStart, int prevStop, int nextStart, int channels, long planeStride);
        /// <summary>ClippingCopyBuffer / CopyBuffer (StreamDecoder.cs:391-415).</summary>
        [DllImport(Lib)] public static extern int nvh_copy_buffer(IntPtr ctx, IntPtr dPlanes, int start, int count, int channels, long planeStride, IntPtr dTarget, int clip, out int clipped);
        /// <summary>_hasPosition / _currentPosition (StreamDecoder.cs:35-39); set when a decoder starts mid-stream.</summary>
        [DllImport(Lib)] public static extern int nvh_stream_position_state(IntPtr stream, out int hasPosition, out long position);
        [DllImport(Lib)] public static extern int nvh_stream_set_position_state(IntPtr stream, int hasPosition, long position);
        [DllImport(Lib)] public static extern int nvh_stream_drop_pending(IntPtr stream);
        /// <summary>StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647) from the first bytes of a packet.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_stream_packet_sample_count(IntPtr stream, byte* packet, int len, int isResync, out int count);
        /// <summary>ResetDecoder (StreamDecoder.cs:295-305).</summary>
        [DllImport(Lib)] public static extern int nvh_stream_reset(IntPtr stream);
        /// <summary>Geometry-only index of a run of audio packets (positions, emitted samples, decodable / lead-in flags).</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_stream_index_packets(IntPtr stream, byte* bytes, long* offsets, long* granules, byte* flags, int n, long* positionAfter, long* emittedAfter, byte* stateAfter, out long totalEmitted);
        [DllImport(Lib)] public static extern int nvh_stream_mode_info(IntPtr stream, int modeIndex, out int blockFlag, out int blockSize, out int mapping);
        [DllImport(Lib)] public static extern int nvh_stream_floor_info(IntPtr stream, int floorIndex, out int type, out int postCount, out int range);
        [DllImport(Lib)] public static extern int nvh_stream_pending(IntPtr stream, out int frames, out long samplesPerChannel);
        /// <summary>[frames][8] ints: block size (0 = drained tail), start, valid, total, ... of every pending frame.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_stream_pending_geometry(IntPtr stream, int* geometry, int capFrames);
        /// <summary>Page table of one logical Ogg stream + IPacketProvider.SeekTo over it (Ogg/PacketProvider.cs:56-295), for hosts
        /// that do not bring NVorbis' own container code.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_ogg_index_open(byte* bytes, UIntPtr len, int streamIndex, out IntPtr index);
        [DllImport(Lib)] public static extern void nvh_ogg_index_close(IntPtr index);
        [DllImport(Lib)] public static extern int nvh_ogg_index_info(IntPtr index, out int pages, out int packets, out int firstDataPage, out long maxGranule, out int hasAllPages);
        [DllImport(Lib)] public static extern int nvh_ogg_index_page(IntPtr index, int page, out long granule, out int flags, out int packetCount, out int firstPacket);
        [DllImport(Lib)] public static extern int nvh_ogg_seek(IntPtr index, IntPtr stream, long granulePos, int preRoll, out long packetIndex, out long granuleOut);
        /// <summary>IStreamDecoder.UpperBitrate / NominalBitrate / LowerBitrate (StreamDecoder.cs:191-199).</summary>
        [DllImport(Lib)] public static extern int nvh_stream_bitrates(IntPtr stream, out int upper, out int nominal, out int lower);
        /// <summary>Packets of the last synthesised batch that made the parser fail (GPU-parse mode), with their positions in its PCM.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_stream_parse_errors(IntPtr stream, int* codes, long* samplesBefore, int cap, out int count);
        /// <summary>Device memory for the managed float[] contract of the plug-in interfaces (GpuFactory.cs).</summary>
        [DllImport(Lib)] public static extern int nvh_dev_alloc(IntPtr ctx, UIntPtr bytes, out IntPtr dPtr);
        [DllImport(Lib)] public static extern void nvh_dev_free(IntPtr ctx, IntPtr dPtr);
        [DllImport(Lib)] public static extern unsafe int nvh_dev_upload(IntPtr ctx, IntPtr dDst, void* hSrc, UIntPtr bytes);
        [DllImport(Lib)] public static extern unsafe int nvh_dev_download(IntPtr ctx, void* hDst, IntPtr dSrc, UIntPtr bytes);
        [DllImport(Lib)] public static extern int nvh_measure_copy(IntPtr ctx, IntPtr dSrc, IntPtr dDst, UIntPtr bytes, int iters, out float ms);
        [DllImport(Lib)] public static extern unsafe int nvh_stream_synth(IntPtr stream, float* pcmHost, IntPtr dPcm, long capacity, out long written);
        /// <summary>Pipelined form (pinned destination): the transfer of one batch overlaps the pushes, parse and kernels of the next.</summary>
        [DllImport(Lib)] public static extern unsafe int nvh_stream_synth_begin(IntPtr stream, float* pcmHost, long capacity, out long expected);
        [DllImport(Lib)] public static extern int nvh_stream_synth_end(IntPtr stream, out long written);

        /// <summary>The corpus gather (include/nvorbis_hip.h, "multi-GPU"): RCCL over xGMI through the library, one process per GPU (GpuCorpusGather.cs).</summary>
        public const int NVH_COMM_ID_BYTES = 128;
        public const int NVH_GATHER_SELF_P2P = 1;
        [DllImport(Lib)] public static extern unsafe int nvh_comm_unique_id(byte* id);
        [DllImport(Lib)] public static extern unsafe int nvh_comm_create(IntPtr ctx, byte* id, int rank, int world, out IntPtr comm);
        [DllImport(Lib)] public static extern void nvh_comm_destroy(IntPtr comm);
        [DllImport(Lib)] public static extern int nvh_comm_info(IntPtr comm, out int rank, out int world);
        [DllImport(Lib)] public static extern unsafe int nvh_comm_allgather_i64(IntPtr comm, long* mine, int n, long* all);
        [DllImport(Lib)] public static extern unsafe int nvh_comm_gather_pcm(IntPtr comm, IntPtr dSend, long sendCount, IntPtr dRecv, long* counts, int root, int flags);

        internal static void Check(int rc)
        {
            switch (rc)
            {
                case NVH_OK: return;
                case NVH_ERR_INVALID_DATA:
                case NVH_ERR_NOT_VORBIS: throw new System.IO.InvalidDataException("nvorbis_hip: invalid Vorbis data (" + rc + ")");
                case NVH_ERR_ARGUMENT: throw new ArgumentOutOfRangeException("nvorbis_hip argument");
                case NVH_ERR_RUNTIME: throw new IndexOutOfRangeException("nvorbis_hip: the managed decoder would have faulted here");
                case NVH_ERR_NOMEM: throw new OutOfMemoryException();
                case NVH_ERR_UNSUPPORTED: throw new NotSupportedException("nvorbis_hip: stream outside documented limits");
                case NVH_ERR_NO_GPU: throw new PlatformNotSupportedException("nvorbis_hip: no HIP device (there is no CPU fallback)");
                default: throw new InvalidOperationException("nvorbis_hip: HIP error " + nvh_last_hip_error());
            }
        }
    }
}
