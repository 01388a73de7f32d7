// GpuStreamDecoder.cs -- IStreamDecoder over libnvorbis_hip.so.
//
// Shape: NVorbis.StreamDecoder (NVorbis/StreamDecoder.cs) with the synthesis plug-ins (IMode -> IMapping ->
// IFloor / IResidue / IMdct, created by IFactory) replaced by one native stream object.  The reference's own
// container layer keeps feeding it: packets come from Contracts.IPacketProvider exactly as before.
// Plug it in through the seam the reference already has (VorbisReader.cs:15):
//
//     VorbisReader.CreateStreamDecoder = pp => new NVorbis.Hip.GpuStreamDecoder(pp, device: 0);
//
// CreateStreamDecoder is `internal static` (VorbisReader.cs:14-15) and Tags / Stats are built from the reference's internal
// TagData / StreamStats classes, so this file is compiled INTO the NVorbis assembly (csharp/README.md), next to
// GpuFactory.cs (the interface-by-interface path) and NativeMethods.cs.
// Source-only here (no .NET toolchain in the build image); the same call sequence is exercised by the
// Python mirror nvorbis_amd/reader.py::StreamDecoder against the identical C ABI.
using System;
using System.Collections.Generic;
using System.Text;
using NVorbis.Contracts;

namespace NVorbis.Hip
{
    public sealed unsafe class GpuStreamDecoder : IStreamDecoder
    {
        readonly Contracts.IPacketProvider _packetProvider;
        IntPtr _ctx, _stream;
        int _channels, _sampleRate, _block0, _block1;
        readonly int _batchPackets;
        float[] _ring = Array.Empty<float>();
        int _ringPos, _ringLen;
        bool _ended, _clip = true;
        long _skip;   // floats to drop in front of the next samples: SeekTo's roll-forward
        int _upperBitrate, _nominalBitrate, _lowerBitrate;
        readonly string _vendor;
        readonly string[] _comments;
        ITagData _tags;
        readonly StreamStats _stats = new StreamStats();
        // exceptions the reference would throw from inside Read, queued at the ring position they belong to
        readonly Queue<KeyValuePair<Exception, int>> _pendingErrors = new Queue<KeyValuePair<Exception, int>>();

        /// <param name="poolParseLanes">0 for a lone decoder.  A host that runs many decoders at once (one per worker thread: a
        /// corpus transcoder) passes 32 -- nvh_ctx_set_parse_lanes: the GPU parser then puts up to 32 packets on a wavefront, every
        /// lane walking its own, so that the parses of all workers fit the chip side by side -- and starts its process with GPU_MAX_HW_QUEUES=16 in the
        /// environment (INTEGRATION.md).  The PCM does not depend on it.</param>
        public GpuStreamDecoder(Contracts.IPacketProvider packetProvider, int device = 0, int batchPackets = 1024, int poolParseLanes = 0)
        {
            _packetProvider = packetProvider ?? throw new ArgumentNullException(nameof(packetProvider));
            _batchPackets = batchPackets;
            NativeMethods.Check(NativeMethods.nvh_ctx_create(device, out _ctx));
            if (poolParseLanes != 0) NativeMethods.Check(NativeMethods.nvh_ctx_set_parse_lanes(_ctx, poolParseLanes));
            // ProcessHeaderPackets (StreamDecoder.cs:107-127): id, comment, setup
            byte[] id = ReadAll(_packetProvider.GetNextPacket());
            byte[] comment = ReadAll(_packetProvider.GetNextPacket());
            byte[] setup = ReadAll(_packetProvider.GetNextPacket());
            fixed (byte* pi = id, pc = comment, ps = setup)
                NativeMethods.Check(NativeMethods.nvh_stream_open(_ctx, pi, id.Length, pc, comment.Length, ps, setup.Length, out _stream));
            NativeMethods.Check(NativeMethods.nvh_stream_info(_stream, out _channels, out _sampleRate, out _block0, out _block1));
            NativeMethods.Check(NativeMethods.nvh_stream_bitrates(_stream, out _upperBitrate, out _nominalBitrate, out _lowerBitrate));
            ParseComments(comment, out _vendor, out _comments);                 // LoadComments (StreamDecoder.cs:206-224)
            _stats.SetSampleRate(_sampleRate);                                   // StreamDecoder.cs:200
            _stats.AddPacket(-1, id.Length * 8, 0, 0);                           // header packets (:201, :221, :286)
            _stats.AddPacket(-1, comment.Length * 8, 0, 0);
            _stats.AddPacket(-1, setup.Length * 8, 0, 0);
            // Parse the packets on the GPU as well when the stream shape allows it (-7 = outside the GPU parser's limits:
            // the host parser stays in charge).  Same PCM either way.
            int rc = NativeMethods.nvh_stream_set_gpu_parse(_stream, 1);
            if (rc != 0 && rc != -7) NativeMethods.Check(rc);
        }

        static byte[] ReadAll(IPacket packet)
        {
            if (packet == null) throw new System.IO.InvalidDataException("missing Vorbis header packet");
            var buf = new byte[(packet.BitsRemaining + 7) / 8];
            int n = packet.Read(buf, 0, buf.Length);   // Extensions.cs:19
            packet.Done();
            if (n != buf.Length) Array.Resize(ref buf, n);
            return buf;
        }

        // "\x03vorbis", vendor string, comment count, comments: 32-bit little-endian lengths, UTF-8 (StreamDecoder.cs:163-177, 206-224)
        static void ParseComments(byte[] p, out string vendor, out string[] comments)
        {
            int pos = 7;
            string ReadString()
            {
                if (pos + 4 > p.Length) throw new System.IO.InvalidDataException("Could not read full string!");
                int len = BitConverter.ToInt32(p, pos); pos += 4;
                if (len == 0) return string.Empty;
                if (len < 0 || pos + len > p.Length) throw new System.IO.InvalidDataException("Could not read full string!");
                var str = Encoding.UTF8.GetString(p, pos, len); pos += len;
                return str;
            }
            vendor = ReadString();
            if (pos + 4 > p.Length) throw new System.IO.InvalidDataException("Could not read full string!");
            int n = BitConverter.ToInt32(p, pos); pos += 4;
            // every comment needs at least its 4-byte length: an untrusted count beyond that (or negative) is a broken header,
            // not an OverflowException / a gigabyte allocation
            if (n < 0 || (long)n * 4 > p.Length - pos) throw new System.IO.InvalidDataException("Could not read full string!");
            comments = new string[n];
            for (int i = 0; i < n; i++) comments[i] = ReadString();
        }

        public int Channels => _channels;
        public int SampleRate => _sampleRate;
        public int UpperBitrate => _upperBitrate;
        public int NominalBitrate => _nominalBitrate;
        public int LowerBitrate => _lowerBitrate;
        public ITagData Tags => _tags ?? (_tags = new TagData(_vendor, _comments));      // StreamDecoder.cs:690
        // The reference counts, per audio packet, the samples it yielded and the bits the decode read / left over
        // (StreamStats.cs:94-121).  Packets are parsed a batch ahead here and the native parser does not report how far
        // into each packet it read, so a packet is booked with its nominal sample count and all of its bits as read.
        public IStreamStats Stats => _stats;
        public bool ClipSamples { get => _clip; set { _clip = value; NativeMethods.Check(NativeMethods.nvh_stream_set_clip(_stream, value ? 1 : 0)); } }
        public bool HasClipped { get { NativeMethods.Check(NativeMethods.nvh_stream_has_clipped(_stream, out int c)); return c != 0; } }
        public bool IsEndOfStream => _ended && _ringPos >= _ringLen;
        public long SamplePosition
        {
            get
            {
                NativeMethods.Check(NativeMethods.nvh_stream_position(_stream, out long pos, out _, out _));
                NativeMethods.Check(NativeMethods.nvh_stream_pending(_stream, out _, out long pending));   // pushed, not yet synthesised (right after a seek)
                return pos - pending - (_ringLen - _ringPos) / _channels + _skip / _channels;
            }
            set => SeekTo(value);
        }
        public TimeSpan TimePosition { get => TimeSpan.FromSeconds((double)SamplePosition / _sampleRate); set => SeekTo(value); }
        public long TotalSamples => _packetProvider.GetGranuleCount();                       // StreamDecoder.cs:700
        public TimeSpan TotalTime => TimeSpan.FromSeconds((double)TotalSamples / _sampleRate);

        // Parse up to _batchPackets packets ahead on the host, synthesise them on the GPU in one go.
        bool Refill()
        {
            while (!_ended)
            {
                Exception pushError = null;
                for (int pushed = 0; pushed < _batchPackets; pushed++)
                {
                    NativeMethods.Check(NativeMethods.nvh_stream_position(_stream, out _, out _, out int eos));
                    if (eos != 0) { _ended = true; break; }                    // _eosFound (StreamDecoder.cs:343-350)
                    var packet = _packetProvider.GetNextPacket();
                    if (packet == null) { NativeMethods.Check(NativeMethods.nvh_stream_push_end(_stream)); _ended = true; break; }
                    int flags = (packet.IsEndOfStream ? NativeMethods.NVH_PKT_EOS : 0) | (packet.IsResync ? NativeMethods.NVH_PKT_RESYNC : 0);
                    long granule = packet.GranulePosition ?? -1;
                    int overhead = packet.ContainerOverheadBits;
                    byte[] data = ReadAll(packet);
                    int rc, nominal = 0;
                    fixed (byte* p = data)
                    {
                        NativeMethods.nvh_stream_packet_sample_count(_stream, p, Math.Min(data.Length, 8), packet.IsResync ? 1 : 0, out nominal);
                        rc = NativeMethods.nvh_stream_push_packet(_stream, p, data.Length, granule, flags);
                    }
                    _stats.AddPacket(nominal, data.Length * 8, 0, overhead);
                    if (rc != 0)
                    {
                        // host-parse mode: this packet makes the managed decoder throw.  It is consumed (packet.Done() in the
                        // reference's finally block); what was parsed before it is synthesised and read first.
                        pushError = ToException(rc);
                        break;
                    }
                }
                NativeMethods.Check(NativeMethods.nvh_stream_pending(_stream, out int frames, out long samples));
                long written = 0;
                if (frames != 0)
                {
                    long need = samples * _channels;
                    if (_ring.Length < need) _ring = new float[need];
                    int rc;
                    fixed (float* dst = _ring)
                        rc = NativeMethods.nvh_stream_synth(_stream, dst, IntPtr.Zero, _ring.Length, out written);
                    _ringPos = 0; _ringLen = (int)written;
                    if (rc != 0)
                    {
                        // GPU-parse mode: packets inside the batch failed; the PCM of all others is complete
                        NativeMethods.Check(NativeMethods.nvh_stream_parse_errors(_stream, null, null, 0, out int n));
                        if (n <= 0) NativeMethods.Check(rc);
                        var codes = new int[n]; var before = new long[n];
                        fixed (int* pc = codes) fixed (long* pb = before)
                            NativeMethods.Check(NativeMethods.nvh_stream_parse_errors(_stream, pc, pb, n, out n));
                        for (int i = 0; i < n; i++)
                            _pendingErrors.Enqueue(new KeyValuePair<Exception, int>(ToException(codes[i]), (int)Math.Min(before[i] * _channels, written)));
                    }
                }
                if (pushError != null) _pendingErrors.Enqueue(new KeyValuePair<Exception, int>(pushError, (int)written));
                if (written > 0) return true;
                if (_pendingErrors.Count > 0) throw _pendingErrors.Dequeue().Key;
            }
            return false;
        }

        static Exception ToException(int rc)
        {
            try { NativeMethods.Check(rc); } catch (Exception e) { return e; }
            return new InvalidOperationException("nvorbis_hip: unexpected success code");
        }

        // StreamDecoder.Read (StreamDecoder.cs:320-389): same argument checks, partial reads, 0 at end of stream.
        public int Read(Span<float> buffer, int offset, int count)
        {
            if (offset < 0 || offset + count > buffer.Length) throw new ArgumentOutOfRangeException(nameof(offset));
            if (count % _channels != 0) throw new ArgumentOutOfRangeException(nameof(count), "Must be a multiple of Channels!");
            if (_stream == IntPtr.Zero) throw new ObjectDisposedException(nameof(GpuStreamDecoder));
            int idx = offset, tgt = offset + count;
            while (idx < tgt)
            {
                // an exception of the packet that follows this ring position: the samples before it have been delivered
                if (_pendingErrors.Count > 0 && _ringPos >= _pendingErrors.Peek().Value) throw _pendingErrors.Dequeue().Key;
                if (_ringPos >= _ringLen && !Refill()) break;
                if (_skip > 0) { int drop = (int)Math.Min(_skip, _ringLen - _ringPos); _ringPos += drop; _skip -= drop; continue; }
                int take = Math.Min(tgt - idx, _ringLen - _ringPos);
                if (_pendingErrors.Count > 0) take = Math.Min(take, _pendingErrors.Peek().Value - _ringPos);
                new Span<float>(_ring, _ringPos, take).CopyTo(buffer.Slice(idx, take));
                _ringPos += take; idx += take;
            }
            return idx - offset;
        }

        public void Dispose()
        {
            if (_stream != IntPtr.Zero) { NativeMethods.nvh_stream_close(_stream); _stream = IntPtr.Zero; }
            if (_ctx != IntPtr.Zero) { NativeMethods.nvh_ctx_destroy(_ctx); _ctx = IntPtr.Zero; }
        }

        // StreamDecoder.SeekTo (StreamDecoder.cs:552-628).  The page / packet search stays the reference's
        // (IPacketProvider.SeekTo, Ogg/PacketProvider.cs:56-260, fed by GetPacketGranules below; nvh_ogg_seek is the same
        // search for hosts without the managed container code); the decoder side is ResetDecoder + pre-roll packet + the packet
        // that holds the target, and the roll-forward is taken off the front of the next samples.
        public void SeekTo(TimeSpan timePosition, System.IO.SeekOrigin seekOrigin = System.IO.SeekOrigin.Begin)
            => SeekTo((long)(SampleRate * timePosition.TotalSeconds), seekOrigin);

        public void SeekTo(long samplePosition, System.IO.SeekOrigin seekOrigin = System.IO.SeekOrigin.Begin)
        {
            if (_stream == IntPtr.Zero) throw new ObjectDisposedException(nameof(GpuStreamDecoder));
            if (!_packetProvider.CanSeek) throw new InvalidOperationException("Seek is not supported by the Contracts.IPacketProvider instance.");
            switch (seekOrigin)
            {
                case System.IO.SeekOrigin.Begin: break;
                case System.IO.SeekOrigin.Current: samplePosition = SamplePosition - samplePosition; break;   // sic (:573)
                case System.IO.SeekOrigin.End: samplePosition = TotalSamples - samplePosition; break;
                default: throw new ArgumentOutOfRangeException(nameof(seekOrigin));
            }
            if (samplePosition < 0) throw new ArgumentOutOfRangeException(nameof(samplePosition));

            int rollForward;
            if (samplePosition == 0) { _packetProvider.SeekTo(0, 0, GetPacketGranules); rollForward = 0; }
            else { var pos = _packetProvider.SeekTo(samplePosition, 1, GetPacketGranules); rollForward = (int)(samplePosition - pos); }

            long oldPosition = SamplePosition;                                      // ResetDecoder leaves _currentPosition alone (:294-305)
            NativeMethods.Check(NativeMethods.nvh_stream_reset(_stream));          // ResetDecoder (:599)
            _ringPos = _ringLen = 0; _ended = false; _skip = 0;
            _pendingErrors.Clear();
            NativeMethods.Check(NativeMethods.nvh_stream_set_position_state(_stream, 1, oldPosition));   // _hasPosition = true (:600)
            // the pre-roll packet (:603-614): a first packet, emits nothing, only provides the overlap
            if (!PushOne(out _))
            {
                _ended = true;
                NativeMethods.Check(NativeMethods.nvh_stream_drop_pending(_stream));
                if (_packetProvider.GetGranuleCount() != samplePosition)
                    throw new InvalidOperationException("Could not read pre-roll packet!  Try seeking again prior to reading more samples.");
                NativeMethods.Check(NativeMethods.nvh_stream_set_position_state(_stream, 1, samplePosition));
                return;
            }
            // the packet that holds the target (:616-621)
            if (!PushOne(out long count))
            {
                NativeMethods.Check(NativeMethods.nvh_stream_reset(_stream));
                _ended = true;
                throw new InvalidOperationException("Could not read pre-roll packet!  Try seeking again prior to reading more samples.");
            }
            if (rollForward > count || rollForward < 0)
            {
                // _prevPacketStart += rollForward past _prevPacketEnd: the reference's Read never returns (copyLen < 0 with
                // start != end, :341-377).  Possible on the first data page, where GetIsVorbisBugDiff takes the first packet's
                // nominal length for the encoder bug (Ogg/PacketProvider.cs:150-172, 224-260).
                NativeMethods.Check(NativeMethods.nvh_stream_reset(_stream));
                _ended = true;
                throw new IndexOutOfRangeException("nvorbis_hip: the managed decoder does not return from the Read that follows this seek");
            }
            // _prevPacketStart += rollForward; _currentPosition = samplePosition (:624-626): the pending frame's `count` samples are
            // ahead of the position, Read drops rollForward of them
            _skip = (long)rollForward * _channels;
            NativeMethods.Check(NativeMethods.nvh_stream_set_position_state(_stream, 1, samplePosition - rollForward + count));
        }

        // ReadNextPacket for the provider's next packet: true when it decoded; `emitted` = samples per channel it adds
        bool PushOne(out long emitted)
        {
            emitted = 0;
            var packet = _packetProvider.GetNextPacket();
            if (packet == null) { NativeMethods.Check(NativeMethods.nvh_stream_push_end(_stream)); return false; }
            int flags = (packet.IsEndOfStream ? NativeMethods.NVH_PKT_EOS : 0) | (packet.IsResync ? NativeMethods.NVH_PKT_RESYNC : 0);
            long granule = packet.GranulePosition ?? -1;
            byte[] data = ReadAll(packet);
            NativeMethods.Check(NativeMethods.nvh_stream_pending(_stream, out int f0, out long s0));
            fixed (byte* p = data)
                NativeMethods.Check(NativeMethods.nvh_stream_push_packet(_stream, p, data.Length, granule, flags));
            NativeMethods.Check(NativeMethods.nvh_stream_pending(_stream, out int f1, out long s1));
            emitted = s1 - s0;
            if (f1 <= f0) return false;
            // a rejected packet drains the previous block into a pseudo-frame of block size 0
            var geo = new int[8 * f1];
            fixed (int* g = geo) NativeMethods.Check(NativeMethods.nvh_stream_pending_geometry(_stream, g, f1));
            return geo[8 * (f1 - 1)] != 0;
        }

        // StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647), computed natively from the packet's first bits
        int GetPacketGranules(IPacket packet)
        {
            if (packet.IsResync) return 0;
            var head = new byte[8];
            int n = packet.Read(head, 0, head.Length);
            fixed (byte* p = head)
                NativeMethods.Check(NativeMethods.nvh_stream_packet_sample_count(_stream, p, n, 0, out int count));
            return count;
        }
    }
}
