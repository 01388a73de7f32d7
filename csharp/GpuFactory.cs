// GpuFactory.cs -- IFactory (Contracts/IFactory.cs:3-12, Factory.cs:5-59) whose plug-ins run their float work on the GPU.
//
//     new StreamDecoder(packetProvider, new NVorbis.Hip.GpuFactory(device: 0))       // StreamDecoder.cs:50
//
// Shape.  StreamDecoder.LoadBooks (StreamDecoder.cs:226-289) asks the factory for codebooks, floors, residues, mappings and
// modes while it walks the setup packet, and each object's Init consumes its own header bits.  That bit parsing stays
// managed: every Gpu* class holds the reference's own class for Init / Unpack (composition, nothing is re-stated), and
// re-routes only the methods that touch float vectors to the level-1 entry points of libnvorbis_hip.so:
//
//     IMdct.Reverse            -> nvh_mdct_reverse        IFloor.Apply (Floor1 / Floor0) -> nvh_floor1_apply / nvh_floor0_apply
//     IResidue.Decode          -> nvh_residue_decode      IMode.Decode                   -> nvh_mode_decode
//     IMapping.DecodePacket    -> the reference's own orchestration (Mapping.cs:95-198) over the Gpu* floors / residues / mdct
//
// The interfaces hand over managed float[] / float[][]; the arrays are staged through device memory call by call
// (nvh_dev_upload / nvh_dev_download).  This is the unit-parity and drop-in-by-interface path: one P/Invoke and two PCIe
// crossings per interface call.  The throughput path is GpuStreamDecoder (level 2: packets in, PCM out, one crossing per
// look-ahead batch) -- INTEGRATION.md says which to use when.
//
// The native stream needs the three header packets; LoadBooks only ever shows plug-ins the setup packet, so the native
// stream is opened lazily at the first IFloor.Init: that call carries (channels, block0Size, block1Size) -- everything
// of the identification header the synthesis depends on -- and the setup packet itself, which IPacket.Reset() lets it
// read from the start (Contracts/IPacket.cs).
//
// Compiled into the NVorbis assembly (the contracts are internal); needs <AllowUnsafeBlocks>.
using System;
using System.IO;
using System.Reflection;
using NVorbis.Contracts;

namespace NVorbis.Hip
{
    internal sealed unsafe class GpuFactory : IFactory, IDisposable
    {
        internal IntPtr Ctx, Stream;
        internal int Channels, Block0, Block1;
        IntPtr _scratch;      // device planes [channels][block1]
        long _scratchFloats;
        int _floors, _residues, _modes;

        public GpuFactory(int device = 0) { NativeMethods.Check(NativeMethods.nvh_ctx_create(device, out Ctx)); }

        // Factory.cs:7-20: entropy decoding stays on the host, in the reference's own classes
        public IHuffman CreateHuffman() => new Huffman();
        public ICodebook CreateCodebook() => new Codebook();
        public IMdct CreateMdct() => new GpuMdct(this);

        public IFloor CreateFloor(IPacket packet)   // Factory.cs:22-31
        {
            switch ((int)packet.ReadBits(16))
            {
                case 0: return new GpuFloor(this, new Floor0(), _floors++, 0);
                case 1: return new GpuFloor(this, new Floor1(), _floors++, 1);
                default: throw new InvalidDataException("Invalid floor type!");
            }
        }

        public IResidue CreateResidue(IPacket packet)   // Factory.cs:48-58
        {
            switch ((int)packet.ReadBits(16))
            {
                case 0: return new GpuResidue(this, new Residue0(), _residues++);
                case 1: return new GpuResidue(this, new Residue1(), _residues++);
                case 2: return new GpuResidue(this, new Residue2(), _residues++);
                default: throw new InvalidDataException("Invalid residue type!");
            }
        }

        public IMapping CreateMapping(IPacket packet)   // Factory.cs:33-41
        {
            if (packet.ReadBits(16) != 0) throw new InvalidDataException("Invalid mapping type!");
            return new GpuMapping(new Mapping());
        }

        public IMode CreateMode() => new GpuMode(this, new Mode(), _modes++);

        // ---- native stream, opened from what the first IFloor.Init is given ----
        internal void EnsureStream(IPacket setupPacket, int channels, int block0, int block1)
        {
            if (Stream != IntPtr.Zero) return;
            Channels = channels; Block0 = block0; Block1 = block1;
            byte[] setup = PacketBytes(setupPacket, out _);
            byte[] id = IdentificationHeader(channels, block0, block1);
            fixed (byte* pi = id, ps = setup)
                NativeMethods.Check(NativeMethods.nvh_stream_open(Ctx, pi, id.Length, null, 0, ps, setup.Length, out Stream));
        }

        // "\x01vorbis", version 0, channels, sample rate, three bit rates, block size nibbles, framing bit (StreamDecoder.cs:179-204).
        // The synthesis path depends on the channel count and the two block sizes only (Floor0 carries its own rate field).
        static byte[] IdentificationHeader(int channels, int block0, int block1)
        {
            var b = new byte[30];
            b[0] = 1; b[1] = (byte)'v'; b[2] = (byte)'o'; b[3] = (byte)'r'; b[4] = (byte)'b'; b[5] = (byte)'i'; b[6] = (byte)'s';
            b[11] = (byte)channels;
            b[12] = 0x44; b[13] = 0xAC;   // 44100, little endian
            b[28] = (byte)(Utils.ilog(block0) - 1 | (Utils.ilog(block1) - 1) << 4);
            b[29] = 1;
            return b;
        }

        // All bytes of a packet, cursor restored (IPacket.Reset + SkipBits); *bitPos = where the cursor stood.
        internal static byte[] PacketBytes(IPacket packet, out int bitPos)
        {
            bitPos = packet.BitsRead;
            int total = packet.BitsRead + packet.BitsRemaining;
            packet.Reset();
            var buf = new byte[(total + 7) / 8];
            for (int i = 0; i < buf.Length; i++) buf[i] = (byte)packet.ReadBits(Math.Min(8, total - 8 * i));
            packet.Reset();
            packet.SkipBits(bitPos);
            return buf;
        }

        internal IntPtr Scratch(long floats)
        {
            if (floats > _scratchFloats)
            {
                if (_scratch != IntPtr.Zero) NativeMethods.nvh_dev_free(Ctx, _scratch);
                NativeMethods.Check(NativeMethods.nvh_dev_alloc(Ctx, (UIntPtr)(ulong)(floats * 4), out _scratch));
                _scratchFloats = floats;
            }
            return _scratch;
        }

        internal void Upload(IntPtr dst, float[] src, int offset, int count)
        { fixed (float* p = src) NativeMethods.Check(NativeMethods.nvh_dev_upload(Ctx, dst, p + offset, (UIntPtr)(ulong)(count * 4L))); }
        internal void Download(float[] dst, int offset, IntPtr src, int count)
        { fixed (float* p = dst) NativeMethods.Check(NativeMethods.nvh_dev_download(Ctx, p + offset, src, (UIntPtr)(ulong)(count * 4L))); }

        public void Dispose()
        {
            if (_scratch != IntPtr.Zero) { NativeMethods.nvh_dev_free(Ctx, _scratch); _scratch = IntPtr.Zero; }
            if (Stream != IntPtr.Zero) { NativeMethods.nvh_stream_close(Stream); Stream = IntPtr.Zero; }
            if (Ctx != IntPtr.Zero) { NativeMethods.nvh_ctx_destroy(Ctx); Ctx = IntPtr.Zero; }
        }
    }

    // IMdct.Reverse(float[] samples, int sampleCount) (Contracts/IMdct.cs:5, Mdct.cs:13-21): reads [0, n/2), writes [0, n).
    internal sealed class GpuMdct : IMdct
    {
        readonly GpuFactory _f;
        public GpuMdct(GpuFactory f) { _f = f; }
        public void Reverse(float[] samples, int sampleCount)
        {
            IntPtr d = _f.Scratch(sampleCount);
            _f.Upload(d, samples, 0, sampleCount / 2);
            NativeMethods.Check(NativeMethods.nvh_mdct_reverse(_f.Ctx, sampleCount, 1, d, sampleCount));
            _f.Download(samples, 0, d, sampleCount);
        }
    }

    // IFloor (Contracts/IFloor.cs): Init and Unpack are the reference's (Floor1.cs:30-184 / Floor0.cs:28-150); Apply
    // (Floor1.cs:186-341 / Floor0.cs:152-212) runs on the GPU from what Unpack left in the floor data.  The data classes are
    // private to Floor0 / Floor1, so their fields are read by reflection (looked up once).
    internal sealed unsafe class GpuFloor : IFloor
    {
        readonly GpuFactory _f;
        readonly IFloor _inner;
        readonly int _index, _type;
        FieldInfo _posts, _postCount, _coeff, _amp;

        public GpuFloor(GpuFactory f, IFloor inner, int index, int type) { _f = f; _inner = inner; _index = index; _type = type; }

        public void Init(IPacket packet, int channels, int block0Size, int block1Size, ICodebook[] codebooks)
        {
            _f.EnsureStream(packet, channels, block0Size, block1Size);
            _inner.Init(packet, channels, block0Size, block1Size, codebooks);
        }

        public IFloorData Unpack(IPacket packet, int blockSize, int channel) => _inner.Unpack(packet, blockSize, channel);

        public void Apply(IFloorData floorData, int blockSize, float[] residue)
        {
            const BindingFlags any = BindingFlags.Instance | BindingFlags.NonPublic | BindingFlags.Public;
            var t = floorData.GetType();
            int half = blockSize / 2, status = 0;
            IntPtr d = _f.Scratch(half);
            _f.Upload(d, residue, 0, half);
            if (_type == 1)
            {
                if (_posts == null) { _posts = t.GetField("Posts", any); _postCount = t.GetField("PostCount", any); }
                if (_posts == null || _postCount == null) throw new ArgumentException("Incorrect packet data!", nameof(floorData));   // Floor1.cs:188
                var posts = (int[])_posts.GetValue(floorData);
                int count = (int)_postCount.GetValue(floorData);
                var p64 = new int[64];
                Array.Copy(posts, p64, Math.Min(64, posts.Length));
                fixed (int* pp = p64)
                    NativeMethods.Check(NativeMethods.nvh_floor1_apply(_f.Stream, _index, blockSize, 1, pp, &count, d, half, &status));
            }
            else
            {
                if (_coeff == null) { _coeff = t.GetField("Coeff", any); _amp = t.GetField("Amp", any); }
                if (_coeff == null || _amp == null) throw new ArgumentException("Incorrect packet data!", nameof(floorData));     // Floor0.cs:154
                var coeff = (float[])_coeff.GetValue(floorData) ?? new float[1];
                float amp = (float)_amp.GetValue(floorData);
                fixed (float* pc = coeff)
                    NativeMethods.Check(NativeMethods.nvh_floor0_apply(_f.Stream, _index, blockSize, 1, &amp, pc, coeff.Length, d, half, &status));
            }
            NativeMethods.Check(status);   // NVH_ERR_RUNTIME: inverse_dB_table / wMap index out of range in the reference
            _f.Download(residue, 0, d, half);
        }
    }

    // IResidue (Contracts/IResidue.cs:5-6): Init is the reference's (Residue0.cs:35-117, Residue2.cs:10-14); Decode
    // (Residue0.cs:119-201, Residue1.cs:8-26, Residue2.cs:16-47) reads the same bits natively and adds the vectors on the GPU.
    internal sealed unsafe class GpuResidue : IResidue
    {
        readonly GpuFactory _f;
        readonly IResidue _inner;
        readonly int _index;
        public GpuResidue(GpuFactory f, IResidue inner, int index) { _f = f; _inner = inner; _index = index; }

        public void Init(IPacket packet, int channels, ICodebook[] codebooks) => _inner.Init(packet, channels, codebooks);

        public void Decode(IPacket packet, bool[] doNotDecodeChannel, int blockSize, float[][] buffer)
        {
            if (Array.IndexOf(doNotDecodeChannel, false) == -1) return;   // Residue0.cs:125: nothing is read
            int channels = buffer.Length, b1 = _f.Block1;
            byte[] bytes = GpuFactory.PacketBytes(packet, out int bitPos);
            IntPtr d = _f.Scratch((long)channels * b1);
            for (int c = 0; c < channels; c++) _f.Upload(d + c * b1 * 4, buffer[c], 0, Math.Min(b1, buffer[c].Length));
            int consumed;
            fixed (byte* pb = bytes)
                NativeMethods.Check(NativeMethods.nvh_residue_decode(_f.Stream, _index, pb, bytes.Length, bitPos, 1, blockSize, d, out consumed));
            packet.SkipBits(consumed);
            for (int c = 0; c < channels; c++) _f.Download(buffer[c], 0, d + c * b1 * 4, Math.Min(b1, buffer[c].Length));
        }
    }

    // IMapping (Contracts/IMapping.cs:5-7): the reference's Mapping, unchanged, over the Gpu* floors, residues and mdct it is
    // initialised with -- its DecodePacket (Mapping.cs:95-198) is orchestration plus the inverse coupling loop; every other
    // float-touching step it calls is one of the classes above.  (GpuMode below does the whole packet in one native call
    // and never reaches this object; it exists so that a mapping taken on its own behaves like the reference's.)
    internal sealed class GpuMapping : IMapping
    {
        readonly Mapping _inner;
        public GpuMapping(Mapping inner) { _inner = inner; }
        public void Init(IPacket packet, int channels, IFloor[] floors, IResidue[] residues, IMdct mdct) => _inner.Init(packet, channels, floors, residues, mdct);
        public void DecodePacket(IPacket packet, int blockSize, int channels, float[][] buffer) => _inner.DecodePacket(packet, blockSize, channels, buffer);
    }

    // IMode (Contracts/IMode.cs:5-9): Init / GetPacketSampleCount are the reference's (Mode.cs:24-67, 172-176); Decode
    // (Mode.cs:153-170: GetPacketInfo, Mapping.DecodePacket, window loop) is ONE native call for the whole packet.
    internal sealed unsafe class GpuMode : IMode
    {
        readonly GpuFactory _f;
        readonly Mode _inner;
        readonly int _index;
        public GpuMode(GpuFactory f, Mode inner, int index) { _f = f; _inner = inner; _index = index; }

        public void Init(IPacket packet, int channels, int block0Size, int block1Size, IMapping[] mappings) => _inner.Init(packet, channels, block0Size, block1Size, mappings);
        public int GetPacketSampleCount(IPacket packet) => _inner.GetPacketSampleCount(packet);

        public bool Decode(IPacket packet, float[][] buffer, out int packetStartindex, out int packetValidLength, out int packetTotalLength)
        {
            // the native parser reads the packet from its first bit (packet type, mode number, window flags, floors, residues)
            byte[] bytes = GpuFactory.PacketBytes(packet, out _);
            int channels = buffer.Length, b1 = _f.Block1;
            IntPtr d = _f.Scratch((long)channels * b1);
            int decoded, blockSize;
            fixed (byte* pb = bytes)
                NativeMethods.Check(NativeMethods.nvh_mode_decode(_f.Stream, pb, bytes.Length, d, out decoded, out blockSize,
                                                                 out packetStartindex, out packetValidLength, out packetTotalLength));
            if (decoded == 0) { packetStartindex = packetValidLength = packetTotalLength = 0; return false; }   // Mode.cs:121-128
            for (int c = 0; c < channels; c++) _f.Download(buffer[c], 0, d + c * b1 * 4, blockSize);
            return true;
        }
    }
}
