// GpuCorpusGather.cs -- the one collective of the path for the C# host: the PCM gather of a file-parallel corpus transcode.
//
// StreamDecoder holds per-stream state only (NVorbis/StreamDecoder.cs:35-39), so a corpus shards by FILE: the host starts one
// process per GPU (rank r of `world`), every process decodes its files on its own device, and the samples meet on rank 0's
// GPU over RCCL / xGMI -- device memory to device memory, no host bounce.  RCCL is reached through libnvorbis_hip.so
// (include/nvorbis_hip.h, "multi-GPU"; nvorbis_amd/csrc/nvh_comm.hip): the managed side needs no binding of its own.
//
//     // rank 0:  File.WriteAllBytes(idPath, GpuCorpusGather.NewId());   every rank: id = File.ReadAllBytes(idPath)
//     using var g = new GpuCorpusGather(ctx, id, rank, world);
//     long[][] perFile = g.AllGather(myFloatCountPerFile);              // [rank][file]
//     g.Gather(dMine, myTotal, dAll /* rank 0: sum of all totals */, totals);
//
// Source-only here (no .NET toolchain in the build image); the same call sequence runs in this repository through the
// Python mirror nvorbis_amd.Comm / nvorbis_amd.corpus.gather_pcm_native (tests/test_multi_rank_gpu.py, on an MI355X).
using System;

namespace NVorbis.Hip
{
    public sealed unsafe class GpuCorpusGather : IDisposable
    {
        IntPtr _comm;
        public int Rank { get; }
        public int World { get; }

        /// <summary>ncclGetUniqueId: rank 0 makes it, every rank passes the same bytes to the constructor.</summary>
        public static byte[] NewId()
        {
            var id = new byte[NativeMethods.NVH_COMM_ID_BYTES];
            fixed (byte* p = id) NativeMethods.Check(NativeMethods.nvh_comm_unique_id(p));
            return id;
        }

        /// <summary>Collective: returns when all `world` processes have constructed theirs (ncclCommInitRank on ctx's device).</summary>
        public GpuCorpusGather(IntPtr ctx, byte[] id, int rank, int world)
        {
            if (id == null || id.Length != NativeMethods.NVH_COMM_ID_BYTES) throw new ArgumentException("id");
            fixed (byte* p = id) NativeMethods.Check(NativeMethods.nvh_comm_create(ctx, p, rank, world, out _comm));
            Rank = rank;
            World = world;
        }

        /// <summary>n words from every rank to every rank, e.g. the float count of each file this rank decoded (0 for the others').</summary>
        public long[][] AllGather(long[] mine)
        {
            var all = new long[World * mine.Length];
            fixed (long* m = mine) fixed (long* a = all) NativeMethods.Check(NativeMethods.nvh_comm_allgather_i64(_comm, m, mine.Length, a));
            var rows = new long[World][];
            for (int r = 0; r < World; r++) { rows[r] = new long[mine.Length]; Array.Copy(all, r * mine.Length, rows[r], 0, mine.Length); }
            return rows;
        }

        /// <summary>Rank r's sendCount floats at dSend arrive at dRecv + sum(totals[0..r-1]) on `root`; waits for the transfers.</summary>
        public void Gather(IntPtr dSend, long sendCount, IntPtr dRecv, long[] totals, int root = 0)
        {
            fixed (long* t = totals) NativeMethods.Check(NativeMethods.nvh_comm_gather_pcm(_comm, dSend, sendCount, dRecv, t, root, 0));
        }

        public void Dispose()
        {
            if (_comm != IntPtr.Zero) { NativeMethods.nvh_comm_destroy(_comm); _comm = IntPtr.Zero; }
        }
    }
}
