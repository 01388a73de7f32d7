// nvh_comm.hip -- the one collective of the path, for a host that has no torch.distributed (the C# host of INTEGRATION.md):
// the file-parallel corpus transcode (SURVEY section 8 row e; StreamDecoder.cs:35-39 holds per-stream state only, so files are
// the unit of parallelism) decodes its shard on every GPU and then gathers the PCM on one of them -- "RCCL over xGMI only for
// the final sample gather".  nvorbis_amd/corpus.py does the same exchange through torch.distributed (backend "nccl" = RCCL);
// these entry points do it through RCCL's C API directly, so that a process which is not Python can: an all-gather of the
// per-rank sample counts, then one flat payload per rank, point to point to the root, every transfer posted inside one group
// (xGMI is point to point: each peer has its own link to the root and the root receives from all of them at once; a ring
// collective would be the wrong shape).
//
// RCCL is loaded on first use (dlopen of its soname -- a process that already holds a copy, e.g. PyTorch's, gets that copy):
// the library itself has no link-time dependency on it, and a single-GPU caller never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without RCCL's development headers still builds the library (the single-GPU path never touches RCCL): the
// handful of types and constants of the NCCL ABI these entry points use, as rccl.h / nccl.h declare them.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt64 = 4, ncclFloat32 = 7 } ncclDataType_t;
#endif

#include <mutex>
#include <vector>

#include "nvh_internal.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

const Rccl& rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      R.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (R.handle) break;
    }
    if (!R.handle) return;
    auto sym = [&](const char* n) { return dlsym(R.handle, n); };
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(sym("ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(sym("ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(sym("ncclCommDestroy"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(sym("ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(sym("ncclGroupEnd"));
    R.Send = reinterpret_cast<decltype(R.Send)>(sym("ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(sym("ncclRecv"));
    R.AllGather = reinterpret_cast<decltype(R.AllGather)>(sym("ncclAllGather"));
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv && R.AllGather;
  });
  return R;
}

// an RCCL failure reads as a device error; nvh_last_hip_error() = 10000 + the ncclResult_t
#define NCCL_TRY(expr)                          \
  do {                                          \
    ncclResult_t r_ = (expr);                   \
    if (r_ != ncclSuccess) {                    \
      g_last_hip_error = 10000 + (int)r_;       \
      return NVH_ERR_DEVICE;                    \
    }                                           \
  } while (0)

}  // namespace

struct nvh_comm {
  nvh_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  int64_t* d_counts = nullptr;  // the all-gather's device buffers (grown on demand)
  size_t d_words = 0;
};

static_assert(NVH_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id travels as RCCL made it");

extern "C" int nvh_comm_unique_id(uint8_t* id) {
  return nvh_guard([&]() -> int {
    if (!id) return NVH_ERR_ARGUMENT;
    const Rccl& R = rccl();
    if (!R.ok) return NVH_ERR_UNSUPPORTED;
    ncclUniqueId u;
    NCCL_TRY(R.GetUniqueId(&u));
    std::memcpy(id, u.internal, NVH_COMM_ID_BYTES);
    return NVH_OK;
  });
}

extern "C" int nvh_comm_create(nvh_ctx* ctx, const uint8_t* id, int rank, int world, nvh_comm** out) {
  return nvh_guard([&]() -> int {
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return NVH_ERR_ARGUMENT;
    *out = nullptr;
    const Rccl& R = rccl();
    if (!R.ok) return NVH_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NVH_COMM_ID_BYTES);
    nvh_comm* c = new nvh_comm;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclResult_t r = R.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
      delete c;
      g_last_hip_error = 10000 + (int)r;
      return NVH_ERR_DEVICE;
    }
    *out = c;
    return NVH_OK;
  });
}

extern "C" void nvh_comm_destroy(nvh_comm* c) {
  if (!c) return;
  nvh_guard([&]() -> int {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) rccl().CommDestroy(c->comm);
    if (c->d_counts) (void)hipFree(c->d_counts);
    delete c;
    return NVH_OK;
  });
}

extern "C" int nvh_comm_info(const nvh_comm* c, int* rank, int* world) {
  if (!c) return NVH_ERR_ARGUMENT;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return NVH_OK;
}

extern "C" int nvh_comm_allgather_i64(nvh_comm* c, const int64_t* mine, int n, int64_t* all) {
  return nvh_guard([&]() -> int {
    if (!c || !mine || !all || n < 1) return NVH_ERR_ARGUMENT;
    const Rccl& R = rccl();
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    const size_t words = (size_t)(c->world + 1) * (size_t)n;  // the receive buffer, then this rank's words
    if (words > c->d_words) {
      HIP_TRY(hipStreamSynchronize(st));
      if (c->d_counts) HIP_TRY(hipFree(c->d_counts));
      c->d_counts = nullptr;
      c->d_words = 0;
      hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_counts), words * sizeof(int64_t));
      if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return NVH_ERR_NOMEM;
      }
      c->d_words = words;
    }
    int64_t* d_mine = c->d_counts + (size_t)c->world * (size_t)n;
    HIP_TRY(hipMemcpyAsync(d_mine, mine, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
    NCCL_TRY(R.AllGather(d_mine, c->d_counts, (size_t)n, ncclInt64, c->comm, st));
    HIP_TRY(hipMemcpyAsync(all, c->d_counts, (size_t)c->world * (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return NVH_OK;
  });
}

extern "C" int nvh_comm_gather_pcm(nvh_comm* c, const float* d_send, int64_t send_count, float* d_recv, const int64_t* counts,
                                   int root, int flags) {
  return nvh_guard([&]() -> int {
    if (!c || !counts || root < 0 || root >= c->world || send_count < 0 || (send_count > 0 && !d_send)) return NVH_ERR_ARGUMENT;
    if (counts[c->rank] != send_count) return NVH_ERR_ARGUMENT;  // every rank passes what the all-gather returned
    int64_t total = 0;
    for (int r = 0; r < c->world; ++r) {
      if (counts[r] < 0) return NVH_ERR_ARGUMENT;
      total += counts[r];
    }
    if (c->rank == root && total > 0 && !d_recv) return NVH_ERR_ARGUMENT;
    const Rccl& R = rccl();
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    const bool self_p2p = (flags & NVH_GATHER_SELF_P2P) != 0;
    NCCL_TRY(R.GroupStart());
    ncclResult_t rc = ncclSuccess;
    if (c->rank == root) {
      int64_t off = 0;
      for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
        if (counts[r] > 0 && (r != root || self_p2p)) rc = R.Recv(d_recv + off, (size_t)counts[r], ncclFloat32, r, c->comm, st);
        off += counts[r];
      }
    }
    if (rc == ncclSuccess && send_count > 0 && (c->rank != root || self_p2p))
      rc = R.Send(d_send, (size_t)send_count, ncclFloat32, root, c->comm, st);
    const ncclResult_t rc_end = R.GroupEnd();
    NCCL_TRY(rc);
    NCCL_TRY(rc_end);
    if (c->rank == root && !self_p2p && send_count > 0) {  // the root's own part never leaves its HBM
      int64_t off = 0;
      for (int r = 0; r < root; ++r) off += counts[r];
      if (d_recv + off != d_send)
        HIP_TRY(hipMemcpyAsync(d_recv + off, d_send, (size_t)send_count * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return NVH_OK;
  });
}
