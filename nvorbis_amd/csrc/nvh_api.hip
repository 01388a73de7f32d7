// nvh_api.hip -- C ABI of libnvorbis_hip.so (see include/nvorbis_hip.h): device management, setup
// upload, batch upload and kernel launches.  There is no CPU fallback anywhere in this file: without a
// HIP device every compute entry point fails with NVH_ERR_NO_GPU / NVH_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <iterator>
#include <new>
#include <string>
#include <vector>

#include "../../include/nvorbis_hip.h"
#include "host_ogg.h"
#include "host_parse.h"
#include "host_setup.h"
#include "kernels_common.h"
#include "nvh_parse_format.h"


extern "C" {
__global__ void k_mdct_reverse(float* buf, int n, long long stride, const float* A, const float* B, const float* C,
                               const uint16_t* BR);
__global__ void k_imdct_window(NvhDevSetup S, NvhDevBatch Bt, float* work);
__global__ void k_imdct_wave(NvhDevSetup S, NvhDevBatch Bt, float* work);
__global__ void k_imdct_compact(NvhDevSetup S, NvhDevBatch Bt, float* work);
__global__ void k_expand_carry(NvhDevSetup S, NvhDevBatch Bt, const float* work, float* carry_out, int f);
__global__ void k_ola_compact(NvhDevSetup S, NvhDevBatch Bt, const float* work, const float* carry, float* pcm, int clip,
                              int* clipped_flag, float* carry_out, int last_decoded);
__global__ void k_spectrum(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent,
                           long long* dbg, int phase_mask);
__global__ void k_spectrum_f0(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent);
__global__ void k_spectrum_imdct(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent,
                                 long long* dbg, int phase_mask);
__global__ void k_spectrum_gen(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent, long long* dbg);
__global__ void k_spectrum_gen8(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent, long long* dbg);
__global__ void k_imdct_ola(NvhDevSetup S, NvhDevBatch Bt, const float* work, const float* carry_in, float* carry_out, float* pcm,
                            int clip, int* clipped_flag, int run_len, int last_decoded);
__global__ void k_mdct_reverse_wave(float* buf, int n, long long stride, const float* A, const float* B, const float* C,
                                    const float* TW);
__global__ void k_parse(NvhDevParse T, const uint8_t* pkt_pool, const NvhPacketRef* refs, int nframes, NvhFrame* frames, NvhChan* chans,
                        NvhResPass* passes, NvhResOp* ops, uint16_t* op_link, uint16_t* entries, uint16_t* posts, int* scratch,
                        NvhParseResult* result, int lanes, int scratch_words, int pkt_words);
__global__ void k_parse_g(NvhDevParse T, const uint8_t* pkt_pool, const NvhPacketRef* refs, int nframes, NvhFrame* frames, NvhChan* chans,
                          NvhResPass* passes, NvhResOp* ops, uint16_t* op_link, uint16_t* entries, uint16_t* posts, int* scratch,
                          NvhParseResult* result, int lanes, int scratch_words, int pkt_words);
__global__ void k_parse_links(int nframes, int channels, NvhFrame* frames, NvhChan* chans, const uint32_t* carry_exec_in,
                              uint32_t* carry_exec_out, int last_decoded);
__global__ void k_inverse_couple(float* magnitude, float* angle, int cnt);
__global__ void k_window_apply(float* buf, const float* window, int n, long long stride, int batch);
__global__ void k_overlap_buffers(const float* previous, float* next, int prev_start, int len, int next_start, int channels,
                                  long long plane_stride);
__global__ void k_copy_buffer(const float* planes, int start, int count, int channels, long long plane_stride, float* target,
                              int clip, int* clipped_flag);
__global__ void k_floor0_apply(NvhDevSetup S, int floor_idx, const float* amps, const float* coeffs, int coeff_stride, int n,
                               float* data, long long stride, int* status);
__global__ void k_floor1_apply(NvhDevSetup S, int floor_idx, const uint16_t* posts, const int32_t* counts, int n, float* data,
                               long long stride, int* status);
__global__ void k_residue(NvhDevSetup S, NvhDevBatch Bt, float* work, int clear);
__global__ void k_couple_floor(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err);
__global__ void k_ola_emit(NvhDevSetup S, NvhDevBatch Bt, const float* work, const float* carry, float* pcm, int clip,
                           int* clipped_flag);
__global__ void k_ola_emit_seq(NvhDevSetup S, NvhDevBatch Bt, float* work, const float* carry, float* pcm, int clip,
                               int* clipped_flag);
}

static thread_local int g_last_hip_error = 0;
static void* g_dbg_buf = nullptr;  // profiling aid: per-workgroup phase timestamps of k_spectrum (nvh_debug_set_buffer)
extern "C" void nvh_debug_set_buffer(void* d_buf) { g_dbg_buf = d_buf; }

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) {                  \
      g_last_hip_error = (int)e_;            \
      return NVH_ERR_DEVICE;                 \
    }                                        \
  } while (0)

namespace {

// Device allocations are recycled through a per-context pool: hipMalloc / hipFree cost 0.1-1 ms each and
// serialise inside the runtime, which is what a file-parallel transcoder (many short streams per context, many
// contexts per GPU) would otherwise spend its time on.  Every buffer of a context is used on that context's HIP
// stream only, so handing a block from a closed stream to the next one is ordered by the stream itself.
struct BufPool {
  bool host = false;  // true: pinned host memory (hipHostMalloc), staging for asynchronous copies
  std::multimap<size_t, void*> free_;
  size_t bytes_ = 0;
  void raw_free(void* p) const { (void)(host ? hipHostFree(p) : hipFree(p)); }
  static constexpr size_t kKeepBytes = (size_t)2 << 30;  // beyond this, returned blocks go back to the runtime
  // size classes with two mantissa bits (<= 25 % slack) so that blocks are interchangeable between streams
  static size_t size_class(size_t bytes) {
    size_t v = bytes < 4096 ? 4096 : bytes;
    size_t p = 1;
    while ((p << 1) <= v) p <<= 1;
    size_t step = p >> 2;
    return (v + step - 1) / step * step;
  }
  void* take(size_t cls) {
    auto it = free_.find(cls);
    if (it == free_.end()) return nullptr;
    void* p = it->second;
    free_.erase(it);
    bytes_ -= cls;
    return p;
  }
  void give(void* p, size_t cls) {
    if (bytes_ + cls > kKeepBytes) {
      raw_free(p);
      return;
    }
    free_.emplace(cls, p);
    bytes_ += cls;
  }
  void clear() {
    for (auto& kv : free_) raw_free(kv.second);
    free_.clear();
    bytes_ = 0;
  }
};

struct DevBuf {  // growable device (or pinned host) allocation, optionally backed by a context's pool
  void* p = nullptr;
  size_t cap = 0;
  BufPool* pool = nullptr;
  bool host = false;  // pinned host memory; must match pool->host
  ~DevBuf() { release(); }
  void release() {
    if (!p) return;
    if (pool) pool->give(p, cap);
    else (void)(host ? hipHostFree(p) : hipFree(p));
    p = nullptr;
    cap = 0;
  }
  int reserve(size_t bytes) {
    if (bytes <= cap) return NVH_OK;
    release();
    const size_t want = BufPool::size_class(bytes + 256);
    if (pool) p = pool->take(want);
    if (!p) {
      if (host) HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
      else HIP_TRY(hipMalloc(&p, want));
    }
    cap = want;
    return NVH_OK;
  }
};

struct MdctDev {
  int n = 0;
  float *a = nullptr, *b = nullptr, *c = nullptr, *tw = nullptr;
  uint16_t* br = nullptr;
};

}  // namespace

// Everything derived from a stream's headers: parsed tables on the host, their device image, kernel-selection
// flags.  Immutable once built, so streams with byte-identical identification + setup packets (the normal case
// inside one corpus: same encoder, same settings) share one entry per context.
struct SharedSetup {
  nvh::Setup setup;
  DevBuf arena;  // setup tables
  NvhDevSetup dev{};
  bool fast_spectrum = false;  // every residue takes the pair path and the fused tail applies: k_spectrum proper
  bool has_floor0 = false;
  // GPU packet parser (kernels_parse.hip): its tables, and whether this stream shape is inside its limits
  DevBuf parse_arena;
  NvhDevParse parse{};
  bool gpu_parse_ok = false;
};

struct nvh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::map<int, MdctDev> mdct_cache;  // Mdct._setupCache (Mdct.cs:11)
  BufPool pool;
  BufPool hpool;  // pinned staging blocks
  std::map<std::string, std::shared_ptr<SharedSetup>> setup_cache;  // key: identification packet + setup packet bytes
};

struct nvh_batch {
  nvh_stream* s = nullptr;
  DevBuf blob;          // all descriptor arrays, one allocation
  DevBuf h_blob;        // pinned staging image of it (the upload is asynchronous)
  DevBuf slabs;         // GPU-parse mode: per-frame output slabs of k_parse + its scratch + result block
  DevBuf work;          // [frames][ch][block1] float planes
  DevBuf carry_in;      // snapshot of the tail this batch overlaps its first frame with
  NvhDevBatch dev{};
  int nframes = 0, chan_frames = 0;
  int64_t pcm_samples = 0;
  int64_t descriptor_bytes = 0;
  int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // frames, chans, passes, ops, entries, posts, coeffs, -
  bool sequential_ola = false;
  int last_decoded = -1;  // last frame with n != 0 (its block becomes the next carried tail)
  const char* slot_name[4] = {"-", "-", "-", "-"};  // kernels behind the four timing slots of the last launch
  bool links_ok = false;  // op_link chains usable (every frame has < 32767 ops): k_spectrum's chain walk
  int max_ops = 0, max_ent = 0, max_pass = 0;  // largest per-frame op / entry / pass slice (LDS staging capacity of k_spectrum)
  bool fused_ola = false;        // geometry admits k_imdct_ola (see its preconditions)
  bool block_only = false;       // nvh_mode_decode: stop after the windowed IMDCT (full blocks in the work planes, no overlap-add)
  bool has_carry_in = false;
};

struct nvh_stream {
  nvh_ctx* ctx = nullptr;
  std::shared_ptr<SharedSetup> shared;
  nvh::Setup& setup;
  DevBuf& arena;
  NvhDevSetup& dev;
  bool& fast_spectrum;
  bool& has_floor0;
  std::unique_ptr<nvh::StreamParser> parser;
  nvh::FrameBatch pending;
  DevBuf carry[2];  // [ch][block1] windowed block of the last decoded frame (ping-pong: read one, write the other)
  int carry_cur = 0;
  DevBuf flags;  // int[2]: device error word, clipped flag
  DevBuf pcm;    // staging for host-destination synth
  DevBuf h_pcm;  // pinned bounce buffer behind it (+ 2 ints: the flag words), read back asynchronously
  int clip = 1;
  int has_clipped = 0;
  bool gpu_parse = false;  // packets are parsed by k_parse; the host parser runs in light mode
  DevBuf carry_exec;       // uint32[2], ping-pong with carry[]: execute flags of the carried block (GPU-parse mode)
  nvh_batch scratch;  // reused by nvh_stream_synth

  nvh_stream(nvh_ctx* c, std::shared_ptr<SharedSetup> sh)
      : ctx(c), shared(std::move(sh)), setup(shared->setup), arena(shared->arena), dev(shared->dev),
        fast_spectrum(shared->fast_spectrum), has_floor0(shared->has_floor0) {
    BufPool* pool = c ? &c->pool : nullptr;
    carry[0].pool = carry[1].pool = flags.pool = pcm.pool = carry_exec.pool = pool;
    scratch.blob.pool = scratch.work.pool = scratch.carry_in.pool = scratch.slabs.pool = pool;
    h_pcm.host = scratch.h_blob.host = true;
    h_pcm.pool = scratch.h_blob.pool = c ? &c->hpool : nullptr;
    scratch.s = this;
  }
};

// ------------------------------------------------------------------------------------------------

// hipEvent that is destroyed on every path out of a function (the HIP_TRY macro returns early).
struct ScopedEvent {
  hipEvent_t e = nullptr;
  ~ScopedEvent() {
    if (e) (void)hipEventDestroy(e);
  }
  int create() { HIP_TRY(hipEventCreate(&e)); return NVH_OK; }
};

static int ensure_device(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_last_hip_error = (int)e;
    return NVH_ERR_NO_GPU;
  }
  if (device < 0 || device >= count) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(device));
  return NVH_OK;
}

extern "C" const char* nvh_version(void) { return "nvorbis_hip 0.1 (gfx950)"; }
extern "C" int nvh_last_hip_error(void) { return g_last_hip_error; }
extern "C" int nvh_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

extern "C" int nvh_ctx_create(int device, nvh_ctx** out) {
  if (!out) return NVH_ERR_ARGUMENT;
  *out = nullptr;
  int rc = ensure_device(device);
  if (rc != NVH_OK) return rc;
  nvh_ctx* c = new (std::nothrow) nvh_ctx();
  if (!c) return NVH_ERR_NOMEM;
  c->device = device;
  c->hpool.host = true;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    delete c;
    return NVH_ERR_DEVICE;
  }
  c->own_stream = true;
  *out = c;
  return NVH_OK;
}

extern "C" void nvh_ctx_destroy(nvh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& kv : c->mdct_cache) {
    (void)hipFree(kv.second.a);
    (void)hipFree(kv.second.b);
    (void)hipFree(kv.second.c);
    (void)hipFree(kv.second.br);
    (void)hipFree(kv.second.tw);
  }
  c->setup_cache.clear();  // streams must have been closed: they share these entries and return their buffers here
  c->pool.clear();
  c->hpool.clear();
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int nvh_ctx_set_hip_stream(nvh_ctx* c, void* hip_stream) {
  if (!c) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  if (c->own_stream && c->stream) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipStreamDestroy(c->stream);
    c->own_stream = false;
    c->stream = nullptr;
  }
  if (hip_stream == nullptr) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  } else {
    c->stream = (hipStream_t)hip_stream;
  }
  return NVH_OK;
}

extern "C" int nvh_ctx_synchronize(nvh_ctx* c) {
  if (!c) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// level 1
// ------------------------------------------------------------------------------------------------

static bool valid_block(int n) { return n >= 64 && n <= 8192 && (n & (n - 1)) == 0; }
// LDS bytes of the wavefront IMDCT: n/4 complex points + 1/8 padding (kernels_imdct.hip Geo<LD>::LDS_FLOATS)
static size_t wave_lds_bytes(int n) { return (size_t)2 * ((size_t)(n / 4) + (size_t)(n / 32)) * sizeof(float); }

extern "C" int nvh_mdct_tables(int n, float* a, float* b, float* c, uint16_t* bitrev) {
  if (!valid_block(n) || !a || !b || !c || !bitrev) return NVH_ERR_ARGUMENT;
  nvh::MdctTables t;
  nvh::build_mdct_tables(n, t);
  std::memcpy(a, t.a.data(), t.a.size() * sizeof(float));
  std::memcpy(b, t.b.data(), t.b.size() * sizeof(float));
  std::memcpy(c, t.c.data(), t.c.size() * sizeof(float));
  std::memcpy(bitrev, t.bitrev.data(), t.bitrev.size() * sizeof(uint16_t));
  return NVH_OK;
}

extern "C" int nvh_calc_window(int prev_block, int block, int next_block, float* out) {
  if (!out || block <= 0 || prev_block <= 0 || next_block <= 0 || prev_block > block || next_block > block) return NVH_ERR_ARGUMENT;
  nvh::calc_window(prev_block, block, next_block, out);
  return NVH_OK;
}

extern "C" int nvh_calc_overlap(int prev_block, int block, int next_block, int* start, int* valid, int* total) {
  if (!start || !valid || !total) return NVH_ERR_ARGUMENT;
  nvh::calc_overlap(prev_block, block, next_block, start, valid, total);
  return NVH_OK;
}

static int get_mdct(nvh_ctx* c, int n, MdctDev** out) {
  auto it = c->mdct_cache.find(n);
  if (it != c->mdct_cache.end()) {
    *out = &it->second;
    return NVH_OK;
  }
  nvh::MdctTables t;
  nvh::build_mdct_tables(n, t);
  MdctDev d;
  d.n = n;
  HIP_TRY(hipMalloc((void**)&d.a, t.a.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&d.b, t.b.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&d.c, t.c.size() * sizeof(float)));
  HIP_TRY(hipMalloc((void**)&d.br, t.bitrev.size() * sizeof(uint16_t)));
  HIP_TRY(hipMemcpy(d.a, t.a.data(), t.a.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.b, t.b.data(), t.b.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.c, t.c.data(), t.c.size() * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d.br, t.bitrev.data(), t.bitrev.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  if (!t.tw.empty()) {
    HIP_TRY(hipMalloc((void**)&d.tw, t.tw.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(d.tw, t.tw.data(), t.tw.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  auto ins = c->mdct_cache.emplace(n, d);
  *out = &ins.first->second;
  return NVH_OK;
}

extern "C" int nvh_inverse_couple(nvh_ctx* c, float* d_magnitude, float* d_angle, int count) {
  if (!c || !d_magnitude || !d_angle || count < 0) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  if (count == 0) return NVH_OK;
  int blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(k_inverse_couple, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_magnitude, d_angle, count);
  HIP_TRY(hipGetLastError());
  return NVH_OK;
}

extern "C" int nvh_mdct_reverse(nvh_ctx* c, int n, int batch, float* d_buf, int64_t stride) {
  if (!c || !d_buf || batch < 0 || !valid_block(n) || stride < n) return NVH_ERR_ARGUMENT;
  if (batch == 0) return NVH_OK;
  HIP_TRY(hipSetDevice(c->device));
  MdctDev* m = nullptr;
  int rc = get_mdct(c, n, &m);
  if (rc != NVH_OK) return rc;
  if (n >= 256)  // wavefront-per-buffer radix-8 path
    hipLaunchKernelGGL(k_mdct_reverse_wave, dim3((unsigned)batch), dim3(64), wave_lds_bytes(n), c->stream, d_buf, n,
                       (long long)stride, m->a, m->b, m->c, m->tw);
  else  // 64 / 128: generic stage-synchronous kernel (the reference's loops over-count there, quirk B-10)
    hipLaunchKernelGGL(k_mdct_reverse, dim3((unsigned)batch), dim3(256), (size_t)n * sizeof(float), c->stream, d_buf, n,
                       (long long)stride, m->a, m->b, m->c, m->br);
  HIP_TRY(hipGetLastError());
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// stream: setup upload
// ------------------------------------------------------------------------------------------------

namespace {

struct ArenaBuilder {
  std::vector<uint8_t> bytes;
  size_t add(const void* src, size_t n, size_t align = 16) {
    size_t off = (bytes.size() + align - 1) / align * align;
    bytes.resize(off + n);
    if (n) std::memcpy(bytes.data() + off, src, n);
    return off;
  }
};

}  // namespace

static int upload_parse_tables(nvh_stream* s);

static int upload_setup(nvh_stream* s) {
  const nvh::Setup& S = s->setup;
  ArenaBuilder ab;
  // documented limits of this build
  if (S.channels > 255) return NVH_ERR_UNSUPPORTED;
  if (S.books.size() > 256) return NVH_ERR_UNSUPPORTED;

  std::vector<float> vq;
  std::vector<uint32_t> lattice;
  std::vector<NvhDevBook> books(S.books.size());
  for (size_t i = 0; i < S.books.size(); i++) {
    const nvh::Codebook& b = S.books[i];
    books[i].lat_values = 0;
    books[i].lat_magic = 0;
    books[i].lat_off = 0;
    books[i].dim_magic16 = b.dimensions >= 1 ? (uint32_t)((65536u + (uint32_t)b.dimensions - 1u) / (uint32_t)b.dimensions) : 0u;
    // lattice fast path: digits via exact reciprocal multiplies (entry < 2^16, powers <= entries)
    if (b.lattice_values >= 1 && b.dimensions >= 1 && b.dimensions <= 16 && b.entries <= 0xFFFF) {
      bool ok = true;
      std::vector<uint32_t> magics;
      uint64_t pw = 1;
      for (int d = 0; d < b.dimensions && ok; d++) {
        if (pw > 0xFFFF) { ok = false; break; }
        magics.push_back(pw > 1 ? (uint32_t)((0x100000000ull + pw - 1) / pw) : 0u);  // 0: divisor 1
        pw *= (uint64_t)b.lattice_values;
      }
      // self-check against the table the reference algorithm builds
      for (int e = 0; ok && e < b.entries; e++) {
        int q = e;
        for (int d = 0; d < b.dimensions; d++) {
          uint32_t bits_t, bits_l;
          float tv = b.lookup[(size_t)e * b.dimensions + d], lv = b.lattice[(size_t)(q % b.lattice_values)];
          std::memcpy(&bits_t, &tv, 4);
          std::memcpy(&bits_l, &lv, 4);
          if (bits_t != bits_l) { ok = false; break; }
          q /= b.lattice_values;
        }
      }
      if (ok) {
        books[i].lat_values = (uint32_t)b.lattice_values;
        books[i].lat_magic = b.lattice_values > 1 ? (uint32_t)((0x100000000ull + (uint64_t)b.lattice_values - 1) / (uint64_t)b.lattice_values) : 0u;
        books[i].lat_off = (uint32_t)lattice.size();
        for (float v : b.lattice) {
          uint32_t bits;
          std::memcpy(&bits, &v, 4);
          lattice.push_back(bits);
        }
        lattice.insert(lattice.end(), magics.begin(), magics.end());
      }
    }
    books[i].entries = (uint32_t)b.entries;
    books[i].dim = (uint32_t)b.dimensions;
    books[i].dim_magic = b.dimensions > 1 ? (uint32_t)((0x100000000ull + (uint64_t)b.dimensions - 1) / (uint64_t)b.dimensions) : 0u;
    if (b.map_type == 0) {
      books[i].tab_off = 0xFFFFFFFFu;
    } else {
      books[i].tab_off = (uint32_t)vq.size();
      vq.insert(vq.end(), b.lookup.begin(), b.lookup.end());
    }
  }

  std::vector<int32_t> ipool;
  std::vector<float> fpool;
  std::vector<NvhDevFloor> floors(S.floors.size());
  for (size_t i = 0; i < S.floors.size(); i++) {
    const nvh::Floor& f = S.floors[i];
    NvhDevFloor& d = floors[i];
    std::memset(&d, 0, sizeof d);
    d.type = f.type;
    if (f.type == 1) {
      int cnt = (int)f.f1.x_list.size();
      d.f1.x_count = cnt;
      d.f1.multiplier = f.f1.multiplier;
      d.f1.range = f.f1.range;
      int lim = cnt < NVH_MAX_POSTS ? cnt : NVH_MAX_POSTS;  // more than 64 posts faults at decode time (host parser)
      int levels = 0;
      for (int k = 0; k < lim; k++) {
        if (f.f1.x_list[k] > 0xFFFF) return NVH_ERR_UNSUPPORTED;
        d.f1.x_list[k] = (uint16_t)f.f1.x_list[k];
        d.f1.l_neigh[k] = (uint8_t)f.f1.l_neigh[k];
        d.f1.h_neigh[k] = (uint8_t)f.f1.h_neigh[k];
        d.f1.sort_idx[k] = (uint8_t)(f.f1.sort_idx[k] < NVH_MAX_POSTS ? f.f1.sort_idx[k] : 0);
        int lv = 0;
        if (k >= 2) {
          int a = d.f1.level[f.f1.l_neigh[k]], b = d.f1.level[f.f1.h_neigh[k]];
          lv = (a > b ? a : b) + 1;
        }
        d.f1.level[k] = (uint8_t)lv;
        if (lv + 1 > levels) levels = lv + 1;
      }
      for (int k = 0; k < lim; k++) {
        const int lo = d.f1.l_neigh[k], hi = d.f1.h_neigh[k];
        d.f1.x_lo[k] = d.f1.x_list[lo < lim ? lo : 0];
        d.f1.x_hi[k] = d.f1.x_list[hi < lim ? hi : 0];
        d.f1.x_sorted[k] = d.f1.x_list[d.f1.sort_idx[k]];
        const int adx = (int)d.f1.x_hi[k] - (int)d.f1.x_lo[k];
        d.f1.adx_magic[k] = (k >= 2 && adx > 0) ? 0xFFFFFFFFu / (uint32_t)adx : 0u;
      }
      d.f1.levels = levels;
    } else {
      d.f0.order = f.f0.order;
      d.f0.amp_ofs = f.f0.amp_ofs;
      d.f0.bark_map_size = f.f0.bark_map_size;
      if (f.f0.order > 255) return NVH_ERR_UNSUPPORTED;
      for (int w = 0; w < 2; w++) {
        d.f0.bark_off[w] = (uint32_t)ipool.size();
        ipool.insert(ipool.end(), f.f0.bark_map[w].begin(), f.f0.bark_map[w].end());
        d.f0.wmap_off[w] = (uint32_t)fpool.size();
        fpool.insert(fpool.end(), f.f0.w_map[w].begin(), f.f0.w_map[w].end());
      }
    }
  }

  std::vector<NvhDevResidue> residues(S.residues.size());
  for (size_t i = 0; i < S.residues.size(); i++) {
    const nvh::Residue& r = S.residues[i];
    NvhDevResidue& d = residues[i];
    d.type = r.type;
    d.begin = r.begin;
    d.end = r.end;
    d.partition_size = r.partition_size;
    d.classifications = r.classifications;
    d.channels = r.channels;
    d.real_channels = r.real_channels;
    bool seq = false;
    if (r.type == 2 && (r.begin % r.real_channels != 0 || r.partition_size % r.real_channels != 0)) seq = true;
    for (int c = 0; c < r.classifications; c++)
      for (int k = 0; k < NVH_MAX_STAGES; k++) {
        int b = r.books[c][k];
        if (b < 0) continue;
        const nvh::Codebook& bk = S.books[(size_t)b];
        if (bk.entries > 0xFFFF) return NVH_ERR_UNSUPPORTED;  // entry stream is 16-bit (0xFFFF = skip)
        if (bk.dimensions > 0 && r.type != 0 && r.partition_size % bk.dimensions != 0) seq = true;  // vector overrun
      }
    d.sequential = seq ? 1 : 0;
    d.psize_magic = r.partition_size > 1 ? (uint32_t)((0x100000000ull + (uint64_t)r.partition_size - 1) / (uint64_t)r.partition_size) : 0u;
    d.rch_magic = r.real_channels > 1 ? (uint32_t)((0x100000000ull + (uint64_t)r.real_channels - 1) / (uint64_t)r.real_channels) : 0u;
    // reciprocal multiplies are exact while index * divisor < 2^32; indices stay below
    // (partitions per stage) * channels * partition_size <= block1/2 * channels (+ one partition of overrun)
    {
      uint64_t max_index = (uint64_t)(S.block1 / 2 + r.partition_size) * (uint64_t)(r.real_channels > 0 ? r.real_channels : 1);
      uint64_t max_div = (uint64_t)r.partition_size;
      if ((uint64_t)r.real_channels > max_div) max_div = (uint64_t)r.real_channels;
      for (int c = 0; c < r.classifications; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++)
          if (r.books[c][k] >= 0 && (uint64_t)S.books[(size_t)r.books[c][k]].dimensions > max_div)
            max_div = (uint64_t)S.books[(size_t)r.books[c][k]].dimensions;
      d.fast = (r.type != 0 && r.partition_size > 1 && max_index * max_div < 0x100000000ull) ? 1 : 0;
      // pair records pack LDS offsets / bin indices into 16 bits and use a 16-bit reciprocal of the book dimension
      bool pairs = d.fast != 0 && !seq && (r.partition_size % 2) == 0 && r.partition_size <= 4096 && lattice.size() <= 0xFFFFu;
      for (int c = 0; c < r.classifications; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++)
          if (r.books[c][k] >= 0 && (books[(size_t)r.books[c][k]].lat_values == 0 || (books[(size_t)r.books[c][k]].dim & 1u))) pairs = false;
      if (std::getenv("NVH_NO_PAIR")) pairs = false;  // A/B aid
      d.pair_path = pairs ? 1 : 0;
      d.hp_magic = r.partition_size / 2 > 1 ? (uint32_t)((0x100000000ull + (uint64_t)(r.partition_size / 2) - 1) / (uint64_t)(r.partition_size / 2)) : 0u;
      d.pad[0] = d.pad[1] = d.pad[2] = 0;
    }
  }

  std::vector<uint8_t> coupling;
  std::vector<NvhDevMapping> mappings(S.mappings.size());
  for (size_t i = 0; i < S.mappings.size(); i++) {
    mappings[i].coupling_steps = (int32_t)S.mappings[i].coupling_angle.size();
    mappings[i].coupling_off = (uint32_t)coupling.size();
    for (size_t k = 0; k < S.mappings[i].coupling_angle.size(); k++) {
      coupling.push_back((uint8_t)S.mappings[i].coupling_magnitude[k]);
      coupling.push_back((uint8_t)S.mappings[i].coupling_angle[k]);
    }
  }
  if (coupling.empty()) coupling.push_back(0);
  if (vq.empty()) vq.push_back(0.0f);
  if (lattice.empty()) lattice.push_back(0u);
  if (ipool.empty()) ipool.push_back(0);
  if (fpool.empty()) fpool.push_back(0.0f);

  size_t o_vq = ab.add(vq.data(), vq.size() * sizeof(float));
  size_t o_lat = ab.add(lattice.data(), lattice.size() * sizeof(uint32_t));
  size_t o_books = ab.add(books.data(), books.size() * sizeof(NvhDevBook));
  size_t o_floors = ab.add(floors.data(), floors.size() * sizeof(NvhDevFloor));
  size_t o_res = ab.add(residues.data(), residues.size() * sizeof(NvhDevResidue));
  size_t o_map = ab.add(mappings.data(), mappings.size() * sizeof(NvhDevMapping));
  size_t o_cpl = ab.add(coupling.data(), coupling.size());
  size_t o_win = ab.add(S.windows.data(), S.windows.size() * sizeof(float));
  // reciprocals of every possible floor segment length (kernels_spectrum.hip: floor_prepare)
  std::vector<uint32_t> recip((size_t)S.block1 / 2 + 1, 0u);
  for (size_t d = 1; d < recip.size(); d++) recip[d] = (uint32_t)(0xFFFFFFFFull / d);
  size_t o_recip = ab.add(recip.data(), recip.size() * sizeof(uint32_t));
  size_t o_ip = ab.add(ipool.data(), ipool.size() * sizeof(int32_t));
  size_t o_fp = ab.add(fpool.data(), fpool.size() * sizeof(float));
  size_t o_a[2], o_b[2], o_c[2], o_br[2], o_tw[2];
  for (int w = 0; w < 2; w++) {
    o_a[w] = ab.add(S.mdct[w].a.data(), S.mdct[w].a.size() * sizeof(float));
    o_b[w] = ab.add(S.mdct[w].b.data(), S.mdct[w].b.size() * sizeof(float));
    o_c[w] = ab.add(S.mdct[w].c.data(), S.mdct[w].c.size() * sizeof(float));
    o_br[w] = ab.add(S.mdct[w].bitrev.data(), S.mdct[w].bitrev.size() * sizeof(uint16_t));
    o_tw[w] = ab.add(S.mdct[w].tw.data(), S.mdct[w].tw.size() * sizeof(float));
  }

  s->has_floor0 = false;
  for (const auto& fl : S.floors) s->has_floor0 = s->has_floor0 || fl.type == 0;
  int rc = s->arena.reserve(ab.bytes.size());
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpy(s->arena.p, ab.bytes.data(), ab.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t* base = (const uint8_t*)s->arena.p;
  NvhDevSetup& D = s->dev;
  D.channels = S.channels;
  D.block0 = S.block0;
  D.block1 = S.block1;
  D.nbooks = (int32_t)S.books.size();
  D.vq = (const float*)(base + o_vq);
  D.lattice = (const uint32_t*)(base + o_lat);
  D.lattice_words = (int32_t)lattice.size();
  {
    bool ok = S.channels <= 2 && !s->has_floor0;
    for (const nvh::Mapping& m : S.mappings) ok = ok && m.coupling_angle.size() <= 1;
    D.fused_tail_ok = ok ? 1 : 0;
    bool all_pairs = true;
    for (const NvhDevResidue& r : residues) all_pairs = all_pairs && r.pair_path != 0;
    s->fast_spectrum = ok && all_pairs;
  }
  D.books = (const NvhDevBook*)(base + o_books);
  D.floors = (const NvhDevFloor*)(base + o_floors);
  D.residues = (const NvhDevResidue*)(base + o_res);
  D.mappings = (const NvhDevMapping*)(base + o_map);
  D.coupling = base + o_cpl;
  D.windows = (const float*)(base + o_win);
  D.recip = (const uint32_t*)(base + o_recip);
  D.ipool = (const int32_t*)(base + o_ip);
  D.fpool = (const float*)(base + o_fp);
  for (int w = 0; w < 2; w++) {
    D.mdct_a[w] = (const float*)(base + o_a[w]);
    D.mdct_b[w] = (const float*)(base + o_b[w]);
    D.mdct_c[w] = (const float*)(base + o_c[w]);
    D.mdct_br[w] = (const uint16_t*)(base + o_br[w]);
    D.mdct_tw[w] = (const float*)(base + o_tw[w]);
  }
  return NVH_OK;
}

extern "C" int nvh_stream_open(nvh_ctx* c, const uint8_t* id_pkt, int id_len, const uint8_t* comment_pkt,
                               int comment_len, const uint8_t* setup_pkt, int setup_len, nvh_stream** out) {
  // ctx == NULL gives a host-only stream: packets can be parsed (and their frame geometry inspected) but
  // every synthesis entry point fails with NVH_ERR_NO_GPU -- there is no CPU synthesis path.
  if (!id_pkt || !setup_pkt || !out) return NVH_ERR_ARGUMENT;
  *out = nullptr;
  if (id_len < 0 || setup_len < 0) return NVH_ERR_ARGUMENT;
  if (c) HIP_TRY(hipSetDevice(c->device));
  int rc;
  if (comment_pkt) {  // signature check only (StreamDecoder.cs:157-175), independent of the other two headers
    nvh::Setup probe;
    rc = probe.parse_comment_sig(comment_pkt, comment_len);
    if (rc != NVH_OK) return rc;
  }
  std::string key;
  std::shared_ptr<SharedSetup> sh;
  if (c) {
    key.assign((const char*)id_pkt, (size_t)id_len);
    key.append((const char*)setup_pkt, (size_t)setup_len);
    auto it = c->setup_cache.find(key);
    if (it != c->setup_cache.end()) sh = it->second;
  }
  const bool cached = (bool)sh;
  if (!cached) {
    sh.reset(new (std::nothrow) SharedSetup());
    if (!sh) return NVH_ERR_NOMEM;
    sh->arena.pool = c ? &c->pool : nullptr;
    rc = sh->setup.parse_id(id_pkt, id_len);
    if (rc != NVH_OK) return rc;
    if (!valid_block(sh->setup.block0) || !valid_block(sh->setup.block1) || sh->setup.block0 > sh->setup.block1)
      return NVH_ERR_UNSUPPORTED;  // Vorbis I allows 64..8192 with block0 <= block1
    rc = sh->setup.parse_setup(setup_pkt, setup_len);
    if (rc != NVH_OK) return rc;
  }
  std::unique_ptr<nvh_stream> s(new (std::nothrow) nvh_stream(c, sh));
  if (!s) return NVH_ERR_NOMEM;
  s->parser.reset(new nvh::StreamParser(&s->setup));
  if (!c) {
    *out = s.release();
    return NVH_OK;
  }
  if (!cached) {
    rc = upload_setup(s.get());
    if (rc != NVH_OK) return rc;
    rc = upload_parse_tables(s.get());
    if (rc != NVH_OK) return rc;
    if (c->setup_cache.size() >= 64) {  // bounded: drop entries no open stream uses
      for (auto it = c->setup_cache.begin(); it != c->setup_cache.end();)
        it = it->second.use_count() == 1 ? c->setup_cache.erase(it) : std::next(it);
    }
    c->setup_cache.emplace(std::move(key), sh);
  }
  size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  for (int k = 0; k < 2; k++) {
    if ((rc = s->carry[k].reserve(plane)) != NVH_OK) return rc;
    HIP_TRY(hipMemsetAsync(s->carry[k].p, 0, plane, c->stream));
  }
  if ((rc = s->flags.reserve(2 * sizeof(int))) != NVH_OK) return rc;
  HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), c->stream));
  if ((rc = s->carry_exec.reserve(2 * sizeof(uint32_t))) != NVH_OK) return rc;
  HIP_TRY(hipMemsetAsync(s->carry_exec.p, 0, 2 * sizeof(uint32_t), c->stream));
  if (getenv("NVH_GPU_PARSE") && s->shared->gpu_parse_ok) {  // opt-in default for whole test runs
    s->gpu_parse = true;
    s->parser->set_light(true);
  }
  *out = s.release();
  return NVH_OK;
}

extern "C" void nvh_stream_close(nvh_stream* s) {
  if (!s) return;
  if (s->ctx) {
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
  }
  delete s;
}

extern "C" int nvh_stream_info(const nvh_stream* s, int* channels, int* sample_rate, int* block0, int* block1) {
  if (!s) return NVH_ERR_ARGUMENT;
  if (channels) *channels = s->setup.channels;
  if (sample_rate) *sample_rate = s->setup.sample_rate;
  if (block0) *block0 = s->setup.block0;
  if (block1) *block1 = s->setup.block1;
  return NVH_OK;
}

extern "C" int nvh_pinned_alloc(size_t bytes, void** out) {
  if (!out) return NVH_ERR_ARGUMENT;
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
  return NVH_OK;
}

extern "C" void nvh_pinned_free(void* p) {
  if (p) (void)hipHostFree(p);
}

extern "C" int nvh_stream_set_gpu_parse(nvh_stream* s, int on) {
  if (!s) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (!s->pending.frames.empty()) return NVH_ERR_ARGUMENT;  // switch between batches only
  if (on && !s->shared->gpu_parse_ok) return NVH_ERR_UNSUPPORTED;
  s->gpu_parse = on != 0;
  s->parser->set_light(s->gpu_parse);
  return NVH_OK;
}

extern "C" int nvh_stream_set_clip(nvh_stream* s, int on) {
  if (!s) return NVH_ERR_ARGUMENT;
  s->clip = on ? 1 : 0;
  return NVH_OK;
}

extern "C" int nvh_stream_has_clipped(nvh_stream* s, int* clipped) {
  if (!s || !clipped) return NVH_ERR_ARGUMENT;
  *clipped = s->has_clipped;
  return NVH_OK;
}

extern "C" int nvh_stream_position(const nvh_stream* s, int64_t* position, int64_t* emitted, int* eos) {
  if (!s) return NVH_ERR_ARGUMENT;
  if (position) *position = s->parser->position();
  if (emitted) *emitted = s->parser->emitted();
  if (eos) *eos = s->parser->eos() ? 1 : 0;
  return NVH_OK;
}

extern "C" int nvh_stream_position_state(const nvh_stream* s, int* has_position, int64_t* position) {
  if (!s) return NVH_ERR_ARGUMENT;
  if (has_position) *has_position = s->parser->has_position() ? 1 : 0;
  if (position) *position = s->parser->position();
  return NVH_OK;
}

extern "C" int nvh_stream_set_position_state(nvh_stream* s, int has_position, int64_t position) {
  if (!s) return NVH_ERR_ARGUMENT;
  s->parser->set_position_state(has_position != 0, position);
  return NVH_OK;
}

// Integer geometry of a run of audio packets as a serial decoder that starts with the first of them sees it (see
// include/nvorbis_hip.h).  A parser of its own in light mode: no bits beyond the packet type, mode number and window
// flags are read, nothing of `s` changes, no GPU is involved.
extern "C" int nvh_stream_index_packets(const nvh_stream* s, const uint8_t* bytes, const int64_t* offsets, const int64_t* granules,
                                        const uint8_t* flags, int n, int64_t* position_after, int64_t* emitted_after,
                                        uint8_t* state_after, int64_t* total_emitted) {
  if (!s || n < 0 || (n > 0 && (!bytes || !offsets))) return NVH_ERR_ARGUMENT;
  nvh::StreamParser one(&s->setup);
  one.set_light(true);
  nvh::FrameBatch fb;
  static const uint8_t empty = 0;
  for (int i = 0; i < n; i++) {
    const int64_t len = offsets[i + 1] - offsets[i];
    if (len < 0 || len > 0x7FFFFFFF) return NVH_ERR_ARGUMENT;
    uint8_t st = 0;
    if (!one.eos()) {
      const size_t before = fb.frames.size();
      int rc = one.push_packet(len ? bytes + offsets[i] : &empty, (int)len, granules ? granules[i] : -1, flags ? flags[i] : 0, fb);
      if (rc != NVH_OK) return rc;
      if (fb.frames.size() > before && fb.frames.back().n > 0) {
        const NvhFrame& f = fb.frames.back();
        st |= 1;                                                      // the packet decodes (mode level)
        if (f.ov_len == 0 || f.start + f.ov_len <= f.valid) st |= 2;  // its overlap stays out of its own tail
      }
    }
    if (one.has_position()) st |= 4;
    if (one.eos()) st |= 8;  // _eosFound: the serial decoder pulls nothing after this packet
    if (position_after) position_after[i] = one.position();
    if (emitted_after) emitted_after[i] = one.emitted();
    if (state_after) state_after[i] = st;
    if (fb.frames.size() >= 1024) {
      fb.clear();
      one.begin_batch();
    }
  }
  if (total_emitted) {
    int rc = one.push_end(fb);  // the provider runs dry: the last block's tail is drained unless _eosFound (StreamDecoder.cs:352-356)
    if (rc != NVH_OK) return rc;
    *total_emitted = one.emitted();
  }
  return NVH_OK;
}

// StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647) -> Mode.GetPacketSampleCount (Mode.cs:172-176): the number of
// samples a packet stands for in the page granule arithmetic of the reference's seek (Ogg/PacketProvider.cs:74-146).
extern "C" int nvh_stream_packet_sample_count(const nvh_stream* s, const uint8_t* pkt, int len, int is_resync, int* count) {
  if (!s || !count || (!pkt && len > 0) || len < 0) return NVH_ERR_ARGUMENT;
  *count = 0;
  if (is_resync) return NVH_OK;  // a resync packet carries no audio data to return
  static const uint8_t empty = 0;
  nvh::BitReader p(pkt ? pkt : &empty, len);
  if (p.read_bit()) return NVH_OK;  // not an audio packet
  const int mode_idx = (int)p.read(s->setup.mode_field_bits);
  if (mode_idx < 0 || mode_idx >= (int)s->setup.modes.size()) return NVH_OK;
  const nvh::Mode& m = s->setup.modes[(size_t)mode_idx];
  if (p.is_short) return NVH_OK;  // Mode.GetPacketInfo: IsShort is looked at before the flag bits (Mode.cs:121-128)
  if (m.block_flag) {
    const bool prev_flag = p.read_bit();
    const bool next_flag = p.read_bit();
    const int wi = (prev_flag ? 1 : 0) + (next_flag ? 2 : 0);
    *count = m.ov_valid[wi] - m.ov_start[wi];
  } else {
    *count = m.block_size / 2;
  }
  return NVH_OK;
}

// ResetDecoder (StreamDecoder.cs:295-305): forget the previous block, the position, end of stream and the clipped flag;
// the next packet pushed is a "first packet" again.  Pending frames are dropped.
extern "C" int nvh_stream_reset(nvh_stream* s) {
  if (!s) return NVH_ERR_ARGUMENT;
  s->pending.clear();
  s->parser.reset(new (std::nothrow) nvh::StreamParser(&s->setup));
  if (!s->parser) return NVH_ERR_NOMEM;
  s->parser->set_light(s->gpu_parse);
  s->has_clipped = 0;
  if (s->ctx && s->flags.p) {
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), s->ctx->stream));
  }
  return NVH_OK;
}

extern "C" int nvh_stream_drop_pending(nvh_stream* s) {
  if (!s) return NVH_ERR_ARGUMENT;
  s->pending.clear();
  s->parser->begin_batch();  // later frames refer to the previous block as a carried tail
  return NVH_OK;
}

extern "C" int nvh_stream_push_packet(nvh_stream* s, const uint8_t* data, int len, int64_t granule, int flags) {
  if (!s || (!data && len > 0) || len < 0) return NVH_ERR_ARGUMENT;
  static const uint8_t empty = 0;
  return s->parser->push_packet(data ? data : &empty, len, granule, flags, s->pending);
}

extern "C" int nvh_stream_push_packets(nvh_stream* s, const uint8_t* bytes, const int64_t* offsets, const int64_t* granules,
                                       const uint8_t* flags, int n, int max_packets, int* consumed) {
  if (!s || !bytes || !offsets || !consumed || n < 0) return NVH_ERR_ARGUMENT;
  static const uint8_t empty = 0;
  int i = 0;
  // the look-ahead loop of a batched caller: stop when the quota is used up or once the stream has seen its
  // end-of-stream packet (StreamDecoder.cs:343-350: no more packets are pulled after _eosFound)
  for (; i < n && i < max_packets && !s->parser->eos(); i++) {
    const int64_t len = offsets[i + 1] - offsets[i];
    if (len < 0 || len > 0x7FFFFFFF) return NVH_ERR_ARGUMENT;
    int rc = s->parser->push_packet(len ? bytes + offsets[i] : &empty, (int)len, granules ? granules[i] : -1,
                                    flags ? (int)flags[i] : 0, s->pending);
    if (rc != NVH_OK) {
      *consumed = i;
      return rc;
    }
  }
  *consumed = i;
  return NVH_OK;
}

extern "C" int nvh_stream_push_end(nvh_stream* s) {
  if (!s) return NVH_ERR_ARGUMENT;
  return s->parser->push_end(s->pending);
}

extern "C" int nvh_stream_pending_geometry(const nvh_stream* s, int32_t* out, int cap_frames) {
  if (!s || !out) return NVH_ERR_ARGUMENT;
  int n = (int)s->pending.frames.size();
  if (cap_frames < n) return NVH_ERR_ARGUMENT;
  for (int i = 0; i < n; i++) {
    const NvhFrame& f = s->pending.frames[(size_t)i];
    int32_t* o = out + (size_t)i * 8;
    o[0] = f.n; o[1] = f.start; o[2] = f.valid; o[3] = f.total;
    o[4] = f.emit_start; o[5] = f.emit_count; o[6] = f.ov_frame; o[7] = f.ov_len;
  }
  return NVH_OK;
}

extern "C" int nvh_stream_pending(const nvh_stream* s, int* frames, int64_t* pcm_samples) {
  if (!s) return NVH_ERR_ARGUMENT;
  if (frames) *frames = (int)s->pending.frames.size();
  if (pcm_samples) *pcm_samples = s->pending.pcm_samples;
  return NVH_OK;
}

// ------------------------------------------------------------------------------------------------
// batches
// ------------------------------------------------------------------------------------------------

// Moves s->pending into `b` (device resident) and advances the stream's batch boundary.
// Device tables of the GPU packet parser (kernels_parse.hip) and the worst-case slab capacities of this setup.
// Streams outside its limits (Floor0, > 8 channels, ...) simply keep the host parser.
static int upload_parse_tables(nvh_stream* s) {
  const nvh::Setup& S = s->setup;
  SharedSetup& sh = *s->shared;
  sh.gpu_parse_ok = false;
  if (S.channels > NVH_PARSE_MAX_CH || S.books.size() > 256) return NVH_OK;
  for (const nvh::Floor& f : S.floors)
    if (f.type != 1) return NVH_OK;
  for (const nvh::Mapping& m : S.mappings)
    if (m.submap_floor.size() > NVH_PARSE_MAX_SUBMAPS || m.coupling_angle.size() > NVH_PARSE_MAX_COUPLING) return NVH_OK;

  std::vector<NvhPBook> books(S.books.size());
  std::vector<uint32_t> prefix;
  std::vector<NvhPOverflow> overflow;
  for (size_t i = 0; i < S.books.size(); i++) {
    const nvh::Codebook& b = S.books[i];
    NvhPBook& d = books[i];
    std::memset(&d, 0, sizeof d);
    if (b.entries > 0xFFFFFF || b.prefix_bits > 16 || b.max_bits > 32 || b.dimensions > 0xFFFF) return NVH_OK;
    d.prefix_off = (uint32_t)prefix.size();
    d.ovf_off = (uint32_t)overflow.size();
    d.entries = (uint32_t)b.entries;
    d.dims = (uint16_t)b.dimensions;
    d.prefix_bits = (uint8_t)b.prefix_bits;
    d.max_bits = (uint8_t)b.max_bits;
    d.has_tree = b.has_tree ? 1 : 0;
    d.has_overflow = b.has_overflow ? 1 : 0;
    // prefix[slot]: a short code, or (for slots only longer codes start with) that slot's group of overflow nodes
    //   present: (value << 8) | 0x80 | length        absent: (group begin << 8) | group count (0x7F = scan the whole list)
    for (size_t k = 0; k < b.prefix.size(); k++) {
      const nvh::HuffNode& n = b.prefix[k];
      if (n.present) {
        if (n.length < 0 || n.length > 0x7F || n.value < 0 || n.value > 0xFFFFFF) return NVH_OK;
        prefix.push_back(((uint32_t)n.value << 8) | 0x80u | (uint32_t)n.length);
      } else {
        uint32_t g = b.has_overflow && k < b.slot_group.size() ? b.slot_group[k] : 0u;
        uint32_t cnt = g & 0xFFu, beg = g >> 8;
        if (cnt >= 0x7Fu || beg > 0xFFFFFFu) {  // oversized group (or the host's own fallback marker): plain scan
          cnt = 0x7Fu;
          beg = 0;
        }
        prefix.push_back((beg << 8) | cnt);
      }
    }
    if (b.prefix.empty()) prefix.push_back(0u);  // has_tree == false: never indexed, keeps offsets valid
    // overflow pool of this book: the whole list in the reference's order, then the same nodes grouped by slot
    auto put = [&](const nvh::HuffNode& n) {
      NvhPOverflow o;
      o.bits = (uint32_t)n.bits;
      o.mask = (uint32_t)n.mask;
      o.value = (uint32_t)n.value;
      o.length = (uint32_t)n.length;
      overflow.push_back(o);
    };
    for (const nvh::HuffNode& n : b.overflow) put(n);
    for (const nvh::HuffNode& n : b.overflow_grouped) put(n);
    d.ovf_count = (uint32_t)b.overflow.size();
  }
  // LDS image: residue VQ books first (most symbols of a packet), then class books, then floor books, while they fit
  std::vector<uint32_t> lds_image;
  {
    const size_t budget = 13 * 1024;  // words (52 KB; + <= 8 KB of book / floor / residue / mapping records): two workgroups per CU
    for (auto& d : books) d.lds_off = 0xFFFFFFFFu;
    std::vector<int> order;
    std::vector<char> seen(S.books.size(), 0);
    auto want = [&](int b) {
      if (b >= 0 && b < (int)S.books.size() && !seen[(size_t)b]) {
        seen[(size_t)b] = 1;
        order.push_back(b);
      }
    };
    for (const nvh::Residue& r : S.residues)
      for (int c = 0; c < r.classifications && c < NVH_MAX_CLASSES; c++)
        for (int k = 0; k < NVH_MAX_STAGES; k++) want(r.books[c][k]);
    for (const nvh::Residue& r : S.residues) want(r.class_book);
    for (const nvh::Floor& fl : S.floors)
      for (int k = 0; k < 16; k++) {
        want(fl.f1.class_masterbook[k]);
        for (int j = 0; j < 8; j++) want(fl.f1.subclass_book[k][j]);
      }
    for (int b : order) {
      const size_t n = S.books[(size_t)b].prefix.size();
      if (n == 0 || lds_image.size() + n > budget) continue;
      books[(size_t)b].lds_off = (uint32_t)lds_image.size();
      lds_image.insert(lds_image.end(), prefix.begin() + books[(size_t)b].prefix_off, prefix.begin() + books[(size_t)b].prefix_off + n);
    }
    if (lds_image.empty()) lds_image.push_back(0u);
  }
  std::vector<NvhPFloor1> floors(S.floors.size());
  for (size_t i = 0; i < S.floors.size(); i++) {
    const nvh::Floor1& f = S.floors[i].f1;
    NvhPFloor1& d = floors[i];
    std::memset(&d, 0, sizeof d);
    d.type = 1;
    d.partition_count = f.partition_count;
    d.y_bits = f.y_bits;
    for (int k = 0; k < 32; k++) d.partition_class[k] = (uint8_t)f.partition_class[k];
    for (int k = 0; k < 16; k++) {
      d.class_dims[k] = (uint8_t)f.class_dimensions[k];
      d.class_sub_bits[k] = (uint8_t)f.class_subclasses[k];
      d.class_master[k] = (int16_t)f.class_masterbook[k];
      for (int j = 0; j < 8; j++) d.sub_book[k][j] = (int16_t)f.subclass_book[k][j];
    }
  }
  std::vector<int32_t> ipool;
  std::vector<NvhPResidue> residues(S.residues.size());
  std::vector<int> r_parts(S.residues.size()), r_ops(S.residues.size()), r_ent(S.residues.size());
  int cap_parts = 1;
  for (size_t i = 0; i < S.residues.size(); i++) {
    const nvh::Residue& r = S.residues[i];
    NvhPResidue& d = residues[i];
    std::memset(&d, 0, sizeof d);
    d.type = r.type; d.begin = r.begin; d.end = r.end; d.partition_size = r.partition_size;
    d.classifications = r.classifications; d.class_book = r.class_book; d.channels = r.channels;
    d.real_channels = r.real_channels; d.max_stages = r.max_stages; d.partvals = r.partvals;
    d.class_dims = S.books[(size_t)r.class_book].dimensions;
    d.decode_map_off = (uint32_t)ipool.size();
    ipool.insert(ipool.end(), r.decode_map.begin(), r.decode_map.end());
    int min_dims = 1 << 30;
    for (int c = 0; c < NVH_MAX_CLASSES; c++) {
      d.cascade[c] = (uint8_t)r.cascade[c];
      for (int k = 0; k < NVH_MAX_STAGES; k++) {
        d.books[c][k] = (int16_t)r.books[c][k];
        if (c < r.classifications && r.books[c][k] >= 0) {
          const int dm = S.books[(size_t)r.books[c][k]].dimensions;
          if (dm > 0 && dm < min_dims) min_dims = dm;
        }
      }
    }
    if (min_dims == (1 << 30)) min_dims = 1;
    // worst case over this setup's largest block: every partition of every channel has a book in every stage
    const int bs = r.type == 2 ? S.block1 * r.real_channels : S.block1;
    const int end = r.end < bs / 2 ? r.end : bs / 2;
    const int n = end - r.begin;
    const int parts = (n > 0 && r.partition_size > 0) ? n / r.partition_size : 0;
    const int cdim = d.class_dims > 0 ? d.class_dims : 1;
    const int words = (parts + cdim - 1) / cdim;
    r_parts[i] = parts;
    r_ops[i] = r.max_stages * parts * r.channels;
    r_ent[i] = r_ops[i] * ((r.partition_size + min_dims - 1) / min_dims);
    const int need = r.channels * std::max(std::max(parts, words), 1);
    if (need > cap_parts) cap_parts = need;
  }
  if (ipool.empty()) ipool.push_back(0);
  std::vector<NvhPMapping> mappings(S.mappings.size());
  int cap_ops = 1, cap_ent = 8, cap_pass = 1;
  for (size_t i = 0; i < S.mappings.size(); i++) {
    const nvh::Mapping& m = S.mappings[i];
    NvhPMapping& d = mappings[i];
    std::memset(&d, 0, sizeof d);
    d.submaps = (int32_t)m.submap_floor.size();
    d.coupling_steps = (int32_t)m.coupling_angle.size();
    int ops = 0, ent = 0;
    for (size_t k = 0; k < m.submap_floor.size(); k++) {
      d.submap_floor[k] = (uint8_t)m.submap_floor[k];
      d.submap_residue[k] = (uint8_t)m.submap_residue[k];
      ops += r_ops[(size_t)m.submap_residue[k]];
      ent += r_ent[(size_t)m.submap_residue[k]];
    }
    for (int c = 0; c < S.channels; c++) {
      d.chan_floor[c] = (uint8_t)m.channel_floor[(size_t)c];
      d.chan_residue[c] = (uint8_t)m.channel_residue[(size_t)c];
    }
    for (size_t k = 0; k < m.coupling_angle.size(); k++) {
      d.coupling_ang[k] = (uint8_t)m.coupling_angle[k];
      d.coupling_mag[k] = (uint8_t)m.coupling_magnitude[k];
    }
    if (ops > cap_ops) cap_ops = ops;
    if (ent > cap_ent) cap_ent = ent;
    if (d.submaps > cap_pass) cap_pass = d.submaps;
  }
  cap_ops = (cap_ops + 7) & ~7;
  cap_ent = (cap_ent + 15) & ~7;
  // keep a frame's slabs within reason (and op indices within the 15-bit links where possible)
  if ((size_t)cap_ops * 10 + (size_t)cap_ent * 2 + (size_t)cap_parts * 8 > ((size_t)1 << 20)) return NVH_OK;

  ArenaBuilder ab;
  // books | floors | residues | mappings back to back: k_parse copies this block into LDS
  size_t o_bk = ab.add(books.data(), books.size() * sizeof(NvhPBook));
  size_t o_fl = ab.add(floors.data(), floors.size() * sizeof(NvhPFloor1));
  size_t o_rs = ab.add(residues.data(), residues.size() * sizeof(NvhPResidue));
  size_t o_mp = ab.add(mappings.data(), mappings.size() * sizeof(NvhPMapping));
  const size_t meta_end = (ab.bytes.size() + 15) / 16 * 16;
  size_t o_px = ab.add(prefix.data(), prefix.size() * sizeof(uint32_t));
  NvhPOverflow none{};
  size_t o_ov = ab.add(overflow.empty() ? &none : overflow.data(), (overflow.empty() ? 1 : overflow.size()) * sizeof(NvhPOverflow));
  size_t o_ip = ab.add(ipool.data(), ipool.size() * sizeof(int32_t));
  size_t o_li = ab.add(lds_image.data(), lds_image.size() * sizeof(uint32_t));
  if (meta_end - o_bk > 8 * 1024) return NVH_OK;  // unusually large setup: keep the host parser
  sh.parse_arena.pool = &s->ctx->pool;
  int rc = sh.parse_arena.reserve(ab.bytes.size());
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpy(sh.parse_arena.p, ab.bytes.data(), ab.bytes.size(), hipMemcpyHostToDevice));
  const uint8_t* base = (const uint8_t*)sh.parse_arena.p;
  NvhDevParse& P = sh.parse;
  P.channels = S.channels;
  P.block1 = S.block1;
  P.cap_pass = cap_pass;
  P.cap_ops = cap_ops;
  P.cap_ent = cap_ent;
  P.cap_parts = cap_parts;
  P.books = (const NvhPBook*)(base + o_bk);
  P.prefix = (const uint32_t*)(base + o_px);
  P.overflow = (const NvhPOverflow*)(base + o_ov);
  P.floors = (const NvhPFloor1*)(base + o_fl);
  P.residues = (const NvhPResidue*)(base + o_rs);
  P.mappings = (const NvhPMapping*)(base + o_mp);
  P.ipool = (const int32_t*)(base + o_ip);
  P.lds_image = (const uint32_t*)(base + o_li);
  P.lds_words = (int32_t)lds_image.size();
  P.meta_words = (int32_t)((meta_end - o_bk) / 4);
  P.meta_floors_off = (int32_t)(o_fl - o_bk);
  P.meta_residues_off = (int32_t)(o_rs - o_bk);
  P.meta_mappings_off = (int32_t)(o_mp - o_bk);
  P.pad = 0;
  sh.gpu_parse_ok = true;
  return NVH_OK;
}

static int collect_parse_result(nvh_stream* s, nvh_batch* b, const NvhParseResult* d_res);

// GPU-parse mode: upload frame geometry + packets, let k_parse produce the descriptors into per-frame slabs.
static int batch_upload_gpu(nvh_stream* s, nvh_batch* b) {
  nvh::FrameBatch& P = s->pending;
  const NvhDevParse& T = s->shared->parse;
  const int ch = s->setup.channels;
  const size_t nf = P.frames.size();
  P.pkt_refs.resize(nf);  // trailing pseudo-frames
  if (P.pkt_pool.empty()) P.pkt_pool.resize(8, 0);
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  // host-written prefix of the blob ...
  const size_t o_fr = 0;
  const size_t o_ch = al(o_fr + std::max<size_t>(nf, 1) * sizeof(NvhFrame));
  const size_t o_rf = al(o_ch + std::max<size_t>(nf * ch, 1) * sizeof(NvhChan));
  const size_t o_pk = al(o_rf + std::max<size_t>(nf, 1) * sizeof(NvhPacketRef));
  const size_t host_bytes = al(o_pk + P.pkt_pool.size() + 8);
  // ... and the device-only slabs behind it
  const size_t o_ps = host_bytes;
  const size_t o_op = al(o_ps + nf * (size_t)T.cap_pass * sizeof(NvhResPass));
  const size_t o_lk = al(o_op + nf * (size_t)T.cap_ops * sizeof(NvhResOp));
  const size_t o_en = al(o_lk + nf * (size_t)T.cap_ops * sizeof(uint16_t));
  const size_t o_po = al(o_en + nf * (size_t)T.cap_ent * sizeof(uint16_t) + 64);
  const size_t o_sc = al(o_po + nf * (size_t)ch * NVH_MAX_POSTS * sizeof(uint16_t));
  const size_t o_rs = al(o_sc + nf * 2 * (size_t)T.cap_parts * sizeof(int));
  const size_t total = al(o_rs + sizeof(NvhParseResult));
  int rc = b->blob.reserve(total);
  if (rc != NVH_OK) return rc;
  if ((rc = b->h_blob.reserve(host_bytes)) != NVH_OK) return rc;
  uint8_t* h = (uint8_t*)b->h_blob.p;
  if (nf) std::memcpy(h + o_fr, P.frames.data(), nf * sizeof(NvhFrame));
  if (nf) std::memcpy(h + o_ch, P.chans.data(), std::min(P.chans.size(), nf * (size_t)ch) * sizeof(NvhChan));
  if (nf) std::memcpy(h + o_rf, P.pkt_refs.data(), nf * sizeof(NvhPacketRef));
  std::memcpy(h + o_pk, P.pkt_pool.data(), P.pkt_pool.size());
  std::memset(h + o_pk + P.pkt_pool.size(), 0, 8);
  b->descriptor_bytes = (int64_t)(nf * (sizeof(NvhFrame) + sizeof(NvhPacketRef)) + P.pkt_pool.size());
  hipStream_t st = s->ctx->stream;
  uint8_t* base = (uint8_t*)b->blob.p;
  HIP_TRY(hipMemcpyAsync(base, h, host_bytes, hipMemcpyHostToDevice, st));
  NvhParseResult init{};
  init.err_frame = 0x7FFFFFFF;
  init.links_ok = 1;
  // (a 32-byte pageable source: staged by the runtime before the call returns)
  HIP_TRY(hipMemcpyAsync(base + o_rs, &init, sizeof init, hipMemcpyHostToDevice, st));
  b->dev.frames = (const NvhFrame*)(base + o_fr);
  b->dev.chans = (const NvhChan*)(base + o_ch);
  b->dev.passes = (const NvhResPass*)(base + o_ps);
  b->dev.ops = (const NvhResOp*)(base + o_op);
  b->dev.op_link = (const uint16_t*)(base + o_lk);
  b->dev.entries = (const uint16_t*)(base + o_en);
  b->dev.posts = (const uint16_t*)(base + o_po);
  b->dev.coeffs = (const float*)(base + o_po);  // no Floor0 in this mode
  b->dev.nframes = b->nframes;
  b->dev.pad = 0;
  if (nf) {
    const unsigned blocks = (unsigned)((nf + 63) / 64);
    // Launch shape.  Packets take different paths through the parser, so the lanes of a wavefront mostly run one after
    // the other, and one wavefront alone issues an instruction every ~5 cycles at best: aim at ~2 wavefronts per SIMD
    // (2048 in all) with as few packets each as that allows; workgroups of 4 wavefronts, two per CU (LDS tables).
    static const int lanes_env = getenv("NVH_PARSE_LANES") ? atoi(getenv("NVH_PARSE_LANES")) : 0;
    static const int waves_env = getenv("NVH_PARSE_WAVES") ? atoi(getenv("NVH_PARSE_WAVES")) : 0;
    const int kParseWaves = (waves_env >= 1 && waves_env <= 16) ? waves_env : 4;
    int lanes = 1;
    while (lanes < 64 && (nf + (size_t)lanes - 1) / (size_t)lanes > 2048) lanes *= 2;
    if (lanes_env >= 1 && lanes_env <= 64) lanes = lanes_env;
    const size_t per_wg = (size_t)kParseWaves * (size_t)lanes;
    const unsigned pblocks = (unsigned)((nf + per_wg - 1) / per_wg);
    // per-lane LDS next to the tables, while two workgroups still fit a CU (2 x 80 KB): the residue scratch rows first,
    // then the packets (sized for the longest packet of the batch)
    size_t max_pkt_words = 1;
    for (size_t i = 0; i < nf; i++) max_pkt_words = std::max<size_t>(max_pkt_words, ((size_t)P.pkt_refs[i].bit_len + 31) / 32 + 1);
    const size_t table_words = (size_t)(T.lds_words + T.meta_words);
    const size_t lds_cap_words = 80 * 1024 / 4;
    int scratch_words = 2 * T.cap_parts, pkt_words = (int)max_pkt_words;
    // LDS variant only when both the rows and the longest packet of the batch fit for every lane; else everything per-lane
    // stays in global memory (k_parse_g)
    const bool in_lds = table_words + per_wg * (size_t)(scratch_words + pkt_words) <= lds_cap_words;
    if (!in_lds) scratch_words = pkt_words = 0;
    const size_t parse_lds = (table_words + per_wg * (size_t)(scratch_words + pkt_words)) * sizeof(uint32_t);
    static bool lds_attr_set = false;
    if (!lds_attr_set) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_g, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
      lds_attr_set = true;
    }
    hipLaunchKernelGGL(in_lds ? k_parse : k_parse_g, dim3(pblocks), dim3(64 * kParseWaves), parse_lds, st, T,
                       (const uint8_t*)(base + o_pk), (const NvhPacketRef*)(base + o_rf),
                       (int)nf, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch), (NvhResPass*)(base + o_ps), (NvhResOp*)(base + o_op),
                       (uint16_t*)(base + o_lk), (uint16_t*)(base + o_en), (uint16_t*)(base + o_po), (int*)(base + o_sc),
                       (NvhParseResult*)(base + o_rs), lanes, scratch_words, pkt_words);
    // the carried block's execute flags ping-pong together with the carried block (nvh_stream_synth flips carry_cur)
    uint32_t* ce = (uint32_t*)s->carry_exec.p;
    hipLaunchKernelGGL(k_parse_links, dim3(blocks), dim3(64), 0, st, (int)nf, ch, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch),
                       (const uint32_t*)(ce + s->carry_cur), ce + (s->carry_cur ^ 1), b->last_decoded);
    HIP_TRY(hipGetLastError());
  }
  rc = collect_parse_result(s, b, (const NvhParseResult*)(base + o_rs));
  if (rc != NVH_OK) return rc;
  size_t plane = (size_t)ch * (size_t)s->setup.block1 * sizeof(float);
  rc = b->work.reserve(std::max<size_t>((size_t)b->nframes, 1) * plane);
  if (rc != NVH_OK) return rc;
  P.clear();
  s->parser->begin_batch();
  return NVH_OK;
}

// Reads k_parse's batch-level result back (one small copy + synchronisation): sizes the LDS staging of the
// spectrum kernel and reports the first packet the reference would have thrown on.
static int collect_parse_result(nvh_stream* s, nvh_batch* b, const NvhParseResult* d_res) {
  hipStream_t st = s->ctx->stream;
  int rc = s->h_pcm.reserve(sizeof(NvhParseResult));
  if (rc != NVH_OK) return rc;
  NvhParseResult* r = (NvhParseResult*)s->h_pcm.p;
  HIP_TRY(hipMemcpyAsync(r, d_res, sizeof *r, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  b->max_ops = r->max_ops;
  b->max_ent = r->max_ent;
  b->max_pass = r->max_pass;
  b->links_ok = r->links_ok != 0;
  if (r->err_frame != 0x7FFFFFFF) {
    // some packet of the batch would have made the managed decoder throw (host parser: the same code from
    // nvh_stream_push_packet); the whole look-ahead batch is dropped
    s->pending.clear();
    s->parser->begin_batch();
    return r->err_code < 0 ? r->err_code : NVH_ERR_RUNTIME;
  }
  return NVH_OK;
}

static int batch_upload(nvh_stream* s, nvh_batch* b) {
  nvh::FrameBatch& P = s->pending;
  b->s = s;
  b->nframes = (int)P.frames.size();
  b->chan_frames = (int)P.chans.size();
  b->pcm_samples = P.pcm_samples;
  b->sequential_ola = P.sequential_ola;
  b->last_decoded = -1;
  b->max_ops = b->max_ent = b->max_pass = 0;
  b->links_ok = P.links_ok && P.op_link.size() == P.ops.size();
  for (const NvhFrame& fr : P.frames) {
    if ((int)fr.op_count > b->max_ops) b->max_ops = (int)fr.op_count;
    if ((int)fr.ent_count > b->max_ent) b->max_ent = (int)fr.ent_count;
    if ((int)(fr.pass_end - fr.pass_begin) > b->max_pass) b->max_pass = (int)(fr.pass_end - fr.pass_begin);
    // kernels index channel records by frame: every frame owns exactly `channels` of them (host_parse.cpp)
    if (fr.chan_off != (uint32_t)((size_t)(&fr - P.frames.data()) * (size_t)s->setup.channels)) return NVH_ERR_RUNTIME;
  }
  {
    // preconditions of the fused IMDCT + overlap-add kernel (kernels_imdct.hip, k_imdct_ola)
    bool ok = !P.sequential_ola && s->setup.block0 >= 256 && s->setup.block1 <= 2048 && s->setup.channels <= 4;
    for (size_t i = 0; ok && i < P.frames.size(); i++) {
      const NvhFrame& fr = P.frames[i];
      if (fr.n == 0) {
        ok = fr.ov_frame == -2;
        continue;
      }
      if (fr.ov_len > 0) {
        const bool src_ok = fr.ov_frame == -2 || (fr.ov_frame == (int)i - 1 && P.frames[i - 1].n != 0);
        ok = src_ok && fr.start + fr.ov_len <= fr.n / 2 && fr.ov_src >= fr.ov_n / 2 && fr.ov_src + fr.ov_len <= fr.ov_n &&
             fr.ov_n >= 256 && fr.ov_n <= 2048;
      }
    }
    b->fused_ola = ok;
  }
  for (int i = b->nframes - 1; i >= 0; --i)
    if (P.frames[(size_t)i].n != 0) {
      b->last_decoded = i;
      break;
    }

  b->stats[0] = (int64_t)P.frames.size(); b->stats[1] = (int64_t)P.chans.size(); b->stats[2] = (int64_t)P.passes.size();
  b->stats[3] = (int64_t)P.ops.size(); b->stats[4] = (int64_t)P.entries.size(); b->stats[5] = (int64_t)P.posts.size();
  b->stats[6] = (int64_t)P.coeffs.size();
  if (s->gpu_parse) return batch_upload_gpu(s, b);
  // the descriptor arrays are laid out back to back (16-byte aligned) in one pinned staging block and go to the
  // device with one asynchronous copy; the caller decides when the stream is synchronised
  auto pad1 = [](size_t n) { return n ? n : (size_t)1; };
  static const uint8_t dummy[64] = {0};
  struct Piece { const void* src; size_t n, off; };
  std::vector<Piece> pieces;
  size_t total = 0;
  auto add = [&](const void* src, size_t count, size_t elem) {
    const size_t n = pad1(count) * elem;
    total = (total + 15) / 16 * 16;
    pieces.push_back({count ? src : (const void*)dummy, count ? n : (n < sizeof dummy ? n : sizeof dummy), total});
    total += n;
    return pieces.back().off;
  };
  size_t o_fr = add(P.frames.data(), P.frames.size(), sizeof(NvhFrame));
  size_t o_ch = add(P.chans.data(), P.chans.size(), sizeof(NvhChan));
  size_t o_ps = add(P.passes.data(), P.passes.size(), sizeof(NvhResPass));
  size_t o_op = add(P.ops.data(), P.ops.size(), sizeof(NvhResOp));
  size_t o_lk = add(P.op_link.data(), P.op_link.size(), sizeof(uint16_t));
  size_t o_en = add(P.entries.data(), P.entries.size(), sizeof(uint16_t));
  size_t o_po = add(P.posts.data(), P.posts.size(), sizeof(uint16_t));
  size_t o_co = add(P.coeffs.data(), P.coeffs.size(), sizeof(float));
  total += 64;  // k_spectrum copies entry slices in whole 16-byte vectors
  b->descriptor_bytes = (int64_t)(P.frames.size() * sizeof(NvhFrame) + P.chans.size() * sizeof(NvhChan) +
                                  P.passes.size() * sizeof(NvhResPass) + P.ops.size() * (sizeof(NvhResOp) + sizeof(uint16_t)) +
                                  P.entries.size() * 2 + P.posts.size() * 2 + P.coeffs.size() * 4);
  int rc = b->blob.reserve(total);
  if (rc != NVH_OK) return rc;
  if ((rc = b->h_blob.reserve(total)) != NVH_OK) return rc;
  for (const Piece& pc : pieces) std::memcpy((uint8_t*)b->h_blob.p + pc.off, pc.src, pc.n);
  hipStream_t st = s->ctx->stream;
  HIP_TRY(hipMemcpyAsync(b->blob.p, b->h_blob.p, total, hipMemcpyHostToDevice, st));
  const uint8_t* base = (const uint8_t*)b->blob.p;
  b->dev.frames = (const NvhFrame*)(base + o_fr);
  b->dev.chans = (const NvhChan*)(base + o_ch);
  b->dev.passes = (const NvhResPass*)(base + o_ps);
  b->dev.ops = (const NvhResOp*)(base + o_op);
  b->dev.op_link = (const uint16_t*)(base + o_lk);
  b->dev.entries = (const uint16_t*)(base + o_en);
  b->dev.posts = (const uint16_t*)(base + o_po);
  b->dev.coeffs = (const float*)(base + o_co);
  b->dev.nframes = b->nframes;
  b->dev.pad = 0;

  size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  rc = b->work.reserve(pad1((size_t)b->nframes) * plane);
  if (rc != NVH_OK) return rc;
  P.clear();
  s->parser->begin_batch();
  return NVH_OK;
}

static int batch_launch(nvh_batch* b, const float* carry, float* carry_out, float* d_pcm, bool timing, float* kernel_ms) {
  nvh_stream* s = b->s;
  hipStream_t st = s->ctx->stream;
  if (b->nframes == 0) return NVH_OK;
  const int ch = s->setup.channels;
  float* work = (float*)b->work.p;
  int* flags = (int*)s->flags.p;
  const size_t lds = (size_t)s->setup.block1 * sizeof(float);
  ScopedEvent sev[5];
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (timing)
    for (int k = 0; k < 5; k++) {
      int rc = sev[k].create();
      if (rc != NVH_OK) return rc;
      ev[k] = sev[k].e;
    }
  if (timing) HIP_TRY(hipEventRecord(ev[0], st));
  // compact hand-over (two independent quarters per block, windowed in the overlap kernel) whenever no overlap
  // ever modifies a tail (the in-place sequential form needs the full windowed blocks)
  static const int no_compact = getenv("NVH_NO_COMPACT") ? 1 : 0;
  static const int no_fused_ola = getenv("NVH_FUSED_OLA") ? 0 : 1;  // experimental run-based kernel: opt-in
  static const int no_fused_imdct = getenv("NVH_NO_FUSED_IMDCT") ? 1 : 0;
  const bool use_fused_ola = b->fused_ola && !no_fused_ola && !b->block_only;
  const bool compact = s->setup.block0 >= 256 && !b->sequential_ola && !no_compact && !use_fused_ola && !b->block_only;
  // spectrum + IMDCT in one kernel: pair-path / fused-tail streams with block sizes the single-pass wavefront IMDCT covers
  const bool fast = s->fast_spectrum && b->links_ok;  // k_spectrum proper (pair path by chain walk, fused tail)
  bool fuse_imdct = compact && fast && s->setup.block1 <= 2048 && !no_fused_imdct;  // and the LDS-resident path is taken (below)
  // Fused spectrum kernel when a frame's spectrum (+ staged side information) fits the default 64 KB dynamic
  // LDS window; LDS map in kernels_spectrum.hip.
  {
    const bool has_floor0 = s->has_floor0;
    // more than four channels without Floor0: 8 wavefronts per workgroup (k_spectrum_gen8), one floor scratch block each
    static const int no_gen8 = getenv("NVH_NO_GEN8") ? 1 : 0;
    const bool gen8 = !has_floor0 && !fast && ch > 4 && !no_gen8;
    const int scratch_blocks = gen8 ? (ch < 8 ? ch : 8) : (ch < 4 ? ch : 4);
    const size_t scratch_words = (size_t)scratch_blocks * NVH_SP_FLOOR_SCRATCH_WORDS;
    // staging capacities; the entry slice is copied from its enclosing 16-byte boundary (up to 7 entries of slack)
    int cap_pass = b->max_pass, cap_ops = (b->max_ops + 7) & ~7, cap_ent = (b->max_ent + 14) & ~7;
    const size_t staging_words = (size_t)s->setup.books.size() * 8 + (size_t)((s->dev.lattice_words + 3) & ~3) + (size_t)cap_pass * 16 +
                                 (size_t)cap_ops * 6 + (size_t)cap_ops / 2 + (size_t)cap_ent / 2;  // ops 2 + pair records 4 + links 1/2 words per op
    // k_spectrum_gen8 overlays the floor scratch on the staged side information (dead by the time the floors are prepared)
    size_t words = (size_t)(has_floor0 ? 512 : 256) + (gen8 ? std::max(scratch_words, staging_words) : scratch_words + staging_words) +
                   (size_t)ch * (size_t)(s->setup.block1 / 2);
    if (getenv("NVH_UNFUSED")) words = 1u << 20;  // test aid: force the unfused kernels below
    if ((words + (size_t)(s->setup.block1 / 16)) * 4 > 64 * 1024) fuse_imdct = false;
    static const int phase_mask = getenv("NVH_DEBUG_SPECTRUM_MASK") ? atoi(getenv("NVH_DEBUG_SPECTRUM_MASK")) : 7;  // profiling aid
    static const size_t lds_pad = getenv("NVH_LDS_PAD") ? (size_t)atoi(getenv("NVH_LDS_PAD")) : 0;  // occupancy experiments
    if (words * 4 <= 64 * 1024) {
      if (timing) HIP_TRY(hipEventRecord(ev[1], st));  // slot 0 stays empty: slot 1 = fused spectrum kernel
      b->slot_name[0] = "-";
      b->slot_name[1] = has_floor0 ? "k_spectrum_f0" : (fast ? "k_spectrum" : "k_spectrum_gen");
      if (has_floor0) {
        hipLaunchKernelGGL(k_spectrum_f0, dim3((unsigned)b->nframes), dim3(256), words * 4, st, s->dev, b->dev, work, flags,
                           cap_pass, cap_ops, cap_ent);
      } else if (gen8) {
        b->slot_name[1] = "k_spectrum_gen8";
        hipLaunchKernelGGL(k_spectrum_gen8, dim3((unsigned)b->nframes), dim3(512), words * 4, st, s->dev, b->dev, work, flags,
                           cap_pass, cap_ops, cap_ent, (long long*)g_dbg_buf);
      } else if (!fast) {
        hipLaunchKernelGGL(k_spectrum_gen, dim3((unsigned)b->nframes), dim3(256), words * 4, st, s->dev, b->dev, work, flags,
                           cap_pass, cap_ops, cap_ent, (long long*)g_dbg_buf);
      } else {
        if (getenv("NVH_DEBUG_OCC")) {
          int nb = -1;
          hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_spectrum, 256, words * 4 + lds_pad);
          hipFuncAttributes fa;
          (void)hipFuncGetAttributes(&fa, (const void*)k_spectrum);
          fprintf(stderr, "k_spectrum: lds %zu B, occupancy %d WG/CU (err %d), regs %d, static lds %zu, max dyn lds %d\n", words * 4 + lds_pad, nb,
                  (int)oe, fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
        }
        if (fuse_imdct) {
          // + the IMDCT padding of the last channel (n/16 floats past the spectrum area)
          b->slot_name[1] = "k_spectrum_imdct";
          hipLaunchKernelGGL(k_spectrum_imdct, dim3((unsigned)b->nframes), dim3(256), words * 4 + (size_t)(s->setup.block1 / 16) * 4 + lds_pad,
                             st, s->dev, b->dev, work, flags, cap_pass, cap_ops, cap_ent, (long long*)g_dbg_buf, phase_mask);
        } else {
          hipLaunchKernelGGL(k_spectrum, dim3((unsigned)b->nframes), dim3(256), words * 4 + lds_pad, st, s->dev, b->dev, work, flags,
                             cap_pass, cap_ops, cap_ent, (long long*)g_dbg_buf, phase_mask);
        }
      }
    } else {
      b->slot_name[0] = "k_residue"; b->slot_name[1] = "k_couple_floor";
      hipLaunchKernelGGL(k_residue, dim3((unsigned)b->nframes), dim3(256), 0, st, s->dev, b->dev, work, 1);
      if (timing) HIP_TRY(hipEventRecord(ev[1], st));
      hipLaunchKernelGGL(k_couple_floor, dim3((unsigned)b->nframes), dim3(256), 0, st, s->dev, b->dev, work, flags);
    }
  }
  if (timing) HIP_TRY(hipEventRecord(ev[2], st));
  static const int run_len_env = getenv("NVH_RUN_LEN") ? atoi(getenv("NVH_RUN_LEN")) : 0;
  const size_t plane_bytes = (size_t)ch * (size_t)s->setup.block1 * sizeof(float);
  if (use_fused_ola) {
    // one workgroup per run of frames, one wavefront per channel; keep >= ~2048 waves in flight
    int run_len = run_len_env > 0 ? run_len_env : 4;
    while (run_len > 1 && (long long)(b->nframes / run_len) * ch < 2048) run_len >>= 1;
    const int runs = (b->nframes + run_len - 1) / run_len;
    const size_t ola_lds = (size_t)ch * (wave_lds_bytes(s->setup.block1) + (size_t)(s->setup.block1 / 2) * sizeof(float));
    b->slot_name[2] = "k_imdct_ola"; b->slot_name[3] = "-";
    hipLaunchKernelGGL(k_imdct_ola, dim3((unsigned)runs), dim3((unsigned)(64 * ch)), ola_lds, st, s->dev, b->dev, (const float*)work,
                       carry, carry_out, d_pcm, s->clip, flags + 1, run_len, b->last_decoded);
    if (timing) HIP_TRY(hipEventRecord(ev[3], st));  // slot 2 = fused IMDCT+OLA, slot 3 empty
  } else {
    b->slot_name[2] = fuse_imdct ? "-" : compact ? "k_imdct_compact" : (s->setup.block0 >= 256 ? "k_imdct_wave" : "k_imdct_window");
    b->slot_name[3] = compact ? "k_ola_compact" : (!b->sequential_ola ? "k_ola_emit" : "k_ola_emit_seq");
    if (fuse_imdct)
      ;  // done inside k_spectrum_imdct
    else if (compact)
      hipLaunchKernelGGL(k_imdct_compact, dim3((unsigned)(b->nframes * ch)), dim3(64), wave_lds_bytes(s->setup.block1), st, s->dev,
                         b->dev, work);
    else if (s->setup.block0 >= 256)
      hipLaunchKernelGGL(k_imdct_wave, dim3((unsigned)(b->nframes * ch)), dim3(64), wave_lds_bytes(s->setup.block1), st, s->dev,
                         b->dev, work);
    else
      hipLaunchKernelGGL(k_imdct_window, dim3((unsigned)(b->nframes * ch)), dim3(256), lds, st, s->dev, b->dev, work);
    if (timing) HIP_TRY(hipEventRecord(ev[3], st));
    if (b->block_only) {
      b->slot_name[3] = "-";  // nvh_mode_decode: the caller wants the windowed blocks themselves
    } else if (compact) {
      // 128 lanes per frame: 21.4 us instead of 24.8 us on its own (more loads in flight per frame); with two batches
      // in flight it is a wash against 64, and 256 lanes start to take wave slots from the other batch's spectrum kernel
      static const int ola_env = getenv("NVH_OLA_THREADS") ? atoi(getenv("NVH_OLA_THREADS")) : 0;
      const int ola_threads = (ola_env == 64 || ola_env == 128 || ola_env == 256) ? ola_env : 128;
      hipLaunchKernelGGL(k_ola_compact, dim3((unsigned)b->nframes), dim3((unsigned)ola_threads), 0, st, s->dev, b->dev,
                         (const float*)work, carry, d_pcm, s->clip, flags + 1, carry_out, b->last_decoded);
    } else if (!b->sequential_ola)
      hipLaunchKernelGGL(k_ola_emit, dim3((unsigned)b->nframes), dim3(256), 0, st, s->dev, b->dev, (const float*)work, carry,
                         d_pcm, s->clip, flags + 1);
    else
      hipLaunchKernelGGL(k_ola_emit_seq, dim3(1), dim3(256), 0, st, s->dev, b->dev, work, carry, d_pcm, s->clip, flags + 1);
    // the last decoded block becomes the carried tail (StreamDecoder's _prevPacketBuf), always fully windowed
    if (!compact && !b->block_only && b->last_decoded >= 0 && carry_out)
      HIP_TRY(hipMemcpyAsync(carry_out, (const uint8_t*)b->work.p + (size_t)b->last_decoded * plane_bytes, plane_bytes,
                             hipMemcpyDeviceToDevice, st));
  }
  if (timing) HIP_TRY(hipEventRecord(ev[4], st));
  HIP_TRY(hipGetLastError());
  if (timing) {
    HIP_TRY(hipEventSynchronize(ev[4]));
    for (int k = 0; k < 4; k++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      kernel_ms[k] += ms;
    }
  }
  return NVH_OK;
}

static unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

extern "C" int nvh_window_apply(nvh_stream* s, int mode_index, int prev_flag, int next_flag, int batch, float* d_buf,
                                int64_t stride) {
  if (!s || mode_index < 0 || mode_index >= (int)s->setup.modes.size() || batch < 0 || (batch > 0 && !d_buf)) return NVH_ERR_ARGUMENT;
  const nvh::Mode& m = s->setup.modes[(size_t)mode_index];
  if (stride < m.block_size) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (batch == 0) return NVH_OK;
  HIP_TRY(hipSetDevice(s->ctx->device));
  // Mode.cs:135: the long-block window is chosen by the packet's two flag bits; a short-block mode has one window
  const int wi = m.block_flag ? ((prev_flag ? 1 : 0) + (next_flag ? 2 : 0)) : 0;
  hipLaunchKernelGGL(k_window_apply, dim3(grid_for((long long)batch * m.block_size)), dim3(256), 0, s->ctx->stream, d_buf,
                     s->dev.windows + m.window_off[wi], m.block_size, (long long)stride, batch);
  HIP_TRY(hipGetLastError());
  return NVH_OK;
}

extern "C" int nvh_overlap_buffers(nvh_ctx* c, const float* d_previous, float* d_next, int prev_start, int prev_stop,
                                   int next_start, int channels, int64_t plane_stride) {
  if (!c || !d_previous || !d_next || prev_start < 0 || next_start < 0 || channels <= 0) return NVH_ERR_ARGUMENT;
  const int len = prev_stop - prev_start;
  if (len <= 0) return NVH_OK;  // the reference's loop does not run
  if (prev_stop > plane_stride || (int64_t)next_start + len > plane_stride) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_overlap_buffers, dim3(grid_for((long long)channels * len)), dim3(256), 0, c->stream, d_previous, d_next,
                     prev_start, len, next_start, channels, (long long)plane_stride);
  HIP_TRY(hipGetLastError());
  return NVH_OK;
}

extern "C" int nvh_copy_buffer(nvh_ctx* c, const float* d_planes, int start, int count, int channels, int64_t plane_stride,
                               float* d_target, int clip, int* clipped) {
  if (!c || start < 0 || count < 0 || channels <= 0 || (int64_t)start + count > plane_stride) return NVH_ERR_ARGUMENT;
  if (clipped) *clipped = 0;
  if (count == 0) return NVH_OK;
  if (!d_planes || !d_target) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  DevBuf flag;
  flag.pool = &c->pool;
  int rc = flag.reserve(sizeof(int));
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_copy_buffer, dim3(grid_for((long long)channels * count)), dim3(256), 0, c->stream, d_planes, start, count,
                     channels, (long long)plane_stride, d_target, clip, (int*)flag.p);
  HIP_TRY(hipGetLastError());
  int h = 0;
  HIP_TRY(hipMemcpyAsync(&h, flag.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (clipped) *clipped = h ? 1 : 0;
  return NVH_OK;
}

// IFloor.Apply for the stream's floor `floor_index` on `batch` device vectors (see include/nvorbis_hip.h).
extern "C" int nvh_stream_floor_info(const nvh_stream* s, int floor_index, int* type, int* post_count, int* range) {
  if (!s || floor_index < 0 || floor_index >= (int)s->setup.floors.size()) return NVH_ERR_ARGUMENT;
  const nvh::Floor& f = s->setup.floors[(size_t)floor_index];
  if (type) *type = f.type;
  if (post_count) *post_count = f.type == 1 ? (int)f.f1.x_list.size() : f.f0.order;
  if (range) *range = f.type == 1 ? f.f1.range : 0;
  return NVH_OK;
}

extern "C" int nvh_floor0_apply(nvh_stream* s, int floor_index, int block_size, int batch, const float* amps, const float* coeffs,
                                int coeff_stride, float* d_residue, int64_t stride, int32_t* status) {
  if (!s || batch < 0 || (batch > 0 && (!amps || !coeffs || !d_residue))) return NVH_ERR_ARGUMENT;
  if (floor_index < 0 || floor_index >= (int)s->setup.floors.size()) return NVH_ERR_ARGUMENT;
  const nvh::Floor& f = s->setup.floors[(size_t)floor_index];
  if (f.type != 0 || f.f0.order > 256 || coeff_stride < f.f0.order) return NVH_ERR_ARGUMENT;
  if (block_size != s->setup.block0 && block_size != s->setup.block1) return NVH_ERR_ARGUMENT;
  if (stride < block_size / 2) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (batch == 0) return NVH_OK;
  HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  DevBuf d_amps, d_coeffs, d_status;
  d_amps.pool = d_coeffs.pool = d_status.pool = &s->ctx->pool;
  const size_t ncoef = (size_t)batch * (size_t)coeff_stride;
  int rc;
  if ((rc = d_amps.reserve((size_t)batch * sizeof(float))) != NVH_OK) return rc;
  if ((rc = d_coeffs.reserve(ncoef * sizeof(float))) != NVH_OK) return rc;
  if ((rc = d_status.reserve((size_t)batch * sizeof(int32_t))) != NVH_OK) return rc;
  HIP_TRY(hipMemcpyAsync(d_amps.p, amps, (size_t)batch * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_coeffs.p, coeffs, ncoef * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_status.p, 0, (size_t)batch * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_floor0_apply, dim3((unsigned)batch), dim3(256), 0, st, s->dev, floor_index, (const float*)d_amps.p,
                     (const float*)d_coeffs.p, coeff_stride, block_size, d_residue, (long long)stride, (int*)d_status.p);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> h_status((size_t)batch, 0);
  HIP_TRY(hipMemcpyAsync(h_status.data(), d_status.p, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int any = NVH_OK;
  for (int b = 0; b < batch; ++b) {
    const int code = h_status[(size_t)b] ? NVH_ERR_RUNTIME : NVH_OK;  // wMap index out of range (Floor0.cs:90, :163)
    if (status) status[b] = code;
    if (code != NVH_OK && any == NVH_OK) any = code;
  }
  return status ? NVH_OK : any;
}

extern "C" int nvh_stream_mode_info(const nvh_stream* s, int mode_index, int* block_flag, int* block_size, int* mapping) {
  if (!s || mode_index < 0 || mode_index >= (int)s->setup.modes.size()) return NVH_ERR_ARGUMENT;
  const nvh::Mode& m = s->setup.modes[(size_t)mode_index];
  if (block_flag) *block_flag = m.block_flag ? 1 : 0;
  if (block_size) *block_size = m.block_size;
  if (mapping) *mapping = m.mapping;
  return NVH_OK;
}

extern "C" int nvh_floor1_apply(nvh_stream* s, int floor_index, int block_size, int batch, const int32_t* posts,
                                const int32_t* post_counts, float* d_residue, int64_t stride, int32_t* status) {
  if (!s || batch < 0 || (batch > 0 && (!posts || !post_counts || !d_residue))) return NVH_ERR_ARGUMENT;
  if (floor_index < 0 || floor_index >= (int)s->setup.floors.size()) return NVH_ERR_ARGUMENT;
  const nvh::Floor& f = s->setup.floors[(size_t)floor_index];
  if (f.type != 1) return NVH_ERR_ARGUMENT;
  if (block_size != s->setup.block0 && block_size != s->setup.block1) return NVH_ERR_ARGUMENT;
  if (stride < block_size / 2) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (batch == 0) return NVH_OK;
  const int pc = (int)f.f1.x_list.size();
  if (pc > NVH_MAX_POSTS) return NVH_ERR_RUNTIME;  // Data.Posts = new int[64] (Floor1.cs:12): Unpack itself throws for such a floor
  // Unpack leaves either no posts or all of them (Floor1.cs:135-184); the values are sums of codebook entries
  std::vector<uint16_t> h_posts((size_t)batch * NVH_MAX_POSTS, 0);
  for (int b = 0; b < batch; ++b) {
    if (post_counts[b] != 0 && post_counts[b] != pc) return NVH_ERR_ARGUMENT;
    for (int i = 0; i < post_counts[b]; ++i) {
      const int32_t v = posts[(size_t)b * NVH_MAX_POSTS + i];
      if (v < 0 || v > 0xFFFF) return NVH_ERR_ARGUMENT;
      h_posts[(size_t)b * NVH_MAX_POSTS + i] = (uint16_t)v;
    }
  }
  HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  DevBuf d_posts, d_counts, d_status;
  d_posts.pool = d_counts.pool = d_status.pool = &s->ctx->pool;
  int rc;
  if ((rc = d_posts.reserve(h_posts.size() * sizeof(uint16_t))) != NVH_OK) return rc;
  if ((rc = d_counts.reserve((size_t)batch * sizeof(int32_t))) != NVH_OK) return rc;
  if ((rc = d_status.reserve((size_t)batch * sizeof(int32_t))) != NVH_OK) return rc;
  HIP_TRY(hipMemcpyAsync(d_posts.p, h_posts.data(), h_posts.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d_counts.p, post_counts, (size_t)batch * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(d_status.p, 0, (size_t)batch * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_floor1_apply, dim3((unsigned)batch), dim3(64), 0, st, s->dev, floor_index, (const uint16_t*)d_posts.p,
                     (const int32_t*)d_counts.p, block_size, d_residue, (long long)stride, (int*)d_status.p);
  HIP_TRY(hipGetLastError());
  std::vector<int32_t> h_status((size_t)batch, 0);
  HIP_TRY(hipMemcpyAsync(h_status.data(), d_status.p, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int any = NVH_OK;
  for (int b = 0; b < batch; ++b) {
    // inverse_dB_table index out of range: IndexOutOfRangeException in the reference (quirk B-7)
    const int code = h_status[(size_t)b] ? NVH_ERR_RUNTIME : NVH_OK;
    if (status) status[b] = code;
    if (code != NVH_OK && any == NVH_OK) any = code;
  }
  return status ? NVH_OK : any;
}

// Reads and clears the device error / clipped words; maps device errors to status codes.
static int collect_flags(nvh_stream* s) {
  int h[2] = {0, 0};
  hipStream_t st = s->ctx->stream;
  HIP_TRY(hipMemcpyAsync(h, s->flags.p, sizeof h, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h[0] || h[1]) HIP_TRY(hipMemsetAsync(s->flags.p, 0, sizeof h, st));
  if (h[1]) s->has_clipped = 1;
  if (h[0]) return NVH_ERR_RUNTIME;  // inverse_dB_table / wMap index out of range in the reference
  return NVH_OK;
}

extern "C" int nvh_stream_synth(nvh_stream* s, float* pcm_host, float* d_pcm, int64_t capacity, int64_t* written) {
  if (!s || (pcm_host && d_pcm)) return NVH_ERR_ARGUMENT;
  if (written) *written = 0;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  HIP_TRY(hipSetDevice(s->ctx->device));
  const int ch = s->setup.channels;
  const int64_t need = s->pending.pcm_samples * ch;
  if (s->pending.frames.empty()) return NVH_OK;
  if (need > 0 && !pcm_host && !d_pcm) return NVH_ERR_ARGUMENT;
  if (capacity < need) return NVH_ERR_ARGUMENT;
  nvh_batch* b = &s->scratch;
  int rc = batch_upload(s, b);
  if (rc != NVH_OK) return rc;
  float* dst = d_pcm;
  if (!dst) {
    if ((rc = s->pcm.reserve((size_t)(need > 0 ? need : 1) * sizeof(float))) != NVH_OK) return rc;
    dst = (float*)s->pcm.p;
  }
  rc = batch_launch(b, (const float*)s->carry[s->carry_cur].p, (float*)s->carry[s->carry_cur ^ 1].p, dst, false, nullptr);
  if (rc != NVH_OK) return rc;
  hipStream_t st = s->ctx->stream;
  if (b->last_decoded >= 0) s->carry_cur ^= 1;  // the batch wrote its last block's tail into the other buffer
  // one read-back, one synchronisation: PCM and the two flag words land in a pinned bounce buffer
  size_t pcm_bytes = pcm_host ? (size_t)need * sizeof(float) : 0;
  // a destination in pinned host memory (nvh_pinned_alloc, hipHostMalloc, hipHostRegister) is written by the copy
  // engine directly; anything else goes through the bounce buffer and one memcpy on this thread
  bool direct = false;
  if (pcm_bytes) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, pcm_host) == hipSuccess) direct = attr.type == hipMemoryTypeHost;
    else (void)hipGetLastError();  // plain pageable memory: not an error
  }
  const size_t bounce = direct ? 0 : pcm_bytes;
  if ((rc = s->h_pcm.reserve(bounce + 2 * sizeof(int))) != NVH_OK) return rc;
  int* h_flags = (int*)((uint8_t*)s->h_pcm.p + bounce);
  if (pcm_bytes) HIP_TRY(hipMemcpyAsync(direct ? (void*)pcm_host : s->h_pcm.p, dst, pcm_bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(h_flags, s->flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (bounce) std::memcpy(pcm_host, s->h_pcm.p, bounce);
  if (h_flags[0] || h_flags[1]) HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), st));
  if (h_flags[1]) s->has_clipped = 1;
  if (h_flags[0]) return NVH_ERR_RUNTIME;  // inverse_dB_table / wMap index out of range in the reference
  if (written) *written = need;
  return NVH_OK;
}

// IMode.Decode on one packet (Mode.cs:153-170): floors, residue, coupling, floor apply, IMDCT and window of that packet
// alone -- the windowed block, before any overlap -- into d_block [channels][block1] (device memory).  Does not
// touch the stream's decode state; the stream must have nothing pending.  *decoded = 0 when the reference would have
// returned without decoding (short packet).
extern "C" int nvh_mode_decode(nvh_stream* s, const uint8_t* pkt, int len, float* d_block, int* decoded, int* block_size,
                               int* start, int* valid, int* total) {
  if (!s || (!pkt && len > 0) || len < 0 || !d_block) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (!s->pending.frames.empty()) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(s->ctx->device));
  if (decoded) *decoded = 0;
  static const uint8_t empty = 0;
  nvh::StreamParser one(&s->setup);  // a fresh parser: Mode.Decode does not depend on what came before
  nvh::FrameBatch fb;
  int rc = one.push_packet(pkt ? pkt : &empty, len, -1, 0, fb);
  if (rc != NVH_OK) return rc;
  if (fb.frames.empty() || fb.frames[0].n == 0) return NVH_OK;
  const NvhFrame f0 = fb.frames[0];
  nvh_batch b;
  b.blob.pool = b.work.pool = b.carry_in.pool = b.slabs.pool = &s->ctx->pool;
  b.h_blob.host = true;
  b.h_blob.pool = &s->ctx->hpool;
  b.block_only = true;
  const bool was_gpu = s->gpu_parse;
  s->gpu_parse = false;
  std::swap(s->pending, fb);
  rc = batch_upload(s, &b);
  std::swap(s->pending, fb);
  s->pending.clear();
  s->gpu_parse = was_gpu;
  if (rc != NVH_OK) return rc;
  rc = batch_launch(&b, (const float*)s->carry[s->carry_cur].p, nullptr, nullptr, false, nullptr);
  if (rc != NVH_OK) return rc;
  hipStream_t st = s->ctx->stream;
  const size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  HIP_TRY(hipMemcpyAsync(d_block, b.work.p, plane, hipMemcpyDeviceToDevice, st));
  rc = collect_flags(s);  // synchronises; a floor curve outside the dB table is NVH_ERR_RUNTIME here as well
  if (rc != NVH_OK) return rc;
  if (decoded) *decoded = 1;
  if (block_size) *block_size = f0.n;
  if (start) *start = f0.start;
  if (valid) *valid = f0.valid;
  if (total) *total = f0.total;
  return NVH_OK;
}

// IResidue.Decode(packet, doNotDecodeChannel, blockSize, buffer) on its own (see include/nvorbis_hip.h): the host reads
// the classifications and entries from the packet, k_residue adds the vectors into the caller's planes.
extern "C" int nvh_residue_decode(nvh_stream* s, int residue_index, const uint8_t* pkt, int len, int bit_offset,
                                  int any_channel_decodes, int block_size, float* d_buffer, int* bits_consumed) {
  if (!s || (!pkt && len > 0) || len < 0 || bit_offset < 0 || !d_buffer) return NVH_ERR_ARGUMENT;
  if (residue_index < 0 || residue_index >= (int)s->setup.residues.size()) return NVH_ERR_ARGUMENT;
  if (block_size != s->setup.block0 && block_size != s->setup.block1) return NVH_ERR_ARGUMENT;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  if (!s->pending.frames.empty()) return NVH_ERR_ARGUMENT;
  if (bits_consumed) *bits_consumed = 0;
  if (!any_channel_decodes) return NVH_OK;  // Array.IndexOf(doNotDecodeChannel, false) == -1 (Residue0.cs:125): nothing is read
  HIP_TRY(hipSetDevice(s->ctx->device));
  static const uint8_t empty = 0;
  nvh::StreamParser one(&s->setup);
  nvh::FrameBatch fb;
  int rc = one.parse_residue(residue_index, pkt ? pkt : &empty, len, bit_offset, block_size, fb, bits_consumed);
  if (rc != NVH_OK) return rc;
  nvh_batch b;
  b.blob.pool = b.work.pool = b.carry_in.pool = b.slabs.pool = &s->ctx->pool;
  b.h_blob.host = true;
  b.h_blob.pool = &s->ctx->hpool;
  const bool was_gpu = s->gpu_parse;
  s->gpu_parse = false;
  std::swap(s->pending, fb);
  rc = batch_upload(s, &b);
  std::swap(s->pending, fb);
  s->pending.clear();
  s->gpu_parse = was_gpu;
  if (rc != NVH_OK) return rc;
  hipStream_t st = s->ctx->stream;
  const size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  HIP_TRY(hipMemcpyAsync(b.work.p, d_buffer, plane, hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(k_residue, dim3(1), dim3(256), 0, st, s->dev, b.dev, (float*)b.work.p, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(d_buffer, b.work.p, plane, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  return NVH_OK;
}

extern "C" int nvh_batch_upload(nvh_stream* s, nvh_batch** out) {
  if (!s || !out) return NVH_ERR_ARGUMENT;
  *out = nullptr;
  if (!s->ctx) return NVH_ERR_NO_GPU;
  HIP_TRY(hipSetDevice(s->ctx->device));
  std::unique_ptr<nvh_batch> b(new (std::nothrow) nvh_batch());
  if (b && s && s->ctx) {
    b->blob.pool = b->work.pool = b->carry_in.pool = b->slabs.pool = &s->ctx->pool;
    b->h_blob.host = true;
    b->h_blob.pool = &s->ctx->hpool;
  }
  if (!b) return NVH_ERR_NOMEM;
  // snapshot the tail this batch starts from so that repeated synthesis is idempotent
  size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  int rc = b->carry_in.reserve(plane);
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpyAsync(b->carry_in.p, s->carry[s->carry_cur].p, plane, hipMemcpyDeviceToDevice, s->ctx->stream));
  b->has_carry_in = true;
  rc = batch_upload(s, b.get());
  if (rc != NVH_OK) return rc;
  *out = b.release();
  return NVH_OK;
}

extern "C" int nvh_batch_info(const nvh_batch* b, int* frames, int* chan_frames, int64_t* pcm_samples,
                              int64_t* descriptor_bytes) {
  if (!b) return NVH_ERR_ARGUMENT;
  if (frames) *frames = b->nframes;
  if (chan_frames) *chan_frames = b->chan_frames;
  if (pcm_samples) *pcm_samples = b->pcm_samples;
  if (descriptor_bytes) *descriptor_bytes = b->descriptor_bytes;
  return NVH_OK;
}

extern "C" int nvh_batch_stats(const nvh_batch* b, int64_t* out8) {
  if (!b || !out8) return NVH_ERR_ARGUMENT;
  for (int i = 0; i < 8; i++) out8[i] = b->stats[i];
  return NVH_OK;
}

extern "C" int nvh_batch_kernels(const nvh_batch* b, char* buf, int cap) {
  if (!b || !buf || cap <= 0) return NVH_ERR_ARGUMENT;
  std::string t;
  for (int k = 0; k < 4; k++) {
    if (k) t += ",";
    t += b->slot_name[k];
  }
  if ((int)t.size() + 1 > cap) return NVH_ERR_ARGUMENT;
  std::memcpy(buf, t.c_str(), t.size() + 1);
  return NVH_OK;
}

extern "C" int nvh_batch_synth(nvh_batch* b, float* d_pcm, int64_t capacity) {
  if (!b || !b->s) return NVH_ERR_ARGUMENT;
  nvh_stream* s = b->s;
  if (capacity < b->pcm_samples * s->setup.channels) return NVH_ERR_ARGUMENT;
  if (b->pcm_samples > 0 && !d_pcm) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(s->ctx->device));
  // the stream keeps the tail of the newest batch (written to its current carry buffer; the batch reads its own snapshot)
  return batch_launch(b, (const float*)b->carry_in.p, (float*)s->carry[s->carry_cur].p, d_pcm, false, nullptr);
}

extern "C" int nvh_batch_time(nvh_batch* b, float* d_pcm, int64_t capacity, int iters, float* total_ms,
                              float* kernel_ms) {
  if (!b || !b->s || iters <= 0) return NVH_ERR_ARGUMENT;
  nvh_stream* s = b->s;
  if (capacity < b->pcm_samples * s->setup.channels) return NVH_ERR_ARGUMENT;
  HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = s->ctx->stream;
  float km[4] = {0, 0, 0, 0};
  if (kernel_ms) {
    for (int i = 0; i < iters; i++) {
      int rc = batch_launch(b, (const float*)b->carry_in.p, (float*)s->carry[s->carry_cur].p, d_pcm, true, km);
      if (rc != NVH_OK) return rc;
    }
    for (int k = 0; k < 4; k++) kernel_ms[k] = km[k] / (float)iters;
  }
  if (total_ms) {
    ScopedEvent e0, e1;
    int rc = e0.create();
    if (rc == NVH_OK) rc = e1.create();
    if (rc != NVH_OK) return rc;
    HIP_TRY(hipEventRecord(e0.e, st));
    for (int i = 0; i < iters; i++) {
      rc = batch_launch(b, (const float*)b->carry_in.p, (float*)s->carry[s->carry_cur].p, d_pcm, false, nullptr);
      if (rc != NVH_OK) return rc;
    }
    HIP_TRY(hipEventRecord(e1.e, st));
    HIP_TRY(hipEventSynchronize(e1.e));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0.e, e1.e));
    *total_ms = ms;
  }
  return collect_flags(s);
}

extern "C" void nvh_batch_free(nvh_batch* b) {
  if (!b) return;
  if (b->s) {
    (void)hipSetDevice(b->s->ctx->device);
    (void)hipStreamSynchronize(b->s->ctx->stream);
  }
  delete b;
}

// ------------------------------------------------------------------------------------------------
// container helper
// ------------------------------------------------------------------------------------------------

extern "C" int nvh_ogg_demux_stream(const uint8_t* bytes, size_t len, int stream_index, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                                    int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                                    int64_t* total_bytes, int* nstreams) {
  if (!bytes || !npackets || !total_bytes || stream_index < 0) return NVH_ERR_ARGUMENT;
  nvh::OggPackets pk;
  int rc = nvh::ogg_demux(bytes, len, pk, stream_index, nstreams);
  if (rc != NVH_OK) return rc;
  int n = (int)pk.granule.size();
  *npackets = n;
  *total_bytes = (int64_t)pk.bytes.size();
  if (!pkt_bytes && !offsets && !granules && !flags) return NVH_OK;  // sizing call
  if (pkt_cap < n || pkt_bytes_cap < (int64_t)pk.bytes.size() || !pkt_bytes || !offsets || !granules || !flags)
    return NVH_ERR_ARGUMENT;
  if (!pk.bytes.empty()) std::memcpy(pkt_bytes, pk.bytes.data(), pk.bytes.size());
  std::memcpy(offsets, pk.offs.data(), sizeof(int64_t) * (size_t)(n + 1));
  if (n) {
    std::memcpy(granules, pk.granule.data(), sizeof(int64_t) * (size_t)n);
    std::memcpy(flags, pk.flags.data(), (size_t)n);
  }
  return NVH_OK;
}

extern "C" int nvh_ogg_demux(const uint8_t* bytes, size_t len, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                             int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                             int64_t* total_bytes) {
  return nvh_ogg_demux_stream(bytes, len, 0, pkt_bytes, pkt_bytes_cap, offsets, granules, flags, pkt_cap, npackets, total_bytes, nullptr);
}
