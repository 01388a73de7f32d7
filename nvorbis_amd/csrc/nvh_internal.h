// nvh_internal.h -- shared by the translation units behind the C ABI of libnvorbis_hip.so:
//   nvh_api.hip     contexts, streams (packets in, PCM out), resident batches, the Ogg helper
//   nvh_setup.hip   device images of a stream's setup (synthesis tables, GPU-parser tables)
//   nvh_launch.hip  batch upload and the launch policy (which kernel variants a batch runs through)
//   nvh_ops.hip     level-1 operators: device-pointer mirrors of the reference's interface methods
// Nothing here is part of the ABI (include/nvorbis_hip.h is).
#pragma once
#include <time.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/nvorbis_hip.h"
#include "host_ogg.h"
#include "host_parse.h"
#include "host_setup.h"
#include "host_slab.h"
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
                              int* clipped_flag, float* carry_out, int last_decoded, int nosym, const int* list, int emitted);
__global__ void k_spectrum(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_spectrum_f0(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent);
__global__ void k_spectrum_imdct(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_spectrum_gen(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_spectrum_gen8(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_spectrum_gen_imdct(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_spectrum_gen8_imdct(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err, int cap_pass, int cap_ops, int cap_ent NVH_DBG_PARAMS);
__global__ void k_mdct_reverse_wave(float* buf, int n, long long stride, const float* A, const float* B, const float* C,
                                    const float* TW);
#define NVH_PARSE_DECL(NAME)                                                                                                         \
  __global__ void NAME(NvhDevParse T, const uint8_t* pkt_pool, const NvhPacketRef* refs, int nframes, NvhFrame* frames, NvhChan* chans,   \
                       NvhResPass* passes, NvhResOp* ops, uint16_t* op_link, uint16_t* entries, uint16_t* posts, int* scratch,            \
                       NvhParseResult* result, int lanes, int scratch_words, int pkt_words, uint4* slabs, const int* order,         \
                       uint32_t* handover NVH_DBG_PARAMS)
NVH_PARSE_DECL(k_parse);         // descriptors out; packets and scratch rows in LDS
NVH_PARSE_DECL(k_parse_g);       // ... in global memory
NVH_PARSE_DECL(k_parse_slab);    // slabs out (kernels_parse.hip: parse_body<.., SLAB>)
NVH_PARSE_DECL(k_parse_slab_g);
NVH_PARSE_DECL(k_parse_slab_u);  // k_parse_slab for one packet per wavefront, wave-uniform control flow (parse_body<.., UNI>)
NVH_PARSE_DECL(k_parse_slab_c);  // several packets per wavefront, one cursor per lane through the residue walk: the parse alone ...
NVH_PARSE_DECL(k_parse_slab_f);  // the lean form of k_parse_slab_c for ordinary packets; the others it leaves to k_parse_slab_c
NVH_PARSE_DECL(k_parse_slab_t);  // ... and the rest of the slab, one packet per wavefront (parse_body<.., CUR, PHASE>)
__global__ void k_parse_result_out(const NvhParseResult* dev, NvhParseResult* host);
__global__ void k_parse_fetch(const uint4* stage_h, uint4* stage_d, long long stage_n16, const uint8_t* pool_h, uint8_t* pool_d,
                              long long pool_bytes, NvhParseResult* result);
__global__ void k_parse_links(int nframes, int channels, NvhFrame* frames, NvhChan* chans, const uint32_t* carry_exec_in,
                              uint32_t* carry_exec_out, int last_decoded, NvhParseResult* result, uint4* slabs, int stride_vecs);
__global__ void k_inverse_couple(float* magnitude, float* angle, int cnt);
__global__ void k_copy_f4(const float4* src, float4* dst, long long n4);
__global__ void k_synth(NvhSynthArgs A NVH_DBG_PARAMS);
__global__ void k_synth_g(NvhSynthArgs A NVH_DBG_PARAMS);     // + the general bin walk (Residue0, odd dimensions, several passes)
__global__ void k_synth_tail(NvhSynthArgs A NVH_DBG_PARAMS);  // + the carried tail written in place (kernels_synth.hip: MODE 1)
__global__ void k_synth_emit(NvhSynthArgs A NVH_DBG_PARAMS);  // + paired emission (MODE 2)
__global__ void k_synth_group2(NvhSynthArgs A NVH_DBG_PARAMS);  // frame groups: two / four frames per workgroup, the overlaps between them on chip
__global__ void k_synth_group4(NvhSynthArgs A NVH_DBG_PARAMS);
__global__ void k_synth8(NvhSynthArgs A NVH_DBG_PARAMS);
__global__ void k_synth8_g(NvhSynthArgs A NVH_DBG_PARAMS);    // + the general bin walk
__global__ void k_synth8_emit(NvhSynthArgs A NVH_DBG_PARAMS);  // wide frames + paired emission through LDS (synth_emit8)
__global__ void k_window_apply(float* buf, const float* window, int n, long long stride, int batch);
__global__ void k_overlap_buffers(const float* previous, float* next, int prev_start, int len, int next_start, int channels,
                                  long long plane_stride);
__global__ void k_copy_buffer(const float* planes, int start, int count, int channels, long long plane_stride, float* target,
                              int clip, int* clipped_flag);
__global__ void k_floor0_apply(const int32_t* bark, const float* qk, int K, const float* amps, const int32_t* skip, int n, float* data,
                               long long stride);
__global__ void k_floor1_apply(NvhDevSetup S, int floor_idx, const uint16_t* posts, const int32_t* counts, int n, float* data,
                               long long stride, int* status);
__global__ void k_residue(NvhDevSetup S, NvhDevBatch Bt, float* work, int clear);
__global__ void k_couple_floor(NvhDevSetup S, NvhDevBatch Bt, float* work, int* err);
__global__ void k_ola_emit(NvhDevSetup S, NvhDevBatch Bt, const float* work, const float* carry, float* pcm, int clip,
                           int* clipped_flag);
__global__ void k_ola_emit_seq(NvhDevSetup S, NvhDevBatch Bt, float* work, const float* carry, float* pcm, int clip,
                               int* clipped_flag);
}

extern thread_local int g_last_hip_error;

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t e_ = (expr);                  \
    if (e_ != hipSuccess) {                  \
      g_last_hip_error = (int)e_;            \
      return NVH_ERR_DEVICE;                 \
    }                                        \
  } while (0)

// No C++ exception may unwind through the C ABI (a P/Invoke or ctypes caller would be torn down with it): every
// extern "C" entry point runs its body inside this barrier.  Header fields are untrusted and size std::vectors.
template <class F>
static inline int nvh_guard(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return NVH_ERR_NOMEM;
  } catch (...) {
    return NVH_ERR_RUNTIME;
  }
}
template <class F>
static inline void nvh_guard_void(F&& body) noexcept {
  try {
    body();
  } catch (...) {
  }
}

// Test / experiment switches from the environment, read once per process (before the first context exists).
struct NvhToggles {
  bool no_compact, no_fused_imdct, no_gen8, unfused, no_pair, debug_occ, gpu_parse_default;
  bool no_ola_sym;  // NVH_NO_OLA_SYM: k_ola_compact without its read-once steady-state path (test / A-B aid)
  bool emit8;       // NVH_EMIT8: accepted, no effect any more (paired emission for wide frames is the default since round 5)
  bool no_emit8;    // NVH_NO_EMIT8: no paired emission for more than two channels / blocks beyond 2048 (k_synth8 + k_ola_compact; A/B aid)
  bool no_emit;     // NVH_NO_EMIT: no paired emission -- every frame's PCM through k_ola_compact (test / A-B aid)
  bool emit_always; // NVH_EMIT_ALWAYS: paired emission for every batch that has a steady-state frame (default: batches that are
                    // at least 7/8 steady state; the parity suite replays itself with this switch to cover the mixed cases)
  bool xcd_map;      // NVH_XCD_MAP: paired-emission launches take their frames in eight per-XCD runs instead of workgroup order (A/B aid)
  bool copy_upload;  // NVH_COPY_UPLOAD: a GPU-parse batch's input goes up by copy commands instead of k_parse_fetch (A/B aid)
  int fpw;           // NVH_FPW: frames per workgroup of the mono / stereo synthesis with paired emission (kernels_synth.hip: frame groups):
                     // 1 = k_synth + k_synth_emit (the round-3..5 form), 2 (default) = k_synth_group2, 4 = k_synth_group4
  bool no_walk_two;  // NVH_NO_WALK_TWO: a frame group's two residue walks one after the other (A/B aid)
  bool no_prefetch;  // NVH_NO_PREFETCH: the odd launch of a paired-emission pass does not touch the even launch's slabs (A/B aid)
  bool uncached_planes;  // NVH_UNCACHED_PLANES: a batch's work planes in hipDeviceMallocUncached memory (the round-4 experiment whose
                         // wrong PCM with GPU-parsed batches was never explained: tools/repro_uncached.py)
  bool poison_planes;    // NVH_POISON_PLANES: work planes filled with NaN patterns at upload (finds reads of regions a batch never wrote)
  bool no_slab;   // NVH_NO_SLAB: the descriptor kernels (k_spectrum_imdct & co.) instead of the slab kernels (test / A-B aid)
  int lds_pad, ola_threads, parse_lanes, parse_waves;
  bool no_sleep_wait;  // NVH_NO_SLEEP_WAIT: worker-pool contexts wait with hipStreamSynchronize like every other (A/B aid)
  bool no_parse_sort;  // NVH_NO_PARSE_SORT: the GPU parser takes a batch's frames in stream order instead of longest packet first (A/B aid)
  int parse_cur;      // NVH_PARSE_CUR: several packets per wavefront in slab mode: 0 = the lockstep nest (k_parse_slab / _g), 1 = the cursor
                      // walk + tail kernel (k_parse_slab_c / _t), 2 = the lean walk in front of it (k_parse_slab_f; the default where
                      // the setup allows it); -1 = not set
  bool no_parse_sub;  // NVH_NO_PARSE_SUB: k_parse_slab_f finds long codes by scanning their groups instead of through the second-level tables (A/B aid)
  bool no_parse_uni;  // NVH_NO_PARSE_UNI: one-packet-per-wavefront batches through k_parse_slab instead of k_parse_slab_u (A/B aid)
  int ola_segs;   // NVH_OLA_SEGS: workgroups per frame in k_ola_compact (default: by frame size)
  int phase_mask;  // debug build only (NVH_DEBUG_SPECTRUM_MASK)
};
const NvhToggles& nvh_toggles();
#ifdef NVH_DEBUG
extern void* g_dbg_buf;  // per-workgroup phase timestamps of the spectrum kernels (nvh_debug_set_buffer)
#define NVH_DBG_LAUNCH , (long long*)g_dbg_buf, nvh_toggles().phase_mask
#else
#define NVH_DBG_LAUNCH
#endif

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
  bool uncached = false;  // experiment (NVH_UNCACHED_PLANES): device memory that bypasses the L2s (hipDeviceMallocUncached), never pooled
  ~DevBuf() { release(); }
  void release() {
    if (!p) return;
    if (pool && !uncached) pool->give(p, cap);
    else (void)(host ? hipHostFree(p) : hipFree(p));
    p = nullptr;
    cap = 0;
  }
  int reserve(size_t bytes) {
    if (bytes <= cap) return NVH_OK;
    release();
    const size_t want = BufPool::size_class(bytes + 256);
    if (pool && !uncached) p = pool->take(want);
    if (!p) {
      if (host) HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
      else if (uncached) HIP_TRY(hipExtMallocWithFlags(&p, want, hipDeviceMallocUncached));
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

// Everything derived from a stream's headers: parsed tables on the host, their device image, kernel-selection
// flags.  Immutable once built, so streams with byte-identical identification + setup packets (the normal case
// inside one corpus: same encoder, same settings) share one entry per context.
struct SharedSetup {
  nvh::Setup setup;
  DevBuf arena;  // setup tables
  NvhDevSetup dev{};
  DevBuf dev_copy;  // the NvhDevSetup block itself in device memory (the run kernel reads it from there, see kernels_run.hip)
  bool fast_spectrum = false;  // every residue takes the pair path and the fused tail applies: k_spectrum proper
  bool has_floor0 = false;
  const uint4* synth_consts = nullptr;  // inverse_dB_table + lattice pool in 16-byte units (kernels_synth.hip), inside `arena`
  int synth_const_vecs = 0;
  int max_posts = 0;            // largest Floor1 post count of the setup (bounds a slab's segment lists)
  nvh::SlabSetup slab;          // host copy of the codebook directory: what host_slab.cpp needs to write pair records
  bool slab_general = false;    // ... and some of its frames need the general bin walk (k_synth_g / k_synth8, no paired emission)
  bool slab_setup_ok = false;   // the setup is inside the slab synthesis kernels' contract (nvh_launch.hip: slab_path)
  bool has_sequential = false;  // some residue replays the reference's partition order (quirk B-1 / vector overrun)
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
  bool big_lds_attr_set = false;    // the general spectrum kernels' 152 KB dynamic-LDS opt-in was made on this context's device
  bool group_lds_attr_set = false;  // ... k_synth_group2 / 4's
  bool synth_lds_attr_set = false;  // k_synth8's 160 KB dynamic-LDS opt-in was made on this context's device
  bool parse_lds_attr_set = false;  // k_parse's 80 KB dynamic-LDS opt-in was made on this context's device
  int parse_lanes = 0;              // nvh_ctx_set_parse_lanes: packets per wavefront of the GPU parser, 0 = automatic
};

// Wait for the context's stream.  A context of a worker pool (nvh_ctx_set_parse_lanes > 0: one of many host threads, each waiting
// milliseconds for its own parse) polls and sleeps instead of calling hipStreamSynchronize, which spins: a pool may then have
// more threads than the host has cores for it (a container's CPU quota), and more parses in flight than cores.
inline hipError_t nvh_wait_stream(const nvh_ctx* c, hipStream_t st) {
  if (!c || c->parse_lanes <= 0 || nvh_toggles().no_sleep_wait) return hipStreamSynchronize(st);
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    struct timespec ts = {0, 50 * 1000};
    nanosleep(&ts, nullptr);
  }
}

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
  int max_vecs = 0;     // GPU-parsed batch in slab mode: its largest slab, as k_parse reported it
  bool block_only = false;       // nvh_mode_decode: stop after the windowed IMDCT (full blocks in the work planes, no overlap-add)
  bool descriptors_only = false; // nvh_residue_decode: the caller launches a descriptor kernel itself (no slabs)
  bool has_carry_in = false;
  DevBuf dev_copy;  // the NvhDevBatch block in device memory
  bool dev_copy_valid = false;
  // slab synthesis kernel (kernels_synth.hip): per-frame slabs, written by the packet parser (host_slab.cpp into `blob`, k_parse into `slab3`)
  DevBuf slab3;
  int slab_stride_vecs = 0;   // 16-byte units between slabs (host-written: the batch's largest slab; GPU-written: the setup's worst case)
  int slab_cap_vecs = 0;      // the batch's largest slab: what the synthesis kernels' LDS slab area holds
  bool slabs_ready = false;
  bool slab_host = false;          // the slabs were written by the host parser's thread (host_slab.cpp) and lie inside `blob`
  const uint4* d_slabs = nullptr;  // ... here
  // paired emission (nvh_format.h: NVH_EMIT_*): frames whose PCM k_synth writes itself, and the frames left to k_ola_compact
  int emit_frames = 0;           // frames with NVH_EMIT_DONE
  int fpw = 1;                   // frames per workgroup the emission flags were laid out for (1: odd / even frames; 2, 4: frame groups)
  bool ola_all = false;          // GPU-parsed batch in which k_parse_links withdrew an emission candidate: k_ola_compact over every frame
  int ola_count = 0;             // entries of d_ola_list
  const int* d_ola_list = nullptr;  // inside the descriptor blob
};

// GPU-parse mode: everything pushed since the last batch boundary, so that a batch in which k_parse found a packet the
// reference would throw on can be replayed through the host parser (same frames kept, same state, same error code as in
// host-parse mode) instead of being dropped.
struct ReplayLog {
  enum { kPacket = 0, kEnd = 1, kPosition = 2 };
  struct Event { int kind; int64_t off, len, granule; int flags; };
  std::vector<uint8_t> bytes;
  std::vector<Event> events;
  void clear() { bytes.clear(); events.clear(); }
};

#define NVH_INTERNAL_REPLAY 1000  // batch_upload_gpu -> batch_upload: parse the logged packets on the host instead

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
  nvh::SlabBatch slab_build;  // scratch of host_slab.cpp, reused from batch to batch
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
  ReplayLog replay;
  std::unique_ptr<nvh::StreamParser> replay_start;  // parser state at the first logged event
  int replay_error = NVH_OK;                         // first error of the last replay (reported by the synthesis call)
  std::vector<std::pair<int, int64_t>> replay_errors;  // every error of it: (code, samples per channel the batch emits before that packet)
  // Pipelined read-back (nvh_stream_synth_begin / _end): the PCM of batch i travels to the host on a stream of its own while
  // batch i+1 is uploaded, parsed and synthesised.  Two batches may be outstanding, ended in the order they were begun.
  struct Flight {
    bool on = false;
    hipEvent_t kernels = nullptr, done = nullptr;
    int64_t need = 0;
    int replay_error = NVH_OK;
    std::vector<std::pair<int, int64_t>> replay_errors;
  };
  hipStream_t copy_stream = nullptr;
  DevBuf pcm2[2];
  DevBuf h_flags2;  // pinned int[4]: the two flag words of each outstanding batch
  Flight flight[2];
  int flight_next = 0, flight_first = 0;
  ~nvh_stream() {
    for (Flight& f : flight) {
      if (f.kernels) (void)hipEventDestroy(f.kernels);
      if (f.done) (void)hipEventDestroy(f.done);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  }

  nvh_stream(nvh_ctx* c, std::shared_ptr<SharedSetup> sh)
      : ctx(c), shared(std::move(sh)), setup(shared->setup), arena(shared->arena), dev(shared->dev),
        fast_spectrum(shared->fast_spectrum), has_floor0(shared->has_floor0) {
    BufPool* pool = c ? &c->pool : nullptr;
    carry[0].pool = carry[1].pool = flags.pool = pcm.pool = carry_exec.pool = pcm2[0].pool = pcm2[1].pool = pool;
    h_flags2.host = true;
    h_flags2.pool = c ? &c->hpool : nullptr;
    scratch.blob.pool = scratch.work.pool = scratch.carry_in.pool = scratch.slabs.pool = scratch.dev_copy.pool = scratch.slab3.pool = pool;
    scratch.work.uncached = nvh_toggles().uncached_planes;
    h_pcm.host = scratch.h_blob.host = true;
    h_pcm.pool = scratch.h_blob.pool = c ? &c->hpool : nullptr;
    scratch.s = this;
    if (c) {  // GPU-parse batches: the packet bytes straight into page-locked memory (host_parse.h: PacketPool)
      pending.pkt_pool.owner = c;
      pending.pkt_pool.grow = &pinned_pool_grow;
    }
  }
  // PacketPool storage from the context's pool of page-locked blocks; new_cap == 0: give `old` (of old_cap bytes) back
  static uint8_t* pinned_pool_grow(void* owner, uint8_t* old, size_t old_cap, size_t new_cap) {
    nvh_ctx* c = static_cast<nvh_ctx*>(owner);
    if (new_cap == 0) {
      if (old) c->hpool.give(old, old_cap);
      return nullptr;
    }
    if (BufPool::size_class(new_cap) != new_cap) return nullptr;  // (the pool asks for powers of two >= 64 KB: size classes of their own)
    void* p = c->hpool.take(new_cap);
    if (!p && hipHostMalloc(&p, new_cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    return static_cast<uint8_t*>(p);
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

static inline bool valid_block(int n) { return n >= 64 && n <= 8192 && (n & (n - 1)) == 0; }
// LDS bytes of the wavefront IMDCT: n/4 complex points + 1/8 padding (kernels_imdct.hip Geo<LD>::LDS_FLOATS)
static inline size_t wave_lds_bytes(int n) { return (size_t)2 * ((size_t)(n / 4) + (size_t)(n / 32)) * sizeof(float); }

int get_mdct(nvh_ctx* c, int n, MdctDev** out);                 // nvh_ops.hip
int upload_setup(nvh_stream* s);                                 // nvh_setup.hip
int upload_parse_tables(nvh_stream* s);                          // nvh_setup.hip
int batch_upload(nvh_stream* s, nvh_batch* b);                   // nvh_launch.hip
int batch_launch(nvh_batch* b, const float* carry, float* carry_out, float* d_pcm, bool timing, float* kernel_ms,
                 hipEvent_t* ext_ev = nullptr);  // nvh_launch.hip
int collect_flags(nvh_stream* s);                                // nvh_launch.hip
void replay_note(nvh_stream* s, int kind, const uint8_t* data, int len, int64_t granule, int flags);  // nvh_launch.hip
