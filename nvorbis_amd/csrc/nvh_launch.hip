// nvh_launch.hip -- moving a parsed batch into HBM and the launch policy: which kernel variants a batch runs through.
#include <chrono>

#include "nvh_internal.h"

static int collect_parse_result(nvh_stream* s, nvh_batch* b, const NvhParseResult* d_res);
static bool slab_shape_ok(const nvh_batch* b);
static bool slab_size_ok(const nvh_batch* b);
static void assign_emission(nvh_stream* s, nvh_batch* b, nvh::FrameBatch& P, int fpw, std::vector<int>& ola_list);

void replay_note(nvh_stream* s, int kind, const uint8_t* data, int len, int64_t granule, int flags) {
  if (!s->gpu_parse) return;
  if (s->replay.events.empty()) s->replay_start.reset(new nvh::StreamParser(*s->parser));
  ReplayLog::Event e{kind, (int64_t)s->replay.bytes.size(), (int64_t)len, granule, flags};
  if (kind == ReplayLog::kPacket && len > 0) s->replay.bytes.insert(s->replay.bytes.end(), data, data + len);
  s->replay.events.push_back(e);
}

// k_parse found a packet the managed decoder would throw on.  The host parser in light mode has already walked past the
// whole look-ahead batch, so: back to the state at the batch boundary, and the logged packets once more through the full
// host parser.  The throwing packet(s) are consumed and leave no frame -- exactly what a caller of the host-parse mode
// gets who catches the exception and keeps reading, as a caller of the reference may (StreamDecoder.cs:465-530: the
// packet is `Done()` in the finally block, the previous-block state is untouched).  The first error code and the number
// of samples emitted before it are kept for the synthesis call to report.
static int replay_on_host(nvh_stream* s) {
  if (!s->replay_start) return NVH_ERR_RUNTIME;
  *s->parser = *s->replay_start;
  s->parser->set_light(false);
  s->pending.clear();
  static const uint8_t empty = 0;
  int first = NVH_OK;
  s->replay_errors.clear();
  for (const ReplayLog::Event& e : s->replay.events) {
    int rc = NVH_OK;
    if (e.kind == ReplayLog::kEnd) rc = s->parser->push_end(s->pending);
    else if (e.kind == ReplayLog::kPosition) s->parser->set_position_state(e.flags != 0, e.granule);
    else rc = s->parser->push_packet(e.len ? s->replay.bytes.data() + e.off : &empty, (int)e.len, e.granule, e.flags, s->pending);
    if (rc != NVH_OK) {
      if (first == NVH_OK) first = rc;
      s->replay_errors.emplace_back(rc, s->pending.pcm_samples);
    }
  }
  s->parser->set_light(true);
  s->replay_error = first;
  return NVH_OK;
}

// NVH_POISON_PLANES (test aid): a batch's work planes start out as NaN bit patterns, so that a kernel which reads a plane region
// nothing wrote in this batch -- and gets by on whatever an earlier decode left in the recycled block -- shows up as NaN PCM.
static int poison_planes(nvh_stream* s, nvh_batch* b, size_t bytes) {
  if (!nvh_toggles().poison_planes || !b->work.p || bytes == 0) return NVH_OK;
  HIP_TRY(hipMemsetAsync(b->work.p, 0xFF, bytes, s->ctx->stream));
  return NVH_OK;
}

// GPU-parse mode: upload frame geometry + packets, let k_parse produce the descriptors into per-frame slabs.
static int batch_upload_gpu(nvh_stream* s, nvh_batch* b, const std::vector<int>& ola_list) {
  static const bool tprint = std::getenv("NVH_TIME_UPLOAD") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_a = tnow();
  nvh::FrameBatch& P = s->pending;
  const NvhDevParse& T = s->shared->parse;
  const int ch = s->setup.channels;
  const size_t nf = P.frames.size();
  P.pkt_refs.resize(nf);  // trailing pseudo-frames
  if (P.pkt_pool.empty()) {
    uint8_t* z = P.pkt_pool.append(8);
    if (!z) return NVH_ERR_NOMEM;
    std::memset(z, 0, 8);
  }
  // the packets lie in page-locked memory when the stream placed the pool there (nvh_api.hip): they go up from where the
  // parser wrote them; otherwise through the staging block like the records
  const bool pool_pinned = P.pkt_pool.grow != nullptr;
  auto al = [](size_t v) { return (v + 255) / 256 * 256; };
  // host-written prefix of the blob ...
  const size_t o_fr = 0;
  const size_t o_ch = al(o_fr + std::max<size_t>(nf, 1) * sizeof(NvhFrame));
  const size_t o_rf = al(o_ch + std::max<size_t>(nf * ch, 1) * sizeof(NvhChan));
  const size_t o_ol = al(o_rf + std::max<size_t>(nf, 1) * sizeof(NvhPacketRef));
  const size_t o_od = al(o_ol + std::max<size_t>(ola_list.size(), 1) * sizeof(int));  // parse order: frames, longest packet first
  const size_t o_pk = al(o_od + std::max<size_t>(nf, 1) * sizeof(int));
  const size_t pool_bytes = P.pkt_pool.size;
  const size_t host_bytes = pool_pinned ? al(o_pk) : al(o_pk + pool_bytes + 8);  // what the staging block holds
  // Slab mode: k_parse writes the synthesis kernels' slabs itself (kernels_parse.hip: parse_body<.., SLAB>), at the stride of the
  // setup's worst case; the LDS the synthesis kernels need is sized from the batch's largest slab, reported with the result.
  // (a setup whose worst-case slabs would not fit a sane allocation for this batch takes the descriptor form instead of failing)
  b->slab_stride_vecs = T.slab_stride_vecs;
  b->slab_cap_vecs = T.slab_stride_vecs;  // for the pre-launch size check: the bound
  const bool slab_fits = (size_t)std::max<size_t>(nf, 1) * (size_t)T.slab_stride_vecs * 16 <= ((size_t)4 << 30);
  const bool slab_mode = T.slab_stride_vecs > 0 && slab_fits && slab_shape_ok(b) && slab_size_ok(b);
  // ... and the device-only slabs behind it (slab mode writes no op / op-link descriptors: every vector write goes straight to its
  // record in the slab, so those two regions shrink to nothing)
  const size_t ops_per_frame = slab_mode ? 0 : (size_t)T.cap_ops;
  const size_t o_ps = al(o_pk + pool_bytes + 8);
  const size_t o_op = al(o_ps + nf * (size_t)T.cap_pass * sizeof(NvhResPass));
  const size_t o_lk = al(o_op + nf * ops_per_frame * sizeof(NvhResOp));
  const size_t o_en = al(o_lk + nf * ops_per_frame * sizeof(uint16_t));
  const size_t o_po = al(o_en + nf * (size_t)T.cap_ent * sizeof(uint16_t) + 64);
  const size_t o_sc = al(o_po + nf * (size_t)ch * NVH_MAX_POSTS * sizeof(uint16_t));
  const size_t row_words = (size_t)T.row_words;  // the residue walk's rows
  const size_t o_rs = al(o_sc + nf * row_words * sizeof(int));
  const size_t o_ho = al(o_rs + sizeof(NvhParseResult));  // hand-over of the split parser (kernels_parse.hip: NVH_PHO_WORDS words per frame)
  const size_t total = al(o_ho + (slab_mode ? std::max<size_t>(nf, 1) * 16 * sizeof(uint32_t) : 0));
  if (slab_mode) {
    int rcs = b->slab3.reserve(((size_t)std::max<size_t>(nf, 1) * (size_t)T.slab_stride_vecs * 16 + 4096 + 255) & ~(size_t)255);
    if (rcs != NVH_OK) return rcs;
  }
  int rc = b->blob.reserve(total);
  if (rc != NVH_OK) return rc;
  if ((rc = b->h_blob.reserve(host_bytes)) != NVH_OK) return rc;
  uint8_t* h = (uint8_t*)b->h_blob.p;
  if (nf) std::memcpy(h + o_fr, P.frames.data(), nf * sizeof(NvhFrame));
  if (nf) std::memcpy(h + o_ch, P.chans.data(), std::min(P.chans.size(), nf * (size_t)ch) * sizeof(NvhChan));
  if (nf) std::memcpy(h + o_rf, P.pkt_refs.data(), nf * sizeof(NvhPacketRef));
  if (!ola_list.empty()) std::memcpy(h + o_ol, ola_list.data(), ola_list.size() * sizeof(int));
  // (where it pays: several packets per wavefront AND packets of both block sizes -- a short block has an eighth of a long one's
  // symbols; the C5 writer's packets at eight per wavefront 3.69 -> 3.39 ms per 3000, 32 768: 3.82 -> 3.64 ms.  Batches of one
  // block size lose 2-4 % to the scattered slab writes, and one packet per wavefront has no lanes to keep together.)
  bool sorted_parse = false;
  if (nf > 1 && !nvh_toggles().no_parse_sort && (nf > 4096 || s->ctx->parse_lanes > 1 || nvh_toggles().parse_lanes > 1)) {
    int n0 = 0;
    for (size_t i = 0; i < nf && !sorted_parse; i++) {
      const int n = P.frames[i].n;
      if (n != 0 && n0 == 0) n0 = n;
      else if (n != 0 && n != n0) sorted_parse = true;
    }
  }
  if (sorted_parse) {
    // counting sort by packet length in 32-byte steps (descending, stable): ~10 us per 4096 frames
    int* od = (int*)(h + o_od);
    constexpr uint32_t kBuckets = 2048;
    uint32_t cnt[kBuckets + 1] = {0};
    auto bucket = [&](size_t i) { const uint32_t k = P.pkt_refs[i].bit_len >> 8; return kBuckets - 1 - (k < kBuckets ? k : kBuckets - 1); };
    for (size_t i = 0; i < nf; i++) cnt[bucket(i) + 1]++;
    for (uint32_t k = 0; k < kBuckets; k++) cnt[k + 1] += cnt[k];
    for (size_t i = 0; i < nf; i++) od[cnt[bucket(i)]++] = (int)i;
  }
  if (!pool_pinned) {
    std::memcpy(h + o_pk, P.pkt_pool.base, pool_bytes);
    std::memset(h + o_pk + pool_bytes, 0, 8);
  }
  b->descriptor_bytes = (int64_t)(nf * (sizeof(NvhFrame) + sizeof(NvhPacketRef)) + pool_bytes);
  hipStream_t st = s->ctx->stream;
  uint8_t* base = (uint8_t*)b->blob.p;
  auto t_b = tnow();
  if (nvh_toggles().copy_upload) {
    HIP_TRY(hipMemcpyAsync(base, h, host_bytes, hipMemcpyHostToDevice, st));
    if (pool_pinned) {
      // (the pool is not touched again before collect_parse_result below has synchronised the stream)
      HIP_TRY(hipMemcpyAsync(base + o_pk, P.pkt_pool.base, pool_bytes, hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemsetAsync(base + o_pk + pool_bytes, 0, 8, st));
    }
    NvhParseResult init{};
    init.err_frame = 0x7FFFFFFF;
    init.links_ok = 1;
    init.emit_ok = 1;
    // (a 32-byte pageable source: staged by the runtime before the call returns)
    HIP_TRY(hipMemcpyAsync(base + o_rs, &init, sizeof init, hipMemcpyHostToDevice, st));
  } else {
    // the device fetches its input itself (kernels_parse.hip: k_parse_fetch); both host blocks are page-locked and mapped
    void *h_dev = nullptr, *pool_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&h_dev, h, 0));
    if (pool_pinned) HIP_TRY(hipHostGetDevicePointer(&pool_dev, P.pkt_pool.base, 0));
    const long long n16 = (long long)(host_bytes / 16);  // (offsets are multiples of 256)
    const long long work16 = n16 + (long long)(pool_pinned ? pool_bytes / 16 : 0);
    const unsigned fblocks = (unsigned)std::min<long long>(2048, std::max<long long>(1, (work16 + 255) / 256));
    hipLaunchKernelGGL(k_parse_fetch, dim3(fblocks), dim3(256), 0, st, (const uint4*)h_dev, (uint4*)base, n16,
                       (const uint8_t*)pool_dev, pool_pinned ? base + o_pk : (uint8_t*)nullptr, (long long)pool_bytes,
                       (NvhParseResult*)(base + o_rs));
    HIP_TRY(hipGetLastError());
  }
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
  b->dev_copy_valid = false;
  b->d_ola_list = (const int*)(base + o_ol);
  if (nf) {
    const unsigned blocks = (unsigned)((nf + 63) / 64);
    // Launch shape.  A batch's parse time has a floor -- one lane's serial decode of one packet, ~0.5 ms -- that no shape moves
    // (tools/sweep_parse_shape.sh: 0.54-0.60 ms per 4096 packets for 1-2 packets per wavefront x 4-16 wavefronts per workgroup);
    // what the shape decides is how many parses share the chip.  Workgroups of 16 wavefronts (one copy of the Huffman tables per
    // CU instead of two to four) and up to 4096 wavefronts: a lone 4096-packet batch 0.55 ms (0.54 with the round-4 shape: 4
    // wavefronts, 2048 in all), 32 768 packets 1.17 against 1.23 ms, the corpus pass (16 host threads, each parsing its own
    // file's batches) 2.57 -> 2.01 s.
    const int lanes_env = nvh_toggles().parse_lanes, waves_env = nvh_toggles().parse_waves;
    int kParseWaves = (waves_env >= 1 && waves_env <= 16) ? waves_env : 16;
    int lanes = 1;
    while (lanes < 64 && (nf + (size_t)lanes - 1) / (size_t)lanes > 4096) lanes *= 2;
    if (s->ctx->parse_lanes >= 1) {
      // nvh_ctx_set_parse_lanes: a host with many parses in flight.  A small batch keeps at least 256 wavefronts -- a parse of
      // 260 packets at eight per wavefront is 33 wavefronts that take four times as long (the corpus at a tenth of its length:
      // decode pass 0.25 s with eight lanes throughout, 0.14 s with one)
      // (that bound is the lockstep nest's, whose wavefronts take longer the more packets they hold; the lean walk's do not --
      // k_parse_slab_f: 0.66 ms per wavefront at 8 or 64 packets -- so a batch of a thousand packets and more takes the pool's
      // figure as it is: the fewer wavefronts a parse is, the more parses run side by side)
      int want = s->ctx->parse_lanes;
      const bool lean_batch = T.slab_stride_vecs > 0 && T.dm_in_lds && !T.slab_general && nvh_toggles().parse_cur != 0 &&
                              nvh_toggles().parse_cur != 1 && nf >= 1024;
      if (!lean_batch) while (want > 1 && nf / (size_t)want < 256) want >>= 1;
      lanes = std::max(lanes, want);
    }
    // the lean walk (below) runs one workgroup of up to eight wavefronts per CU (its tables and per-packet rows take most of a CU's
    // LDS): a batch that is several packets per wavefront anyway gets enough of them for its workgroups to run in one round
    if (lanes > 1 && T.slab_stride_vecs > 0 && T.dm_in_lds && !T.slab_general && nvh_toggles().parse_cur != 0 && nvh_toggles().parse_cur != 1)
      while (lanes < 32 && (nf + (size_t)lanes * 8 - 1) / ((size_t)lanes * 8) > 256) lanes *= 2;
    if (lanes_env >= 1 && lanes_env <= 64) lanes = lanes_env;
    // Several packets per wavefront, slab mode: the cursor form (kernels_parse.hip: CUR) in two kernels -- the parse, then the rest
    // of the slab with one wavefront per packet (NVH_PARSE_CUR=0: the lockstep nest, k_parse_slab / _g, the form of rounds 3-5).
    // The walk keeps two bytes per (channel, partition) and NVH_PSTG = 32 entries of staging per packet in LDS next to the tables;
    // workgroups of eight wavefronts where that fits, else four, two, one -- else the lockstep nest.
    const int cur_env = nvh_toggles().parse_cur;
    const bool lean_ok = T.dm_in_lds && !T.slab_general;  // (k_parse_slab_f: one residue pass per frame, visit descriptors in LDS)
    bool cur = slab_mode && lanes > 1 && (cur_env > 0 || (cur_env < 0 && lean_ok));
    const size_t cls_words = 2 * (((size_t)T.cap_parts + 3) / 4);  // class bytes + stage-mask bytes
    size_t sub_words = 0;  // k_parse_slab_f: the second-level tables of the long codes, if they fit behind the records
    if (cur) {
      // (rows per packet of the workgroup: up to eight wavefronts -- two per SIMD -- where they hold 32 packets or fewer each)
      const size_t lds_cap = (size_t)156 * 1024 / 4, tabw = (size_t)(T.lds_words + T.meta_words);
      const size_t subw = (T.sub_words > 1 && !nvh_toggles().no_parse_sub) ? (size_t)T.sub_words : 0;
      int w = (waves_env >= 1 && waves_env <= 8) ? waves_env : 8;
      while (w > 1 && ((size_t)w * (size_t)lanes > 256 || tabw + subw + (size_t)w * (size_t)lanes * (16 + cls_words) > lds_cap)) w >>= 1;
      if (tabw + subw + (size_t)w * (size_t)lanes * (16 + cls_words) <= lds_cap) sub_words = subw;
      else if (tabw + (size_t)w * (size_t)lanes * (16 + cls_words) > lds_cap) cur = false;
      if (cur) kParseWaves = w;
    }
    const bool lean = cur && lean_ok && cur_env != 1;
    const size_t per_wg = (size_t)kParseWaves * (size_t)lanes;
    const unsigned pblocks = (unsigned)((nf + per_wg - 1) / per_wg);
    // per-lane LDS next to the tables, while two workgroups still fit a CU (2 x 80 KB): the residue scratch rows first,
    // then the packets (sized for the longest packet of the batch)
    size_t max_pkt_words = 1;
    for (size_t i = 0; i < nf; i++) max_pkt_words = std::max<size_t>(max_pkt_words, ((size_t)P.pkt_refs[i].bit_len + 31) / 32 + 1);
    const size_t table_words = (size_t)(T.lds_words + T.meta_words);
    // (workgroups of eight and more wavefronts may take a CU's whole LDS: one workgroup per CU still is 2+ wavefronts per SIMD)
    const size_t lds_cap_words = (size_t)(kParseWaves >= 8 ? 156 : 80) * 1024 / 4;
    int scratch_words = T.row_words, pkt_words = (int)max_pkt_words;
    // LDS variant only when both the rows and the longest packet of the batch fit for every lane; else everything per-lane
    // stays in global memory (k_parse_g)
    // slab mode: + one floor scratch block and an error word per wavefront (kernels_parse.hip: floor_to_slab_wave)
    const size_t floor_words = (slab_mode && !cur) ? (size_t)kParseWaves * (NVH_SP_FLOOR_SCRATCH_WORDS + 1) + 8 : 0;  // (cur: the tail kernel's)
    const bool in_lds = !cur && table_words + per_wg * (size_t)(scratch_words + pkt_words) + floor_words <= lds_cap_words;
    if (!in_lds) scratch_words = pkt_words = 0;
    const size_t stage_words = cur ? (size_t)kParseWaves * (size_t)lanes * (16 + cls_words) + sub_words : 0;
    const size_t parse_lds = (table_words + std::max(per_wg * (size_t)(scratch_words + pkt_words), stage_words) + floor_words) * sizeof(uint32_t);
    if (!s->ctx->parse_lds_attr_set) {  // the opt-in is per device: once per context (contexts are single-threaded)
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_g, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab_g, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab_u, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab_c, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab_f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_parse_slab_t, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      s->ctx->parse_lds_attr_set = true;
    }
    // one packet per wavefront (every batch of up to 4096 packets): the wave-uniform form of the slab parser
    const bool uni = slab_mode && in_lds && lanes == 1 && T.dm_in_lds && !nvh_toggles().no_parse_uni;
    if (lean) {
      hipLaunchKernelGGL(k_parse_slab_f, dim3(pblocks), dim3(64 * kParseWaves), parse_lds, st, T,
                         (const uint8_t*)(base + o_pk), (const NvhPacketRef*)(base + o_rf),
                         (int)nf, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch), (NvhResPass*)(base + o_ps), (NvhResOp*)(base + o_op),
                         (uint16_t*)(base + o_lk), (uint16_t*)(base + o_en), (uint16_t*)(base + o_po), (int*)(base + o_sc),
                         (NvhParseResult*)(base + o_rs), lanes, (int)sub_words, 0, (uint4*)b->slab3.p,
                         sorted_parse ? (const int*)(base + o_od) : (const int*)nullptr, (uint32_t*)(base + o_ho) NVH_DBG_LAUNCH);
      pkt_words = -1;  // k_parse_slab_c below: the frames k_parse_slab_f marked, nothing else
    }
    hipLaunchKernelGGL(cur ? k_parse_slab_c : uni ? k_parse_slab_u : slab_mode ? (in_lds ? k_parse_slab : k_parse_slab_g) : (in_lds ? k_parse : k_parse_g),
                       dim3(pblocks), dim3(64 * kParseWaves), parse_lds, st, T,
                       (const uint8_t*)(base + o_pk), (const NvhPacketRef*)(base + o_rf),
                       (int)nf, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch), (NvhResPass*)(base + o_ps), (NvhResOp*)(base + o_op),
                       (uint16_t*)(base + o_lk), (uint16_t*)(base + o_en), (uint16_t*)(base + o_po), (int*)(base + o_sc),
                       (NvhParseResult*)(base + o_rs), lanes, scratch_words, pkt_words, slab_mode ? (uint4*)b->slab3.p : (uint4*)nullptr,
                       sorted_parse ? (const int*)(base + o_od) : (const int*)nullptr, (uint32_t*)(base + o_ho) NVH_DBG_LAUNCH);
    if (cur) {
      // the rest of the slab: one wavefront per packet, four per workgroup; LDS = the setup records + a floor scratch block and
      // an error word per wavefront
      constexpr int kTailWaves = 4;  // (sixteen per workgroup: 207 -> 250 us per 32 768 packets)
      const size_t tail_lds = ((size_t)T.meta_words + 3 + (size_t)kTailWaves * (NVH_SP_FLOOR_SCRATCH_WORDS + 1) + 8) * sizeof(uint32_t);
      hipLaunchKernelGGL(k_parse_slab_t, dim3((unsigned)((nf + kTailWaves - 1) / kTailWaves)), dim3(64 * kTailWaves), tail_lds, st, T,
                         (const uint8_t*)(base + o_pk), (const NvhPacketRef*)(base + o_rf),
                         (int)nf, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch), (NvhResPass*)(base + o_ps), (NvhResOp*)(base + o_op),
                         (uint16_t*)(base + o_lk), (uint16_t*)(base + o_en), (uint16_t*)(base + o_po), (int*)(base + o_sc),
                         (NvhParseResult*)(base + o_rs), 1, 0, 0, (uint4*)b->slab3.p, (const int*)nullptr, (uint32_t*)(base + o_ho) NVH_DBG_LAUNCH);
    }
    // the carried block's execute flags ping-pong together with the carried block (nvh_stream_synth flips carry_cur)
    uint32_t* ce = (uint32_t*)s->carry_exec.p;
    hipLaunchKernelGGL(k_parse_links, dim3(blocks), dim3(64), 0, st, (int)nf, ch, (NvhFrame*)(base + o_fr), (NvhChan*)(base + o_ch),
                       (const uint32_t*)(ce + s->carry_cur), ce + (s->carry_cur ^ 1), b->last_decoded, (NvhParseResult*)(base + o_rs),
                       slab_mode ? (uint4*)b->slab3.p : (uint4*)nullptr, (int)T.slab_stride_vecs);
    HIP_TRY(hipGetLastError());
  }
  auto t_c = tnow();
  rc = collect_parse_result(s, b, (const NvhParseResult*)(base + o_rs));
  auto t_d = tnow();
  if (tprint) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    fprintf(stderr, "batch_upload_gpu: host prep %.2f ms, enqueue %.2f ms, wait for k_parse %.2f ms\n", ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d));
  }
  if (rc != NVH_OK) return rc;
  size_t plane = (size_t)ch * (size_t)s->setup.block1 * sizeof(float);
  rc = b->work.reserve(std::max<size_t>((size_t)b->nframes, 1) * plane);
  if (rc != NVH_OK) return rc;
  if ((rc = poison_planes(s, b, std::max<size_t>((size_t)b->nframes, 1) * plane)) != NVH_OK) return rc;
  if (slab_mode) {
    size_t cap = std::max<size_t>((size_t)b->max_vecs, (size_t)s->setup.block1 / 64 + 8);  // the IMDCT padding of channel 0 overlays the slab area
    b->slab_cap_vecs = (int)((cap + 3) & ~(size_t)3);
    b->slabs_ready = true;
  } else {
    b->slab_stride_vecs = b->slab_cap_vecs = 0;
  }
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
  if (nvh_toggles().copy_upload) {
    HIP_TRY(hipMemcpyAsync(r, d_res, sizeof *r, hipMemcpyDeviceToHost, st));
  } else {
    static_assert(sizeof(NvhParseResult) % 4 == 0 && sizeof(NvhParseResult) <= 256, "k_parse_result_out: one word per lane");
    void* r_dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&r_dev, r, 0));
    hipLaunchKernelGGL(k_parse_result_out, dim3(1), dim3(64), 0, st, d_res, (NvhParseResult*)r_dev);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(nvh_wait_stream(s->ctx, st));
  b->max_ops = r->max_ops;
  b->max_ent = r->max_ent;
  b->max_pass = r->max_pass;
  b->links_ok = r->links_ok != 0;
  b->max_vecs = r->max_vecs;
  b->ola_all = r->emit_ok == 0;  // k_parse_links withdrew an emission candidate: the host's list of left-over frames is short
  // some packet of the batch would have made the managed decoder throw: the batch is parsed again on the host
  if (r->err_frame != 0x7FFFFFFF) return NVH_INTERNAL_REPLAY;
  return NVH_OK;
}

// Paired emission: marks the steady-state overlaps the slab synthesis kernels emit themselves (NvhFrame::emit_flags) for workgroups
// of `fpw` consecutive frames, and lists what is left for k_ola_compact.
static void assign_emission(nvh_stream* s, nvh_batch* b, nvh::FrameBatch& P, int fpw, std::vector<int>& ola_list) {
  // ---- paired emission (nvh_format.h: NVH_EMIT_*): which steady-state overlaps the slab synthesis kernel emits itself ----
  // g "steady": its whole first half lies over the whole second half of frame g - 1, same size, every channel executing in
  // both, whole groups of four samples (k_ola_compact's read-once path has the same contract: kernels.hip, ola_sym).  Even
  // frames emit; an odd steady frame's PCM comes from the even frame in front of it.  Host-parsed batches of mono / stereo
  // streams with blocks 256..2048 only (the execute flags of a GPU-parsed batch are not known here).
  ola_list.clear();
  b->emit_frames = 0;
  b->fpw = 1;
  {
    const int ch = s->setup.channels;
    // mono / stereo with blocks up to 2048 (k_synth_emit: from registers and LDS-staged quarters), or the wide kernel's shapes --
    // up to eight channels, blocks up to 4096 (k_synth8_emit: through the planes and LDS rows)
    // (streams whose frames need the general bin walk run k_synth_g / k_synth8 without paired emission)
    const bool narrow = ch <= 2 && s->setup.block1 <= 2048 && !s->shared->slab_general;
    // (round 4's form -- planes out, read back, LDS rows -- lost to k_synth8 + k_ola_compact on C4, six channels at n = 4096: 123.5 against
    // 116.5 us per 2048 frames, and stayed opt-in.  Round 5's direct form (kernels_synth.hip: synth_emit8_direct -- own quarters stay
    // in LDS, the even frames write no plane, PCM leaves in whole lines through a per-wavefront transposition) wins: 110.3 -> 98.8 us
    // on one stream, 109.8 -> 87.1 us over three; psize 32: 148.7 -> 138.9 / 146.4 -> 119.4.  NVH_NO_EMIT8 switches it off.)
    const bool wide_emit = !narrow && !s->shared->slab_general && ch <= NVH_SLAB_MAX_CH && s->setup.block1 <= 4096 && !nvh_toggles().no_emit8;
    const bool can = !nvh_toggles().no_emit && (narrow || wide_emit) && !P.sequential_ola && s->setup.block0 >= 256;
    const unsigned all_ch = (1u << ch) - 1u;
    // GPU-parse mode: the execute flags are decided inside k_parse (Mapping.cs:104-131); the host marks the candidates from the
    // geometry and k_parse_links has the last word (kernels_parse.hip)
    const bool gp = s->gpu_parse;
    auto full = [&](uint32_t m) { return gp || (m & all_ch) == all_ch; };
    const int nf = (int)P.frames.size();
    auto steady = [&](int g) {
      if (!can || g < 1 || g >= nf) return false;
      const NvhFrame& fr = P.frames[(size_t)g];
      const NvhFrame& pv = P.frames[(size_t)g - 1];
      const int half = fr.n >> 1;
      return fr.n >= 256 && pv.n == fr.n && fr.ov_frame == g - 1 && fr.ov_n == fr.n && fr.start == 0 && fr.emit_start == 0 &&
             fr.emit_count == half && fr.ov_src == half && fr.ov_len == half && full(fr.exec_mask) &&
             full(fr.ov_exec_mask) && full(pv.exec_mask) && ((fr.out_pos * ch) & 3) == 0 &&
             fr.out_pos >= 0 && fr.out_pos < 0x7FFFFFFFll && fr.window_off < 0x7FFFFFFFu;
    };
    // Frame groups (fpw > 1) also emit through block-size switches (Mode.cs:102-151, StreamDecoder.cs:417-463): the overlap of
    // two blocks of n and pn samples is ola_sym's read-once form on m = min(n, pn) -- sample times i and m/2 - 1 - i need the last
    // m/4 values of the later block's first quarter and of the earlier block's third quarter --, and what a long block emits
    // beside its overlaps (the flat parts of its window next to a short neighbour: [start + m/2, n/2) and [n/2, valid)) only that
    // block contributes to: its own workgroup emits it.  "pairable": the geometry the window flags promise is the geometry the
    // stream has (consistent flags, no end-of-stream trim), everything in whole groups of four samples.
    auto pairable = [&](int g) {
      if (!can || g < 1 || g >= nf) return false;
      const NvhFrame& fr = P.frames[(size_t)g];
      const NvhFrame& pv = P.frames[(size_t)g - 1];
      if (fr.n < 256 || pv.n < 256) return false;
      const int m = fr.n < pv.n ? fr.n : pv.n, half = fr.n >> 1;
      return fr.ov_frame == g - 1 && fr.ov_n == pv.n && fr.ov_window_off == pv.window_off && fr.start == (fr.n >> 2) - (m >> 2) &&
             fr.ov_len == (m >> 1) && fr.ov_src == 3 * (pv.n >> 2) - (m >> 2) && fr.emit_start == fr.start &&
             fr.emit_count == fr.valid - fr.start && fr.valid >= half && fr.valid <= half + (fr.n >> 2) && (fr.valid & 63) == 0 &&
             full(fr.exec_mask) && full(fr.ov_exec_mask) && full(pv.exec_mask) && ((fr.out_pos * ch) & 3) == 0 &&
             fr.out_pos >= 0 && fr.out_pos < 0x7FFFFFFFll && fr.window_off < 0x7FFFFFFFu;
    };
    // Who emits an overlap.  Frames go to workgroups in groups of fpw consecutive frames (1: k_synth / k_synth_emit, the odd
    // frames first; 2, 4: kernels_synth.hip, frame groups -- the groups with an odd index first); a steady overlap inside a group
    // is emitted by that group from LDS, one between two groups by the group of the second launch (even index), which finds the
    // other group's quarter in the planes.  SELF: the frame's own workgroup emits its PCM; NEXT: it emits frame g + 1's.
    if (narrow && can && fpw > 1) b->fpw = fpw;
    const int gw = b->fpw;
    auto emits = [&](int g) { return gw > 1 ? pairable(g) : steady(g); };
    for (int g = 0; g < nf; g++) {
      NvhFrame& fr = P.frames[(size_t)g];
      fr.emit_flags = 0;
      const int k = g % gw;
      const bool second_launch = ((g / gw) & 1) == 0;
      if (emits(g)) {
        fr.emit_flags |= NVH_EMIT_DONE;
        if (k > 0 || second_launch) fr.emit_flags |= NVH_EMIT_SELF;
        b->emit_frames++;
      }
      if ((k < gw - 1 || second_launch) && emits(g + 1)) fr.emit_flags |= NVH_EMIT_NEXT;
    }
    // the batch's first frame over the carried tail of the batch before (same geometry, the tail stored fully windowed; frame
    // groups: a long block in front of a short one emits the flat part behind its first half as well)
    if (can && nf > 0) {
      NvhFrame& fr = P.frames[0];
      const int half = fr.n >> 1;
      const bool tail_ok = gw > 1 ? (fr.emit_count == fr.valid - fr.start && fr.valid >= half && fr.valid <= half + (fr.n >> 2) && (fr.valid & 63) == 0)
                                  : fr.emit_count == half;
      if (fr.n >= 256 && fr.ov_frame == -2 && fr.ov_n == fr.n && fr.start == 0 && fr.emit_start == 0 && tail_ok &&
          fr.ov_src == half && fr.ov_len == half && full(fr.exec_mask) && ((fr.out_pos * ch) & 3) == 0 &&
          fr.out_pos >= 0 && fr.out_pos < 0x7FFFFFFFll) {
        fr.emit_flags |= NVH_EMIT_SELF | NVH_EMIT_SELF_CARRY | NVH_EMIT_DONE;
        b->emit_frames++;
      }
    }
    // Worth it for batches that are mostly steady state: the second launch and the list-driven k_ola_compact cost a stream of
    // mixed block sizes more than the emission saves when it runs alone (C3, 256/2048 Markov chain, one HIP stream: 45.0 us
    // per 4096 frames against 41.6 us; three streams: 24.5 against 25.1).  NVH_EMIT_ALWAYS lifts the threshold.
    {
      int decoded = 0;
      for (const NvhFrame& fr : P.frames) decoded += fr.n != 0;
      if (!nvh_toggles().emit_always && (long long)b->emit_frames * 8 < (long long)decoded * 7) {
        for (NvhFrame& fr : P.frames) fr.emit_flags = 0;
        b->emit_frames = 0;
        b->fpw = 1;
      }
    }
    // what is left for k_ola_compact: every other frame that emits samples.  The block that becomes the carried tail is written
    // by its own workgroup (whole groups of four samples per channel: always, n >= 256).
    int last = -1;
    for (int i = nf - 1; i >= 0; --i)
      if (P.frames[(size_t)i].n != 0) { last = i; break; }
    if (b->emit_frames > 0 && last >= 0 && P.frames[(size_t)last].n >= 256) P.frames[(size_t)last].emit_flags |= NVH_EMIT_CARRY_OUT;
    for (int g = 0; g < nf; g++) {
      const NvhFrame& fr = P.frames[(size_t)g];
      if ((fr.emit_count > 0 && !(fr.emit_flags & NVH_EMIT_DONE)) || (g == last && !(fr.emit_flags & NVH_EMIT_CARRY_OUT))) ola_list.push_back(g);
    }
  }
}

int batch_upload(nvh_stream* s, nvh_batch* b) {
  nvh::FrameBatch& P = s->pending;
  b->s = s;
  b->nframes = (int)P.frames.size();
  b->chan_frames = (int)P.chans.size();
  b->pcm_samples = P.pcm_samples;
  b->sequential_ola = P.sequential_ola;
  b->last_decoded = -1;
  b->max_ops = b->max_ent = b->max_pass = 0;
  b->slabs_ready = false;
  b->slab_host = false;
  b->d_slabs = nullptr;
  b->slab_stride_vecs = b->slab_cap_vecs = 0;
  b->links_ok = P.links_ok && P.op_link.size() == P.ops.size();
  for (const NvhFrame& fr : P.frames) {
    if ((int)fr.op_count > b->max_ops) b->max_ops = (int)fr.op_count;
    if ((int)fr.ent_count > b->max_ent) b->max_ent = (int)fr.ent_count;
    if ((int)(fr.pass_end - fr.pass_begin) > b->max_pass) b->max_pass = (int)(fr.pass_end - fr.pass_begin);
    // kernels index channel records by frame: every frame owns exactly `channels` of them (host_parse.cpp)
    if (fr.chan_off != (uint32_t)((size_t)(&fr - P.frames.data()) * (size_t)s->setup.channels)) return NVH_ERR_RUNTIME;
  }
  for (int i = b->nframes - 1; i >= 0; --i)
    if (P.frames[(size_t)i].n != 0) {
      b->last_decoded = i;
      break;
    }

  std::vector<int> ola_list;
  assign_emission(s, b, P, nvh_toggles().fpw, ola_list);
  b->ola_count = (int)ola_list.size();
  b->d_ola_list = nullptr;
  b->ola_all = false;

  b->stats[0] = (int64_t)P.frames.size(); b->stats[1] = (int64_t)P.chans.size(); b->stats[2] = (int64_t)P.passes.size();
  b->stats[3] = (int64_t)P.ops.size(); b->stats[4] = (int64_t)P.entries.size(); b->stats[5] = (int64_t)P.posts.size();
  b->stats[6] = (int64_t)P.coeffs.size();
  if (s->gpu_parse) {
    if (b->fpw > 1) {  // a setup whose worst-case slabs do not leave room for a group's LDS: one frame per workgroup
      b->slab_stride_vecs = b->slab_cap_vecs = s->shared->parse.slab_stride_vecs;
      if (!slab_size_ok(b)) assign_emission(s, b, P, 1, ola_list);
      b->ola_count = (int)ola_list.size();
    }
    int rc = batch_upload_gpu(s, b, ola_list);
    if (rc != NVH_INTERNAL_REPLAY) {
      if (rc == NVH_OK) s->replay.clear();
      return rc;
    }
    if ((rc = replay_on_host(s)) != NVH_OK) return rc;
    s->gpu_parse = false;  // this batch goes up as host-parsed descriptors
    rc = batch_upload(s, b);
    s->gpu_parse = true;
    s->replay.clear();
    return rc;
  }
  // ---- host-parsed batches inside the slab kernels' contract: the parser's thread writes the slabs themselves (host_slab.cpp:
  // Floor1 unwrap + segment lists, chain-major pair records) and only they travel -- with the frame / channel records, which
  // k_ola_compact reads for the frames outside the steady state.  No descriptor arrays, no conversion kernel.
  if (slab_shape_ok(b)) {
    nvh::SlabBatch& SB = s->slab_build;
    int rc = nvh::build_slabs(s->setup, s->shared->slab, P, SB);
    if (rc != NVH_OK && rc != NVH_ERR_UNSUPPORTED) return rc;
    if (rc == NVH_OK) {
      size_t stride = std::max<size_t>(SB.max_vecs, (size_t)s->setup.block1 / 64 + 8);  // the IMDCT padding of channel 0 overlays the slab area
      stride = (stride + 3) & ~(size_t)3;
      b->slab_stride_vecs = b->slab_cap_vecs = (int)stride;
      b->slab_host = true;
      if (!slab_size_ok(b) && b->fpw > 1) {  // no room for a group's LDS: one frame per workgroup, the slab headers written again
        assign_emission(s, b, P, 1, ola_list);
        b->ola_count = (int)ola_list.size();
        rc = nvh::build_slabs(s->setup, s->shared->slab, P, SB);
        if (rc != NVH_OK) return rc;
      }
      if (!slab_size_ok(b)) b->slab_host = false;
    }
    if (b->slab_host) {
      const size_t nf = P.frames.size(), stride = (size_t)b->slab_stride_vecs;
      auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
      const size_t o_fr = 0;
      const size_t o_ch = al(o_fr + std::max<size_t>(nf, 1) * sizeof(NvhFrame));
      const size_t o_ol = al(o_ch + std::max<size_t>(P.chans.size(), 1) * sizeof(NvhChan));
      const size_t o_sl = al(o_ol + std::max<size_t>(ola_list.size(), 1) * sizeof(int));
      const size_t total = al(o_sl + nf * stride * 16 + 4096);  // the speculative first DMA piece of the last slab stays inside the allocation
      if ((rc = b->blob.reserve(total)) != NVH_OK) return rc;
      if ((rc = b->h_blob.reserve(total)) != NVH_OK) return rc;
      uint8_t* h = (uint8_t*)b->h_blob.p;
      if (nf) std::memcpy(h + o_fr, P.frames.data(), nf * sizeof(NvhFrame));
      if (!P.chans.empty()) std::memcpy(h + o_ch, P.chans.data(), P.chans.size() * sizeof(NvhChan));
      if (!ola_list.empty()) std::memcpy(h + o_ol, ola_list.data(), ola_list.size() * sizeof(int));
      int64_t slab_bytes = 0;
      for (size_t f = 0; f < nf; f++) {
        const size_t v = (size_t)(SB.first[f + 1] - SB.first[f]);
        std::memcpy(h + o_sl + f * stride * 16, &SB.data[SB.first[f]], v * 16);
        slab_bytes += (int64_t)v * 16;
      }
      b->descriptor_bytes = (int64_t)(nf * sizeof(NvhFrame) + P.chans.size() * sizeof(NvhChan)) + slab_bytes;
      hipStream_t st = s->ctx->stream;
      HIP_TRY(hipMemcpyAsync(b->blob.p, h, o_sl + nf * stride * 16, hipMemcpyHostToDevice, st));
      const uint8_t* base = (const uint8_t*)b->blob.p;
      b->dev.frames = (const NvhFrame*)(base + o_fr);
      b->dev.chans = (const NvhChan*)(base + o_ch);
      b->dev.passes = nullptr; b->dev.ops = nullptr; b->dev.op_link = nullptr; b->dev.entries = nullptr; b->dev.posts = nullptr; b->dev.coeffs = nullptr;
      b->d_ola_list = (const int*)(base + o_ol);
      b->d_slabs = (const uint4*)(base + o_sl);
      b->dev.nframes = b->nframes;
      b->dev.pad = 0;
      b->dev_copy_valid = false;
      b->slabs_ready = true;
      const size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
      if ((rc = b->work.reserve(std::max<size_t>((size_t)b->nframes, 1) * plane)) != NVH_OK) return rc;
      if ((rc = poison_planes(s, b, std::max<size_t>((size_t)b->nframes, 1) * plane)) != NVH_OK) return rc;
      P.clear();
      s->parser->begin_batch();
      return NVH_OK;
    }
  }
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
  size_t o_ol = add(ola_list.data(), ola_list.size(), sizeof(int));
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
  b->d_ola_list = (const int*)(base + o_ol);
  b->dev.nframes = b->nframes;
  b->dev.pad = 0;
  b->dev_copy_valid = false;

  size_t plane = (size_t)s->setup.channels * (size_t)s->setup.block1 * sizeof(float);
  rc = b->work.reserve(pad1((size_t)b->nframes) * plane);
  if (rc != NVH_OK) return rc;
  if ((rc = poison_planes(s, b, pad1((size_t)b->nframes) * plane)) != NVH_OK) return rc;
  P.clear();
  s->parser->begin_batch();
  return NVH_OK;
}

// The batch's parameter block in device memory (the frame-loop kernels read it from there instead of holding its pointers
// in scalar registers); 80 bytes from pageable memory, staged by the runtime before the call returns, once per upload.
static int upload_dev_copy(nvh_batch* b, hipStream_t st) {
  if (b->dev_copy_valid && b->dev_copy.p) return NVH_OK;
  int rc = b->dev_copy.reserve(sizeof(NvhDevBatch));
  if (rc != NVH_OK) return rc;
  HIP_TRY(hipMemcpyAsync(b->dev_copy.p, &b->dev, sizeof(NvhDevBatch), hipMemcpyHostToDevice, st));
  b->dev_copy_valid = true;
  return NVH_OK;
}

// The wide form (k_synth8: 512 threads, up to eight channels, blocks up to 4096) against k_synth (256 threads, mono / stereo,
// blocks up to 2048, 8 workgroups per CU).
#ifndef NVH_SYNTH_NT
#define NVH_SYNTH_NT 256  // threads of a k_synth workgroup (kernels_synth.hip; build variants: tools/build_variant.py)
#endif
static bool slab_wide(const nvh_stream* s) { return s->setup.channels > 2 || s->setup.block1 > 2048; }

// The batch's largest slab in 16-byte units (nvh_format.h: NvhSlabHdr): known exactly for slabs the host parser's thread wrote,
// reported by k_parse for GPU-parsed batches (before its launch: the setup's worst case, for the size check).
static size_t slab_bound_vecs(const nvh_batch* b) { return (size_t)b->slab_cap_vecs; }

// A mono / stereo batch with paired emission whose frames go to workgroups in groups (k_synth_group2 / 4)
static bool slab_groups(const nvh_batch* b) { return !slab_wide(b->s) && b->emit_frames > 0 && b->fpw > 1; }

// The LDS slab area in 16-byte units: the batch's largest slab -- and, for a batch with paired emission (k_synth only), room
// for the neighbours' quarters that are staged over constants + slab in front of the first transform slice's padding
// (kernels_synth.hip: synth_emit).
static size_t slab_lds_vecs(const nvh_batch* b) {
  const nvh_stream* s = b->s;
  size_t v = slab_bound_vecs(b);
  if (!slab_wide(s) && b->emit_frames > 0) {
    const size_t ch = (size_t)s->setup.channels, b1 = (size_t)s->setup.block1;
    if (b->fpw > 1) {
      // frame groups: the two staged quarters of every channel lie over the constants and the group's fpw slab areas together
      const size_t need_words = ch * (b1 / 2), have = (size_t)s->shared->synth_const_vecs * 4, g = (size_t)b->fpw;
      if (need_words > have) v = std::max(v, (need_words - have + 4 * g - 1) / (4 * g));
    } else {
      const size_t need_words = ch * (b1 / 2) + b1 / 16, have = (size_t)s->shared->synth_const_vecs * 4;
      if (need_words > have) v = std::max(v, (need_words - have + 3) / 4);
    }
  }
  return (v + 3) & ~(size_t)3;
}

// Dynamic LDS of the slab synthesis kernel for this batch (kernels_synth.hip: LDS map): constants + the largest slab + the
// spectra, and room for the transforms' slices where they are laid over everything (k_synth8).
static size_t slab_lds_bytes(const nvh_batch* b) {
  const nvh_stream* s = b->s;
  const size_t ch = (size_t)s->setup.channels, b1 = (size_t)s->setup.block1;
  if (slab_groups(b)) {
    // kernels_synth.hip, frame groups: the larger of the walk's map (constants | fpw slab areas | fpw x spectra) and the transforms'
    // (staged quarters | fpw x channels slices), + the overlaps' parameter table
    const size_t g = (size_t)b->fpw;
    const size_t walk = (size_t)s->shared->synth_const_vecs * 4 + g * (slab_lds_vecs(b) * 4 + ch * (b1 / 2));
    const size_t xform = ch * (b1 / 2) + g * ch * (b1 / 2 + b1 / 16);
    return (std::max(walk, xform) + 8 * (2 * g + 1)) * sizeof(float);
  }
  size_t words = (size_t)s->shared->synth_const_vecs * 4 + slab_lds_vecs(b) * 4 + ch * (b1 / 2) + b1 / 16;
  if (slab_wide(s)) words = std::max(words, ch * (b1 / 2 + b1 / 16));
  if (b1 > 4096) words += ch * (b1 / 2 + b1 / 16);  // n = 8192: the transforms' slices lie behind the spectra, not over them
  return words * sizeof(float);
}

// Whether a batch takes the slab synthesis kernels (kernels_synth.hip): Floor1 only, every residue on the pair path (lattice
// books of even dimension, no aliasing partitions), Residue2 over more than two channels only with partitions that are whole
// multiples of two bins of every channel, at most NVH_SLAB_MAX_COUPLE coupling steps, at most one residue pass per frame,
// up to eight channels, blocks 256..4096, slabs within 16-bit section offsets and the CU's LDS.
static bool slab_shape_ok(const nvh_batch* b) {
  const nvh_stream* s = b->s;
  const NvhToggles& T = nvh_toggles();
  if (!s->shared->slab_setup_ok || !b->links_ok || b->sequential_ola || b->block_only || b->descriptors_only || (b->max_pass > 1 && !s->shared->slab_general)) return false;
  if (T.no_slab || T.unfused || T.no_fused_imdct || T.no_compact) return false;
  if (slab_wide(s) && T.no_gen8) return false;  // NVH_NO_GEN8 keeps its meaning: more than four channels through k_spectrum_gen
  return true;
}

static bool slab_size_ok(const nvh_batch* b) {
  const nvh_stream* s = b->s;
  if (slab_bound_vecs(b) > 0xFFFFu) return false;
  return slab_lds_bytes(b) + (size_t)nvh_toggles().lds_pad <= ((slab_wide(s) || slab_groups(b)) ? (size_t)160 * 1024 - 1024 : (size_t)64 * 1024);
}

static bool slab_path(const nvh_batch* b) { return slab_shape_ok(b) && slab_size_ok(b); }

int batch_launch(nvh_batch* b, const float* carry, float* carry_out, float* d_pcm, bool timing, float* kernel_ms, hipEvent_t* ext_ev) {
  nvh_stream* s = b->s;
  hipStream_t st = s->ctx->stream;
  if (b->nframes == 0) return NVH_OK;
  const int ch = s->setup.channels;
  float* work = (float*)b->work.p;
  int* flags = (int*)s->flags.p;
  const size_t lds = (size_t)s->setup.block1 * sizeof(float);
  ScopedEvent sev[5];
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // timing: five event records bracket the four slots.  With the caller's events (ext_ev) nothing is waited for here: the caller
  // queues many launches back to back and reads the events afterwards, so a slot is a kernel between two markers in a full queue
  // rather than a kernel dispatched onto an idle GPU.
  if (timing)
    for (int k = 0; k < 5; k++) {
      if (ext_ev) {
        ev[k] = ext_ev[k];
        continue;
      }
      int rc = sev[k].create();
      if (rc != NVH_OK) return rc;
      ev[k] = sev[k].e;
    }
  if (timing) HIP_TRY(hipEventRecord(ev[0], st));
  // compact hand-over (two independent quarters per block, windowed in the overlap kernel) whenever no overlap
  // ever modifies a tail (the in-place sequential form needs the full windowed blocks)
  const NvhToggles& T = nvh_toggles();
  const bool no_compact = T.no_compact, no_fused_imdct = T.no_fused_imdct;
  const bool compact = s->setup.block0 >= 256 && !b->sequential_ola && !no_compact && !b->block_only;
  // spectrum + IMDCT in one kernel: pair-path / fused-tail streams with block sizes the single-pass wavefront IMDCT covers
  const bool fast = s->fast_spectrum && b->links_ok;  // k_spectrum proper (pair path by chain walk, fused tail)
  bool fuse_imdct = compact && fast && s->setup.block1 <= 2048 && !no_fused_imdct;  // and the LDS-resident path is taken (below)
  bool fuse_gen8 = false;  // k_spectrum_gen8_imdct (below)
  // ---- slab synthesis kernels (kernels_synth.hip): spectrum + inverse MDCT from per-frame slabs fetched by LDS-DMA ----
  bool slab_done = false;
  bool emitted = false;  // paired emission ran: k_synth wrote the PCM of the frames marked NVH_EMIT_DONE
  if (b->slabs_ready && compact && !no_fused_imdct) {
    NvhSynthArgs A;
    A.consts = s->shared->synth_consts;
    A.slabs = b->slab_host ? b->d_slabs : (const uint4*)b->slab3.p;
    A.work = work;
    A.err = flags;
    for (int w = 0; w < 2; w++) {
      A.mdct_a[w] = s->dev.mdct_a[w]; A.mdct_b[w] = s->dev.mdct_b[w]; A.mdct_c[w] = s->dev.mdct_c[w]; A.mdct_tw[w] = s->dev.mdct_tw[w];
    }
    A.ipool = s->dev.ipool;
    A.vq = s->dev.vq;
    A.const_vecs = s->shared->synth_const_vecs;
    A.stride_vecs = b->slab_stride_vecs;
    A.cap_vecs = b->slab_cap_vecs;
    A.lds_vecs = (int)slab_lds_vecs(b);
    A.channels = ch;
    A.block1 = s->setup.block1;
    A.f0 = 0; A.fstep = 1;
    A.nframes = b->nframes;
    A.prefetch_prev = 0;
    A.xcd_map = 0;
    A.walk_two = nvh_toggles().no_walk_two ? 0 : 1;
    const bool wide = slab_wide(s);
    // paired emission (nvh_format.h: NVH_EMIT_*): the host marked the frames at upload; it needs the PCM buffer and the slabs
    // in frame order
    emitted = b->emit_frames > 0 && d_pcm != nullptr && !b->block_only && !T.no_emit && !s->shared->slab_general;
    A.pcm = emitted ? d_pcm : nullptr;
    A.windows = s->dev.windows;
    A.clip = s->clip;
    A.clipped_flag = flags + 1;
    A.carry = carry;
    A.carry_out = emitted ? carry_out : nullptr;
    A.frames = b->dev.frames;
    const size_t synth_lds = slab_lds_bytes(b) + (size_t)T.lds_pad;
    if (synth_lds > 64 * 1024 && !s->ctx->synth_lds_attr_set) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_synth8, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_synth8_emit, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_synth8_g, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      s->ctx->synth_lds_attr_set = true;
    }
    if (timing) HIP_TRY(hipEventRecord(ev[1], st));  // slot 0 stays empty: slot 1 = the synthesis kernel
    b->slot_name[0] = "-";
    const bool narrow_general = !wide && s->shared->slab_general;  // (never with paired emission: batch_upload)
    const bool wide_general = wide && s->shared->slab_general;
    b->slot_name[1] = wide ? (wide_general ? "k_synth8_g" : "k_synth8") : (narrow_general ? "k_synth_g" : "k_synth");
    if (wide && emitted) {
      // odd frames first, then the even frames, which overlap-add the steady-state overlaps they take part in (synth_emit8)
      A.fstep = 2;
      A.f0 = 1;
      if (b->nframes > 1) hipLaunchKernelGGL(k_synth8, dim3((unsigned)(b->nframes / 2)), dim3(512), synth_lds, st, A NVH_DBG_LAUNCH);
      A.f0 = 0;
      hipLaunchKernelGGL(k_synth8_emit, dim3((unsigned)((b->nframes + 1) / 2)), dim3(512), synth_lds, st, A NVH_DBG_LAUNCH);
    } else if (wide_general) hipLaunchKernelGGL(k_synth8_g, dim3((unsigned)b->nframes), dim3(512), synth_lds, st, A NVH_DBG_LAUNCH);
    else if (wide) hipLaunchKernelGGL(k_synth8, dim3((unsigned)b->nframes), dim3(512), synth_lds, st, A NVH_DBG_LAUNCH);
    else if (narrow_general) hipLaunchKernelGGL(k_synth_g, dim3((unsigned)b->nframes), dim3(256), synth_lds, st, A NVH_DBG_LAUNCH);
    else if (!emitted) hipLaunchKernelGGL(k_synth, dim3((unsigned)b->nframes), dim3(NVH_SYNTH_NT), synth_lds, st, A NVH_DBG_LAUNCH);
    else if (b->fpw > 1) {
      // frame groups (kernels_synth.hip): fpw consecutive frames per workgroup, the overlaps inside a group on chip.  The groups
      // with an odd index first (they leave their outer quarters in the planes), then the even ones, which emit the overlaps
      // between groups as well.
      const int gw = b->fpw, ngroups = (b->nframes + gw - 1) / gw;
      if (synth_lds > 64 * 1024 && !s->ctx->group_lds_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_synth_group2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_synth_group4, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        s->ctx->group_lds_attr_set = true;
      }
      auto kern = gw == 2 ? k_synth_group2 : k_synth_group4;
      const unsigned nt = gw == 2 ? 256u : 512u;
      if (T.debug_occ) {
        int nb = -1;
        hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, (int)nt, synth_lds);
        hipFuncAttributes fa;
        (void)hipFuncGetAttributes(&fa, (const void*)kern);
        fprintf(stderr, "frame groups of %d: %u threads, lds %zu B (constants %d, slab area %d, largest slab %d vecs), occupancy %d WG/CU (err %d), regs %d\n",
                gw, nt, synth_lds, A.const_vecs, A.lds_vecs, A.cap_vecs, nb, (int)oe, fa.numRegs);
      }
      A.fstep = 2 * gw;
      A.f0 = gw;
      static const bool group_prefetch = std::getenv("NVH_GROUP_PREFETCH") != nullptr;  // (measured: 181.6 with, 184.5 M frames/s without)
      A.prefetch_prev = group_prefetch ? 1 : 0;
      if (ngroups / 2 > 0) hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups / 2)), dim3(nt), synth_lds, st, A NVH_DBG_LAUNCH);
      A.f0 = 0;
      A.prefetch_prev = 0;
      hipLaunchKernelGGL(kern, dim3((unsigned)((ngroups + 1) / 2)), dim3(nt), synth_lds, st, A NVH_DBG_LAUNCH);
    } else {
      // odd frames first (their planes are what the even frames overlap-add with), then the even frames, which emit
      // (the odd frames never emit: the plain kernel, or the one that can write the carried tail when the last decoded block is odd)
      A.fstep = 2;
      A.f0 = 1;
      A.prefetch_prev = nvh_toggles().no_prefetch ? 0 : 1;
      A.xcd_map = nvh_toggles().xcd_map ? 1 : 0;  // opt-in: one stream 36.3 -> 35.6 us per pass, three streams 174 -> 172 M frames/s
      if (b->nframes > 1) {
        if (b->last_decoded >= 0 && (b->last_decoded & 1))
          hipLaunchKernelGGL(k_synth_tail, dim3((unsigned)(b->nframes / 2)), dim3(NVH_SYNTH_NT), synth_lds, st, A NVH_DBG_LAUNCH);
        else
          hipLaunchKernelGGL(k_synth, dim3((unsigned)(b->nframes / 2)), dim3(NVH_SYNTH_NT), synth_lds, st, A NVH_DBG_LAUNCH);
      }
      A.f0 = 0;
      A.prefetch_prev = 0;
      hipLaunchKernelGGL(k_synth_emit, dim3((unsigned)((b->nframes + 1) / 2)), dim3(NVH_SYNTH_NT), synth_lds, st, A NVH_DBG_LAUNCH);
    }
    if (emitted) b->slot_name[1] = wide ? "k_synth8+k_synth8_emit" : (b->fpw == 2 ? "k_synth_group2" : b->fpw == 4 ? "k_synth_group4" : "k_synth+k_synth_emit");  // odd frames, then the emitting even frames
    slab_done = true;
    fuse_gen8 = true;  // the inverse MDCT is inside: no transform kernel behind it
  }
  // Fused spectrum kernel when a frame's spectrum (+ staged side information) fits the default 64 KB dynamic
  // LDS window; LDS map in kernels_spectrum.hip.
  if (!slab_done) {
    const bool has_floor0 = s->has_floor0;
    // more than four channels without Floor0: 8 wavefronts per workgroup (k_spectrum_gen8), one floor scratch block each
    const bool no_gen8 = T.no_gen8;
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
    if (T.unfused) words = 1u << 20;  // test aid: force the unfused kernels below
    if ((words + (size_t)(s->setup.block1 / 16)) * 4 > 64 * 1024) fuse_imdct = false;
    const size_t lds_pad = (size_t)T.lds_pad;  // occupancy experiments
    // A frame that does not fit the default 64 KB dynamic-LDS window (six channels at n = 4096 with full-depth packets:
    // 48 KB of spectrum + ~25 KB of staged ops and entries) still fits the CU's 160 KB: the general kernels opt in to a larger
    // window, one workgroup per CU, instead of falling back to the global-memory kernels (measured on the C4 full-depth
    // stream: k_residue + k_couple_floor 510 us per 2048 frames).
    size_t lds_limit = 64 * 1024;
    // (not for residues that replay the reference's partition order, one vector write at a time: a lone 8-wavefront workgroup
    // per CU is the worst shape for that -- 1135 us against 840 us for the global-memory kernels on the psize-32 C4 stream)
    if (!has_floor0 && !fast && !T.unfused && !s->shared->has_sequential && words * 4 > lds_limit && words * 4 <= 152 * 1024) {
      if (!s->ctx->big_lds_attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_spectrum_gen, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_spectrum_gen8, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_spectrum_gen8_imdct, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)k_spectrum_gen_imdct, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        s->ctx->big_lds_attr_set = true;
      }
      lds_limit = 152 * 1024;
    }
    if (words * 4 <= lds_limit) {
      if (timing) HIP_TRY(hipEventRecord(ev[1], st));  // slot 0 stays empty: slot 1 = fused spectrum kernel
      b->slot_name[0] = "-";
      b->slot_name[1] = has_floor0 ? "k_spectrum_f0" : (fast ? "k_spectrum" : "k_spectrum_gen");
      if (has_floor0) {
        hipLaunchKernelGGL(k_spectrum_f0, dim3((unsigned)b->nframes), dim3(256), words * 4, st, s->dev, b->dev, work, flags,
                           cap_pass, cap_ops, cap_ent);
      } else if (gen8) {
        // the inverse MDCT in the same workgroup (one wavefront per channel, slices laid over the dead floor scratch, side
        // information and spectra): block sizes the in-register spectrum covers, and the padding has to fit in front of the spectra
        const size_t front_words = words - (size_t)ch * (size_t)(s->setup.block1 / 2);
        fuse_gen8 = compact && !no_fused_imdct && ch <= 8 && s->setup.block1 <= 4096 &&
                    256 + (size_t)ch * (size_t)(s->setup.block1 / 16) <= front_words;
        b->slot_name[1] = fuse_gen8 ? "k_spectrum_gen8_imdct" : "k_spectrum_gen8";
        hipLaunchKernelGGL(fuse_gen8 ? k_spectrum_gen8_imdct : k_spectrum_gen8, dim3((unsigned)b->nframes), dim3(512), words * 4, st, s->dev,
                           b->dev, work, flags, cap_pass, cap_ops, cap_ent NVH_DBG_LAUNCH);
      } else if (!fast) {
        // the inverse MDCT in the same workgroup where a wavefront per channel exists: mono / stereo setups take the kernel's
        // fused tail and k_spectrum_imdct's transform (one n/16-float pad behind the spectra), three and four channels the general
        // tail and the transform of k_spectrum_gen8_imdct (slices over the dead front of the LDS area)
        size_t extra = 0;
        if (compact && !no_fused_imdct && ch <= 4) {
          if (s->dev.fused_tail_ok) {
            fuse_gen8 = s->setup.block1 <= 2048 && (words + (size_t)(s->setup.block1 / 16)) * 4 <= lds_limit;
            if (fuse_gen8) extra = (size_t)(s->setup.block1 / 16) * 4;
          } else {
            const size_t front_words = words - (size_t)ch * (size_t)(s->setup.block1 / 2);
            fuse_gen8 = s->setup.block1 <= 4096 && 256 + (size_t)ch * (size_t)(s->setup.block1 / 16) <= front_words;
          }
        }
        if (fuse_gen8) b->slot_name[1] = "k_spectrum_gen_imdct";
        hipLaunchKernelGGL(fuse_gen8 ? k_spectrum_gen_imdct : k_spectrum_gen, dim3((unsigned)b->nframes), dim3(256), words * 4 + extra, st,
                           s->dev, b->dev, work, flags, cap_pass, cap_ops, cap_ent NVH_DBG_LAUNCH);
      } else {
        if (T.debug_occ) {
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
                             st, s->dev, b->dev, work, flags, cap_pass, cap_ops, cap_ent NVH_DBG_LAUNCH);
        } else {
          hipLaunchKernelGGL(k_spectrum, dim3((unsigned)b->nframes), dim3(256), words * 4 + lds_pad, st, s->dev, b->dev, work, flags,
                             cap_pass, cap_ops, cap_ent NVH_DBG_LAUNCH);
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
  const size_t plane_bytes = (size_t)ch * (size_t)s->setup.block1 * sizeof(float);
  {
    b->slot_name[2] = (fuse_imdct || fuse_gen8) ? "-" : compact ? "k_imdct_compact" : (s->setup.block0 >= 256 ? "k_imdct_wave" : "k_imdct_window");
    b->slot_name[3] = compact ? "k_ola_compact" : (!b->sequential_ola ? "k_ola_emit" : "k_ola_emit_seq");
    if (fuse_imdct || fuse_gen8)
      ;  // done inside k_spectrum_imdct / k_spectrum_gen8_imdct
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
      // (frame order by XCD -- blocks f and f-1, which share a quarter, behind the same L2 -- measured slower: 23.3 us, not 21.8)
      // 128 lanes per frame: 21.4 us instead of 24.8 us on its own (more loads in flight per frame); with two batches
      // in flight it is a wash against 64, and 256 lanes start to take wave slots from the other batch's spectrum kernel
      const int ola_env = T.ola_threads;
      const int ola_threads = (ola_env == 64 || ola_env == 128 || ola_env == 256) ? ola_env : 128;
      // large frames are shared by several workgroups (kernels.hip): aim at one group of four sample times per lane
      int segs = T.ola_segs > 0 ? T.ola_segs : (int)(((size_t)s->setup.block1 / 8 * (size_t)ch) / (size_t)(2 * 256));
      if (segs < 1) segs = 1;
      if (segs > 8) segs = 8;
      if (segs > (s->setup.block1 / 8) / ola_threads) segs = (s->setup.block1 / 8) / ola_threads;
      if (segs < 1) segs = 1;
      // more than two channels: the steady-state path splits a frame into runs of NVH_OLA_GW groups of four sample times, one
      // workgroup each, and interleaves through LDS (ola_sym_lds)
      if (ch > 2 && !T.no_ola_sym && T.ola_segs <= 0) segs = ((s->setup.block1 / 16) + NVH_OLA_GW - 1) / NVH_OLA_GW;
      if (!emitted)
        hipLaunchKernelGGL(k_ola_compact, dim3((unsigned)b->nframes, (unsigned)segs), dim3((unsigned)ola_threads), 0, st, s->dev, b->dev,
                           (const float*)work, carry, d_pcm, s->clip, flags + 1, carry_out, b->last_decoded, T.no_ola_sym ? 1 : 0,
                           (const int*)nullptr, 0);
      else if (b->ola_all)  // GPU-parsed batch, some candidate withdrawn on the device: every frame, the emitted ones return at once
        hipLaunchKernelGGL(k_ola_compact, dim3((unsigned)b->nframes, (unsigned)segs), dim3((unsigned)ola_threads), 0, st, s->dev, b->dev,
                           (const float*)work, carry, d_pcm, s->clip, flags + 1, (float*)nullptr /* k_synth wrote the carried tail */,
                           b->last_decoded, T.no_ola_sym ? 1 : 0, (const int*)nullptr, 1);
      else if (b->ola_count == 0)
        b->slot_name[3] = "-";  // paired emission covered every frame, the carried tail included: no launch
      else  // paired emission: only the frames k_synth left over
        hipLaunchKernelGGL(k_ola_compact, dim3((unsigned)b->ola_count, (unsigned)segs), dim3((unsigned)ola_threads), 0, st, s->dev, b->dev,
                           (const float*)work, carry, d_pcm, s->clip, flags + 1, (float*)nullptr /* k_synth wrote the carried tail */,
                           b->last_decoded, T.no_ola_sym ? 1 : 0, b->d_ola_list, 1);
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
  if (timing && !ext_ev) {
    HIP_TRY(hipEventSynchronize(ev[4]));
    for (int k = 0; k < 4; k++) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      kernel_ms[k] += ms;
    }
  }
  return NVH_OK;
}

// Reads and clears the device error / clipped words; maps device errors to status codes.
int collect_flags(nvh_stream* s) {
  int h[2] = {0, 0};
  hipStream_t st = s->ctx->stream;
  HIP_TRY(hipMemcpyAsync(h, s->flags.p, sizeof h, hipMemcpyDeviceToHost, st));
  HIP_TRY(nvh_wait_stream(s->ctx, st));
  if (h[0] || h[1]) HIP_TRY(hipMemsetAsync(s->flags.p, 0, sizeof h, st));
  if (h[1]) s->has_clipped = 1;
  if (h[0]) return NVH_ERR_RUNTIME;  // inverse_dB_table / wMap index out of range in the reference
  return NVH_OK;
}

