// nvh_ops.hip -- level-1 entry points: device-pointer mirrors of the reference's interface methods (IMdct.Reverse,
// IFloor.Apply, IResidue.Decode, IMode.Decode, the window loop, OverlapBuffers, CopyBuffer) for unit parity and for the
// GpuMdct / GpuFloor / GpuResidue / GpuMode classes of the C# shim.
#include "nvh_internal.h"

// ------------------------------------------------------------------------------------------------
// level 1
// ------------------------------------------------------------------------------------------------


extern "C" int nvh_mdct_tables(int n, float* a, float* b, float* c, uint16_t* bitrev) {
  return nvh_guard([&]() -> int {
    if (!valid_block(n) || !a || !b || !c || !bitrev) return NVH_ERR_ARGUMENT;
    nvh::MdctTables t;
    nvh::build_mdct_tables(n, t);
    std::memcpy(a, t.a.data(), t.a.size() * sizeof(float));
    std::memcpy(b, t.b.data(), t.b.size() * sizeof(float));
    std::memcpy(c, t.c.data(), t.c.size() * sizeof(float));
    std::memcpy(bitrev, t.bitrev.data(), t.bitrev.size() * sizeof(uint16_t));
    return NVH_OK;
  });
}

extern "C" int nvh_calc_window(int prev_block, int block, int next_block, float* out) {
  return nvh_guard([&]() -> int {
    if (!out || block <= 0 || prev_block <= 0 || next_block <= 0 || prev_block > block || next_block > block) return NVH_ERR_ARGUMENT;
    nvh::calc_window(prev_block, block, next_block, out);
    return NVH_OK;
  });
}

extern "C" int nvh_calc_overlap(int prev_block, int block, int next_block, int* start, int* valid, int* total) {
  return nvh_guard([&]() -> int {
    if (!start || !valid || !total) return NVH_ERR_ARGUMENT;
    nvh::calc_overlap(prev_block, block, next_block, start, valid, total);
    return NVH_OK;
  });
}

int get_mdct(nvh_ctx* c, int n, MdctDev** out) {
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

// Device memory for callers without a HIP binding of their own (the C# GpuMdct / GpuFloor / GpuResidue / GpuMode classes keep
// the managed float[] contract of the plug-in interfaces and stage through these).  Copies are ordered on the context's
// stream and complete before the call returns.
extern "C" int nvh_dev_alloc(nvh_ctx* c, size_t bytes, void** out) {
  return nvh_guard([&]() -> int {
    if (!c || !out) return NVH_ERR_ARGUMENT;
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return NVH_OK;
  });
}

extern "C" void nvh_dev_free(nvh_ctx* c, void* d_ptr) {
  nvh_guard_void([&] {
    if (!c || !d_ptr) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_ptr);
  });
}

extern "C" int nvh_dev_upload(nvh_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  return nvh_guard([&]() -> int {
    if (!c || (bytes && (!d_dst || !h_src))) return NVH_ERR_ARGUMENT;
    if (!bytes) return NVH_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NVH_OK;
  });
}

extern "C" int nvh_dev_download(nvh_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  return nvh_guard([&]() -> int {
    if (!c || (bytes && (!h_dst || !d_src))) return NVH_ERR_ARGUMENT;
    if (!bytes) return NVH_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NVH_OK;
  });
}

// Measured HBM ceiling for the roofline report: `iters` passes of a float4 copy kernel over `bytes` (a multiple of 16),
// timed with HIP events on the context's stream.  *ms = total time of the `iters` passes.
extern "C" int nvh_measure_copy(nvh_ctx* c, const void* d_src, void* d_dst, size_t bytes, int iters, float* ms) {
  return nvh_guard([&]() -> int {
    if (!c || !d_src || !d_dst || !ms || iters <= 0 || bytes < 16 || (bytes & 15)) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    ScopedEvent e0, e1;
    int rc = e0.create();
    if (rc == NVH_OK) rc = e1.create();
    if (rc != NVH_OK) return rc;
    const long long n4 = (long long)(bytes / 16);
    HIP_TRY(hipEventRecord(e0.e, c->stream));
    for (int i = 0; i < iters; i++)
      hipLaunchKernelGGL(k_copy_f4, dim3(256 * 8 * 2), dim3(256), 0, c->stream, (const float4*)d_src, (float4*)d_dst, n4);
    HIP_TRY(hipEventRecord(e1.e, c->stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(e1.e));
    HIP_TRY(hipEventElapsedTime(ms, e0.e, e1.e));
    return NVH_OK;
  });
}

extern "C" int nvh_inverse_couple(nvh_ctx* c, float* d_magnitude, float* d_angle, int count) {
  return nvh_guard([&]() -> int {
    if (!c || !d_magnitude || !d_angle || count < 0) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    if (count == 0) return NVH_OK;
    int blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_inverse_couple, dim3((unsigned)blocks), dim3(256), 0, c->stream, d_magnitude, d_angle, count);
    HIP_TRY(hipGetLastError());
    return NVH_OK;
  });
}

extern "C" int nvh_mdct_reverse(nvh_ctx* c, int n, int batch, float* d_buf, int64_t stride) {
  return nvh_guard([&]() -> int {
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
  });
}

static unsigned grid_for(long long total) {
  long long blocks = (total + 255) / 256;
  return (unsigned)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

extern "C" int nvh_window_apply(nvh_stream* s, int mode_index, int prev_flag, int next_flag, int batch, float* d_buf,
                                int64_t stride) {
  return nvh_guard([&]() -> int {
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
  });
}

extern "C" int nvh_overlap_buffers(nvh_ctx* c, const float* d_previous, float* d_next, int prev_start, int prev_stop,
                                   int next_start, int channels, int64_t plane_stride) {
  return nvh_guard([&]() -> int {
    if (!c || !d_previous || !d_next || prev_start < 0 || next_start < 0 || channels <= 0) return NVH_ERR_ARGUMENT;
    const int len = prev_stop - prev_start;
    if (len <= 0) return NVH_OK;  // the reference's loop does not run
    if (prev_stop > plane_stride || (int64_t)next_start + len > plane_stride) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_overlap_buffers, dim3(grid_for((long long)channels * len)), dim3(256), 0, c->stream, d_previous, d_next,
                       prev_start, len, next_start, channels, (long long)plane_stride);
    HIP_TRY(hipGetLastError());
    return NVH_OK;
  });
}

extern "C" int nvh_copy_buffer(nvh_ctx* c, const float* d_planes, int start, int count, int channels, int64_t plane_stride,
                               float* d_target, int clip, int* clipped) {
  return nvh_guard([&]() -> int {
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
  });
}

// IFloor.Apply for the stream's floor `floor_index` on `batch` device vectors (see include/nvorbis_hip.h).
extern "C" int nvh_stream_floor_info(const nvh_stream* s, int floor_index, int* type, int* post_count, int* range) {
  return nvh_guard([&]() -> int {
    if (!s || floor_index < 0 || floor_index >= (int)s->setup.floors.size()) return NVH_ERR_ARGUMENT;
    const nvh::Floor& f = s->setup.floors[(size_t)floor_index];
    if (type) *type = f.type;
    if (post_count) *post_count = f.type == 1 ? (int)f.f1.x_list.size() : f.f0.order;
    if (range) *range = f.type == 1 ? f.f1.range : 0;
    return NVH_OK;
  });
}

extern "C" int nvh_floor0_apply(nvh_stream* s, int floor_index, int block_size, int batch, const float* amps, const float* coeffs,
                                int coeff_stride, float* d_residue, int64_t stride, int32_t* status) {
  return nvh_guard([&]() -> int {
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
    // one value per Bark section and item, evaluated here (the reference's float / double expression shapes on the host's
    // libm: bit-exact with the CPU restatement by construction); the kernel gathers by barkMap[i] and multiplies
    const int K = f.f0.bark_map_size, slot = block_size == s->setup.block1 ? 1 : 0, half = block_size / 2;
    std::vector<float> qk((size_t)batch * (size_t)std::max(K, 1), 0.0f);
    std::vector<int32_t> skip((size_t)batch, 0);
    for (int b = 0; b < batch; ++b)
      if (amps[b] > 0.0f && !nvh::floor0_section_values(f.f0, slot, half, amps[b], coeffs + (size_t)b * (size_t)coeff_stride, &qk[(size_t)b * (size_t)K]))
        skip[(size_t)b] = 1;  // wMap index out of range (Floor0.cs:90, :163)
    DevBuf d_amps, d_qk, d_skip;
    d_amps.pool = d_qk.pool = d_skip.pool = &s->ctx->pool;
    int rc;
    if ((rc = d_amps.reserve((size_t)batch * sizeof(float))) != NVH_OK) return rc;
    if ((rc = d_qk.reserve(qk.size() * sizeof(float))) != NVH_OK) return rc;
    if ((rc = d_skip.reserve((size_t)batch * sizeof(int32_t))) != NVH_OK) return rc;
    HIP_TRY(hipMemcpyAsync(d_amps.p, amps, (size_t)batch * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_qk.p, qk.data(), qk.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_skip.p, skip.data(), (size_t)batch * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_floor0_apply, dim3((unsigned)batch), dim3(256), 0, st, s->dev.ipool + s->shared->slab.floor0_bark_off[slot][(size_t)floor_index],
                       (const float*)d_qk.p, K, (const float*)d_amps.p, (const int32_t*)d_skip.p, block_size, d_residue, (long long)stride);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));  // (the staging vectors above are pageable)
    int any = NVH_OK;
    for (int b = 0; b < batch; ++b) {
      const int code = skip[(size_t)b] ? NVH_ERR_RUNTIME : NVH_OK;
      if (status) status[b] = code;
      if (code != NVH_OK && any == NVH_OK) any = code;
    }
    return status ? NVH_OK : any;
  });
}

extern "C" int nvh_stream_codebook_info(const nvh_stream* s, int book_index, int* dimensions, int* entries, int* map_type,
                                        int* prefix_bits, int* max_bits, int* n_prefix, int* n_overflow) {
  return nvh_guard([&]() -> int {
    if (!s || book_index < 0 || book_index >= (int)s->setup.books.size()) return NVH_ERR_ARGUMENT;
    const nvh::Codebook& b = s->setup.books[(size_t)book_index];
    if (dimensions) *dimensions = b.dimensions;
    if (entries) *entries = b.entries;
    if (map_type) *map_type = b.map_type;
    if (prefix_bits) *prefix_bits = b.prefix_bits;
    if (max_bits) *max_bits = b.max_bits;
    if (n_prefix) *n_prefix = b.has_tree ? (int)b.prefix.size() : 0;
    if (n_overflow) *n_overflow = b.has_overflow ? (int)b.overflow.size() : -1;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_codebook_tables(const nvh_stream* s, int book_index, int32_t* lengths, float* lookup, int32_t* prefix,
                                          int32_t* overflow) {
  return nvh_guard([&]() -> int {
    if (!s || book_index < 0 || book_index >= (int)s->setup.books.size()) return NVH_ERR_ARGUMENT;
    const nvh::Codebook& b = s->setup.books[(size_t)book_index];
    if (lengths)
      for (size_t i = 0; i < b.lengths.size(); i++) lengths[i] = b.lengths[i];
    if (lookup && b.map_type != 0)
      for (size_t i = 0; i < b.lookup.size(); i++) lookup[i] = b.lookup[i];
    auto nodes = [](const std::vector<nvh::HuffNode>& v, int32_t* out) {
      for (size_t i = 0; i < v.size(); i++) {
        out[5 * i] = v[i].present ? 1 : 0;
        out[5 * i + 1] = v[i].value;
        out[5 * i + 2] = v[i].length;
        out[5 * i + 3] = v[i].bits;
        out[5 * i + 4] = v[i].mask;
      }
    };
    if (prefix && b.has_tree) nodes(b.prefix, prefix);
    if (overflow && b.has_overflow) nodes(b.overflow, overflow);
    return NVH_OK;
  });
}

extern "C" int nvh_stream_mode_info(const nvh_stream* s, int mode_index, int* block_flag, int* block_size, int* mapping) {
  return nvh_guard([&]() -> int {
    if (!s || mode_index < 0 || mode_index >= (int)s->setup.modes.size()) return NVH_ERR_ARGUMENT;
    const nvh::Mode& m = s->setup.modes[(size_t)mode_index];
    if (block_flag) *block_flag = m.block_flag ? 1 : 0;
    if (block_size) *block_size = m.block_size;
    if (mapping) *mapping = m.mapping;
    return NVH_OK;
  });
}

extern "C" int nvh_floor1_apply(nvh_stream* s, int floor_index, int block_size, int batch, const int32_t* posts,
                                const int32_t* post_counts, float* d_residue, int64_t stride, int32_t* status) {
  return nvh_guard([&]() -> int {
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
  });
}

// IMode.Decode on one packet (Mode.cs:153-170): floors, residue, coupling, floor apply, IMDCT and window of that packet
// alone -- the windowed block, before any overlap -- into d_block [channels][block1] (device memory).  Does not
// touch the stream's decode state; the stream must have nothing pending.  *decoded = 0 when the reference would have
// returned without decoding (short packet).
extern "C" int nvh_mode_decode(nvh_stream* s, const uint8_t* pkt, int len, float* d_block, int* decoded, int* block_size,
                               int* start, int* valid, int* total) {
  return nvh_guard([&]() -> int {
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
    b.blob.pool = b.work.pool = b.carry_in.pool = b.slabs.pool = b.dev_copy.pool = &s->ctx->pool;
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
  });
}

// IResidue.Decode(packet, doNotDecodeChannel, blockSize, buffer) on its own (see include/nvorbis_hip.h): the host reads
// the classifications and entries from the packet, k_residue adds the vectors into the caller's planes.
extern "C" int nvh_residue_decode(nvh_stream* s, int residue_index, const uint8_t* pkt, int len, int bit_offset,
                                  int any_channel_decodes, int block_size, float* d_buffer, int* bits_consumed) {
  return nvh_guard([&]() -> int {
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
    b.blob.pool = b.work.pool = b.carry_in.pool = b.slabs.pool = b.dev_copy.pool = &s->ctx->pool;
    b.h_blob.host = true;
    b.h_blob.pool = &s->ctx->hpool;
    b.descriptors_only = true;  // k_residue below reads the op list
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
  });
}
