// nvh_api.hip -- C ABI of libnvorbis_hip.so (see include/nvorbis_hip.h): contexts, streams (packets in, PCM out),
// resident batches, the Ogg helper.  There is no CPU fallback anywhere behind this ABI: without a HIP device every compute
// entry point fails with NVH_ERR_NO_GPU / NVH_ERR_DEVICE.
#include "nvh_internal.h"

thread_local int g_last_hip_error = 0;

const NvhToggles& nvh_toggles() {
  static const NvhToggles t = [] {
    auto on = [](const char* k) { return std::getenv(k) != nullptr; };
    auto num = [](const char* k) { const char* v = std::getenv(k); return v ? std::atoi(v) : 0; };
    NvhToggles x{};
    x.no_compact = on("NVH_NO_COMPACT");
    x.no_fused_imdct = on("NVH_NO_FUSED_IMDCT");
    x.no_gen8 = on("NVH_NO_GEN8");
    x.unfused = on("NVH_UNFUSED");
    x.no_pair = on("NVH_NO_PAIR");
    x.no_slab = on("NVH_NO_SLAB");
    x.uncached_planes = on("NVH_UNCACHED_PLANES");
    x.poison_planes = on("NVH_POISON_PLANES");
    x.no_ola_sym = on("NVH_NO_OLA_SYM");
    x.no_emit = on("NVH_NO_EMIT");
    x.emit8 = on("NVH_EMIT8");
    x.no_emit8 = on("NVH_NO_EMIT8");
    x.no_prefetch = on("NVH_NO_PREFETCH");
    x.no_walk_two = on("NVH_NO_WALK_TWO");
    x.fpw = std::getenv("NVH_FPW") ? num("NVH_FPW") : 2;
    if (x.fpw != 1 && x.fpw != 2 && x.fpw != 4) x.fpw = 2;
    x.copy_upload = on("NVH_COPY_UPLOAD");
    x.xcd_map = on("NVH_XCD_MAP");
    x.emit_always = on("NVH_EMIT_ALWAYS");
    x.debug_occ = on("NVH_DEBUG_OCC");
    x.gpu_parse_default = on("NVH_GPU_PARSE");
    x.lds_pad = num("NVH_LDS_PAD");
    x.ola_threads = num("NVH_OLA_THREADS");
    x.parse_lanes = num("NVH_PARSE_LANES");
    x.parse_waves = num("NVH_PARSE_WAVES");
    x.no_parse_uni = on("NVH_NO_PARSE_UNI");
    x.no_parse_sub = on("NVH_NO_PARSE_SUB");
    x.parse_cur = std::getenv("NVH_PARSE_CUR") ? num("NVH_PARSE_CUR") : -1;
    if (x.parse_cur > 2) x.parse_cur = 2;
    x.no_sleep_wait = on("NVH_NO_SLEEP_WAIT");
    x.no_parse_sort = on("NVH_NO_PARSE_SORT");
    x.ola_segs = num("NVH_OLA_SEGS");
    x.phase_mask = std::getenv("NVH_DEBUG_SPECTRUM_MASK") ? num("NVH_DEBUG_SPECTRUM_MASK") : 15;
    return x;
  }();
  return t;
}

#ifdef NVH_DEBUG
void* g_dbg_buf = nullptr;
extern "C" void nvh_debug_set_buffer(void* d_buf) { g_dbg_buf = d_buf; }
#endif

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

#ifndef NVH_SRC_HASH
#define NVH_SRC_HASH "unknown"  // nvorbis_amd/build.py passes the hash of the sources; a hand-rolled build has none
#endif
// "nvh-src-hash=<16 hex digits>": nvorbis_amd/build.py finds the marker in the file, native.lib() compares it with the
// hash of the sources it sits next to and refuses (rebuilds) a stale binary.
extern "C" const char* nvh_version(void) { return "nvorbis_hip 0.3 (gfx950) nvh-src-hash=" NVH_SRC_HASH; }
extern "C" int nvh_last_hip_error(void) { return g_last_hip_error; }
extern "C" int nvh_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

extern "C" int nvh_ctx_create(int device, nvh_ctx** out) {
  return nvh_guard([&]() -> int {
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
  });
}

extern "C" void nvh_ctx_destroy(nvh_ctx* c) {
  nvh_guard_void([&] {
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
  });
}

extern "C" int nvh_ctx_set_parse_lanes(nvh_ctx* c, int lanes) {
  return nvh_guard([&]() -> int {
    if (!c || lanes < 0 || lanes > 64 || (lanes & (lanes - 1)) != 0) return NVH_ERR_ARGUMENT;
    c->parse_lanes = lanes;
    return NVH_OK;
  });
}

extern "C" int nvh_ctx_set_hip_stream(nvh_ctx* c, void* hip_stream) {
  return nvh_guard([&]() -> int {
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
  });
}

extern "C" int nvh_ctx_synchronize(nvh_ctx* c) {
  return nvh_guard([&]() -> int {
    if (!c) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NVH_OK;
  });
}

extern "C" int nvh_stream_open(nvh_ctx* c, const uint8_t* id_pkt, int id_len, const uint8_t* comment_pkt,
                               int comment_len, const uint8_t* setup_pkt, int setup_len, nvh_stream** out) {
  return nvh_guard([&]() -> int {
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
    key.assign((const char*)id_pkt, (size_t)id_len);
    key.append((const char*)setup_pkt, (size_t)setup_len);
    // host-only streams (the index pass of a corpus: one per file, on a pool of threads) share their setups per thread: parsing
    // the codebooks costs ~1.2 ms, as much as demultiplexing and indexing a one-minute file
    static thread_local std::map<std::string, std::shared_ptr<SharedSetup>> t_host_setups;
    if (c) {
      auto it = c->setup_cache.find(key);
      if (it != c->setup_cache.end()) sh = it->second;
    } else {
      auto it = t_host_setups.find(key);
      if (it != t_host_setups.end()) sh = it->second;
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
      if (!cached) {  // host-only stream: the codebook directory the host slab writer needs (nvh_stream_pending_slabs)
        std::vector<float> vq;
        std::vector<uint32_t> lattice;
        nvh::build_book_directory(sh->setup, sh->slab, vq, lattice);
        {  // the int pool's layout (nvh_setup.hip): Floor0 Bark maps in floor order, block0 then block1
          uint32_t at = 0;
          for (int w = 0; w < 2; w++) sh->slab.floor0_bark_off[w].assign(sh->setup.floors.size(), 0xFFFFFFFFu);
          for (size_t i = 0; i < sh->setup.floors.size(); i++)
            if (sh->setup.floors[i].type == 0)
              for (int w = 0; w < 2; w++) {
                sh->slab.floor0_bark_off[w][i] = at;
                at += (uint32_t)sh->setup.floors[i].f0.bark_map[w].size();
              }
        }
        nvh::classify_residues(sh->setup, sh->slab, nvh_toggles().no_pair || lattice.size() > 0xFFFFu);
        sh->slab.lattice = lattice;
        sh->slab.vq = vq;
        if (t_host_setups.size() >= 8) t_host_setups.clear();
        t_host_setups.emplace(std::move(key), sh);
      }
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
    if (nvh_toggles().gpu_parse_default && s->shared->gpu_parse_ok) {  // opt-in default for whole test runs
      s->gpu_parse = true;
      s->parser->set_light(true);
    }
    *out = s.release();
    return NVH_OK;
  });
}

extern "C" void nvh_stream_close(nvh_stream* s) {
  nvh_guard_void([&] {
    if (!s) return;
    if (s->ctx) {
      (void)hipSetDevice(s->ctx->device);
      (void)hipStreamSynchronize(s->ctx->stream);
      if (s->copy_stream) (void)hipStreamSynchronize(s->copy_stream);
    }
    delete s;
  });
}

extern "C" int nvh_stream_info(const nvh_stream* s, int* channels, int* sample_rate, int* block0, int* block1) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (channels) *channels = s->setup.channels;
    if (sample_rate) *sample_rate = s->setup.sample_rate;
    if (block0) *block0 = s->setup.block0;
    if (block1) *block1 = s->setup.block1;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_bitrates(const nvh_stream* s, int* upper, int* nominal, int* lower) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (upper) *upper = s->setup.upper_bitrate;
    if (nominal) *nominal = s->setup.nominal_bitrate;
    if (lower) *lower = s->setup.lower_bitrate;
    return NVH_OK;
  });
}

extern "C" int nvh_pinned_alloc(size_t bytes, void** out) {
  return nvh_guard([&]() -> int {
    if (!out) return NVH_ERR_ARGUMENT;
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return NVH_OK;
  });
}

extern "C" void nvh_pinned_free(void* p) {
  nvh_guard_void([&] {
    if (p) (void)hipHostFree(p);
  });
}

extern "C" int nvh_stream_set_gpu_parse(nvh_stream* s, int on) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (!s->ctx) return NVH_ERR_NO_GPU;
    if (!s->pending.frames.empty()) return NVH_ERR_ARGUMENT;  // switch between batches only
    if (on && !s->shared->gpu_parse_ok) return NVH_ERR_UNSUPPORTED;
    s->gpu_parse = on != 0;
    s->parser->set_light(s->gpu_parse);
    return NVH_OK;
  });
}

extern "C" int nvh_stream_set_clip(nvh_stream* s, int on) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    s->clip = on ? 1 : 0;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_has_clipped(nvh_stream* s, int* clipped) {
  return nvh_guard([&]() -> int {
    if (!s || !clipped) return NVH_ERR_ARGUMENT;
    *clipped = s->has_clipped;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_position(const nvh_stream* s, int64_t* position, int64_t* emitted, int* eos) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (position) *position = s->parser->position();
    if (emitted) *emitted = s->parser->emitted();
    if (eos) *eos = s->parser->eos() ? 1 : 0;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_position_state(const nvh_stream* s, int* has_position, int64_t* position) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (has_position) *has_position = s->parser->has_position() ? 1 : 0;
    if (position) *position = s->parser->position();
    return NVH_OK;
  });
}

extern "C" int nvh_stream_set_position_state(nvh_stream* s, int has_position, int64_t position) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    replay_note(s, ReplayLog::kPosition, nullptr, 0, position, has_position != 0);
    s->parser->set_position_state(has_position != 0, position);
    return NVH_OK;
  });
}

// Integer geometry of a run of audio packets as a serial decoder that starts with the first of them sees it (see
// include/nvorbis_hip.h).  A parser of its own in light mode: no bits beyond the packet type, mode number and window
// flags are read, nothing of `s` changes, no GPU is involved.
extern "C" int nvh_stream_index_packets(const nvh_stream* s, const uint8_t* bytes, const int64_t* offsets, const int64_t* granules,
                                        const uint8_t* flags, int n, int64_t* position_after, int64_t* emitted_after,
                                        uint8_t* state_after, int64_t* total_emitted) {
  return nvh_guard([&]() -> int {
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
  });
}

// StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647) -> Mode.GetPacketSampleCount (Mode.cs:172-176): the number of
// samples a packet stands for in the page granule arithmetic of the reference's seek (Ogg/PacketProvider.cs:74-146).
extern "C" int nvh_stream_packet_sample_count(const nvh_stream* s, const uint8_t* pkt, int len, int is_resync, int* count) {
  return nvh_guard([&]() -> int {
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
  });
}

// ResetDecoder (StreamDecoder.cs:295-305): forget the previous block, the position, end of stream and the clipped flag;
// the next packet pushed is a "first packet" again.  Pending frames are dropped.
extern "C" int nvh_stream_reset(nvh_stream* s) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    s->pending.clear();
    s->replay.clear();
    s->replay_error = NVH_OK;
    s->parser.reset(new (std::nothrow) nvh::StreamParser(&s->setup));
    if (!s->parser) return NVH_ERR_NOMEM;
    s->parser->set_light(s->gpu_parse);
    s->has_clipped = 0;
    if (s->ctx && s->copy_stream) {  // outstanding pipelined batches are abandoned: let their copies land first
      HIP_TRY(hipSetDevice(s->ctx->device));
      HIP_TRY(hipStreamSynchronize(s->ctx->stream));
      HIP_TRY(hipStreamSynchronize(s->copy_stream));
      s->flight[0].on = s->flight[1].on = false;
      s->flight_next = s->flight_first = 0;
    }
    if (s->ctx && s->flags.p) {
      HIP_TRY(hipSetDevice(s->ctx->device));
      HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), s->ctx->stream));
    }
    return NVH_OK;
  });
}

extern "C" int nvh_stream_drop_pending(nvh_stream* s) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    s->pending.clear();
    s->replay.clear();
    s->parser->begin_batch();  // later frames refer to the previous block as a carried tail
    return NVH_OK;
  });
}

extern "C" int nvh_stream_push_packet(nvh_stream* s, const uint8_t* data, int len, int64_t granule, int flags) {
  return nvh_guard([&]() -> int {
    if (!s || (!data && len > 0) || len < 0) return NVH_ERR_ARGUMENT;
    static const uint8_t empty = 0;
    replay_note(s, ReplayLog::kPacket, data, len, granule, flags);
    return s->parser->push_packet(data ? data : &empty, len, granule, flags, s->pending);
  });
}

extern "C" int nvh_stream_push_packets(nvh_stream* s, const uint8_t* bytes, const int64_t* offsets, const int64_t* granules,
                                       const uint8_t* flags, int n, int max_packets, int* consumed) {
  return nvh_guard([&]() -> int {
    if (!s || !bytes || !offsets || !consumed || n < 0) return NVH_ERR_ARGUMENT;
    static const uint8_t empty = 0;
    int i = 0;
    // the look-ahead loop of a batched caller: stop when the quota is used up or once the stream has seen its
    // end-of-stream packet (StreamDecoder.cs:343-350: no more packets are pulled after _eosFound)
    for (; i < n && i < max_packets && !s->parser->eos(); i++) {
      const int64_t len = offsets[i + 1] - offsets[i];
      if (len < 0 || len > 0x7FFFFFFF) return NVH_ERR_ARGUMENT;
      replay_note(s, ReplayLog::kPacket, bytes + offsets[i], (int)len, granules ? granules[i] : -1, flags ? (int)flags[i] : 0);
      int rc = s->parser->push_packet(len ? bytes + offsets[i] : &empty, (int)len, granules ? granules[i] : -1,
                                      flags ? (int)flags[i] : 0, s->pending);
      if (rc != NVH_OK) {
        *consumed = i;
        return rc;
      }
    }
    *consumed = i;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_push_end(nvh_stream* s) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    replay_note(s, ReplayLog::kEnd, nullptr, 0, -1, 0);
    return s->parser->push_end(s->pending);
  });
}

extern "C" int nvh_stream_pending_geometry(const nvh_stream* s, int32_t* out, int cap_frames) {
  return nvh_guard([&]() -> int {
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
  });
}

extern "C" int nvh_stream_pending_slabs(const nvh_stream* s, uint8_t* buf, int64_t cap, int64_t* bytes, uint32_t* first_unit,
                                        int cap_frames) {
  return nvh_guard([&]() -> int {
    if (!s || !bytes || cap < 0 || (cap > 0 && !buf)) return NVH_ERR_ARGUMENT;
    if (s->parser->light()) return NVH_ERR_UNSUPPORTED;  // GPU-parse mode: the host never sees floors and residues
    nvh::SlabBatch sb;
    int rc = nvh::build_slabs(s->setup, s->shared->slab, s->pending, sb);
    if (rc != NVH_OK) return rc;
    *bytes = (int64_t)sb.data.size() * 16;
    const int nf = (int)s->pending.frames.size();
    if (*bytes > cap || (first_unit && cap_frames < nf + 1)) return NVH_ERR_ARGUMENT;
    if (*bytes) std::memcpy(buf, sb.data.data(), (size_t)*bytes);
    if (first_unit) std::memcpy(first_unit, sb.first.data(), (size_t)(nf + 1) * sizeof(uint32_t));
    return NVH_OK;
  });
}

extern "C" int nvh_stream_lattice_pool(const nvh_stream* s, uint32_t* out, int64_t cap_words, int64_t* words) {
  return nvh_guard([&]() -> int {
    if (!s || !words || cap_words < 0 || (cap_words > 0 && !out)) return NVH_ERR_ARGUMENT;
    // lattice pool, then the value pool of the digit form: what a slab record's 12-bit offset points into
    const std::vector<uint32_t>& L = s->shared->slab.lattice;
    const std::vector<uint32_t>& V = s->shared->slab.val_pool;
    const size_t lw = V.empty() ? L.size() : std::max<size_t>(L.size(), 1);  // (the device image pads an empty lattice pool to one word)
    *words = (int64_t)(lw + V.size());
    if (*words > cap_words) return NVH_ERR_ARGUMENT;
    if (lw > L.size()) out[0] = 0u;
    if (!L.empty()) std::memcpy(out, L.data(), L.size() * sizeof(uint32_t));
    if (!V.empty()) std::memcpy(out + lw, V.data(), V.size() * sizeof(uint32_t));
    return NVH_OK;
  });
}

extern "C" int nvh_stream_vq_pool(const nvh_stream* s, float* out, int64_t cap_floats, int64_t* floats) {
  return nvh_guard([&]() -> int {
    if (!s || !floats || cap_floats < 0 || (cap_floats > 0 && !out)) return NVH_ERR_ARGUMENT;
    const std::vector<float>& V = s->shared->slab.vq;
    *floats = (int64_t)V.size();
    if (*floats > cap_floats) return NVH_ERR_ARGUMENT;
    if (!V.empty()) std::memcpy(out, V.data(), V.size() * sizeof(float));
    return NVH_OK;
  });
}

extern "C" int nvh_stream_pending(const nvh_stream* s, int* frames, int64_t* pcm_samples) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (frames) *frames = (int)s->pending.frames.size();
    if (pcm_samples) *pcm_samples = s->pending.pcm_samples;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_synth(nvh_stream* s, float* pcm_host, float* d_pcm, int64_t capacity, int64_t* written) {
  return nvh_guard([&]() -> int {
    if (!s || (pcm_host && d_pcm)) return NVH_ERR_ARGUMENT;
    if (written) *written = 0;
    if (!s->ctx) return NVH_ERR_NO_GPU;
    if (s->flight[0].on || s->flight[1].on) return NVH_ERR_ARGUMENT;  // pipelined batches outstanding: end them first
    HIP_TRY(hipSetDevice(s->ctx->device));
    const int ch = s->setup.channels;
    int64_t need = s->pending.pcm_samples * ch;
    if (s->pending.frames.empty()) return NVH_OK;
    if (need > 0 && !pcm_host && !d_pcm) return NVH_ERR_ARGUMENT;
    if (capacity < need) return NVH_ERR_ARGUMENT;
    nvh_batch* b = &s->scratch;
    s->replay_error = NVH_OK;
    s->replay_errors.clear();
    int rc = batch_upload(s, b);
    if (rc != NVH_OK) return rc;
    // GPU-parse mode: a batch with a throwing packet was parsed again on the host (nvh_launch.hip: replay_on_host); the
    // throwing packet contributes nothing, so the batch may emit less than the look-ahead said
    need = b->pcm_samples * ch;
    if (capacity < need) return NVH_ERR_ARGUMENT;
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
    HIP_TRY(nvh_wait_stream(s->ctx, st));
    if (bounce) std::memcpy(pcm_host, s->h_pcm.p, bounce);
    if (h_flags[0] || h_flags[1]) HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), st));
    if (h_flags[1]) s->has_clipped = 1;
    if (h_flags[0]) return NVH_ERR_RUNTIME;  // inverse_dB_table / wMap index out of range in the reference
    if (written) *written = need;
    // a packet of this batch made the parser fail (the code nvh_stream_push_packet returns in host-parse mode): the PCM of
    // every other packet is complete and *written says so; nvh_stream_parse_errors tells where the exceptions belong
    return s->replay_error;
  });
}

// Pipelined form of nvh_stream_synth for a destination in page-locked host memory: begin queues upload, (GPU parse,) synthesis and
// -- on a copy stream of its own -- the transfer of the PCM, and returns; end waits for the OLDEST outstanding batch.  The
// transfer of batch i (8 KB per stereo long frame over PCIe: the longest step of the end-to-end path) then runs while the host
// pushes batch i+1 and the GPU parses and synthesises it.
extern "C" int nvh_stream_synth_begin(nvh_stream* s, float* pcm_host, int64_t capacity, int64_t* expected) {
  return nvh_guard([&]() -> int {
    if (!s || !pcm_host) return NVH_ERR_ARGUMENT;
    if (expected) *expected = 0;
    if (!s->ctx) return NVH_ERR_NO_GPU;
    HIP_TRY(hipSetDevice(s->ctx->device));
    const int slot = s->flight_next;
    nvh_stream::Flight& F = s->flight[slot];
    if (F.on) return NVH_ERR_ARGUMENT;  // two batches outstanding already
    {
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, pcm_host) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return NVH_ERR_ARGUMENT;  // the copy engine needs page-locked memory (nvh_pinned_alloc)
      }
    }
    const int ch = s->setup.channels;
    hipStream_t st = s->ctx->stream;
    if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    if (!F.kernels) HIP_TRY(hipEventCreateWithFlags(&F.kernels, hipEventDisableTiming));
    if (!F.done) HIP_TRY(hipEventCreateWithFlags(&F.done, hipEventDisableTiming));
    // the staging image of the previous batch's descriptors is about to be overwritten: its upload (and kernels) must be through
    HIP_TRY(hipStreamSynchronize(st));
    F.need = 0;
    F.replay_error = NVH_OK;
    F.replay_errors.clear();
    if (s->pending.frames.empty()) {  // nothing to do: an outstanding "batch" of zero samples keeps begin / end paired
      HIP_TRY(hipEventRecord(F.done, st));
      F.on = true;
      s->flight_next ^= 1;
      return NVH_OK;
    }
    if (capacity < s->pending.pcm_samples * ch) return NVH_ERR_ARGUMENT;
    nvh_batch* b = &s->scratch;
    s->replay_error = NVH_OK;
    s->replay_errors.clear();
    int rc = batch_upload(s, b);
    if (rc != NVH_OK) return rc;
    const int64_t need = b->pcm_samples * ch;
    if (capacity < need) return NVH_ERR_ARGUMENT;
    if ((rc = s->pcm2[slot].reserve((size_t)(need > 0 ? need : 1) * sizeof(float))) != NVH_OK) return rc;
    if ((rc = s->h_flags2.reserve(4 * sizeof(int))) != NVH_OK) return rc;
    float* dst = (float*)s->pcm2[slot].p;
    rc = batch_launch(b, (const float*)s->carry[s->carry_cur].p, (float*)s->carry[s->carry_cur ^ 1].p, dst, false, nullptr);
    if (rc != NVH_OK) return rc;
    if (b->last_decoded >= 0) s->carry_cur ^= 1;
    // this batch's flag words, then a clean pair for the next one (all on the launch stream, in order)
    HIP_TRY(hipMemcpyAsync((int*)s->h_flags2.p + 2 * slot, s->flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemsetAsync(s->flags.p, 0, 2 * sizeof(int), st));
    HIP_TRY(hipEventRecord(F.kernels, st));
    HIP_TRY(hipStreamWaitEvent(s->copy_stream, F.kernels, 0));
    if (need > 0) HIP_TRY(hipMemcpyAsync(pcm_host, dst, (size_t)need * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
    HIP_TRY(hipEventRecord(F.done, s->copy_stream));
    F.need = need;
    F.replay_error = s->replay_error;
    F.replay_errors = s->replay_errors;
    F.on = true;
    s->flight_next ^= 1;
    if (expected) *expected = need;
    return NVH_OK;
  });
}

extern "C" int nvh_stream_synth_end(nvh_stream* s, int64_t* written) {
  return nvh_guard([&]() -> int {
    if (!s) return NVH_ERR_ARGUMENT;
    if (written) *written = 0;
    if (!s->ctx) return NVH_ERR_NO_GPU;
    nvh_stream::Flight& F = s->flight[s->flight_first];
    if (!F.on) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipEventSynchronize(F.done));
    F.on = false;
    const int slot = s->flight_first;
    s->flight_first ^= 1;
    s->replay_error = F.replay_error;  // nvh_stream_parse_errors then describes THIS batch
    s->replay_errors = F.replay_errors;
    if (F.need > 0) {
      const int* h = (const int*)s->h_flags2.p + 2 * slot;
      if (h[1]) s->has_clipped = 1;
      if (h[0]) return NVH_ERR_RUNTIME;
    }
    if (written) *written = F.need;
    return F.replay_error;
  });
}

extern "C" int nvh_stream_parse_errors(const nvh_stream* s, int32_t* codes, int64_t* samples_before, int cap, int* count) {
  return nvh_guard([&]() -> int {
    if (!s || !count || cap < 0 || (cap > 0 && (!codes || !samples_before))) return NVH_ERR_ARGUMENT;
    const int n = s->replay_error != NVH_OK ? (int)s->replay_errors.size() : 0;
    *count = n;
    for (int i = 0; i < n && i < cap; i++) {
      codes[i] = s->replay_errors[(size_t)i].first;
      samples_before[i] = s->replay_errors[(size_t)i].second;
    }
    return NVH_OK;
  });
}

extern "C" int nvh_batch_upload(nvh_stream* s, nvh_batch** out) {
  return nvh_guard([&]() -> int {
    if (!s || !out) return NVH_ERR_ARGUMENT;
    *out = nullptr;
    if (!s->ctx) return NVH_ERR_NO_GPU;
    HIP_TRY(hipSetDevice(s->ctx->device));
    std::unique_ptr<nvh_batch> b(new (std::nothrow) nvh_batch());
    if (b && s && s->ctx) {
      b->blob.pool = b->work.pool = b->carry_in.pool = b->slabs.pool = b->dev_copy.pool = b->slab3.pool = &s->ctx->pool;
      b->work.uncached = nvh_toggles().uncached_planes;
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
    s->replay_error = NVH_OK;
    rc = batch_upload(s, b.get());
    if (rc != NVH_OK) return rc;
    if (s->replay_error != NVH_OK) return s->replay_error;  // resident batches are all-or-nothing
    *out = b.release();
    return NVH_OK;
  });
}

extern "C" int nvh_batch_info(const nvh_batch* b, int* frames, int* chan_frames, int64_t* pcm_samples,
                              int64_t* descriptor_bytes) {
  return nvh_guard([&]() -> int {
    if (!b) return NVH_ERR_ARGUMENT;
    if (frames) *frames = b->nframes;
    if (chan_frames) *chan_frames = b->chan_frames;
    if (pcm_samples) *pcm_samples = b->pcm_samples;
    if (descriptor_bytes) *descriptor_bytes = b->descriptor_bytes;
    return NVH_OK;
  });
}

extern "C" int nvh_batch_stats(const nvh_batch* b, int64_t* out8) {
  return nvh_guard([&]() -> int {
    if (!b || !out8) return NVH_ERR_ARGUMENT;
    for (int i = 0; i < 8; i++) out8[i] = b->stats[i];
    return NVH_OK;
  });
}

extern "C" int nvh_batch_kernels(const nvh_batch* b, char* buf, int cap) {
  return nvh_guard([&]() -> int {
    if (!b || !buf || cap <= 0) return NVH_ERR_ARGUMENT;
    std::string t;
    for (int k = 0; k < 4; k++) {
      if (k) t += ",";
      t += b->slot_name[k];
    }
    if ((int)t.size() + 1 > cap) return NVH_ERR_ARGUMENT;
    std::memcpy(buf, t.c_str(), t.size() + 1);
    return NVH_OK;
  });
}

extern "C" int nvh_stream_kernels(const nvh_stream* s, char* buf, int cap) {
  return nvh_batch_kernels(s ? &s->scratch : nullptr, buf, cap);
}

extern "C" int nvh_batch_synth(nvh_batch* b, float* d_pcm, int64_t capacity) {
  return nvh_guard([&]() -> int {
    if (!b || !b->s) return NVH_ERR_ARGUMENT;
    nvh_stream* s = b->s;
    if (capacity < b->pcm_samples * s->setup.channels) return NVH_ERR_ARGUMENT;
    if (b->pcm_samples > 0 && !d_pcm) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(s->ctx->device));
    // the stream keeps the tail of the newest batch (written to its current carry buffer; the batch reads its own snapshot)
    return batch_launch(b, (const float*)b->carry_in.p, (float*)s->carry[s->carry_cur].p, d_pcm, false, nullptr);
  });
}

extern "C" int nvh_batch_time(nvh_batch* b, float* d_pcm, int64_t capacity, int iters, float* total_ms,
                              float* kernel_ms) {
  return nvh_guard([&]() -> int {
    if (!b || !b->s || iters <= 0) return NVH_ERR_ARGUMENT;
    nvh_stream* s = b->s;
    if (capacity < b->pcm_samples * s->setup.channels) return NVH_ERR_ARGUMENT;
    HIP_TRY(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    float km[4] = {0, 0, 0, 0};
    if (kernel_ms) {
      // all iterations queued back to back, five events each, read afterwards (batch_launch: ext_ev)
      std::vector<ScopedEvent> evs((size_t)iters * 5);
      std::vector<hipEvent_t> raw((size_t)iters * 5);
      for (size_t k = 0; k < evs.size(); k++) {
        int rc = evs[k].create();
        if (rc != NVH_OK) return rc;
        raw[k] = evs[k].e;
      }
      for (int i = 0; i < iters; i++) {
        int rc = batch_launch(b, (const float*)b->carry_in.p, (float*)s->carry[s->carry_cur].p, d_pcm, true, km, &raw[(size_t)i * 5]);
        if (rc != NVH_OK) return rc;
      }
      HIP_TRY(hipEventSynchronize(raw.back()));
      for (int i = 0; i < iters; i++)
        for (int k = 0; k < 4; k++) {
          float ms = 0;
          HIP_TRY(hipEventElapsedTime(&ms, raw[(size_t)i * 5 + k], raw[(size_t)i * 5 + k + 1]));
          km[k] += ms;
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
  });
}

extern "C" void nvh_batch_free(nvh_batch* b) {
  nvh_guard_void([&] {
    if (!b) return;
    if (b->s) {
      (void)hipSetDevice(b->s->ctx->device);
      (void)hipStreamSynchronize(b->s->ctx->stream);
    }
    delete b;
  });
}

// ------------------------------------------------------------------------------------------------
// container helper
// ------------------------------------------------------------------------------------------------

// The two-call protocol (sizing call, then the call that fills the caller's arrays) would demultiplex a file twice: the
// sizing call parks its result here, per thread, and the fill call that follows on the same bytes takes it.  "The same bytes"
// is checked by address, length, stream and a hash of the whole buffer, so a caller that reuses or edits a buffer is not handed
// the old packets.
namespace {
struct DemuxMemo {
  const uint8_t* bytes = nullptr;
  size_t len = 0;
  int stream = -1, nstreams = 0;
  bool forward = false;
  uint64_t print = 0;
  nvh::OggPackets pk;
};
thread_local DemuxMemo g_demux_memo;

uint64_t demux_fingerprint(const uint8_t* bytes, size_t len) {
  // every byte of the buffer (four independent multiply chains; well under the cost of the demultiplex, which checksums every
  // page): a caller that edits a buffer in place between the sizing call and the fill call is not handed the old packets
  uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ (uint64_t)len, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  size_t off = 0;
  for (; off + 32 <= len; off += 32) {
    uint64_t w[4];
    std::memcpy(w, bytes + off, 32);
    for (int k = 0; k < 4; k++) h[k] = (h[k] ^ w[k]) * 0x100000001B3ull;
  }
  for (; off < len; off += 8) {
    uint64_t w = 0;
    std::memcpy(&w, bytes + off, len - off >= 8 ? 8 : len - off);
    h[0] = (h[0] ^ w) * 0x100000001B3ull;
  }
  uint64_t r = h[0];
  for (int k = 1; k < 4; k++) r = (r ^ (h[k] >> 29) ^ h[k]) * 0x9E3779B97F4A7C15ull;
  return r ^ (r >> 31);
}

template <typename Demux>
int demux_two_call(bool forward, Demux demux, const uint8_t* bytes, size_t len, int stream_index, uint8_t* pkt_bytes,
                   int64_t pkt_bytes_cap, int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                   int64_t* total_bytes, int* nstreams) {
  if (!bytes || !npackets || !total_bytes || stream_index < 0) return NVH_ERR_ARGUMENT;
  DemuxMemo& M = g_demux_memo;
  const bool sizing = !pkt_bytes && !offsets && !granules && !flags;
  // (the fingerprint is a pass over the file: only where it decides something -- a sizing call, or a fill call that could be
  // the second half of one; a one-call caller with buffers of its own pays for the demultiplex alone)
  const bool memo_candidate = !sizing && M.bytes == bytes && M.len == len && M.stream == stream_index && M.forward == forward;
  const uint64_t print = (sizing || memo_candidate) ? demux_fingerprint(bytes, len) : 0;
  nvh::OggPackets local;
  nvh::OggPackets* pk = &local;
  int ns = 0;
  if (memo_candidate && M.print == print) {
    pk = &M.pk;  // the sizing call's result
    ns = M.nstreams;
  } else {
    if (sizing) pk = &M.pk;
    M.bytes = nullptr;
    int rc = demux(bytes, len, *pk, stream_index, &ns);
    if (rc != NVH_OK) return rc;
    if (sizing) {
      M.bytes = bytes; M.len = len; M.stream = stream_index; M.forward = forward; M.print = print; M.nstreams = ns;
    }
  }
  if (nstreams) *nstreams = ns;
  const int n = (int)pk->granule.size();
  *npackets = n;
  *total_bytes = (int64_t)pk->bytes.size();
  if (sizing) return NVH_OK;
  int rc = NVH_OK;
  if (pkt_cap < n || pkt_bytes_cap < (int64_t)pk->bytes.size() || !pkt_bytes || !offsets || !granules || !flags) {
    rc = NVH_ERR_ARGUMENT;
  } else {
    if (!pk->bytes.empty()) std::memcpy(pkt_bytes, pk->bytes.data(), pk->bytes.size());
    std::memcpy(offsets, pk->offs.data(), sizeof(int64_t) * (size_t)(n + 1));
    if (n) {
      std::memcpy(granules, pk->granule.data(), sizeof(int64_t) * (size_t)n);
      std::memcpy(flags, pk->flags.data(), (size_t)n);
    }
  }
  if (pk == &M.pk) {  // taken: release the memory (a thread that decodes one big file does not keep a copy of it)
    M.bytes = nullptr;
    nvh::OggPackets().bytes.swap(M.pk.bytes);
  }
  return rc;
}
}  // namespace

extern "C" int nvh_ogg_demux_stream(const uint8_t* bytes, size_t len, int stream_index, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                                    int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                                    int64_t* total_bytes, int* nstreams) {
  return nvh_guard([&]() -> int {
    return demux_two_call(false, [](const uint8_t* b, size_t l, nvh::OggPackets& o, int si, int* ns) { return nvh::ogg_demux(b, l, o, si, ns); },
                          bytes, len, stream_index, pkt_bytes, pkt_bytes_cap, offsets, granules, flags, pkt_cap, npackets, total_bytes, nstreams);
  });
}

// The index form (host_ogg.h: OggIndexMode): no checksums, packet heads only -- see include/nvorbis_hip.h
extern "C" int nvh_ogg_index_packets(const uint8_t* bytes, size_t len, int stream_index, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                                     int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                                     int64_t* total_bytes, int64_t* payload_bytes, int* nstreams) {
  return nvh_guard([&]() -> int {
    if (!bytes || !npackets || !total_bytes || stream_index < 0 || !pkt_bytes || !offsets || !granules || !flags) return NVH_ERR_ARGUMENT;
    nvh::OggPackets pk;
    nvh::OggIndexMode mode;
    int ns = 0;
    int rc = nvh::ogg_demux(bytes, len, pk, stream_index, &ns, false, &mode);
    if (rc != NVH_OK) return rc;
    const int n = (int)pk.granule.size();
    *npackets = n;
    *total_bytes = (int64_t)pk.bytes.size();
    if (payload_bytes) *payload_bytes = mode.payload_bytes;
    if (nstreams) *nstreams = ns;
    if (pkt_cap < n || pkt_bytes_cap < (int64_t)pk.bytes.size()) return NVH_ERR_ARGUMENT;
    if (!pk.bytes.empty()) std::memcpy(pkt_bytes, pk.bytes.data(), pk.bytes.size());
    std::memcpy(offsets, pk.offs.data(), sizeof(int64_t) * (size_t)(n + 1));
    if (n) {
      std::memcpy(granules, pk.granule.data(), sizeof(int64_t) * (size_t)n);
      std::memcpy(flags, pk.flags.data(), (size_t)n);
    }
    return NVH_OK;
  });
}

// ForwardOnlyPageReader / ForwardOnlyPacketProvider (host_ogg.cpp): the packet list a non-seekable source yields
extern "C" int nvh_ogg_demux_forward(const uint8_t* bytes, size_t len, int stream_index, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                                     int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                                     int64_t* total_bytes, int* nstreams) {
  return nvh_guard([&]() -> int {
    return demux_two_call(true, [](const uint8_t* b, size_t l, nvh::OggPackets& o, int si, int* ns) { return nvh::ogg_demux_forward(b, l, o, si, ns); },
                          bytes, len, stream_index, pkt_bytes, pkt_bytes_cap, offsets, granules, flags, pkt_cap, npackets, total_bytes, nstreams);
  });
}

extern "C" int nvh_ogg_demux(const uint8_t* bytes, size_t len, uint8_t* pkt_bytes, int64_t pkt_bytes_cap,
                             int64_t* offsets, int64_t* granules, uint8_t* flags, int pkt_cap, int* npackets,
                             int64_t* total_bytes) {
  return nvh_guard([&]() -> int {
    return nvh_ogg_demux_stream(bytes, len, 0, pkt_bytes, pkt_bytes_cap, offsets, granules, flags, pkt_cap, npackets, total_bytes, nullptr);
  });
}

// ------------------------------------------------------------------------------------------------
// seeking: page table + PacketProvider.SeekTo (host_ogg.cpp)
// ------------------------------------------------------------------------------------------------
struct nvh_ogg_index {
  nvh::OggPackets pk;
};

extern "C" int nvh_ogg_index_open(const uint8_t* bytes, size_t len, int stream_index, nvh_ogg_index** out) {
  return nvh_guard([&]() -> int {
    if (!bytes || !out || stream_index < 0) return NVH_ERR_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<nvh_ogg_index> ix(new (std::nothrow) nvh_ogg_index());
    if (!ix) return NVH_ERR_NOMEM;
    int rc = nvh::ogg_demux(bytes, len, ix->pk, stream_index, nullptr, true);
    if (rc != NVH_OK) return rc;
    ix->pk.bytes.clear();  // the seek search works on the page table alone
    ix->pk.bytes.shrink_to_fit();
    *out = ix.release();
    return NVH_OK;
  });
}

extern "C" void nvh_ogg_index_close(nvh_ogg_index* ix) {
  nvh_guard_void([&]() { delete ix; });
}

extern "C" int nvh_ogg_index_info(const nvh_ogg_index* ix, int* npages, int* npackets, int* first_data_page, int64_t* max_granule,
                                  int* has_all_pages) {
  return nvh_guard([&]() -> int {
    if (!ix) return NVH_ERR_ARGUMENT;
    if (npages) *npages = (int)ix->pk.pages.size();
    if (npackets) *npackets = (int)ix->pk.granule.size();
    if (first_data_page) *first_data_page = ix->pk.first_data_page;
    if (max_granule) *max_granule = ix->pk.max_granule;
    if (has_all_pages) *has_all_pages = ix->pk.has_all_pages ? 1 : 0;
    return NVH_OK;
  });
}

extern "C" int nvh_ogg_index_page(const nvh_ogg_index* ix, int page, int64_t* granule, int* flags, int* packet_count, int* first_packet) {
  return nvh_guard([&]() -> int {
    if (!ix || page < 0 || page >= (int)ix->pk.pages.size()) return NVH_ERR_ARGUMENT;
    const nvh::OggPageInfo& pg = ix->pk.pages[(size_t)page];
    if (granule) *granule = pg.granule;
    if (flags) *flags = (pg.resync ? 1 : 0) | (pg.continuation ? 2 : 0) | (pg.continued ? 4 : 0);
    if (packet_count) *packet_count = pg.packet_count;
    if (first_packet) {
      *first_packet = -1;
      for (int32_t f : pg.flat)
        if (f >= 0) {
          *first_packet = f;
          break;
        }
    }
    return NVH_OK;
  });
}

extern "C" int nvh_ogg_seek(const nvh_ogg_index* ix, const nvh_stream* s, int64_t granule_pos, int pre_roll, int64_t* packet_index,
                            int64_t* granule_out) {
  return nvh_guard([&]() -> int {
    if (!ix || !s || !packet_index || !granule_out) return NVH_ERR_ARGUMENT;
    auto count = [](void* user, const uint8_t* head, int len, bool is_resync) -> int {
      int n = 0;
      (void)nvh_stream_packet_sample_count((const nvh_stream*)user, head, len, is_resync ? 1 : 0, &n);
      return n;
    };
    return nvh::ogg_seek(ix->pk, count, (void*)s, granule_pos, pre_roll, packet_index, granule_out);
  });
}
