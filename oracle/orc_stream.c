/* oracle/orc_stream.c -- StreamDecoder.cs + Factory.cs + VorbisReader.ReadSamples restatement
 * (test infrastructure, see orc.h). */
#include "orc_internal.h"

static int validate_header(orc_packet *p, const uint8_t *expected, int n) { /* StreamDecoder.cs:145-155 */
  int i;
  for (i = 0; i < n; i++)
    if (expected[i] != orc_read_bits(p, 8)) return 0;
  return 1;
}

/* StreamDecoder.cs:179-204 */
static int load_stream_header(orc_decoder *d, orc_packet *p) {
  static const uint8_t sig[11] = {0x01, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73, 0x00, 0x00, 0x00, 0x00};
  if (!validate_header(p, sig, 11)) return ORC_ERR_NOT_VORBIS;
  d->channels = (int)(uint8_t)orc_read_bits(p, 8);
  d->sample_rate = (int)orc_read_bits(p, 32);
  (void)orc_read_bits(p, 32); /* UpperBitrate   */
  (void)orc_read_bits(p, 32); /* NominalBitrate */
  (void)orc_read_bits(p, 32); /* LowerBitrate   */
  d->block0 = 1 << (int)orc_read_bits(p, 4);
  d->block1 = 1 << (int)orc_read_bits(p, 4);
  return ORC_OK;
}

/* StreamDecoder.cs:206-224: only the signature matters to this path */
static int load_comments(orc_packet *p) {
  static const uint8_t sig[7] = {0x03, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73};
  return validate_header(p, sig, 7) ? ORC_OK : ORC_ERR_NOT_VORBIS;
}

/* StreamDecoder.cs:226-289 (+ Factory.cs:22-58 type switches) */
static int load_books(orc_decoder *d, orc_packet *p) {
  static const uint8_t sig[7] = {0x05, 0x76, 0x6f, 0x72, 0x62, 0x69, 0x73};
  int i, rc, times;
  if (!validate_header(p, sig, 7)) return ORC_ERR_NOT_VORBIS;

  d->nbooks = (int)orc_read_bits(p, 8) + 1;
  d->books = (orc_codebook *)calloc((size_t)d->nbooks, sizeof *d->books);
  if (!d->books) return ORC_ERR_NOMEM;
  for (i = 0; i < d->nbooks; i++) {
    rc = orc_codebook_init(&d->books[i], p);
    if (rc) return rc;
  }

  times = (int)orc_read_bits(p, 6) + 1;
  orc_skip_bits(p, 16 * times);

  d->nfloors = (int)orc_read_bits(p, 6) + 1;
  d->floors = (orc_floor *)calloc((size_t)d->nfloors, sizeof *d->floors);
  if (!d->floors) return ORC_ERR_NOMEM;
  for (i = 0; i < d->nfloors; i++) {
    int type = (int)orc_read_bits(p, 16); /* Factory.cs:22-31 */
    if (type != 0 && type != 1) return ORC_ERR_INVALID_DATA;
    rc = orc_floor_init(&d->floors[i], type, p, d->channels, d->block0, d->block1, d->books, d->nbooks);
    if (rc) return rc;
  }

  d->nresidues = (int)orc_read_bits(p, 6) + 1;
  d->residues = (orc_residue *)calloc((size_t)d->nresidues, sizeof *d->residues);
  if (!d->residues) return ORC_ERR_NOMEM;
  for (i = 0; i < d->nresidues; i++) {
    int type = (int)orc_read_bits(p, 16); /* Factory.cs:48-58 */
    if (type < 0 || type > 2) return ORC_ERR_INVALID_DATA;
    rc = orc_residue_init(&d->residues[i], type, p, d->channels, d->books, d->nbooks);
    if (rc) return rc;
  }

  d->nmappings = (int)orc_read_bits(p, 6) + 1;
  d->mappings = (orc_mapping *)calloc((size_t)d->nmappings, sizeof *d->mappings);
  if (!d->mappings) return ORC_ERR_NOMEM;
  for (i = 0; i < d->nmappings; i++) {
    if (orc_read_bits(p, 16) != 0) return ORC_ERR_INVALID_DATA; /* Factory.cs:33-41 */
    rc = orc_mapping_init(&d->mappings[i], p, d->channels, d->nfloors, d->nresidues);
    if (rc) return rc;
  }

  d->nmodes = (int)orc_read_bits(p, 6) + 1;
  d->modes = (orc_mode *)calloc((size_t)d->nmodes, sizeof *d->modes);
  if (!d->modes) return ORC_ERR_NOMEM;
  for (i = 0; i < d->nmodes; i++) {
    rc = orc_mode_init(&d->modes[i], p, d->block0, d->block1, d->nmappings);
    if (rc) return rc;
  }

  if (!orc_read_bit(p)) return ORC_ERR_INVALID_DATA; /* :281 */
  d->mode_field_bits = orc_ilog(d->nmodes - 1);      /* :284 */
  return ORC_OK;
}

static float **alloc_planes(int ch, int n) {
  float **pl = (float **)calloc((size_t)ch, sizeof(float *));
  int i;
  if (!pl) return NULL;
  for (i = 0; i < ch; i++) {
    pl[i] = (float *)calloc((size_t)n, sizeof(float));
    if (!pl[i]) return NULL;
  }
  return pl;
}

static void free_planes(float **pl, int ch) {
  int i;
  if (!pl) return;
  for (i = 0; i < ch; i++) free(pl[i]);
  free(pl);
}

static int get_packet(orc_decoder *d, int idx, orc_packet *p) {
  if (idx >= d->npackets) return 0;
  orc_packet_init(p, d->bytes + d->offs[idx], (int)(d->offs[idx + 1] - d->offs[idx]));
  p->has_granule = d->granule[idx] >= 0;
  p->granule = d->granule[idx];
  p->is_eos = (d->flags[idx] & 1) != 0;
  p->is_resync = (d->flags[idx] & 2) != 0;
  return 1;
}

orc_decoder *orc_open_packets(const uint8_t *bytes, const int64_t *offs, const int64_t *granule, const uint8_t *flags,
                              int npackets, int *err) {
  orc_decoder *d = (orc_decoder *)calloc(1, sizeof *d);
  orc_packet p;
  int rc = ORC_OK;
  size_t total;
  if (err) *err = ORC_OK;
  if (!d) {
    if (err) *err = ORC_ERR_NOMEM;
    return NULL;
  }
  if (npackets < 3) {
    rc = ORC_ERR_NOT_VORBIS;
    goto fail;
  }
  total = (size_t)offs[npackets];
  d->bytes = (uint8_t *)malloc(total ? total : 1);
  d->offs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(npackets + 1));
  d->granule = (int64_t *)malloc(sizeof(int64_t) * (size_t)npackets);
  d->flags = (uint8_t *)malloc((size_t)npackets);
  if (!d->bytes || !d->offs || !d->granule || !d->flags) {
    rc = ORC_ERR_NOMEM;
    goto fail;
  }
  memcpy(d->bytes, bytes, total);
  memcpy(d->offs, offs, sizeof(int64_t) * (size_t)(npackets + 1));
  memcpy(d->granule, granule, sizeof(int64_t) * (size_t)npackets);
  memcpy(d->flags, flags, (size_t)npackets);
  d->npackets = npackets;
  d->clip_samples = 1; /* StreamDecoder.cs:58 */

  /* ProcessHeaderPackets (StreamDecoder.cs:107-127) */
  get_packet(d, 0, &p);
  if ((rc = load_stream_header(d, &p)) != ORC_OK) goto fail;
  get_packet(d, 1, &p);
  if ((rc = load_comments(&p)) != ORC_OK) goto fail;
  get_packet(d, 2, &p);
  if ((rc = load_books(d, &p)) != ORC_OK) goto fail;
  d->next_packet = 3;
  d->current_position = 0;
  /* ResetDecoder (:295-305) -- all zero already */
  return d;
fail:
  if (err) *err = rc;
  orc_close(d);
  return NULL;
}

orc_decoder *orc_open_ogg(const uint8_t *bytes, size_t len, int *err) {
  uint8_t *pb = NULL, *fl = NULL;
  int64_t *offs = NULL, *gr = NULL;
  int n = 0;
  orc_decoder *d;
  int rc = orc_ogg_demux(bytes, len, &pb, &offs, &gr, &fl, &n);
  if (rc) {
    if (err) *err = rc;
    return NULL;
  }
  d = orc_open_packets(pb, offs, gr, fl, n, err);
  free(pb);
  free(offs);
  free(gr);
  free(fl);
  if (d) {
    d->ogg_bytes = (uint8_t *)malloc(len ? len : 1);
    if (d->ogg_bytes) {
      memcpy(d->ogg_bytes, bytes, len);
      d->ogg_len = len;
      (void)orc_ogg_max_granule(bytes, len, &d->max_granule);
    }
  }
  return d;
}

void orc_close(orc_decoder *d) {
  int i;
  if (!d) return;
  for (i = 0; i < d->nbooks && d->books; i++) orc_codebook_free(&d->books[i]);
  for (i = 0; i < d->nfloors && d->floors; i++) orc_floor_free(&d->floors[i]);
  for (i = 0; i < d->nresidues && d->residues; i++) orc_residue_free(&d->residues[i]);
  for (i = 0; i < d->nmodes && d->modes; i++) orc_mode_free(&d->modes[i]);
  free(d->books);
  free(d->floors);
  free(d->residues);
  free(d->mappings);
  free(d->modes);
  free_planes(d->buf_a, d->channels);
  free_planes(d->buf_b, d->channels);
  free(d->ogg_bytes);
  free(d->bytes);
  free(d->offs);
  free(d->granule);
  free(d->flags);
  free(d->trace);
  free(d);
}

int orc_channels(const orc_decoder *d) { return d->channels; }
int orc_sample_rate(const orc_decoder *d) { return d->sample_rate; }
int orc_block0(const orc_decoder *d) { return d->block0; }
int orc_block1(const orc_decoder *d) { return d->block1; }
int orc_packet_count(const orc_decoder *d) { return d->npackets; }
void orc_set_clip_samples(orc_decoder *d, int on) { d->clip_samples = on; }
int orc_has_clipped(const orc_decoder *d) { return d->has_clipped; }
int orc_is_end_of_stream(const orc_decoder *d) { return d->eos_found && d->prev_buf == NULL; } /* :733 */
int64_t orc_sample_position(const orc_decoder *d) { return d->current_position; }
int orc_last_error(const orc_decoder *d) { return d->last_error; }
void orc_enable_trace(orc_decoder *d, int on) { d->trace_on = on; }
int orc_trace_count(const orc_decoder *d) { return d->trace_n; }
const orc_frame_trace *orc_trace_data(const orc_decoder *d) { return d->trace; }

static void trace_push(orc_decoder *d, int start, int valid, int total, int ok, int bs, int wi) {
  if (!d->trace_on) return;
  if (d->trace_n == d->trace_cap) {
    int cap = d->trace_cap ? d->trace_cap * 2 : 256;
    orc_frame_trace *t = (orc_frame_trace *)realloc(d->trace, sizeof *t * (size_t)cap);
    if (!t) return;
    d->trace = t;
    d->trace_cap = cap;
  }
  d->trace[d->trace_n].start = start;
  d->trace[d->trace_n].valid = valid;
  d->trace[d->trace_n].total = total;
  d->trace[d->trace_n].ok = ok;
  d->trace[d->trace_n].block_size = bs;
  d->trace[d->trace_n].window_index = wi;
  d->trace_n++;
}

/* StreamDecoder.cs:465-530.  Returns the decoded planes (== d->next_buf) or NULL. */
static float **decode_next_packet(orc_decoder *d, int *start, int *valid, int *total, int *is_eos, int *has_pos,
                                  int64_t *pos, int *err) {
  orc_packet p;
  *err = ORC_OK;
  *start = *valid = *total = 0;
  *has_pos = 0;
  *pos = 0;
  if (!get_packet(d, d->next_packet, &p)) {
    *is_eos = 1; /* no packet => end of stream (:472-475) */
    return NULL;
  }
  d->next_packet++;
  *is_eos = p.is_eos;
  if (p.is_resync) d->has_position = 0; /* :481-484 */

  if (orc_read_bit(&p)) { /* :490 */
    trace_push(d, 0, 0, 0, 0, 0, 0);
    return NULL;
  } else {
    int mode_idx = (int)orc_read_bits(&p, d->mode_field_bits);
    int rc, wi = 0;
    if (mode_idx >= d->nmodes) { /* quirk B-15: IndexOutOfRangeException */
      *err = ORC_ERR_RUNTIME;
      return NULL;
    }
    if (!d->next_buf) { /* :498-505 */
      if (!d->buf_a) {
        d->buf_a = alloc_planes(d->channels, d->block1);
        d->next_buf = d->buf_a;
      } else if (!d->buf_b) {
        d->buf_b = alloc_planes(d->channels, d->block1);
        d->next_buf = d->buf_b;
      } else {
        /* the managed code allocates a fresh zeroed array; reuse the one not referenced by prev_buf */
        int i;
        d->next_buf = (d->prev_buf == d->buf_a) ? d->buf_b : d->buf_a;
        for (i = 0; i < d->channels; i++) memset(d->next_buf[i], 0, sizeof(float) * (size_t)d->block1);
      }
      if (!d->next_buf) {
        *err = ORC_ERR_NOMEM;
        return NULL;
      }
    }
    rc = orc_mode_decode(d, &d->modes[mode_idx], &p, d->next_buf, start, valid, total, &wi);
    if (rc < 0) {
      *err = rc;
      return NULL;
    }
    if (rc == 1) {
      *has_pos = p.has_granule; /* samplePosition = packet.GranulePosition (:509) */
      *pos = p.granule;
      trace_push(d, *start, *valid, *total, 1, d->modes[mode_idx].block_size, wi);
      return d->next_buf;
    }
    *start = *valid = *total = 0;
    trace_push(d, 0, 0, 0, 0, 0, 0);
    return NULL;
  }
}

/* StreamDecoder.cs:532-541 */
static void overlap_buffers(float **previous, float **next, int prev_start, int prev_len, int next_start,
                            int channels) {
  int c;
  for (; prev_start < prev_len; prev_start++, next_start++)
    for (c = 0; c < channels; c++) next[c][next_start] += previous[c][prev_start];
}

/* StreamDecoder.cs:417-463 */
static int read_next_packet(orc_decoder *d, int buffered_samples, int *has_pos, int64_t *pos, int *err) {
  int start, valid, total, is_eos;
  float **cur = decode_next_packet(d, &start, &valid, &total, &is_eos, has_pos, pos, err);
  d->eos_found |= is_eos;
  if (cur == NULL) return 0;

  if (*has_pos && is_eos) { /* :429-437 */
    int64_t actual_end = d->current_position + buffered_samples + valid - start;
    int diff = (int)(*pos - actual_end);
    if (diff < 0) valid += diff;
  }

  if (d->prev_end > 0) { /* :440-445 */
    overlap_buffers(d->prev_buf, cur, d->prev_start, d->prev_stop, start, d->channels);
    d->prev_start = start;
  } else if (d->prev_buf == NULL) { /* :446-450 */
    d->prev_start = valid;
  }

  d->next_buf = d->prev_buf; /* :456 */
  d->prev_end = valid;
  d->prev_stop = total;
  d->prev_buf = cur;
  return 1;
}

/* StreamDecoder.cs:320-389 */
static int stream_read(orc_decoder *d, float *buffer, int buffer_len, int offset, int count) {
  int idx, tgt;
  if (offset < 0 || offset + count > buffer_len) return ORC_ERR_ARGUMENT;
  if (count % d->channels != 0) return ORC_ERR_ARGUMENT;
  if (count == 0) return 0;

  idx = offset;
  tgt = offset + count;
  while (idx < tgt) {
    int copy_len;
    if (d->prev_start == d->prev_end) {
      int has_pos = 0, err = ORC_OK;
      int64_t pos = 0;
      if (d->eos_found) {
        d->next_buf = NULL;
        d->prev_buf = NULL;
        break;
      }
      if (!read_next_packet(d, (idx - offset) / d->channels, &has_pos, &pos, &err)) {
        if (err) {
          d->last_error = err;
          return err;
        }
        d->prev_end = d->prev_stop; /* drain (:352-356) */
        has_pos = 0;                /* samplePosition = null on the failure path (:520) */
      }
      if (has_pos && !d->has_position) { /* :359-363 */
        d->has_position = 1;
        d->current_position = pos - (d->prev_end - d->prev_start) - (idx - offset) / d->channels;
      }
    }

    copy_len = (tgt - idx) / d->channels;
    if (copy_len > d->prev_end - d->prev_start) copy_len = d->prev_end - d->prev_start;
    if (copy_len > 0) {
      int ch;
      /* ClippingCopyBuffer / CopyBuffer (:391-415) */
      for (; copy_len > 0; d->prev_start++, copy_len--) {
        for (ch = 0; ch < d->channels; ch++) {
          float s = d->prev_buf[ch][d->prev_start];
          buffer[idx++] = d->clip_samples ? orc_clip_value(s, &d->has_clipped) : s;
        }
      }
    } else if (d->prev_start != d->prev_end) {
      /* validLen < startIndex after an EOS trim: the managed loop would spin forever; stop instead */
      d->last_error = ORC_ERR_RUNTIME;
      return ORC_ERR_RUNTIME;
    }
  }
  count = idx - offset;
  d->current_position += count / d->channels;
  return count;
}

/* StreamDecoder.cs:294-305 */
static void reset_decoder(orc_decoder *d) {
  d->prev_buf = NULL;
  d->prev_start = 0;
  d->prev_end = 0;
  d->prev_stop = 0;
  d->next_buf = NULL;
  d->eos_found = 0;
  d->has_clipped = 0;
  d->has_position = 0;
}

int64_t orc_total_samples(const orc_decoder *d) { return d->max_granule; }

/* StreamDecoder.cs:562-628, SeekOrigin.Begin */
int orc_seek_to(orc_decoder *d, int64_t sample_position) {
  int64_t k = 0, pos = 0;
  int roll_forward, rc, has_pos = 0, err = ORC_OK;
  int64_t p64 = 0;
  if (!d || !d->ogg_bytes) return ORC_ERR_STATE; /* "Seek is not supported by the Contracts.IPacketProvider instance." */
  if (sample_position < 0) return ORC_ERR_ARGUMENT;
  if (sample_position == 0) {
    rc = orc_ogg_seek(d->ogg_bytes, d->ogg_len, d, 0, 0, &k, &pos); /* "short circuit for the looping case" */
    roll_forward = 0;
  } else {
    rc = orc_ogg_seek(d->ogg_bytes, d->ogg_len, d, sample_position, 1, &k, &pos);
    roll_forward = (int)(sample_position - pos);
  }
  if (rc != ORC_OK) return rc;
  d->next_packet = (int)k;
  reset_decoder(d);
  d->has_position = 1;
  if (!read_next_packet(d, 0, &has_pos, &p64, &err)) { /* the pre-roll packet */
    if (err) return err;
    d->eos_found = 1;
    if (d->max_granule != sample_position) return ORC_ERR_STATE; /* "Could not read pre-roll packet!" */
    d->prev_start = d->prev_stop;
    d->current_position = sample_position;
    return ORC_OK;
  }
  if (!read_next_packet(d, 0, &has_pos, &p64, &err)) { /* the actual packet */
    if (err) return err;
    reset_decoder(d);
    d->eos_found = 1;
    return ORC_ERR_STATE;
  }
  d->prev_start += roll_forward;
  d->current_position = sample_position;
  return ORC_OK;
}

/* VorbisReader.cs:336-345 */
int orc_read_samples(orc_decoder *d, float *buffer, int buffer_len, int offset, int count) {
  count -= count % d->channels;
  if (count > 0) return stream_read(d, buffer, buffer_len, offset, count);
  return 0;
}

int orc_decode_packet_block(orc_decoder *d, const uint8_t *pkt, int len, float *planes, int *start, int *valid,
                            int *total, int *block_size) {
  orc_packet p;
  float **tmp;
  int mode_idx, rc, wi = 0, c;
  orc_packet_init(&p, pkt, len);
  *start = *valid = *total = *block_size = 0;
  if (orc_read_bit(&p)) return 0;
  mode_idx = (int)orc_read_bits(&p, d->mode_field_bits);
  if (mode_idx >= d->nmodes) return ORC_ERR_RUNTIME;
  tmp = alloc_planes(d->channels, d->block1);
  if (!tmp) return ORC_ERR_NOMEM;
  rc = orc_mode_decode(d, &d->modes[mode_idx], &p, tmp, start, valid, total, &wi);
  if (rc == 1) {
    *block_size = d->modes[mode_idx].block_size;
    for (c = 0; c < d->channels; c++) memcpy(planes + (size_t)c * d->block1, tmp[c], sizeof(float) * (size_t)d->block1);
  }
  free_planes(tmp, d->channels);
  return rc;
}

/* Operator-level entry points for the parity tests. */
int orc_floor1_apply_posts(orc_decoder *d, int floor_index, int block_size, const int *posts, int post_count,
                           float *residue, int reslen) {
  orc_floor_data data;
  int i;
  if (!d || floor_index < 0 || floor_index >= d->nfloors || d->floors[floor_index].type != 1) return ORC_ERR_ARGUMENT;
  if (post_count < 0 || post_count > 256) return ORC_ERR_ARGUMENT;
  memset(&data, 0, sizeof data);
  data.type = 1;
  data.post_count = post_count;
  for (i = 0; i < post_count; i++) data.posts[i] = posts[i];
  return orc_floor_apply(&d->floors[floor_index], &data, block_size, residue, reslen);
}

int orc_book_count(const orc_decoder *d) { return d ? d->nbooks : 0; }

int orc_codebook_info(const orc_decoder *d, int book_index, int *dimensions, int *entries, int *map_type, int *prefix_bits,
                      int *max_bits, int *n_prefix, int *n_overflow) {
  if (!d || book_index < 0 || book_index >= d->nbooks) return -1;
  const orc_codebook *b = &d->books[book_index];
  if (dimensions) *dimensions = b->dimensions;
  if (entries) *entries = b->entries;
  if (map_type) *map_type = b->map_type;
  if (prefix_bits) *prefix_bits = b->prefix_bits;
  if (max_bits) *max_bits = b->max_bits;
  if (n_prefix) *n_prefix = b->prefix ? b->prefix_count : 0;
  if (n_overflow) *n_overflow = b->overflow ? b->overflow_count : -1;
  return 0;
}

static void orc_copy_nodes(const orc_huff_node *v, int n, int32_t *out) {
  for (int i = 0; i < n; i++) {
    out[5 * i] = v[i].present ? 1 : 0;
    out[5 * i + 1] = v[i].value;
    out[5 * i + 2] = v[i].length;
    out[5 * i + 3] = v[i].bits;
    out[5 * i + 4] = v[i].mask;
  }
}

int orc_codebook_tables(const orc_decoder *d, int book_index, int32_t *lengths, float *lookup, int32_t *prefix, int32_t *overflow) {
  if (!d || book_index < 0 || book_index >= d->nbooks) return -1;
  const orc_codebook *b = &d->books[book_index];
  if (lengths)
    for (int i = 0; i < b->entries; i++) lengths[i] = b->lengths[i];
  if (lookup && b->lookup)
    for (int i = 0; i < b->entries * b->dimensions; i++) lookup[i] = b->lookup[i];
  if (prefix && b->prefix) orc_copy_nodes(b->prefix, b->prefix_count, prefix);
  if (overflow && b->overflow) orc_copy_nodes(b->overflow, b->overflow_count, overflow);
  return 0;
}

int orc_floor_info(const orc_decoder *d, int floor_index, int *type, int *post_count, int *range) {
  if (!d) return 0;
  if (floor_index >= 0 && floor_index < d->nfloors) {
    const orc_floor *f = &d->floors[floor_index];
    if (type) *type = f->type;
    if (post_count) *post_count = f->type == 1 ? f->f1.x_count : f->f0.order;
    if (range) *range = f->type == 1 ? f->f1.range : 0;
  }
  return d->nfloors;
}

int orc_residue_decode_at(orc_decoder *d, int residue_index, const uint8_t *pkt, int len, int bit_offset,
                          int any_channel_decodes, int block_size, float *planes, int *bits_consumed) {
  orc_packet p;
  float **rows;
  int no_decode = !any_channel_decodes, rc, c;
  if (!d || residue_index < 0 || residue_index >= d->nresidues) return ORC_ERR_ARGUMENT;
  rows = (float **)calloc((size_t)d->channels, sizeof *rows);
  if (!rows) return ORC_ERR_NOMEM;
  for (c = 0; c < d->channels; c++) rows[c] = planes + (size_t)c * d->block1;
  orc_packet_init(&p, pkt, len);
  orc_skip_bits(&p, bit_offset);
  rc = orc_residue_decode(&d->residues[residue_index], d->books, &p, &no_decode, 1, block_size, rows, d->block1);
  if (bits_consumed) *bits_consumed = p.pos - bit_offset;
  free(rows);
  return rc;
}

int orc_last_residue_calls(const orc_decoder *d, int *pos, int *idx, int cap, int *any) {
  int i;
  if (!d) return 0;
  for (i = 0; i < d->res_calls && i < cap; i++) {
    pos[i] = d->res_call_pos[i];
    idx[i] = d->res_call_idx[i];
  }
  if (any) *any = d->res_call_any;
  return d->res_calls;
}

/* Test hook: IFloorData of channel `channel` as Mapping.DecodePacket left it for the last packet given to
 * orc_decode_packet_block: *execute = ExecuteChannel, Floor1: posts[64] + *post_count, Floor0: *amp + coeff[order]. */
int orc_last_floor_data(const orc_decoder *d, int channel, int *execute, int *posts, int *post_count, float *amp, float *coeff,
                        int coeff_cap) {
  const orc_floor_data *f;
  int i;
  if (!d || channel < 0 || channel >= d->last_floor_n) return ORC_ERR_ARGUMENT;
  f = &d->last_floor[channel];
  if (execute) *execute = orc_floor_execute_channel(f);
  if (post_count) *post_count = f->post_count;
  if (posts)
    for (i = 0; i < 64; i++) posts[i] = f->posts[i];
  if (amp) *amp = f->amp;
  if (coeff)
    for (i = 0; i < coeff_cap && i < 257; i++) coeff[i] = f->coeff[i];
  return f->type;
}

/* Mapping.cs:16-93 of mapping `mapping_index`: coupling pairs (magnitude, angle), floor index per channel. */
int orc_mapping_info(const orc_decoder *d, int mapping_index, int *coupling_steps, int *magnitude, int *angle, int cap,
                     int *channel_floor, int ch_cap) {
  const orc_mapping *m;
  int i;
  if (!d || mapping_index < 0 || mapping_index >= d->nmappings) return ORC_ERR_ARGUMENT;
  m = &d->mappings[mapping_index];
  if (coupling_steps) *coupling_steps = m->coupling_steps;
  for (i = 0; i < m->coupling_steps && i < cap; i++) {
    if (magnitude) magnitude[i] = m->coupling_magnitude[i];
    if (angle) angle[i] = m->coupling_angle[i];
  }
  for (i = 0; i < m->channels && i < ch_cap; i++)
    if (channel_floor) channel_floor[i] = m->channel_floor[i];
  return ORC_OK;
}

int orc_mode_info(const orc_decoder *d, int mode_index, int *block_flag, int *block_size, int *mapping) {
  if (!d) return 0;
  if (mode_index >= 0 && mode_index < d->nmodes) {
    if (block_flag) *block_flag = d->modes[mode_index].block_flag;
    if (block_size) *block_size = d->modes[mode_index].block_size;
    if (mapping) *mapping = d->modes[mode_index].mapping;
  }
  return d->nmodes;
}

int orc_floor0_apply_coeffs(orc_decoder *d, int floor_index, int block_size, float amp, const float *coeff,
                            float *residue, int reslen) {
  orc_floor_data data;
  int i;
  if (!d || floor_index < 0 || floor_index >= d->nfloors || d->floors[floor_index].type != 0) return ORC_ERR_ARGUMENT;
  memset(&data, 0, sizeof data);
  data.type = 0;
  data.amp = amp;
  for (i = 0; i < d->floors[floor_index].f0.order && i < 257; i++) data.coeff[i] = coeff[i];
  return orc_floor_apply(&d->floors[floor_index], &data, block_size, residue, reslen);
}
