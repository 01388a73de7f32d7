/* oracle/orc_ogg.c -- minimal forward Ogg demux for the oracle (test infrastructure, see orc.h).
 *
 * Restates only what the synthesis path observes of the container:
 *   - page sync "OggS" + CRC-32 (poly 0x04c11db7, no reflection)      Ogg/PageReaderBase.cs:33-70,227-292, Ogg/Crc.cs:5-40
 *   - lacing -> packets, zero-length packets dropped                  Ogg/PageReader.cs:27-93
 *   - pages without packets dropped                                   Ogg/PageReader.cs:131
 *   - continuation across pages, granule attached to the packet that is LAST on the page it
 *     completes on, EOS on that packet of the EOS-flagged page        Ogg/PacketProvider.cs:324-438
 *   - pages after the EOS-flagged page ignored                        Ogg/StreamPageReader.cs:44-91
 * Only the first logical stream (serial of the first page) is demuxed; multiplexing is out of scope.
 * Seeking, orc_ogg_seek at the end of the file:
 *   - StreamPageReader.FindPage / FindPageBisection / FindPageForward   Ogg/StreamPageReader.cs:122-264
 *   - PacketProvider.SeekTo / FindPacket / granule-bug workaround        Ogg/PacketProvider.cs:56-260
 *   - PacketProvider.NormalizePacketIndex                                 Ogg/PacketProvider.cs:262-295
 *   - StreamDecoder.GetPacketGranules                                     StreamDecoder.cs:630-647
 */
#include "orc_internal.h"

typedef struct {
  size_t data_off;    /* offset of the first data byte */
  const uint8_t *seg; /* lacing table */
  int seg_cnt;
  int flags;
  int64_t granule;
  int is_resync, is_continued, packet_count;
  /* per-packet spans on this page */
  int *pk_off, *pk_len;
} ogg_page;

static uint32_t g_crc_table[256];
static int g_crc_ready;

static void crc_init(void) { /* Ogg/Crc.cs:8-20 */
  uint32_t i;
  int j;
  for (i = 0; i < 256; i++) {
    uint32_t s = i << 24;
    for (j = 0; j < 8; ++j) s = (s << 1) ^ (s >= (1u << 31) ? 0x04c11db7u : 0);
    g_crc_table[i] = s;
  }
  g_crc_ready = 1;
}

static int verify_page(const uint8_t *pg, size_t total) { /* Ogg/PageReaderBase.cs:33-70 */
  uint32_t crc = 0, want;
  size_t i;
  for (i = 0; i < total; i++) {
    uint8_t b = (i >= 22 && i < 26) ? 0 : pg[i];
    crc = (crc << 8) ^ g_crc_table[b ^ (crc >> 24)];
  }
  want = (uint32_t)pg[22] | ((uint32_t)pg[23] << 8) | ((uint32_t)pg[24] << 16) | ((uint32_t)pg[25] << 24);
  return crc == want;
}

typedef struct {
  uint8_t *bytes;
  size_t nbytes, cap;
  int64_t *offs, *granule;
  uint8_t *flags;
  int n, ncap;
} pkt_list;

static int pl_begin(pkt_list *l) {
  if (l->n == l->ncap) {
    int cap = l->ncap ? l->ncap * 2 : 256;
    int64_t *o = (int64_t *)realloc(l->offs, sizeof(int64_t) * (size_t)(cap + 1));
    int64_t *g = (int64_t *)realloc(l->granule, sizeof(int64_t) * (size_t)cap);
    uint8_t *f = (uint8_t *)realloc(l->flags, (size_t)cap);
    if (!o || !g || !f) return ORC_ERR_NOMEM;
    l->offs = o;
    l->granule = g;
    l->flags = f;
    l->ncap = cap;
  }
  l->offs[l->n] = (int64_t)l->nbytes;
  l->granule[l->n] = -1;
  l->flags[l->n] = 0;
  return ORC_OK;
}

static int pl_append(pkt_list *l, const uint8_t *src, size_t len) {
  if (l->nbytes + len > l->cap) {
    size_t cap = l->cap ? l->cap * 2 : 65536;
    uint8_t *b;
    while (cap < l->nbytes + len) cap *= 2;
    b = (uint8_t *)realloc(l->bytes, cap);
    if (!b) return ORC_ERR_NOMEM;
    l->bytes = b;
    l->cap = cap;
  }
  memcpy(l->bytes + l->nbytes, src, len);
  l->nbytes += len;
  return ORC_OK;
}

typedef struct {
  ogg_page *pages;
  int npages, has_all_pages, first_data_page;
  int64_t max_granule;
} page_table;

static void free_pages(page_table *t) {
  int i;
  for (i = 0; i < t->npages; i++) {
    free(t->pages[i].pk_off);
    free(t->pages[i].pk_len);
  }
  free(t->pages);
  memset(t, 0, sizeof *t);
}

/* The pages of the first logical stream, as StreamPageReader.AddPage accepts them (Ogg/StreamPageReader.cs:44-91). */
static int scan_pages(const uint8_t *bytes, size_t len, page_table *out) {
  ogg_page *pages = NULL;
  int npages = 0, pcap = 0, have_serial = 0, has_all_pages = 0, rc = ORC_OK;
  int32_t serial = 0;
  size_t pos = 0;
  int resync = 0;
  /* StreamPageReader.AddPage state (Ogg/StreamPageReader.cs:44-91) */
  int32_t last_seq = 0;
  int have_first_data_page = 0, first_data_page = -1;
  int64_t max_granule = 0;
  if (!g_crc_ready) crc_init();

  /* ---- page scan (ReadNextPage loop) ---- */
  while (pos + 27 <= len && !has_all_pages) {
    const uint8_t *h = bytes + pos;
    int seg_cnt, data_len = 0, s;
    size_t total;
    if (!(h[0] == 0x4f && h[1] == 0x67 && h[2] == 0x67 && h[3] == 0x53)) {
      pos++;
      resync = 1;
      continue;
    }
    seg_cnt = h[26];
    if (pos + 27 + (size_t)seg_cnt > len) {
      pos++;
      resync = 1;
      continue;
    }
    for (s = 0; s < seg_cnt; s++) data_len += h[27 + s];
    total = 27 + (size_t)seg_cnt + (size_t)data_len;
    if (pos + total > len || !verify_page(h, total)) {
      pos++;
      resync = 1;
      continue;
    }
    {
      int32_t pg_serial = (int32_t)((uint32_t)h[14] | ((uint32_t)h[15] << 8) | ((uint32_t)h[16] << 16) | ((uint32_t)h[17] << 24));
      int pkt_cnt = 0, size = 0, is_continued = 0;
      /* ParsePageHeader (Ogg/PageReader.cs:27-60) */
      for (s = 0; s < seg_cnt; s++) {
        int seg = h[27 + s];
        size += seg;
        if (seg < 255) {
          if (size > 0) ++pkt_cnt;
          size = 0;
        }
      }
      if (size > 0) {
        is_continued = h[seg_cnt + 26] == 255;
        ++pkt_cnt;
      }
      if (!have_serial) {
        have_serial = 1;
        serial = pg_serial;
      }
      if (pg_serial == serial && pkt_cnt == 0) {
        /* PageReader.AddPage returns false for a page without packets (Ogg/PageReader.cs:131) and the
         * base class then puts the serial on its ignore list (Ogg/PageReaderBase.cs:72-85): nothing
         * more is ever delivered for this stream, and an EOS flag on such a page is never seen. */
        break;
      }
      if (pg_serial == serial && pkt_cnt > 0) {
        ogg_page *pg;
        int k = 0, off = 0;
        if (npages == pcap) {
          int cap = pcap ? pcap * 2 : 64;
          ogg_page *np = (ogg_page *)realloc(pages, sizeof *np * (size_t)cap);
          if (!np) {
            rc = ORC_ERR_NOMEM;
            goto done;
          }
          pages = np;
          pcap = cap;
        }
        pg = &pages[npages++];
        memset(pg, 0, sizeof *pg);
        pg->data_off = pos + 27 + (size_t)seg_cnt;
        pg->seg = h + 27;
        pg->seg_cnt = seg_cnt;
        pg->flags = h[5];
        memcpy(&pg->granule, h + 6, 8); /* little-endian host */
        {
          /* StreamPageReader.AddPage (Ogg/StreamPageReader.cs:50-86): granule sanity, then the resync mark -- lost page sync
           * or a page sequence number that does not follow the previous one */
          const int32_t seq = (int32_t)((uint32_t)h[18] | ((uint32_t)h[19] << 8) | ((uint32_t)h[20] << 16) | ((uint32_t)h[21] << 24));
          if (pg->granule != -1) {
            if (!have_first_data_page && pg->granule > 0) {
              have_first_data_page = 1;
              first_data_page = npages - 1; /* _firstDataPageIndex = _pageOffsets.Count (this page is not added yet) */
            } else if (max_granule > pg->granule) {
              rc = ORC_ERR_INVALID_DATA; /* "Granule Position regressed?!" */
              npages--;
              goto done;
            }
            max_granule = pg->granule;
          } else if (have_first_data_page && (!is_continued || pkt_cnt != 1)) {
            rc = ORC_ERR_INVALID_DATA; /* "Granule Position was -1 but page does not have exactly 1 continued packet." */
            npages--;
            goto done;
          }
          pg->is_resync = resync || (last_seq != 0 && (int32_t)((uint32_t)last_seq + 1u) != seq);
          last_seq = seq;
        }
        pg->is_continued = is_continued;
        pg->packet_count = pkt_cnt;
        pg->pk_off = (int *)calloc((size_t)pkt_cnt, sizeof(int));
        pg->pk_len = (int *)calloc((size_t)pkt_cnt, sizeof(int));
        if (!pg->pk_off || !pg->pk_len) {
          rc = ORC_ERR_NOMEM;
          goto done;
        }
        /* ReadPackets (Ogg/PageReader.cs:62-91) */
        size = 0;
        for (s = 0; s < seg_cnt; s++) {
          int seg = h[27 + s];
          size += seg;
          if (seg < 255) {
            if (size > 0) {
              pg->pk_off[k] = off;
              pg->pk_len[k] = size;
              k++;
              off += size;
            }
            size = 0;
          }
        }
        if (size > 0) {
          pg->pk_off[k] = off;
          pg->pk_len[k] = size;
        }
        if (pg->flags & 0x04) has_all_pages = 1; /* PageFlags.EndOfStream (StreamPageReader.cs:72-75) */
      }
      resync = 0;
      pos += total;
    }
  }

done:
  out->pages = pages;
  out->npages = npages;
  out->has_all_pages = has_all_pages;
  out->first_data_page = first_data_page;
  out->max_granule = max_granule;
  if (rc != ORC_OK) free_pages(out);
  return rc;
}

int orc_ogg_demux(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                  uint8_t **out_flags, int *out_n) {
  page_table tab;
  ogg_page *pages;
  int npages, has_all_pages, rc;
  pkt_list pl;
  memset(&pl, 0, sizeof pl);
  memset(&tab, 0, sizeof tab);
  rc = scan_pages(bytes, len, &tab);
  if (rc != ORC_OK) return rc;
  pages = tab.pages;
  npages = tab.npages;
  has_all_pages = tab.has_all_pages;

  /* ---- packet assembly (PacketProvider.GetNextPacket/CreatePacket) ---- */
  {
    int page_index = 0, packet_index = 0;
    while (page_index < npages) {
      ogg_page *pg = &pages[page_index];
      int64_t granule_pos = pg->granule;
      int is_resync = pg->is_resync, is_continued = pg->is_continued, packet_count = pg->packet_count;
      int is_last_packet, final_page = page_index, truncated = 0;
      if (packet_index >= packet_count) { /* defensive; cannot happen for pages with >=1 packet */
        page_index++;
        packet_index = 0;
        continue;
      }
      if ((rc = pl_begin(&pl)) != ORC_OK) goto done;
      if ((rc = pl_append(&pl, bytes + pg->data_off + pg->pk_off[packet_index], (size_t)pg->pk_len[packet_index])) != ORC_OK) goto done;

      if (is_continued && packet_index == packet_count - 1) {
        int cont = page_index;
        while (is_continued) {
          ogg_page *np;
          int is_continuation;
          if (++cont >= npages) { /* GetPage failed: "we can't satisfy the request" -> null packet => end */
            truncated = 1;
            break;
          }
          np = &pages[cont];
          granule_pos = np->granule;
          is_resync = np->is_resync;
          is_continuation = (np->flags & 0x01) != 0;
          is_continued = np->is_continued;
          packet_count = np->packet_count;
          if (!is_continuation || is_resync) break;
          if (is_continued && packet_count > 1) is_continued = 0;
          if ((rc = pl_append(&pl, bytes + np->data_off + np->pk_off[0], (size_t)np->pk_len[0])) != ORC_OK) goto done;
        }
        if (truncated) {
          pl.nbytes = (size_t)pl.offs[pl.n]; /* drop the partial packet; stream ends here */
          break;
        }
        is_last_packet = packet_count == 1;
        final_page = cont;
      } else {
        is_last_packet = packet_index == packet_count - 1;
      }

      if (is_resync) pl.flags[pl.n] |= 2;
      if (is_last_packet) {
        pl.granule[pl.n] = granule_pos;
        if (granule_pos < 0) pl.granule[pl.n] = -1; /* -1 == "no position" on continued-only pages */
        if (has_all_pages && final_page == npages - 1) pl.flags[pl.n] |= 1;
      }
      pl.n++;

      /* advance (:411-433) */
      if (final_page != page_index) {
        page_index = final_page;
        packet_index = 0;
      }
      if (packet_index == packet_count - 1) {
        ++page_index;
        packet_index = 0;
      } else {
        ++packet_index;
      }
    }
  }
  if ((rc = pl_begin(&pl)) != ORC_OK) goto done; /* terminal offset */

done:
  free_pages(&tab);
  if (rc != ORC_OK) {
    free(pl.bytes);
    free(pl.offs);
    free(pl.granule);
    free(pl.flags);
    return rc;
  }
  if (!pl.bytes) pl.bytes = (uint8_t *)malloc(1);
  *out_bytes = pl.bytes;
  *out_offs = pl.offs;
  *out_granule = pl.granule;
  *out_flags = pl.flags;
  *out_n = pl.n;
  return ORC_OK;
}

/* ================================================================================================
 * Seeking: IPacketProvider.SeekTo(granulePos, preRoll, getPacketGranuleCount) restated method by method
 * (test infrastructure; the reader is taken to have read every page already, as after TotalSamples).
 * ================================================================================================ */

typedef struct {
  const uint8_t *bytes;
  const page_table *t;
  orc_decoder *d;
  int err; /* ORC_ERR_* raised by a helper */
} seek_ctx;

/* StreamDecoder.GetPacketGranules (StreamDecoder.cs:630-647) on an assembled packet */
static int packet_granules(seek_ctx *c, const uint8_t *data, int len, int is_resync) {
  orc_packet p;
  int mode_idx;
  const orc_mode *m;
  if (is_resync) return 0;
  orc_packet_init(&p, data, len);
  if (orc_read_bit(&p)) return 0;
  mode_idx = (int)orc_read_bits(&p, c->d->mode_field_bits);
  if (mode_idx < 0 || mode_idx >= c->d->nmodes) return 0;
  m = &c->d->modes[mode_idx];
  /* Mode.GetPacketSampleCount -> GetPacketInfo (Mode.cs:119-152, 172-177): valid - start */
  if (p.is_short) return 0;
  if (m->block_flag) {
    int prev_flag = orc_read_bit(&p);
    int next_flag = orc_read_bit(&p);
    int wi = (prev_flag ? 1 : 0) + (next_flag ? 2 : 0);
    return m->ov_valid[wi] - m->ov_start[wi];
  }
  return m->block_size / 2;
}

/* PacketProvider.CreatePacket with advance = false (Ogg/PacketProvider.cs:324-400), then getPacketGranuleCount on it.
 * Returns 0 and sets *granules, or -1 where CreatePacket returns null, or -2 with c->err set. */
static int create_packet_granules(seek_ctx *c, int page_index, int packet_index, int is_resync, int is_continued, int packet_count,
                                  int *granules) {
  const ogg_page *pg = &c->t->pages[page_index];
  uint8_t *buf;
  size_t n = 0, cap;
  if (packet_index < 0 || packet_index >= pg->packet_count) { /* GetPagePackets(pageIndex)[packetIndex] */
    c->err = ORC_ERR_RUNTIME;
    return -2;
  }
  cap = (size_t)pg->pk_len[packet_index] + 16;
  buf = (uint8_t *)malloc(cap);
  if (!buf) {
    c->err = ORC_ERR_NOMEM;
    return -2;
  }
  memcpy(buf, c->bytes + pg->data_off + pg->pk_off[packet_index], (size_t)pg->pk_len[packet_index]);
  n = (size_t)pg->pk_len[packet_index];
  if (is_continued && packet_index == packet_count - 1) {
    int cont = page_index;
    while (is_continued) {
      const ogg_page *np;
      int is_continuation;
      if (++cont >= c->t->npages) {
        free(buf);
        return -1;
      }
      np = &c->t->pages[cont];
      is_resync = np->is_resync;
      is_continuation = (np->flags & 0x01) != 0;
      is_continued = np->is_continued;
      packet_count = np->packet_count;
      if (!is_continuation || is_resync) break;
      if (is_continued && packet_count > 1) is_continued = 0;
      if (n + (size_t)np->pk_len[0] > cap) {
        uint8_t *nb;
        cap = (n + (size_t)np->pk_len[0]) * 2;
        nb = (uint8_t *)realloc(buf, cap);
        if (!nb) {
          free(buf);
          c->err = ORC_ERR_NOMEM;
          return -2;
        }
        buf = nb;
      }
      memcpy(buf + n, c->bytes + np->data_off + np->pk_off[0], (size_t)np->pk_len[0]);
      n += (size_t)np->pk_len[0];
    }
  }
  *granules = packet_granules(c, buf, (int)n, is_resync);
  free(buf);
  return 0;
}

/* Ogg/PacketProvider.cs:224-260 */
static int get_is_vorbis_bug_diff(int64_t diff) {
  int64_t temp;
  int short_block_bits = 0, long_block_bits;
  if (diff < 0) diff = -diff;
  temp = diff;
  while (temp > 0 && (temp & 1) == 0) {
    ++short_block_bits;
    temp >>= 1;
  }
  long_block_bits = short_block_bits;
  while ((temp & 1) == 1) {
    ++long_block_bits;
    temp >>= 1;
  }
  /* C#: (1 << longBlockBits) - (1 << shortBlockBits) on int -- shift counts are taken mod 32, the subtraction wraps */
  {
    int32_t a = (int32_t)(1u << (long_block_bits & 31)), b = (int32_t)(1u << (short_block_bits & 31));
    return temp == 0 && diff == (int64_t)(int32_t)((uint32_t)a - (uint32_t)b);
  }
}

/* Ogg/StreamPageReader.cs:232-264 */
static int find_page_bisection(const page_table *t, int64_t granule_pos, int low, int high, int64_t high_granule_pos, int *fault) {
  int64_t low_granule_pos = 0;
  int dist;
  while ((dist = high - low) > 0) {
    int index = low + (int)(dist * ((granule_pos - low_granule_pos) / (double)(high_granule_pos - low_granule_pos)));
    int64_t idx_granule_pos;
    if (index < 0 || index >= t->npages) { /* _pageOffsets[index] */
      *fault = 1;
      return -1;
    }
    idx_granule_pos = t->pages[index].granule;
    if (idx_granule_pos > granule_pos) {
      high = index;
      high_granule_pos = idx_granule_pos;
    } else if (idx_granule_pos < granule_pos) {
      low = index + 1;
      low_granule_pos = idx_granule_pos + 1;
    } else {
      return index + 1;
    }
  }
  return low;
}

/* Ogg/StreamPageReader.cs:122-160, every page already read */
static int find_page(const page_table *t, int64_t granule_pos, int *fault) {
  int page_index = -1;
  if (granule_pos == 0) {
    page_index = t->first_data_page;
  } else {
    int last_page_index = t->npages - 1;
    if (last_page_index >= 0) {
      int64_t page_gp = t->pages[last_page_index].granule;
      if (granule_pos < page_gp) {
        page_index = find_page_bisection(t, granule_pos, t->first_data_page, last_page_index, page_gp, fault);
      } else if (granule_pos > page_gp) {
        /* FindPageForward (:171-199): the next index is past the last page, GetNextPageGranulePos finds nothing more to
         * read and marks the reader complete (:201-230); "allow finding the last granulePos" */
        page_index = last_page_index + 1;
        if (t->max_granule < granule_pos) page_index = -1;
      } else {
        page_index = last_page_index + 1;
      }
    }
  }
  return page_index;
}

/* position of slot (page, packet) in the list GetNextPacket produces from the start of the stream, or -1 */
static int64_t list_index_of(const page_table *t, int page, int packet) {
  int page_index = 0, packet_index = 0;
  int64_t k = 0;
  while (page_index < t->npages) {
    const ogg_page *pg = &t->pages[page_index];
    int is_continued = pg->is_continued, packet_count = pg->packet_count, final_page = page_index;
    if (page_index == page && packet_index == packet) return k;
    if (page_index > page) return -1;
    if (is_continued && packet_index == packet_count - 1) {
      int cont = page_index;
      while (is_continued) {
        const ogg_page *np;
        if (++cont >= t->npages) return -1;
        np = &t->pages[cont];
        is_continued = np->is_continued;
        packet_count = np->packet_count;
        if (!(np->flags & 0x01) || np->is_resync) break;
        if (is_continued && packet_count > 1) is_continued = 0;
      }
      final_page = cont;
    }
    k++;
    if (final_page != page_index) {
      page_index = final_page;
      packet_index = 0;
    }
    if (packet_index == packet_count - 1) {
      ++page_index;
      packet_index = 0;
    } else {
      ++packet_index;
    }
  }
  return -1;
}

int orc_ogg_seek(const uint8_t *bytes, size_t len, orc_decoder *d, int64_t granule_pos, int pre_roll, int64_t *packet_index_out,
                 int64_t *granule_out) {
  page_table tab;
  seek_ctx c;
  int rc, page_index, packet_index = 0, fault = 0;
  int64_t last_page_granule_pos = 0, end_gp, *gps = NULL;
  int last_page_packet_length = 0, first_real_packet = 0, packet_count, i, found;
  const ogg_page *pg;
  memset(&tab, 0, sizeof tab);
  if (!bytes || !d || !packet_index_out || !granule_out) return ORC_ERR_ARGUMENT;
  if ((rc = scan_pages(bytes, len, &tab)) != ORC_OK) return rc;
  c.bytes = bytes;
  c.t = &tab;
  c.d = d;
  c.err = ORC_OK;
  rc = ORC_OK;

  /* SeekTo (:56-72) */
  page_index = find_page(&tab, granule_pos, &fault);
  if (fault) { rc = ORC_ERR_RUNTIME; goto done; }
  if (page_index == -1) { rc = ORC_ERR_ARGUMENT; goto done; } /* FindPage throws ArgumentOutOfRangeException */

  /* FindPacket(pageIndex, preRoll, ref granulePos, ..) (:204-222): GetPreviousPageInfo (:74-108) */
  if (page_index > 0) {
    const ogg_page *prev;
    if (page_index - 1 >= tab.npages) { rc = ORC_ERR_INVALID_DATA; goto done; } /* "Could not get preceding page?!" */
    prev = &tab.pages[page_index - 1];
    last_page_granule_pos = prev->granule;
    if (page_index > tab.first_data_page) {
      int r = create_packet_granules(&c, page_index - 1, prev->packet_count - 1, 0, prev->is_continued, prev->packet_count,
                                     &last_page_packet_length);
      if (r == -1) { rc = ORC_ERR_INVALID_DATA; goto done; } /* "Could not find end of continuation!" */
      if (r == -2) { rc = c.err; goto done; }
    } else {
      last_page_packet_length = 0;
    }
    first_real_packet = prev->is_continued ? 1 : 0;
  }
  /* GetTargetPageInfo (:110-146) */
  if (page_index >= tab.npages) { rc = ORC_ERR_INVALID_DATA; goto done; } /* "Could not get found page?!" */
  pg = &tab.pages[page_index];
  packet_count = pg->packet_count;
  if (pg->is_continued) packet_count--;
  gps = (int64_t *)calloc((size_t)(packet_count > 0 ? packet_count : 1), sizeof *gps);
  if (!gps) { rc = ORC_ERR_NOMEM; goto done; }
  end_gp = pg->granule;
  for (i = packet_count - 1; i >= first_real_packet; i--) {
    int g = 0, r;
    gps[i] = end_gp;
    r = create_packet_granules(&c, page_index, i, i == 0 && pg->is_resync, pg->is_continued, packet_count, &g);
    if (r == -1) { rc = ORC_ERR_INVALID_DATA; goto done; }
    if (r == -2) { rc = c.err; goto done; }
    end_gp -= g;
  }
  if (first_real_packet == 1) {
    if (packet_count < 1) { rc = ORC_ERR_RUNTIME; goto done; } /* gps[0] of an empty array */
    gps[0] = end_gp;
    end_gp -= last_page_packet_length;
  }
  /* FindPacket(pageIndex, gps, endGP, lastPageGranulePos, lastPagePacketLength, ref granulePos) (:148-202) */
  found = 0;
  if (end_gp != last_page_granule_pos) {
    int64_t diff = end_gp - last_page_granule_pos;
    if (get_is_vorbis_bug_diff(diff)) {
      if (diff > 0) {
        if (granule_pos <= end_gp) {
          granule_pos = end_gp - last_page_packet_length;
          packet_index = -1;
          found = 1;
        }
      } else {
        for (i = 0; i < packet_count; i++) gps[i] -= diff;
      }
    } else if (page_index > tab.first_data_page) {
      rc = ORC_ERR_INVALID_DATA; /* "GranulePos mismatch" */
      goto done;
    }
  }
  if (!found) {
    for (i = 0; i < packet_count; i++) {
      if (gps[i] >= granule_pos) {
        granule_pos = i == 0 ? end_gp : gps[i - 1];
        packet_index = i;
        found = 1;
        break;
      }
    }
    if (!found) { rc = ORC_ERR_INVALID_DATA; goto done; } /* "Could not find seek packet?!" */
  }
  if (end_gp > 0 || packet_index > 1) packet_index -= pre_roll; /* :216-220 */

  /* NormalizePacketIndex (:262-295) */
  {
    int is_resync = pg->is_resync, is_continuation = (pg->flags & 0x01) != 0;
    int pg_idx = page_index, pkt_idx = packet_index;
    while (pkt_idx < (is_continuation ? 1 : 0)) {
      int was_continuation = is_continuation;
      const ogg_page *pp;
      if (is_continuation && is_resync) { rc = ORC_ERR_ARGUMENT; goto done; }
      if (--pg_idx < 0) { rc = ORC_ERR_ARGUMENT; goto done; }
      pp = &tab.pages[pg_idx];
      is_resync = pp->is_resync;
      is_continuation = (pp->flags & 0x01) != 0;
      if (was_continuation && !pp->is_continued) { rc = ORC_ERR_ARGUMENT; goto done; }
      pkt_idx += pp->packet_count - (was_continuation ? 1 : 0);
    }
    page_index = pg_idx;
    packet_index = pkt_idx;
  }
  {
    int64_t k = list_index_of(&tab, page_index, packet_index);
    if (k < 0) { rc = ORC_ERR_RUNTIME; goto done; }
    *packet_index_out = k;
    *granule_out = granule_pos;
  }
done:
  free(gps);
  free_pages(&tab);
  return rc;
}

int orc_ogg_max_granule(const uint8_t *bytes, size_t len, int64_t *max_granule) {
  page_table tab;
  int rc;
  memset(&tab, 0, sizeof tab);
  if ((rc = scan_pages(bytes, len, &tab)) != ORC_OK) return rc;
  *max_granule = tab.max_granule;
  free_pages(&tab);
  return ORC_OK;
}

/* ================================================================================================
 * Forward-only reader: ForwardOnlyPageReader.AddPage (Ogg/ForwardOnlyPageReader.cs:21-52) and
 * ForwardOnlyPacketProvider (Ogg/ForwardOnlyPacketProvider.cs), first logical stream, as a packet list.
 * ================================================================================================ */

typedef struct {
  /* the page queue: pages of the stream in file order, pulled one at a time */
  const uint8_t **queue_buf;
  uint8_t *queue_resync;
  int queue_n, queue_pos;
  /* ForwardOnlyPacketProvider fields (:10-22) */
  int last_seq_no;
  const uint8_t *page_buf;
  int packet_index;
  int is_end_of_stream;
  int data_start;
} fwd_provider;

/* :36-67 */
static int fwd_add_page(fwd_provider *pp, const uint8_t *buf, int is_resync, int *resync_out) {
  int ttl = 0, i, seq_no = (int32_t)((uint32_t)buf[18] | ((uint32_t)buf[19] << 8) | ((uint32_t)buf[20] << 16) | ((uint32_t)buf[21] << 24));
  if (buf[5] & 0x02) { /* PageFlags.BeginningOfStream */
    if (pp->is_end_of_stream) return 0;
    is_resync = 1;
    pp->last_seq_no = seq_no;
  } else {
    is_resync |= seq_no != (int32_t)((uint32_t)pp->last_seq_no + 1u);
    pp->last_seq_no = seq_no;
  }
  for (i = 0; i < buf[26]; i++) ttl += buf[27 + i];
  if (ttl == 0) return 0;
  *resync_out = is_resync;
  return 1;
}

/* :270-284 */
static int fwd_get_packet_length(const uint8_t *page_buf, int *packet_index) {
  int len = 0;
  while (*packet_index < page_buf[26] + 27 && page_buf[*packet_index] == 255) {
    len += page_buf[*packet_index];
    ++*packet_index;
  }
  if (*packet_index < page_buf[26] + 27) {
    len += page_buf[*packet_index];
    ++*packet_index;
  }
  return len;
}

/* :248-268 */
static int fwd_read_next_page(fwd_provider *pp, const uint8_t **page_buf, int *is_resync, int *data_start, int *packet_index,
                              int *is_continuation, int *is_continued) {
  if (pp->queue_pos >= pp->queue_n) { /* queue empty; _isEndOfStream, or the reader finds no further page */
    *page_buf = NULL;
    *is_resync = 0;
    *data_start = 0;
    *packet_index = 0;
    *is_continuation = 0;
    *is_continued = 0;
    return 0;
  }
  *page_buf = pp->queue_buf[pp->queue_pos];
  *is_resync = pp->queue_resync[pp->queue_pos];
  pp->queue_pos++;
  *data_start = (*page_buf)[26] + 27;
  *packet_index = 27;
  *is_continuation = ((*page_buf)[5] & 0x01) != 0;
  *is_continued = (*page_buf)[26 + (*page_buf)[26]] == 255;
  return 1;
}

int orc_ogg_demux_forward(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                          uint8_t **out_flags, int *out_n) {
  fwd_provider pp;
  pkt_list pl;
  size_t pos = 0;
  int resync = 0, have_serial = 0, ignored = 0, rc = ORC_OK, qcap = 0, gone = 0;
  int32_t serial = 0;
  memset(&pp, 0, sizeof pp);
  memset(&pl, 0, sizeof pl);
  pp.packet_index = 0x7fffffff; /* "force the first page to read" (:29-30) */
  if (!g_crc_ready) crc_init();

  /* PageReaderBase.ReadNextPage (Ogg/PageReaderBase.cs:227-292) over the whole input */
  while (pos + 27 <= len) {
    const uint8_t *h = bytes + pos;
    int seg_cnt, data_len = 0, s;
    size_t total;
    if (!(h[0] == 0x4f && h[1] == 0x67 && h[2] == 0x67 && h[3] == 0x53)) {
      pos++;
      resync = 1;
      continue;
    }
    seg_cnt = h[26];
    if (pos + 27 + (size_t)seg_cnt > len) {
      pos++;
      resync = 1;
      continue;
    }
    for (s = 0; s < seg_cnt; s++) data_len += h[27 + s];
    total = 27 + (size_t)seg_cnt + (size_t)data_len;
    if (pos + total > len || !verify_page(h, total)) {
      pos++;
      resync = 1;
      continue;
    }
    {
      int32_t pg_serial = (int32_t)((uint32_t)h[14] | ((uint32_t)h[15] << 8) | ((uint32_t)h[16] << 16) | ((uint32_t)h[17] << 24));
      if (!have_serial) {
        have_serial = 1;
        serial = pg_serial;
      }
      if (pg_serial == serial && !ignored && !gone) {
        int pg_resync = 0;
        if (fwd_add_page(&pp, h, resync, &pg_resync)) {
          if (pp.queue_n == qcap) {
            int cap = qcap ? qcap * 2 : 64;
            const uint8_t **nb = (const uint8_t **)realloc((void *)pp.queue_buf, sizeof *nb * (size_t)cap);
            uint8_t *nr = nb ? (uint8_t *)realloc(pp.queue_resync, (size_t)cap) : NULL;
            if (nb) pp.queue_buf = nb;
            if (nr) pp.queue_resync = nr;
            if (!nb || !nr) {
              rc = ORC_ERR_NOMEM;
              goto done;
            }
            qcap = cap;
          }
          pp.queue_buf[pp.queue_n] = h;
          pp.queue_resync[pp.queue_n] = (uint8_t)pg_resync;
          pp.queue_n++;
          if (h[5] & 0x04) { /* ForwardOnlyPageReader.cs:28-33: SetEndOfStream, the provider leaves the reader's list */
            pp.is_end_of_stream = 1;
            gone = 1; /* a later page with this serial would open another stream: not this one's business */
          }
        } else {
          ignored = 1; /* PageReaderBase.AddPage: the serial goes on the ignore list (:72-85) */
        }
      }
      resync = 0;
      pos += total;
    }
  }
  pp.is_end_of_stream = 0; /* the flag is raised when the EOS page is READ, which happens when the queue runs dry: see below */

  /* GetNextPacket until null: GetPacket (:119-246) */
  for (;;) {
    const uint8_t *page_buf;
    int is_resync, data_start, packet_index, is_cont, is_cntd, is_first, data_len, is_last, is_eos = 0, has_granule = 0;
    int64_t granule_pos = 0;
    size_t first_off;
    if (pp.page_buf != NULL && pp.packet_index < 27 + pp.page_buf[26]) {
      page_buf = pp.page_buf;
      is_resync = 0;
      data_start = pp.data_start;
      packet_index = pp.packet_index;
      is_cont = 0;
      is_cntd = page_buf[26 + page_buf[26]] == 255;
    } else {
      if (!fwd_read_next_page(&pp, &page_buf, &is_resync, &data_start, &packet_index, &is_cont, &is_cntd)) break;
    }
    is_first = packet_index == 27;
    if (is_cont) {
      if (is_first) {
        is_resync = 1;
        (void)fwd_get_packet_length(page_buf, &packet_index); /* contOverhead += ...: the data offset is NOT advanced (:152) */
        if (packet_index == 27 + page_buf[26]) continue;      /* return GetPacket(): the saved page is still the previous one */
      }
    }
    data_len = fwd_get_packet_length(page_buf, &packet_index);
    if ((rc = pl_begin(&pl)) != ORC_OK) goto done;
    first_off = (size_t)(page_buf - bytes) + (size_t)data_start;
    {
      size_t avail = first_off <= len ? len - first_off : 0, take = (size_t)data_len <= avail ? (size_t)data_len : avail;
      if ((rc = pl_append(&pl, bytes + first_off, take)) != ORC_OK) goto done;
      while (take < (size_t)data_len) { /* (cannot happen: the stale offset only ever points earlier) */
        static const uint8_t zero = 0;
        if ((rc = pl_append(&pl, &zero, 1)) != ORC_OK) goto done;
        take++;
      }
    }
    data_start += data_len;
    is_last = packet_index == 27 + page_buf[26];
    if (is_cntd) {
      if (is_last) {
        is_last = 0;
      } else {
        int pi = packet_index;
        (void)fwd_get_packet_length(page_buf, &pi);
        is_last = pi == 27 + page_buf[26];
      }
    }
    if (is_last) {
      memcpy(&granule_pos, page_buf + 6, 8);
      has_granule = 1;
      if ((page_buf[5] & 0x04) != 0) is_eos = 1; /* (_isEndOfStream && _pageQueue.Count == 0): never true before this page */
    } else {
      while (is_cntd && packet_index == 27 + page_buf[26]) {
        if (fwd_read_next_page(&pp, &page_buf, &is_resync, &data_start, &packet_index, &is_cont, &is_cntd) && !is_resync && is_cont) {
          int cont_sz = fwd_get_packet_length(page_buf, &packet_index);
          if ((rc = pl_append(&pl, page_buf + data_start, (size_t)cont_sz)) != ORC_OK) goto done;
          data_start += cont_sz;
        } else {
          break;
        }
      }
    }
    if (is_resync) pl.flags[pl.n] |= 2;
    if (has_granule) pl.granule[pl.n] = granule_pos;
    if (is_eos) pl.flags[pl.n] |= 1;
    pl.n++;
    pp.page_buf = page_buf;
    pp.data_start = data_start;
    pp.packet_index = packet_index;
    if (page_buf == NULL) pp.packet_index = 0x7fffffff;
  }
  if ((rc = pl_begin(&pl)) != ORC_OK) goto done;

done:
  free((void *)pp.queue_buf);
  free(pp.queue_resync);
  if (rc != ORC_OK) {
    free(pl.bytes);
    free(pl.offs);
    free(pl.granule);
    free(pl.flags);
    return rc;
  }
  if (!pl.bytes) pl.bytes = (uint8_t *)malloc(1);
  *out_bytes = pl.bytes;
  *out_offs = pl.offs;
  *out_granule = pl.granule;
  *out_flags = pl.flags;
  *out_n = pl.n;
  return ORC_OK;
}

void orc_free(void *p) { free(p); }
