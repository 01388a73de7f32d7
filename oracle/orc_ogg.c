/* oracle/orc_ogg.c -- minimal forward Ogg demux for the oracle (test infrastructure, see orc.h).
 *
 * Restates only what the synthesis path observes of the container:
 *   - page sync "OggS" + CRC-32 (poly 0x04c11db7, no reflection)      Ogg/PageReaderBase.cs:33-70,227-292, Ogg/Crc.cs:5-40
 *   - lacing -> packets, zero-length packets dropped                  Ogg/PageReader.cs:27-93
 *   - pages without packets dropped                                   Ogg/PageReader.cs:131
 *   - continuation across pages, granule attached to the packet that is LAST on the page it
 *     completes on, EOS on that packet of the EOS-flagged page        Ogg/PacketProvider.cs:324-438
 *   - pages after the EOS-flagged page ignored                        Ogg/StreamPageReader.cs:44-91
 * Only the first logical stream (serial of the first page) is demuxed.  Seeking, multiplexing and
 * the libvorbis granule-bug workaround are out of scope.
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

int orc_ogg_demux(const uint8_t *bytes, size_t len, uint8_t **out_bytes, int64_t **out_offs, int64_t **out_granule,
                  uint8_t **out_flags, int *out_n) {
  ogg_page *pages = NULL;
  int npages = 0, pcap = 0, have_serial = 0, has_all_pages = 0, rc = ORC_OK;
  int32_t serial = 0;
  size_t pos = 0;
  int resync = 0, i;
  /* StreamPageReader.AddPage state (Ogg/StreamPageReader.cs:44-91) */
  int32_t last_seq = 0;
  int have_first_data_page = 0;
  int64_t max_granule = 0;
  pkt_list pl;
  memset(&pl, 0, sizeof pl);
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
  for (i = 0; i < npages; i++) {
    free(pages[i].pk_off);
    free(pages[i].pk_len);
  }
  free(pages);
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
