/* oracle/orc_residue.c -- Residue0.cs / Residue1.cs / Residue2.cs restatement (test infrastructure, see orc.h). */
#include "orc_internal.h"

static int icount(int v) { /* Residue0.cs:10-19 */
  int ret = 0;
  while (v != 0) {
    ret += (v & 1);
    v = (int)((unsigned)v >> 1); /* values are non-negative here */
  }
  return ret;
}

/* Residue0.cs:35-117 (Residue2.Init :10-14 passes channels=1 to the base and keeps the real count) */
int orc_residue_init(orc_residue *r, int type, orc_packet *p, int channels, const orc_codebook *books, int nbooks) {
  int acc = 0, i, j, k, maxstage = 0;
  int book_nums[64 * 8];
  int entries, dim, partvals;
  memset(r, 0, sizeof *r);
  r->type = type;
  r->real_channels = channels;
  r->begin = (int)orc_read_bits(p, 24);
  r->end = (int)orc_read_bits(p, 24);
  r->partition_size = (int)orc_read_bits(p, 24) + 1;
  r->classifications = (int)orc_read_bits(p, 6) + 1;
  r->class_book = (int)orc_read_bits(p, 8);
  if (r->class_book >= nbooks) return ORC_ERR_RUNTIME;

  for (i = 0; i < r->classifications; i++) {
    int low_bits = (int)orc_read_bits(p, 3);
    if (orc_read_bit(p))
      r->cascade[i] = ((int)orc_read_bits(p, 5) << 3) | low_bits;
    else
      r->cascade[i] = low_bits;
    acc += icount(r->cascade[i]);
  }
  for (i = 0; i < acc; i++) {
    book_nums[i] = (int)orc_read_bits(p, 8);
    if (book_nums[i] >= nbooks) return ORC_ERR_RUNTIME;
    if (books[book_nums[i]].map_type == 0) return ORC_ERR_INVALID_DATA;
  }

  entries = books[r->class_book].entries;
  dim = books[r->class_book].dimensions;
  partvals = 1;
  while (dim > 0) {
    partvals *= r->classifications;
    if (partvals > entries) return ORC_ERR_INVALID_DATA;
    --dim;
  }

  acc = 0;
  for (j = 0; j < r->classifications; j++) {
    int stages = orc_ilog(r->cascade[j]);
    r->stages[j] = stages;
    for (k = 0; k < 8; k++) r->books[j][k] = -1;
    if (stages > 0) {
      if (stages > maxstage) maxstage = stages;
      for (k = 0; k < stages; k++) {
        if ((r->cascade[j] & (1 << k)) > 0) r->books[j][k] = book_nums[acc++];
      }
    }
  }
  r->max_stages = maxstage;

  dim = books[r->class_book].dimensions;
  r->partvals = partvals;
  r->decode_map = (int *)calloc((size_t)partvals * (size_t)(dim > 0 ? dim : 1), sizeof(int));
  if (!r->decode_map) return ORC_ERR_NOMEM;
  for (j = 0; j < partvals; j++) {
    int val = j;
    int mult = partvals / r->classifications;
    for (k = 0; k < dim; k++) {
      int deco;
      if (mult == 0) return ORC_ERR_RUNTIME; /* DivideByZeroException */
      deco = val / mult;
      val -= deco * mult;
      mult /= r->classifications;
      r->decode_map[j * dim + k] = deco;
    }
  }
  r->channels = (type == 2) ? 1 : channels;
  return ORC_OK;
}

void orc_residue_free(orc_residue *r) {
  free(r->decode_map);
  memset(r, 0, sizeof *r);
}

/* Test instrumentation (not part of the restated algorithm): which bins each cascade stage added to.
 * orc_coverage_begin(channels, plane_len) arms it for the calling thread; every residue add then ORs (1 << stage)
 * into mask[channel * plane_len + bin]; orc_coverage_end copies the mask out and disarms. */
static __thread unsigned char *g_cov_mask = NULL;
static __thread int g_cov_channels = 0, g_cov_len = 0, g_cov_stage = 0;

int orc_coverage_begin(int channels, int plane_len) {
  free(g_cov_mask);
  g_cov_mask = (unsigned char *)calloc((size_t)channels * (size_t)plane_len + 1, 1);
  if (!g_cov_mask) return ORC_ERR_NOMEM;
  g_cov_channels = channels;
  g_cov_len = plane_len;
  return ORC_OK;
}

int orc_coverage_end(unsigned char *mask_out) {
  if (!g_cov_mask) return ORC_ERR_ARGUMENT;
  if (mask_out) memcpy(mask_out, g_cov_mask, (size_t)g_cov_channels * (size_t)g_cov_len);
  free(g_cov_mask);
  g_cov_mask = NULL;
  return ORC_OK;
}

static void cov_mark(int channel, int offset) {
  if (g_cov_mask && channel >= 0 && channel < g_cov_channels && offset >= 0 && offset < g_cov_len)
    g_cov_mask[(size_t)channel * (size_t)g_cov_len + (size_t)offset] |= (unsigned char)(1u << (g_cov_stage & 7));
}

/* WriteVectors: Residue0.cs:180-201, Residue1.cs:8-26, Residue2.cs:23-47.
 * returns 1 = "bad packet, stop", 0 = ok, <0 = runtime fault */
static int write_vectors(const orc_residue *r, const orc_codebook *cb, orc_packet *p, float **residue, int buflen,
                         int channel, int offset, int partition_size) {
  int dims = cb->dimensions;
  if (r->type == 0) {
    float *res = residue[channel];
    int steps, i, dim, step;
    int *entry_cache;
    if (dims == 0) return ORC_ERR_RUNTIME;
    steps = partition_size / dims;
    entry_cache = (int *)malloc(sizeof(int) * (size_t)(steps > 0 ? steps : 1));
    if (!entry_cache) return ORC_ERR_NOMEM;
    for (i = 0; i < steps; i++) {
      int e = orc_decode_scalar(cb, p);
      if (e == -2) {
        free(entry_cache);
        return ORC_ERR_RUNTIME;
      }
      if ((entry_cache[i] = e) == -1) {
        free(entry_cache);
        return 1;
      }
    }
    for (dim = 0; dim < dims; dim++) {
      for (step = 0; step < steps; step++, offset++) {
        if (offset < 0 || offset >= buflen) {
          free(entry_cache);
          return ORC_ERR_RUNTIME;
        }
        res[offset] += cb->lookup[entry_cache[step] * dims + dim];
        cov_mark(channel, offset);
      }
    }
    free(entry_cache);
    return 0;
  } else if (r->type == 1) {
    float *res = residue[channel];
    int i, j;
    for (i = 0; i < partition_size;) {
      int entry = orc_decode_scalar(cb, p);
      if (entry == -2) return ORC_ERR_RUNTIME;
      if (entry == -1) return 1;
      if (dims == 0) return ORC_ERR_RUNTIME; /* would spin forever in the reference */
      for (j = 0; j < dims; i++, j++) {
        if (offset + i < 0 || offset + i >= buflen) return ORC_ERR_RUNTIME;
        res[offset + i] += cb->lookup[entry * dims + j];
        cov_mark(channel, offset + i);
      }
    }
    return 0;
  } else {
    int ch_ptr = 0, c, d;
    offset /= r->real_channels; /* Residue2.cs:27 (quirk B-1) */
    for (c = 0; c < partition_size;) {
      int entry = orc_decode_scalar(cb, p);
      if (entry == -2) return ORC_ERR_RUNTIME;
      if (entry == -1) return 1;
      if (dims == 0) return ORC_ERR_RUNTIME;
      for (d = 0; d < dims; d++, c++) {
        if (offset < 0 || offset >= buflen) return ORC_ERR_RUNTIME;
        residue[ch_ptr][offset] += cb->lookup[entry * dims + d];
        cov_mark(ch_ptr, offset);
        if (++ch_ptr == r->real_channels) {
          ch_ptr = 0;
          offset++;
        }
      }
    }
    return 0;
  }
}

/* Residue0.cs:119-178 (Residue2.Decode :16-21 multiplies blockSize by the channel count first) */
int orc_residue_decode(const orc_residue *r, const orc_codebook *books, orc_packet *p, const int *do_not_decode,
                       int nflags, int block_size, float **buffer, int buflen) {
  const orc_codebook *class_book = &books[r->class_book];
  int end, n, any = 0, i;
  if (r->type == 2) block_size *= r->real_channels;
  end = r->end < block_size / 2 ? r->end : block_size / 2;
  n = end - r->begin;
  for (i = 0; i < nflags; i++)
    if (!do_not_decode[i]) any = 1;

  if (n > 0 && any) {
    int partition_count = n / r->partition_size;
    int cdim = class_book->dimensions;
    int partition_words, stage;
    const int **part_word_cache;
    if (cdim == 0) return ORC_ERR_RUNTIME; /* DivideByZeroException */
    partition_words = (partition_count + cdim - 1) / cdim;
    part_word_cache = (const int **)calloc((size_t)r->channels * (size_t)(partition_words > 0 ? partition_words : 1),
                                           sizeof(int *));
    if (!part_word_cache) return ORC_ERR_NOMEM;

    for (stage = 0; stage < r->max_stages; stage++) {
      int partition_idx, entry_idx;
      g_cov_stage = stage;
      for (partition_idx = 0, entry_idx = 0; partition_idx < partition_count; entry_idx++) {
        int dimension_idx, ch;
        if (stage == 0) {
          for (ch = 0; ch < r->channels; ch++) {
            int idx = orc_decode_scalar(class_book, p);
            if (idx == -2) {
              free(part_word_cache);
              return ORC_ERR_RUNTIME;
            }
            if (idx >= 0 && idx < r->partvals) {
              part_word_cache[ch * partition_words + entry_idx] = &r->decode_map[idx * cdim];
            } else {
              partition_idx = partition_count;
              stage = r->max_stages;
              break;
            }
          }
        }
        for (dimension_idx = 0; partition_idx < partition_count && dimension_idx < cdim;
             dimension_idx++, partition_idx++) {
          int offset = r->begin + partition_idx * r->partition_size;
          for (ch = 0; ch < r->channels; ch++) {
            int idx = part_word_cache[ch * partition_words + entry_idx][dimension_idx];
            if ((r->cascade[idx] & (1 << stage)) != 0) {
              int book = r->books[idx][stage];
              if (book >= 0) {
                int rc = write_vectors(r, &books[book], p, buffer, buflen, ch, offset, r->partition_size);
                if (rc < 0) {
                  free(part_word_cache);
                  return rc;
                }
                if (rc) {
                  partition_idx = partition_count;
                  stage = r->max_stages;
                  break;
                }
              }
            }
          }
        }
      }
    }
    free(part_word_cache);
  }
  return ORC_OK;
}
