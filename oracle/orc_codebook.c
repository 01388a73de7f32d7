/* oracle/orc_codebook.c -- Huffman.cs + Codebook.cs restatement (test infrastructure, see orc.h). */
#include "orc_internal.h"

/* Huffman.cs:78-86 comparer: by Length, then Bits (the reference subtracts ints; for the codes
 * that matter -- distinct (length,bits) pairs of a prefix code -- any consistent order gives the
 * same decode because at most one node can match a given bit string). */
static int node_cmp(const void *a, const void *b) {
  const orc_huff_node *x = (const orc_huff_node *)a, *y = (const orc_huff_node *)b;
  if (x->length != y->length) return x->length < y->length ? -1 : 1;
  if (x->bits != y->bits) return x->bits < y->bits ? -1 : 1;
  return 0;
}

/* Huffman.cs:15-76 GenerateTable */
static int generate_table(orc_codebook *cb, const int *values /* NULL => FastRange(0,n) */, const int *length_list,
                          const int *code_list, int n) {
  const int MAX_TABLE_BITS = 10; /* Huffman.cs:9 */
  orc_huff_node *list = (orc_huff_node *)calloc((size_t)(n > 0 ? n : 1), sizeof *list);
  int max_len = 0, i, table_bits;
  if (!list) return ORC_ERR_NOMEM;
  for (i = 0; i < n; i++) {
    list[i].value = values ? values[i] : i;
    list[i].length = length_list[i] <= 0 ? 99999 : length_list[i];
    list[i].bits = code_list[i];
    list[i].mask = (int)((1u << (length_list[i] & 31)) - 1u); /* C# int shift masks the count */
    list[i].present = 1;
    if (length_list[i] > 0 && max_len < length_list[i]) max_len = length_list[i];
  }
  qsort(list, (size_t)n, sizeof *list, node_cmp);

  table_bits = max_len > MAX_TABLE_BITS ? MAX_TABLE_BITS : max_len;
  cb->prefix_count = 1 << table_bits;
  cb->prefix = (orc_huff_node *)calloc((size_t)cb->prefix_count, sizeof *cb->prefix);
  cb->overflow = NULL;
  cb->overflow_count = 0;
  if (!cb->prefix) {
    free(list);
    return ORC_ERR_NOMEM;
  }
  for (i = 0; i < n && list[i].length < 99999; i++) {
    int item_bits = list[i].length;
    if (item_bits > table_bits) {
      int cnt = 0, j;
      for (j = i; j < n && list[j].length < 99999; j++) cnt++;
      cb->overflow = (orc_huff_node *)calloc((size_t)(cnt > 0 ? cnt : 1), sizeof *cb->overflow);
      if (!cb->overflow) {
        free(list);
        return ORC_ERR_NOMEM;
      }
      for (; i < n && list[i].length < 99999; i++) cb->overflow[cb->overflow_count++] = list[i];
    } else {
      int max_val = 1 << (table_bits - item_bits), j;
      for (j = 0; j < max_val; j++) {
        int idx = (j << item_bits) | list[i].bits;
        if (idx >= 0 && idx < cb->prefix_count) cb->prefix[idx] = list[i];
      }
    }
  }
  cb->prefix_bits = table_bits;
  free(list);
  return ORC_OK;
}

/* Codebook.cs:172-220 ComputeCodewords + AddEntry */
static int compute_codewords(int sparse, int *codewords, int *codeword_lengths, const int *len, int n, int *values) {
  int i, k, m = 0;
  uint32_t available[33];
  memset(available, 0, sizeof available);
  for (k = 0; k < n; ++k)
    if (len[k] > 0) break;
  if (k == n) return 1;

#define ADD_ENTRY(code, symbol, count, l)      \
  do {                                         \
    if (sparse) {                              \
      codewords[count] = (int)(code);          \
      codeword_lengths[count] = (l);           \
      values[count] = (symbol);                \
    } else {                                   \
      codewords[symbol] = (int)(code);         \
    }                                          \
  } while (0)

  ADD_ENTRY(0u, k, m, len[k]);
  m++;
  if (len[k] > 31) return -1; /* available[32] is out of range in the reference (uint[32]) */
  for (i = 1; i <= len[k]; ++i) available[i] = 1u << (32 - i);

  for (i = k + 1; i < n; ++i) {
    uint32_t res;
    int z = len[i], y;
    if (z <= 0) continue;
    if (z > 31) return -1;
    while (z > 0 && available[z] == 0) --z;
    if (z == 0) return 0;
    res = available[z];
    available[z] = 0;
    ADD_ENTRY(orc_bit_reverse(res, 32), i, m, len[i]);
    m++;
    if (z != len[i]) {
      for (y = len[i]; y > z; --y) available[y] = res + (1u << (32 - y));
    }
  }
#undef ADD_ENTRY
  return 1;
}

/* Codebook.cs:76-170 InitTree */
static int init_tree(orc_codebook *cb, orc_packet *p) {
  int sparse, total = 0, max_len, i;
  int entries = cb->entries;
  if (orc_read_bit(p)) {
    /* ordered */
    int len = (int)orc_read_bits(p, 5) + 1;
    for (i = 0; i < entries;) {
      int cnt = (int)orc_read_bits(p, orc_ilog(entries - i));
      while (--cnt >= 0) {
        if (i >= entries) return ORC_ERR_RUNTIME; /* IndexOutOfRangeException */
        cb->lengths[i++] = len;
      }
      ++len;
    }
    total = 0;
    sparse = 0;
    max_len = len;
  } else {
    /* unordered */
    max_len = -1;
    sparse = orc_read_bit(p);
    for (i = 0; i < entries; i++) {
      if (!sparse || orc_read_bit(p)) {
        cb->lengths[i] = (int)orc_read_bits(p, 5) + 1;
        ++total;
      } else {
        cb->lengths[i] = -1;
      }
      if (cb->lengths[i] > max_len) max_len = cb->lengths[i];
    }
  }

  cb->max_bits = max_len;
  if (max_len > -1) {
    int *codeword_lengths = NULL, *values = NULL, *codewords = NULL;
    int sorted_count, rc, ncodes;
    if (sparse && total >= (entries >> 2)) {
      codeword_lengths = (int *)malloc(sizeof(int) * (size_t)(entries > 0 ? entries : 1));
      if (!codeword_lengths) return ORC_ERR_NOMEM;
      memcpy(codeword_lengths, cb->lengths, sizeof(int) * (size_t)entries);
      sparse = 0;
    }
    sorted_count = sparse ? total : 0;
    if (!sparse) {
      codewords = (int *)calloc((size_t)(entries > 0 ? entries : 1), sizeof(int));
      ncodes = entries;
    } else {
      codeword_lengths = (int *)calloc((size_t)(sorted_count > 0 ? sorted_count : 1), sizeof(int));
      codewords = (int *)calloc((size_t)(sorted_count > 0 ? sorted_count : 1), sizeof(int));
      values = (int *)calloc((size_t)(sorted_count > 0 ? sorted_count : 1), sizeof(int));
      ncodes = sorted_count;
    }
    if (!codewords) return ORC_ERR_NOMEM;
    rc = compute_codewords(sparse, codewords, codeword_lengths, cb->lengths, entries, values);
    if (rc <= 0) {
      free(codeword_lengths);
      free(codewords);
      free(values);
      return rc < 0 ? ORC_ERR_RUNTIME : ORC_ERR_INVALID_DATA; /* Codebook.cs:161 */
    }
    rc = generate_table(cb, values, codeword_lengths ? codeword_lengths : cb->lengths, codewords, ncodes);
    free(codeword_lengths);
    free(codewords);
    free(values);
    if (rc != ORC_OK) return rc;
  }
  return ORC_OK;
}

/* Codebook.cs:285-292 */
static int lookup1_values(int entries, int dimensions) {
  int r = (int)floor(exp(log((double)entries) / dimensions));
  if (floor(pow((double)(r + 1), (double)dimensions)) <= entries) ++r;
  return r;
}

/* Codebook.cs:222-283 InitLookupTable */
static int init_lookup_table(orc_codebook *cb, orc_packet *p) {
  float min_value, delta_value;
  int value_bits, sequence_p, lookup_value_count, i, idx;
  uint32_t *multiplicands;
  float *table;
  cb->map_type = (int)orc_read_bits(p, 4);
  if (cb->map_type == 0) return ORC_OK;

  min_value = orc_convert_from_vorbis_float32((uint32_t)orc_read_bits(p, 32));
  delta_value = orc_convert_from_vorbis_float32((uint32_t)orc_read_bits(p, 32));
  value_bits = (int)orc_read_bits(p, 4) + 1;
  sequence_p = orc_read_bit(p);

  lookup_value_count = cb->entries * cb->dimensions;
  table = (float *)calloc((size_t)(lookup_value_count > 0 ? lookup_value_count : 1), sizeof(float));
  if (!table) return ORC_ERR_NOMEM;
  if (cb->map_type == 1) {
    if (cb->dimensions == 0 || cb->entries == 0) { /* DivideByZero / log(0) paths in the reference */
      free(table);
      return ORC_ERR_RUNTIME;
    }
    lookup_value_count = lookup1_values(cb->entries, cb->dimensions);
  }
  multiplicands = (uint32_t *)calloc((size_t)(lookup_value_count > 0 ? lookup_value_count : 1), sizeof(uint32_t));
  if (!multiplicands) {
    free(table);
    return ORC_ERR_NOMEM;
  }
  for (i = 0; i < lookup_value_count; i++) multiplicands[i] = (uint32_t)orc_read_bits(p, value_bits);

  if (cb->map_type == 1) {
    if (lookup_value_count <= 0) {
      free(table);
      free(multiplicands);
      return ORC_ERR_RUNTIME;
    }
    for (idx = 0; idx < cb->entries; idx++) {
      double last = 0.0;
      int idx_div = 1; /* C# int: wraps silently on overflow (unchecked) */
      for (i = 0; i < cb->dimensions; i++) {
        int moff;
        double value;
        if (idx_div == 0) {
          free(table);
          free(multiplicands);
          return ORC_ERR_RUNTIME; /* DivideByZeroException */
        }
        moff = (idx / idx_div) % lookup_value_count;
        /* (float)mult * delta + min are float ops; + last promotes to double (Codebook.cs:255) */
        value = (double)((float)((float)multiplicands[moff] * delta_value) + min_value) + last;
        table[idx * cb->dimensions + i] = (float)value;
        if (sequence_p) last = value;
        idx_div = (int)((uint32_t)idx_div * (uint32_t)lookup_value_count);
      }
    }
  } else {
    for (idx = 0; idx < cb->entries; idx++) {
      double last = 0.0;
      int moff = idx * cb->dimensions;
      for (i = 0; i < cb->dimensions; i++) {
        /* uint * float -> float (Codebook.cs:272) */
        double value = (double)((float)((float)multiplicands[moff] * delta_value) + min_value) + last;
        table[idx * cb->dimensions + i] = (float)value;
        if (sequence_p) last = value;
        ++moff;
      }
    }
  }
  free(multiplicands);
  cb->lookup = table;
  return ORC_OK;
}

/* Codebook.cs:59-74 Init */
int orc_codebook_init(orc_codebook *cb, orc_packet *p) {
  int rc;
  memset(cb, 0, sizeof *cb);
  if (orc_read_bits(p, 24) != 0x564342ull) return ORC_ERR_INVALID_DATA;
  cb->dimensions = (int)orc_read_bits(p, 16);
  cb->entries = (int)orc_read_bits(p, 24);
  cb->lengths = (int *)calloc((size_t)(cb->entries > 0 ? cb->entries : 1), sizeof(int));
  if (!cb->lengths) return ORC_ERR_NOMEM;
  rc = init_tree(cb, p);
  if (rc != ORC_OK) return rc;
  return init_lookup_table(cb, p);
}

void orc_codebook_free(orc_codebook *cb) {
  free(cb->lengths);
  free(cb->lookup);
  free(cb->prefix);
  free(cb->overflow);
  memset(cb, 0, sizeof *cb);
}

/* Codebook.cs:294-320 DecodeScalar */
int orc_decode_scalar(const orc_codebook *cb, orc_packet *p) {
  int bits_read, i;
  int data = (int)orc_try_peek_bits(p, cb->prefix_bits, &bits_read);
  if (bits_read == 0) return -1;
  if (!cb->prefix) return -2; /* NullReferenceException in the reference (tree never built) */

  if (cb->prefix[data].present) {
    orc_skip_bits(p, cb->prefix[data].length);
    return cb->prefix[data].value;
  }

  data = (int)orc_try_peek_bits(p, cb->max_bits, &bits_read);
  if (!cb->overflow) return -2; /* _overflowList == null -> NullReferenceException */
  for (i = 0; i < cb->overflow_count; i++) {
    const orc_huff_node *node = &cb->overflow[i];
    if (node->bits == (data & node->mask)) {
      orc_skip_bits(p, node->length);
      return node->value;
    }
  }
  return -1;
}
