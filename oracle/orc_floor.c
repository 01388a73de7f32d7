/* oracle/orc_floor.c -- Floor0.cs + Floor1.cs restatement (test infrastructure, see orc.h). */
#include "orc_internal.h"

static const float k_inverse_db[256] = {
#include "floor1_db_table.inc"
};

float orc_inverse_db(int i) { return k_inverse_db[i]; }

/* ======================= Floor1 ======================= */

static const int k_range_lookup[4] = {256, 128, 86, 64}; /* Floor1.cs:27 */
static const int k_ybits_lookup[4] = {8, 7, 7, 6};       /* Floor1.cs:28 */

/* Floor1.cs:30-133 */
static int floor1_init(orc_floor1 *f, orc_packet *p, int nbooks) {
  int maximum_class = -1, i, j;
  int range_bits;
  memset(f, 0, sizeof *f);
  f->partition_count = (int)orc_read_bits(p, 5);
  for (i = 0; i < f->partition_count; i++) {
    f->partition_class[i] = (int)orc_read_bits(p, 4);
    if (f->partition_class[i] > maximum_class) maximum_class = f->partition_class[i];
  }
  ++maximum_class;
  f->class_count = maximum_class;
  for (i = 0; i < maximum_class; i++) {
    int nsub;
    f->class_dimensions[i] = (int)orc_read_bits(p, 3) + 1;
    f->class_subclasses[i] = (int)orc_read_bits(p, 2);
    f->class_masterbook[i] = -1;
    if (f->class_subclasses[i] > 0) {
      f->class_masterbook[i] = (int)orc_read_bits(p, 8);
      if (f->class_masterbook[i] >= nbooks) return ORC_ERR_RUNTIME; /* codebooks[...] out of range */
    }
    nsub = 1 << f->class_subclasses[i];
    for (j = 0; j < 8; j++) f->subclass_book[i][j] = -1;
    for (j = 0; j < nsub; j++) {
      int book_num = (int)orc_read_bits(p, 8) - 1;
      if (book_num >= nbooks) return ORC_ERR_RUNTIME;
      f->subclass_book[i][j] = book_num;
    }
  }

  f->multiplier = (int)orc_read_bits(p, 2);
  f->range = k_range_lookup[f->multiplier];
  f->y_bits = k_ybits_lookup[f->multiplier];
  ++f->multiplier; /* Floor1.cs:74 -- header field plus one (quirk B-8) */

  range_bits = (int)orc_read_bits(p, 4);
  f->x_count = 0;
  f->x_list[f->x_count++] = 0;
  f->x_list[f->x_count++] = 1 << range_bits;
  for (i = 0; i < f->partition_count; i++) {
    int class_num = f->partition_class[i];
    for (j = 0; j < f->class_dimensions[class_num]; j++) {
      if (f->x_count >= 256) return ORC_ERR_RUNTIME;
      f->x_list[f->x_count++] = (int)orc_read_bits(p, range_bits);
    }
  }

  /* neighbours + sort table (Floor1.cs:93-115) */
  f->sort_idx[0] = 0;
  f->sort_idx[1] = 1;
  for (i = 2; i < f->x_count; i++) {
    f->l_neigh[i] = 0;
    f->h_neigh[i] = 1;
    f->sort_idx[i] = i;
    for (j = 2; j < i; j++) {
      int temp = f->x_list[j];
      if (temp < f->x_list[i]) {
        if (temp > f->x_list[f->l_neigh[i]]) f->l_neigh[i] = j;
      } else {
        if (temp < f->x_list[f->h_neigh[i]]) f->h_neigh[i] = j;
      }
    }
  }
  /* Floor1.cs:118-132 */
  for (i = 0; i < f->x_count - 1; i++) {
    for (j = i + 1; j < f->x_count; j++) {
      if (f->x_list[i] == f->x_list[j]) return ORC_ERR_INVALID_DATA;
      if (f->x_list[f->sort_idx[i]] > f->x_list[f->sort_idx[j]]) {
        int temp = f->sort_idx[i];
        f->sort_idx[i] = f->sort_idx[j];
        f->sort_idx[j] = temp;
      }
    }
  }
  return ORC_OK;
}

/* Floor1.cs:135-184 */
static int floor1_unpack(const orc_floor1 *f, const orc_codebook *books, orc_packet *p, orc_floor_data *data) {
  int i, j;
  data->post_count = 0;
  if (orc_read_bit(p)) {
    int post_count = 2;
    data->posts[0] = (int)orc_read_bits(p, f->y_bits);
    data->posts[1] = (int)orc_read_bits(p, f->y_bits);

    for (i = 0; i < f->partition_count; i++) {
      int cls_num = f->partition_class[i];
      int cdim = f->class_dimensions[cls_num];
      int cbits = f->class_subclasses[cls_num];
      int csub = (1 << cbits) - 1;
      uint32_t cval = 0;
      if (cbits > 0) {
        int r = orc_decode_scalar(&books[f->class_masterbook[cls_num]], p);
        if (r == -2) return ORC_ERR_RUNTIME;
        cval = (uint32_t)r;
        if (cval == 0xFFFFFFFFu) {
          post_count = 0;
          break;
        }
      }
      for (j = 0; j < cdim; j++) {
        int book = f->subclass_book[cls_num][cval & (uint32_t)csub];
        cval >>= cbits;
        if (book >= 0) {
          int r;
          if (post_count >= 64) return ORC_ERR_RUNTIME; /* Posts = new int[64] (Floor1.cs:12) */
          r = orc_decode_scalar(&books[book], p);
          if (r == -2) return ORC_ERR_RUNTIME;
          if ((data->posts[post_count] = r) == -1) {
            post_count = 0;
            i = f->partition_count;
            break;
          }
        }
        ++post_count;
      }
    }
    data->post_count = post_count;
  }
  return ORC_OK;
}

/* Floor1.cs:299-314 */
int orc_render_point(int x0, int y0, int x1, int y1, int X) {
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = abs(dy);
  int err = ady * (X - x0);
  int off = err / adx;
  if (dy < 0) return y0 - off;
  return y0 + off;
}

/* Floor1.cs:316-341; returns 0 ok, <0 when the reference would index inverse_dB_table / v out of range */
int orc_render_line_multi(int x0, int y0, int x1, int y1, float *v, int vlen) {
  int dy = y1 - y0;
  int adx = x1 - x0;
  int ady = abs(dy);
  int sy = 1 - (((dy >> 31) & 1) * 2);
  int b, x = x0, y = y0, err;
  if (adx == 0) return ORC_ERR_RUNTIME; /* DivideByZeroException */
  b = dy / adx;
  err = -adx;

  if (y0 < 0 || y0 > 255 || x0 < 0 || x0 >= vlen) return ORC_ERR_RUNTIME;
  v[x0] *= k_inverse_db[y0];
  ady -= abs(b) * adx;

  while (++x < x1) {
    y += b;
    err += ady;
    if (err >= 0) {
      err -= adx;
      y += sy;
    }
    if (y < 0 || y > 255 || x >= vlen) return ORC_ERR_RUNTIME;
    v[x] *= k_inverse_db[y];
  }
  return ORC_OK;
}

/* Floor1.cs:224-297; writes step flags, rewrites data->posts with final Y */
static void floor1_unwrap_posts(const orc_floor1 *f, orc_floor_data *data, int *step_flags /*[256]*/) {
  int final_y[256];
  int i;
  memset(step_flags, 0, sizeof(int) * 256);
  step_flags[0] = 1;
  step_flags[1] = 1;
  final_y[0] = data->posts[0];
  final_y[1] = data->posts[1];

  for (i = 2; i < data->post_count; i++) {
    int low_ofs = f->l_neigh[i];
    int high_ofs = f->h_neigh[i];
    int predicted =
        orc_render_point(f->x_list[low_ofs], final_y[low_ofs], f->x_list[high_ofs], final_y[high_ofs], f->x_list[i]);
    int val = data->posts[i];
    int highroom = f->range - predicted;
    int lowroom = predicted;
    int room;
    if (highroom < lowroom)
      room = highroom * 2;
    else
      room = lowroom * 2;
    if (val != 0) {
      step_flags[low_ofs] = 1;
      step_flags[high_ofs] = 1;
      step_flags[i] = 1;
      if (val >= room) {
        if (highroom > lowroom)
          final_y[i] = val - lowroom + predicted;
        else
          final_y[i] = predicted - val + highroom - 1;
      } else {
        if ((val % 2) == 1)
          final_y[i] = predicted - ((val + 1) / 2);
        else
          final_y[i] = predicted + (val / 2);
      }
    } else {
      step_flags[i] = 0;
      final_y[i] = predicted;
    }
  }
  for (i = 0; i < data->post_count; i++) data->posts[i] = final_y[i];
}

/* Floor1.cs:186-222 */
static int floor1_apply(const orc_floor1 *f, orc_floor_data *data, int block_size, float *residue, int reslen) {
  int n = block_size / 2;
  if (data->post_count > 0) {
    int step_flags[256];
    int lx = 0, ly, i, rc;
    floor1_unwrap_posts(f, data, step_flags);
    ly = data->posts[0] * f->multiplier;
    for (i = 1; i < data->post_count; i++) {
      int idx = f->sort_idx[i];
      if (step_flags[idx]) {
        int hx = f->x_list[idx];
        int hy = data->posts[idx] * f->multiplier;
        if (lx < n) {
          rc = orc_render_line_multi(lx, ly, hx < n ? hx : n, hy, residue, reslen);
          if (rc) return rc;
        }
        lx = hx;
        ly = hy;
      }
      if (lx >= n) break;
    }
    if (lx < n) {
      rc = orc_render_line_multi(lx, ly, n, ly, residue, reslen);
      if (rc) return rc;
    }
  } else {
    memset(residue, 0, sizeof(float) * (size_t)n);
  }
  return ORC_OK;
}

/* ======================= Floor0 ======================= */

/* Floor0.cs:81-84 */
static float to_bark(double lsp) {
  return (float)(13.1 * atan(0.00074 * lsp) + 2.24 * atan(0.0000000185 * lsp * lsp) + .0001 * lsp);
}

/* Floor0.cs:67-79 */
static int *synthesize_bark_curve(const orc_floor0 *f, int n) {
  float scale = (float)f->bark_map_size / to_bark((double)(f->rate / 2));
  int *map = (int *)calloc((size_t)n + 1, sizeof(int));
  int i;
  if (!map) return NULL;
  for (i = 0; i < n - 1; i++) {
    float hz = (float)((float)((float)f->rate / 2.0f) / (float)n) * (float)i; /* (_rate / 2f) / n * i */
    float t = (float)(to_bark((double)hz) * scale);
    int v = (int)floor((double)t);
    map[i] = (f->bark_map_size - 1) < v ? (f->bark_map_size - 1) : v;
  }
  map[n] = -1;
  return map;
}

/* Floor0.cs:86-96 */
static float *synthesize_wdel_map(const orc_floor0 *f, int n) {
  float wdel = (float)(3.14159265358979323846 / f->bark_map_size);
  float *map = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
  int i;
  if (!map) return NULL;
  for (i = 0; i < n; i++) map[i] = 2.0f * (float)cos((double)(float)(wdel * (float)i));
  return map;
}

/* Floor0.cs:28-65 */
static int floor0_init(orc_floor0 *f, orc_packet *p, int block0, int block1, const orc_codebook *books, int nbooks) {
  int i;
  memset(f, 0, sizeof *f);
  f->order = (int)orc_read_bits(p, 8);
  f->rate = (int)orc_read_bits(p, 16);
  f->bark_map_size = (int)orc_read_bits(p, 16);
  f->amp_bits = (int)orc_read_bits(p, 6);
  f->amp_ofs = (int)orc_read_bits(p, 8);
  f->book_count = (int)orc_read_bits(p, 4) + 1;
  if (f->order < 1 || f->rate < 1 || f->bark_map_size < 1 || f->book_count == 0) return ORC_ERR_INVALID_DATA;
  f->amp_div = (int)((1u << (f->amp_bits & 31)) - 1u);
  for (i = 0; i < f->book_count; i++) {
    int num = (int)orc_read_bits(p, 8);
    if (num < 0 || num >= nbooks) return ORC_ERR_INVALID_DATA;
    if (books[num].map_type == 0 || books[num].dimensions < 1) return ORC_ERR_INVALID_DATA;
    f->books[i] = num;
  }
  f->book_bits = orc_ilog(f->book_count);
  f->block_size[0] = block0;
  f->block_size[1] = block1;
  f->bark_map[0] = synthesize_bark_curve(f, block0 / 2);
  f->bark_map[1] = synthesize_bark_curve(f, block1 / 2);
  f->w_map[0] = synthesize_wdel_map(f, block0 / 2);
  f->w_map[1] = synthesize_wdel_map(f, block1 / 2);
  if (!f->bark_map[0] || !f->bark_map[1] || !f->w_map[0] || !f->w_map[1]) return ORC_ERR_NOMEM;
  return ORC_OK;
}

/* Floor0.cs:98-150 */
static int floor0_unpack(const orc_floor0 *f, const orc_codebook *books, orc_packet *p, orc_floor_data *data) {
  int i, j, k;
  memset(data->coeff, 0, sizeof(float) * (size_t)(f->order + 1));
  data->amp = (float)orc_read_bits(p, f->amp_bits); /* ulong -> float */
  if (data->amp > 0.0f) {
    uint32_t book_num;
    const orc_codebook *book;
    float last;
    data->amp = data->amp / (float)f->amp_div * (float)f->amp_ofs;
    book_num = (uint32_t)orc_read_bits(p, f->book_bits);
    if (book_num >= (uint32_t)f->book_count) {
      data->amp = 0;
      return ORC_OK;
    }
    book = &books[f->books[book_num]];
    for (i = 0; i < f->order;) {
      int entry = orc_decode_scalar(book, p);
      if (entry == -2) return ORC_ERR_RUNTIME;
      if (entry == -1) {
        data->amp = 0;
        return ORC_OK;
      }
      for (j = 0; i < f->order && j < book->dimensions; j++, i++) data->coeff[i] = book->lookup[entry * book->dimensions + j];
    }
    last = 0.0f;
    for (j = 0; j < f->order;) {
      for (k = 0; j < f->order && k < book->dimensions; j++, k++) data->coeff[j] += last;
      last = data->coeff[j - 1];
    }
  }
  return ORC_OK;
}

/* Floor0.cs:152-212 */
static int floor0_apply(const orc_floor0 *f, orc_floor_data *data, int block_size, float *residue, int reslen) {
  int n = block_size / 2;
  (void)reslen;
  if (data->amp > 0.0f) {
    int which = (block_size == f->block_size[0]) ? 0 : 1; /* Dictionary lookup by blockSize; block0==block1 -> same maps */
    const int *bark_map = f->bark_map[which];
    const float *w_map = f->w_map[which];
    int i, j;
    for (i = 0; i < f->order; i++) data->coeff[i] = 2.0f * (float)cos((double)data->coeff[i]);

    i = 0;
    while (i < n) {
      int k = bark_map[i];
      float pp = .5f, q = .5f, w;
      if (k < 0 || k >= n) return ORC_ERR_RUNTIME; /* wMap is sized n (Floor0.cs:90) */
      w = w_map[k];
      for (j = 1; j < f->order; j += 2) {
        q = q * (float)(w - data->coeff[j - 1]);
        pp = pp * (float)(w - data->coeff[j]);
      }
      if (j == f->order) {
        /* odd order filter */
        q = q * (float)(w - data->coeff[j - 1]);
        pp = pp * (float)(pp * (float)(4.0f - (float)(w * w)));
        q = q * q;
      } else {
        /* even order filter */
        pp = pp * (float)(pp * (float)(2.0f - w));
        q = q * (float)(q * (float)(2.0f + w));
      }
      q = (float)(data->amp / (float)sqrt((double)(float)(pp + q))) - (float)f->amp_ofs;
      q = (float)exp((double)(float)(q * 0.11512925f));
      residue[i] *= q;
      while (bark_map[++i] == k) residue[i] *= q;
    }
  } else {
    memset(residue, 0, sizeof(float) * (size_t)n);
  }
  return ORC_OK;
}

/* ======================= dispatch ======================= */

int orc_floor_init(orc_floor *f, int type, orc_packet *p, int channels, int block0, int block1,
                   const orc_codebook *books, int nbooks) {
  (void)channels;
  memset(f, 0, sizeof *f);
  f->type = type;
  if (type == 0) return floor0_init(&f->f0, p, block0, block1, books, nbooks);
  return floor1_init(&f->f1, p, nbooks);
}

void orc_floor_free(orc_floor *f) {
  if (f->type == 0) {
    if (f->f0.bark_map[0]) free(f->f0.bark_map[0]);
    if (f->f0.bark_map[1]) free(f->f0.bark_map[1]);
    free(f->f0.w_map[0]);
    free(f->f0.w_map[1]);
  }
  memset(f, 0, sizeof *f);
}

int orc_floor_unpack(const orc_floor *f, const orc_codebook *books, orc_packet *p, int block_size,
                     orc_floor_data *out) {
  (void)block_size;
  memset(out, 0, sizeof *out);
  out->type = f->type;
  if (f->type == 0) return floor0_unpack(&f->f0, books, p, out);
  return floor1_unpack(&f->f1, books, p, out);
}

int orc_floor_execute_channel(const orc_floor_data *d) {
  /* Floor1.cs:15, Floor0.cs:16 */
  int energy = d->type == 0 ? (d->amp > 0.0f) : (d->post_count > 0);
  return (d->force_energy || energy) && !d->force_no_energy;
}

int orc_floor_apply(const orc_floor *f, orc_floor_data *d, int block_size, float *residue, int reslen) {
  if (d->type != f->type) return ORC_ERR_ARGUMENT; /* ArgumentException "Incorrect packet data!" */
  if (f->type == 0) return floor0_apply(&f->f0, d, block_size, residue, reslen);
  return floor1_apply(&f->f1, d, block_size, residue, reslen);
}
