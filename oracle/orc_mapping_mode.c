/* oracle/orc_mapping_mode.c -- Mapping.cs + Mode.cs restatement (test infrastructure, see orc.h). */
#include "orc_internal.h"

/* ======================= Mapping ======================= */

/* Mapping.cs:16-93 */
int orc_mapping_init(orc_mapping *m, orc_packet *p, int channels, int nfloors, int nresidues) {
  int submap_count = 1, coupling_steps = 0, coupling_bits, j, c;
  int mux[256];
  memset(m, 0, sizeof *m);
  if (orc_read_bit(p)) submap_count += (int)orc_read_bits(p, 4);
  if (orc_read_bit(p)) coupling_steps = (int)orc_read_bits(p, 8) + 1;

  coupling_bits = orc_ilog(channels - 1);
  m->coupling_steps = coupling_steps;
  for (j = 0; j < coupling_steps; j++) {
    int magnitude = (int)orc_read_bits(p, coupling_bits);
    int angle = (int)orc_read_bits(p, coupling_bits);
    if (magnitude == angle || magnitude > channels - 1 || angle > channels - 1) return ORC_ERR_INVALID_DATA;
    m->coupling_angle[j] = angle;
    m->coupling_magnitude[j] = magnitude;
  }
  if (0 != orc_read_bits(p, 2)) return ORC_ERR_INVALID_DATA;

  memset(mux, 0, sizeof mux);
  if (submap_count > 1) {
    for (c = 0; c < channels; c++) {
      mux[c] = (int)orc_read_bits(p, 4);
      if (mux[c] > submap_count) return ORC_ERR_INVALID_DATA; /* sic: '>' (Mapping.cs:53) */
    }
  }
  m->submap_count = submap_count;
  for (j = 0; j < submap_count; j++) {
    int floor_num, residue_num;
    orc_skip_bits(p, 8);
    floor_num = (int)orc_read_bits(p, 8);
    if (floor_num >= nfloors) return ORC_ERR_INVALID_DATA;
    residue_num = (int)orc_read_bits(p, 8);
    if (residue_num >= nresidues) return ORC_ERR_INVALID_DATA;
    m->submap_floor[j] = floor_num;
    m->submap_residue[j] = residue_num;
  }
  m->channels = channels;
  for (c = 0; c < channels; c++) {
    if (mux[c] >= submap_count) return ORC_ERR_RUNTIME; /* IndexOutOfRangeException (mux == submapCount) */
    m->channel_floor[c] = m->submap_floor[mux[c]];
    m->channel_residue[c] = m->submap_residue[mux[c]];
  }
  return ORC_OK;
}

/* Mapping.cs:137-182 */
void orc_inverse_couple(float *magnitude, float *angle, int cnt) {
  int j;
  for (j = 0; j < cnt; j++) {
    float new_m, new_a;
    float old_m = magnitude[j];
    float old_a = angle[j];
    if (old_m > 0) {
      if (old_a > 0) {
        new_m = old_m;
        new_a = old_m - old_a;
      } else {
        new_a = old_m;
        new_m = old_m + old_a;
      }
    } else {
      if (old_a > 0) {
        new_m = old_m;
        new_a = old_m + old_a;
      } else {
        new_a = old_m;
        new_m = old_m - old_a;
      }
    }
    magnitude[j] = new_m;
    angle[j] = new_a;
  }
}

/* Mapping.cs:95-198 */
int orc_mapping_decode_packet(orc_decoder *d, const orc_mapping *m, orc_packet *p, int block_size, float **buffer) {
  int half = block_size >> 1;
  int nch = m->channels, i, j, c, rc = ORC_OK;
  orc_floor_data *floor_data = (orc_floor_data *)calloc((size_t)nch, sizeof *floor_data);
  int *no_execute = (int *)calloc((size_t)nch, sizeof(int));
  if (!floor_data || !no_execute) {
    free(floor_data);
    free(no_execute);
    return ORC_ERR_NOMEM;
  }

  /* read the noise floor data (:100-109) */
  for (i = 0; i < nch; i++) {
    rc = orc_floor_unpack(&d->floors[m->channel_floor[i]], d->books, p, block_size, &floor_data[i]);
    if (rc) goto done;
    no_execute[i] = !orc_floor_execute_channel(&floor_data[i]);
    memset(buffer[i], 0, sizeof(float) * (size_t)half);
  }

  /* (:112-119) */
  for (i = 0; i < m->coupling_steps; i++) {
    if (orc_floor_execute_channel(&floor_data[m->coupling_angle[i]]) ||
        orc_floor_execute_channel(&floor_data[m->coupling_magnitude[i]])) {
      floor_data[m->coupling_angle[i]].force_energy = 1;
      floor_data[m->coupling_magnitude[i]].force_energy = 1;
    }
  }

  /* decode the submaps into the residue buffer (:122-134).  Object identity of floors/residues
   * in the reference == index identity here (each index is a distinct object). */
  d->res_calls = 0;
  d->res_call_any = 0;
  for (j = 0; j < nch; j++)
    if (!no_execute[j]) d->res_call_any = 1;
  for (i = 0; i < m->submap_count; i++) {
    for (j = 0; j < nch; j++) {
      if (m->submap_floor[i] != m->channel_floor[j] || m->submap_residue[i] != m->channel_residue[j])
        floor_data[j].force_no_energy = 1;
    }
    if (d->res_calls < 16) {
      d->res_call_pos[d->res_calls] = p->pos;
      d->res_call_idx[d->res_calls] = m->submap_residue[i];
      d->res_calls++;
    }
    rc = orc_residue_decode(&d->residues[m->submap_residue[i]], d->books, p, no_execute, nch, block_size, buffer,
                            d->block1);
    if (rc) goto done;
  }

  /* inverse coupling (:137-182) */
  for (i = m->coupling_steps - 1; i >= 0; i--) {
    if (orc_floor_execute_channel(&floor_data[m->coupling_angle[i]]) ||
        orc_floor_execute_channel(&floor_data[m->coupling_magnitude[i]])) {
      orc_inverse_couple(buffer[m->coupling_magnitude[i]], buffer[m->coupling_angle[i]], half);
    }
  }

  d->last_floor_n = nch < 8 ? nch : 8; /* test hook (orc_last_floor_data) */
  for (c = 0; c < d->last_floor_n; c++) d->last_floor[c] = floor_data[c];

  /* floor apply + IMDCT (:185-197) */
  for (c = 0; c < nch; c++) {
    if (orc_floor_execute_channel(&floor_data[c])) {
      rc = orc_floor_apply(&d->floors[m->channel_floor[c]], &floor_data[c], block_size, buffer[c], d->block1);
      if (rc) goto done;
      orc_mdct_reverse(buffer[c], block_size);
    } else {
      memset(buffer[c] + half, 0, sizeof(float) * (size_t)half);
    }
  }
done:
  free(floor_data);
  free(no_execute);
  return rc;
}

/* ======================= Mode ======================= */

static const float M_PI2_F = 3.1415926539f / 2; /* Mode.cs:15 */

/* Mode.cs:69-100 */
void orc_calc_window(int prev_block, int block, int next_block, float *array) {
  int left = prev_block / 2;
  int wnd = block;
  int right = next_block / 2;
  int leftbegin = wnd / 4 - left / 2;
  int rightbegin = wnd - wnd / 4 - right / 2;
  int i;
  memset(array, 0, sizeof(float) * (size_t)block);
  for (i = 0; i < left; i++) {
    float x = (float)sin((i + .5) / left * (double)M_PI2_F);
    x *= x;
    array[leftbegin + i] = (float)sin((double)(float)(x * M_PI2_F));
  }
  for (i = leftbegin + left; i < rightbegin; i++) array[i] = 1.0f;
  for (i = 0; i < right; i++) {
    float x = (float)sin((right - i - .5) / right * (double)M_PI2_F);
    x *= x;
    array[rightbegin + i] = (float)sin((double)(float)(x * M_PI2_F));
  }
}

/* Mode.cs:102-117 */
void orc_calc_overlap(int prev_block, int block, int next_block, int *start, int *valid, int *total) {
  int left_half = prev_block / 4;
  int right_half = next_block / 4;
  *start = block / 4 - left_half;
  *total = block / 4 * 3 + right_half;
  *valid = *total - right_half * 2;
}

/* Mode.cs:24-67 */
int orc_mode_init(orc_mode *m, orc_packet *p, int block0, int block1, int nmappings) {
  int i;
  memset(m, 0, sizeof *m);
  m->block_flag = orc_read_bit(p);
  if (0 != orc_read_bits(p, 32)) return ORC_ERR_INVALID_DATA;
  m->mapping = (int)orc_read_bits(p, 8);
  if (m->mapping >= nmappings) return ORC_ERR_INVALID_DATA;
  if (m->block_flag) {
    static const int prevsel[4] = {0, 1, 0, 1}, nextsel[4] = {0, 0, 1, 1};
    m->block_size = block1;
    for (i = 0; i < 4; i++) {
      int pb = prevsel[i] ? block1 : block0, nb = nextsel[i] ? block1 : block0;
      m->windows[i] = (float *)malloc(sizeof(float) * (size_t)block1);
      if (!m->windows[i]) return ORC_ERR_NOMEM;
      orc_calc_window(pb, block1, nb, m->windows[i]);
      orc_calc_overlap(pb, block1, nb, &m->ov_start[i], &m->ov_valid[i], &m->ov_total[i]);
    }
  } else {
    m->block_size = block0;
    m->windows[0] = (float *)malloc(sizeof(float) * (size_t)block0);
    if (!m->windows[0]) return ORC_ERR_NOMEM;
    orc_calc_window(block0, block0, block0, m->windows[0]);
  }
  return ORC_OK;
}

void orc_mode_free(orc_mode *m) {
  int i;
  for (i = 0; i < 4; i++) free(m->windows[i]);
  memset(m, 0, sizeof *m);
}

/* Mode.cs:119-170 (GetPacketInfo + Decode) */
int orc_mode_decode(orc_decoder *d, const orc_mode *m, orc_packet *p, float **buffer, int *start, int *valid,
                    int *total, int *window_index) {
  int wi, rc, i, ch;
  const float *window;
  if (p->is_short) { /* :121-128 */
    *window_index = 0;
    *start = *valid = *total = 0;
    return 0;
  }
  if (m->block_flag) {
    int prev_flag = orc_read_bit(p);
    int next_flag = orc_read_bit(p);
    wi = (prev_flag ? 1 : 0) + (next_flag ? 2 : 0);
    *start = m->ov_start[wi];
    *valid = m->ov_valid[wi];
    *total = m->ov_total[wi];
  } else {
    wi = 0;
    *start = 0;
    *valid = m->block_size / 2;
    *total = m->block_size;
  }
  *window_index = wi;

  rc = orc_mapping_decode_packet(d, &d->mappings[m->mapping], p, m->block_size, buffer);
  if (rc) return rc;

  window = m->windows[wi];
  for (i = 0; i < m->block_size; i++)
    for (ch = 0; ch < d->channels; ch++) buffer[ch][i] *= window[i];
  return 1;
}
