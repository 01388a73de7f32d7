/* oracle/orc_mdct.c -- Mdct.cs restatement (test infrastructure, see orc.h).
 *
 * Operation order and association of every float expression follow Mdct.cs verbatim; the file
 * must be compiled with -ffp-contract=off so that a*b-c*d is two rounded products and a rounded
 * difference, exactly as RyuJIT emits it (mulss/mulss/subss).
 */
#include "orc_internal.h"

static const float M_PI_F = 3.14159265358979323846264f; /* Mdct.cs:9 */

/* Mdct.cs:30-63 */
void orc_mdct_tables(int n, float *a, float *b, float *c, uint16_t *bitrev) {
  int n2 = n >> 1, n4 = n2 >> 1, n8 = n4 >> 1;
  int ld = orc_ilog(n) - 1;
  int k, k2, i;
  for (k = k2 = 0; k < n4; ++k, k2 += 2) {
    /* 4 * k * M_PI / n : int*float -> float, float/int -> float, then widened for Math.Cos */
    float arg_a = (float)((float)(4 * k) * M_PI_F) / (float)n;
    float arg_b = (float)((float)((float)(k2 + 1) * M_PI_F) / (float)n) / 2.0f; /* (k2+1)*M_PI/n/2 */
    a[k2] = (float)cos((double)arg_a);
    a[k2 + 1] = (float)-sin((double)arg_a);
    b[k2] = (float)cos((double)arg_b) * .5f;
    b[k2 + 1] = (float)sin((double)arg_b) * .5f;
  }
  for (k = k2 = 0; k < n8; ++k, k2 += 2) {
    float arg_c = (float)((float)(2 * (k2 + 1)) * M_PI_F) / (float)n;
    c[k2] = (float)cos((double)arg_c);
    c[k2 + 1] = (float)-sin((double)arg_c);
  }
  for (i = 0; i < n8; ++i) bitrev[i] = (uint16_t)(orc_bit_reverse((uint32_t)i, ld - 3) << 2);
}

typedef struct {
  int n;
  float *a, *b, *c;
  uint16_t *bitrev;
} mdct_impl;

/* Mdct.cs:315-359 */
static void step3_iter0_loop(const float *A, int n, float *e, int i_off, int k_off) {
  int ee0 = i_off, ee2 = ee0 + k_off, a = 0, i, q;
  for (i = n >> 2; i > 0; --i) {
    for (q = 0; q < 8; q += 2) {
      float k00_20 = e[ee0 - q] - e[ee2 - q];
      float k01_21 = e[ee0 - q - 1] - e[ee2 - q - 1];
      e[ee0 - q] += e[ee2 - q];
      e[ee0 - q - 1] += e[ee2 - q - 1];
      e[ee2 - q] = k00_20 * A[a] - k01_21 * A[a + 1];
      e[ee2 - q - 1] = k01_21 * A[a] + k00_20 * A[a + 1];
      a += 8;
    }
    ee0 -= 8;
    ee2 -= 8;
  }
}

/* Mdct.cs:361-410 */
static void step3_inner_r_loop(const float *A, int lim, float *e, int d0, int k_off, int k1) {
  int e0 = d0, e2 = e0 + k_off, a = 0, i, q;
  for (i = lim >> 2; i > 0; --i) {
    for (q = 0; q < 8; q += 2) {
      float k00_20 = e[e0 - q] - e[e2 - q];
      float k01_21 = e[e0 - q - 1] - e[e2 - q - 1];
      e[e0 - q] += e[e2 - q];
      e[e0 - q - 1] += e[e2 - q - 1];
      e[e2 - q] = k00_20 * A[a] - k01_21 * A[a + 1];
      e[e2 - q - 1] = k01_21 * A[a] + k00_20 * A[a + 1];
      a += k1;
    }
    e0 -= 8;
    e2 -= 8;
  }
}

/* Mdct.cs:412-461 */
static void step3_inner_s_loop(const float *A, int n, float *e, int i_off, int k_off, int a, int a_off, int k0) {
  float A0 = A[a], A1 = A[a + 1];
  float A2 = A[a + a_off], A3 = A[a + a_off + 1];
  float A4 = A[a + a_off * 2], A5 = A[a + a_off * 2 + 1];
  float A6 = A[a + a_off * 3], A7 = A[a + a_off * 3 + 1];
  float k00, k11;
  int ee0 = i_off, ee2 = ee0 + k_off, i;
  for (i = n; i > 0; --i) {
    k00 = e[ee0] - e[ee2];
    k11 = e[ee0 - 1] - e[ee2 - 1];
    e[ee0] += e[ee2];
    e[ee0 - 1] += e[ee2 - 1];
    e[ee2] = k00 * A0 - k11 * A1;
    e[ee2 - 1] = k11 * A0 + k00 * A1;

    k00 = e[ee0 - 2] - e[ee2 - 2];
    k11 = e[ee0 - 3] - e[ee2 - 3];
    e[ee0 - 2] += e[ee2 - 2];
    e[ee0 - 3] += e[ee2 - 3];
    e[ee2 - 2] = k00 * A2 - k11 * A3;
    e[ee2 - 3] = k11 * A2 + k00 * A3;

    k00 = e[ee0 - 4] - e[ee2 - 4];
    k11 = e[ee0 - 5] - e[ee2 - 5];
    e[ee0 - 4] += e[ee2 - 4];
    e[ee0 - 5] += e[ee2 - 5];
    e[ee2 - 4] = k00 * A4 - k11 * A5;
    e[ee2 - 5] = k11 * A4 + k00 * A5;

    k00 = e[ee0 - 6] - e[ee2 - 6];
    k11 = e[ee0 - 7] - e[ee2 - 7];
    e[ee0 - 6] += e[ee2 - 6];
    e[ee0 - 7] += e[ee2 - 7];
    e[ee2 - 6] = k00 * A6 - k11 * A7;
    e[ee2 - 7] = k11 * A6 + k00 * A7;

    ee0 -= k0;
    ee2 -= k0;
  }
}

/* Mdct.cs:509-535 */
static void iter_54(float *e, int z) {
  float k00, k11, k22, k33, y0, y1, y2, y3;
  k00 = e[z] - e[z - 4];
  y0 = e[z] + e[z - 4];
  y2 = e[z - 2] + e[z - 6];
  k22 = e[z - 2] - e[z - 6];

  e[z] = y0 + y2;
  e[z - 2] = y0 - y2;

  k33 = e[z - 3] - e[z - 7];

  e[z - 4] = k00 + k33;
  e[z - 6] = k00 - k33;

  k11 = e[z - 1] - e[z - 5];
  y1 = e[z - 1] + e[z - 5];
  y3 = e[z - 3] + e[z - 7];

  e[z - 1] = y1 + y3;
  e[z - 3] = y1 - y3;
  e[z - 5] = k11 - k22;
  e[z - 7] = k11 + k22;
}

/* Mdct.cs:463-507 */
static void step3_inner_s_loop_ld654(const float *A, int n, float *e, int i_off, int base_n) {
  int a_off = base_n >> 3;
  float A2 = A[a_off];
  int z = i_off;
  int base = z - 16 * n;
  while (z > base) {
    float k00, k11;
    k00 = e[z] - e[z - 8];
    k11 = e[z - 1] - e[z - 9];
    e[z] += e[z - 8];
    e[z - 1] += e[z - 9];
    e[z - 8] = k00;
    e[z - 9] = k11;

    k00 = e[z - 2] - e[z - 10];
    k11 = e[z - 3] - e[z - 11];
    e[z - 2] += e[z - 10];
    e[z - 3] += e[z - 11];
    e[z - 10] = (k00 + k11) * A2;
    e[z - 11] = (k11 - k00) * A2;

    k00 = e[z - 12] - e[z - 4];
    k11 = e[z - 5] - e[z - 13];
    e[z - 4] += e[z - 12];
    e[z - 5] += e[z - 13];
    e[z - 12] = k11;
    e[z - 13] = k00;

    k00 = e[z - 14] - e[z - 6];
    k11 = e[z - 7] - e[z - 15];
    e[z - 6] += e[z - 14];
    e[z - 7] += e[z - 15];
    e[z - 14] = (k00 + k11) * A2;
    e[z - 15] = (k00 - k11) * A2;

    iter_54(e, z);
    iter_54(e, z - 8);
    z -= 16;
  }
}

/* Mdct.cs:65-313 */
static void calc_reverse(const mdct_impl *m, float *buffer) {
  const int n = m->n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
  const int ld = orc_ilog(n) - 1;
  const float *A = m->a, *B = m->b, *C = m->c;
  float *u, *v;
  float *buf2 = (float *)calloc((size_t)n2, sizeof(float)); /* new float[_n2] :69 */

  /* step 0 (:74-97) */
  {
    int d = n2 - 2, AA = 0, e = 0, e_stop = n2;
    while (e != e_stop) {
      buf2[d + 1] = (buffer[e] * A[AA] - buffer[e + 2] * A[AA + 1]);
      buf2[d] = (buffer[e] * A[AA + 1] + buffer[e + 2] * A[AA]);
      d -= 2;
      AA += 2;
      e += 4;
    }
    e = n2 - 3;
    while (d >= 0) {
      buf2[d + 1] = (-buffer[e + 2] * A[AA] - -buffer[e] * A[AA + 1]);
      buf2[d] = (-buffer[e + 2] * A[AA + 1] + -buffer[e] * A[AA]);
      d -= 2;
      AA += 2;
      e -= 4;
    }
  }

  u = buffer;
  v = buf2;

  /* step 2 (:105-139) */
  {
    int AA = n2 - 8, e0 = n4, e1 = 0, d0 = n4, d1 = 0;
    while (AA >= 0) {
      float v40_20, v41_21;
      v41_21 = v[e0 + 1] - v[e1 + 1];
      v40_20 = v[e0] - v[e1];
      u[d0 + 1] = v[e0 + 1] + v[e1 + 1];
      u[d0] = v[e0] + v[e1];
      u[d1 + 1] = v41_21 * A[AA + 4] - v40_20 * A[AA + 5];
      u[d1] = v40_20 * A[AA + 4] + v41_21 * A[AA + 5];

      v41_21 = v[e0 + 3] - v[e1 + 3];
      v40_20 = v[e0 + 2] - v[e1 + 2];
      u[d0 + 3] = v[e0 + 3] + v[e1 + 3];
      u[d0 + 2] = v[e0 + 2] + v[e1 + 2];
      u[d1 + 3] = v41_21 * A[AA] - v40_20 * A[AA + 1];
      u[d1 + 2] = v40_20 * A[AA] + v41_21 * A[AA + 1];

      AA -= 8;
      d0 += 4;
      d1 += 4;
      e0 += 4;
      e1 += 4;
    }
  }

  /* step 3 (:144-186) */
  step3_iter0_loop(A, n >> 4, u, n2 - 1 - n4 * 0, -n8);
  step3_iter0_loop(A, n >> 4, u, n2 - 1 - n4 * 1, -n8);

  step3_inner_r_loop(A, n >> 5, u, n2 - 1 - n8 * 0, -(n >> 4), 16);
  step3_inner_r_loop(A, n >> 5, u, n2 - 1 - n8 * 1, -(n >> 4), 16);
  step3_inner_r_loop(A, n >> 5, u, n2 - 1 - n8 * 2, -(n >> 4), 16);
  step3_inner_r_loop(A, n >> 5, u, n2 - 1 - n8 * 3, -(n >> 4), 16);

  {
    int l = 2;
    for (; l < (ld - 3) >> 1; ++l) {
      int k0 = n >> (l + 2);
      int k0_2 = k0 >> 1;
      int lim = 1 << (l + 1);
      int i;
      for (i = 0; i < lim; ++i) step3_inner_r_loop(A, n >> (l + 4), u, n2 - 1 - k0 * i, -k0_2, 1 << (l + 3));
    }
    for (; l < ld - 6; ++l) {
      int k0 = n >> (l + 2);
      int k1 = 1 << (l + 3);
      int k0_2 = k0 >> 1;
      int rlim = n >> (l + 6);
      int lim = 1 << (l + 1); /* `1 << l + 1` parses as 1 << (l+1) */
      int i_off = n2 - 1;
      int A0 = 0, r;
      for (r = rlim; r > 0; --r) {
        step3_inner_s_loop(A, lim, u, i_off, -k0_2, A0, k1, k0);
        A0 += k1 * 4;
        i_off -= 8;
      }
    }
  }

  step3_inner_s_loop_ld654(A, n >> 5, u, n2 - 1, n);

  /* steps 4, 5, 6 (:189-214) */
  {
    int bit = 0, d0 = n4 - 4, d1 = n2 - 4;
    while (d0 >= 0) {
      int k4;
      k4 = m->bitrev[bit];
      v[d1 + 3] = u[k4];
      v[d1 + 2] = u[k4 + 1];
      v[d0 + 3] = u[k4 + 2];
      v[d0 + 2] = u[k4 + 3];

      k4 = m->bitrev[bit + 1];
      v[d1 + 1] = u[k4];
      v[d1] = u[k4 + 1];
      v[d0 + 1] = u[k4 + 2];
      v[d0] = u[k4 + 3];

      d0 -= 4;
      d1 -= 4;
      bit += 2;
    }
  }

  /* step 7 (:217-258) */
  {
    int c = 0, d = 0, e = n2 - 4;
    while (d < e) {
      float a02, a11, b0, b1, b2, b3;
      a02 = v[d] - v[e + 2];
      a11 = v[d + 1] + v[e + 3];

      b0 = C[c + 1] * a02 + C[c] * a11;
      b1 = C[c + 1] * a11 - C[c] * a02;

      b2 = v[d] + v[e + 2];
      b3 = v[d + 1] - v[e + 3];

      v[d] = b2 + b0;
      v[d + 1] = b3 + b1;
      v[e + 2] = b2 - b0;
      v[e + 3] = b1 - b3;

      a02 = v[d + 2] - v[e];
      a11 = v[d + 3] + v[e + 1];

      b0 = C[c + 3] * a02 + C[c + 2] * a11;
      b1 = C[c + 3] * a11 - C[c + 2] * a02;

      b2 = v[d + 2] + v[e];
      b3 = v[d + 3] - v[e + 1];

      v[d + 2] = b2 + b0;
      v[d + 3] = b3 + b1;
      v[e] = b2 - b0;
      v[e + 1] = b1 - b3;

      c += 4;
      d += 4;
      e -= 4;
    }
  }

  /* step 8 + decode (:261-312) */
  {
    int b = n2 - 8, e = n2 - 8, d0 = 0, d1 = n2 - 4, d2 = n2, d3 = n - 4;
    while (e >= 0) {
      float p0, p1, p2, p3;
      p3 = buf2[e + 6] * B[b + 7] - buf2[e + 7] * B[b + 6];
      p2 = -buf2[e + 6] * B[b + 6] - buf2[e + 7] * B[b + 7];

      buffer[d0] = p3;
      buffer[d1 + 3] = -p3;
      buffer[d2] = p2;
      buffer[d3 + 3] = p2;

      p1 = buf2[e + 4] * B[b + 5] - buf2[e + 5] * B[b + 4];
      p0 = -buf2[e + 4] * B[b + 4] - buf2[e + 5] * B[b + 5];

      buffer[d0 + 1] = p1;
      buffer[d1 + 2] = -p1;
      buffer[d2 + 1] = p0;
      buffer[d3 + 2] = p0;

      p3 = buf2[e + 2] * B[b + 3] - buf2[e + 3] * B[b + 2];
      p2 = -buf2[e + 2] * B[b + 2] - buf2[e + 3] * B[b + 3];

      buffer[d0 + 2] = p3;
      buffer[d1 + 1] = -p3;
      buffer[d2 + 2] = p2;
      buffer[d3 + 1] = p2;

      p1 = buf2[e] * B[b + 1] - buf2[e + 1] * B[b];
      p0 = -buf2[e] * B[b] - buf2[e + 1] * B[b + 1];

      buffer[d0 + 3] = p1;
      buffer[d1] = -p1;
      buffer[d2 + 3] = p0;
      buffer[d3] = p0;

      b -= 8;
      e -= 8;
      d0 += 4;
      d2 += 4;
      d1 -= 4;
      d3 -= 4;
    }
  }
  free(buf2);
}

/* Mdct.cs:11-21: per-n cache of MdctImpl.  n is a power of two in [64, 8192].  The reference's cache belongs to one Mdct
 * instance (one per StreamDecoder, used by one thread); here it is per thread, so that several oracle decoders may run
 * on different threads (bench.py's all-cores CPU baseline) without sharing mutable state. */
static __thread mdct_impl g_cache[16];

void orc_mdct_reverse(float *buf, int n) {
  int slot = orc_ilog(n) & 15;
  mdct_impl *m = &g_cache[slot];
  if (m->n != n) {
    free(m->a);
    free(m->b);
    free(m->c);
    free(m->bitrev);
    m->n = n;
    m->a = (float *)malloc(sizeof(float) * (size_t)(n / 2));
    m->b = (float *)malloc(sizeof(float) * (size_t)(n / 2));
    m->c = (float *)malloc(sizeof(float) * (size_t)(n / 4));
    m->bitrev = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(n / 8));
    orc_mdct_tables(n, m->a, m->b, m->c, m->bitrev);
  }
  calc_reverse(m, buf);
}
