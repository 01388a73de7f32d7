// imdct_wave.h -- the wavefront IMDCT as device functions (shared by kernels_imdct.hip and the fused
// spectrum + IMDCT kernel in kernels_spectrum.hip).
//
//
// The reference's inverse MDCT (Mdct.cs:65-313, stb_vorbis lineage) is a decimation-in-frequency FFT over
// N = n/4 complex points wrapped in a pre-twiddle (step 0) and a post-twiddle (steps 7, 8):
//
//   step 0   spectrum -> v                       N items, each one complex multiply with _a
//   stages   D = N/2, N/4, ..., 8 (complex)      radix-2 butterflies  hi' = hi + lo, lo' = (hi - lo) * tw
//            ("step 2" is the D = N/2 stage: same expression shapes, twiddle index n2-4-4c, Mdct.cs:105-139;
//             stage l of "step 3" has D = N >> (l+2), Mdct.cs:144-183)
//   ld654    D = 4, 2, 1 fused, trivial twiddles  Mdct.cs:463-535
//   4-6      bit reversal                          Mdct.cs:189-214
//   7, 8     post-twiddles with _c and _b, 4-way symmetric expansion to n outputs   Mdct.cs:217-312
//
// For every radix-2 stage the twiddle of the pair (c_lo, c_lo + D) is _a[t], _a[t+1] with
//   t = (D - 1 - (c_lo & (D-1))) * (n2 / D)
// (derived from the loop nests of step3_iter0_loop / step3_inner_r_loop / step3_inner_s_loop; it reproduces
// `AA = n2-8-8i` of step 2 as the D = N/2 case).  All butterflies of a stage touch disjoint elements, so the
// stages are regrouped here into register passes of up to three stages (radix 8): a lane loads the 8 complex
// points base + S*k, runs the butterflies of stages D = 4S, 2S, S on them with the reference's exact per-
// butterfly arithmetic (separately rounded mul / add / sub, -ffp-contract=off) and stores them back.  One
// 64-lane wavefront owns one channel of one frame; its LDS slice (n/2 floats + padding) is private, so the
// only synchronisation is the wave's own program order.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_common.h"

namespace {

// A wavefront's LDS slice is private to it: ordering between its own DS writes and reads only needs the
// compiler not to reorder them (the hardware executes one wave's DS instructions in order).
// The fences are scoped to the LDS address space ("local") so that independent global loads (twiddles, window,
// the next tables) may still be scheduled across them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A workgroup barrier that orders LDS accesses only: global loads and LDS-DMA in flight (vmcnt) are not waited for, as
// __syncthreads() would (callers that have staging DMA under way across the barrier: kernels_synth.hip, frame groups).
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS layout: complex point c (float2) lives at float2 slot c + 8*(c>>6): the pad keeps the stride-8 and
// stride-64 set patterns of the radix passes off each other's banks.
__device__ __forceinline__ int phys(int c) { return c + ((c >> 6) << 3); }
// The same slots with the four 16-byte chunks of every 8-point block rotated by (block / 4) mod 4: the layout between the
// last radix pass and the end of the looped transform.  The D = 4, 2, 1 pass has every lane read a whole block (4 x 16 bytes,
// all lanes at multiples of 64 bytes): unrotated, a 16-lane group of a ds_read_b128 hits 4 of the 16 slot residues -- 4-way
// bank conflicts on all of its 8 LDS instructions (rocprofv3: 44 % of the transform's LDS cycles were conflict cycles).
__device__ __forceinline__ int phys_rot(int c) {
  const int b = c >> 3, p = c & 7;
  return ((b + (b >> 3)) << 3) + ((((p >> 1) + (b >> 2)) & 3) << 1) + (p & 1);
}

template <int LD>
struct Geo {
  static constexpr int n = 1 << LD;
  static constexpr int n2 = n >> 1;
  static constexpr int n4 = n >> 2;
  static constexpr int n8 = n >> 3;
  static constexpr int N = n >> 2;                       // complex points
  static constexpr int LDS_FLOATS = 2 * (N + (N >> 3));  // padded
};

// Floats of a block size's pass twiddles in the lane-ordered table (host_setup.cpp: build_mdct_tables), rounded up to a 16-byte
// boundary: where the output stage's gather-address table begins.
template <int LD>
constexpr int tw_pass_floats() {
  int total = 0, remain = LD - 5;
  while (remain > 0) {
    const int R = remain >= 3 ? 3 : remain;
    total += 2 * ((1 << R) - 1) * ((1 << (LD - 2)) >> R);
    remain -= R;
  }
  return (total + 3) & ~3;
}

// One register pass over R radix-2 stages whose smallest distance is S complex points.
// TW: this pass's lane-ordered twiddles, [pair component][set] (host_setup.cpp build_mdct_tables).
template <int LD, int R, int S, bool ROT_OUT = false>
__device__ __forceinline__ void radix_pass(float2* __restrict__ l2, const float* __restrict__ TW, int lane) {
  using G = Geo<LD>;
  constexpr int K = 1 << R;
  constexpr int NSETS = G::N >> R;
#pragma unroll 1
  for (int s = lane; s < NSETS; s += 64) {
    const int r = s & (S - 1);
    const int blk = s / S;
    const int base = blk * (S << R) + r;
    (void)r;
    // twiddles of this set: coalesced across the wave (consecutive sets = consecutive floats)
    float tw[2 * (K - 1)];
#pragma unroll
    for (int j = 0; j < 2 * (K - 1); ++j) tw[j] = TW[j * NSETS + s];
    float2 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = l2[phys(base + S * k)];
#pragma unroll
    for (int st = R - 1; st >= 0; --st) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (k & (1 << st)) continue;
        const int lo = k, hi = k | (1 << st);
        // pair index: stages from the largest distance down, kk = k mod 2^st ascending
        const int pi = (K - (2 << st)) + (k & ((1 << st) - 1));
        const float a0 = tw[2 * pi], a1 = tw[2 * pi + 1];
        // Mdct.cs:324-329: k00 = e[ee0]-e[ee2] (odd slot), k01 = e[ee0-1]-e[ee2-1] (even slot)
        const float d1 = v[hi].y - v[lo].y;
        const float d0 = v[hi].x - v[lo].x;
        v[hi].y = v[hi].y + v[lo].y;
        v[hi].x = v[hi].x + v[lo].x;
        v[lo].y = d1 * a0 - d0 * a1;
        v[lo].x = d0 * a0 + d1 * a1;
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) l2[ROT_OUT ? phys_rot(base + S * k) : phys(base + S * k)] = v[k];
  }
}

// Mdct.cs:509-535 on a register block: e[] holds u[z-15 .. z], so u[z-k] == e[15-k]
__device__ __forceinline__ void iter_54_regs(float* e, int zo /* index of u[z] inside e */) {
#define U(k) e[zo - (k)]
  float k00 = U(0) - U(4);
  float y0 = U(0) + U(4);
  float y2 = U(2) + U(6);
  float k22 = U(2) - U(6);
  U(0) = y0 + y2;
  U(2) = y0 - y2;
  float k33 = U(3) - U(7);
  U(4) = k00 + k33;
  U(6) = k00 - k33;
  float k11 = U(1) - U(5);
  float y1 = U(1) + U(5);
  float y3 = U(3) + U(7);
  U(1) = y1 + y3;
  U(3) = y1 - y3;
  U(5) = k11 - k22;
  U(7) = k11 + k22;
#undef U
}

// Fused last three stages (Mdct.cs:463-507): blocks of 8 consecutive complex points.
template <int LD, bool ROT = false>
__device__ __forceinline__ void ld654_pass(float2* __restrict__ l2, const float* __restrict__ A, int lane) {
  using G = Geo<LD>;
  const float A2 = A[G::n >> 3];
#pragma unroll 1
  for (int q = lane; q < (G::N >> 3); q += 64) {
    const int c0 = G::N - 8 - 8 * q;  // complex index of u[z-15], z = n2-1-16q
    float e[16];
    float4* p4 = reinterpret_cast<float4*>(l2 + phys(c0));
    const int rot = ROT ? (c0 >> 5) & 3 : 0;  // (block / 4) mod 4, see phys_rot
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 t = p4[(j + rot) & 3];
      e[4 * j] = t.x; e[4 * j + 1] = t.y; e[4 * j + 2] = t.z; e[4 * j + 3] = t.w;
    }
#define U(k) e[15 - (k)]
    float k00, k11;
    k00 = U(0) - U(8);
    k11 = U(1) - U(9);
    U(0) = U(0) + U(8);
    U(1) = U(1) + U(9);
    U(8) = k00;
    U(9) = k11;

    k00 = U(2) - U(10);
    k11 = U(3) - U(11);
    U(2) = U(2) + U(10);
    U(3) = U(3) + U(11);
    U(10) = (k00 + k11) * A2;
    U(11) = (k11 - k00) * A2;

    k00 = U(12) - U(4);
    k11 = U(5) - U(13);
    U(4) = U(4) + U(12);
    U(5) = U(5) + U(13);
    U(12) = k11;
    U(13) = k00;

    k00 = U(14) - U(6);
    k11 = U(7) - U(15);
    U(6) = U(6) + U(14);
    U(7) = U(7) + U(15);
    U(14) = (k00 + k11) * A2;
    U(15) = (k00 - k11) * A2;
#undef U
    iter_54_regs(e, 15);
    iter_54_regs(e, 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) p4[(j + rot) & 3] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
  }
}

template <int LD, int REMAIN>
struct Passes {
  // REMAIN = number of radix-2 stages still to run whose distances are 8<<(REMAIN-1) ... 8
  static __device__ __forceinline__ void run(float2* l2, const float* TW, int lane) {
    constexpr int R = REMAIN >= 3 ? 3 : REMAIN;
    constexpr int S = 8 << (REMAIN - R);
    radix_pass<LD, R, S, (REMAIN - R == 0)>(l2, TW, lane);  // the last pass leaves the rotated layout (phys_rot)
    wave_sync();
    Passes<LD, REMAIN - R>::run(l2, TW + 2 * ((1 << R) - 1) * (Geo<LD>::N >> R), lane);
  }
};
template <int LD>
struct Passes<LD, 0> {
  static __device__ __forceinline__ void run(float2*, const float*, int) {}
};

// The same passes with every pass's twiddles in registers one pass ahead of their use (callers with registers to spare: a
// workgroup whose occupancy is set by LDS, not by VGPRs): the L2 round trip of a pass's table is hidden behind the arithmetic
// of the pass in front of it instead of standing at its head.  Same loads, same arithmetic, same order as Passes<>.
template <int LD, int REMAIN>
struct PassesPF {
  using G = Geo<LD>;
  static constexpr int R = REMAIN >= 3 ? 3 : REMAIN;
  static constexpr int S = 8 << (REMAIN - R);
  static constexpr int K = 1 << R;
  static constexpr int NSETS = G::N >> R;
  static constexpr int ITER = (NSETS + 63) / 64;
  static constexpr int NTW = 2 * (K - 1);
  static __device__ __forceinline__ void load(const float* __restrict__ TW, int lane, float (&tw)[ITER * NTW]) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int s = lane + 64 * it;
      const int sc = s < NSETS ? s : NSETS - 1;
#pragma unroll
      for (int j = 0; j < NTW; ++j) tw[it * NTW + j] = TW[j * NSETS + sc];
    }
  }
  static __device__ __forceinline__ void compute(float2* __restrict__ l2, const float (&tw)[ITER * NTW], int lane) {
    constexpr bool ROT_OUT = (REMAIN - R == 0);  // the last pass leaves the rotated layout (phys_rot)
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int s = lane + 64 * it;
      if (s >= NSETS) continue;
      const int r = s & (S - 1);
      const int blk = s / S;
      const int base = blk * (S << R) + r;
      float2 v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = l2[phys(base + S * k)];
#pragma unroll
      for (int st = R - 1; st >= 0; --st) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (k & (1 << st)) continue;
          const int lo = k, hi = k | (1 << st);
          const int pi = (K - (2 << st)) + (k & ((1 << st) - 1));
          const float a0 = tw[it * NTW + 2 * pi], a1 = tw[it * NTW + 2 * pi + 1];
          // Mdct.cs:324-329: k00 = e[ee0]-e[ee2] (odd slot), k01 = e[ee0-1]-e[ee2-1] (even slot)
          const float d1 = v[hi].y - v[lo].y;
          const float d0 = v[hi].x - v[lo].x;
          v[hi].y = v[hi].y + v[lo].y;
          v[hi].x = v[hi].x + v[lo].x;
          v[lo].y = d1 * a0 - d0 * a1;
          v[lo].x = d0 * a0 + d1 * a1;
        }
      }
#pragma unroll
      for (int k = 0; k < K; ++k) l2[ROT_OUT ? phys_rot(base + S * k) : phys(base + S * k)] = v[k];
    }
  }
  // tw: this pass's twiddles, already in registers
  static __device__ __forceinline__ void run(float2* l2, const float* TW, int lane, const float (&tw)[ITER * NTW]) {
    using Next = PassesPF<LD, REMAIN - R>;
    const float* TWn = TW + NTW * NSETS;
    if constexpr (REMAIN - R > 0) {
      float twn[Next::ITER * Next::NTW];
      Next::load(TWn, lane, twn);
      compute(l2, tw, lane);
      wave_sync();
      Next::run(l2, TWn, lane, twn);
    } else {
      compute(l2, tw, lane);
      wave_sync();
    }
  }
};

// Full IMDCT of one channel-frame by one wavefront.  X: n/2 spectrum floats (global), out: n floats (global),
// w: window (n floats) applied on the way out (Mode.cs:160-166).
// sink(slot, idx, v): receives the 8 float4 output chunks a lane produces per pair index (slot = 8 * iteration + 0..7 is a
// compile-time constant after unrolling; blocks up to 2048 have one iteration), idx = position of the chunk inside the n-sample block.
// INPLACE: X may be (part of) the wavefront's own LDS slice `lds` -- the whole spectrum is then read into
// registers before the first store (n <= 2048: at most 4 float4 per lane).
// WGSYNC (with INPLACE): the slice may overlay OTHER wavefronts' spectra as well -- a workgroup barrier, not the
// wavefront's own program order, separates the spectrum loads from the first store (every wavefront of the workgroup
// passes exactly one __syncthreads; those without a transform of their own call it themselves).
// PRESYNC (with INPLACE): the caller's workgroup barrier in front of the transform (the spectrum is other wavefronts' work too) is
// taken HERE, after the step-0 twiddle loads have been issued, so that their L2 round trip overlaps the wait.
// LDSBAR (with WGSYNC): that barrier orders LDS accesses only (lds_barrier above).
template <int LD, bool WIN, typename Sink, bool INPLACE = false, bool WGSYNC = false, bool PRESYNC = false, bool PF = false, bool LDSBAR = false>
__device__ __forceinline__ void imdct_wave_sink(const float* X, const float* __restrict__ w, float* lds,
                                                const float* __restrict__ A, const float* __restrict__ B,
                                                const float* __restrict__ C, const float* __restrict__ TW, int lane,
                                                Sink sink, long long* stamp = nullptr, int dbg_skip = 0) {
  using G = Geo<LD>;
  float2* l2 = reinterpret_cast<float2*>(lds);
  // profiling builds: shader-clock stamps of lane 0 (0 entry, 1 step 0 done, 2 radix passes done, 3 D=4,2,1 done, 4 end)
#define IMDCT_T(k) do { if (stamp && lane == 0) stamp[k] = clock64(); } while (0)
  IMDCT_T(0);

  // step 0 (Mdct.cs:74-97).  The float4 at X[4j] feeds the first-half item j (even elements) and the
  // second-half item n8-1-j (odd elements): complex points N-1-j and j.
  auto step0 = [&](int j, const float4 x, const float2 a_lo, const float2 a_hi) {
    float2 hi, lo;
    hi.y = (x.x * a_lo.x - x.z * a_lo.y);      // buf2[d+1], d = n2-2-2j
    hi.x = (x.x * a_lo.y + x.z * a_lo.x);      // buf2[d]
    lo.y = (-x.w * a_hi.x - -x.y * a_hi.y);    // buf2[d'+1], d' = 2j
    lo.x = (-x.w * a_hi.y + -x.y * a_hi.x);    // buf2[d']
    l2[phys(G::N - 1 - j)] = hi;
    l2[phys(j)] = lo;
  };
  const float2* A2 = reinterpret_cast<const float2*>(A);  // a_lo = A[2j], A[2j+1]; a_hi = A[n2-2-2j], A[n2-1-2j]
  using PF1 = PassesPF<LD, LD - 5>;
  float tw_first[PF ? PF1::ITER * PF1::NTW : 1];
  if constexpr (PF) PF1::load(TW, lane, tw_first);  // in flight across the barrier and step 0
  if constexpr (INPLACE) {
    constexpr int J = (G::n8 + 63) / 64;
    static_assert(J <= (WGSYNC ? 8 : 4), "in-place form keeps the spectrum in registers");
    float2 pa_lo[J], pa_hi[J];
#pragma unroll
    for (int r = 0; r < J; ++r) {
      const int j = lane + 64 * r;
      const int jc = j < G::n8 ? j : G::n8 - 1;
      pa_lo[r] = A2[jc];
      pa_hi[r] = A2[G::n4 - 1 - jc];
    }
    if constexpr (PRESYNC) __syncthreads();
    float4 xin[J];
#pragma unroll
    for (int r = 0; r < J; ++r) {
      const int j = lane + 64 * r;
      xin[r] = reinterpret_cast<const float4*>(X)[j < G::n8 ? j : G::n8 - 1];
    }
    if constexpr (WGSYNC && LDSBAR) lds_barrier(); else if constexpr (WGSYNC) __syncthreads(); else wave_sync();
#pragma unroll
    for (int r = 0; r < J; ++r) {
      const int j = lane + 64 * r;
      if (j < G::n8) step0(j, xin[r], pa_lo[r], pa_hi[r]);
    }
  } else {
#pragma unroll 1
    for (int j = lane; j < G::n8; j += 64) step0(j, reinterpret_cast<const float4*>(X)[j], A2[j], A2[G::n4 - 1 - j]);
  }
  wave_sync();
  IMDCT_T(1);

  // radix-2 stages D = N/2 ... 8, three per register pass (dbg_skip: profiling builds time the kernel without them)
  if constexpr (PF) {
    PF1::run(l2, TW, lane, tw_first);
  } else {
    if (!(dbg_skip & 1)) Passes<LD, LD - 5>::run(l2, TW, lane);
  }
  IMDCT_T(2);

  // the output stage's tables (_c, _b): fetched here, a phase ahead of their use -- with one pair index per lane in the
  // register-lean form, for every pair index of the lane where registers are plentiful (PF)
  constexpr bool kPre = (INPLACE && !WGSYNC && (G::n >> 5) <= 64) || PF;
  constexpr int ITERO = kPre ? (((G::n >> 5) + 63) / 64) : 1;
  float4 pcc[ITERO][2], pbl[ITERO][2], pbh[ITERO][2];
  // ... and, one pair per lane, its eight gather slots from the table behind the pass twiddles instead of ~100 integer
  // instructions of bit reversal and layout arithmetic
  constexpr bool kFinTab = kPre && ITERO <= 2;
  uint4 fin[ITERO];
  if constexpr (kFinTab) {
#pragma unroll
    for (int it = 0; it < ITERO; ++it) {
      const int pq = lane + 64 * it;
      const int pp = pq < (G::n >> 5) ? pq : (G::n >> 5) - 1;
      fin[it] = reinterpret_cast<const uint4*>(TW + tw_pass_floats<LD>())[pp];
    }
  }
  if constexpr (kPre) {
#pragma unroll
    for (int it = 0; it < ITERO; ++it) {
      const int pq = lane + 64 * it;
      const int pp = pq < (G::n >> 5) ? pq : (G::n >> 5) - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pcc[it][h] = reinterpret_cast<const float4*>(C)[2 * pp + h];
        const int i8 = h == 0 ? pp : (G::n >> 4) - 1 - pp;
        const int b = G::n2 - 8 - 8 * i8;
        pbl[it][h] = *reinterpret_cast<const float4*>(B + b);
        pbh[it][h] = *reinterpret_cast<const float4*>(B + b + 4);
      }
    }
  }
  // D = 4, 2, 1
  if (!(dbg_skip & 2)) ld654_pass<LD, true>(l2, A, lane);
  wave_sync();
  IMDCT_T(3);

  // steps 4-6 (bit reversal), 7 and 8 fused.  Pair index p covers step-7 iterations 2p and 2p+1, whose
  // results are exactly the inputs of step-8 iterations p and n/16-1-p.
  const float* lf = lds;
#pragma unroll
  for (int it = 0; it < (((G::n >> 5) + 63) / 64); ++it) {
    const int p = lane + 64 * it;
    if (p >= (G::n >> 5)) continue;
    float vd[8], ve[8];  // v[8p .. 8p+7] and v[n2-8-8p .. n2-1-8p]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 2 * p + h;
      const int ir = (G::n >> 4) - 1 - i;
      // _bitrev[j] = BitReverse(j, ld-3) << 2 (Mdct.cs:59-62)
      const int kE0 = (int)(__brev((unsigned)(2 * i)) >> (32 - (LD - 3))) << 2;
      const int kE1 = (int)(__brev((unsigned)(2 * i + 1)) >> (32 - (LD - 3))) << 2;
      const int kD0 = (int)(__brev((unsigned)(2 * ir)) >> (32 - (LD - 3))) << 2;
      const int kD1 = (int)(__brev((unsigned)(2 * ir + 1)) >> (32 - (LD - 3))) << 2;
      // v[d1+3]=u[k], v[d1+2]=u[k+1] (k=BR[2i]); v[d1+1]=u[k'], v[d1]=u[k'+1] (k'=BR[2i+1]); d1 = n2-4-4i
      float2 e0, e1, g0, g1;
      if constexpr (kFinTab) {
        const uint4 ft = fin[it < ITERO ? it : 0];
        const unsigned we = h == 0 ? ft.x : ft.z, wg = h == 0 ? ft.y : ft.w;
        const float2* l2c = reinterpret_cast<const float2*>(lf);
        e0 = l2c[we & 0xFFFFu]; e1 = l2c[we >> 16];
        g0 = l2c[wg & 0xFFFFu]; g1 = l2c[wg >> 16];
      } else {
      e0 = *reinterpret_cast<const float2*>(lf + 2 * phys_rot(kE0 >> 1));
      e1 = *reinterpret_cast<const float2*>(lf + 2 * phys_rot(kE1 >> 1));
      // v[d0+3]=u[k+2], v[d0+2]=u[k+3] (k=BR[2i']); v[d0+1]=u[k'+2], v[d0]=u[k'+3]; d0 = n4-4-4i' = 4i
      g0 = *reinterpret_cast<const float2*>(lf + 2 * phys_rot((kD0 >> 1) + 1));
      g1 = *reinterpret_cast<const float2*>(lf + 2 * phys_rot((kD1 >> 1) + 1));
      }
      float vD0 = g1.y, vD1 = g1.x, vD2 = g0.y, vD3 = g0.x;  // v[4i .. 4i+3]
      float vE0 = e1.y, vE1 = e1.x, vE2 = e0.y, vE3 = e0.x;  // v[n2-4-4i .. n2-1-4i]
      // step 7 (Mdct.cs:217-258) for iteration i: c = d = 4i, e = n2-4-4i
      float4 cc;
      if constexpr (kPre) cc = pcc[it < ITERO ? it : 0][h]; else cc = reinterpret_cast<const float4*>(C)[i];
      float a02, a11, b0, b1, b2, b3;
      a02 = vD0 - vE2;
      a11 = vD1 + vE3;
      b0 = cc.y * a02 + cc.x * a11;
      b1 = cc.y * a11 - cc.x * a02;
      b2 = vD0 + vE2;
      b3 = vD1 - vE3;
      const float nD0 = b2 + b0, nD1 = b3 + b1, nE2 = b2 - b0, nE3 = b1 - b3;
      a02 = vD2 - vE0;
      a11 = vD3 + vE1;
      b0 = cc.w * a02 + cc.z * a11;
      b1 = cc.w * a11 - cc.z * a02;
      b2 = vD2 + vE0;
      b3 = vD3 - vE1;
      const float nD2 = b2 + b0, nD3 = b3 + b1, nE0 = b2 - b0, nE1 = b1 - b3;
      vd[4 * h] = nD0; vd[4 * h + 1] = nD1; vd[4 * h + 2] = nD2; vd[4 * h + 3] = nD3;
      // v[n2-8-8p+k]: iteration 2p covers k = 4..7, iteration 2p+1 covers k = 0..3
      ve[4 * (1 - h)] = nE0; ve[4 * (1 - h) + 1] = nE1; ve[4 * (1 - h) + 2] = nE2; ve[4 * (1 - h) + 3] = nE3;
    }
    // step 8 (Mdct.cs:261-312) for iterations i8 = p (reads ve, e = n2-8-8p) and i8 = n/16-1-p (reads vd, e = 8p)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* vv = h == 0 ? ve : vd;
      const int i8 = h == 0 ? p : (G::n >> 4) - 1 - p;
      const int b = G::n2 - 8 - 8 * i8;
      float4 b_lo, b_hi;  // B[b .. b+3], B[b+4 .. b+7]
      if constexpr (kPre) {
        b_lo = pbl[it < ITERO ? it : 0][h];
        b_hi = pbh[it < ITERO ? it : 0][h];
      } else {
        b_lo = *reinterpret_cast<const float4*>(B + b);
        b_hi = *reinterpret_cast<const float4*>(B + b + 4);
      }
      float p0, p1, p2, p3;
      float4 o0, o1, o2, o3;
      p3 = vv[6] * b_hi.w - vv[7] * b_hi.z;
      p2 = -vv[6] * b_hi.z - vv[7] * b_hi.w;
      o0.x = p3; o1.w = -p3; o2.x = p2; o3.w = p2;
      p1 = vv[4] * b_hi.y - vv[5] * b_hi.x;
      p0 = -vv[4] * b_hi.x - vv[5] * b_hi.y;
      o0.y = p1; o1.z = -p1; o2.y = p0; o3.z = p0;
      p3 = vv[2] * b_lo.w - vv[3] * b_lo.z;
      p2 = -vv[2] * b_lo.z - vv[3] * b_lo.w;
      o0.z = p3; o1.y = -p3; o2.z = p2; o3.y = p2;
      p1 = vv[0] * b_lo.y - vv[1] * b_lo.x;
      p0 = -vv[0] * b_lo.x - vv[1] * b_lo.y;
      o0.w = p1; o1.x = -p1; o2.w = p0; o3.x = p0;
      const int d0 = 4 * i8, d1 = G::n2 - 4 - 4 * i8, d2 = G::n2 + 4 * i8, d3 = G::n - 4 - 4 * i8;
      if (WIN) {
        const float4 w0 = *reinterpret_cast<const float4*>(w + d0), w1 = *reinterpret_cast<const float4*>(w + d1);
        const float4 w2 = *reinterpret_cast<const float4*>(w + d2), w3 = *reinterpret_cast<const float4*>(w + d3);
        o0 = make_float4(o0.x * w0.x, o0.y * w0.y, o0.z * w0.z, o0.w * w0.w);
        o1 = make_float4(o1.x * w1.x, o1.y * w1.y, o1.z * w1.z, o1.w * w1.w);
        o2 = make_float4(o2.x * w2.x, o2.y * w2.y, o2.z * w2.z, o2.w * w2.w);
        o3 = make_float4(o3.x * w3.x, o3.y * w3.y, o3.z * w3.z, o3.w * w3.w);
      }
      // (slot = 8 it + 4 h + q: callers that keep chunks in registers tell the pair-index iterations apart; q is slot & 3)
      sink(8 * it + 4 * h + 0, d0, o0);
      sink(8 * it + 4 * h + 1, d1, o1);
      sink(8 * it + 4 * h + 2, d2, o2);
      sink(8 * it + 4 * h + 3, d3, o3);
    }
  }
  IMDCT_T(4);
#undef IMDCT_T
}

// ---- latency-optimised form for n <= 2048 ---------------------------------------------------------------
// For N <= 512 complex points every pass has at most one set per lane, so the transform is straight-line code
// per lane.  A wave spends most of its life parked on memory round trips (rocprof: SQ_WAIT_ANY ~63 % of wave
// cycles in the looped form, each phase fetching its own tables right before use); here every table access
// whose address depends only on the lane -- step-0 twiddles, the radix-pass twiddles, _c, _b and the window --
// is issued up front, next to the spectrum load, so that one memory latency covers all of them.
template <int LD>
struct PassPlan {  // stages D = N/2 .. 8 are LD-5 radix-2 stages: first pass takes 3 (or all), second the rest
  static constexpr int TOTAL = LD - 5;
  static constexpr int R1 = TOTAL >= 3 ? 3 : TOTAL;
  static constexpr int S1 = 8 << (TOTAL - R1);
  static constexpr int R2 = TOTAL - R1;  // 0..3 (LD <= 11)
  static constexpr int S2 = 8;
  static constexpr int NSETS1 = (1 << (LD - 2)) >> R1;
  static constexpr int NSETS2 = R2 > 0 ? ((1 << (LD - 2)) >> R2) : 1;
  static constexpr int TW1 = 2 * ((1 << R1) - 1);
  static constexpr int TW2 = R2 > 0 ? 2 * ((1 << R2) - 1) : 0;
};

template <int LD, int R, int S, int NT>
__device__ __forceinline__ void radix_pass_regs(float2* __restrict__ l2, const float (&tw)[NT], int s, bool on) {
  using G = Geo<LD>;
  constexpr int K = 1 << R;
  if (!on) return;
  const int r = s & (S - 1);
  const int blk = s / S;
  const int base = blk * (S << R) + r;
  float2 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = l2[phys(base + S * k)];
#pragma unroll
  for (int st = R - 1; st >= 0; --st) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (k & (1 << st)) continue;
      const int lo = k, hi = k | (1 << st);
      const int pi = (K - (2 << st)) + (k & ((1 << st) - 1));
      const float a0 = tw[2 * pi], a1 = tw[2 * pi + 1];
      const float d1 = v[hi].y - v[lo].y;
      const float d0 = v[hi].x - v[lo].x;
      v[hi].y = v[hi].y + v[lo].y;
      v[hi].x = v[hi].x + v[lo].x;
      v[lo].y = d1 * a0 - d0 * a1;
      v[lo].x = d0 * a0 + d1 * a1;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) l2[phys(base + S * k)] = v[k];
  (void)G::n;
}

template <int LD, bool WIN, typename Sink>
__device__ __forceinline__ void imdct_wave_fast(const float* X, const float* __restrict__ w, float* lds,
                                                const float* __restrict__ A, const float* __restrict__ B,
                                                const float* __restrict__ C, const float* __restrict__ TW, int lane,
                                                Sink sink) {
  static_assert(LD >= 8 && LD <= 11, "single-set-per-lane form");
  using G = Geo<LD>;
  using P = PassPlan<LD>;
  float2* l2 = reinterpret_cast<float2*>(lds);
  constexpr int J = (G::n8 + 63) / 64;  // step-0 items per lane

  // ---- every global load whose address is known now ----
  float4 x[J];
  float2 a_lo[J], a_hi[J];
#pragma unroll
  for (int r = 0; r < J; ++r) {
    const int j = lane + 64 * r;
    const int jc = j < G::n8 ? j : G::n8 - 1;
    x[r] = reinterpret_cast<const float4*>(X)[jc];
    a_lo[r] = reinterpret_cast<const float2*>(A)[jc];
    a_hi[r] = reinterpret_cast<const float2*>(A)[G::n4 - 1 - jc];
  }
  float tw1[P::TW1];
  const int s1 = lane < P::NSETS1 ? lane : P::NSETS1 - 1;
#pragma unroll
  for (int j = 0; j < P::TW1; ++j) tw1[j] = TW[j * P::NSETS1 + s1];
  float tw2[P::TW2 > 0 ? P::TW2 : 1];
  const int s2 = lane < P::NSETS2 ? lane : P::NSETS2 - 1;
  if (P::R2 > 0) {
    const float* TW2p = TW + P::TW1 * P::NSETS1;
#pragma unroll
    for (int j = 0; j < P::TW2; ++j) tw2[j] = TW2p[j * P::NSETS2 + s2];
  }
  // ---- step 0 (Mdct.cs:74-97) ----
#pragma unroll
  for (int r = 0; r < J; ++r) {
    const int j = lane + 64 * r;
    if (j < G::n8) {
      float2 hi, lo;
      hi.y = (x[r].x * a_lo[r].x - x[r].z * a_lo[r].y);
      hi.x = (x[r].x * a_lo[r].y + x[r].z * a_lo[r].x);
      lo.y = (-x[r].w * a_hi[r].x - -x[r].y * a_hi[r].y);
      lo.x = (-x[r].w * a_hi[r].y + -x[r].y * a_hi[r].x);
      l2[phys(G::N - 1 - j)] = hi;
      l2[phys(j)] = lo;
    }
  }
  // second wave of table loads: one phase of lead time is enough, and the step-0 registers are free now
  const float A2 = A[G::n >> 3];
  const bool pon = lane < (G::n >> 5);
  const int p = pon ? lane : (G::n >> 5) - 1;
  const float4 cc0 = reinterpret_cast<const float4*>(C)[2 * p];
  const float4 cc1 = reinterpret_cast<const float4*>(C)[2 * p + 1];
  const int i8a = p, i8b = (G::n >> 4) - 1 - p;
  const int ba = G::n2 - 8 - 8 * i8a, bb = G::n2 - 8 - 8 * i8b;
  const float4 ba_lo = *reinterpret_cast<const float4*>(B + ba), ba_hi = *reinterpret_cast<const float4*>(B + ba + 4);
  const float4 bb_lo = *reinterpret_cast<const float4*>(B + bb), bb_hi = *reinterpret_cast<const float4*>(B + bb + 4);
  const int da[4] = {4 * i8a, G::n2 - 4 - 4 * i8a, G::n2 + 4 * i8a, G::n - 4 - 4 * i8a};
  const int db[4] = {4 * i8b, G::n2 - 4 - 4 * i8b, G::n2 + 4 * i8b, G::n - 4 - 4 * i8b};
  wave_sync();
  radix_pass_regs<LD, P::R1, P::S1>(l2, tw1, s1, lane < P::NSETS1);
  float4 wa[4], wb[4];
  if (WIN) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wa[q] = *reinterpret_cast<const float4*>(w + da[q]);
      wb[q] = *reinterpret_cast<const float4*>(w + db[q]);
    }
  }

  wave_sync();
  if (P::R2 > 0) {
    radix_pass_regs<LD, (P::R2 > 0 ? P::R2 : 1), P::S2>(l2, tw2, s2, lane < P::NSETS2);
    wave_sync();
  }

  // ---- D = 4, 2, 1 (Mdct.cs:463-535) ----
  if (lane < (G::N >> 3)) {
    const int q = lane;
    const int c0 = G::N - 8 - 8 * q;
    float e[16];
    float4* p4 = reinterpret_cast<float4*>(l2 + phys(c0));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 t = p4[j];
      e[4 * j] = t.x; e[4 * j + 1] = t.y; e[4 * j + 2] = t.z; e[4 * j + 3] = t.w;
    }
#define U(k) e[15 - (k)]
    float k00, k11;
    k00 = U(0) - U(8);  k11 = U(1) - U(9);
    U(0) = U(0) + U(8); U(1) = U(1) + U(9);
    U(8) = k00;         U(9) = k11;
    k00 = U(2) - U(10); k11 = U(3) - U(11);
    U(2) = U(2) + U(10); U(3) = U(3) + U(11);
    U(10) = (k00 + k11) * A2; U(11) = (k11 - k00) * A2;
    k00 = U(12) - U(4); k11 = U(5) - U(13);
    U(4) = U(4) + U(12); U(5) = U(5) + U(13);
    U(12) = k11;        U(13) = k00;
    k00 = U(14) - U(6); k11 = U(7) - U(15);
    U(6) = U(6) + U(14); U(7) = U(7) + U(15);
    U(14) = (k00 + k11) * A2; U(15) = (k00 - k11) * A2;
#undef U
    iter_54_regs(e, 15);
    iter_54_regs(e, 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) p4[j] = make_float4(e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]);
  }
  wave_sync();

  // ---- steps 4-6 (bit reversal), 7, 8 (Mdct.cs:189-312) ----
  if (pon) {
    const float* lf = lds;
    float vd[8], ve[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 2 * p + h;
      const int ir = (G::n >> 4) - 1 - i;
      const int kE0 = (int)(__brev((unsigned)(2 * i)) >> (32 - (LD - 3))) << 2;
      const int kE1 = (int)(__brev((unsigned)(2 * i + 1)) >> (32 - (LD - 3))) << 2;
      const int kD0 = (int)(__brev((unsigned)(2 * ir)) >> (32 - (LD - 3))) << 2;
      const int kD1 = (int)(__brev((unsigned)(2 * ir + 1)) >> (32 - (LD - 3))) << 2;
      const float2 e0 = *reinterpret_cast<const float2*>(lf + 2 * phys(kE0 >> 1));
      const float2 e1 = *reinterpret_cast<const float2*>(lf + 2 * phys(kE1 >> 1));
      const float2 g0 = *reinterpret_cast<const float2*>(lf + 2 * phys((kD0 >> 1) + 1));
      const float2 g1 = *reinterpret_cast<const float2*>(lf + 2 * phys((kD1 >> 1) + 1));
      float vD0 = g1.y, vD1 = g1.x, vD2 = g0.y, vD3 = g0.x;
      float vE0 = e1.y, vE1 = e1.x, vE2 = e0.y, vE3 = e0.x;
      const float4 cc = h == 0 ? cc0 : cc1;
      float a02, a11, b0, b1, b2, b3;
      a02 = vD0 - vE2;
      a11 = vD1 + vE3;
      b0 = cc.y * a02 + cc.x * a11;
      b1 = cc.y * a11 - cc.x * a02;
      b2 = vD0 + vE2;
      b3 = vD1 - vE3;
      const float nD0 = b2 + b0, nD1 = b3 + b1, nE2 = b2 - b0, nE3 = b1 - b3;
      a02 = vD2 - vE0;
      a11 = vD3 + vE1;
      b0 = cc.w * a02 + cc.z * a11;
      b1 = cc.w * a11 - cc.z * a02;
      b2 = vD2 + vE0;
      b3 = vD3 - vE1;
      const float nD2 = b2 + b0, nD3 = b3 + b1, nE0 = b2 - b0, nE1 = b1 - b3;
      vd[4 * h] = nD0; vd[4 * h + 1] = nD1; vd[4 * h + 2] = nD2; vd[4 * h + 3] = nD3;
      ve[4 * (1 - h)] = nE0; ve[4 * (1 - h) + 1] = nE1; ve[4 * (1 - h) + 2] = nE2; ve[4 * (1 - h) + 3] = nE3;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* vv = h == 0 ? ve : vd;
      const float4 b_lo = h == 0 ? ba_lo : bb_lo;
      const float4 b_hi = h == 0 ? ba_hi : bb_hi;
      float p0, p1, p2, p3;
      float4 o0, o1, o2, o3;
      p3 = vv[6] * b_hi.w - vv[7] * b_hi.z;
      p2 = -vv[6] * b_hi.z - vv[7] * b_hi.w;
      o0.x = p3; o1.w = -p3; o2.x = p2; o3.w = p2;
      p1 = vv[4] * b_hi.y - vv[5] * b_hi.x;
      p0 = -vv[4] * b_hi.x - vv[5] * b_hi.y;
      o0.y = p1; o1.z = -p1; o2.y = p0; o3.z = p0;
      p3 = vv[2] * b_lo.w - vv[3] * b_lo.z;
      p2 = -vv[2] * b_lo.z - vv[3] * b_lo.w;
      o0.z = p3; o1.y = -p3; o2.z = p2; o3.y = p2;
      p1 = vv[0] * b_lo.y - vv[1] * b_lo.x;
      p0 = -vv[0] * b_lo.x - vv[1] * b_lo.y;
      o0.w = p1; o1.x = -p1; o2.w = p0; o3.x = p0;
      if (WIN) {
        const float4 w0 = h == 0 ? wa[0] : wb[0], w1 = h == 0 ? wa[1] : wb[1];
        const float4 w2 = h == 0 ? wa[2] : wb[2], w3 = h == 0 ? wa[3] : wb[3];
        o0 = make_float4(o0.x * w0.x, o0.y * w0.y, o0.z * w0.z, o0.w * w0.w);
        o1 = make_float4(o1.x * w1.x, o1.y * w1.y, o1.z * w1.z, o1.w * w1.w);
        o2 = make_float4(o2.x * w2.x, o2.y * w2.y, o2.z * w2.z, o2.w * w2.w);
        o3 = make_float4(o3.x * w3.x, o3.y * w3.y, o3.z * w3.z, o3.w * w3.w);
      }
      const int* dd = h == 0 ? da : db;
      sink(4 * h + 0, dd[0], o0);
      sink(4 * h + 1, dd[1], o1);
      sink(4 * h + 2, dd[2], o2);
      sink(4 * h + 3, dd[3], o3);
    }
  }
}

// In-place form: out may alias X (all of X is consumed by step 0 before anything is stored).
// COMPACT: store only the two independent quarters of the (un-windowed) result -- out[0, n/4) and
// out[n/2, 3n/4); the other two follow from out[n/2-1-x] = -out[x] and out[n-1-x] = out[n/2+x] (Mdct.cs:275-303)
// and are rebuilt, windowed, by k_ola_compact.  Halves the bytes this kernel writes and the next one reads.
// LEAN: the looped form with the spectrum preloaded (fits 64 VGPRs; X may alias the wavefront's LDS slice) instead
// of the everything-prefetched form (~100 VGPRs): for callers that bring their own occupancy (kernels_spectrum.hip).
template <int LD, bool WIN, bool COMPACT = false, bool LEAN = false, bool WGSYNC = false, bool PRESYNC = false, bool PF = false>
__device__ __forceinline__ void imdct_wave(const float* X, float* out, const float* __restrict__ w, float* lds,
                                           const float* __restrict__ A, const float* __restrict__ B,
                                           const float* __restrict__ C, const float* __restrict__ TW, int lane,
                                           long long* stamp = nullptr, int dbg_skip = 0) {
  auto sink = [=](int slot, int idx, float4 v) {
#ifdef NVH_ABL_NO_PLANE
    if (v.x == 1.2345e-30f)  // (ablation build: the plane stores left out, the arithmetic kept alive)
#endif
#ifdef NVH_PLANE_NT
    if (!COMPACT || (slot & 1) == 0) pcm_store4(reinterpret_cast<float4*>(out + idx), v.x, v.y, v.z, v.w);
#else
    if (!COMPACT || (slot & 1) == 0) *reinterpret_cast<float4*>(out + idx) = v;
#endif
  };
  if constexpr (LEAN)
    imdct_wave_sink<LD, WIN, decltype(sink), true, WGSYNC, PRESYNC, PF>(X, w, lds, A, B, C, TW, lane, sink, stamp, dbg_skip);
  else if constexpr (LD <= 11)
    imdct_wave_fast<LD, WIN>(X, w, lds, A, B, C, TW, lane, sink);
  else
    imdct_wave_sink<LD, WIN>(X, w, lds, A, B, C, TW, lane, sink);
}

}  // namespace
