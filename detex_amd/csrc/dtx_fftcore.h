// Core of the frequency-domain engine, shared by the scan (dtx_fft.hip) and the all-pairs
// cross-correlation (dtx_xcfft.hip): complex helpers, the wave-decimated 8192-point transform
// in LDS and the block-spectra kernel.  See dtx_fft.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>

#include "dtx_hankel.h"
#include "dtx_internal.h"

namespace {


constexpr int kN = DTX_FFT_N;        // transform length (complex points)
constexpr int kT = 512;              // threads per workgroup, 16 points each
constexpr int kLdsC = kN;            // complex slots of the (swizzled, unpadded) LDS image
// one stored half spectrum of a real series: slots f = 1 .. kN/2 - 1 hold X[f]; slot 0 holds
// (X[0], X[kN/2]), both real; X[kN - f] = conj X[f] is never stored
constexpr int kXh = kN / 2;
constexpr int kXlMax = 3;            // channels whose block spectra fit in LDS beside the image
// L2-sized tiles of the (unit, block) work space: the workgroups an XCD runs together
// share kTileU units' spectra and kTileB blocks' spectra (see ds_fft_kernel)
constexpr int kTileU = 1, kTileB = 16 / 1, kTile = kTileU * kTileB;
static_assert(kN == 8192, "radix plan 16 x 8 x 8 x 8 is for 8192 points");

typedef float2 c32;

__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (SG * i)
template <int SG>
__device__ __forceinline__ c32 muli(c32 a) {
  return SG > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// In-place radix-2 decimation-in-frequency DFT of R = 8 or 16 register values, kernel
// e^{SG 2 pi i s q / R}; the result is left in BIT-REVERSED order (X[q] = u[rev(q)]), which
// costs nothing: the callers index their stores with compile-time constants.  Working in
// place stage by stage keeps the live set at R values plus one temporary -- the natural-
// order split-radix form needed 238 registers for the 8192-point transform, this one fits
// the 128 of two workgroups per CU.
template <int SG, int LEN, int Q>
__device__ __forceinline__ c32 mul_root(c32 t) {   // t * e^{SG 2 pi i Q / LEN}
  if constexpr (Q == 0) {
    return t;
  } else if constexpr (4 * Q == LEN) {
    return muli<SG>(t);
  } else if constexpr (8 * Q == LEN) {             // (1 + SG i) / sqrt 2
    constexpr float h = 0.70710678118654752440f;
    return SG > 0 ? make_float2((t.x - t.y) * h, (t.y + t.x) * h)
                  : make_float2((t.x + t.y) * h, (t.y - t.x) * h);
  } else if constexpr (8 * Q == 3 * LEN) {         // (-1 + SG i) / sqrt 2
    constexpr float h = 0.70710678118654752440f;
    return SG > 0 ? make_float2((-t.x - t.y) * h, (t.x - t.y) * h)
                  : make_float2((t.y - t.x) * h, (-t.x - t.y) * h);
  } else {
    static_assert(LEN == 16, "general roots only in the 16-point stage");
    constexpr float cs[8] = {1.f, 0.92387953251128675613f, 0.70710678118654752440f,
                             0.38268343236508977173f, 0.f, -0.38268343236508977173f,
                             -0.70710678118654752440f, -0.92387953251128675613f};
    constexpr float sn[8] = {0.f, 0.38268343236508977173f, 0.70710678118654752440f,
                             0.92387953251128675613f, 1.f, 0.92387953251128675613f,
                             0.70710678118654752440f, 0.38268343236508977173f};
    return cmul(t, make_float2(cs[Q], (float)SG * sn[Q]));
  }
}
template <int SG, int R, int LEN>
__device__ __forceinline__ void dif_stage(c32 (&u)[R]) {
  constexpr int H = LEN / 2;
  dtx::static_for<0, R / 2>([&](auto bc) {
    constexpr int b = bc, blk = (b / H) * LEN, q = b % H;
    const c32 x = u[blk + q], y = u[blk + q + H];
    u[blk + q] = cadd(x, y);
    u[blk + q + H] = mul_root<SG, LEN, q>(csub(x, y));
  });
}
template <int SG, int R>
__device__ __forceinline__ void dft_bitrev(c32 (&u)[R]) {
  if constexpr (R >= 16) dif_stage<SG, R, 16>(u);
  if constexpr (R >= 8) dif_stage<SG, R, 8>(u);
  dif_stage<SG, R, 4>(u);
  dif_stage<SG, R, 2>(u);
}
// (x, y) -> (x + T y, x - T y) in six fused multiply-adds (the difference as 2 x - (x + T y))
__device__ __forceinline__ void bfly_tw(c32& x, c32& y, c32 T) {
  const float px = __fmaf_rn(T.x, y.x, __fmaf_rn(-T.y, y.y, x.x));
  const float py = __fmaf_rn(T.x, y.y, __fmaf_rn(T.y, y.x, x.y));
  y = make_float2(__fmaf_rn(2.f, x.x, -px), __fmaf_rn(2.f, x.y, -py));
  x = make_float2(px, py);
}
// t * e^{SG 2 pi i E / R}, 0 <= E < R / 2: first-quadrant roots by mul_root, the second quadrant as
// i times those (a register swap and a sign the consumer's fma absorbs)
template <int SG, int R, int E>
__device__ __forceinline__ c32 rot_root(c32 t) {
  if constexpr (4 * E >= R) return muli<SG>(rot_root<SG, R, E - R / 4>(t));
  else return mul_root<SG, R, E>(t);
}
// Twiddled radix-R DFT: X[q] = sum_s (u[s] w^s) e^{SG 2 pi i s q / R}, left bit-reversed
// (X[q] = u[rev(q)]) like dft_bitrev.  wp[j] = w^(R >> (j + 1)): w^(R/2), .., w^2, w -- the only
// powers needed.  Same data flow as the decimation-in-frequency stages of dft_bitrev, but the
// inter-pass twiddles w^s are folded into the butterflies: stage LEN, block j multiplies the lower
// input by T = w^(LEN/2) e^{SG 2 pi i a LEN / (2 R)} (a = the block's path, bit-reversed j) and
// forms x +- T y in six fused multiply-adds; nothing is multiplied afterwards.  Against
// "multiply the inputs by w^s, then dft_bitrev": 208 instead of 272 instructions for R = 16
// (60 twiddle products + 44 to derive w^3 .. w^15 + 168 butterfly), 76 instead of ~115 for R = 8.
template <int SG, int R, int LEN, int STAGE, int LG>
__device__ __forceinline__ void tw_stage(c32 (&u)[R], const c32 (&wp)[LG]) {
  constexpr int H = LEN / 2, NB = R / LEN;
  dtx::static_for<0, NB>([&](auto jc) {
    constexpr int j = jc;
    // a = sum_t j_t R / (2 LEN 2^t): the odd branches taken on the way down, most recent first
    constexpr int a = [] { int s = 0; for (int t = 0; (1 << t) < NB; ++t) s += ((j >> t) & 1) * (R / (2 * LEN) >> t); return s; }();
    constexpr int e = (a * H) % R;
    const c32 T = rot_root<SG, R, e>(wp[STAGE]);
    dtx::static_for<0, H>([&](auto qc) {
      constexpr int q = qc;
      bfly_tw(u[j * LEN + q], u[j * LEN + q + H], T);
    });
  });
}
template <int SG, int R, int LG>
__device__ __forceinline__ void dft_tw_bitrev(c32 (&u)[R], const c32 (&wp)[LG]) {
  static_assert((1 << LG) == R, "wp holds log2 R powers");
  if constexpr (R >= 16) tw_stage<SG, R, 16, LG - 4, LG>(u, wp);
  if constexpr (R >= 8) tw_stage<SG, R, 8, LG - 3, LG>(u, wp);
  tw_stage<SG, R, 4, LG - 2, LG>(u, wp);
  tw_stage<SG, R, 2, LG - 1, LG>(u, wp);
}
// (x, y) -> (x + T y, x - T y) for the constant T = e^{SG 2 pi i E / R}, 0 <= E < R / 2: four adds
// for T = 1, i; six instructions otherwise (the decimation-in-frequency form pays four adds and a
// four-instruction rotation of the difference)
template <int SG, int R, int E>
__device__ __forceinline__ void bfly_const(c32& x, c32& y) {
  constexpr float h = 0.70710678118654752440f;
  if constexpr (E == 0) {
    const c32 t = y;
    y = csub(x, t);
    x = cadd(x, t);
  } else if constexpr (4 * E == R) {
    const c32 t = muli<SG>(y);
    y = csub(x, t);
    x = cadd(x, t);
  } else if constexpr (8 * E == R) {               // T = (1 + SG i) / sqrt 2
    const float d1 = SG > 0 ? y.x - y.y : y.x + y.y, d2 = SG > 0 ? y.y + y.x : y.y - y.x;
    y = make_float2(__fmaf_rn(-h, d1, x.x), __fmaf_rn(-h, d2, x.y));
    x = make_float2(__fmaf_rn(h, d1, x.x), __fmaf_rn(h, d2, x.y));
  } else if constexpr (8 * E == 3 * R) {           // T = (-1 + SG i) / sqrt 2
    const float d1 = SG > 0 ? y.x + y.y : y.x - y.y, d2 = SG > 0 ? y.y - y.x : y.y + y.x;   // T y = -h (d1, d2)
    y = make_float2(__fmaf_rn(h, d1, x.x), __fmaf_rn(h, d2, x.y));
    x = make_float2(__fmaf_rn(-h, d1, x.x), __fmaf_rn(-h, d2, x.y));
  } else {
    static_assert(R == 16, "general constant roots only in the 16-point transform");
    constexpr float cs[8] = {1.f, 0.92387953251128675613f, 0.70710678118654752440f,
                             0.38268343236508977173f, 0.f, -0.38268343236508977173f,
                             -0.70710678118654752440f, -0.92387953251128675613f};
    constexpr float sn[8] = {0.f, 0.38268343236508977173f, 0.70710678118654752440f,
                             0.92387953251128675613f, 1.f, 0.92387953251128675613f,
                             0.70710678118654752440f, 0.38268343236508977173f};
    bfly_tw(x, y, make_float2(cs[E], (float)SG * sn[E]));
  }
}
// dft_bitrev by the same route: X[q] = sum_s u[s] e^{SG 2 pi i s q / R} at u[rev(q)], 148 instead
// of 168 instructions for R = 16
template <int SG, int R, int LEN>
__device__ __forceinline__ void const_stage(c32 (&u)[R]) {
  constexpr int H = LEN / 2, NB = R / LEN;
  dtx::static_for<0, NB>([&](auto jc) {
    constexpr int j = jc;
    constexpr int a = [] { int s = 0; for (int t = 0; (1 << t) < NB; ++t) s += ((j >> t) & 1) * (R / (2 * LEN) >> t); return s; }();
    constexpr int e = (a * H) % R;
    dtx::static_for<0, H>([&](auto qc) {
      constexpr int q = qc;
      bfly_const<SG, R, e>(u[j * LEN + q], u[j * LEN + q + H]);
    });
  });
}
template <int SG, int R>
__device__ __forceinline__ void dft_const_bitrev(c32 (&u)[R]) {
  if constexpr (R >= 16) const_stage<SG, R, 16>(u);
  if constexpr (R >= 8) const_stage<SG, R, 8>(u);
  const_stage<SG, R, 4>(u);
  const_stage<SG, R, 2>(u);
}
// the untwiddled 16-point pass
template <int SG>
__device__ __forceinline__ void dft16_plain(c32 (&v)[16]) {
  dft_const_bitrev<SG, 16>(v);
}
__host__ __device__ constexpr int rev2(int q) { return ((q & 1) << 1) | (q >> 1); }
__host__ __device__ constexpr int rev3(int q) { return ((q & 1) << 2) | (q & 2) | ((q >> 2) & 1); }
__host__ __device__ constexpr int rev4(int q) {
  return ((q & 1) << 3) | ((q & 2) << 1) | ((q & 4) >> 1) | ((q >> 3) & 1);
}

// The 8192-point transform, decimated over the workgroup's 8 waves:  n = 8 n2 + w.
//   phase A  wave w transforms ITS 1024-point subsequence z[8 n2 + w] alone: 16 points per
//            lane (lane l, register r <-> n2 = l + 64 r), Stockham radices 16 x 16 x 4, two
//            exchanges through the wave's own 8 KB of the LDS image.  A wave's LDS
//            instructions execute in order, so these exchanges need no s_barrier: the waves
//            of a workgroup drift apart and one wave's L2 loads overlap another's butterflies.
//   phase B  one exchange across the waves (Y_w[k2] at slot 1024 w + k2), then thread tid
//            combines Y_0..7[k2] for k2 = tid, tid + 512 with twiddles e^{2 pi i w k2 / N} in a
//            radix-8 pass: X[k2 + 1024 k1] -> register m = 2 k1 + h holds point tid + 512 m,
//            which is what the epilogue's coalesced DS stores want.
// Two workgroup barriers per transform (before the first scatter: the previous transform's
// phase-B reads are done; before the phase-B reads) instead of seven.
//
// LDS image: in-region slot a lives at a ^ g(a), g(a) = ((a >> 4) & 15) ^ (((a >> 8) & 1) << 4):
// an XOR swizzle instead of padding (the image is exactly 64 KB, which is what lets three
// channels of block spectra sit beside it in the 160 KB).  Every access is conflict-free
// under it in the bank model of MI355X_MICROARCH.md -- 8-byte stores go in groups of 16 contiguous
// lanes over 32 banks, ds_read_b64 in half-waves over 64 banks, ds_read2st64_b64 again in groups of
// 16 over 32 banks (checked exhaustively by profiles/attic/fft_swizzle_check2.py).  Rounds 1-2 used
// (a >> 5) & 15, checked every access as half-waves on 64 banks: under the real model their first
// scatter (slot 16 l + s) put lanes 2 j and 2 j + 1 of a 16-lane group on one bank pair -- the 13 % of
// conflict cycles rocprofv3 counted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  Every address is a
// per-lane base combined with a compile-time constant:
//   gather / phase-B write   point l + 64 r             -> (g ^ (4 (r & 3) | 16 ((r >> 2) & 1))) + 64 r
//   scatter 1                slot 16 l + s              -> s1 ^ s
//   scatter 2                256 (l >> 4) + (l & 15) + 16 s  -> s2 ^ (16 s ^ s)
//   phase-B read             region w', point tid + 512 h    -> f + 512 h + 1024 w'
struct LdsBases {
  int g, s1, s2, f;
};
__device__ __forceinline__ LdsBases lds_bases(int tid) {
  const int l = tid & 63, region = (tid >> 6) * 1024, hb = (l >> 4) & 1;
  LdsBases b;
  b.g = region + (l ^ (l >> 4));
  b.s1 = region + ((16 * l) ^ ((l & 15) ^ (hb << 4)));
  b.s2 = region + ((256 * (l >> 4) + (l & 15)) ^ (16 * hb));
  b.f = tid ^ (((tid >> 4) & 15) ^ (((tid >> 8) & 1) << 4));
  return b;
}
__host__ __device__ constexpr int swz_gr(int r) { return ((r & 3) << 2) | (16 * ((r >> 2) & 1)); }

// LDS element at a BYTE offset: the swizzled addresses are formed in bytes (base << 3 once, then
// XOR / add of compile-time byte constants), so no per-access shift is left for the VALU
__device__ __forceinline__ c32& lds_at(c32* lds, int byte_off) {
  return *reinterpret_cast<c32*>(reinterpret_cast<char*>(lds) + byte_off);
}

// a wave's LDS writes are visible to its own later reads (in-order LDS pipeline); this only
// keeps hipcc from moving the accesses across the exchange point
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// twiddles of one thread, already conjugated for SG < 0
struct Twid {
  c32 a[4];    // pass 2 (radix 16, sub-transforms of 16): w, w^2, w^4, w^8, w = e^{2 pi i (l & 15) / 256}
  c32 b[2];    // pass 3 (radix 4, sub-transforms of 256), butterfly 0: w, w^2, w = e^{2 pi i l / 1024}
  c32 p4[3];   // phase B (radix 8): w, w^2, w^4, w = e^{2 pi i tid / 8192}; point tid + 512 derives its own
};
template <int SG>
__device__ __forceinline__ void load_twiddles(Twid& w, const c32* __restrict__ tab, int tid) {
  auto ld = [&](int idx) {
    c32 t = tab[idx];
    if (SG < 0) t.y = -t.y;
    return t;
  };
  const int l = tid & 63;
#pragma unroll
  for (int e = 0; e < 4; ++e) w.a[e] = ld(((l & 15) * (kN / 256)) << e);
  w.b[0] = ld(l * (kN / 1024));
  w.b[1] = ld(l * (kN / 512));
#pragma unroll
  for (int e = 0; e < 3; ++e) w.p4[e] = ld(tid << e);
}

// one radix-8 butterfly of phase B: points v[2 s + H], s = 0..7, twiddled by w^s; the
// transform is left bit-reversed in u
template <int SG, int H>
__device__ __forceinline__ void bfly8(const c32 (&v)[16], c32 (&u)[8], c32 w1, c32 w2, c32 w4) {
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx) u[sidx] = v[2 * sidx + H];
  const c32 wp[3] = {w4, w2, w1};
  dft_tw_bitrev<SG, 8>(u, wp);
}

// a <-> b transposed over (which of the two registers, lane bit 4 / lane bit 5)
__device__ __forceinline__ void lane_swap16(c32& a, c32& b) {
  auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
  b = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
__device__ __forceinline__ void lane_swap32(c32& a, c32& b) {
  auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
  b = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// v[r] = point 8 (l + 64 r) + wave in, natural-order transform point (tid + 512 m) out in v[m]
// LEAN: for the 128-register kernels -- each pass loads its twiddles just before it uses them
// (L1 / L2 hits; four waves per SIMD cover the latency) and the two radix-8 butterflies of
// phase B run strictly one after the other, so that at most one set of temporaries is live
// the register-only tail of the transform: the two radix-8 butterflies across the waves, from
// the values the phase-B reads left in v.  Separate so that a kernel can release the LDS image
// (a workgroup barrier) between the reads and this arithmetic.
template <int SG>
__device__ __forceinline__ void fft8192_back_twiddles(c32 (&p4)[3], const c32* __restrict__ tab, int tid) {
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    p4[e] = tab[tid << e];
    if (SG < 0) p4[e].y = -p4[e].y;
  }
}
template <int SG>
__device__ __forceinline__ void fft8192_back_compute(c32 (&v)[16], const c32 (&p4)[3]) {
  // butterfly h = 1 is point tid + 512: its twiddle angle is pi/8 ahead, so w, w^2, w^4 pick
  // up e^{i pi/8}, e^{i pi/4}, e^{i pi/2}
  c32 u[8];
  bfly8<SG, 0>(v, u, p4[0], p4[1], p4[2]);
  dtx::static_for<0, 8>([&](auto sc) { constexpr int s = sc; v[2 * s] = u[rev3(s)]; });
  __builtin_amdgcn_sched_barrier(0);
  bfly8<SG, 1>(v, u, mul_root<SG, 16, 1>(p4[0]), mul_root<SG, 8, 1>(p4[1]), muli<SG>(p4[2]));
  dtx::static_for<0, 8>([&](auto sc) { constexpr int s = sc; v[2 * s + 1] = u[rev3(s)]; });
}
template <int SG>
__device__ __forceinline__ void fft8192_back(c32 (&v)[16], const c32* __restrict__ tab, int tid) {
  int opaque;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
  c32 p4[3];
  fft8192_back_twiddles<SG>(p4, tab, tid + opaque);
  fft8192_back_compute<SG>(v, p4);
}

template <int SG>
__device__ __forceinline__ void fft8192_pass1(c32 (&v)[16]) { dft16_plain<SG>(v); }
// p1: where pass 1 runs -- 0 here, before the barrier; 1 the caller has run it (fft8192_pass1)
template <int SG, bool LEAN = false, bool FRONT_ONLY = false>
__device__ __forceinline__ void fft8192(c32 (&v)[16], c32* lds, const LdsBases& lb0,
                                        const c32* __restrict__ tab, int tid, int p1 = 0) {
  // The ~50 swizzled LDS addresses below and the 9 twiddles are loop invariants of the
  // caller's pair loop; hoisted there they cost 70 registers and the kernel spills.  An opaque
  // zero keeps them inside (the twiddle loads are L1 / L2 hits, 9 per transform).
  int opaque;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
  Twid w;
  auto ldtw = [&](int idx) {
    c32 t = tab[idx];
    if (SG < 0) t.y = -t.y;
    return t;
  };
  if constexpr (!LEAN) load_twiddles<SG>(w, tab, tid + opaque);
  if constexpr (LEAN && FRONT_ONLY) {
    // the front's own twiddles (passes 2 and 3), issued now: pass 1 and the first exchange
    // cover their latency.  12 registers; phase B's three are loaded by fft8192_back
    const int l = (tid + opaque) & 63;
#pragma unroll
    for (int e = 0; e < 4; ++e) w.a[e] = ldtw(((l & 15) * (kN / 256)) << e);
    w.b[0] = ldtw(l * (kN / 1024));
    w.b[1] = ldtw(l * (kN / 512));
  }
  LdsBases lb;   // in bytes from here on
  lb.g = (lb0.g + opaque) << 3; lb.s1 = (lb0.s1 + opaque) << 3; lb.s2 = (lb0.s2 + opaque) << 3;
  lb.f = (lb0.f + opaque) << 3;
  // ---- phase A, pass 1: radix 16 over r, no twiddles
  if (p1 == 0) dft16_plain<SG>(v);
  __syncthreads();   // every wave has finished the phase-B reads of the previous transform
  dtx::static_for<0, 16>([&](auto sc) { constexpr int s = sc; lds_at(lds, lb.s1 ^ (8 * s)) = v[rev4(s)]; });
  wave_sync();
  dtx::static_for<0, 16>([&](auto rc) { constexpr int r = rc; v[r] = lds_at(lds, (lb.g ^ (8 * swz_gr(r))) + 512 * r); });
  wave_sync();
  // ---- pass 2: radix 16, sub-transforms of 16 -> 256
  {
    if constexpr (LEAN && !FRONT_ONLY) {
      const int l15 = ((tid + opaque) & 15) * (kN / 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) w.a[e] = ldtw(l15 << e);
    }
    const c32 wp[4] = {w.a[3], w.a[2], w.a[1], w.a[0]};
    dft_tw_bitrev<SG, 16>(v, wp);
  }
  // Second exchange without LDS.  Output k2 = s of pass 2 sits in v[rev4(s)] of lane
  // l = k1 + 16 x3; pass 3 wants point j = k1 + 16 k2 of sub-sequence x3 in lane j & 63, register
  // (j >> 6) + 4 x3: the low two bits of k2 trade places with lane bits 4 and 5.  gfx950's
  // v_permlane16_swap / v_permlane32_swap transpose exactly (one register bit, lane bit 4 / 5).
  {
    c32 y[16];
    dtx::static_for<0, 16>([&](auto sc) { constexpr int s = sc; y[s] = v[rev4(s)]; });
    dtx::static_for<0, 16>([&](auto sc) {
      constexpr int s = sc;
      if constexpr ((s & 1) == 0) lane_swap16(y[s], y[s + 1]);
    });
    dtx::static_for<0, 16>([&](auto sc) {
      constexpr int s = sc;
      if constexpr ((s & 2) == 0) lane_swap32(y[s], y[s + 2]);
    });
    // y index: bits 1:0 = x3, bits 3:2 = k2 >> 2 = b
    dtx::static_for<0, 16>([&](auto rc) { constexpr int r = rc; v[r] = y[(r >> 2) + 4 * (r & 3)]; });
  }
  // ---- pass 3: radix 4, sub-transforms of 256 -> 1024; butterfly b takes registers b + 4 s
  // and point index l + 64 b: its twiddle angle is b/16 of a turn ahead of butterfly 0's
  if constexpr (LEAN && !FRONT_ONLY) {
    const int l = (tid + opaque) & 63;
    w.b[0] = ldtw(l * (kN / 1024));
    w.b[1] = ldtw(l * (kN / 512));
  }
  dtx::static_for<0, 4>([&](auto bc) {
    constexpr int b = bc;
    const c32 w1 = mul_root<SG, 16, b>(w.b[0]), w2 = mul_root<SG, 8, b>(w.b[1]);
    c32 u[4];
    u[0] = v[b]; u[1] = v[b + 4]; u[2] = v[b + 8]; u[3] = v[b + 12];
    const c32 wp[2] = {w2, w1};
    dft_tw_bitrev<SG, 4>(u, wp);
    dtx::static_for<0, 4>([&](auto qc) { constexpr int q = qc; v[b + 4 * q] = u[rev2(q)]; });
  });
  // ---- phase B: Y_w[k2 = l + 64 r] to the image, then radix 8 across the waves
  dtx::static_for<0, 16>([&](auto rc) { constexpr int r = rc; lds_at(lds, (lb.g ^ (8 * swz_gr(r))) + 512 * r) = v[r]; });
  __syncthreads();
  dtx::static_for<0, 16>([&](auto mc) {
    constexpr int m = mc;
    v[m] = lds_at(lds, lb.f + 8 * (512 * (m & 1) + 1024 * (m >> 1)));
  });
  if constexpr (FRONT_ONLY) return;
  // butterfly h = 1 is point tid + 512: its twiddle angle is pi/8 ahead, so w, w^2, w^4 pick
  // up e^{i pi/8}, e^{i pi/4}, e^{i pi/2}
  if constexpr (LEAN) {
#pragma unroll
    for (int e = 0; e < 3; ++e) w.p4[e] = ldtw((tid + opaque) << e);
    c32 u[8];
    bfly8<SG, 0>(v, u, w.p4[0], w.p4[1], w.p4[2]);
    dtx::static_for<0, 8>([&](auto sc) { constexpr int s = sc; v[2 * s] = u[rev3(s)]; });
    __builtin_amdgcn_sched_barrier(0);
    bfly8<SG, 1>(v, u, mul_root<SG, 16, 1>(w.p4[0]), mul_root<SG, 8, 1>(w.p4[1]), muli<SG>(w.p4[2]));
    dtx::static_for<0, 8>([&](auto sc) { constexpr int s = sc; v[2 * s + 1] = u[rev3(s)]; });
  } else {
    c32 u0[8], u1[8];
    bfly8<SG, 0>(v, u0, w.p4[0], w.p4[1], w.p4[2]);
    bfly8<SG, 1>(v, u1, mul_root<SG, 16, 1>(w.p4[0]), mul_root<SG, 8, 1>(w.p4[1]), muli<SG>(w.p4[2]));
    dtx::static_for<0, 8>([&](auto sc) {
      constexpr int s = sc;
      v[2 * s] = u0[rev3(s)];
      v[2 * s + 1] = u1[rev3(s)];
    });
  }
}

// Two independent transforms at once, each on an image of its own (a and b): the same passes as
// fft8192, written pass by pass for both so that one's LDS round trips and dependent butterflies are
// covered by the other's -- with two waves per SIMD a single transform leaves the vector ALU idle
// most of the time (the cross-correlation kernel ran 5.3 us per transform against 1.7 us of issue).
// Same twiddles (they depend on the lane only), same two workgroup barriers for the pair.
template <int SG>
__device__ __forceinline__ void fft8192_x2(c32 (&va)[16], c32 (&vb)[16], c32* ldsa, c32* ldsb,
                                           const LdsBases& lb0, const c32* __restrict__ tab, int tid) {
  int opaque;
  asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
  Twid w;
  load_twiddles<SG>(w, tab, tid + opaque);
  LdsBases lb;   // in bytes
  lb.g = (lb0.g + opaque) << 3; lb.s1 = (lb0.s1 + opaque) << 3; lb.s2 = (lb0.s2 + opaque) << 3;
  lb.f = (lb0.f + opaque) << 3;
  // ---- pass 1
  dft16_plain<SG>(va);
  dft16_plain<SG>(vb);
  __syncthreads();   // every wave has finished its reads of both images
  dtx::static_for<0, 16>([&](auto sc) {
    constexpr int s = sc;
    lds_at(ldsa, lb.s1 ^ (8 * s)) = va[rev4(s)];
    lds_at(ldsb, lb.s1 ^ (8 * s)) = vb[rev4(s)];
  });
  wave_sync();
  dtx::static_for<0, 16>([&](auto rc) {
    constexpr int r = rc;
    va[r] = lds_at(ldsa, (lb.g ^ (8 * swz_gr(r))) + 512 * r);
    vb[r] = lds_at(ldsb, (lb.g ^ (8 * swz_gr(r))) + 512 * r);
  });
  wave_sync();
  // ---- pass 2: radix 16, sub-transforms of 16 -> 256, then the in-wave exchange by lane swaps
  {
    const c32 wp[4] = {w.a[3], w.a[2], w.a[1], w.a[0]};
    dft_tw_bitrev<SG, 16>(va, wp);
    dft_tw_bitrev<SG, 16>(vb, wp);
  }
  auto exchange = [&](c32 (&v)[16]) {
    c32 y[16];
    dtx::static_for<0, 16>([&](auto sc) { constexpr int s = sc; y[s] = v[rev4(s)]; });
    dtx::static_for<0, 16>([&](auto sc) {
      constexpr int s = sc;
      if constexpr ((s & 1) == 0) lane_swap16(y[s], y[s + 1]);
    });
    dtx::static_for<0, 16>([&](auto sc) {
      constexpr int s = sc;
      if constexpr ((s & 2) == 0) lane_swap32(y[s], y[s + 2]);
    });
    dtx::static_for<0, 16>([&](auto rc) { constexpr int r = rc; v[r] = y[(r >> 2) + 4 * (r & 3)]; });
  };
  exchange(va);
  exchange(vb);
  // ---- pass 3: radix 4, sub-transforms of 256 -> 1024
  auto pass3 = [&](c32 (&v)[16]) {
    dtx::static_for<0, 4>([&](auto bc) {
      constexpr int b = bc;
      const c32 w1 = mul_root<SG, 16, b>(w.b[0]), w2 = mul_root<SG, 8, b>(w.b[1]);
      c32 u[4];
      u[0] = v[b]; u[1] = v[b + 4]; u[2] = v[b + 8]; u[3] = v[b + 12];
      const c32 wp[2] = {w2, w1};
      dft_tw_bitrev<SG, 4>(u, wp);
      dtx::static_for<0, 4>([&](auto qc) { constexpr int q = qc; v[b + 4 * q] = u[rev2(q)]; });
    });
  };
  pass3(va);
  pass3(vb);
  // ---- phase B
  dtx::static_for<0, 16>([&](auto rc) {
    constexpr int r = rc;
    lds_at(ldsa, (lb.g ^ (8 * swz_gr(r))) + 512 * r) = va[r];
    lds_at(ldsb, (lb.g ^ (8 * swz_gr(r))) + 512 * r) = vb[r];
  });
  __syncthreads();
  dtx::static_for<0, 16>([&](auto mc) {
    constexpr int m = mc;
    va[m] = lds_at(ldsa, lb.f + 8 * (512 * (m & 1) + 1024 * (m >> 1)));
    vb[m] = lds_at(ldsb, lb.f + 8 * (512 * (m & 1) + 1024 * (m >> 1)));
  });
  fft8192_back_compute<SG>(va, w.p4);
  fft8192_back_compute<SG>(vb, w.p4);
}

// z += w p and z += w conj(p) in two packed fused multiply-adds each, the operand selection and the signs in the
// instruction's modifiers (no (-w.y, w.y) or swapped copies in registers).  Used by the opt-in 16 384-point lag
// search only: v_pk_*_f32 run at half the instruction rate on gfx950 (the Makefile builds the transform files with
// -fno-slp-vectorize for that reason), and in the scan kernel these forms measured 3.2 against 2.75 ms for the
// compiler's scalar fused multiply-adds (profiles/g26_cmac_ab.txt) -- what they save there is registers, not time.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cmac(f2& z, f2 w, f2 p) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
      : "+v"(z) : "v"(w), "v"(p));
}
__device__ __forceinline__ void cmac_conj(f2& z, f2 w, f2 p) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
      : "+v"(z) : "v"(w), "v"(p));
}
// z = w p / z = w conj(p): the first product of a sum (no zero fill of z)
__device__ __forceinline__ f2 cmul2(f2 w, f2 p) {
  f2 z;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
      : "=&v"(z) : "v"(w), "v"(p));
  return z;
}
__device__ __forceinline__ f2 cmul2_conj(f2 w, f2 p) {
  f2 z;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
      : "=&v"(z) : "v"(w), "v"(p));
  return z;
}

constexpr int kN16 = 16384;
constexpr int kT16 = 1024;
constexpr int kXh16 = kN16 / 2;

// Basis pair spectra in memory: thread (wave w, lane l) of a MAC holds frequencies
// f = 8 (l + 64 r) + w, r = 0 .. 15.  Registers r = 2 j, 2 j + 1 of one lane are adjacent, so a
// lane fetches its 16 values of a channel with eight 16-byte loads (eight 1 KiB wave
// instructions) instead of sixteen 8-byte ones.
__host__ __device__ constexpr int w_slot(int f) {
  const int w = f & 7, n2 = f >> 3, l = n2 & 63, r = n2 >> 6;
  return w * 1024 + (r >> 1) * 128 + 2 * l + (r & 1);
}
// pw[r] = W_c[8 (l + 64 r) + w]; wl = 512 w + l in float4 units from the channel base
__device__ __forceinline__ void w_load16(const c32* __restrict__ wc, int wl, c32 (&pw)[16]) {
  const float4* q = reinterpret_cast<const float4*>(wc);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float4 t = (q + 64 * j)[wl];
    pw[2 * j] = make_float2(t.x, t.y);
    pw[2 * j + 1] = make_float2(t.z, t.w);
  }
}

// ---------------------------------------------------------------- block spectra
// Xh[chunk * nblk + blk][c][slot(f)] = sum_t xp[chunk][c][blk V + t] e^{-2 pi i t f / kN}, f < kN / 2,
// slot(f) = (f & 7) * 512 + (f >> 3): the wave-decimated order the scan kernel reads (wave w
// takes the frequencies f = 8 n2 + w); X[kN / 2] rides in the imaginary part of slot 0.
// keep: samples of a block that can reach a kept output (V + L - 1 for a single-segment
// template: window p < V reads samples p .. p + L - 1); the rest of the 8192 are transformed as
// zeros -- exact, and it keeps a burst that lies only in that tail out of this block's float32
// rounding (the dynamic-range guard looks at the block's own windows).  kN: no masking.
// nsub > 1 (the cross-correlation's lag search): stored block j = blk * nsub + s starts at sample
// blk V + s sub_stride -- the data that segment s of a template cut into segments of sub_stride
// samples meets in lag block blk (advance V != segment length: every (block, segment) has its own spectrum).
__global__ __launch_bounds__(kT) void xhat_kernel(const float* __restrict__ xp, int64_t Tpad,
                                                  int Nc, int V, int nblk, int keep,
                                                  const c32* __restrict__ tab,
                                                  c32* __restrict__ xh, int nsub = 1, int sub_stride = 0) {
  __shared__ c32 lds[kLdsC];
  const int tid = threadIdx.x, blk = blockIdx.x, c = blockIdx.y, chunk = blockIdx.z;
  const LdsBases lb = lds_bases(tid);
  const float* x = xp + ((int64_t)chunk * Nc + c) * Tpad;
  const int tb = 8 * (tid & 63) + (tid >> 6);          // sample within the block
  const int64_t ts = (int64_t)(blk / nsub) * V + (int64_t)(blk % nsub) * sub_stride;   // the block's first sample
  c32 v[16];
  // The transform wants v[r] = sample tb + 512 r: taken from memory directly that is sixteen 4-byte loads a thread
  // whose lanes sit 32 bytes apart (rounds 1-5).  Instead the workgroup reads the block with four coalesced 16-byte
  // loads a thread (dword-aligned: a cross-correlation block starts anywhere) and hands the samples over through
  // the still unused image; element i at word i + (i >> 3): the reads (lane stride 9 words) are conflict-free.
  // Step - scan kernel at configs[1] 0.435 -> 0.410 ms (scripts/probe/ab_xhat.sh, alternating builds on one box).
  {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    float* stage = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * tid + 2048 * k;
      const int64_t t = ts + i;
      float e[4];
      if (t + 3 < Tpad) {
        const f4u q = *reinterpret_cast<const f4u*>(x + t);
        e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = t + j < Tpad ? x[t + j] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) stage[i + j + ((i + j) >> 3)] = i + j < keep ? e[j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = tb + 512 * r;
      v[r] = make_float2(stage[i + (i >> 3)], 0.f);
    }
    __syncthreads();                                   // (the transform's first pass writes the image)
  }
  fft8192<-1>(v, lds, lb, tab, tid);
  // Thread tid holds X[tid + 512 m] in v[m]; slot (f & 7) * 512 + (f >> 3) puts a wave's eight
  // values of one m into eight 64-byte pieces 4 KB apart.  Through the (free) image instead: wave w
  // stores row f & 7 = w, 64 consecutive slots (512 bytes) per instruction.  Image layout
  // [m][tid + (tid >> 3)]: the reads (lane stride 9 elements) are conflict-free.
  c32* row = xh + (((int64_t)chunk * nblk + blk) * Nc + c) * kXh + (tid >> 6) * 512 + (tid & 63);   // nblk: blocks STORED per chunk
  __syncthreads();                                     // every wave's reads of the image are done
#pragma unroll
  for (int m = 0; m < 8; ++m) lds[m * 576 + tid + (tid >> 3)] = v[m];
  if (tid == 0) lds[8 * 576] = v[8];                   // X[kN / 2]
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int t2 = 8 * (tid & 63) + (tid >> 6);        // the thread that held f = 512 j + 8 lane + wave
    c32 o = lds[j * 576 + t2 + (t2 >> 3)];
    if (tid == 0 && j == 0) o.y = lds[8 * 576].x;
    row[64 * j] = o;
  }
}

}  // namespace
