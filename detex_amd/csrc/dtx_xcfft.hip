// All-pairs normalised cross-correlation through the frequency-domain machinery of the scan
// (dtx_fftcore.h): replaces the lag-by-lag contraction of dtx_xcorr.hip for long events.
//
// For a fixed later event c the reference's _CCX2 (detex/construct.py:425-466) correlates every
// earlier event b against c at the 2S+1 kept lags.  With c's zero-padded channel planes z_c as
// the "stream" and the earlier events as "basis vectors" this is the scan's problem:
//   Y_b[p] = sum_ch sum_j x_b,ch[j] z_c,ch[p + j],   p = 0 .. 2S  (lag index)
// evaluated by overlap-save blocks of 8192 samples: block spectra of z_c once per event,
// spectra of event PAIRS (b, b+1) as one complex sequence once per call (two correlations per
// transform), events longer than 5633 samples cut into segments of 4096 that meet the block
// spectra at shifted block indices (virtual channels; 12 per pair at L = 13 000).  The epilogue
// is the contraction kernels' own: cc = (Y - sum(x_b) mu_w) rs_b rs_w, scrub, and the running
// (max, first argmax) over lags as a 64-bit atomicMax key per pair; the winning lag is then
// re-evaluated in float64 by xc_finish_kernel exactly as before, so the float32 transforms
// only have to pick the lag.
#include <cstdlib>
#include "dtx_fftcore.h"
#include "dtx_plan.h"

namespace {

constexpr float kScrubTolF = 1e-5f;   // dtx_xcorr.hip: kScrubTol
// Work list (scripts/ab_xc.sh, configs[2], kernel ms): rounds 1-2 gave a workgroup 8 pairs, which it walked two at
// a time, streaming its block spectra from HBM once per two pairs (12 MB of them in flight per XCD) past 4.3 MB
// of pair spectra: UNIT 8 / GROUP 1 181.4, 4 / 1 174.3, 2 / 1 181.3, 16 / 1 195.9.  Now a workgroup takes ONE
// step of two pairs, the workgroups of a later event's GROUP units run side by side and share its block
// spectra in L2 (2 / 2 167.4, 2 / 16 165.1, 2 / 32 216: the group's pair spectra must stay in the Infinity
// Cache), and CTILE later events are walked together inside a group so that a unit's pair spectra serve
// them all from L2: CTILE 2 / GROUP 16 162.4, 2 / 8 157.5, 4 / 8 156.1, 6 / 8 160.0, 8 / 8 162.2, 4 / 4 161.8.
// What is left is the sum of the kernel's two phases (products, transforms), which one workgroup per CU runs
// one after the other.
#ifndef DTX_XC_UNIT
#define DTX_XC_UNIT 2
#endif
#ifndef DTX_XC_GROUP
#define DTX_XC_GROUP 8
#endif
#ifndef DTX_XC_CTILE
#define DTX_XC_CTILE 4
#endif
constexpr int kCtileX = DTX_XC_CTILE; // later events walked together inside a group
constexpr int kGroupX = DTX_XC_GROUP; // units whose workgroups run side by side for the same later event
constexpr int kUnitX = DTX_XC_UNIT;   // event pairs per workgroup

// Wave-wide max / min by data-parallel primitives (row operations of the vector ALU: no LDS crossbar,
// no ds_bpermute round trips): after the four in-row steps every lane of a row of 16 holds the row's
// result, row_bcast15 / row_bcast31 carry it up the rows; lane 63 ends with the wave's.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_mov(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxf(v, __int_as_float(dpp_mov<0xB1, 0xF>(__float_as_int(v))));     // quad_perm [1,0,3,2]
  v = fmaxf(v, __int_as_float(dpp_mov<0x4E, 0xF>(__float_as_int(v))));     // quad_perm [2,3,0,1]
  v = fmaxf(v, __int_as_float(dpp_mov<0x141, 0xF>(__float_as_int(v))));    // row_half_mirror
  v = fmaxf(v, __int_as_float(dpp_mov<0x140, 0xF>(__float_as_int(v))));    // row_mirror
  v = fmaxf(v, __int_as_float(dpp_mov<0x142, 0xA>(__float_as_int(v))));    // row_bcast15 -> rows 1, 3
  v = fmaxf(v, __int_as_float(dpp_mov<0x143, 0xC>(__float_as_int(v))));    // row_bcast31 -> rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)dpp_mov<0xB1, 0xF>((int)v));
  v = min(v, (unsigned)dpp_mov<0x4E, 0xF>((int)v));
  v = min(v, (unsigned)dpp_mov<0x141, 0xF>((int)v));
  v = min(v, (unsigned)dpp_mov<0x140, 0xF>((int)v));
  v = min(v, (unsigned)dpp_mov<0x142, 0xA>((int)v));
  v = min(v, (unsigned)dpp_mov<0x143, 0xC>((int)v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ inline int64_t condensed_index_x(int64_t N, int64_t b, int64_t c) {
  return N * b - b * (b + 1) / 2 + (c - b - 1);
}

// W[pair][vc][wave-major f] = 1/kN sum_j (u_2p,ch[j0 + j] + i u_2p+1,ch[j0 + j]) e^{+2 pi i j f / kN},
// vc = ch * nseg + seg, j0 = seg * seglen, j < seglen (and j0 + j < L); float32 on the device
// (2000 events x 12 virtual channels is 12 000 transforms: seconds on the host, < 1 ms here).
// u_b = x_b * rs_b, rs_b = 1 / sqrt(ssd(x_b)): the two events of a pair share one complex
// transform, whose float32 rounding is relative to the larger of the two -- catalogue events
// differ by orders of magnitude in amplitude, unit-energy rows do not.
__global__ __launch_bounds__(kT) void xc_wspec_kernel(const float* __restrict__ td, int N, int n,
                                                      int Nc, int L, int nseg, int seglen,
                                                      const float* __restrict__ frs,
                                                      const c32* __restrict__ tab,
                                                      c32* __restrict__ W) {
  __shared__ c32 lds[kLdsC];
  const int tid = threadIdx.x;
  const int nvc = Nc * nseg;
  const int pr = blockIdx.x / nvc, vc = blockIdx.x - pr * nvc;
  const int ch = vc / nseg, sg = vc - ch * nseg;
  const int a = 2 * pr, b = a + 1;
  const int j0 = sg * seglen;
  const float ra = frs[a], rb = b < N ? frs[b] : 0.f;
  const LdsBases lb = lds_bases(tid);
  c32 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = 8 * ((tid & 63) + 64 * r) + (tid >> 6);     // point index the transform expects
    float re = 0.f, im = 0.f;
    if (j < seglen && j0 + j < L) {
      const int64_t q = (int64_t)(j0 + j) * Nc + ch;
      re = td[(int64_t)a * n + q] * ra;
      if (b < N) im = td[(int64_t)b * n + q] * rb;
    }
    v[r] = make_float2(re, im);
  }
  fft8192<1>(v, lds, lb, tab, tid);
  c32* dst = W + ((int64_t)pr * nvc + vc) * kN;       // register m holds frequency tid + 512 m
  const float sc = 1.0f / (float)kN;
#pragma unroll
  for (int m = 0; m < 16; ++m) dst[w_slot(tid + 512 * m)] = make_float2(v[m].x * sc, v[m].y * sc);
}

#ifndef DTX_XC_BLKLOOP
#define DTX_XC_BLKLOOP 0     // 1: a workgroup walks ALL lag blocks of its (later event, unit) and keeps the rows' best
#endif                       // candidate in registers (one atomicMax per row instead of three): 168 against 152 ms, not kept
#ifndef DTX_XC_DUAL
#define DTX_XC_DUAL 1        // the two transforms of an iteration taken together (fft8192_x2), the second on the X staging buffers
#endif
#ifndef DTX_XC_ABLATE
#define DTX_XC_ABLATE 0      // timing-only ablations (results are wrong): 1 no pair-spectrum loads, 2 no block-spectrum loads,
                             // 4 no epilogue (normalise / max / atomics), 8 no transforms, 16 epilogue without the atomics,
                             // 32 (2 x 2 kernel): what ONE lag block on a 16 384-point transform would cost --
                             //    only lag block 0 is launched, its product loop runs DTX_XC_A16_STEPS steps (24 = the MACs of
                             //    12 double-length steps, 32 = their L1 traffic: 256 KB per 32 768 MACs against 192 KB here),
                             //    every finish runs TWO transforms (2 x 14 / 13 of one is priced on top by hand) and searches
                             //    three blocks' worth of lags
#endif
#ifndef DTX_XC_A16_STEPS
#define DTX_XC_A16_STEPS 24
#endif
struct XcFftArgs {
  const c32* xh;        // [N][nblk][nseg][Nc][kXh] block spectra of the padded planes of every event
  const c32* W;         // [ceil(N/2)][Nc*nseg][kN] pair spectra
  const c32* tab;
  const int2* items;    // (c, first pair of the unit)
  const float* fsum;    // [N] sum(x_b)
  const float* frs;     // [N] 1/sqrt(ssd(x_b)), 0 for a dead event
  const float* mean_w;  // [N][Ppad]
  const float* rsw;     // [N][Ppad]
  unsigned long long* keys;
  int64_t Ppad, pair_begin, pair_end;
  int N, Nc, nseg, nblk, V, Pz;
  int seglen, zend;     // template segment length; first sample of the padded planes behind the data (S + L)
  int64_t per_xcd, nwork;
};

// one workgroup: event c, block blk of its lags, a unit of up to kUnitX earlier-event pairs
template <int VM>
__global__ __launch_bounds__(kT, DTX_FFT_WAVES) void xc_fft_kernel(const XcFftArgs a) {
  __shared__ c32 lds[kLdsC];
  __shared__ c32 xst[2][kXh];     // the block spectrum of the current / next virtual channel (see the MAC loop)
  __shared__ unsigned long long wred[4][kT / 64];   // the waves' lag-search candidates of an iteration's four rows
  const int tid = threadIdx.x;
  // workgroups are dealt to the 8 XCDs round-robin: give every XCD a contiguous run of the
  // (unit-major) work list so that the workgroups sharing an L2 share a unit's pair spectra
  const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (logical >= a.nwork) return;
#if DTX_XC_BLKLOOP
  const int item = (int)logical, blk_lo = 0, blk_hi = a.nblk;
#else
  const int item = (int)(logical / a.nblk), blk_lo = (int)(logical - (int64_t)item * a.nblk), blk_hi = blk_lo + 1;
#endif
  const int2 it = a.items[item];
  const int c = it.x, pr0 = it.y;
  const int npr = min(kUnitX, (c + 1) / 2 - pr0);          // pairs with at least one row < c
  const LdsBases lb = lds_bases(tid);
  // state of the lag block being walked (set by load_block)
  int p0 = 0, nse = 1, nsteps = 1;
  const c32* xh = a.xh;
  float mu[VM], rw[VM];
  auto load_block = [&](int blk) {
    p0 = blk * a.V;
#pragma unroll
    for (int m = 0; m < VM; ++m) {
      const int p = p0 + tid + 512 * m;
      const bool ok = p < a.Pz;
      mu[m] = ok ? a.mean_w[(int64_t)c * a.Ppad + p] : 0.f;
      rw[m] = ok ? a.rsw[(int64_t)c * a.Ppad + p] : __builtin_nanf("");       // (see reduce_rows)
    }
    // Segment s of the earlier events meets the later event's padded planes at samples blk V + s seglen ..:
    // a (block, segment) whose 8192 samples lie wholly in the zero padding behind the data contributes
    // nothing and is skipped (3 of 16 products at L = 13 000 with segments of 4096; 1 of 12 with the
    // 3 x 4 layout the host picks).  nse: segments of this lag block that see data.
    nse = min(a.nseg, max(1, (a.zend - p0 + a.seglen - 1) / a.seglen));
    nsteps = a.Nc * nse;
    xh = a.xh + ((int64_t)c * a.nblk + blk) * a.nseg * a.Nc * kXh;
  };
  // the thread's 16 frequencies in a stored half spectrum (see ds_fft_kernel)
  const int wv = tid >> 6, ln = tid & 63;
  const int xlo = 512 * wv + ln, xhi = wv ? 512 * (8 - wv) + 1023 - ln : 1024 - ln;
  auto fix_x = [&](c32 xx, int r) {
    if (r == 0 && tid == 0) xx.y = 0.f;
    if (r == 8 && tid == 0) return make_float2(xx.y, 0.f);
    if (r >= 8) xx.y = -xx.y;
    return xx;
  };
  auto x_slot = [&](int r) { return r < 8 ? xlo + 64 * r : ((r == 8 && tid == 0) ? 0 : xhi - 64 * r); };
  const int wlo = 512 * wv + ln;              // pair spectra: w_slot order, float4 units
  const int nvc = a.Nc * a.nseg;
  auto xsrc = [&](int i) {                    // block spectrum of step i = (channel, segment): uniform
    const int ch = i / nse, sg = i - ch * nse;
    return reinterpret_cast<const float4*>(xh + ((int64_t)sg * a.Nc + ch) * kXh) + tid;
  };
  auto woff = [&](int i) {                    // its pair spectrum inside a pair's [nvc][kN]
    const int ch = i / nse, sg = i - ch * nse;
    return (int64_t)(ch * a.nseg + sg) * (kN / 2) + wlo;      // float4 units
  };

  // rows 2 pr (real part) and 2 pr + 1 (imaginary part) of a finished transform: normalise,
  // scrub, max / first argmax over the block's lags, one atomicMax per row and wave.
  // The thread's VM lags are compared as floats (first of equals: the smaller lag), the wave's winner is
  // found by row operations -- the max of cc, then the smallest lag among the lanes that hold it -- and
  // only that one candidate is packed into the 64-bit key.  (Rounds 1-2 packed a key per lag and reduced
  // keys through 12 ds_bpermute round trips per row: the epilogue was 22 of the kernel's 190 ms.)
  // rw[m] is NaN for lags past Pz: their cc is NaN and never wins a comparison.
  // (r1v / s1v: the rows' 1 / sqrt(ssd) and sum x_b / sqrt(ssd), fetched before the products: read here, the two
  // dependent scalar loads per row stood exposed in front of every row's arithmetic)
  // The wave's candidate of each row goes to wred[row of the iteration][wave]; after the iteration's closing barrier
  // four threads fold the eight waves' candidates and issue ONE atomicMax per row (eight per row: 9 of the kernel's
  // 156 ms were atomics).
  auto reduce_rows = [&](const c32 (&z)[16], int pr, const float* r1v, const float* s1v, int which) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = 2 * pr + h;
      const bool rowok = b < c;                                   // uniform
      // the pair spectra hold unit-energy rows x_b rs_b: cc = (Y' - (sum x_b) rs_b mu_w) rs_w
      const float r1 = r1v[h], s1 = s1v[h];
      float best = -INFINITY;
      int bm = 0;
#pragma unroll
      for (int m = 0; m < VM; ++m) {
        const float y = h ? z[m].y : z[m].x;
        float cc = (y - s1 * mu[m]) * rw[m];
        cc = (fabsf(cc) > 1.f + kScrubTolF) ? 0.f : cc;           // construct.py:458-462 (NaN stays NaN)
        const bool gt = cc > best;
        best = gt ? cc : best;
        bm = gt ? m : bm;
      }
      const float wmax = wave_max_f32(best);                      // uniform
      const unsigned pcand = (best == wmax) ? (unsigned)(p0 + tid + 512 * bm) : 0xFFFFFFFFu;
      const unsigned pmin = wave_min_u32(pcand);
#if DTX_XC_ABLATE & 16
      if (pmin == 12345u && wmax == 0.5f) a.keys[0] = 1ull;
#else
      if ((tid & 63) == 0) {
        unsigned long long key = 0ull;                            // (0 = no candidate: below every real key)
        if (rowok && r1 > 0.f && wmax > -INFINITY) {
          const unsigned bits = __float_as_uint(wmax);
          const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
          key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - pmin);
        }
        wred[2 * which + h][tid >> 6] = key;
      }
#endif
    }
  };
  // Pairs are taken two at a time: the block spectra (the operand both share) are staged once
  // per step for two MACs -- the MAC streams ~1 MB per transform from L2 and is what bounds this kernel.
  //
  // One step = one (channel, segment): z += W_a X, zb += W_b X at the thread's 16 frequencies.
  //   * X (32 KB, the same for every thread's mirrored reads) goes through LDS, double-buffered a step
  //     ahead: four 16-byte loads per thread at the top of the step, four 16-byte LDS stores at its end.
  //     (They are held in four NAMED registers: as an array the compiler moved them to LDS itself --
  //     global load, wait, LDS store, LDS load, LDS store -- and the wait sat right behind the loads.)
  //   * W (2 x 64 KB) comes in half-step batches of eight 16-byte loads, the batch of the NEXT step issued as
  //     soon as its registers are free: a load has a whole step to land instead of being waited for at once.
  for (int q = 0; q < npr; q += 2) {
    const int pr = pr0 + q;
    const bool has2 = q + 1 < npr;
    const float4* wqa = reinterpret_cast<const float4*>(a.W + (int64_t)pr * nvc * kN);
    const float4* wqb = wqa + (has2 ? (int64_t)nvc * (kN / 2) : 0);
    float r1v[4], s1v[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int b = 2 * pr + h;
      const bool rowok = b < c && (h < 2 || has2);                // uniform
      r1v[h] = rowok ? a.frs[b] : 0.f;
      s1v[h] = rowok ? a.fsum[b] * r1v[h] : 0.f;
    }
    unsigned long long kacc = 0ull;          // threads 0 .. 3: the best candidate of their row over the lag blocks
    float4 wa[2][4], wb[2][4];
    c32 z[16], zb[16];
    auto issue_w = [&](int i, auto hc) {
      constexpr int h = hc;
      const int64_t o = woff(i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if DTX_XC_ABLATE & 1     // timing only: no pair-spectrum loads
        wa[h][j] = wb[h][j] = make_float4(1.f, 0.5f, 1.f, 0.5f);
#else
        wa[h][j] = (wqa + o + 64 * (4 * h + j))[0];
        wb[h][j] = (wqb + o + 64 * (4 * h + j))[0];
#endif
      }
    };
    auto mac_half = [&](const c32* xc, auto hc) {
      constexpr int h = hc;
#pragma unroll
      for (int g = 0; g < 2; ++g) {             // four frequencies at a time: eight live X registers
        c32 px[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 8 * h + 4 * g + e;
#if DTX_XC_ABLATE & 2     // timing only: no block-spectrum loads
          px[e] = make_float2(0.25f, 1.f);
#else
          px[e] = fix_x(xc[x_slot(r)], r);
#endif
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = 8 * h + 4 * g + e;
          const float4 ta = wa[h][2 * g + (e >> 1)], tb = wb[h][2 * g + (e >> 1)];
          const c32 pa = (e & 1) ? make_float2(ta.z, ta.w) : make_float2(ta.x, ta.y);
          const c32 pb = (e & 1) ? make_float2(tb.z, tb.w) : make_float2(tb.x, tb.y);
          // two fused multiply-adds per component (written as a sum of products the compiler
          // emits mul + fma + add)
          z[m].x = __fmaf_rn(-pa.y, px[e].y, __fmaf_rn(pa.x, px[e].x, z[m].x));
          z[m].y = __fmaf_rn(pa.y, px[e].x, __fmaf_rn(pa.x, px[e].y, z[m].y));
          zb[m].x = __fmaf_rn(-pb.y, px[e].y, __fmaf_rn(pb.x, px[e].x, zb[m].x));
          zb[m].y = __fmaf_rn(pb.y, px[e].x, __fmaf_rn(pb.x, px[e].y, zb[m].y));
        }
      }
    };
#pragma nounroll
    for (int blk = blk_lo; blk < blk_hi; ++blk) {
    load_block(blk);
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = zb[m] = make_float2(0.f, 0.f);
    {
      const float4* src = xsrc(0);
      const float4 x0 = src[0], x1 = src[512], x2 = src[1024], x3 = src[1536];
      issue_w(0, std::integral_constant<int, 0>{});
      issue_w(0, std::integral_constant<int, 1>{});
      float4* dst = reinterpret_cast<float4*>(xst[0]) + tid;
      dst[0] = x0; dst[512] = x1; dst[1024] = x2; dst[1536] = x3;
    }
    __syncthreads();
#pragma nounroll
    for (int i = 0; i < nsteps; ++i) {
      // (the last step fetches its own operands again instead of branching: with the step straight-line
      // the scheduling barriers below hold, and a redundant fetch per 11 .. 12 steps costs less than a
      // compiler that sinks half the MACs behind the next loads)
      const int nx = min(i + 1, nsteps - 1);    // uniform
      const float4* src = xsrc(nx);
      const float4 x0 = src[0], x1 = src[512], x2 = src[1024], x3 = src[1536];
      __builtin_amdgcn_sched_barrier(0);
      const c32* xc = xst[i & 1];
      mac_half(xc, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      issue_w(nx, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      mac_half(xc, std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      issue_w(nx, std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      float4* dst = reinterpret_cast<float4*>(xst[(i + 1) & 1]) + tid;
      dst[0] = x0; dst[512] = x1; dst[1024] = x2; dst[1536] = x3;
      __syncthreads();
    }
#if DTX_XC_ABLATE & 8
    if (z[0].x == 1.2345f && zb[3].y == 5.4321f) a.keys[1] = 1ull;     // (keeps the MAC alive)
#else
    if (DTX_XC_DUAL && has2) {
      // both transforms together, the second on the (now idle) staging buffers of the block spectra:
      // 64 KB like the image; the barrier that follows keeps the next iteration's staging stores
      // behind every wave's last reads of it
      fft8192_x2<1>(z, zb, lds, &xst[0][0], lb, a.tab, tid);
    } else {
      fft8192<1>(z, lds, lb, a.tab, tid);
    }
#endif
#if DTX_XC_ABLATE & 4
    if (z[0].x == 1.2345f && z[5].y == 5.4321f) a.keys[1] = 1ull;
#else
    reduce_rows(z, pr, r1v, s1v, 0);
#endif
    if (has2) {
#if !(DTX_XC_ABLATE & 8)
      if (!DTX_XC_DUAL) fft8192<1>(zb, lds, lb, a.tab, tid);
#endif
#if DTX_XC_ABLATE & 4
      if (zb[0].x == 1.2345f && zb[5].y == 5.4321f) a.keys[1] = 1ull;
#else
      reduce_rows(zb, pr + 1, r1v + 2, s1v + 2, 1);
#endif
    }
    __syncthreads();
#if !(DTX_XC_ABLATE & (4 | 16))
    if (tid < (has2 ? 4 : 2)) {
      unsigned long long key = wred[tid][0];
#pragma unroll
      for (int w = 1; w < kT / 64; ++w) key = max(key, wred[tid][w]);
      kacc = max(kacc, key);
    }
#endif
    }      // lag blocks
#if !(DTX_XC_ABLATE & (4 | 16))
    if (tid < (has2 ? 4 : 2)) {
      const int b = 2 * pr + tid;
      if (kacc != 0ull && b < c) {
        const int64_t idx = condensed_index_x(a.N, b, c);
        if (idx >= a.pair_begin && idx < a.pair_end) atomicMax(a.keys + idx, kacc);
      }
    }
#endif
  }
}

// ---- the 2 x 2 workgroup ------------------------------------------------------------------------------------
// What bounds xc_fft_kernel is the sum of its phases under two waves per SIMD, and 80 KB through a CU's L1 per
// product (a pair spectrum to registers + half of a staged block spectrum).  Here a workgroup of 1024 threads takes
// TWO later events x TWO pair spectra of one lag block: half h (512 threads, the transform's own layout) holds pair
// spectrum pr0 + h in registers and multiplies it with BOTH later events' block spectra, which the halves stage
// together through LDS (each loads one): 2 x 64 + 2 x 32 = 192 KB per four products = 48 KB each, sixteen waves
// of 128 registers instead of eight of 198.  The staging buffers (4 x 32 KB: two events, double-buffered) and the
// halves' two transform images (2 x 64 KB) are the same 128 KB of LDS, used one phase after the other.
#ifndef DTX_XC_T22
#define DTX_XC_T22 1         // the 2 x 2 workgroup is the lag search's kernel (0: xc_fft_kernel)
#endif
#ifndef DTX_XC16_ABLATE
#define DTX_XC16_ABLATE 0    // timing only (results wrong): 1 no pair-spectrum loads, 2 no block-spectrum staging, 4 no lag search, 8 no transforms
#endif
#ifndef DTX_XC16_DMA
#define DTX_XC16_DMA 1       // xc_fft16_kernel: the later events' block spectra staged by LDS-DMA (0: through registers)
#endif
#ifndef DTX_XC_NW
#define DTX_XC_NW 2          // pair-spectrum batches (2 x 16 bytes per thread) in flight: 4 = a whole step ahead, 2 = half a step
#endif
constexpr int kT2 = 2 * kT;
template <int VM>
__global__ __launch_bounds__(kT2, 1) void xc_fft22_kernel(const XcFftArgs a) {
  __shared__ c32 lds[2 * kLdsC];
  __shared__ unsigned long long wred[8][kT / 64];   // [(half, later event, row)][wave of the half]
  const int tid = threadIdx.x, hf = tid >> 9, ht = tid & 511;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (logical >= a.nwork) return;
#if DTX_XC_ABLATE & 32
  const int item = (int)logical, blk = 0;
#else
  const int item = (int)(logical / a.nblk), blk = (int)(logical - (int64_t)item * a.nblk);
#endif
  const int2 it = a.items[item];
  const int c0 = it.x, has1 = (it.y >> 30) & 1, pr0 = it.y & 0x3fffffff;
  const int c1 = has1 ? c0 + 1 : c0;                       // (without a second event: the first again, its rows dead)
  const int nprt = (a.N + 1) / 2;
  const bool prok = pr0 + hf < nprt;                        // uniform per half
  const int pr = prok ? pr0 + hf : pr0;
  const LdsBases lb = lds_bases(ht);
  const int p0 = blk * a.V;
  const int nse = min(a.nseg, max(1, (a.zend - p0 + a.seglen - 1) / a.seglen));
#if DTX_XC_ABLATE & 32
  const int nsteps_real = a.Nc * nse, nsteps = DTX_XC_A16_STEPS;
#else
  const int nsteps = a.Nc * nse;
#endif
  const int64_t xstride = (int64_t)a.nblk * a.nseg * a.Nc * kXh;
  const c32* xhh = a.xh + (hf ? c1 : c0) * xstride + (int64_t)blk * a.nseg * a.Nc * kXh;   // the event this half stages
  const int wv = ht >> 6, ln = ht & 63;
  const int xlo = 512 * wv + ln, xhi = wv ? 512 * (8 - wv) + 1023 - ln : 1024 - ln;
  auto fix_x = [&](c32 xx, int r) {
    if (r == 0 && ht == 0) xx.y = 0.f;
    if (r == 8 && ht == 0) return make_float2(xx.y, 0.f);
    if (r >= 8) xx.y = -xx.y;
    return xx;
  };
  auto x_slot = [&](int r) { return r < 8 ? xlo + 64 * r : ((r == 8 && ht == 0) ? 0 : xhi - 64 * r); };
  const int nvc = a.Nc * a.nseg;
  auto xsrc = [&](int i) {
#if DTX_XC_ABLATE & 32
    i %= nsteps_real;
#endif
    const int ch = i / nse, sg = i - ch * nse;
    return reinterpret_cast<const float4*>(xhh + ((int64_t)sg * a.Nc + ch) * kXh) + ht;
  };
  const float4* wq = reinterpret_cast<const float4*>(a.W + (int64_t)pr * nvc * kN) + (512 * wv + ln);
  auto woff = [&](int i) {
#if DTX_XC_ABLATE & 32
    i %= nsteps_real;
#endif
    const int ch = i / nse, sg = i - ch * nse;
    return (int64_t)(ch * a.nseg + sg) * (kN / 2);
  };
  // rows 2 pr, 2 pr + 1 against the two later events: 1 / sqrt(ssd) and sum x_b / sqrt(ssd), 0 for a dead row
  float r1v[2][2], s1v[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = 2 * pr + h;
      const bool rowok = prok && b < (k ? c1 : c0) && (k == 0 || has1);
      r1v[k][h] = rowok ? a.frs[b] : 0.f;
      s1v[k][h] = rowok ? a.fsum[b] * r1v[k][h] : 0.f;
    }
  c32 z0[16], z1[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) z0[m] = z1[m] = make_float2(0.f, 0.f);
  float4 w[DTX_XC_NW][2];
  auto issue_w = [&](int i, auto qc, auto sc) {       // batch q of step i (frequencies 4 q .. 4 q + 3) into slot s
    constexpr int q = qc, sl = sc;
    const float4* src = wq + woff(i);
#if DTX_XC_ABLATE & 1     // timing only: no pair-spectrum loads
    w[sl][0] = w[sl][1] = make_float4(1.f, 0.5f, 1.f, 0.5f);
    if (src == nullptr) w[sl][0].x = 2.f;
#else
    w[sl][0] = src[64 * (2 * q)];
    w[sl][1] = src[64 * (2 * q + 1)];
#endif
  };
  auto mac_q = [&](const c32* x0, const c32* x1, auto qc, auto sc) {
    constexpr int q = qc, sl = sc;
    c32 pa[4], pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
#if DTX_XC_ABLATE & 2     // timing only: no block-spectrum reads from LDS
      pa[e] = make_float2(0.25f, 1.f); pb[e] = make_float2(0.5f, 1.f);
#else
      pa[e] = fix_x(x0[x_slot(r)], r);
      pb[e] = fix_x(x1[x_slot(r)], r);
#endif
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = 4 * q + e;
      const float4 t = w[sl][e >> 1];
      const c32 pw = (e & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
      z0[m].x = __fmaf_rn(-pw.y, pa[e].y, __fmaf_rn(pw.x, pa[e].x, z0[m].x));
      z0[m].y = __fmaf_rn(pw.y, pa[e].x, __fmaf_rn(pw.x, pa[e].y, z0[m].y));
      z1[m].x = __fmaf_rn(-pw.y, pb[e].y, __fmaf_rn(pw.x, pb[e].x, z1[m].x));
      z1[m].y = __fmaf_rn(pw.y, pb[e].x, __fmaf_rn(pw.x, pb[e].y, z1[m].y));
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2c = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // One step = one (channel, segment): four batches of four frequencies.  Behind the products of a batch the
  // batch DTX_XC_NW further on is fetched into the slot that fell free (NW = 2: half a step ahead, 4: a whole step).
  auto issue_ahead = [&](int i, int nx, auto qc, auto slc) {     // batch q + NW, counted from step i's batch 0
    constexpr int q = qc, g = q + DTX_XC_NW;
    issue_w(g < 4 ? i : nx, std::integral_constant<int, (g & 3)>{}, slc);
  };
  auto step = [&](int i, auto phc) {
    constexpr int ph = phc;
    const int nx = min(i + 1, nsteps - 1);    // (the last step fetches its own operands again: straight-line steps)
    const float4* src = xsrc(nx);
    const c32* x0 = lds + (2 * (i & 1)) * kXh;
    const c32* x1 = x0 + kXh;
    float4* dst = reinterpret_cast<float4*>(lds + (2 * ((i + 1) & 1) + hf) * kXh) + ht;
    float4 xa0 = src[0], xa1 = src[512];
    using S0 = std::integral_constant<int, (ph + 0) % DTX_XC_NW>; using S1 = std::integral_constant<int, (ph + 1) % DTX_XC_NW>;
    using S2 = std::integral_constant<int, (ph + 2) % DTX_XC_NW>; using S3 = std::integral_constant<int, (ph + 3) % DTX_XC_NW>;
    __builtin_amdgcn_sched_barrier(0);
    mac_q(x0, x1, I0{}, S0{});
    __builtin_amdgcn_sched_barrier(0);
    issue_ahead(i, nx, I0{}, S0{});
    __builtin_amdgcn_sched_barrier(0);
    mac_q(x0, x1, I1{}, S1{});
    __builtin_amdgcn_sched_barrier(0);
    issue_ahead(i, nx, I1{}, S1{});
    __builtin_amdgcn_sched_barrier(0);
    dst[0] = xa0; dst[512] = xa1;
    xa0 = src[1024]; xa1 = src[1536];
    __builtin_amdgcn_sched_barrier(0);
    mac_q(x0, x1, I2c{}, S2{});
    __builtin_amdgcn_sched_barrier(0);
    issue_ahead(i, nx, I2c{}, S2{});
    __builtin_amdgcn_sched_barrier(0);
    mac_q(x0, x1, I3{}, S3{});
    __builtin_amdgcn_sched_barrier(0);
    issue_ahead(i, nx, I3{}, S3{});
    __builtin_amdgcn_sched_barrier(0);
    dst[1024] = xa0; dst[1536] = xa1;
    __syncthreads();
  };
  {
    const float4* src = xsrc(0);
    const float4 x0 = src[0], x1 = src[512], x2 = src[1024], x3 = src[1536];
    issue_w(0, I0{}, I0{});
    issue_w(0, I1{}, I1{});
    if constexpr (DTX_XC_NW == 4) { issue_w(0, I2c{}, I2c{}); issue_w(0, I3{}, I3{}); }
    float4* dst = reinterpret_cast<float4*>(lds + hf * kXh) + ht;
    dst[0] = x0; dst[512] = x1; dst[1024] = x2; dst[1536] = x3;
  }
  __syncthreads();
  // (NW = 3 -- the slots rotate, three compiled phases of the step -- spills 239 registers; 4 spills 17: 162 ms)
  static_assert(DTX_XC_NW == 2 || DTX_XC_NW == 4, "slots that do not rotate");
#pragma nounroll
  for (int i = 0; i < nsteps; ++i) step(i, I0{});
  // the transforms: each half on its own image (the staging buffers' memory; the loop's last barrier is
  // behind every read of them), one later event after the other
  c32* img = lds + hf * kLdsC;
  auto finish = [&](c32 (&z)[16], auto kc) {
    constexpr int k = kc;
    const int c = k ? c1 : c0;
#if DTX_XC_ABLATE & 8     // timing only: no transforms
    if (z[0].x == 1.2345f && z[3].y == 5.4321f) a.keys[1] = 1ull;
#else
    fft8192<1>(z, img, lb, a.tab, ht);
#if DTX_XC_ABLATE & 32
    fft8192<1>(z, img, lb, a.tab, ht);
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#if DTX_XC_ABLATE & 4     // timing only: no epilogue
    if (z[0].x == 1.2345f && z[5].y == 5.4321f) a.keys[1] = 1ull;
    if ((tid & 63) == 0) { wred[4 * hf + 2 * k][wv] = 0ull; wred[4 * hf + 2 * k + 1][wv] = 0ull; }
    return;
#endif
    // cc = (Y' - (sum x_b) rs_b mu_w) rs_w at the thread's VM lags, max / first argmax of both rows.  The lags'
    // window statistics are fetched behind the transform (in front of it they cost the transform its registers:
    // 133 against 127 ms) by a uniform row base + a 32-bit lane offset at a clamped index, then a select
    // (predicated loads became a branch per lag, eighteen 64-bit lane addresses spilled: 139 ms), kStat lags at a
    // time: with all of them in flight at once the epilogue of more than nine lags per thread ran out of registers
    // beside the other later event's accumulators.
    const char* mwc = reinterpret_cast<const char*>(a.mean_w + (int64_t)c * a.Ppad);
    const char* rwc = reinterpret_cast<const char*>(a.rsw + (int64_t)c * a.Ppad);
    const unsigned last = 4u * (unsigned)(a.Pz - 1);
    float best[2] = {-INFINITY, -INFINITY};
    int bm[2] = {0, 0};
    constexpr int kStat = VM <= 9 ? VM : (VM + 1) / 2;
#if DTX_XC_ABLATE & 32
#pragma nounroll
    for (int rep = 0, p0 = 0; rep < 3; ++rep, p0 += a.V)       // (shadows the block's p0: three blocks' lags)
#endif
    dtx::static_for<0, (VM + kStat - 1) / kStat>([&](auto chc) {
      constexpr int m0 = chc * kStat, m1 = m0 + kStat < VM ? m0 + kStat : VM;
      float mu[m1 - m0], rw[m1 - m0];
#pragma unroll
      for (int m = m0; m < m1; ++m) {
        const unsigned p4 = 4u * (unsigned)(p0 + ht + 512 * m), o = min(p4, last);
        mu[m - m0] = *reinterpret_cast<const float*>(mwc + o);
        const float rv = *reinterpret_cast<const float*>(rwc + o);
        rw[m - m0] = p4 <= last ? rv : __builtin_nanf("");      // NaN (a lag past Pz) never wins a comparison
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = m0; m < m1; ++m) {
          const float y = h ? z[m].y : z[m].x;
          float cc = (y - s1v[k][h] * mu[m - m0]) * rw[m - m0];
          cc = (fabsf(cc) > 1.f + kScrubTolF) ? 0.f : cc;           // construct.py:458-462 (NaN stays NaN)
          const bool gt = cc > best[h];
          best[h] = gt ? cc : best[h];
          bm[h] = gt ? m : bm[h];
        }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float r1 = r1v[k][h];
      const float wmax = wave_max_f32(best[h]);
      const unsigned pcand = (best[h] == wmax) ? (unsigned)(p0 + ht + 512 * bm[h]) : 0xFFFFFFFFu;
      const unsigned pmin = wave_min_u32(pcand);
      if ((tid & 63) == 0) {
        unsigned long long key = 0ull;
        if (r1 > 0.f && wmax > -INFINITY) {
          const unsigned bits = __float_as_uint(wmax);
          const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
          key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - pmin);
        }
        wred[4 * hf + 2 * k + h][wv] = key;
      }
    }
  };
  finish(z0, I0{});
  finish(z1, I1{});
  __syncthreads();
  if (tid < 8) {
    unsigned long long key = wred[tid][0];
#pragma unroll
    for (int q = 1; q < kT / 64; ++q) key = max(key, wred[tid][q]);
    const int h2 = tid >> 2, k = (tid >> 1) & 1, h = tid & 1;
    const int b = 2 * (pr0 + h2) + h, c = k ? c1 : c0;
    if (key != 0ull && b < c && (k == 0 || has1)) {
      const int64_t idx = condensed_index_x(a.N, b, c);
      if (idx >= a.pair_begin && idx < a.pair_end) atomicMax(a.keys + idx, key);
    }
  }
}

// ---- two workgroups per CU: one pair spectrum x two later events ------------------------------------------------
// The 2 x 2 workgroup runs its phases one after the other -- products (bound by the bytes through the L1), transforms
// (vector issue and dependency chains), lag search -- and nothing else fits on its CU.  Here a workgroup is HALF the
// size: 512 threads (the transform's own layout), ONE pair spectrum in registers x TWO later events, whose block
// spectra it stages through LDS by LDS-DMA -- single-buffered, 2 x 32 KB, the same 64 KB that are the transform image
// afterwards -- in 128 registers: TWO such workgroups share a CU, and while one waits for its staging or runs its
// transforms the other multiplies.  A product costs 64 KB through the L1 instead of 48 (the pair spectrum serves two
// later events, the staged spectra one pair instead of two); what it buys is the overlap of the phases.
#ifndef DTX_XC_T12
#define DTX_XC_T12 1
#endif
#ifndef DTX_XC12_HALVES
#define DTX_XC12_HALVES 0    // 1 (measured slower: 134.8 against 128.1 ms at configs[2], 22.3 against 20.9 at L = 3000): the step in two phases on half spectra (lower / upper 256 slots of the eight rows), double-buffered
#endif                       // inside the 64 KB: a staging load has half a step to land instead of being waited for at once
#ifndef DTX_XC12_ABLATE
#define DTX_XC12_ABLATE 0    // timing only (results wrong): 1 no pair-spectrum loads, 2 no staging, 4 no lag search, 8 no transforms
#endif
template <int VM>
__global__ __launch_bounds__(kT, 4) void xc_fft12_kernel(const XcFftArgs a) {
  __shared__ c32 lds[kLdsC];                    // [later event][kXh] staged block spectra; then the transform image
  static_assert(kLdsC == 2 * kXh, "two staged half spectra fill the image");
  __shared__ unsigned long long wred[4][kT / 64];   // [(later event, row)][wave]
  const int tid = threadIdx.x;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (logical >= a.nwork) return;
  // (item, which of its two pair spectra, lag block): the two pair spectra of an item side by side
  const int blk = (int)(logical % a.nblk);
  const int64_t ip = logical / a.nblk;
  const int item = (int)(ip >> 1), hp = (int)(ip & 1);
  const int2 it = a.items[item];
  const int ity = __builtin_amdgcn_readfirstlane(it.y);
  const int c0 = __builtin_amdgcn_readfirstlane(it.x), has1 = (ity >> 30) & 1, pr = (ity & 0x3fffffff) + hp;
  const int c1 = has1 ? c0 + 1 : c0;
  if (pr >= (a.N + 1) / 2 || 2 * pr >= c1) return;          // no such pair spectrum / no row below the later events
  const LdsBases lb = lds_bases(tid);
  const int p0 = blk * a.V;
  const int nse = min(a.nseg, max(1, (a.zend - p0 + a.seglen - 1) / a.seglen));
  const int nsteps = a.Nc * nse;
  const int64_t xstride = (int64_t)a.nblk * a.nseg * a.Nc * kXh;
  const c32* xh0 = a.xh + c0 * xstride + (int64_t)blk * a.nseg * a.Nc * kXh;     // (uniform: scalar registers)
  const c32* xh1 = a.xh + c1 * xstride + (int64_t)blk * a.nseg * a.Nc * kXh;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
  const int xlo = 512 * wv + ln, xhi = wv ? 512 * (8 - wv) + 1023 - ln : 1024 - ln;
  auto x_slot = [&](int r) { return r < 8 ? xlo + 64 * r : ((r == 8 && tid == 0) ? 0 : xhi - 64 * r); };
  auto fix_x = [&](c32 xx, int r) {
    if (r == 0 && tid == 0) xx.y = 0.f;
    if (r == 8 && tid == 0) return make_float2(xx.y, 0.f);
    if (r >= 8) xx.y = -xx.y;
    return xx;
  };
  const int nvc = a.Nc * a.nseg;
  const char* wq = reinterpret_cast<const char*>(a.W + (int64_t)pr * nvc * kN);       // uniform
  const unsigned wlane = 16u * (unsigned)(512 * wv + ln);
  auto voff = [&](int i) {                      // (channel, segment) of step i as an index into [nseg][Nc] / [Nc][nseg]
    const int ch = i / nse, sg = i - ch * nse;
    return make_int2(sg * a.Nc + ch, ch * a.nseg + sg);
  };
  float r1v[2][2], s1v[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = 2 * pr + h;
      const bool rowok = b < (k ? c1 : c0) && (k == 0 || has1);
      r1v[k][h] = rowok ? a.frs[b] : 0.f;
      s1v[k][h] = rowok ? a.fsum[b] * r1v[k][h] : 0.f;
    }
  c32 z0[16], z1[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) z0[m] = z1[m] = make_float2(0.f, 0.f);
  float4 w[2][2];
  auto issue_w = [&](int i, auto qc, auto sc) {       // batch q of step i (frequency registers 4 q .. 4 q + 3) into slot s
    constexpr int q = qc, sl = sc;
    const char* src = wq + (int64_t)voff(i).y * (kN * 8);
#if DTX_XC12_ABLATE & 1
    w[sl][0] = w[sl][1] = make_float4(1.f, 0.5f, 1.f, 0.5f);
    if (src == nullptr) w[sl][0].x = 2.f;
#else
    w[sl][0] = *reinterpret_cast<const float4*>(src + wlane + 16 * 64 * (2 * q));
    w[sl][1] = *reinterpret_cast<const float4*>(src + wlane + 16 * 64 * (2 * q + 1));
#endif
  };
  // the step's block spectra, both later events: 2 x 4096 complex = 4096 float4, eight per thread, a wave's 64 lanes
  // side by side at a wave-uniform LDS address (LDS-DMA: no staging registers, no ds_write)
  [[maybe_unused]] auto stage = [&](int i) {
    const int64_t o = (int64_t)voff(i).x * kXh * 8;
    const char* s0 = reinterpret_cast<const char*>(xh0) + o + 16u * (unsigned)tid;
    const char* s1 = reinterpret_cast<const char*>(xh1) + o + 16u * (unsigned)tid;
    float4* d = reinterpret_cast<float4*>(lds) + 64 * wv;
#if DTX_XC12_ABLATE & 2
    if (s0 == nullptr)
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + 16 * 512 * k),
                                       (__attribute__((address_space(3))) void*)(d + 512 * k), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s1 + 16 * 512 * k),
                                       (__attribute__((address_space(3))) void*)(d + 2048 + 512 * k), 16, 0, 0);
    }
  };
  [[maybe_unused]] auto mac_q = [&](auto qc, auto sc) {
    constexpr int q = qc, sl = sc;
    c32 pa[4], pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      pa[e] = fix_x(lds[x_slot(r)], r);
      pb[e] = fix_x(lds[kXh + x_slot(r)], r);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = 4 * q + e;
      const float4 t = w[sl][e >> 1];
      const c32 pw = (e & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
      z0[m].x = __fmaf_rn(-pw.y, pa[e].y, __fmaf_rn(pw.x, pa[e].x, z0[m].x));
      z0[m].y = __fmaf_rn(pw.y, pa[e].x, __fmaf_rn(pw.x, pa[e].y, z0[m].y));
      z1[m].x = __fmaf_rn(-pw.y, pb[e].y, __fmaf_rn(pw.x, pb[e].x, z1[m].x));
      z1[m].y = __fmaf_rn(pw.y, pb[e].x, __fmaf_rn(pw.x, pb[e].y, z1[m].y));
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2c = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
#if DTX_XC12_HALVES
  // Phase L: frequency registers 0..3 and 12..15 (own + mirrored) read the LOWER 256 slots of the eight 512-slot rows
  // of a stored half spectrum, phase U: registers 4..11 the upper ones.  Buffer L = lds[0 .. 4096) (both later events'
  // lower halves, 2048 slots each), buffer U = lds[4096 .. 8192): while one is read the other is filled.  Thread 0's
  // two mirrored reads outside their half (slot 256 in phase L, the (X[0], X[4096]) pair of slot 0 in phase U) come
  // through a uniform load and a select.
  constexpr int kHalf = 2048;
  const int hlo = 256 * wv + ln;                                        // r < 8: + 64 r (phase U: - 256)
  const int hhi = wv ? 256 * (8 - wv) + 1023 - ln : 1024 - ln;          // r >= 8: - 64 r (phase U: - 256)
  auto h_slot = [&](int r) {
    if (r < 4) return hlo + 64 * r;
    if (r < 8) return hlo + 64 * r - 256;
    if (r < 12) return hhi - 256 - 64 * r;
    return hhi - 64 * r;
  };
  c32 st0 = make_float2(0.f, 0.f), st1 = st0;
  auto load_stray = [&](int i, int ph) {        // slot 256 for phase L, slot 0 (its .y = X[4096]) for phase U
    const int64_t o = (int64_t)voff(i).x * kXh + (ph ? 0 : 256);
    const c32 a0 = xh0[o], a1 = xh1[o];
    st0 = ph ? make_float2(a0.y, 0.f) : a0;
    st1 = ph ? make_float2(a1.y, 0.f) : a1;
  };
  const unsigned sthr = 16u * (unsigned)((tid >> 7) * 256 + (tid & 127));     // the thread's float4 of a half: row (tid >> 7) + 4 k
  auto stage_half = [&](int i, int ph) {        // both later events' halves of step i into buffer ph
    const int64_t o = (int64_t)voff(i).x * kXh * 8 + 16 * 128 * ph;
    const char* s0 = reinterpret_cast<const char*>(xh0) + o + sthr;
    const char* s1 = reinterpret_cast<const char*>(xh1) + o + sthr;
    float4* d = reinterpret_cast<float4*>(lds + 2 * kHalf * ph) + 64 * wv;
#if DTX_XC12_ABLATE & 2
    if (s0 == nullptr)
#endif
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s0 + 16 * 1024 * k),
                                       (__attribute__((address_space(3))) void*)(d + 512 * k), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s1 + 16 * 1024 * k),
                                       (__attribute__((address_space(3))) void*)(d + 1024 + 512 * k), 16, 0, 0);
    }
  };
  const bool t0 = tid == 0;
  auto mac_h = [&](int ph, auto qc, auto sc, c32 sv0, c32 sv1) {
    constexpr int q = qc, sl = sc;
    const c32* x0 = lds + 2 * kHalf * ph;
    const c32* x1 = x0 + kHalf;
    c32 pa[4], pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      pa[e] = x0[h_slot(r)];
      pb[e] = x1[h_slot(r)];
      if (r == 0) { pa[e].y = t0 ? 0.f : pa[e].y; pb[e].y = t0 ? 0.f : pb[e].y; }
      if (r == 8 || r == 12) {
        pa[e].x = t0 ? sv0.x : pa[e].x; pa[e].y = t0 ? sv0.y : pa[e].y;
        pb[e].x = t0 ? sv1.x : pb[e].x; pb[e].y = t0 ? sv1.y : pb[e].y;
      }
      if (r >= 8) { pa[e].y = -pa[e].y; pb[e].y = -pb[e].y; }            // X[kN - f] = conj X[f]
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = 4 * q + e;
      const float4 t = w[sl][e >> 1];
      const c32 pw = (e & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
      z0[m].x = __fmaf_rn(-pw.y, pa[e].y, __fmaf_rn(pw.x, pa[e].x, z0[m].x));
      z0[m].y = __fmaf_rn(pw.y, pa[e].x, __fmaf_rn(pw.x, pa[e].y, z0[m].y));
      z1[m].x = __fmaf_rn(-pw.y, pb[e].y, __fmaf_rn(pw.x, pb[e].x, z1[m].x));
      z1[m].y = __fmaf_rn(pw.y, pb[e].x, __fmaf_rn(pw.x, pb[e].y, z1[m].y));
    }
  };
  // batch order in time: (i, 0) (i, 3) | (i, 1) (i, 2) | (i + 1, 0) ...; slot = position in that order & 1, the batch
  // two further on is fetched into a slot as soon as its products are done
  issue_w(0, I0{}, I0{});
  issue_w(0, I3{}, I1{});
  stage_half(0, 0);
  load_stray(0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma nounroll
  for (int i = 0; i < nsteps; ++i) {
    const int nx = min(i + 1, nsteps - 1);      // (the last step fetches its own operands again: straight-line steps)
    // ---- phase L (buffer U is free: every wave passed the barrier behind its last read of it)
    // (the strays of the half-step just begun are in registers; the next one's are fetched into a second pair)
    stage_half(i, 1);
    const c32 sa0 = st0, sa1 = st1;
    load_stray(i, 1);
    __builtin_amdgcn_sched_barrier(0);
    mac_h(0, I0{}, I0{}, sa0, sa1);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(i, I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    mac_h(0, I3{}, I1{}, sa0, sa1);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(i, I2c{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    // the four DMA loads (and the strays) are older than the four pair-spectrum loads issued behind them (loads
    // return in order): wait for all but those, which stay in flight across the barrier
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- phase U
    stage_half(nx, 0);
    const c32 sb0 = st0, sb1 = st1;
    load_stray(nx, 0);
    __builtin_amdgcn_sched_barrier(0);
    mac_h(1, I1{}, I0{}, sb0, sb1);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(nx, I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    mac_h(1, I2c{}, I1{}, sb0, sb1);
    __builtin_amdgcn_sched_barrier(0);
    issue_w(nx, I3{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
#else
  issue_w(0, I0{}, I0{});
  issue_w(0, I1{}, I1{});
#pragma nounroll
  for (int i = 0; i < nsteps; ++i) {
    const int nx = min(i + 1, nsteps - 1);      // (the last step fetches its own operands again: straight-line steps)
    __syncthreads();                            // every wave has read the previous step's spectra
    stage(i);
    // the eight DMA loads are younger than the step's first two pair-spectrum batches: everything has landed
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    mac_q(I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    issue_w(i, I2c{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    mac_q(I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    issue_w(i, I3{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    mac_q(I2c{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    issue_w(nx, I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    mac_q(I3{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    issue_w(nx, I1{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
  // the transforms on the staging memory (fft8192's first barrier is behind every wave's last read of it)
  auto finish = [&](c32 (&z)[16], auto kc) {
    constexpr int k = kc;
    const int c = k ? c1 : c0;
#if DTX_XC12_ABLATE & 8
    if (z[0].x == 1.2345f && z[3].y == 5.4321f) a.keys[1] = 1ull;
#else
    fft8192<1>(z, lds, lb, a.tab, tid);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if DTX_XC12_ABLATE & 4
    if (z[0].x == 1.2345f && z[5].y == 5.4321f) a.keys[1] = 1ull;
    if ((tid & 63) == 0) { wred[2 * k][wv] = 0ull; wred[2 * k + 1][wv] = 0ull; }
    return;
#endif
    const char* mwc = reinterpret_cast<const char*>(a.mean_w + (int64_t)c * a.Ppad);
    const char* rwc = reinterpret_cast<const char*>(a.rsw + (int64_t)c * a.Ppad);
    const unsigned last = 4u * (unsigned)(a.Pz - 1);
    float best[2] = {-INFINITY, -INFINITY};
    int bm[2] = {0, 0};
    constexpr int kStat = VM <= 9 ? VM : (VM + 1) / 2;
    dtx::static_for<0, (VM + kStat - 1) / kStat>([&](auto chc) {
      constexpr int m0 = chc * kStat, m1 = m0 + kStat < VM ? m0 + kStat : VM;
      float mu[m1 - m0], rw[m1 - m0];
#pragma unroll
      for (int m = m0; m < m1; ++m) {
        const unsigned p4 = 4u * (unsigned)(p0 + tid + 512 * m), o = min(p4, last);
        mu[m - m0] = *reinterpret_cast<const float*>(mwc + o);
        const float rv = *reinterpret_cast<const float*>(rwc + o);
        rw[m - m0] = p4 <= last ? rv : __builtin_nanf("");      // NaN (a lag past Pz) never wins a comparison
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = m0; m < m1; ++m) {
          const float y = h ? z[m].y : z[m].x;
          float cc = (y - s1v[k][h] * mu[m - m0]) * rw[m - m0];
          cc = (fabsf(cc) > 1.f + kScrubTolF) ? 0.f : cc;           // construct.py:458-462 (NaN stays NaN)
          const bool gt = cc > best[h];
          best[h] = gt ? cc : best[h];
          bm[h] = gt ? m : bm[h];
        }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float r1 = r1v[k][h];
      const float wmax = wave_max_f32(best[h]);
      const unsigned pcand = (best[h] == wmax) ? (unsigned)(p0 + tid + 512 * bm[h]) : 0xFFFFFFFFu;
      const unsigned pmin = wave_min_u32(pcand);
      if ((tid & 63) == 0) {
        unsigned long long key = 0ull;
        if (r1 > 0.f && wmax > -INFINITY) {
          const unsigned bits = __float_as_uint(wmax);
          const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
          key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - pmin);
        }
        wred[2 * k + h][wv] = key;
      }
    }
  };
  finish(z0, I0{});
  finish(z1, I1{});
  __syncthreads();
  if (tid < 4) {
    unsigned long long key = wred[tid][0];
#pragma unroll
    for (int q = 1; q < kT / 64; ++q) key = max(key, wred[tid][q]);
    const int k = tid >> 1, h = tid & 1;
    const int b = 2 * pr + h, c = k ? c1 : c0;
    if (key != 0ull && b < c && (k == 0 || has1)) {
      const int64_t idx = condensed_index_x(a.N, b, c);
      if (idx >= a.pair_begin && idx < a.pair_end) atomicMax(a.keys + idx, key);
    }
  }
}

template <int VM>
void launch_xc12(dtx_ctx* ctx, const XcFftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((xc_fft12_kernel<VM>), dim3(grid), dim3(kT), 0, ctx->stream, a);
}

#ifndef DTX_XC16
#define DTX_XC16 0           // the 16 384-point lag search: built, parity-green, measured SLOWER than the 2 x 2 kernel at configs[2]
#endif                       // (157 against 126 ms: profiles/x11_xc16_ablations.txt, profiles/README.md); make EXTRA=-DDTX_XC16=1 builds it
#if DTX_XC16
// ---- one lag block of 16 384 -------------------------------------------------------------------------------------
// At configs[2] (L = 13 000: 13 001 lags) the 8192-point kernels search three lag blocks of 4608 -- 0.56 of every
// transform's outputs are used, and the (block, segment) products that see data add up to 11 per pair spectrum and
// later event.  On a 16 384-point transform the lags are ONE block (13 per thread of a 1024-thread workgroup, 0.79 of
// the outputs) and the products 4: -27 % of the multiply-adds, -28 % of the transform work (two 16 384-point
// transforms = 2 x 14 / 13 of four 8192-point ones, against six), measured beforehand by a timing-only build of
// the 2 x 2 kernel (profiles/x10_ablate16k.txt: 126 -> 86 .. 103 ms predicted).
//
// A workgroup = ONE pair spectrum x TWO later events x one lag block, sixteen waves, a thread's 16 frequencies
// f = 16 (l + 64 r) + w.  The pair spectrum streams through registers (128 KB per step, half a step ahead), the two
// later events' block half-spectra (2 x 64 KB per step) through LDS -- in HALVES, so that the double-buffered
// staging is the same 128 KB the transform image needs afterwards: the thread's frequency registers r = 0..3 and
// 12..15 (own + mirrored) read the LOWER 256 slots of the sixteen 512-slot rows of a stored half spectrum, r = 4..11
// the upper ones; a step is phase L (batches r 0..3, 12..15) then phase U (r 4..7, 8..11), one workgroup barrier
// each.  Two mirrored reads of thread 0 fall outside their half (X[4096]: slot 256, wanted in phase L; the
// (X[0], X[8192]) pair of slot 0, wanted in phase U): those two values per event and step come through the scalar
// path (a uniform address: s_load into scalar registers in front of the previous barrier) and replace what thread 0
// read from LDS by a select -- a branch here cut the half-step into basic blocks and cost 200 spilled registers.
constexpr int kHalfX = 4096;                  // slots of one staged half
template <int VM>
__global__ __launch_bounds__(kT16, 1) void xc_fft16_kernel(const XcFftArgs a) {
  __shared__ c32 lds[4 * kHalfX];               // [phase L / U buffer][later event][kHalfX]; then the 128 KB image
  static_assert(4 * kHalfX >= kN16, "the staging buffers hold the transform image");
  __shared__ unsigned long long wred[4][kT16 / 64];   // [(later event, row)][wave]
  // (wave-uniform values are SAID to be: the compiler then keeps them, and every address built from them, in scalar
  // registers -- the vector registers are all spoken for)
  const int tid = threadIdx.x, hf = __builtin_amdgcn_readfirstlane(tid >> 9), ht = tid & 511;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (logical >= a.nwork) return;
  // (item, which of its two pair spectra, lag block): the two pair spectra of an item side by side
  const int blk = (int)(logical % a.nblk);
  const int64_t ip = logical / a.nblk;
  const int item = (int)(ip >> 1), hp = (int)(ip & 1);
  const int2 it = a.items[item];
  const int ity = __builtin_amdgcn_readfirstlane(it.y);
  const int c0 = __builtin_amdgcn_readfirstlane(it.x), has1 = (ity >> 30) & 1, pr = (ity & 0x3fffffff) + hp;
  const int c1 = has1 ? c0 + 1 : c0;
  if (pr >= (a.N + 1) / 2 || 2 * pr >= c1) return;          // no such pair spectrum / no row below the later events
  const LdsBases lb = lds_bases(tid);
  const int p0 = blk * a.V;
  const int nse = min(a.nseg, max(1, (a.zend - p0 + a.seglen - 1) / a.seglen));
  const int nsteps = a.Nc * nse;
  const int64_t xstride = (int64_t)a.nblk * a.nseg * a.Nc * kXh16;
  const c32* xhh = a.xh + (hf ? c1 : c0) * xstride + (int64_t)blk * a.nseg * a.Nc * kXh16;   // the event this half stages
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
  // the thread's reads of a staged half spectrum (positions inside an event's half-buffer)
  const int xlo = 256 * wv + ln;                                        // r < 8: + 64 r (phase U: - 256)
  const int xhi = wv ? 256 * (16 - wv) + 1023 - ln : 1024 - ln;         // r >= 8: - 64 r (phase U: - 256)
  auto x_slot = [&](int r) {                     // (thread 0 at r = 8, 12: a valid slot whose value is replaced)
    if (r < 4) return xlo + 64 * r;
    if (r < 8) return xlo + 64 * r - 256;
    if (r < 12) return xhi - 256 - 64 * r;
    return xhi - 64 * r;
  };
  const int nvc = a.Nc * a.nseg;
  // staging: half hf loads event hf; float4 q = ht + 512 k of a half = row (ht >> 7) + 4 k, float4 (ht & 127) of its 128
  const unsigned sthr = 16u * (unsigned)((ht >> 7) * 256 + (ht & 127));      // bytes inside a stored half spectrum
  auto xsrc = [&](int i, int ph) {              // uniform
    const int ch = i / nse, sg = i - ch * nse;
    return reinterpret_cast<const char*>(xhh + ((int64_t)sg * a.Nc + ch) * kXh16) + 16 * 128 * ph;
  };
  const c32* xh0 = a.xh + c0 * xstride + (int64_t)blk * a.nseg * a.Nc * kXh16;     // (uniform: the scalar path)
  const c32* xh1 = a.xh + c1 * xstride + (int64_t)blk * a.nseg * a.Nc * kXh16;
  f2 st0 = {0.f, 0.f}, st1 = st0;               // thread 0's stray values of the CURRENT half-step, both later events
  auto load_stray = [&](int i, int ph) {        // slot 256 for phase L, slot 0 for phase U
    const int ch = i / nse, sg = i - ch * nse;
    const int64_t o = ((int64_t)sg * a.Nc + ch) * kXh16 + (ph ? 0 : 256);
    const c32 a0 = xh0[o], a1 = xh1[o];
    st0 = ph ? (f2){a0.y, 0.f} : (f2){a0.x, a0.y};      // phase U wants X[8192] = the imaginary part of slot 0
    st1 = ph ? (f2){a1.y, 0.f} : (f2){a1.x, a1.y};
  };
  const char* wq = reinterpret_cast<const char*>(a.W + (int64_t)pr * nvc * kN16);      // uniform
  const unsigned wlane = 16u * (unsigned)(512 * wv + ln);                              // the thread's float4 inside a [kN16]
  auto woff = [&](int i) {
    const int ch = i / nse, sg = i - ch * nse;
    return (int64_t)(ch * a.nseg + sg) * (kN16 * 8);                                   // bytes, uniform
  };
  float r1v[2][2], s1v[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = 2 * pr + h;
      const bool rowok = b < (k ? c1 : c0) && (k == 0 || has1);
      r1v[k][h] = rowok ? a.frs[b] : 0.f;
      s1v[k][h] = rowok ? a.fsum[b] * r1v[k][h] : 0.f;
    }
  f2 z0[16], z1[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) z0[m] = z1[m] = (f2){0.f, 0.f};
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 w[2][2];
  auto issue_w = [&](int i, auto qc, auto sc) {       // batch q of step i (frequency registers 4 q .. 4 q + 3) into slot s
    constexpr int q = qc, sl = sc;
    const char* src = wq + woff(i);                   // scalar base + 32-bit lane offset + immediate
#if DTX_XC16_ABLATE & 1
    w[sl][0] = w[sl][1] = (f4){1.f, 0.5f, 1.f, 0.5f};
    if (src == nullptr) w[sl][0].x = 2.f;
#else
    w[sl][0] = *reinterpret_cast<const f4*>(src + wlane + 16 * 64 * (2 * q));
    w[sl][1] = *reinterpret_cast<const f4*>(src + wlane + 16 * 64 * (2 * q + 1));
#endif
  };
  const bool t0 = tid == 0;
  auto mac_q = [&](const c32* x0, const c32* x1, auto qc, auto sc) {
    constexpr int q = qc, sl = sc;
    f2 pa[4], pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      pa[e] = *reinterpret_cast<const f2*>(x0 + x_slot(r));
      pb[e] = *reinterpret_cast<const f2*>(x1 + x_slot(r));
      if (r == 0) {                             // X[0] is real: slot 0's imaginary part is X[8192]
        pa[e].y = t0 ? 0.f : pa[e].y;
        pb[e].y = t0 ? 0.f : pb[e].y;
      }
      if (r == 8 || r == 12) {                  // thread 0's mirrored reads outside the staged half
        pa[e] = t0 ? st0 : pa[e];
        pb[e] = t0 ? st1 : pb[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = 4 * q + e;
      const f4 t = w[sl][e >> 1];
      const f2 pw = (e & 1) ? (f2){t.z, t.w} : (f2){t.x, t.y};
      if constexpr (q < 2) {                    // own frequencies / mirrored ones: X[kN16 - f] = conj X[f]
        cmac(z0[m], pw, pa[e]);
        cmac(z1[m], pw, pb[e]);
      } else {
        cmac_conj(z0[m], pw, pa[e]);
        cmac_conj(z1[m], pw, pb[e]);
      }
    }
  };
  const int wvh = wv & 7;                                        // wave of the half: 0 .. 7 (uniform)
  auto stage_dma = [&](const char* src, c32* dbuf) {
    float4* d = reinterpret_cast<float4*>(dbuf) + 64 * wvh;
#if DTX_XC16_ABLATE & 2
    if (src == nullptr)
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + sthr + 16 * 1024 * k),
                                       (__attribute__((address_space(3))) void*)(d + 512 * k), 16, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2c = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  // batch order in time: (i, 0) (i, 3) | (i, 1) (i, 2) | (i + 1, 0) ...; pair-spectrum slot = position in that order & 1,
  // the batch two further on is fetched into a slot as soon as its products are done
  auto half = [&](int i, auto phc) {
    constexpr int ph = phc;
    const int nx = min(i + 1, nsteps - 1);      // (the last step fetches its own operands again: straight-line steps)
    // the NEXT half-step's block spectra: phase U of this step, or phase L of the next
    const char* src = ph ? xsrc(nx, 0) : xsrc(i, 1);
    const c32* x0 = lds + (2 * ph) * kHalfX;
    const c32* x1 = x0 + kHalfX;
    c32* dbuf = lds + (2 * (1 - ph) + hf) * kHalfX;
#if DTX_XC16_DMA
    // the next half-step's block spectra straight into the other buffer (LDS-DMA: no staging registers, no
    // ds_write): a wave's 64 lanes x 16 bytes land side by side at a wave-uniform LDS address
    stage_dma(src, dbuf);
#else
    float4* dst = reinterpret_cast<float4*>(dbuf) + ht;
    auto ldx = [&](int k) { return *reinterpret_cast<const float4*>(src + sthr + 16 * 1024 * k); };
    float4 xa0 = ldx(0), xa1 = ldx(1);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ph == 0) mac_q(x0, x1, I0{}, I0{}); else mac_q(x0, x1, I1{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ph == 0) issue_w(i, I1{}, I0{}); else issue_w(nx, I0{}, I0{});
    __builtin_amdgcn_sched_barrier(0);
#if !DTX_XC16_DMA
    dst[0] = xa0; dst[512] = xa1;
    xa0 = ldx(2); xa1 = ldx(3);
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (ph == 0) mac_q(x0, x1, I3{}, I1{}); else mac_q(x0, x1, I2c{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ph == 0) issue_w(i, I2c{}, I1{}); else issue_w(nx, I3{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
#if !DTX_XC16_DMA
    dst[1024] = xa0; dst[1536] = xa1;
#endif
    if constexpr (ph == 0) load_stray(i, 1); else load_stray(nx, 0);     // the next half-step's, in front of the barrier
#if DTX_XC16_DMA
    // the four DMA loads are older than the four pair-spectrum loads issued behind them (loads return in order):
    // wait for all but those, leave them in flight across the barrier
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
  };
  {
    const float4* src = reinterpret_cast<const float4*>(xsrc(0, 0) + sthr);
    const float4 x0 = src[0], x1 = src[1024], x2 = src[2048], x3 = src[3072];
    issue_w(0, I0{}, I0{});
    issue_w(0, I3{}, I1{});
    c32* dbuf = lds + hf * kHalfX;
    float4* dst = reinterpret_cast<float4*>(dbuf) + ht;
    dst[0] = x0; dst[512] = x1; dst[1024] = x2; dst[1536] = x3;
    load_stray(0, 0);
  }
  __syncthreads();
#pragma nounroll
  for (int i = 0; i < nsteps; ++i) {
    half(i, I0{});
    half(i, I1{});
  }
  // the transforms: all sixteen waves on one image (the staging buffers' memory; the loop's last barrier is behind
  // every read of them), one later event after the other
  auto finish = [&](f2 (&zz)[16], auto kc) {
    constexpr int k = kc;
    const int c = k ? c1 : c0;
    c32 z[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = make_float2(zz[m].x, zz[m].y);
#if DTX_XC16_ABLATE & 8
    if (z[0].x == 1.2345f && z[3].y == 5.4321f) a.keys[1] = 1ull;
#else
    fft16384<1>(z, lds, lb, a.tab, tid);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if DTX_XC16_ABLATE & 4
    if (z[0].x == 1.2345f && z[5].y == 5.4321f) a.keys[1] = 1ull;
    if ((tid & 63) == 0) { wred[2 * k][wv] = 0ull; wred[2 * k + 1][wv] = 0ull; }
    return;
#endif
    // cc = (Y' - (sum x_b) rs_b mu_w) rs_w at the thread's VM lags p0 + tid + 1024 m (see xc_fft22_kernel)
    const char* mwc = reinterpret_cast<const char*>(a.mean_w + (int64_t)c * a.Ppad);
    const char* rwc = reinterpret_cast<const char*>(a.rsw + (int64_t)c * a.Ppad);
    const unsigned last = 4u * (unsigned)(a.Pz - 1);
    float best[2] = {-INFINITY, -INFINITY};
    int bm[2] = {0, 0};
    constexpr int kStat = VM <= 9 ? VM : (VM + 1) / 2;
    dtx::static_for<0, (VM + kStat - 1) / kStat>([&](auto chc) {
      constexpr int m0 = chc * kStat, m1 = m0 + kStat < VM ? m0 + kStat : VM;
      float mu[m1 - m0], rw[m1 - m0];
#pragma unroll
      for (int m = m0; m < m1; ++m) {
        const unsigned p4 = 4u * (unsigned)(p0 + tid + 1024 * m), o = min(p4, last);
        mu[m - m0] = *reinterpret_cast<const float*>(mwc + o);
        const float rv = *reinterpret_cast<const float*>(rwc + o);
        rw[m - m0] = p4 <= last ? rv : __builtin_nanf("");      // NaN (a lag past Pz) never wins a comparison
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int m = m0; m < m1; ++m) {
          const float y = h ? z[m].y : z[m].x;
          float cc = (y - s1v[k][h] * mu[m - m0]) * rw[m - m0];
          cc = (fabsf(cc) > 1.f + kScrubTolF) ? 0.f : cc;           // construct.py:458-462 (NaN stays NaN)
          const bool gt = cc > best[h];
          best[h] = gt ? cc : best[h];
          bm[h] = gt ? m : bm[h];
        }
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float r1 = r1v[k][h];
      const float wmax = wave_max_f32(best[h]);
      const unsigned pcand = (best[h] == wmax) ? (unsigned)(p0 + tid + 1024 * bm[h]) : 0xFFFFFFFFu;
      const unsigned pmin = wave_min_u32(pcand);
      if ((tid & 63) == 0) {
        unsigned long long key = 0ull;
        if (r1 > 0.f && wmax > -INFINITY) {
          const unsigned bits = __float_as_uint(wmax);
          const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
          key = ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - pmin);
        }
        wred[2 * k + h][wv] = key;
      }
    }
  };
  finish(z0, I0{});
  finish(z1, I1{});
  __syncthreads();
  if (tid < 4) {
    unsigned long long key = wred[tid][0];
#pragma unroll
    for (int q = 1; q < kT16 / 64; ++q) key = max(key, wred[tid][q]);
    const int k = tid >> 1, h = tid & 1;
    const int b = 2 * pr + h, c = k ? c1 : c0;
    if (key != 0ull && b < c && (k == 0 || has1)) {
      const int64_t idx = condensed_index_x(a.N, b, c);
      if (idx >= a.pair_begin && idx < a.pair_end) atomicMax(a.keys + idx, key);
    }
  }
}

// pair spectra for the 16 384-point lag search (xc_wspec_kernel one size up): W[pair][vc][w_slot16 order]
__global__ __launch_bounds__(kT16) void xc_wspec16_kernel(const float* __restrict__ td, int N, int n, int Nc, int L,
                                                          int nseg, int seglen, const float* __restrict__ frs,
                                                          const c32* __restrict__ tab, c32* __restrict__ W) {
  __shared__ c32 lds[kN16];
  const int tid = threadIdx.x;
  const int nvc = Nc * nseg;
  const int pr = blockIdx.x / nvc, vc = blockIdx.x - pr * nvc;
  const int ch = vc / nseg, sg = vc - ch * nseg;
  const int a = 2 * pr, b = a + 1;
  const int j0 = sg * seglen;
  const float ra = frs[a], rb = b < N ? frs[b] : 0.f;
  const LdsBases lb = lds_bases(tid);
  c32 v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = 16 * ((tid & 63) + 64 * r) + (tid >> 6);    // point index the transform expects
    float re = 0.f, im = 0.f;
    if (j < seglen && j0 + j < L) {
      const int64_t q = (int64_t)(j0 + j) * Nc + ch;
      re = td[(int64_t)a * n + q] * ra;
      if (b < N) im = td[(int64_t)b * n + q] * rb;
    }
    v[r] = make_float2(re, im);
  }
  fft16384<1>(v, lds, lb, tab, tid);
  c32* dst = W + ((int64_t)pr * nvc + vc) * kN16;     // register m holds frequency tid + 1024 m
  const float sc = 1.0f / (float)kN16;
#pragma unroll
  for (int m = 0; m < 16; ++m) dst[w_slot16(tid + 1024 * m)] = make_float2(v[m].x * sc, v[m].y * sc);
}

template <int VM>
void launch_xc16(dtx_ctx* ctx, const XcFftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((xc_fft16_kernel<VM>), dim3(grid), dim3(kT16), 0, ctx->stream, a);
}

#endif   // DTX_XC16

template <int VM>
void launch_xc22(dtx_ctx* ctx, const XcFftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((xc_fft22_kernel<VM>), dim3(grid), dim3(kT2), 0, ctx->stream, a);
}

template <int VM>
void launch_xc(dtx_ctx* ctx, const XcFftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((xc_fft_kernel<VM>), dim3(grid), dim3(kT), 0, ctx->stream, a);
}

}  // namespace

// Lag search of dtx_xcorr_allpairs through the frequency domain.  z / stats / keys are the
// buffers dtx_xcorr_allpairs prepared; rows b in [b_begin, b_end] hold the pairs of the range.
int dtx_xcfft_run(dtx_ctx* ctx, const float* td_dev, int N, int n, int Nc, int L, int Pz,
                  int64_t Ppad, int64_t Tz, const float* z, const float* fsum, const float* frs,
                  const float* mean_w, const float* rsw, unsigned long long* keys,
                  int64_t pair_begin, int64_t pair_end, int64_t b_begin, int64_t b_end) {
  // Events up to kFftMaxL samples: one segment.  Longer ones are cut into nseg segments of seglen samples;
  // a lag block advances by V = 512 floor((kN - seglen + 1) / 512) lags and (block, segment) meets the later
  // event's padded planes at sample blk V + s seglen (its own block spectrum: nblk x nseg per event and
  // channel).  nseg is chosen by cost: one transform + epilogue per lag block (~7.5 MAC steps' worth, from
  // the kernel's timing-only ablations) and one MAC step per (block, segment, channel) that sees data.  At
  // L = 13 000: segments of 4096 -> 4 blocks x 4 segments (13 products see data); 4 segments of 3250 ->
  // V = 4608, 3 blocks, 11 products -- a quarter fewer transforms and a sixth fewer MAC steps.
  // (host arithmetic: dtxplan::plan_xc_segments / plan_xc_items, dtx_plan.h -- covered by the CPU sanitizer build)
  const int S = (Pz - 1) / 2;
  dtxplan::XcSegPlan sp = dtxplan::plan_xc_segments(L, Pz, Nc, kN, kFftMaxL);
  // One size up (round 5): a 16 384-point kernel searches configs[2]'s 13 001 lags in ONE block
  // (42 against 58 steps' worth per two pair spectra and two later events in the planner's cost model).  Built and
  // parity-green, it ran 157 ms against the 2 x 2 kernel's 126: a workgroup holds one pair spectrum where the 2 x 2
  // kernel holds two, so the bytes through the L1 per multiply-add rise by a third and the products -- bound by
  // that traffic, ~20 TB/s either way -- take as long as before, while the sixteen-wave transforms came out no
  // cheaper (profiles/x11_xc16_ablations.txt).  Not the default anywhere.
  bool t16 = false;
#if DTX_XC16
  // (opt-in build; never the default: measured slower.  DETEX_AMD_XC16=1 selects it where a plan exists)
  const dtxplan::XcSegPlan16 sp16 = dtxplan::plan_xc_segments16(L, Pz, Nc, kN16);
  if (const char* e = std::getenv("DETEX_AMD_XC16")) t16 = sp16.ok && e[0] == '1';
  if (t16) { sp.nseg = sp16.nseg; sp.seglen = sp16.seglen; sp.V = sp16.V; sp.nblk = sp16.nblk; }
#endif
  const int nseg = sp.nseg, seglen = sp.seglen, V = sp.V;
  const int nblk = sp.nblk, nblkx = nblk * nseg;
  const int nvc = Nc * nseg, npr = (N + 1) / 2;
  const int kNx = t16 ? kN16 : kN;                  // transform length of the kernel chosen

  // work items (c, first pair of a unit), unit-major so that concurrent workgroups share the
  // unit's pair spectra in L2.
  // kGroupX units are walked together, kCtileX later events at a time: the workgroups of one later event
  // (one per unit of the group) share its block spectra in L2, those of one unit (one per later event of
  // the tile and lag block) its pair spectra
  static_assert(sizeof(int2) == sizeof(dtxplan::I2), "work items are int2");
  // (DETEX_AMD_XC22=0 / 1 overrides the build's choice of kernel: A/B runs and the tests of both)
  // The 2 x 2 workgroup pays where the products' traffic counts: with more than nine lags per thread (events up
  // to ~3000 samples: one segment, three products per transform, a workgroup's fixed costs on top) the transform
  // of the first later event spills beside the second one's accumulators and the round-3 kernel is as fast or
  // faster (N = 1000 events, scripts/xc_sweep_L.sh, kernel ms: L = 1024 7.9 against 7.3, 2000 7.2 / 7.1,
  // 3000 7.1 / 7.0; 3500 6.4 / 7.0, 4000 6.2 / 7.1, 5633 16.0 / 19.3, 9000 23.4 / 23.8, 20 000 77 / 95;
  // 13 000 at N = 2000: 127 / 152).
  bool t22 = t16 || (DTX_XC_T22 != 0 && V / 512 <= 9);       // (the 16 384-point kernel walks the 2 x 2 kernel's items)
  if (const char* e = std::getenv("DETEX_AMD_XC22")) t22 = t16 || e[0] == '1';
  // The half-size workgroup, two per CU (xc_fft12_kernel), is the default: as fast or faster than either of the others
  // from L = 1024 to 32 768 (scripts/xc_sweep_L.sh, N = 1000, Nc = 3, kernel ms: L = 1024 6.8 / 7.9 / 7.4 for this
  // kernel / the 2 x 2 workgroup / the round-3 kernel, 3000 5.7 / 7.0 / 7.0, 5000 9.9 / 11.3 / 13.3, 9000 19.5 / 22.6 /
  // 23.7, 20 000 74.4 / 75.4 / 95.1) -- except where a lag block has twelve or more product steps (configs[2]: 12;
  // L = 6500: 15; six channels at L = 6000: 18): there the bytes per product decide and the 2 x 2 workgroup moves a
  // quarter fewer (126 against 128 ms at configs[2], 16.6 / 17.2 at L = 6500, 18.1 / 19.8 at Nc = 6).  It walks the
  // 2 x 2 kernel's items, one pair spectrum each.  DETEX_AMD_XC12=1 / 0 forces / forbids it; DETEX_AMD_XC22 names one of
  // the other two.
  int maxsteps = 0;
  for (int b = 0; b < nblk; ++b)
    maxsteps = std::max(maxsteps, Nc * std::min(nseg, std::max(1, (S + L - b * V + seglen - 1) / seglen)));
  bool t12 = DTX_XC_T12 != 0 && !t16 && !std::getenv("DETEX_AMD_XC22") && !(V / 512 <= 9 && maxsteps >= 12);
  if (const char* e = std::getenv("DETEX_AMD_XC12")) t12 = !t16 && e[0] == '1';
  if (t12) t22 = true;
  const std::vector<dtxplan::I2> items = t22 ? dtxplan::plan_xc_items22(N, b_begin, b_end, kGroupX, kCtileX)
                                             : dtxplan::plan_xc_items(N, b_begin, b_end, kUnitX, kGroupX, kCtileX);
  if (items.empty()) return DTX_OK;
  const int64_t nwork = (int64_t)items.size() * ((t16 || t12) ? 2 * nblk : ((DTX_XC_BLKLOOP && !t22) || (DTX_XC_ABLATE & 32) ? 1 : nblk));
  const int64_t per_xcd = (nwork + 7) / 8, grid = per_xcd * 8;
  if (grid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "xcorr (FFT): grid too large");

  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_w = carve((size_t)npr * nvc * kNx * sizeof(float2));
  const size_t o_x = carve((size_t)N * nblkx * Nc * (kNx / 2) * sizeof(float2));
  const size_t o_t = carve((size_t)kNx * sizeof(float2));
  const size_t o_i = carve(items.size() * sizeof(int2));
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_xhat, off))) return rc;
  char* base = (char*)ctx->ws_xhat.p;
  c32* W = (c32*)(base + o_w);
  c32* xh = (c32*)(base + o_x);
  c32* tab = (c32*)(base + o_t);
  {
    std::vector<float2> htab(kNx);
    const double pi = 3.14159265358979323846264338327950288;
    for (int m = 0; m < kNx; ++m)
      htab[m] = make_float2((float)std::cos(2 * pi * m / kNx), (float)std::sin(2 * pi * m / kNx));
    DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the arena may be in use by queued work
    DTX_HIP(ctx, hipMemcpy(tab, htab.data(), htab.size() * sizeof(float2), hipMemcpyHostToDevice));
    DTX_HIP(ctx, hipMemcpy(base + o_i, items.data(), items.size() * sizeof(int2),
                           hipMemcpyHostToDevice));
  }
#if DTX_XC16
  if (t16) {
    hipLaunchKernelGGL(xc_wspec16_kernel, dim3((unsigned)(npr * nvc)), dim3(kT16), 0, ctx->stream, td_dev,
                       N, n, Nc, L, nseg, seglen, frs, (const c32*)tab, W);
    DTX_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(xhat16_kernel, dim3((unsigned)nblkx, (unsigned)Nc, (unsigned)N), dim3(kT16), 0,
                       ctx->stream, z, Tz, Nc, V, nblkx, (const c32*)tab, xh, nseg, nseg == 1 ? 0 : seglen);
    DTX_HIP(ctx, hipGetLastError());
  } else
#endif
  {
  hipLaunchKernelGGL(xc_wspec_kernel, dim3((unsigned)(npr * nvc)), dim3(kT), 0, ctx->stream, td_dev,
                     N, n, Nc, L, nseg, seglen, frs, (const c32*)tab, W);
  DTX_HIP(ctx, hipGetLastError());
  hipLaunchKernelGGL(xhat_kernel, dim3((unsigned)nblkx, (unsigned)Nc, (unsigned)N), dim3(kT), 0,
                     ctx->stream, z, Tz, Nc, V, nblkx, kN, (const c32*)tab, xh, nseg, nseg == 1 ? 0 : seglen);
  DTX_HIP(ctx, hipGetLastError());
  }

  XcFftArgs a;
  a.xh = xh; a.W = W; a.tab = tab; a.items = (const int2*)(base + o_i);
  a.fsum = fsum; a.frs = frs; a.mean_w = mean_w; a.rsw = rsw; a.keys = keys;
  a.Ppad = Ppad; a.pair_begin = pair_begin; a.pair_end = pair_end;
  a.N = N; a.Nc = Nc; a.nseg = nseg; a.nblk = nblk; a.V = V; a.Pz = Pz;
  a.seglen = seglen; a.zend = S + L;
  a.per_xcd = per_xcd; a.nwork = nwork;
  { const int rck = dtx_kev_next(ctx); if (rck) return rck; }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
#if DTX_XC16
#ifdef DTX_XC16_ONLY       // (quick builds while tuning: one instantiation)
  if (t16) switch (V / 1024) {
    case DTX_XC16_ONLY: launch_xc16<DTX_XC16_ONLY>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "xcorr (FFT 16384): built for one block advance only");
  }
  else
#endif
  if (t16) switch (V / 1024) {
    case 4: launch_xc16<4>(ctx, a, (unsigned)grid); break;
    case 5: launch_xc16<5>(ctx, a, (unsigned)grid); break;
    case 6: launch_xc16<6>(ctx, a, (unsigned)grid); break;
    case 7: launch_xc16<7>(ctx, a, (unsigned)grid); break;
    case 8: launch_xc16<8>(ctx, a, (unsigned)grid); break;
    case 9: launch_xc16<9>(ctx, a, (unsigned)grid); break;
    case 10: launch_xc16<10>(ctx, a, (unsigned)grid); break;
    case 11: launch_xc16<11>(ctx, a, (unsigned)grid); break;
    case 12: launch_xc16<12>(ctx, a, (unsigned)grid); break;
    case 13: launch_xc16<13>(ctx, a, (unsigned)grid); break;
    case 14: launch_xc16<14>(ctx, a, (unsigned)grid); break;
    case 15: launch_xc16<15>(ctx, a, (unsigned)grid); break;
    case 16: launch_xc16<16>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "xcorr (FFT 16384): bad block advance %d", V);
  }
  else
#endif
  if (t12) switch (V / 512) {
    case 4: launch_xc12<4>(ctx, a, (unsigned)grid); break;
    case 5: launch_xc12<5>(ctx, a, (unsigned)grid); break;
    case 6: launch_xc12<6>(ctx, a, (unsigned)grid); break;
    case 7: launch_xc12<7>(ctx, a, (unsigned)grid); break;
    case 8: launch_xc12<8>(ctx, a, (unsigned)grid); break;
    case 9: launch_xc12<9>(ctx, a, (unsigned)grid); break;
    case 10: launch_xc12<10>(ctx, a, (unsigned)grid); break;
    case 11: launch_xc12<11>(ctx, a, (unsigned)grid); break;
    case 12: launch_xc12<12>(ctx, a, (unsigned)grid); break;
    case 13: launch_xc12<13>(ctx, a, (unsigned)grid); break;
    case 14: launch_xc12<14>(ctx, a, (unsigned)grid); break;
    case 15: launch_xc12<15>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "xcorr (FFT): bad block advance %d", V);
  }
  else if (t22) switch (V / 512) {
    case 4: launch_xc22<4>(ctx, a, (unsigned)grid); break;
    case 5: launch_xc22<5>(ctx, a, (unsigned)grid); break;
    case 6: launch_xc22<6>(ctx, a, (unsigned)grid); break;
    case 7: launch_xc22<7>(ctx, a, (unsigned)grid); break;
    case 8: launch_xc22<8>(ctx, a, (unsigned)grid); break;
    case 9: launch_xc22<9>(ctx, a, (unsigned)grid); break;
    case 10: launch_xc22<10>(ctx, a, (unsigned)grid); break;
    case 11: launch_xc22<11>(ctx, a, (unsigned)grid); break;
    case 12: launch_xc22<12>(ctx, a, (unsigned)grid); break;
    case 13: launch_xc22<13>(ctx, a, (unsigned)grid); break;
    case 14: launch_xc22<14>(ctx, a, (unsigned)grid); break;
    case 15: launch_xc22<15>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "xcorr (FFT): bad block advance %d", V);
  }
  else switch (V / 512) {
    case 4: launch_xc<4>(ctx, a, (unsigned)grid); break;
    case 5: launch_xc<5>(ctx, a, (unsigned)grid); break;
    case 6: launch_xc<6>(ctx, a, (unsigned)grid); break;
    case 7: launch_xc<7>(ctx, a, (unsigned)grid); break;
    case 8: launch_xc<8>(ctx, a, (unsigned)grid); break;
    case 9: launch_xc<9>(ctx, a, (unsigned)grid); break;
    case 10: launch_xc<10>(ctx, a, (unsigned)grid); break;
    case 11: launch_xc<11>(ctx, a, (unsigned)grid); break;
    case 12: launch_xc<12>(ctx, a, (unsigned)grid); break;
    case 13: launch_xc<13>(ctx, a, (unsigned)grid); break;
    case 14: launch_xc<14>(ctx, a, (unsigned)grid); break;
    case 15: launch_xc<15>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "xcorr (FFT): bad block advance %d", V);
  }
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  dtx_kev_commit(ctx);
  return DTX_OK;
}
