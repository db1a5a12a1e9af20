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
// Work list (profiles/attic/ab_xc.sh, configs[2], kernel ms): rounds 1-2 gave a workgroup 8 pairs, which it walked two at
// a time, streaming its block spectra from HBM once per two pairs (12 MB of them in flight per XCD) past 4.3 MB
// of pair spectra: UNIT 8 / GROUP 1 181.4, 4 / 1 174.3, 2 / 1 181.3, 16 / 1 195.9.  Now a workgroup takes ONE
// step of two pairs, the workgroups of a later event's GROUP units run side by side and share its block
// spectra in L2 (2 / 2 167.4, 2 / 16 165.1, 2 / 32 216: the group's pair spectra must stay in the Infinity
// Cache), and CTILE later events are walked together inside a group so that a unit's pair spectra serve
// them all from L2: CTILE 2 / GROUP 16 162.4, 2 / 8 157.5, 4 / 8 156.1, 6 / 8 160.0, 8 / 8 162.2, 4 / 4 161.8.
// What is left is the sum of the kernel's two phases (products, transforms), which one workgroup per CU runs
// one after the other.
constexpr int kCtileX = 4; // later events walked together inside a group
constexpr int kGroupX = 8; // units whose workgroups run side by side for the same later event

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

// ---- the 2 x 2 workgroup ------------------------------------------------------------------------------------
// Round 3's kernel (one workgroup of 512 threads per later event and unit of two pair spectra: removed in round 6,
// profiles/LADDERS_r2_r5.md) was bound by the sum of its phases under two waves per SIMD, and by 80 KB through a CU's
// L1 per product (a pair spectrum to registers + half of a staged block spectrum).  Here a workgroup of 1024 threads takes
// TWO later events x TWO pair spectra of one lag block: half h (512 threads, the transform's own layout) holds pair
// spectrum pr0 + h in registers and multiplies it with BOTH later events' block spectra, which the halves stage
// together through LDS (each loads one): 2 x 64 + 2 x 32 = 192 KB per four products = 48 KB each, sixteen waves
// of 128 registers instead of eight of 198.  The staging buffers (4 x 32 KB: two events, double-buffered) and the
// halves' two transform images (2 x 64 KB) are the same 128 KB of LDS, used one phase after the other.
constexpr int kT2 = 2 * kT;
template <int VM>
__global__ __launch_bounds__(kT2, 1) void xc_fft22_kernel(const XcFftArgs a) {
  __shared__ c32 lds[2 * kLdsC];
  __shared__ unsigned long long wred[8][kT / 64];   // [(half, later event, row)][wave of the half]
  const int tid = threadIdx.x, hf = tid >> 9, ht = tid & 511;
  const int64_t logical = (int64_t)(blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
  if (logical >= a.nwork) return;
  const int item = (int)(logical / a.nblk), blk = (int)(logical - (int64_t)item * a.nblk);
  const int2 it = a.items[item];
  const int c0 = it.x, has1 = (it.y >> 30) & 1, pr0 = it.y & 0x3fffffff;
  const int c1 = has1 ? c0 + 1 : c0;                       // (without a second event: the first again, its rows dead)
  const int nprt = (a.N + 1) / 2;
  const bool prok = pr0 + hf < nprt;                        // uniform per half
  const int pr = prok ? pr0 + hf : pr0;
  const LdsBases lb = lds_bases(ht);
  const int p0 = blk * a.V;
  const int nse = min(a.nseg, max(1, (a.zend - p0 + a.seglen - 1) / a.seglen));
  const int nsteps = a.Nc * nse;
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
    const int ch = i / nse, sg = i - ch * nse;
    return reinterpret_cast<const float4*>(xhh + ((int64_t)sg * a.Nc + ch) * kXh) + ht;
  };
  const float4* wq = reinterpret_cast<const float4*>(a.W + (int64_t)pr * nvc * kN) + (512 * wv + ln);
  auto woff = [&](int i) {
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
  float4 w[2][2];
  auto issue_w = [&](int i, auto qc, auto sc) {       // batch q of step i (frequencies 4 q .. 4 q + 3) into slot s
    constexpr int q = qc, sl = sc;
    const float4* src = wq + woff(i);
    w[sl][0] = src[64 * (2 * q)];
    w[sl][1] = src[64 * (2 * q + 1)];
  };
  auto mac_q = [&](const c32* x0, const c32* x1, auto qc, auto sc) {
    constexpr int q = qc, sl = sc;
    c32 pa[4], pb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * q + e;
      pa[e] = fix_x(x0[x_slot(r)], r);
      pb[e] = fix_x(x1[x_slot(r)], r);
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
    constexpr int q = qc, g = q + 2;
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
    using S0 = std::integral_constant<int, (ph + 0) % 2>; using S1 = std::integral_constant<int, (ph + 1) % 2>;
    using S2 = std::integral_constant<int, (ph + 2) % 2>; using S3 = std::integral_constant<int, (ph + 3) % 2>;
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
    if constexpr (2 == 4) { issue_w(0, I2c{}, I2c{}); issue_w(0, I3{}, I3{}); }
    float4* dst = reinterpret_cast<float4*>(lds + hf * kXh) + ht;
    dst[0] = x0; dst[512] = x1; dst[1024] = x2; dst[1536] = x3;
  }
  __syncthreads();
  // (NW = 3 -- the slots rotate, three compiled phases of the step -- spills 239 registers; 4 spills 17: 162 ms)
  static_assert(2 == 2 || 2 == 4, "slots that do not rotate");
#pragma nounroll
  for (int i = 0; i < nsteps; ++i) step(i, I0{});
  // the transforms: each half on its own image (the staging buffers' memory; the loop's last barrier is
  // behind every read of them), one later event after the other
  c32* img = lds + hf * kLdsC;
  auto finish = [&](c32 (&z)[16], auto kc) {
    constexpr int k = kc;
    const int c = k ? c1 : c0;
    fft8192<1>(z, img, lb, a.tab, ht);
    __builtin_amdgcn_sched_barrier(0);
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
    w[sl][0] = *reinterpret_cast<const float4*>(src + wlane + 16 * 64 * (2 * q));
    w[sl][1] = *reinterpret_cast<const float4*>(src + wlane + 16 * 64 * (2 * q + 1));
  };
  // the step's block spectra, both later events: 2 x 4096 complex = 4096 float4, eight per thread, a wave's 64 lanes
  // side by side at a wave-uniform LDS address (LDS-DMA: no staging registers, no ds_write)
  [[maybe_unused]] auto stage = [&](int i) {
    const int64_t o = (int64_t)voff(i).x * kXh * 8;
    const char* s0 = reinterpret_cast<const char*>(xh0) + o + 16u * (unsigned)tid;
    const char* s1 = reinterpret_cast<const char*>(xh1) + o + 16u * (unsigned)tid;
    float4* d = reinterpret_cast<float4*>(lds) + 64 * wv;
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
  // the transforms on the staging memory (fft8192's first barrier is behind every wave's last read of it)
  auto finish = [&](c32 (&z)[16], auto kc) {
    constexpr int k = kc;
    const int c = k ? c1 : c0;
    fft8192<1>(z, lds, lb, a.tab, tid);
    __builtin_amdgcn_sched_barrier(0);
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


template <int VM>
void launch_xc22(dtx_ctx* ctx, const XcFftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((xc_fft22_kernel<VM>), dim3(grid), dim3(kT2), 0, ctx->stream, a);
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
  const int nseg = sp.nseg, seglen = sp.seglen, V = sp.V;
  const int nblk = sp.nblk, nblkx = nblk * nseg;
  const int nvc = Nc * nseg, npr = (N + 1) / 2;

  // work items (c, first pair of a unit), unit-major so that concurrent workgroups share the
  // unit's pair spectra in L2.
  // kGroupX units are walked together, kCtileX later events at a time: the workgroups of one later event
  // (one per unit of the group) share its block spectra in L2, those of one unit (one per later event of
  // the tile and lag block) its pair spectra
  static_assert(sizeof(int2) == sizeof(dtxplan::I2), "work items are int2");
  // Two kernels walk the same items (two later events x two pair spectra).  The half-size workgroup, two per CU
  // (xc_fft12_kernel, one pair spectrum per workgroup), is as fast or faster than the 2 x 2 workgroup from L = 1024 to
  // 32 768 (profiles/x12_sweep_L.txt, N = 1000, Nc = 3, kernel ms: L = 1024 6.8 / 7.9, 3000 5.7 / 7.0, 5000 9.9 /
  // 11.3, 9000 19.5 / 22.6, 20 000 74.4 / 75.4) -- except where a lag block has twelve or more product steps
  // (configs[2]: 12; L = 6500: 15; six channels at L = 6000: 18): there the bytes per product decide and the 2 x 2
  // workgroup moves a quarter fewer (126 against 128 ms at configs[2], 16.6 / 17.2 at L = 6500, 18.1 / 19.8 at Nc = 6).
  // DETEX_AMD_XC=12 / 22 names one (A/B runs, and the tests that hold the two bit-identical).
  int maxsteps = 0;
  for (int b = 0; b < nblk; ++b)
    maxsteps = std::max(maxsteps, Nc * std::min(nseg, std::max(1, (S + L - b * V + seglen - 1) / seglen)));
  bool t12 = !(V / 512 <= 9 && maxsteps >= 12);
  if (const char* e = std::getenv("DETEX_AMD_XC")) t12 = !(e[0] == '2' && e[1] == '2');
  const std::vector<dtxplan::I2> items = dtxplan::plan_xc_items22(N, b_begin, b_end, kGroupX, kCtileX);
  if (items.empty()) return DTX_OK;
  const int64_t nwork = (int64_t)items.size() * (t12 ? 2 * nblk : nblk);
  const int64_t per_xcd = (nwork + 7) / 8, grid = per_xcd * 8;
  if (grid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "xcorr (FFT): grid too large");

  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_w = carve((size_t)npr * nvc * kN * sizeof(float2));
  const size_t o_x = carve((size_t)N * nblkx * Nc * (kN / 2) * sizeof(float2));
  const size_t o_t = carve((size_t)kN * sizeof(float2));
  const size_t o_i = carve(items.size() * sizeof(int2));
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_xhat, off))) return rc;
  char* base = (char*)ctx->ws_xhat.p;
  c32* W = (c32*)(base + o_w);
  c32* xh = (c32*)(base + o_x);
  c32* tab = (c32*)(base + o_t);
  {
    std::vector<float2> htab(kN);
    const double pi = 3.14159265358979323846264338327950288;
    for (int m = 0; m < kN; ++m)
      htab[m] = make_float2((float)std::cos(2 * pi * m / kN), (float)std::sin(2 * pi * m / kN));
    DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the arena may be in use by queued work
    DTX_HIP(ctx, hipMemcpy(tab, htab.data(), htab.size() * sizeof(float2), hipMemcpyHostToDevice));
    DTX_HIP(ctx, hipMemcpy(base + o_i, items.data(), items.size() * sizeof(int2),
                           hipMemcpyHostToDevice));
  }
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
  {
    char nm[64];
    snprintf(nm, sizeof nm, "%s<%d>", t12 ? "xc_fft12_kernel" : "xc_fft22_kernel", V / 512);
    ctx->last_kernel = nm;
  }
  { const int rck = dtx_kev_next(ctx); if (rck) return rck; }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
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
  else switch (V / 512) {
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
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  dtx_kev_commit(ctx);
  return DTX_OK;
}
