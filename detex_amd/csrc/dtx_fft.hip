// Frequency-domain (overlap-save) engine of the subspace detection statistic.
//
// The reference computes U_k . w_p for all p with one FFT-sized product per basis
// vector (detex/detect.py:559-578: ifft(ssFD * MPconFD), float64, one 2^21-point IFFT per
// vector and chunk).  This engine is the same idea laid out for CDNA4:
//
//   * the channel planes x_c of a chunk are cut into blocks of kN = 8192 samples that
//     advance by V = 512 * floor((kN - L + 1) / 512) samples; block spectra
//     X_c[f] = sum_t x_c[t0 + t] e^{-2 pi i t f / kN} are computed once per chunk
//     (xhat_kernel) and stay in HBM / L2 ([chunk][block][c][kN] complex64);
//   * basis vectors are taken in PAIRS (k, k+1): with
//         W_c[f] = 1/kN sum_j (u_k,c[j] + i u_k+1,c[j]) e^{+2 pi i j f / kN}
//     (built once per upload, in float64 on the host), the inverse transform of
//         Z[f] = sum_c W_c[f] X_c[f]
//     holds U_k . w_p in its real part and U_k+1 . w_p in its imaginary part for the V
//     windows p of the block: two real correlations per complex FFT, no Hermitian
//     bookkeeping;
//   * one workgroup (512 threads) owns one (unit, chunk, block): a unit is one detector
//     (or two single-vector detectors sharing a pair).  It loops over the unit's pairs:
//     16 frequencies per thread x Nc complex MACs (basis spectra straight from L2), an
//     in-LDS 8192-point FFT, and the epilogue (Y - S_k mu_p)^2 accumulated in registers; DS leaves the
//     chip once, already normalised -- the K x P matrix of projections never exists.
//
// The FFT is decimated over the workgroup's 8 waves (n = 8 n2 + w): each wave transforms its
// own 1024-point subsequence (radices 16 x 16 x 4, 16 points per lane) through wave-private
// LDS exchanges that need no s_barrier, then one exchange across the waves feeds a radix-8
// pass that leaves window (tid + 512 m) in register m: coalesced DS stores, and only
// m < V / 512 is kept.  Basis and block spectra are stored in that wave-major order, the
// block's own stay in LDS beside the 64 KB image (three channels: exactly the CU's 160 KB).
//
// Work: per (pair, block) 13 * 8192 / 2 butterflies instead of 2 * 2 * Nc * L * V MACs --
// ~150x fewer flops than the direct contraction; float32 error grows with log kN
// instead of sqrt(n) (measured 2e-7 at DS = 1 against 2e-6 for the MFMA engines).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dtx_fftcore.h"
#include "dtx_plan.h"

namespace {

// ---------------------------------------------------------------- the scan
// Dynamic-range guard (see dtx_subspace_ds): block cb = chunk * nblk + blk is "loud" when the
// loudest window over the samples its transforms read holds > kFftGuard^2 times the energy of
// its quietest live own window.  One workgroup per block; a loud block is appended to DtxRepairWs::list (with the two
// energies): spike_split_kernel decides what becomes of it.  wspan: windows whose samples the block's transforms cover -- V for a
// single-segment template (xhat_kernel zeroes the tail no kept output reads), 2 V for a
// segmented one (segment s meets block blk + s: samples up to (nseg + 1) V are transformed, and
// windows w0 .. w0 + 2 V - 1 reach sample w0 + 2 V + L - 2 >= w0 + (nseg + 1) V - 1).
__global__ __launch_bounds__(256) void guard_kernel(const float* __restrict__ den, int64_t Ppad,
                                                    int64_t P, int V, int nblk, int wspan,
                                                    const DtxRepairWs ws, const int* __restrict__ pb) {
  __shared__ float red[8];
  const int cb = blockIdx.x, tid = threadIdx.x;
  const int chunk = cb / nblk, blk = cb - chunk * nblk;
  if (pb != nullptr) P = min(P, (int64_t)pb[chunk]);     // ragged batch: the chunk's own windows
  const int64_t w0 = (int64_t)blk * V, w1 = min(w0 + V, P), w2 = min(w0 + wspan, P);
  const float* d = den + (int64_t)chunk * Ppad;
  float mn = 3.0e38f, mx = 0.f;
  // (eight loads in flight: one at a time the 20 .. 32 trips were a chain of exposed latencies, 34 us per
  // station-day for 35 MB)
  for (int64_t p0 = w0 + tid; p0 < w2; p0 += 8 * 256) {
    float dn[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dn[k] = p0 + 256 * k < w2 ? d[p0 + 256 * k] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (dn[k] > 0.f) {
        if (p0 + 256 * k < w1) mn = fminf(mn, dn[k]);
        mx = fmaxf(mx, dn[k]);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if ((tid & 63) == 0) { red[tid >> 6] = mn; red[4 + (tid >> 6)] = mx; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int q = 1; q < 4; ++q) { mn = fminf(mn, red[q]); mx = fmaxf(mx, red[4 + q]); }
    if (mx > kFftGuard * kFftGuard * mn) {
      ws.list[atomicAdd(ws.count, 1)] = cb;
      ws.gmn[cb] = mn;
      ws.gmx[cb] = mx;
    }
  }
}

// What becomes of a listed block (one workgroup each; a launch that finds an empty list on ordinary data).
//
// (1) Outliers.  Most listings on real records are a glitch: a few samples thousands of times the noise.  The
//     transforms are linear, so such samples need not go through them: a sample whose square exceeds a quarter of
//     what the guard tolerates for a WINDOW (theta = kFftGuard^2 / 4 x the quietest live window's energy) is taken
//     out of the block (spk: at most kMaxSpk per block, in (frame, channel) order) -- xhat_clean_kernel transforms
//     the block again without them, ds_fft_rescan_kernel scans it again and adds their products U[k][t - p] v to
//     the projections directly: K L multiply-adds per outlier instead of a block of the float32 contraction
//     (18 GFLOP at configs[1]).  Only single-segment templates (the block's samples fit the workgroup's LDS).
// (2) What is still out of range.  emax = the loudest window energy of what the transforms now see (prefix sums
//     of the frame energies, float64 in LDS).  A window is in need when emax > kFftGuard^2 x its own energy (den:
//     a window that holds an outlier is loud and never in need); every column tile of kRepairPT windows that holds
//     one becomes a repair task for the float32 contraction, which stores the windows below emax / kFftKeep.
//     (Round 5 sent ALL 22 column tiles of a listed block there; a zero-filled gap's edge next to an event has
//     its windows in need in one or two.)
constexpr int kSplitT = 512;
__global__ __launch_bounds__(kSplitT) void spike_split_kernel(const float* __restrict__ xp, int64_t Tpad, int Nc,
                                                              int L, const float* __restrict__ den, int64_t Ppad,
                                                              int64_t P, int V, int nblk, int nseg,
                                                              const DtxRepairWs ws, const int* __restrict__ pb) {
  __shared__ double sS[kN + 1];                // prefix sums of the frame energies
  __shared__ double swave[kSplitT / 64];
  __shared__ float sred[kSplitT / 64];
  __shared__ DtxSpike sp[kMaxSpk];
  __shared__ int scnt;
  __shared__ int sneed[kN / kRepairPT];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nlist = ws.count[0];
  for (int e = blockIdx.x; e < nlist; e += gridDim.x) {
    const int cb = ws.list[e];
    const int chunk = cb / nblk, blk = cb - chunk * nblk;
    const int64_t Pc = pb != nullptr ? min(P, (int64_t)pb[chunk]) : P;
    const int64_t w0 = (int64_t)blk * V;
    const int nwin = (int)(min(w0 + V, Pc) - w0);
    float emax = ws.gmx[cb];
    int ns = 0;
    if (nseg == 1) {
      const int F = (int)min((int64_t)(nwin + L - 1), Tpad - w0);     // frames the block's windows read (<= kN)
      const float theta = 0.25f * kFftGuard * kFftGuard * ws.gmn[cb];
      const float* xb = xp + (int64_t)chunk * Nc * Tpad + w0;
      if (tid == 0) scnt = 0;
      __syncthreads();
      for (int c = 0; c < Nc; ++c) {
        const float* x = xb + (int64_t)c * Tpad;
        for (int t = tid; t < F; t += kSplitT) {
          const float v = x[t];
          if (v * v > theta) {
            const int k = atomicAdd(&scnt, 1);
            if (k < kMaxSpk) sp[k] = DtxSpike{(int)w0 + t, c, v};
          }
        }
      }
      __syncthreads();
      ns = scnt;
      const bool split = ns >= 1 && ns <= kMaxSpk;
      if (!split) ns = 0;
      if (split) {                                     // (frame, channel) order: the products are added in a fixed order
        DtxSpike mine = DtxSpike{0, 0, 0.f};
        int rank = 0;
        if (tid < ns) {
          mine = sp[tid];
          const int64_t key = (int64_t)mine.t * Nc + mine.c;
          for (int j = 0; j < ns; ++j) rank += ((int64_t)sp[j].t * Nc + sp[j].c) < key;
        }
        __syncthreads();
        if (tid < ns) sp[rank] = mine;
        __syncthreads();
      }
      // frame energies of what the transforms see, a thread's kN / kSplitT consecutive frames; inclusive scan
      constexpr int kPer = kN / kSplitT;
      double loc[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) loc[q] = 0.0;
      for (int c = 0; c < Nc; ++c) {
        const float* x = xb + (int64_t)c * Tpad;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const int t = kPer * tid + q;
          const float v = t < F ? x[t] : 0.f;
          if (!(split && v * v > theta)) loc[q] += (double)v * (double)v;
        }
      }
#pragma unroll
      for (int q = 1; q < kPer; ++q) loc[q] += loc[q - 1];
      double incl = loc[kPer - 1];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (lane == 63) swave[wv] = incl;
      __syncthreads();
      double base = incl - loc[kPer - 1];
      for (int w = 0; w < wv; ++w) base += swave[w];
      if (tid == 0) sS[0] = 0.0;
#pragma unroll
      for (int q = 0; q < kPer; ++q) sS[kPer * tid + q + 1] = base + loc[q];
      __syncthreads();
      float m = 0.f;
      for (int p = tid; p < nwin; p += kSplitT) m = fmaxf(m, (float)(sS[min(p + L, F)] - sS[p]));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) sred[wv] = m;
      __syncthreads();
      emax = sred[0];
#pragma unroll
      for (int w = 1; w < kSplitT / 64; ++w) emax = fmaxf(emax, sred[w]);
    }
    if (tid < kN / kRepairPT) sneed[tid] = 0;
    __syncthreads();
    const float* d = den + (int64_t)chunk * Ppad + w0;
    for (int p = tid; p < nwin; p += kSplitT) {
      const float dn = d[p];
      if (dn > 0.f && kFftGuard * kFftGuard * dn < emax) sneed[p / kRepairPT] = 1;
    }
    __syncthreads();
    if (tid < (nwin + kRepairPT - 1) / kRepairPT && sneed[tid])
      ws.tasks[atomicAdd(ws.count + 1, 1)] =
          make_int4(chunk, (int)(w0 / kRepairPT) + tid, __float_as_int(emax * (1.f / kFftKeep)), cb);
    if (tid == 0) {
      ws.nspk[cb] = ns;
      if (ns > 0) ws.slist[atomicAdd(ws.count + 2, 1)] = cb;
    }
    if (tid < ns) ws.spk[(int64_t)cb * kMaxSpk + tid] = sp[tid];
    __syncthreads();
  }
}

// The block spectra of the blocks whose outliers were split off, again without them, into a buffer of their own
// (same layout as xhat_kernel's: on the side stream, beside it -- the scan keeps reading the originals).  Grid (any, Nc).
__global__ __launch_bounds__(kT) void xhat_clean_kernel(const float* __restrict__ xp, int64_t Tpad, int Nc, int V,
                                                        int nblk, int nblkx, int keep,
                                                        const c32* __restrict__ tab, c32* __restrict__ xh,
                                                        const DtxRepairWs ws) {
  __shared__ c32 lds[kLdsC];
  const int tid = threadIdx.x, c = blockIdx.y;
  const LdsBases lb = lds_bases(tid);
  const int tb = 8 * (tid & 63) + (tid >> 6);          // sample within the block
  const int nsl = ws.count[2];
  for (int e = blockIdx.x; e < nsl; e += gridDim.x) {
    const int cb = ws.slist[e];
    const int chunk = cb / nblk, blk = cb - chunk * nblk;
    const float* x = xp + ((int64_t)chunk * Nc + c) * Tpad;
    const int64_t t0 = (int64_t)blk * V + tb;
    c32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t t = t0 + 512 * r;
      v[r] = make_float2((t < Tpad && tb + 512 * r < keep) ? x[t] : 0.f, 0.f);
    }
    const int ns = ws.nspk[cb];
    for (int s = 0; s < ns; ++s) {
      const DtxSpike sk = ws.spk[(int64_t)cb * kMaxSpk + s];
      const int rel = sk.t - blk * V - tb;             // this thread's sample r when rel == 512 r
      if (sk.c == c && rel >= 0 && (rel & 511) == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (rel == 512 * r) v[r].x = 0.f;
      }
    }
    fft8192<-1>(v, lds, lb, tab, tid);
    c32* row = xh + (((int64_t)chunk * nblkx + blk) * Nc + c) * kXh + (tid >> 6) * 512 + (tid & 63);
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
    __syncthreads();
  }
}

constexpr int kMaxGroups = 32;
struct FftArgs {
  const c32* xh;        // [nchunk][nblkx][Nc][kXh] half spectra, nblkx = nblk + nseg - 1
  const c32* xhc;       // the same layout: spectra WITHOUT the outliers of the blocks that had some split off (xhat_clean_kernel)
  const c32* W;         // [npairs][Nc][kN]
  const c32* tab;       // [kN] e^{+2 pi i m / kN}
  const int4* unit;     // [nunits] {first pair, pairs, -, -}: consecutive pairs of whole detectors
  const float4* pinfo;  // [npairs] {S_k, S_k+1, -, -} row sums of the real- and imaginary-part vectors
  const int2* pdet;     // [npairs] {detector closed by this pair or -1, single-vector detector of the imaginary part or -1}
  const float* mean;    // [nchunk][Ppad]
  const float* den;     // [nchunk][Ppad]
  float* ds;            // [nchunk][D][ldp]
  int64_t Ppad, ldp, P;
  int Nc, D, V, nblk, nchunk;
  int nunits, ntile_b, per_xcd;   // tiling of the grid (launch_fft)
  // ping-pong kernel: the pairs are stored half 0's first (every unit's share, in unit order), then
  // half 1's; upre[u] = {pairs of half 0, pairs of half 1} in the units before u.  The first nfull
  // blocks get one workgroup each that runs ALL units; each of the other nrem blocks is split over
  // ngroups workgroups, group g running units gb[g] .. gb[g + 1] - 1 (dtx_fft_scan sizes this so
  // that the last wave of workgroups is short)
  const int2* upre;     // [nunits + 1]
  int nfull, nrem, ngroups;
  short gb[kMaxGroups + 1];
  int nseg, nblkx;      // template segments (virtual channels per real channel), block spectra stored per chunk
  // ping-pong kernel with DTX_PP_INLINE_X: the block spectra are taken inside the kernel from the channel planes
  const float* xp;      // [nchunk][Nc][Tpad]
  int64_t Tpad;
  int keep;
  const int* pb;        // ragged batch: [nchunk] windows of each chunk (NULL: P for all)
  // re-scan of the blocks whose outliers were split off (ds_fft_rescan_kernel)
  DtxRepairWs rws;
  const int2* runs;     // [nruns] {first pair, pairs}: runs of whole detectors
  const int2* prow;     // [npairs] basis rows of the real / imaginary part
  const float* U;       // [Ktot][n] the basis itself
  int nruns, n, L;
};

#define DTX_DS_STORE(v, p) __builtin_nontemporal_store((v), (p))
// epilogue of one pair (uses z, pr, ui, mean, den, acc0, acc1 of the enclosing kernel): add the
// squared projections; when the pair closes its detector(s) (pdet.x >= 0), normalise and store.
// pdet.y >= 0: the imaginary part is a different, single-vector detector (acc1).
#define DTX_FFT_EPILOGUE \
    add_spikes(ui.x + pr); \
    const float4 pi = a.pinfo[ui.x + pr]; \
    const int2 pd = a.pdet[ui.x + pr]; \
    const bool two = TWO && pd.y >= 0; \
_Pragma("unroll") \
    for (int m = 0; m < VM; ++m) { \
      const float mu = (mean + 512 * m)[tid]; \
      /* explicit fma chain: every instantiation of the kernel rounds the same way */ \
      const float re = __fmaf_rn(-pi.x, mu, z[m].x), im = __fmaf_rn(-pi.y, mu, z[m].y); \
      const float i0 = two ? 0.f : im, i1 = two ? im : 0.f; \
      acc0[m] = __fmaf_rn(i0, i0, __fmaf_rn(re, re, acc0[m])); \
      if (TWO) acc1[m] = __fmaf_rn(i1, i1, acc1[m]); \
    } \
    if (pd.x >= 0) { \
_Pragma("unroll") \
      for (int m = 0; m < VM; ++m) { \
        const int64_t p = p0 + tid + 512 * m; \
        if (p < a.P) { \
          const float dn = (den + 512 * m)[tid]; \
          float v0 = (dn > 0.f) ? __fdiv_rn(acc0[m], dn) : 0.f; \
          if (__builtin_isinf(v0)) v0 = 0.f; \
          DTX_DS_STORE(v0, &a.ds[((int64_t)chunk * a.D + pd.x) * a.ldp + p]); \
          if (two) { \
            float v1 = (dn > 0.f) ? __fdiv_rn(acc1[m], dn) : 0.f; \
            if (__builtin_isinf(v1)) v1 = 0.f; \
            DTX_DS_STORE(v1, &a.ds[((int64_t)chunk * a.D + pd.y) * a.ldp + p]); \
          } \
        } \
        acc0[m] = 0.f; \
        if (TWO) acc1[m] = 0.f; \
      } \
    }

// VM = V / 512 outputs kept per thread; TWO: some pair holds two single-vector detectors
// NCH > 0: Nc == NCH <= 3 and the block's spectra stay in LDS beside the image (64 + 32 NCH
// KB: with three channels exactly the CU's 160 KB), so each pair reads only its basis
// spectra from L2 -- half the bytes; NCH == 0: any Nc, block spectra re-read per pair
// SPLIT (the re-scan): the block's outliers (a.rws.spk) did not go through the transforms; their products with
// the pair's two basis rows are added to the finished projections, window by window
template <int VM, bool TWO, int NCH, bool SPLIT>
__device__ __forceinline__ void ds_fft_block(const FftArgs& a, const int cb, const int4 ui, c32* lds, c32* xst,
                                             const int tid) {
  const int chunk = cb / a.nblk, blk = cb - chunk * a.nblk;
  const int64_t p0 = (int64_t)blk * a.V;
  [[maybe_unused]] const int nspk = SPLIT ? a.rws.nspk[cb] : 0;
  [[maybe_unused]] const DtxSpike* spk = SPLIT ? a.rws.spk + (int64_t)cb * kMaxSpk : nullptr;

  const LdsBases lb = lds_bases(tid);
  float acc0[VM], acc1[TWO ? VM : 1];
#pragma unroll
  for (int m = 0; m < VM; ++m) {
    acc0[m] = 0.f;
    if (TWO) acc1[m] = 0.f;
  }
  // p0 + 512 VM <= Ppad + V: mean / den are read unguarded (rows are padded, see dtx_fft_scan)
  const float* mean = a.mean + (int64_t)chunk * a.Ppad + p0;   // uniform bases, indexed + tid
  const float* den = a.den + (int64_t)chunk * a.Ppad + p0;
  // the thread's 16 frequencies: wave w, lane l, register r <-> f = 8 (l + 64 r) + w.  In a
  // stored half spectrum (slot(f) = (f & 7) * 512 + (f >> 3), kXh slots per channel):
  // r < 8: slot 512 w + l + 64 r;  r >= 8: kN - f, conjugated, i.e. slot xhi - 64 r with
  // xhi = 512 (8 - w) + 1023 - l (w > 0) or 1024 - l (w = 0) -- except thread 0, whose f = 0
  // and f = kN / 2 are the two reals of slot 0
  const int wv = tid >> 6, ln = tid & 63;
  const int xlo = 512 * wv + ln, xhi = wv ? 512 * (8 - wv) + 1023 - ln : 1024 - ln;
  auto fix_x = [&](c32 xx, int r) {
    if (r == 0 && tid == 0) xx.y = 0.f;
    if (r == 8 && tid == 0) return make_float2(xx.y, 0.f);
    if (r >= 8) xx.y = -xx.y;
    return xx;
  };
  auto x_slot = [&](int r) { return r < 8 ? xlo + 64 * r : ((r == 8 && tid == 0) ? 0 : xhi - 64 * r); };
  const int wlo = 512 * wv + ln;              // basis spectra: see w_slot (float4 units)
  // block spectra of this block; a template longer than one segment also reads the following
  // blocks': segment s of the template meets the data of block blk + s (advance == segment length)
  const c32* xh = (SPLIT ? a.xhc : a.xh) + ((int64_t)chunk * a.nblkx + blk) * a.Nc * kXh;
  c32* xs = lds + kLdsC;                      // NCH > 0: the block's spectra, [NCH][kXh]
  if constexpr (NCH > 0) {
    const float4* src = reinterpret_cast<const float4*>(xh);
    float4* dst = reinterpret_cast<float4*>(xs);
#pragma unroll
    for (int q = 0; q < NCH * kXh / 2 / kT; ++q) dst[tid + kT * q] = src[tid + kT * q];
    __syncthreads();
  }
  // MAC of one channel: z += W_c X_c at the thread's 16 frequencies
  auto mac16 = [&](c32 (&z)[16], const c32 (&ww)[16], const c32 (&xx)[16]) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      // (two fused multiply-adds per component; a sum of products compiles to mul + fma + add)
      z[m].x = __fmaf_rn(-ww[m].y, xx[m].y, __fmaf_rn(ww[m].x, xx[m].x, z[m].x));
      z[m].y = __fmaf_rn(ww[m].y, xx[m].x, __fmaf_rn(ww[m].x, xx[m].y, z[m].y));
    }
  };
  c32 z[16];
  [[maybe_unused]] auto add_spikes = [&](int pair) {
    if constexpr (SPLIT) {
      const int2 rows = a.prow[pair];
      const float* ua = a.U + (size_t)rows.x * a.n;
      const float* ub = a.U + (size_t)(rows.y >= 0 ? rows.y : rows.x) * a.n;
      const float sb = rows.y >= 0 ? 1.f : 0.f;
      for (int s = 0; s < nspk; ++s) {
        const DtxSpike sk = spk[s];                            // (uniform)
        const int j0 = sk.t - (int)p0 - tid;                   // sample of the template that meets the outlier, window m = 0
#pragma unroll
        for (int m = 0; m < VM; ++m) {
          const int j = j0 - 512 * m;
          if ((unsigned)j < (unsigned)a.L) {
            const int q = j * a.Nc + sk.c;
            z[m].x = __fmaf_rn(ua[q], sk.v, z[m].x);
            z[m].y = __fmaf_rn(ub[q] * sb, sk.v, z[m].y);
          }
        }
      }
    }
  };
  if constexpr (NCH == 0) {
    // Block spectra streamed from L2 (Nc not 1 / 3, or a segmented template): pairs are taken two
    // at a time so that one load of a block spectrum feeds two MACs -- the MAC's L2 traffic is
    // what bounds this path (the same step took the cross-correlation kernel from 516 to 402 ms).
    const int nvc = a.Nc * a.nseg;
    for (int pr0 = 0; pr0 < ui.y; pr0 += 2) {
      const bool has2 = pr0 + 1 < ui.y;
      const c32* wpa = a.W + (int64_t)(ui.x + pr0) * nvc * kN;   // uniform
      const c32* wpb = wpa + (has2 ? (int64_t)nvc * kN : 0);
      c32 zb[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) z[m] = zb[m] = make_float2(0.f, 0.f);
      // The block spectrum of a virtual channel is staged through LDS, a channel ahead: four 16-byte
      // loads per thread instead of sixteen mirrored 8-byte ones (as in xc_fft_kernel, where the
      // sixteen cost as much as all the pair spectra; this kernel too runs one workgroup per CU, so
      // the LDS beside the image is free).  As there (round 3): the four staged values live in NAMED
      // registers (as an array the compiler moved them to LDS itself -- load, wait, LDS store, LDS load,
      // LDS store), the basis spectra come in half-step batches issued a step ahead, and the step is
      // straight-line (its last pass re-fetches its own operands instead of branching).
      auto xsrc = [&](int vc) {
        const int c = vc / a.nseg, sg = vc - c * a.nseg;   // (channel, template segment)
        return reinterpret_cast<const float4*>(xh + ((int64_t)sg * a.Nc + c) * kXh) + tid;
      };
      const float4* wqa = reinterpret_cast<const float4*>(wpa) + wlo;
      const float4* wqb = reinterpret_cast<const float4*>(wpb) + wlo;
      float4 wa[2][4], wb[2][4];
      auto issue_w = [&](int vc, auto hc) {
        constexpr int h = hc;
        const int64_t o = (int64_t)vc * (kN / 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wa[h][j] = (wqa + o + 64 * (4 * h + j))[0];
          wb[h][j] = (wqb + o + 64 * (4 * h + j))[0];
        }
      };
      auto mac_half = [&](const c32* xc, auto hc) {
        constexpr int h = hc;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          c32 px[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) px[e] = fix_x(xc[x_slot(8 * h + 4 * g + e)], 8 * h + 4 * g + e);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int m = 8 * h + 4 * g + e;
            const float4 ta = wa[h][2 * g + (e >> 1)], tb = wb[h][2 * g + (e >> 1)];
            const c32 pa = (e & 1) ? make_float2(ta.z, ta.w) : make_float2(ta.x, ta.y);
            const c32 pb = (e & 1) ? make_float2(tb.z, tb.w) : make_float2(tb.x, tb.y);
            z[m].x = __fmaf_rn(-pa.y, px[e].y, __fmaf_rn(pa.x, px[e].x, z[m].x));
            z[m].y = __fmaf_rn(pa.y, px[e].x, __fmaf_rn(pa.x, px[e].y, z[m].y));
            zb[m].x = __fmaf_rn(-pb.y, px[e].y, __fmaf_rn(pb.x, px[e].x, zb[m].x));
            zb[m].y = __fmaf_rn(pb.y, px[e].x, __fmaf_rn(pb.x, px[e].y, zb[m].y));
          }
        }
      };
      {
        const float4* src = xsrc(0);
        const float4 x0 = src[0], x1 = src[kT], x2 = src[2 * kT], x3 = src[3 * kT];
        issue_w(0, std::integral_constant<int, 0>{});
        issue_w(0, std::integral_constant<int, 1>{});
        float4* dst = reinterpret_cast<float4*>(xst) + tid;
        dst[0] = x0; dst[kT] = x1; dst[2 * kT] = x2; dst[3 * kT] = x3;
      }
      __syncthreads();
#pragma nounroll
      for (int vc = 0; vc < nvc; ++vc) {
        const int nx = min(vc + 1, nvc - 1);      // uniform
        const float4* src = xsrc(nx);
        const float4 x0 = src[0], x1 = src[kT], x2 = src[2 * kT], x3 = src[3 * kT];
        __builtin_amdgcn_sched_barrier(0);
        const c32* xc = xst + (vc & 1) * kXh;
        mac_half(xc, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(nx, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mac_half(xc, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        issue_w(nx, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        float4* dst = reinterpret_cast<float4*>(xst + ((vc + 1) & 1) * kXh) + tid;
        dst[0] = x0; dst[kT] = x1; dst[2 * kT] = x2; dst[3 * kT] = x3;
        __syncthreads();
      }
      {
        const int pr = pr0;
        fft8192<1>(z, lds, lb, a.tab, tid);   // (its leading barrier covers the reuse of the image)
        DTX_FFT_EPILOGUE
      }
      if (has2) {
        const int pr = pr0 + 1;
#pragma unroll
        for (int m = 0; m < 16; ++m) z[m] = zb[m];
        fft8192<1>(z, lds, lb, a.tab, tid);
        DTX_FFT_EPILOGUE
      }
    }
  } else {
    for (int pr = 0; pr < ui.y; ++pr) {
      const c32* wp = a.W + (int64_t)(ui.x + pr) * NCH * kN;   // uniform
#pragma unroll
      for (int m = 0; m < 16; ++m) z[m] = make_float2(0.f, 0.f);
      // a whole channel batch of loads ahead of its MACs: memory-level parallelism is what
      // this phase lives on (the same loop with each load next to its MAC ran 40 % slower).
      // Addresses are kept as (uniform base) + lane offset: SGPRs + one VGPR.
#pragma nounroll
      for (int c = 0; c < NCH; ++c) {
        c32 pw[16], px[16];
        const c32* wc = wp + (int64_t)c * kN;
        w_load16(wc, wlo, pw);
#pragma unroll
        for (int r = 0; r < 16; ++r) px[r] = fix_x(xs[c * kXh + x_slot(r)], r);
        mac16(z, pw, px);
      }
      fft8192<1>(z, lds, lb, a.tab, tid);   // (its leading barrier covers the reuse of the image)
      DTX_FFT_EPILOGUE
    }
  }
}

// VM = V / 512 outputs kept per thread; TWO: some pair holds two single-vector detectors
// NCH > 0: Nc == NCH <= 3 and the block's spectra stay in LDS beside the image (64 + 32 NCH
// KB: with three channels exactly the CU's 160 KB), so each pair reads only its basis
// spectra from L2 -- half the bytes; NCH == 0: any Nc, block spectra re-read per pair
template <int VM, bool TWO, int NCH>
__global__ __launch_bounds__(kT, 2) void ds_fft_kernel(const FftArgs a) {
  __shared__ c32 lds[kLdsC + (NCH > 0 ? NCH * kXh : 0)];
  __shared__ c32 xst[NCH == 0 ? 2 * kXh : 1];   // NCH == 0: the current / next virtual channel's block spectrum
  // Workgroups are dealt to the 8 XCDs round-robin; each XCD has its own 4 MB L2.  Give
  // every XCD a contiguous run of the logical work list, and order that list in tiles of
  // kTileU units x kTileB blocks (1 x 16): the workgroups an XCD runs at a time then read the same
  // unit's basis spectra and the same few blocks' spectra instead of unrelated ones; consecutive
  // tiles are the same blocks under the next unit, so the block spectra are fetched into the
  // L2 once per tile, not once per unit.
  const int logical = (int)(blockIdx.x & 7) * a.per_xcd + (int)(blockIdx.x >> 3);
  const int tile = logical / kTile, slot = logical - tile * kTile;
  const int ntu = (a.nunits + kTileU - 1) / kTileU;
  const int tb = tile / ntu, tu = tile - tb * ntu;     // consecutive tiles: the same blocks under the next unit
  const int unit = tu * kTileU + slot % kTileU;
  const int cb = tb * kTileB + slot / kTileU;          // chunk * nblk + blk
  if (unit >= a.nunits || cb >= a.nchunk * a.nblk) return;
  ds_fft_block<VM, TWO, NCH, false>(a, cb, a.unit[unit], lds, xst, threadIdx.x);
}

// The blocks whose outliers were split off (spike_split_kernel), scanned again on their clean spectra
// (xhat_clean_kernel) with the outliers' products added directly.  Work items (block, run of whole detectors'
// pairs), walked by a grid sized for the chip in tiles of 16 blocks x all runs (the workgroups running at a time
// share a few blocks' spectra and walk the same pair spectra); finds an empty list on ordinary data.
template <int VM, bool TWO, int NCH>
__global__ __launch_bounds__(kT, 2) void ds_fft_rescan_kernel(const FftArgs a) {
  __shared__ c32 lds[kLdsC + (NCH > 0 ? NCH * kXh : 0)];
  __shared__ c32 xst[NCH == 0 ? 2 * kXh : 1];
  constexpr int kTB = 16;
  const int nsl = a.rws.count[2];
  const int nwork = nsl * a.nruns;
  for (int e = blockIdx.x; e < nwork; e += gridDim.x) {
    const int tile = e / (kTB * a.nruns), r = e - tile * kTB * a.nruns;
    const int nb = min(kTB, nsl - tile * kTB);           // blocks of this tile (every tile but the last is full)
    const int run = r / nb, slot = tile * kTB + (r - run * nb);
    const int2 rn = a.runs[run];
    ds_fft_block<VM, TWO, NCH, true>(a, a.rws.slist[slot], make_int4(rn.x, rn.y, 0, 0), lds, xst, threadIdx.x);
    __syncthreads();
  }
}

// ---------------------------------------------------------------- ping-pong variant
// The same work with TWO half-workgroups of 512 threads in one 1024-thread workgroup (16 waves,
// four per SIMD, <= 128 registers, no scratch).  The halves share the block spectra staged in
// LDS and take turns on the single 64 KB transform image.  In round r half h is in phase
// q = r - h:
//   even q  MAC role: the products of pair q / 2 (basis spectra from L2 in half-channel batches,
//           three in flight; block spectra from LDS) between the round's two barriers;
//   odd q   FRONT role: the whole transform of that pair -- first pass in registers, the
//           image-bound passes between the two barriers, the cross-wave reads, then (in the gap
//           to the next round's first barrier, where the other half runs ITS first pass) the
//           register-only last pass, the accumulation (Y - S mu)^2 and, when the pair closes its
//           detector, the statistic and its stores.
// Every round holds exactly two workgroup barriers on both sides (the transform's own two), which
// is all the synchronisation the shared image needs: a round's leading barrier separates one
// half's last reads of the image from the other half's first writes.  A workgroup runs every unit
// of its block (or a group of units, for the blocks of the last partial wave of workgroups); the
// detectors are dealt to the halves whole, so each half keeps its own DS accumulators and stores
// its own rows: no exchange between them.
// Register discipline (the compiler spills 100+ registers otherwise): opaque zeros keep loop
// invariants (LDS slots, row addresses) inside the round, scheduling barriers pin the load
// batches, each role is a straight-line stretch of a two-phase loop body.
// What bounds it is vector-ALU issue and the latency of the few waves, not memory: DESIGN.md 4.1
// has the ladder (5.6 ms for the round-1 kernel -> 2.9 ms) and the ablations behind each step.
template <int VM, bool TWO, int NCH>
__global__ __launch_bounds__(2 * kT) void ds_fft_pp_kernel(const FftArgs a) {
  static_assert(NCH > 0, "the ping-pong kernel keeps the block spectra in LDS");
  __shared__ c32 lds[kLdsC + NCH * kXh];
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 9), tid = threadIdx.x & (kT - 1);   // (uniform per wave)
  // Work order: the first nfull workgroups each take one whole block (all units: the block spectra
  // are staged once and the round pipeline fills and drains once); the rest take (group of units,
  // block), group-major.  Workgroups that run at the same time walk through the same basis
  // spectra, which therefore stay in the XCDs' L2s whatever blocks they are paired with.
  int cb, u0, u1;
  if ((int)blockIdx.x < a.nfull) {
    cb = blockIdx.x; u0 = 0; u1 = a.nunits;
  } else {
    const int j = (int)blockIdx.x - a.nfull, grp = j / a.nrem;
    cb = a.nfull + j - grp * a.nrem; u0 = a.gb[grp]; u1 = a.gb[grp + 1];
  }
  const int chunk = cb / a.nblk, blk = cb - chunk * a.nblk;   // cb = chunk * nblk + blk
  // ragged batch: a block wholly past its chunk's own windows has nothing to do (uniform: before any barrier)
  if (a.pb != nullptr && (int64_t)blk * a.V >= (int64_t)a.pb[chunk]) return;
  // this half's pairs: its share of every unit of the group -- whole detectors, contiguous in memory
  const int2 g0 = a.upre[u0], g1 = a.upre[u1];
  const int pbase = half ? a.upre[a.nunits].x + g0.y : g0.x;
  const int nmine = half ? g1.y - g0.y : g1.x - g0.x;
  // half h runs phases q = 0 .. 2 n_h - 1 (products of pair i in phase 2 i, its transform in 2 i + 1), one
  // round later for h = 1
  const int rounds = max(2 * (g1.x - g0.x), 2 * (g1.y - g0.y) + 1);
  const int64_t p0 = (int64_t)blk * a.V;
  const LdsBases lb = lds_bases(tid);
  float acc0[VM], acc1[TWO ? VM : 1];
#pragma unroll
  for (int m = 0; m < VM; ++m) {
    acc0[m] = 0.f;
    if (TWO) acc1[m] = 0.f;
  }
  const float* mean = a.mean + (int64_t)chunk * a.Ppad + p0;
  const float* den = a.den + (int64_t)chunk * a.Ppad + p0;
  const int wv = tid >> 6, ln = tid & 63;
  const int xlo = 512 * wv + ln, xhi = wv ? 512 * (8 - wv) + 1023 - ln : 1024 - ln;
  auto fix_x = [&](c32 xx, int r) {
    if (r == 0 && tid == 0) xx.y = 0.f;
    if (r == 8 && tid == 0) return make_float2(xx.y, 0.f);
    if (r >= 8) xx.y = -xx.y;
    return xx;
  };
  const int wlo = 512 * wv + ln;
  c32* xs = lds + kLdsC;
  const c32* xh = a.xh + ((int64_t)chunk * a.nblkx + blk) * a.Nc * kXh;
  {
    const float4* src = reinterpret_cast<const float4*>(xh);
    float4* dst = reinterpret_cast<float4*>(xs);
#pragma unroll
    for (int q = 0; q < NCH * kXh / 2 / (2 * kT); ++q)
      dst[threadIdx.x + 2 * kT * q] = src[threadIdx.x + 2 * kT * q];
    __syncthreads();
  }
  const int p1_mode = 0;   // where the transform's register-only first pass runs (measured equal)
  c32 z[16];
  // One lambda per role; each half runs them alternately (half 0 starts with the MAC role, half 1
  // one round later), written out as a two-phase loop body so that z flows from one role to the
  // next in place (one loop with a branch on the phase's parity cost 64 register moves per pair
  // where the branches met).
  // the block's window means: the same ten for every pair -- resident while the register budget allows
  constexpr bool kMuRes = VM <= 10 && !TWO;
  // the window norms a closing pair divides by: loaded before the barrier (latency hidden) where the
  // registers allow, else at their use (the kernel must stay within 128 registers, no scratch)
  constexpr bool kDnEarly = NCH == 1 ? (TWO ? VM <= 12 : true) : (TWO ? VM <= 5 : VM <= 11);
  constexpr bool kMuEarly = kDnEarly;       // likewise the window means, where they are not resident
  float mures[kMuRes ? VM : 1];
  if constexpr (kMuRes) {
#pragma unroll
    for (int m = 0; m < VM; ++m) mures[m] = (mean + 512 * m)[tid];
  }
  // a closing pair's statistic (division by the window norms, DS stores) is formed by the NEXT role of this half,
  // behind the round's first barrier and under the first pair-spectrum loads of that role's products: between a
  // round's second barrier and the next round's first this half has the longer instruction stream (transform
  // tail + accumulation + statistic against the other half's first pass)
  // (not where two accumulator sets leave no room for the window norms beside the first loads: no scratch)
  constexpr bool kDefer = !(TWO && VM >= 13);
  [[maybe_unused]] int2 pend = make_int2(-1, -1);          // detectors whose accumulators wait to be closed (uniform)
  // Roles.  FRONT: the whole transform of the pair whose products this half formed in its previous
  // round -- first pass, the image-bound passes between the round's two barriers, and, after the
  // cross-wave reads, the register-only tail, the accumulation and (closing pair) the statistic and
  // its stores; those run while the OTHER half does the first pass of its next transform, between
  // this round's second barrier and the next round's first.  MAC: nothing but the products of the
  // next pair between the two barriers.  (With the tail and the epilogue at the head of the MAC round
  // the interval between the barriers held 2.2 ms-equivalents of MAC-role work against 1.4 of
  // transform passes: timing-only ablations, profiles/g9_ablations.txt.)
  // accumulation of a finished transform (z) and, when the pair closes its detector, the statistic
  // and its stores
  // (the loads it waits for are issued by fetch_pair, ahead of whatever hides their latency)
  auto fetch_pair = [&](int pr, int te, float4& pi, int2& pd, float (&mu)[kMuRes ? 1 : VM], float (&dn)[VM]) {
        pi = a.pinfo[pr];
        pd = a.pdet[pr];
        if constexpr (!kMuRes && kMuEarly) {
#pragma unroll
          for (int m = 0; m < VM; ++m) mu[m] = (mean + 512 * m)[te];
        }
#pragma unroll
        for (int m = 0; m < VM; ++m) dn[m] = 0.f;
        if (kDnEarly && !kDefer && pd.x >= 0) {
#pragma unroll
          for (int m = 0; m < VM; ++m) dn[m] = (den + 512 * m)[te];
        }
  };
  auto close_pair = [&](int te, const int2 pd, float (&dn)[VM]) {
          const bool two = TWO && pd.y >= 0;
          float* row0 = a.ds + ((int64_t)chunk * a.D + pd.x) * a.ldp + p0;         // uniform
          float* row1 = a.ds + ((int64_t)chunk * a.D + (two ? pd.y : pd.x)) * a.ldp + p0;
          const int lim = (int)min((int64_t)kN, a.P - p0);                    // windows left in the chunk
          if constexpr (!kDnEarly && !kDefer) {
#pragma unroll
            for (int m = 0; m < VM; ++m) dn[m] = (den + 512 * m)[te];
          }
#pragma unroll
          for (int m = 0; m < VM; ++m) {
            // DS = acc / den by reciprocal (v_rcp_f32, 1 ulp) and one Newton step on the quotient:
            // within 1 ulp of the correctly rounded quotient in 5 instructions instead of 12
            const float rc = __builtin_amdgcn_rcpf(dn[m]);
            const bool live = dn[m] > 0.f && !__builtin_isinf(rc);
            float v0 = acc0[m] * rc;
            v0 = __fmaf_rn(__fmaf_rn(-v0, dn[m], acc0[m]), rc, v0);
            v0 = (live && !__builtin_isinf(v0)) ? v0 : 0.f;
            if (te + 512 * m < lim) DTX_DS_STORE(v0, &(row0 + 512 * m)[te]);
            acc0[m] = 0.f;
            if (TWO) {
              float v1 = acc1[m] * rc;
              v1 = __fmaf_rn(__fmaf_rn(-v1, dn[m], acc1[m]), rc, v1);
              v1 = (live && !__builtin_isinf(v1)) ? v1 : 0.f;
              if (two && te + 512 * m < lim) DTX_DS_STORE(v1, &(row1 + 512 * m)[te]);
              acc1[m] = 0.f;
            }
          }
  };
  auto finish_pair = [&](int te, const float4 pi, const int2 pd, float (&mu)[kMuRes ? 1 : VM], float (&dn)[VM]) {
        const bool two = TWO && pd.y >= 0;
        const bool closes = pd.x >= 0;
#pragma unroll
        for (int m = 0; m < VM; ++m) {
          /* explicit fma chain: every instantiation of the kernels rounds the same way */
          if constexpr (!kMuRes && !kMuEarly) mu[m] = (mean + 512 * m)[te];
          const float mum = kMuRes ? mures[m] : mu[kMuRes ? 0 : m];
          const float re = __fmaf_rn(-pi.x, mum, z[m].x), im = __fmaf_rn(-pi.y, mum, z[m].y);
          const float i0 = two ? 0.f : im, i1 = two ? im : 0.f;
          acc0[m] = __fmaf_rn(i0, i0, __fmaf_rn(re, re, acc0[m]));
          if (TWO) acc1[m] = __fmaf_rn(i1, i1, acc1[m]);
        }
        if (closes) {
          if constexpr (kDefer) pend = pd;
          else close_pair(te, pd, dn);
        }
  };
  auto front_role = [&](int q) {
      const int pi_ = q >> 1;
      if (q >= 0 && pi_ < nmine) {
        int oe;                                  // nothing below is hoisted out of the round loop
        asm volatile("v_mov_b32 %0, 0" : "=v"(oe));
        const int te = tid + oe;
        const int pr = pbase + pi_;              // global pair index
        // issued now, landing under the transform: the tail's twiddles, the pair's constants, the
        // window means and norms
        c32 p4[3];
        float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
        int2 pd = make_int2(-1, -1);
        float mu[kMuRes ? 1 : VM], dn[VM];
        fft8192_back_twiddles<1>(p4, a.tab, te);
        fetch_pair(pr, te, pi, pd, mu, dn);
        __builtin_amdgcn_sched_barrier(0);
        fft8192<1, true, true>(z, lds, lb, a.tab, tid, p1_mode);    // two workgroup barriers inside
        fft8192_back_compute<1>(z, p4);
        finish_pair(te, pi, pd, mu, dn);
      } else {
        __syncthreads();
        __syncthreads();
      }
  };
  auto mac_role = [&](int q) {
      const bool mac = q >= 0 && (q >> 1) < nmine;       // a pair's products to form
      __syncthreads();       // the image is the other half's from here (our cross-wave reads are done)
      [[maybe_unused]] const int2 pnd = pend;
      [[maybe_unused]] float dnp[kDefer ? VM : 1];
      [[maybe_unused]] int ot = 0;
      if constexpr (kDefer) {
        pend = make_int2(-1, -1);
        asm volatile("v_mov_b32 %0, 0" : "=v"(ot));
        if (pnd.x >= 0) {
#pragma unroll
          for (int m = 0; m < VM; ++m) dnp[m] = (den + 512 * m)[tid + ot];
        }
      }
      // ---- MAC of pair q / 2: half-channel batches of four 16-byte loads, three batches in flight
      if (mac) {
        const int pr = pbase + (q >> 1);
        int oz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(oz));
        const int xlo_ = xlo + oz, xhi_ = xhi + oz, wlo_ = wlo + oz;
        auto x_slot = [&](int r) { return r < 8 ? xlo_ + 64 * r : ((r == 8 && tid == 0) ? 0 : xhi_ - 64 * r); };
        const float4* wq = reinterpret_cast<const float4*>(a.W + (int64_t)pr * NCH * kN) + wlo_;   // + lane
        // batches in flight: three, two where the accumulators leave no room (no scratch)
        constexpr int kBatches = 2 * NCH, kDepth = (NCH == 1 || !kDnEarly) ? 2 : 3;
        float4 pw[kDepth][4];
        auto issue = [&](auto bc) {
          constexpr int b = bc, c = b >> 1, h2 = b & 1;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pw[b % kDepth][j] = (wq + c * (kN / 2) + 64 * (4 * h2 + j))[0];
        };
        dtx::static_for<0, kDepth>([&](auto bc) { issue(bc); });
        if constexpr (kDefer) {
          __builtin_amdgcn_sched_barrier(0);
          if (pnd.x >= 0) close_pair(tid + ot, pnd, dnp);     // under the loads just issued
        }
        __builtin_amdgcn_sched_barrier(0);
        dtx::static_for<0, kBatches>([&](auto bc) {
          constexpr int b = bc, c = b >> 1, h2 = b & 1;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            c32 px[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) px[e] = fix_x(xs[c * kXh + x_slot(8 * h2 + 4 * hh + e)], 8 * h2 + 4 * hh + e);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int m = 8 * h2 + 4 * hh + e;
              const float4 t = pw[b % kDepth][2 * hh + (e >> 1)];
              const c32 w = (e & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);
              if constexpr (c == 0) {             // first channel: z starts here (no zero fill)
                z[m].x = __fmaf_rn(-w.y, px[e].y, w.x * px[e].x);
                z[m].y = __fmaf_rn(w.y, px[e].x, w.x * px[e].y);
              } else {                            // two fused multiply-adds per component
                z[m].x = __fmaf_rn(-w.y, px[e].y, __fmaf_rn(w.x, px[e].x, z[m].x));
                z[m].y = __fmaf_rn(w.y, px[e].x, __fmaf_rn(w.x, px[e].y, z[m].y));
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (b + kDepth < kBatches) {
            issue(std::integral_constant<int, b + kDepth>{});
            __builtin_amdgcn_sched_barrier(0);
          }
        });
        if (p1_mode == 1) fft8192_pass1<1>(z);     // register-only first pass of the transform
      }
      else if constexpr (kDefer) {
        if (pnd.x >= 0) close_pair(tid + ot, pnd, dnp);
      }
      __syncthreads();                           // the second barrier of the round
  };
  const int rounds2 = (rounds + 1) >> 1;       // (an odd count is padded with a round of bare barriers)
  if (half == 0) {
#pragma nounroll
    for (int i = 0; i < rounds2; ++i) {
      mac_role(2 * i);
      front_role(2 * i + 1);
    }
  } else {
#pragma nounroll
    for (int i = 0; i < rounds2; ++i) {
      front_role(2 * i - 1);
      mac_role(2 * i);
    }
  }
  if constexpr (kDefer) {
    if (pend.x >= 0) {        // half 0 ends on a transform: its last statistic
      float dnp[VM];
#pragma unroll
      for (int m = 0; m < VM; ++m) dnp[m] = (den + 512 * m)[tid];
      close_pair(tid, pend, dnp);
    }
  }
}

template <int VM>
void launch_fft(dtx_ctx* ctx, const FftArgs& a, unsigned grid, bool two) {
  const bool xl = a.nseg == 1;   // block spectra LDS-resident: single-segment templates
  if (ctx->fft.pingpong) {
    if (two && a.Nc == 3) hipLaunchKernelGGL((ds_fft_pp_kernel<VM, true, 3>), dim3(grid), dim3(2 * kT), 0, ctx->stream, a);
    else if (two) hipLaunchKernelGGL((ds_fft_pp_kernel<VM, true, 1>), dim3(grid), dim3(2 * kT), 0, ctx->stream, a);
    else if (a.Nc == 3) hipLaunchKernelGGL((ds_fft_pp_kernel<VM, false, 3>), dim3(grid), dim3(2 * kT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((ds_fft_pp_kernel<VM, false, 1>), dim3(grid), dim3(2 * kT), 0, ctx->stream, a);
    return;
  }
  if (two && a.Nc == 3 && xl) {   // some pair holds two single-vector detectors
    hipLaunchKernelGGL((ds_fft_kernel<VM, true, 3>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  } else if (two) {
    hipLaunchKernelGGL((ds_fft_kernel<VM, true, 0>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  } else if (a.Nc == 3 && xl) {
    hipLaunchKernelGGL((ds_fft_kernel<VM, false, 3>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  } else if (a.Nc == 1 && xl) {
    hipLaunchKernelGGL((ds_fft_kernel<VM, false, 1>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  } else {
    hipLaunchKernelGGL((ds_fft_kernel<VM, false, 0>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  }
}

// the re-scan: the one-role kernel whatever the main scan ran (block spectra LDS-resident for three channels,
// streamed otherwise)
template <int VM>
void launch_rescan(dtx_ctx* ctx, const FftArgs& a, unsigned grid, bool two) {
  if (two) hipLaunchKernelGGL((ds_fft_rescan_kernel<VM, true, 0>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  else if (a.Nc == 3) hipLaunchKernelGGL((ds_fft_rescan_kernel<VM, false, 3>), dim3(grid), dim3(kT), 0, ctx->stream, a);
  else hipLaunchKernelGGL((ds_fft_rescan_kernel<VM, false, 0>), dim3(grid), dim3(kT), 0, ctx->stream, a);
}

// Pair spectra in FLOAT64 on the device: W[task][wave-major f] = 1/kN sum_j (u_ra,c[j0 + j] +
// i u_rb,c[j0 + j]) e^{+2 pi i j f / kN}, task = pair * nvc + (c * nseg + seg).  One workgroup
// per task; a plain radix-2 transform of 8192 complex128 points in 128 KB of LDS (twiddles from
// sincospi in float64), rounded to float32 only on the way out -- the same values the host
// used to build with a float64 FFT (300 transforms at configs[1]: 80 ms on one core, ~0.1 ms here).
__global__ __launch_bounds__(512) void wspec64_kernel(const float* __restrict__ U, int n, int Nc,
                                                      int L, int nseg, int seglen,
                                                      const int2* __restrict__ prow,
                                                      c32* __restrict__ W) {
  __shared__ double2 z[kN];
  const int tid = threadIdx.x;
  const int nvc = Nc * nseg;
  const int pr = blockIdx.x / nvc, vc = blockIdx.x - pr * nvc;
  const int c = vc / nseg, sg = vc - c * nseg;
  const int j0 = sg * seglen, j1 = min(L, j0 + seglen);
  const int2 rows = prow[pr];
  constexpr int kBits = 13;
  static_assert((1 << kBits) == kN, "bit reversal is for 8192 points");
  for (int j = tid; j < kN; j += 512) {
    double re = 0.0, im = 0.0;
    if (j0 + j < j1) {
      const size_t q = (size_t)(j0 + j) * Nc + c;
      re = (double)U[(size_t)rows.x * n + q];
      if (rows.y >= 0) im = (double)U[(size_t)rows.y * n + q];
    }
    z[__brev((unsigned)j) >> (32 - kBits)] = make_double2(re, im);
  }
  __syncthreads();
  for (int half = 1; half < kN; half <<= 1) {
    for (int b = tid; b < kN / 2; b += 512) {
      const int k = b & (half - 1);
      const int i = ((b - k) << 1) + k;
      double sn, cs;
      sincospi((double)k / (double)half, &sn, &cs);        // e^{+i pi k / half}
      const double2 u = z[i], w = z[i + half];
      const double2 v = make_double2(w.x * cs - w.y * sn, w.x * sn + w.y * cs);
      z[i] = make_double2(u.x + v.x, u.y + v.y);
      z[i + half] = make_double2(u.x - v.x, u.y - v.y);
    }
    __syncthreads();
  }
  c32* dst = W + (size_t)blockIdx.x * kN;
  // stored in the order the scan reads (w_slot): wave w's frequencies f = 8 n2 + w, register pairs adjacent
  for (int q = tid; q < kN; q += 512)
    dst[w_slot(q)] = make_float2((float)(z[q].x / kN), (float)(z[q].y / kN));
}

// pinfo[pair] = {S_ra, S_rb, 0, 0} from the row sums
__global__ void pinfo_kernel(const int2* __restrict__ prow, const float* __restrict__ srow, int npairs,
                             float4* __restrict__ pinfo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npairs) return;
  const int2 r = prow[p];
  pinfo[p] = make_float4(srow[r.x], r.y >= 0 ? srow[r.y] : 0.f, 0.f, 0.f);
}

}  // namespace

void dtx_fft_free(dtx_ctx* ctx) {
  DtxFft& f = ctx->fft;
  if (f.W_dev) (void)hipFree(f.W_dev);
  if (f.tab_dev) (void)hipFree(f.tab_dev);
  if (f.unit_dev) (void)hipFree(f.unit_dev);
  if (f.pinfo_dev) (void)hipFree(f.pinfo_dev);
  if (f.pdet_dev) (void)hipFree(f.pdet_dev);
  if (f.prow_dev) (void)hipFree(f.prow_dev);
  if (f.upre_dev) (void)hipFree(f.upre_dev);
  if (f.runs_dev) (void)hipFree(f.runs_dev);
  f = DtxFft();
}

bool dtx_fft_eligible(int L) { return L >= kFftMinL && L <= kFftLongMaxL; }

// Units / pairs of the uploaded basis (host, from det_off alone) and their spectra (device, from
// the context's device copy of U and its row sums)
int dtx_fft_basis_build(dtx_ctx* ctx, const int32_t* det_off, int D, int n, int Nc) {
  dtx_fft_free(ctx);
  const int L = n / Nc;
  if (!dtx_fft_eligible(L)) return DTX_OK;
  DtxFft& f = ctx->fft;
  // Templates up to kFftMaxL: one segment, V = 512 floor((kN - L + 1) / 512) windows per block.
  // Longer ones are cut into segments of kSeg = kN / 2 samples and the block advance is kSeg
  // too: Y[p] = sum_s corr(segment s, data at p + s kSeg), and with advance == segment length
  // "data at p + s kSeg" is simply block (blk + s) -- its spectrum is already there.  Each
  // (channel, segment) is a virtual channel of the MAC.
  // The pairs, the units (runs of whole detectors' pairs, closed at DTX_FFT_UNIT = 8: the block spectra a
  // workgroup stages in LDS serve all its pairs) and the split of every unit between the two halves of the
  // ping-pong kernel are host arithmetic on det_off alone: dtxplan::plan_fft_basis (dtx_plan.h, the
  // translation unit the CPU sanitizer build covers).
  constexpr int kSeg = kN / 2;
  dtxplan::FftBasisPlan pl = dtxplan::plan_fft_basis(det_off, D, L, kN, kFftMaxL, 8);
  f.nseg = pl.nseg;
  f.V = pl.V;
  f.two = pl.two;
  f.nunits = pl.nunits;
  f.npairs = pl.npairs;
  // the ping-pong kernel keeps the block spectra in LDS (single-segment templates, one or three
  // channels) and is worth it when the halves are balanced: at least 0.8 pairs per round
  f.pingpong = f.nseg == 1 && (Nc == 1 || Nc == 3) && pl.balanced;
  if (f.pingpong) {
    // store the pairs half 0's first: each half-workgroup then walks a contiguous run of pairs
    dtxplan::reorder_pairs_for_halves(pl);
    f.unit_pairs = pl.unit_pairs;
  }
  static_assert(sizeof(int4) == sizeof(dtxplan::I4) && sizeof(int2) == sizeof(dtxplan::I2), "plan tables are int2 / int4");
  const std::vector<dtxplan::I2> runs = dtxplan::plan_fft_runs(pl, f.pingpong);
  f.nruns = (int)runs.size();
  const std::vector<dtxplan::I4>& units = pl.units;
  const std::vector<dtxplan::I2>&pdet = pl.pdet, &prow = pl.prow, &upre = pl.upre;

  const int nvc = Nc * f.nseg;
  static const std::vector<float2> tab = [] {          // e^{+2 pi i m / kN}, once per process
    std::vector<float2> t(kN);
    const double pi = 3.14159265358979323846264338327950288;
    for (int m = 0; m < kN; ++m)
      t[m] = make_float2((float)std::cos(2 * pi * m / kN), (float)std::sin(2 * pi * m / kN));
    return t;
  }();
  DTX_HIP(ctx, hipMalloc(&f.W_dev, (size_t)f.npairs * nvc * kN * sizeof(float2)));
  DTX_HIP(ctx, hipMalloc(&f.tab_dev, tab.size() * sizeof(float2)));
  DTX_HIP(ctx, hipMalloc(&f.unit_dev, units.size() * sizeof(int4)));
  DTX_HIP(ctx, hipMalloc(&f.pinfo_dev, (size_t)f.npairs * sizeof(float4)));
  DTX_HIP(ctx, hipMalloc(&f.pdet_dev, pdet.size() * sizeof(int2)));
  DTX_HIP(ctx, hipMalloc(&f.prow_dev, prow.size() * sizeof(int2)));
  DTX_HIP(ctx, hipMalloc(&f.upre_dev, upre.size() * sizeof(int2)));
  DTX_HIP(ctx, hipMalloc(&f.runs_dev, runs.size() * sizeof(int2)));
  // small synchronous copies into fresh buffers
  hipError_t e = hipMemcpy(f.prow_dev, prow.data(), prow.size() * sizeof(int2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f.pdet_dev, pdet.data(), pdet.size() * sizeof(int2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f.tab_dev, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f.unit_dev, units.data(), units.size() * sizeof(int4), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f.upre_dev, upre.data(), upre.size() * sizeof(int2), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(f.runs_dev, runs.data(), runs.size() * sizeof(int2), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pinfo_kernel, dim3((unsigned)((f.npairs + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const int2*)f.prow_dev, (const float*)ctx->srow_all.p, f.npairs,
                       (float4*)f.pinfo_dev);
    hipLaunchKernelGGL(wspec64_kernel, dim3((unsigned)(f.npairs * nvc)), dim3(512), 0, ctx->stream,
                       (const float*)ctx->U_buf.p, n, Nc, L, f.nseg, f.nseg == 1 ? L : kSeg,
                       (const int2*)f.prow_dev, (c32*)f.W_dev);
    e = hipGetLastError();
  }
  if (e != hipSuccess)
    return dtx_fail(ctx, DTX_E_HIP, "FFT engine: building the pair spectra failed: %s", hipGetErrorString(e));
  f.ok = true;
  return DTX_OK;
}

// xp: the channel planes; mean / den: the window statistics, being computed on the context's side
// stream (forked by the caller): the guard pass follows them there, the block spectra are taken on the
// main stream meanwhile, and the two streams meet in front of the scan kernel
int dtx_fft_scan(dtx_ctx* ctx, const float* xp, int64_t Tpad, const float* mean, const float* den,
                 int64_t Ppad, int64_t P, int nchunk, float* ds, int64_t ldp, const DtxRepairWs& ws,
                 const int* pb) {
  DtxFft& f = ctx->fft;
  if (!f.ok) return dtx_fail(ctx, DTX_E_STATE, "FFT engine: no spectra for this basis");
  const int Nc = ctx->Nc, V = f.V;
  const int nblk = (int)((P + V - 1) / V);
  const bool pp = f.pingpong;
  const int ntile_u = (f.nunits + kTileU - 1) / kTileU;
  const int ntile_b = (int)(((int64_t)nchunk * nblk + kTileB - 1) / kTileB);
  const int64_t per_xcd = ((int64_t)ntile_u * ntile_b * kTile + 7) / 8;
  int64_t grid = per_xcd * 8;
  // ping-pong kernel: one workgroup per block runs every unit (the block spectra are staged once,
  // the pipeline fills and drains once) for as many blocks as fill the CUs a whole number of
  // times; the remaining nrem < ncu blocks are split over ng groups of units each, ng chosen so
  // that nrem * ng short workgroups again fill the CUs evenly.
  int nfull = 0, nrem = 0, ng = 1;
  short gb[kMaxGroups + 1] = {0};
  if (pp) {
    static_assert(kMaxGroups == dtxplan::kPlanMaxGroups, "group table size");
    const char* e = getenv("DTX_PP_GROUPS");     // experiments: this many groups for every block
    const dtxplan::PpGridPlan gp = dtxplan::plan_pp_grid((int64_t)nchunk * nblk, ctx->ncu, f.nunits, f.npairs,
                                                         f.unit_pairs.data(), e ? atoi(e) : 0);
    nfull = gp.nfull; nrem = gp.nrem; ng = gp.ng;
    std::copy(gp.gb, gp.gb + kMaxGroups + 1, gb);
    grid = gp.grid;
  }
  if (grid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "FFT engine: grid too large");
  int rc;
  const int nblkx = nblk + f.nseg - 1;   // a segmented template also reads the blocks behind its own
  // (single-segment templates: a second set for the blocks scanned again without their outliers -- 172 MB at
  // configs[1], written only where the guard lists something)
  const size_t xh_bytes = (size_t)nchunk * nblkx * Nc * kXh * sizeof(float2);
  if ((rc = dtx_reserve(ctx, ctx->ws_xhat, (f.nseg == 1 ? 2 : 1) * xh_bytes))) return rc;
  c32* xh = (c32*)ctx->ws_xhat.p;
  c32* xhc = f.nseg == 1 ? (c32*)((char*)ctx->ws_xhat.p + xh_bytes) : xh;
  const int keep = f.nseg == 1 ? std::min(kN, V + ctx->n / Nc - 1) : kN;
  hipLaunchKernelGGL(xhat_kernel, dim3((unsigned)nblkx, (unsigned)Nc, (unsigned)nchunk), dim3(kT), 0,
                     ctx->stream, xp, Tpad, Nc, V, nblkx, keep, (const c32*)f.tab_dev, xh, 1, 0);
  DTX_HIP(ctx, hipGetLastError());
  FftArgs a;
  a.xh = xh; a.xhc = xhc; a.W = (const c32*)f.W_dev; a.tab = (const c32*)f.tab_dev;
  a.unit = (const int4*)f.unit_dev; a.pinfo = (const float4*)f.pinfo_dev;
  a.pdet = (const int2*)f.pdet_dev;
  a.mean = mean; a.den = den; a.ds = ds; a.Ppad = Ppad; a.ldp = ldp; a.P = P;
  a.Nc = Nc; a.D = ctx->D; a.V = V; a.nblk = nblk; a.nchunk = nchunk;
  a.nunits = f.nunits; a.ntile_b = ntile_b; a.per_xcd = (int)per_xcd;
  a.nseg = f.nseg; a.nblkx = nblkx;
  a.xp = xp; a.Tpad = Tpad; a.keep = keep;
  a.pb = pp ? pb : nullptr;      // (the one-group kernel of long templates runs every block)
  a.upre = (const int2*)f.upre_dev; a.nfull = nfull; a.nrem = std::max(nrem, 1); a.ngroups = ng;
  std::copy(gb, gb + kMaxGroups + 1, a.gb);
  a.rws = ws; a.runs = (const int2*)f.runs_dev; a.nruns = f.nruns; a.prow = (const int2*)f.prow_dev;
  a.U = (const float*)ctx->U_buf.p; a.n = ctx->n; a.L = ctx->L;
  hipLaunchKernelGGL(guard_kernel, dim3((unsigned)(nchunk * nblk)), dim3(256), 0, ctx->aux, den,
                     Ppad, P, V, nblk, f.nseg == 1 ? V : 2 * V, ws, pb);
  DTX_HIP(ctx, hipGetLastError());
  // What becomes of the listed blocks (outliers split off, repair tasks; the spectra of the blocks with outliers split
  // off, without them): launches sized for the chip that find empty lists on ordinary data.  Still on the side stream,
  // beside the block spectra -- the chain statistics -> guard -> these is shorter than `xhat` there -- unless the call
  // is small (fewer blocks than 2 x CUs: configs[4] one hour per call): then `xhat` is the short side and they follow
  // it on the main stream.
  const bool small_call = (int64_t)nchunk * nblk < 2 * (int64_t)ctx->ncu;
  if (small_call) {
    const int rcj = dtx_aux_join(ctx);
    if (rcj) return rcj;
  }
  const hipStream_t sst = small_call ? ctx->stream : ctx->aux;
  hipLaunchKernelGGL(spike_split_kernel, dim3((unsigned)ctx->ncu), dim3(kSplitT), 0, sst, xp, Tpad, Nc, ctx->L,
                     den, Ppad, P, V, nblk, f.nseg, ws, pb);
  DTX_HIP(ctx, hipGetLastError());
  if (f.nseg == 1) {
    hipLaunchKernelGGL(xhat_clean_kernel, dim3((unsigned)ctx->ncu, (unsigned)Nc), dim3(kT), 0, sst, xp, Tpad,
                       Nc, V, nblk, nblkx, keep, (const c32*)f.tab_dev, xhc, ws);
    DTX_HIP(ctx, hipGetLastError());
  }
  if (!small_call) {
    const int rcj = dtx_aux_join(ctx);
    if (rcj) return rcj;
  }
  {
    char nm[96];
    if (pp) snprintf(nm, sizeof nm, "ds_fft_pp_kernel<%d,%s,%d>", V / 512, f.two ? "true" : "false", Nc);
    else snprintf(nm, sizeof nm, "ds_fft_kernel<%d,%s,%d>", V / 512, f.two ? "true" : "false",
                  (f.nseg == 1 && !f.two && (Nc == 1 || Nc == 3)) || (f.nseg == 1 && f.two && Nc == 3) ? Nc : 0);
    ctx->last_kernel = nm;
  }
  { const int rcw = dtx_wait_post(ctx); if (rcw) return rcw; }     // (an earlier scan's post pass may still be reading this DS tensor)
  { const int rck = dtx_kev_next(ctx); if (rck) return rck; }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));   // times the dominant kernel only
  switch (V / 512) {
#ifndef DTX_FFT_ONLY_VM
    case 1: launch_fft<1>(ctx, a, (unsigned)grid, f.two); break;
    case 2: launch_fft<2>(ctx, a, (unsigned)grid, f.two); break;
    case 3: launch_fft<3>(ctx, a, (unsigned)grid, f.two); break;
    case 4: launch_fft<4>(ctx, a, (unsigned)grid, f.two); break;
    case 5: launch_fft<5>(ctx, a, (unsigned)grid, f.two); break;
    case 6: launch_fft<6>(ctx, a, (unsigned)grid, f.two); break;
    case 7: launch_fft<7>(ctx, a, (unsigned)grid, f.two); break;
    case 8: launch_fft<8>(ctx, a, (unsigned)grid, f.two); break;
    case 9: launch_fft<9>(ctx, a, (unsigned)grid, f.two); break;
#endif
    case 10: launch_fft<10>(ctx, a, (unsigned)grid, f.two); break;
#ifndef DTX_FFT_ONLY_VM
    case 11: launch_fft<11>(ctx, a, (unsigned)grid, f.two); break;
    case 12: launch_fft<12>(ctx, a, (unsigned)grid, f.two); break;
    case 13: launch_fft<13>(ctx, a, (unsigned)grid, f.two); break;
    case 14: launch_fft<14>(ctx, a, (unsigned)grid, f.two); break;
    case 15: launch_fft<15>(ctx, a, (unsigned)grid, f.two); break;
#endif
    default: return dtx_fail(ctx, DTX_E_STATE, "FFT engine: bad block advance %d", V);
  }
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  dtx_kev_commit(ctx);
  if (f.nseg == 1) {
    // blocks with outliers split off: again on the one-role kernel, on their clean spectra (a launch sized for the
    // chip that finds an empty list on ordinary data)
    const unsigned rgrid = (unsigned)(4 * ctx->ncu);
    switch (V / 512) {
#ifndef DTX_FFT_ONLY_VM
      case 1: launch_rescan<1>(ctx, a, rgrid, f.two); break;
      case 2: launch_rescan<2>(ctx, a, rgrid, f.two); break;
      case 3: launch_rescan<3>(ctx, a, rgrid, f.two); break;
      case 4: launch_rescan<4>(ctx, a, rgrid, f.two); break;
      case 5: launch_rescan<5>(ctx, a, rgrid, f.two); break;
      case 6: launch_rescan<6>(ctx, a, rgrid, f.two); break;
      case 7: launch_rescan<7>(ctx, a, rgrid, f.two); break;
      case 8: launch_rescan<8>(ctx, a, rgrid, f.two); break;
      case 9: launch_rescan<9>(ctx, a, rgrid, f.two); break;
#endif
      case 10: launch_rescan<10>(ctx, a, rgrid, f.two); break;
#ifndef DTX_FFT_ONLY_VM
      case 11: launch_rescan<11>(ctx, a, rgrid, f.two); break;
      case 12: launch_rescan<12>(ctx, a, rgrid, f.two); break;
      case 13: launch_rescan<13>(ctx, a, rgrid, f.two); break;
      case 14: launch_rescan<14>(ctx, a, rgrid, f.two); break;
      case 15: launch_rescan<15>(ctx, a, rgrid, f.two); break;
#endif
      default: break;
    }
    DTX_HIP(ctx, hipGetLastError());
  }
  return DTX_OK;
}
