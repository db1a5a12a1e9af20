// All-pairs normalised cross-correlation on CDNA4 (gfx950): replaces the
// Python pair loop of the reference (detex/construct.py:369-394 _makeDFcclags
// calling _CCX2, 425-466, once per pair with one FFT each) by the same
// implicit-Hankel FP32 MFMA contraction as the detection statistic.
//
// For a fixed "later" event c, zero-pad its channel planes by S = L - L//2
// samples on both sides (z_c) -- the central half of the lag range the
// reference keeps after result[trunc:-trunc] (construct.py:439, 453).  Then for
// every earlier event b and every kept lag s = p - S
//     Y[b][p] = sum_ch sum_t x_b,ch[t] * z_c,ch[p + t]        (= x_b . w_s)
// is a GEMM [events x n] * Hankel(z_c): the events are the dense operand
// (stored transposed + channel-planar once), z_c is the stream.  Steps of the
// contraction whose z window is entirely padding are skipped.  The epilogue
// normalises (means / stds over the zero-padded
// windows), scrubs |cc| > 1, and reduces max / first-argmax over the lags with
// one 64-bit atomicMax per (pair, wave).  A finishing kernel re-evaluates the
// winning lag and its two neighbours in float64 and applies the cosine
// sub-sample fit (construct.py:397-422).
#include <algorithm>

#include "dtx_hankel.h"
#include "dtx_hankel_bf16.h"
#include "dtx_internal.h"

using namespace dtx;


namespace {

constexpr float kScrubTol = 1e-5f;   // see DESIGN.md "knife-edge scrub"

__host__ __device__ inline int64_t condensed_index(int64_t N, int64_t b, int64_t c) {
  return N * b - b * (b + 1) / 2 + (c - b - 1);
}

// XT[kt][ch*Lp + t][208] = td[208*kt + k][Nc*t + ch]   (0 where t >= L or event >= N)
__global__ __launch_bounds__(256) void xc_transpose_kernel(const float* __restrict__ td, int N,
                                                           int n, int Nc, int L, int Lp,
                                                           float* __restrict__ xt) {
  const int kt = blockIdx.z;
  const int64_t rows = (int64_t)Nc * Lp;
  for (int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); q < rows;
       q += (int64_t)gridDim.x * 4) {
    const int ch = (int)(q / Lp), t = (int)(q - (int64_t)ch * Lp);
    for (int k = threadIdx.x & 63; k < kKtMax; k += 64) {
      const int e = kt * kKtMax + k;
      float v = 0.f;
      if (e < N && t < L) v = td[(int64_t)e * n + (int64_t)t * Nc + ch];
      xt[((int64_t)kt * rows + q) * kKtMax + k] = v;
    }
  }
}

// split-bf16 image of the event tiles: XT16[kt][ch][Lp/8][h][208][8] (dtx_hankel_bf16.h)
__global__ __launch_bounds__(256) void xc_transpose16_kernel(const float* __restrict__ td, int N,
                                                             int n, int Nc, int L, int Lp,
                                                             uint16_t* __restrict__ xt16) {
  const int kt = blockIdx.z;
  const int64_t nqb = (int64_t)Nc * (Lp / 8);
  for (int64_t qb = blockIdx.x; qb < nqb; qb += gridDim.x) {
    const int ch = (int)(qb / (Lp / 8)), j8 = (int)(qb - (int64_t)ch * (Lp / 8)) * 8;
    for (int v = threadIdx.x; v < kKtMax * 8; v += 256) {
      const int k = v >> 3, j = v & 7;
      const int e = kt * kKtMax + k, t = j8 + j;
      float val = 0.f;
      if (e < N && t < L) val = td[(int64_t)e * n + (int64_t)t * Nc + ch];
      const unsigned short hi = __builtin_bit_cast(unsigned short, (__bf16)val);
      const float hf = __uint_as_float((unsigned)hi << 16);
      const unsigned short lo = __builtin_bit_cast(unsigned short, (__bf16)(val - hf));
      uint16_t* base = xt16 + (((int64_t)kt * nqb + qb) * 2) * kKtMax * 8;
      base[(int64_t)k * 8 + j] = hi;
      base[(int64_t)kKtMax * 8 + (int64_t)k * 8 + j] = lo;
    }
  }
}

// z[e][ch][tau] = td[e][Nc*(tau - S) + ch] for 0 <= tau - S < L, else 0
__global__ __launch_bounds__(256) void xc_pad_kernel(const float* __restrict__ td, int n, int Nc,
                                                     int L, int S, int64_t Tz,
                                                     float* __restrict__ z) {
  const int e = blockIdx.z, ch = blockIdx.y;
  const float* src = td + (int64_t)e * n;
  float* dst = z + ((int64_t)e * Nc + ch) * Tz;
  for (int64_t tau = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tau < Tz;
       tau += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = tau - S;
    dst[tau] = (t >= 0 && t < L) ? src[t * Nc + ch] : 0.f;
  }
}

// per event: sum, sum of squared deviations (float64), 1/sqrt(ssd) (float32)
__global__ __launch_bounds__(256) void xc_event_stats_kernel(const float* __restrict__ td, int n,
                                                             double* __restrict__ esum,
                                                             double* __restrict__ essd,
                                                             double* __restrict__ esq,
                                                             float* __restrict__ fsum,
                                                             float* __restrict__ frs) {
  __shared__ double sh[8];
  const int e = blockIdx.x, tid = threadIdx.x;
  const float* x = td + (int64_t)e * n;
  double s1 = 0.0, s2 = 0.0;
  for (int j = tid; j < n; j += 256) {
    const double v = (double)x[j];
    s1 += v;
    s2 += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  if ((tid & 63) == 0) { sh[tid >> 6] = s1; sh[4 + (tid >> 6)] = s2; }
  __syncthreads();
  if (tid == 0) {
    s1 = sh[0] + sh[1] + sh[2] + sh[3];
    s2 = sh[4] + sh[5] + sh[6] + sh[7];
    const double ssd = clamp_var(s2, s1, (double)n);
    esum[e] = s1;
    essd[e] = ssd;
    esq[e] = s2;
    fsum[e] = (float)s1;
    frs[e] = ssd > 0.0 ? (float)(1.0 / sqrt(ssd)) : 0.f;
  }
}

struct XcArgs {
  const float* xt;      // [nkt][Nc*Lp][208]
  const float* z;       // [N][Nc][Tz]
  const float* fsum;    // [N]   sum(x_b)
  const float* frs;     // [N]   1/sqrt(ssd(x_b)), 0 for a dead event
  const float* mean_w;  // [N][Ppad] mean of the padded window of event c at lag p
  const float* rsw;     // [N][Ppad] 1/sqrt(ssd) of that window, 0 if zero variance
  unsigned long long* keys;  // [npairs] (orderable cc << 32) | ~p
  const int2* items;    // (c, kt) work items of this launch
  const uint16_t* xt16; // split-bf16 engine: event tiles
  const i32x4* g16;     // split-bf16 engine: shifted copies of z, [N][Nc][2][8][V]
  int64_t V;
  int64_t Tz, Ppad, pair_begin, pair_end;
  int N, Nc, L, Lp, S, Pz, nptiles;
};

// normalise, scrub, max / first argmax over lags; shared by both engines
template <int RM, int TN>
__device__ __forceinline__ void xc_epilogue(f32x4 (&acc)[RM][TN], const XcArgs& a, int c, int kt,
                                            int p0) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int col0 = wave * 16 * TN;
  float mu[TN], rw[TN];
  const int pcol = p0 + col0 + i;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int p = pcol + 16 * t;   // < Ppad
    mu[t] = a.mean_w[(int64_t)c * a.Ppad + p];
    rw[t] = a.rsw[(int64_t)c * a.Ppad + p];
  }
#pragma unroll
  for (int r = 0; r < RM; ++r) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int b = kt * kKtMax + 16 * r + 4 * kq + e;
      const bool rowok = b < c;
      const float s1 = rowok ? a.fsum[b] : 0.f;
      const float r1 = rowok ? a.frs[b] : 0.f;
      unsigned long long key = 0ull;
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int p = pcol + 16 * t;
        float cc = (acc[r][t][e] - s1 * mu[t]) * (r1 * rw[t]);
        if (cc > 1.f + kScrubTol || cc < -1.f - kScrubTol) cc = 0.f;   // construct.py:458-462
        if (p < a.Pz && cc == cc) {
          const unsigned bits = __float_as_uint(cc);
          const unsigned ord = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
          const unsigned long long k2 =
              ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)p);
          key = k2 > key ? k2 : key;
        }
      }
      // the 16 lanes sharing kq hold the other columns of this row
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        const unsigned long long o = __shfl_xor(key, m, 64);
        key = o > key ? o : key;
      }
      if (i == 0 && rowok && r1 > 0.f && key != 0ull) {
        const int64_t idx = condensed_index(a.N, b, c);
        if (idx >= a.pair_begin && idx < a.pair_end) atomicMax(a.keys + idx, key);
      }
    }
  }
}

template <int RM, int TN>
__global__ __launch_bounds__(kThreads, 2) void xc_mfma_kernel(const XcArgs a) {
  constexpr int PT = 128 * TN;
  __shared__ __attribute__((aligned(16))) float lds[lds_floats(RM)];

  const int item = blockIdx.x / a.nptiles;
  const int pt = blockIdx.x - item * a.nptiles;
  const int2 it = a.items[item];
  const int c = it.x, kt = it.y;
  const int p0 = pt * PT;

  // contraction range with a non-padding z window: S <= p + t < S + L
  int jlo = a.S - p0 - PT + 1;
  jlo = jlo > 0 ? (jlo / kBQ) * kBQ : 0;
  int jhi = a.S + a.L - p0;
  jhi = min(a.Lp, ((jhi + kBQ - 1) / kBQ) * kBQ);

  f32x4 acc[RM][TN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // RM == 13: the 208-wide event tile is copied whole -> LDS-DMA staging
  hankel_mainloop<RM, TN, RM == kRmMax>(acc, a.xt + (int64_t)kt * a.Nc * a.Lp * kKtMax, kKtMax, a.Lp,
                          a.z + (int64_t)c * a.Nc * a.Tz + p0, a.Tz, a.Nc, jlo, jhi, lds);

  xc_epilogue<RM, TN>(acc, a, c, kt, p0);
}

template <int RM, int TN>
__global__ __launch_bounds__(kThreads, 2) void xc_mfma_kernel_s(const XcArgs a) {
  constexpr int PT = 128 * TN;
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes_s(RM)];
  const int item = blockIdx.x / a.nptiles;
  const int pt = blockIdx.x - item * a.nptiles;
  const int2 it = a.items[item];
  const int c = it.x, kt = it.y;
  const int p0 = pt * PT;
  int jlo = a.S - p0 - PT + 1;
  jlo = jlo > 0 ? (jlo / kBQ) * kBQ : 0;
  int jhi = a.S + a.L - p0;
  jhi = min(a.Lp, ((jhi + kBQ - 1) / kBQ) * kBQ);
  f32x4 acc[RM][TN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the tile image always has 208 rows per slot-row; RM < 13 cannot use it (row stride), so the
  // split engine serves the full 208-event tiles and the float32 engine the remainder tiles
  hankel_mainloop_bf16x3<RM, TN>(
      acc, a.xt16 + (int64_t)kt * a.Nc * (a.Lp / 8) * 2 * kKtMax * 8, a.Lp,
      a.g16 + (int64_t)c * a.Nc * 16 * a.V, a.V, p0 / 8, a.Nc, jlo, jhi, lds);
  xc_epilogue<RM, TN>(acc, a, c, kt, p0);
}

static constexpr size_t kFin2PrefixBudget = (size_t)4 << 30;   // bytes of float64 prefix sums xc_finish2 may carve
// One wave per pair: float64 re-evaluation of the winning lag and its two
// neighbours, reference scrub, cosine sub-sample fit, outputs.
__global__ __launch_bounds__(256) void xc_finish_kernel(
    const float* __restrict__ td, const unsigned long long* __restrict__ keys,
    const double* __restrict__ esum, const double* __restrict__ essd,
    const double* __restrict__ esq, int N, int n, int Nc, int S, int Pz, int trunc,
    int64_t pair_begin, int64_t pair_end, float* __restrict__ cc_out, int32_t* __restrict__ lag_out,
    float* __restrict__ ss_out, int T) {
  const int lane = threadIdx.x & 63;
  // Pairs are walked tile by tile (T x T events, upper triangle of tiles): the events a tile
  // touches (2 T n floats: 160 MB at T = 512, n = 39 000) stay in the Infinity Cache while the
  // tile's T^2 pairs are evaluated.  In plain condensed order every pair streamed its later
  // event from HBM (312 GB for 2000 events of 39 000 samples: 169 ms).
  const int per_tile = T * T / 4;                       // blocks per tile, 4 pairs (waves) each
  int t = (int)(blockIdx.x / per_tile);
  const int r = (int)(blockIdx.x - (int64_t)t * per_tile);
  const int ntile = (N + T - 1) / T;
  int ti = 0;
  while (t >= ntile - ti) { t -= ntile - ti; ++ti; }    // tile (ti, tj), ti <= tj
  const int tj = ti + t;
  int lb, lc;
  if (T % 128 == 0) {
    // Inside a tile, L2 locality per XCD (blocks are dealt to the 8 XCDs round-robin): an XCD
    // owns whole groups of 16 later events (2.5 MB at n = 39 000: resident in its 4 MB L2) and
    // walks all T earlier events against a group before taking the next, so only the earlier
    // event streams in -- once per 16 pairs.
    const int xcd = r & 7, seq = r >> 3;
    const int per_cg = T * 4;                           // blocks per group: T earlier events x 16 / 4
    const int cg = (seq / per_cg) * 8 + xcd, rem = seq % per_cg;
    lb = rem >> 2;
    lc = cg * 16 + (rem & 3) * 4 + (threadIdx.x >> 6);
  } else {
    const int local = r * 4 + (threadIdx.x >> 6);
    lb = local / T;
    lc = local % T;
  }
  const int64_t b = (int64_t)ti * T + lb, c = (int64_t)tj * T + lc;
  // the four waves of a block share the earlier event b (lc differs): b is staged through LDS
  // once per block, and the loop below is organised around ONE pass over the later event c --
  // per element one global load instead of four (x_b, and x_c at the three lags)
  constexpr int kCh = 4096, kPad = 64;
  __shared__ float x1s[kCh + 2 * kPad];
  const bool stage = Nc <= kPad;                        // uniform; else: the plain loop
  bool live = b < c && c < N;
  int64_t idx = 0;
  if (live) {
    idx = condensed_index(N, b, c);
    live = idx >= pair_begin && idx < pair_end;
  }
  unsigned long long key = 0ull;
  double ssd1 = 0.0, sum1 = 0.0;
  if (live) {
    key = keys[idx];
    ssd1 = essd[b]; sum1 = esum[b];
    if (key == 0ull || !(ssd1 > 0.0) || !(esq[c] > 0.0)) {
      // construct.py:463-464: all-NaN CC vector -> (0, 0, 0)
      if (lane == 0) { cc_out[idx] = 0.f; lag_out[idx] = 0; ss_out[idx] = 0.f; }
      live = false;
    }
  }
  if (!stage && !live) return;
  // a block whose four pairs are all dead (below the diagonal of a diagonal tile, past the last event,
  // outside the rank's pair range) would still stage the earlier event through LDS ten times over
  if (stage && !__syncthreads_or(live)) return;
  const int p = live ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0;
  const float* x1 = td + (b < N ? b : 0) * (int64_t)n;
  const float* x2 = td + (c < N ? c : 0) * (int64_t)n;
  double dot[3] = {0, 0, 0}, sw[3] = {0, 0, 0}, sq[3] = {0, 0, 0}, swc = 0.0, sqc = 0.0;
  const int sh0 = (p - 1 - S) * Nc;                     // lag l pairs x_b[j] with x_c[j + sh0 + l Nc]
  if (stage) {
    // index jm = the x_b index of the MIDDLE lag; x_c element t = jm + sh0 + Nc meets
    // x_b[jm + Nc] (lag 0), x_b[jm] (lag 1), x_b[jm - Nc] (lag 2)
    const int sh1 = sh0 + Nc;
    for (int J0 = -Nc; J0 < n + Nc; J0 += kCh) {
      __syncthreads();
      for (int i = threadIdx.x; i < kCh + 2 * Nc; i += 256) {
        const int j = J0 - Nc + i;
        x1s[i] = (j >= 0 && j < n && b < N) ? x1[j] : 0.f;
      }
      __syncthreads();
      if (live) {
        const int jend = min(J0 + kCh, n + Nc);
        for (int jb = J0; jb < jend; jb += 64) {        // jb: the iteration's first index (uniform)
          constexpr int kB = 4;           // interior iterations taken together
          if (kB > 1 && jb + 64 * kB <= jend && jb - Nc >= 0 && jb + 64 * kB - 1 + Nc < n && jb + sh1 >= 0 &&
              jb + 64 * kB - 1 + sh1 < n) {
            // kB interior iterations: their loads of x_c first (all in flight), then the same sums in
            // the same order
            float vb[kB];
#pragma unroll
            for (int k = 0; k < kB; ++k) vb[k] = x2[jb + 64 * k + lane + sh1];
#pragma unroll
            for (int k = 0; k < kB; ++k) {
              const int i = jb + 64 * k + lane - J0 + Nc;
              const double v = (double)vb[k];
              dot[0] += (double)x1s[i + Nc] * v;
              dot[1] += (double)x1s[i] * v;
              dot[2] += (double)x1s[i - Nc] * v;
              swc += v;
              sqc += v * v;
            }
            jb += 64 * (kB - 1);
            continue;
          }
          const int jm = jb + lane, t = jm + sh1;
          const int i = jm - J0 + Nc;                   // x1s index of x_b[jm]
          if (jb - Nc >= 0 && jb + 63 + Nc < n && jb + sh1 >= 0 && jb + 63 + sh1 < n) {
            // interior: every lane valid at all three lags, which all see the same x_c element
            const double v = (double)x2[t];
            dot[0] += (double)x1s[i + Nc] * v;
            dot[1] += (double)x1s[i] * v;
            dot[2] += (double)x1s[i - Nc] * v;
            swc += v;
            sqc += v * v;
          } else if (jm < jend) {
            const bool tok = t >= 0 && t < n;
            const double v = tok ? (double)x2[min(max(t, 0), n - 1)] : 0.0;
#pragma unroll
            for (int l = 0; l < 3; ++l) {
              const int jl = jm + (1 - l) * Nc;
              const bool ok = tok && jl >= 0 && jl < n;
              const double a1 = (double)x1s[i + (1 - l) * Nc];
              const double vv = ok ? v : 0.0;
              dot[l] += a1 * vv;
              sw[l] += vv;
              sq[l] += vv * vv;
            }
          }
        }
      }
    }
#pragma unroll
    for (int l = 0; l < 3; ++l) { sw[l] += swc; sq[l] += sqc; }
    if (!live) return;
  } else {
  // branch-free: out-of-range lags read a clamped address and contribute 0, so the four loads of
  // an iteration (and of the unrolled neighbours) are in flight together
#pragma unroll 4
  for (int j = lane; j < n; j += 64) {
    const double a1 = (double)x1[j];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const int jj = j + sh0 + l * Nc;
      const bool ok = jj >= 0 && jj < n;
      const double v = ok ? (double)x2[min(max(jj, 0), n - 1)] : 0.0;
      dot[l] += a1 * v;
      sw[l] += v;
      sq[l] += v * v;
    }
  }
  }
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      dot[l] += __shfl_down(dot[l], off, 64);
      sw[l] += __shfl_down(sw[l], off, 64);
      sq[l] += __shfl_down(sq[l], off, 64);
    }
  if (lane != 0) return;
  const double dn = (double)n;
  double cv[3];
  for (int l = 0; l < 3; ++l) {
    const double vw = clamp_var(sq[l], sw[l], dn);
    // zero-variance window: the reference gets +-inf and scrubs it to 0
    double v = vw > 0.0 ? (dot[l] - sum1 * sw[l] / dn) / sqrt(ssd1 * vw) : 0.0;
    if (v > 1.0 + (double)kScrubTol || v < -1.0 - (double)kScrubTol) v = 0.0;
    cv[l] = v;
  }
  double tau = 0.0;
  if (p != 0 && p != Pz - 1) {                       // construct.py:409-410
    const double cb4 = cv[0], cn = cv[1], caf = cv[2];
    const double alpha = acos((cb4 + caf) / (2.0 * cn));
    const double alsi = sin(alpha);
    tau = -(atan((cb4 - caf) / (2.0 * cn * alsi)) / alpha);
    if (fabs(tau) > 0.5) tau = (double)p;            // construct.py:418-421 (returns ind)
  }
  cc_out[idx] = (float)cv[1];
  lag_out[idx] = (p + 1 + trunc) * Nc - n;           // construct.py:466
  ss_out[idx] = (float)tau;
}

// Float64 prefix sums of every event and of its squares: ps[e][t] = sum_{u < t} x_e[u],
// pq[e][t] = sum_{u < t} x_e[u]^2, t = 0 .. n.  The finishing kernel reads the zero-padded window sums
// of the reference's normalisation off them in O(1) per lag.
__global__ __launch_bounds__(256) void xc_prefix_kernel(const float* __restrict__ td, int n,
                                                        double* __restrict__ ps, double* __restrict__ pq) {
  __shared__ double ws1[2][4], ws2[2][4];
  const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* x = td + (int64_t)e * n;
  double* o1 = ps + (int64_t)e * (n + 1);
  double* o2 = pq + (int64_t)e * (n + 1);
  double c1 = 0.0, c2 = 0.0;                            // sums of the tiles before this one
  int par = 0;
  for (int j0 = 0; j0 < n; j0 += 256, par ^= 1) {       // coalesced tiles of 256 samples
    const int j = j0 + tid;
    const double v = j < n ? (double)x[j] : 0.0;
    const double s1 = wave_incl_scan(v, lane), s2 = wave_incl_scan(v * v, lane);
    if (lane == 63) { ws1[par][wave] = s1; ws2[par][wave] = s2; }
    __syncthreads();                                    // (the other parity's slots are free: one barrier per tile)
    double b1 = c1, b2 = c2;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) { b1 += ws1[par][w]; b2 += ws2[par][w]; }
      c1 += ws1[par][w]; c2 += ws2[par][w];
    }
    if (j < n) {
      o1[j] = b1 + s1 - v; o2[j] = b2 + s2 - v * v;     // exclusive
      if (j == n - 1) { o1[n] = b1 + s1; o2[n] = b2 + s2; }
    }
  }
}

// The finishing kernel, second version (events with Nc <= 64): eight pairs per workgroup -- one wave
// each -- share the earlier event b.  The winning lag and its two neighbours are re-evaluated in
// float64 exactly as before; what changed is where the time went (rocprofv3, profiles/x4_*: waves
// parked 43 % of the time, issue-stalled 30 %, 9 double-rate instructions per element):
//   * x_b is staged through LDS ALREADY CONVERTED to float64, a chunk ahead in the other half of a
//     double buffer (one barrier per chunk instead of two with the loads exposed between them);
//     per element a wave converts only its own x_c sample: 1 convert + 3 fused multiply-adds;
//   * the window sums sum(w), sum(w^2) of the zero-padded shifted x_c come from the prefix sums
//     (xc_prefix_kernel) instead of two more accumulations per element;
//   * a wave walks only the samples its pair's shift leaves inside both events.
//   * round 4 (NC = the channel count at compile time, 1 .. 6): a lane takes kFinE CONSECUTIVE samples of a step
//     instead of every 64th.  The three lags pair sample e with x_b[e + 2 Nc], x_b[e + Nc], x_b[e]: one window
//     of kFinE + 2 Nc doubles per lane, fetched by 16-byte LDS reads, serves all 3 kFinE products (at Nc = 3:
//     one LDS instruction per sample instead of 3), and the lane's x_c samples come by 16- and 8-byte loads
//     (dword aligned: the pair's shift is arbitrary).  kFinE = 6: the lanes' windows start 48 bytes apart, an
//     odd number of 16-byte units, which is what keeps ds_read_b128's lane groups on distinct banks
//     (MI355X_MICROARCH.md, LDS; 8 doubles apart = 64 bytes would be a 4-way conflict).  What is left is the
//     stream of the later events from L2: 156 KB per pair, 0.31 TB per launch at configs[2] = 14 TB/s.
constexpr int kFinE = 6;   // samples per lane and step of the window form (2 x an odd number)
constexpr int kFinStep = 64 * kFinE; // samples per wave and step
constexpr int kFinCh = 4 * kFinStep; // x_b samples per staged chunk (2560)
constexpr int kFinPad = 64;
constexpr int kFinWaves = 8;
typedef float fin_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float fin_f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef double fin_d2 __attribute__((ext_vector_type(2)));
template <int NC>
__global__ __launch_bounds__(64 * kFinWaves) void xc_finish2_kernel(
    const float* __restrict__ td, const unsigned long long* __restrict__ keys,
    const double* __restrict__ esum, const double* __restrict__ essd,
    const double* __restrict__ esq, const double* __restrict__ ps, const double* __restrict__ pq,
    int N, int n, int Nc, int S, int Pz, int trunc, int64_t pair_begin, int64_t pair_end,
    float* __restrict__ cc_out, int32_t* __restrict__ lag_out, float* __restrict__ ss_out, int T) {
  __shared__ __attribute__((aligned(16))) double x1s[2][kFinCh + 2 * kFinPad];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // tile walk as in xc_finish_kernel: T x T events per tile, upper triangle of tiles, inside a tile an
  // XCD owns whole groups of 16 later events and walks every earlier event against them
  const int per_tile = T * T / kFinWaves;
  int t = (int)(blockIdx.x / per_tile);
  const int r = (int)(blockIdx.x - (int64_t)t * per_tile);
  const int ntile = (N + T - 1) / T;
  int ti = 0;
  while (t >= ntile - ti) { t -= ntile - ti; ++ti; }
  const int tj = ti + t;
  int lb, lc;
  if (T % 128 == 0) {
    const int xcd = r & 7, seq = r >> 3;
    const int per_cg = T * (16 / kFinWaves);            // blocks per group: T earlier events x 16 / 8
    const int cg = (seq / per_cg) * 8 + xcd, rem = seq % per_cg;
    lb = rem / (16 / kFinWaves);
    lc = cg * 16 + (rem % (16 / kFinWaves)) * kFinWaves + wave;
  } else {
    const int local = r * kFinWaves + wave;
    lb = local / T;
    lc = local % T;
  }
  const int64_t b = (int64_t)ti * T + lb, c = (int64_t)tj * T + lc;
  bool live = b < c && c < N;
  int64_t idx = 0;
  if (live) {
    idx = condensed_index(N, b, c);
    live = idx >= pair_begin && idx < pair_end;
  }
  unsigned long long key = 0ull;
  double ssd1 = 0.0, sum1 = 0.0;
  if (live) {
    key = keys[idx];
    ssd1 = essd[b]; sum1 = esum[b];
    if (key == 0ull || !(ssd1 > 0.0) || !(esq[c] > 0.0)) {
      // construct.py:463-464: all-NaN CC vector -> (0, 0, 0)
      if (lane == 0) { cc_out[idx] = 0.f; lag_out[idx] = 0; ss_out[idx] = 0.f; }
      live = false;
    }
  }
  if (!__syncthreads_or(live)) return;                  // (uniform: nothing to stage for)
  const int p = live ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0;
  const float* x1 = td + (b < N ? b : 0) * (int64_t)n;
  const float* x2 = td + (c < N ? c : 0) * (int64_t)n;
  const int sh0 = (p - 1 - S) * Nc;                     // lag l pairs x_b[j] with x_c[j + sh0 + l Nc]
  const int sh1 = sh0 + Nc;
  // index jm = the x_b index of the MIDDLE lag: x_c element t = jm + sh1 meets x_b[jm + Nc] (lag 0),
  // x_b[jm] (lag 1), x_b[jm - Nc] (lag 2); x_b is zero outside [0, n) in the staged image, so only
  // t has to stay inside x_c: jm in [max(-Nc, -sh1), min(n + Nc, n - sh1))
  const int jlo = live ? max(-Nc, -sh1) : 0, jhi = live ? min(n + Nc, n - sh1) : 0;
  double dot[3] = {0, 0, 0};
  auto stage_load = [&](int J0, float (&v)[kFinCh / (64 * kFinWaves) + 1]) {
#pragma unroll
    for (int q = 0; q <= kFinCh / (64 * kFinWaves); ++q) {
      const int i = threadIdx.x + 64 * kFinWaves * q;       // image index: x_b[J0 - Nc + i]
      const int j = J0 - Nc + i;
      v[q] = (i < kFinCh + 2 * Nc && j >= 0 && j < n && b < N) ? x1[j] : 0.f;
    }
  };
  auto stage_store = [&](int buf, const float (&v)[kFinCh / (64 * kFinWaves) + 1]) {
#pragma unroll
    for (int q = 0; q <= kFinCh / (64 * kFinWaves); ++q) {
      const int i = threadIdx.x + 64 * kFinWaves * q;
      if (i < kFinCh + 2 * Nc) x1s[buf][i] = (double)v[q];
    }
  };
  float sv[kFinCh / (64 * kFinWaves) + 1];
  stage_load(-Nc, sv);
  stage_store(0, sv);
  __syncthreads();
  int buf = 0;
  for (int J0 = -Nc; J0 < n + Nc; J0 += kFinCh, buf ^= 1) {
    const bool more = J0 + kFinCh < n + Nc;             // uniform
    if (more) stage_load(J0 + kFinCh, sv);
    const double* xs = x1s[buf];
    // this wave's share of the chunk, in steps of 64 samples from the chunk's start
    const int c0 = max(J0, jlo), c1 = min(J0 + kFinCh, jhi);
    // samples lo .. hi - 1, 64 at a time from a multiple of 64 behind the chunk's start
    auto walk64 = [&](int lo, int hi) {
      for (int jb = J0 + (lo - J0) / 64 * 64; jb < hi; jb += 64) {
        const int jm = jb + lane;
        const bool ok = jm >= lo && jm < hi;
        const double v = ok ? (double)x2[min(max(jm + sh1, 0), n - 1)] : 0.0;
        const int i = jm - J0 + Nc;
        dot[0] += xs[i + Nc] * v;
        dot[1] += xs[i] * v;
        dot[2] += xs[i - Nc] * v;
      }
    };
    if (NC > 0 && c0 < c1) {
      for (int jb = J0; jb < J0 + kFinCh; jb += kFinStep) {
        if (jb + kFinStep <= c0 || jb >= c1) continue;            // (uniform)
        if (jb < c0 || jb + kFinStep > c1) { walk64(max(jb, c0), min(jb + kFinStep, c1)); continue; }
        const int jm0 = jb + kFinE * lane;
        const float* src = x2 + jm0 + sh1;
        float v[kFinE];
        static_assert(kFinE % 4 == 2, "16-byte loads and one of 8 bytes; an odd number of 16-byte units of doubles");
#pragma unroll
        for (int q = 0; q < kFinE / 4; ++q) {
          const fin_f4u t = *reinterpret_cast<const fin_f4u*>(src + 4 * q);
          v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        {
          const fin_f2u t = *reinterpret_cast<const fin_f2u*>(src + kFinE - 2);
          v[kFinE - 2] = t.x; v[kFinE - 1] = t.y;
        }
        constexpr int kW = kFinE + 2 * (NC > 0 ? NC : 1);
        double w[kW];
        const fin_d2* wp = reinterpret_cast<const fin_d2*>(xs + (jm0 - J0));
#pragma unroll
        for (int q = 0; q < kW / 2; ++q) { const fin_d2 t = wp[q]; w[2 * q] = t.x; w[2 * q + 1] = t.y; }
#pragma unroll
        for (int e = 0; e < kFinE; ++e) {
          const double dv = (double)v[e];
          dot[0] += w[e + 2 * NC] * dv;
          dot[1] += w[e + NC] * dv;
          dot[2] += w[e] * dv;
        }
      }
    } else if (c0 < c1) {
      int jb = J0 + (c0 - J0) / 64 * 64;
      constexpr int kB = 4;
      for (; jb < c1; jb += 64) {
        if (jb >= c0 && jb + 64 * kB <= c1) {           // kB whole steps: their loads first, all in flight
          float vb[kB];
#pragma unroll
          for (int k = 0; k < kB; ++k) vb[k] = x2[jb + 64 * k + lane + sh1];
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int i = jb + 64 * k + lane - J0 + Nc;
            const double v = (double)vb[k];
            dot[0] += xs[i + Nc] * v;
            dot[1] += xs[i] * v;
            dot[2] += xs[i - Nc] * v;
          }
          jb += 64 * (kB - 1);
          continue;
        }
        const int jm = jb + lane;
        const bool ok = jm >= c0 && jm < c1;
        const double v = ok ? (double)x2[min(max(jm + sh1, 0), n - 1)] : 0.0;
        const int i = jm - J0 + Nc;
        dot[0] += xs[i + Nc] * v;
        dot[1] += xs[i] * v;
        dot[2] += xs[i - Nc] * v;
      }
    }
    if (more) stage_store(buf ^ 1, sv);
    __syncthreads();
  }
  if (!live) return;
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot[l] += __shfl_down(dot[l], off, 64);
  if (lane != 0) return;
  const double dn = (double)n;
  const double* p1 = ps + c * (int64_t)(n + 1);
  const double* p2 = pq + c * (int64_t)(n + 1);
  double cv[3];
  for (int l = 0; l < 3; ++l) {
    // x_c samples t = j + s with j in [0, n): the rest of the window is zero padding
    const int s = sh0 + l * Nc;
    const int t0 = min(max(s, 0), n), t1 = min(max(n + s, 0), n);
    const double sw = t1 > t0 ? p1[t1] - p1[t0] : 0.0, sq = t1 > t0 ? p2[t1] - p2[t0] : 0.0;
    const double vw = clamp_var(sq, sw, dn);
    // zero-variance window: the reference gets +-inf and scrubs it to 0
    double v = vw > 0.0 ? (dot[l] - sum1 * sw / dn) / sqrt(ssd1 * vw) : 0.0;
    if (v > 1.0 + (double)kScrubTol || v < -1.0 - (double)kScrubTol) v = 0.0;
    cv[l] = v;
  }
  double tau = 0.0;
  if (p != 0 && p != Pz - 1) {                       // construct.py:409-410
    const double cb4 = cv[0], cn = cv[1], caf = cv[2];
    const double alpha = acos((cb4 + caf) / (2.0 * cn));
    const double alsi = sin(alpha);
    tau = -(atan((cb4 - caf) / (2.0 * cn * alsi)) / alpha);
    if (fabs(tau) > 0.5) tau = (double)p;            // construct.py:418-421 (returns ind)
  }
  cc_out[idx] = (float)cv[1];
  lag_out[idx] = (p + 1 + trunc) * Nc - n;           // construct.py:466
  ss_out[idx] = (float)tau;
}

template <int RM>
int launch_items(dtx_ctx* ctx, XcArgs a, const int2* items_dev, int nitems) {
  if constexpr (RM == kRmMax) {
    if (a.xt16) {                       // split-bf16 engine for the full 208-event tiles
      constexpr int TNs = tn_s(RM);
      constexpr int PTs = 128 * TNs;
      a.items = items_dev;
      a.nptiles = (a.Pz + PTs - 1) / PTs;
      const int64_t nblk = (int64_t)a.nptiles * nitems;
      const int64_t per = ((int64_t)1 << 20) / a.nptiles * a.nptiles;
      for (int64_t off = 0; off < nblk; off += per) {
        XcArgs b = a;
        b.items = items_dev + off / a.nptiles;
        const int64_t cnt = std::min(per, nblk - off);
        hipLaunchKernelGGL((xc_mfma_kernel_s<RM, TNs>), dim3((unsigned)cnt), dim3(kThreads), 0,
                           ctx->stream, b);
        DTX_HIP(ctx, hipGetLastError());
      }
      return DTX_OK;
    }
  }
  constexpr int TN = tn_for(RM);
  constexpr int PT = 128 * TN;
  a.items = items_dev;
  a.nptiles = (a.Pz + PT - 1) / PT;
  const int64_t nblk = (int64_t)a.nptiles * nitems;
  // grid.x limit is 2^31-1 blocks; split anyway to keep launches bounded
  const int64_t per = ((int64_t)1 << 20) / a.nptiles * a.nptiles;
  for (int64_t off = 0; off < nblk; off += per) {
    XcArgs b = a;
    b.items = items_dev + off / a.nptiles;
    const int64_t cnt = std::min(per, nblk - off);
    hipLaunchKernelGGL((xc_mfma_kernel<RM, TN>), dim3((unsigned)cnt), dim3(kThreads), 0,
                       ctx->stream, b);
    DTX_HIP(ctx, hipGetLastError());
  }
  return DTX_OK;
}

int launch_items_rm(dtx_ctx* ctx, int rm, const XcArgs& a, const int2* items_dev, int nitems) {
  switch (rm) {
    case 1: return launch_items<1>(ctx, a, items_dev, nitems);
    case 2: return launch_items<2>(ctx, a, items_dev, nitems);
    case 3: return launch_items<3>(ctx, a, items_dev, nitems);
    case 4: return launch_items<4>(ctx, a, items_dev, nitems);
    case 5: return launch_items<5>(ctx, a, items_dev, nitems);
    case 6: return launch_items<6>(ctx, a, items_dev, nitems);
    case 7: return launch_items<7>(ctx, a, items_dev, nitems);
    case 8: return launch_items<8>(ctx, a, items_dev, nitems);
    case 9: return launch_items<9>(ctx, a, items_dev, nitems);
    case 10: return launch_items<10>(ctx, a, items_dev, nitems);
    case 11: return launch_items<11>(ctx, a, items_dev, nitems);
    case 12: return launch_items<12>(ctx, a, items_dev, nitems);
    case 13: return launch_items<13>(ctx, a, items_dev, nitems);
  }
  return dtx_fail(ctx, DTX_E_ARG, "bad row-block count %d", rm);
}

// first row b of the condensed index idx
int64_t row_of_index(int64_t N, int64_t idx) {
  int64_t lo = 0, hi = N - 2;
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) / 2;
    if (condensed_index(N, mid, mid + 1) <= idx) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

}  // namespace

extern "C" {

int dtx_xcorr_allpairs(dtx_ctx* ctx, const float* td_dev, int N, int n, int Nc, float* cc_dev,
                       int32_t* lag_dev, float* subsamp_dev, int64_t pair_begin,
                       int64_t pair_end) {
  if (!ctx) return DTX_E_ARG;
  if (!td_dev || !cc_dev || !lag_dev || !subsamp_dev)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_xcorr_allpairs: null pointer");
  if (N < 2 || n <= 0 || Nc <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_xcorr_allpairs: need N >= 2, n > 0, Nc > 0");
  if (n % Nc != 0)
    // construct.py:434-436: 'Lengths not equal on multiplexed data, cannot correlate'
    return dtx_fail(ctx, DTX_E_ARG, "event length n=%d is not a multiple of Nc=%d", n, Nc);
  const int64_t npairs = (int64_t)N * (N - 1) / 2;
  if (pair_begin < 0 || pair_end > npairs || pair_begin > pair_end)
    return dtx_fail(ctx, DTX_E_ARG, "pair range [%lld, %lld) outside [0, %lld]",
                    (long long)pair_begin, (long long)pair_end, (long long)npairs);
  if (pair_begin == pair_end) return DTX_OK;
  DtxDeviceGuard g(ctx->device);
  const int64_t cnt = pair_end - pair_begin;
  const int L = n / Nc;
  const int trunc = n / (2 * Nc) - 1;               // construct.py:439
  if (trunc <= 0) {
    // result[trunc:-trunc] is empty -> ValueError path -> (0, 0, 0) (construct.py:463-464)
    DTX_HIP(ctx, hipMemsetAsync(cc_dev + pair_begin, 0, cnt * sizeof(float), ctx->stream));
    DTX_HIP(ctx, hipMemsetAsync(lag_dev + pair_begin, 0, cnt * sizeof(int32_t), ctx->stream));
    DTX_HIP(ctx, hipMemsetAsync(subsamp_dev + pair_begin, 0, cnt * sizeof(float), ctx->stream));
    return DTX_OK;
  }
  const int S = (L - 1) - trunc;
  const int Pz = 2 * S + 1;
  const int Lp = (int)dtx_round_up(L, kBQ);
  const int64_t Ppad = dtx_round_up(Pz, 1024);
  const int64_t Tz = Ppad + Lp + 64;
  const int nkt = (N + kKtMax - 1) / kKtMax;

  // AUTO / FFT: events of kXcFftMinL .. kFftLongMaxL samples per channel go through the
  // frequency domain (dtx_xcfft.hip); the contraction kernels below serve the rest
  const bool use_fft = (ctx->engine == DTX_ENGINE_AUTO || ctx->engine == DTX_ENGINE_FFT) &&
                       L >= kXcFftMinL && L <= kFftLongMaxL;
  // ---- workspaces (one arena)
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_xt = carve(use_fft ? 0 : (size_t)nkt * Nc * Lp * kKtMax * sizeof(float));
  const size_t o_z = carve((size_t)N * Nc * Tz * sizeof(float));
  const size_t o_mw = carve((size_t)N * Ppad * sizeof(float));
  const size_t o_rw = carve((size_t)N * Ppad * sizeof(float));
  const size_t o_es = carve((size_t)N * sizeof(double));
  const size_t o_ed = carve((size_t)N * sizeof(double));
  const size_t o_eq = carve((size_t)N * sizeof(double));
  const size_t o_fs = carve((size_t)N * sizeof(float));
  const size_t o_fr = carve((size_t)N * sizeof(float));
  const size_t o_keys = carve((size_t)npairs * sizeof(unsigned long long));
  // the second finishing kernel (float64 image of x_b, prefix sums of EVERY event: 16 (n + 1) bytes each --
  // 1.25 GB at configs[2]).  Past kFin2PrefixBudget the first finishing kernel (no prefix workspace) takes
  // over: 20 000 events would otherwise ask for 12 GB per context.
  const bool fin2 = Nc <= kFinPad &&
                    (size_t)2 * N * ((size_t)n + 1) * sizeof(double) <= kFin2PrefixBudget;
  const size_t o_ps = carve(fin2 ? (size_t)N * (n + 1) * sizeof(double) : 0);
  const size_t o_pq = carve(fin2 ? (size_t)N * (n + 1) * sizeof(double) : 0);
  const bool split = !use_fft && ctx->engine != DTX_ENGINE_FP32 && L >= kSplitMinL;
  const int64_t Vz = Tz / 8;
  const size_t o_xt16 = carve(split ? (size_t)nkt * Nc * (Lp / 8) * 2 * kKtMax * 8 * sizeof(uint16_t) : 0);
  const size_t o_g = carve(split ? (size_t)N * Nc * 16 * Vz * sizeof(i32x4) : 0);
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc, off))) return rc;
  char* base = (char*)ctx->ws_misc.p;
  float* xt = (float*)(base + o_xt);
  float* z = (float*)(base + o_z);
  float* mean_w = (float*)(base + o_mw);
  float* rsw = (float*)(base + o_rw);
  double* esum = (double*)(base + o_es);
  double* essd = (double*)(base + o_ed);
  double* esq = (double*)(base + o_eq);
  float* fsum = (float*)(base + o_fs);
  float* frs = (float*)(base + o_fr);
  unsigned long long* keys = (unsigned long long*)(base + o_keys);
  uint16_t* xt16 = (uint16_t*)(base + o_xt16);
  i32x4* g16 = (i32x4*)(base + o_g);

  // ---- work items: (c, kt) for every column c and row tile kt that holds a pair in range
  const int64_t b_begin = row_of_index(N, pair_begin);
  const int64_t b_end = row_of_index(N, pair_end - 1);
  std::vector<std::vector<int2>> items(kRmMax + 1);
  for (int kt = 0; kt < (use_fft ? 0 : nkt); ++kt) {   // kt-major: concurrent blocks share the XT tile
    const int64_t r0 = (int64_t)kt * kKtMax;
    for (int c = 1; c < N; ++c) {
      const int64_t lo = std::max<int64_t>(r0, b_begin);
      const int64_t hi = std::min<int64_t>({r0 + kKtMax - 1, (int64_t)c - 1, b_end});
      if (lo > hi) continue;
      int rm = (int)((hi - r0) / 16 + 1);
      // the split engine works on whole 208-row tiles and is ~3.6x faster per row than the
      // float32 engine: a remainder tile of >= 4 row blocks is cheaper as a full split tile
      if (split && rm >= 4) rm = kRmMax;
      items[rm].push_back(make_int2(c, kt));
    }
  }
  size_t nitems = 0;
  for (auto& v : items) nitems += v.size();
  if ((rc = dtx_reserve(ctx, ctx->ws_misc2, std::max<size_t>(nitems, 1) * sizeof(int2)))) return rc;
  {
    std::vector<int2> flat;
    flat.reserve(nitems);
    for (auto& v : items) flat.insert(flat.end(), v.begin(), v.end());
    // ws_misc2 may be read by a previous launch still in flight on the stream
    DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DTX_HIP(ctx, hipMemcpy(ctx->ws_misc2.p, flat.data(), flat.size() * sizeof(int2),
                           hipMemcpyHostToDevice));
  }

  // ---- prep kernels
  // the finishing kernel's prefix sums depend on the events only: on the side stream, beside the preparation and
  // the head of the lag search (joined in front of the finishing kernel)
  if (fin2) {
    if ((rc = dtx_aux_fork(ctx))) return rc;
    hipLaunchKernelGGL(xc_prefix_kernel, dim3((unsigned)N), dim3(256), 0, ctx->aux, td_dev, n,
                       (double*)(base + o_ps), (double*)(base + o_pq));
    DTX_HIP(ctx, hipGetLastError());
  }
  // (an error return below leaves nothing running on the side stream that the next call does not wait for)
  auto bail = [&](int r) { if (fin2) dtx_aux_join(ctx); return r; };
  if (!use_fft) {
    dim3 grid((unsigned)std::min<int64_t>(((int64_t)Nc * Lp + 3) / 4, 8192), 1, (unsigned)nkt);
    hipLaunchKernelGGL(xc_transpose_kernel, grid, dim3(256), 0, ctx->stream, td_dev, N, n, Nc, L,
                       Lp, xt);
    DTX_HIP(ctx, hipGetLastError());
  }
  {
    dim3 grid((unsigned)((Tz + 255) / 256), (unsigned)Nc, (unsigned)N);
    hipLaunchKernelGGL(xc_pad_kernel, grid, dim3(256), 0, ctx->stream, td_dev, n, Nc, L, S, Tz, z);
    DTX_HIP(ctx, hipGetLastError());
  }
  if (split) {
    dim3 grid((unsigned)std::min<int64_t>((int64_t)Nc * (Lp / 8), 4096), 1, (unsigned)nkt);
    hipLaunchKernelGGL(xc_transpose16_kernel, grid, dim3(256), 0, ctx->stream, td_dev, N, n, Nc, L,
                       Lp, xt16);
    DTX_HIP(ctx, hipGetLastError());
    if ((rc = dtx_launch_demux16(ctx, z, Tz, N * Nc, g16))) return bail(rc);
  }
  hipLaunchKernelGGL(xc_event_stats_kernel, dim3((unsigned)N), dim3(256), 0, ctx->stream, td_dev, n,
                     esum, essd, esq, fsum, frs);
  DTX_HIP(ctx, hipGetLastError());
  {
    dim3 grid((unsigned)((Pz + kStatsPB - 1) / kStatsPB), (unsigned)N);
    hipLaunchKernelGGL(win_stats_kernel, grid, dim3(256), 0, ctx->stream, (const float*)z, Tz, Nc, L,
                       (int64_t)Pz, Ppad, -1.0 /* 1/sqrt mode */, mean_w, rsw);
    DTX_HIP(ctx, hipGetLastError());
  }
  DTX_HIP(ctx, hipMemsetAsync(keys + pair_begin, 0, cnt * sizeof(unsigned long long), ctx->stream));

  // ---- the MFMA launches, one per row-block count
  XcArgs a;
  a.xt = xt; a.z = z; a.fsum = fsum; a.frs = frs; a.mean_w = mean_w; a.rsw = rsw; a.keys = keys;
  a.xt16 = split ? xt16 : nullptr; a.g16 = g16; a.V = Vz;
  a.items = nullptr; a.Tz = Tz; a.Ppad = Ppad; a.pair_begin = pair_begin; a.pair_end = pair_end;
  a.N = N; a.Nc = Nc; a.L = L; a.Lp = Lp; a.S = S; a.Pz = Pz; a.nptiles = 0;
  if (use_fft) {
    if ((rc = dtx_xcfft_run(ctx, td_dev, N, n, Nc, L, Pz, Ppad, Tz, z, fsum, frs, mean_w, rsw, keys,
                            pair_begin, pair_end, b_begin, b_end)))
      return bail(rc);
  } else {
  { const int rck = dtx_kev_next(ctx); if (rck) return bail(rck); }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
  {
    const int2* p = (const int2*)ctx->ws_misc2.p;
    for (int rm = kRmMax; rm >= 1; --rm) {
      size_t skip = 0;
      for (int q = 0; q < rm; ++q) skip += items[q].size();
      if (items[rm].empty()) continue;
      if ((rc = launch_items_rm(ctx, rm, a, p + skip, (int)items[rm].size()))) return bail(rc);
    }
  }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  dtx_kev_commit(ctx);
  }

  if (fin2) {
    double* ps = (double*)(base + o_ps);
    double* pq = (double*)(base + o_pq);
    if ((rc = dtx_aux_join(ctx))) return rc;
    const int T = std::min(512, (N + 7) / 8 * 8);
    const int64_t ntile = (N + T - 1) / T;
    const int64_t fgrid = ntile * (ntile + 1) / 2 * ((int64_t)T * T / kFinWaves);
    if (fgrid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "too many events");
    auto fin = [&](auto ncc) {
      constexpr int NC = ncc;
      hipLaunchKernelGGL((xc_finish2_kernel<NC>), dim3((unsigned)fgrid), dim3(64 * kFinWaves), 0, ctx->stream,
                         td_dev, (const unsigned long long*)keys, (const double*)esum,
                         (const double*)essd, (const double*)esq, (const double*)ps, (const double*)pq, N, n,
                         Nc, S, Pz, trunc, pair_begin, pair_end, cc_dev, lag_dev, subsamp_dev, T);
    };
    switch (Nc) {              // (the window form: channel count at compile time)
      case 1: fin(std::integral_constant<int, 1>{}); break;
      case 2: fin(std::integral_constant<int, 2>{}); break;
      case 3: fin(std::integral_constant<int, 3>{}); break;
      case 4: fin(std::integral_constant<int, 4>{}); break;
      case 6: fin(std::integral_constant<int, 6>{}); break;
      default: fin(std::integral_constant<int, 0>{}); break;
    }
  } else {
    const int T = std::min(512, (N + 3) / 4 * 4);
    const int64_t ntile = (N + T - 1) / T;
    const int64_t fgrid = ntile * (ntile + 1) / 2 * ((int64_t)T * T / 4);
    if (fgrid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "too many events");
    hipLaunchKernelGGL(xc_finish_kernel, dim3((unsigned)fgrid), dim3(256), 0, ctx->stream,
                       td_dev, (const unsigned long long*)keys, (const double*)esum,
                       (const double*)essd, (const double*)esq, N, n, Nc, S, Pz, trunc, pair_begin,
                       pair_end, cc_dev, lag_dev, subsamp_dev, T);
  }
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

int dtx_xcorr_allpairs_host(dtx_ctx* ctx, const float* td_host, int N, int n, int Nc,
                            float* cc_host, int32_t* lag_host, float* subsamp_host) {
  if (!ctx) return DTX_E_ARG;
  if (!td_host || !cc_host || !lag_host || !subsamp_host)
    return dtx_fail(ctx, DTX_E_ARG, "null pointer");
  if (N < 2 || n <= 0) return dtx_fail(ctx, DTX_E_ARG, "need N >= 2, n > 0");
  DtxDeviceGuard g(ctx->device);
  const int64_t npairs = (int64_t)N * (N - 1) / 2;
  const size_t tdb = (size_t)N * n * sizeof(float);
  const size_t o1 = (tdb + 255) / 256 * 256, o2 = o1 + (npairs * 4 + 255) / 256 * 256,
               o3 = o2 + (npairs * 4 + 255) / 256 * 256, tot = o3 + npairs * 4;
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc3, tot))) return rc;
  char* base = (char*)ctx->ws_misc3.p;
  DTX_HIP(ctx, hipMemcpyAsync(base, td_host, tdb, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dtx_xcorr_allpairs(ctx, (const float*)base, N, n, Nc, (float*)(base + o1),
                               (int32_t*)(base + o2), (float*)(base + o3), 0, npairs)))
    return rc;
  DTX_HIP(ctx, hipMemcpyAsync(cc_host, base + o1, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  DTX_HIP(ctx, hipMemcpyAsync(lag_host, base + o2, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
  DTX_HIP(ctx, hipMemcpyAsync(subsamp_host, base + o3, npairs * 4, hipMemcpyDeviceToHost,
                              ctx->stream));
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DTX_OK;
}

}  // extern "C"
