// Trigger-time estimates on the device, one workgroup per trigger:
//   * the STA / LTA of the detection statistic at the trigger (detex/detect.py:501-524
//     _getStaLtaArray: centred rolling means of |DS|, evaluated only where a trigger fell),
//   * the ingredients of the magnitude / SNR estimates of detex/detect.py:447-499 (_estMag,
//     _estPEMag / _estSTDMag 637-664): mean and std of the triggering data window, the variance of
//     its projection into the subspace U^T (U w) (never the reference's n x n UtU, detect.py:367),
//     its dot product with every training event (the lag-0 value of fast_normcorr,
//     construct.py:469-483) and the median rolling standard deviation of the pre-event noise.
// Everything is accumulated in float64; the host turns the numbers into Mag / ProEnMag / SNR
// (a handful of scalar operations per trigger, detex_amd/detect.py).
#include "dtx_internal.h"

namespace {

constexpr int kTT = 512;          // threads per workgroup (8 waves)
constexpr int kTrigHead = 6;      // leading doubles of an output row, then one dot product per event

struct TrigArgs {
  const float* x;        // multiplexed chunks, chunk b at x + b * stride
  int64_t M, stride;
  const int64_t* ctab;   // ragged batch: chunk b = min(ctab[2 b + 1], M) samples at x + ctab[2 b] (NULL: uniform)
  const float* ds;       // [nchunk][D][ldp]
  int64_t P, ldp;
  const int32_t* trig;   // [ntrig][5]: chunk, detector, window index, first sample of the STA window (-1: none), of the LTA window
  int ntrig, sta_len, lta_len;
  const float* U;        // [Ktot][n] stacked basis
  const int32_t* doff;   // [D + 1] detector rows
  const float* ewf;      // [Etot][n] training-event waveforms (trimmed, as the detector saw them)
  const int32_t* eoff;   // [D + 1] events of each detector
  double* out;           // [ntrig][kTrigHead + emax]
  float* scratch;        // [gridDim.x][scap] rolling standard deviations of one trigger
  int64_t scap;
  int D, n, Nc, emax;
  int full;              // 0: STA / LTA only (no magnitude ingredients asked for)
};

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the workgroup; every thread gets the result (red: 8 doubles of LDS)
__device__ inline double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < kTT / 64; ++w) s += red[w];
  return s;
}

// k-th smallest (0-based) of count non-negative floats, and the (k + 1)-th if next != NULL (the two middle
// values of an even count): radix select on the bit patterns, 4 bits per pass, counted in REGISTERS and reduced
// over the workgroup -- no atomics, no walk of the buckets by one thread, two barriers a pass.  (Rounds 1-3
// counted 8 bits per pass with LDS atomics from values in a global scratch buffer: nine passes of L2 round trips
// one behind the other.)  Returns the value(s) to every thread.  hist: 8 x 16 words of LDS.
// each(f): calls f(bit pattern) for every value this thread contributes (any partition of the multiset over the
// workgroup's threads will do)
template <class Each>
__device__ inline float select_rank_impl(Each each, int k, unsigned* hist, unsigned* sh, float* next) {
  unsigned prefix = 0u, mask = 0u;
  int rank = k;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int shift = 28; shift >= 0; shift -= 4) {
    unsigned c[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = 0u;
    each([&](unsigned u) {
      const unsigned d = (u & mask) == prefix ? (u >> shift) & 15u : 16u;
#pragma unroll
      for (int b = 0; b < 16; ++b) c[b] += d == (unsigned)b;
    });
#pragma unroll
    for (int b = 0; b < 16; ++b) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c[b] += __shfl_xor(c[b], off, 64);
    }
    __syncthreads();                       // (the previous pass' readers of hist are done)
    if (ln == 0) {
#pragma unroll
      for (int b = 0; b < 16; ++b) hist[16 * wv + b] = c[b];
    }
    __syncthreads();
    // every wave finds the digit for itself (no serial walk by one thread, no broadcast, no third barrier): lane b
    // adds bucket b over the eight waves, an inclusive scan over the 16 buckets, the first bucket whose running
    // total passes the rank
    int cb = 0;
#pragma unroll
    for (int w = 0; w < kTT / 64; ++w) cb += (int)hist[16 * w + (ln & 15)];
    int incl = cb;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int o = __shfl_up(incl, off, 16);
      if ((ln & 15) >= off) incl += o;
    }
    const unsigned long long over = __ballot(incl > rank) & 0xffffull;
    const int dg = over ? __ffsll((long long)over) - 1 : 15;
    const int acc = __shfl(incl - cb, dg, 16);
    prefix |= (unsigned)dg << shift;
    mask |= 15u << shift;
    rank -= acc;
  }
  const float val = __uint_as_float(prefix);
  if (next != nullptr) {
    // the (k + 1)-th smallest: val again if more than k + 1 values are <= val, else the smallest value above it
    unsigned le = 0u, mn = 0xffffffffu;
    each([&](unsigned u) {
      le += u <= prefix;
      if (u > prefix) mn = min(mn, u);
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      le += __shfl_xor(le, off, 64);
      mn = min(mn, (unsigned)__shfl_xor(mn, off, 64));
    }
    __syncthreads();
    if (ln == 0) { hist[2 * wv] = le; hist[2 * wv + 1] = mn; }
    __syncthreads();
    unsigned tle = 0u, tmn = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < kTT / 64; ++w) { tle += hist[2 * w]; tmn = min(tmn, hist[2 * w + 1]); }
    *next = (int)tle > k + 1 ? val : __uint_as_float(tmn);
    __syncthreads();
  }
  return val;
}
// ... of count values in memory
__device__ inline float select_rank(const float* __restrict__ v, int count, int k, unsigned* hist,
                                    unsigned* sh, float* next = nullptr) {
  return select_rank_impl([&](auto f) {
    for (int i = threadIdx.x; i < count; i += kTT) f(__float_as_uint(v[i]));
  }, k, hist, sh, next);
}
// ... of the values the threads hold in registers: r[j], j < NR, 0xffffffff (never a non-negative float's
// pattern: it sorts behind everything and is never reached) where a thread has none
template <int NR>
__device__ inline float select_rank_regs(const unsigned (&r)[NR], int k, unsigned* hist, unsigned* sh,
                                         float* next = nullptr) {
  return select_rank_impl([&](auto f) {
#pragma unroll
    for (int j = 0; j < NR; ++j) f(r[j]);
  }, k, hist, sh, next);
}
constexpr int kRegW = 96;
// four consecutive floats at any 4-byte aligned address in ONE load (global memory takes dword-aligned
// multi-dword loads): a thread that walks its own run of samples issues a quarter of the instructions -- its lanes
// are 70 .. 90 samples apart, so every load instruction is 64 separate cache-line requests whatever its width
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load4_guarded(const float* __restrict__ p, int idx, int limit, float (&o)[4]) {
  if (idx + 4 <= limit) {
    const f4u v = *reinterpret_cast<const f4u*>(p + idx);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = idx + e < limit ? p[idx + e] : 0.f;
  }
}     // rolling standard deviations a thread keeps in registers (fast path of trig_noise)

// pre-event noise of one trigger (every thread of the workgroup calls it; thread 0's return value is stored)
__device__ inline double trig_noise(const TrigArgs& a, const float* xc, int64_t Mc, int64_t a0, float* scr, double* red,
                                    double* seg1, double* seg2, unsigned* hist, unsigned* sh) {
  const int tid = threadIdx.x, n = a.n;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  double s;
  // median of the rolling std (ddof = 1) of 5 n samples before the
  // trigger, or of up to 7 n samples from it when there is not enough data before
  // (detect.py:466-472)
  int64_t pe0;
  int npe;
  if (a0 > 5 * (int64_t)n) { pe0 = a0 - 5 * (int64_t)n; npe = 5 * n; }
  else { pe0 = a0; npe = (int)min((int64_t)7 * n, Mc - a0); }
  const float* pe = xc + pe0;
  const int count = npe - n + 1;
  double noise = nan;
  if (count >= 1 && count <= a.scap) {
    s = 0.0;
    for (int q = tid; q < npe; q += kTT) s += (double)pe[q];
    const double shift = block_sum(s, red) / npe;       // std is shift-invariant: kills the cancellation
    // prefix sums of y = pe - shift and y^2: per-thread segment sums, scanned by thread 0 .. 511
    const int seglen = (npe + kTT - 1) / kTT;
    const int q0 = min(tid * seglen, npe), q1 = min(q0 + seglen, npe);
    double t1 = 0.0, t2 = 0.0;
    for (int q = q0; q < q1; q += 4) {
      float f[4];
      load4_guarded(pe, q, q1, f);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q + e < q1) {
          const double y = (double)f[e] - shift;
          t1 += y;
          t2 += y * y;
        }
    }
    __syncthreads();
    seg1[tid] = t1;
    seg2[tid] = t2;
    __syncthreads();
    if (tid < 64) {          // exclusive scan of 512 segment sums by one wave, 8 per lane
      double l1[8], l2[8], c1 = 0.0, c2 = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        l1[j] = c1; l2[j] = c2;
        c1 += seg1[8 * tid + j]; c2 += seg2[8 * tid + j];
      }
      double p1 = c1, p2 = c2;     // inclusive scan over lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const double o1 = __shfl_up(p1, off, 64), o2 = __shfl_up(p2, off, 64);
        if (tid >= off) { p1 += o1; p2 += o2; }
      }
      p1 -= c1; p2 -= c2;          // exclusive
#pragma unroll
      for (int j = 0; j < 8; ++j) { seg1[8 * tid + j] = p1 + l1[j]; seg2[8 * tid + j] = p2 + l2[j]; }
    }
    __syncthreads();
    auto prefix = [&](int j, double& o1, double& o2) {      // sums of y, y^2 over [0, j)
      int sg = j / seglen;
      if (sg >= kTT) sg = kTT - 1;
      double b1 = seg1[sg], b2 = seg2[sg];
      for (int q = sg * seglen; q < j; ++q) {
        const double y = (double)pe[q] - shift;
        b1 += y;
        b2 += y * y;
      }
      o1 = b1; o2 = b2;
    };
    const int wper = (count + kTT - 1) / kTT;
    const int i0 = min(tid * wper, count), i1 = min(i0 + wper, count);
    if (wper <= kRegW) {
      // the thread's rolling standard deviations stay in its registers: the selection passes read them nine times,
      // and from the scratch buffer (written by other waves: L2 round trips, one behind the other) those reads
      // were the kernel's longest chain -- 0.3 ms for a station-day's triggers
      unsigned r[kRegW];
      double A1 = 0.0, A2 = 0.0, B1 = 0.0, B2 = 0.0;
      if (i0 < i1) {
        prefix(i0, A1, A2);
        prefix(i0 + n, B1, B2);
      }
#pragma unroll
      for (int g = 0; g < kRegW / 4; ++g) {
        float fa[4], fb[4];
        if (i0 + 4 * g < i1) {                 // (the leaving and the entering samples of four windows, 16 bytes each)
          load4_guarded(pe, i0 + 4 * g, npe, fa);
          load4_guarded(pe, i0 + n + 4 * g, npe, fb);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 4 * g + e, i = i0 + j;
          if (i < i1) {
            const double S1 = B1 - A1, S2 = B2 - A2;
            const double var = (S2 - S1 * S1 / n) / (n - 1.0);
            r[j] = __float_as_uint((float)sqrt(var > 0.0 ? var : 0.0));
            const double ya = (double)fa[e] - shift;
            A1 += ya; A2 += ya * ya;
            if (i + n < npe) {
              const double yb = (double)fb[e] - shift;
              B1 += yb; B2 += yb * yb;
            }
          } else {
            r[j] = 0xffffffffu;
          }
        }
      }
      if (count & 1) {
        noise = (double)select_rank_regs(r, count / 2, hist, sh);
      } else {                         // the mean of the two middle values
        float hi;
        const float lo = select_rank_regs(r, count / 2 - 1, hist, sh, &hi);
        noise = 0.5 * ((double)hi + (double)lo);
      }
      return noise;
    }
    if (i0 < i1) {
      double A1, A2, B1, B2;
      prefix(i0, A1, A2);
      prefix(i0 + n, B1, B2);
      for (int i = i0; i < i1; ++i) {
        const double S1 = B1 - A1, S2 = B2 - A2;
        const double var = (S2 - S1 * S1 / n) / (n - 1.0);
        scr[i] = (float)sqrt(var > 0.0 ? var : 0.0);
        const double ya = (double)pe[i] - shift;
        A1 += ya; A2 += ya * ya;
        if (i + n < npe) {
          const double yb = (double)pe[i + n] - shift;
          B1 += yb; B2 += yb * yb;
        }
      }
    }
    __threadfence_block();
    __syncthreads();
    if (count & 1) {
      noise = (double)select_rank(scr, count, count / 2, hist, sh);
    } else {                         // the mean of the two middle values
      float hi;
      const float lo = select_rank(scr, count, count / 2 - 1, hist, sh, &hi);
      noise = 0.5 * ((double)hi + (double)lo);
    }
  }
  return noise;
}

__global__ __launch_bounds__(kTT) void trig_stats_kernel(const TrigArgs a) {
  __shared__ double red[kTT / 64];
  __shared__ double csh[256];            // projections c_k of the detector's rows (<= 208)
  __shared__ double seg1[kTT], seg2[kTT];
  __shared__ unsigned hist[256];
  __shared__ unsigned sh[2];
  const int tid = threadIdx.x, wv = tid >> 6, ln = tid & 63;
  const int n = a.n;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  float* scr = a.scratch + (int64_t)blockIdx.x * a.scap;
  // blockIdx.y = task: 0 STA / LTA + the window's statistics and projection, 1 the dot products with
  // the training events, 2 the pre-event noise -- three workgroups per trigger instead of one (a
  // station-day has tens of triggers: one workgroup each left most of the chip idle for 0.5 ms)
  const int task = blockIdx.y;
  for (int t = blockIdx.x; t < a.ntrig; t += gridDim.x) {
    const int32_t* tr = a.trig + 5 * (int64_t)t;
    const int chunk = tr[0], det = tr[1], p = tr[2], sta_lo = tr[3], lta_lo = tr[4];
    double* out = a.out + (int64_t)t * (kTrigHead + a.emax);
    const float* xc = a.ctab ? a.x + a.ctab[2 * chunk] : a.x + (int64_t)chunk * a.stride;
    const int64_t Mc = a.ctab ? max((int64_t)0, min(a.M, a.ctab[2 * chunk + 1])) : a.M;   // the chunk's samples
    const int64_t a0 = (int64_t)p * a.Nc;
    const float* w = xc + a0;
    if (task == 1) {
      // ---- dot products with the training events (fast_normcorr(x, ConDat)[0], d1 of 494)
      const int e0 = a.eoff[det], ne = a.eoff[det + 1] - e0;
      for (int e = wv; e < ne; e += kTT / 64) {
        const float* ev = a.ewf + (int64_t)(e0 + e) * n;
        double d = 0.0;
        for (int q = ln; q < n; q += 64) d += (double)ev[q] * (double)w[q];
        d = wave_sum(d);
        if (ln == 0) out[kTrigHead + e] = d;
      }
      continue;
    }
    if (task == 2) {
      const double noise = trig_noise(a, xc, Mc, a0, scr, red, seg1, seg2, hist, sh);
      if (tid == 0) out[5] = noise;
      __syncthreads();
      continue;
    }
    // ---- STA / LTA: means of |DS| over the (edge-adjusted) centred windows
    const float* row = a.ds + ((int64_t)chunk * a.D + det) * a.ldp;
    double st = nan, lt = nan;
    if (sta_lo >= 0) {
      double s = 0.0;
      for (int i = tid; i < a.sta_len; i += kTT) s += fabs((double)row[sta_lo + i]);
      st = block_sum(s, red) / (double)a.sta_len;
    }
    if (lta_lo >= 0) {
      double s = 0.0;
      for (int i = tid; i < a.lta_len; i += kTT) s += fabs((double)row[lta_lo + i]);
      lt = block_sum(s, red) / (double)a.lta_len;
    }
    if (!a.full) {
      if (tid == 0) { out[0] = st; out[1] = lt; }
      continue;
    }
    // ---- the triggering window ConDat = MPcon[p Nc : p Nc + n]           (detect.py:456-457)
    double s = 0.0;
    for (int q = tid; q < n; q += kTT) s += (double)w[q];
    const double mean_c = block_sum(s, red) / n;
    s = 0.0;
    for (int q = tid; q < n; q += kTT) {
      const double d = (double)w[q] - mean_c;
      s += d * d;
    }
    const double std_c = sqrt(block_sum(s, red) / n);                    // np.std: ddof = 0
    // ---- projection into the subspace: c = U w, ssCon = U^T c           (detect.py:460-463)
    const int r0 = a.doff[det], k = a.doff[det + 1] - r0;
    for (int r = wv; r < k; r += kTT / 64) {
      const float* u = a.U + (int64_t)(r0 + r) * n;
      double d = 0.0;
      for (int q = ln; q < n; q += 64) d += (double)u[q] * (double)w[q];
      d = wave_sum(d);
      if (ln == 0) csh[r] = d;
    }
    __syncthreads();
    double s1 = 0.0;
    for (int q = tid; q < n; q += kTT) {
      double v = 0.0;
      for (int r = 0; r < k; ++r) v += csh[r] * (double)a.U[(int64_t)(r0 + r) * n + q];
      s1 += v;
    }
    const double mean_s = block_sum(s1, red) / n;
    double s2 = 0.0;
    for (int q = tid; q < n; q += kTT) {
      double v = 0.0;
      for (int r = 0; r < k; ++r) v += csh[r] * (double)a.U[(int64_t)(r0 + r) * n + q];
      v -= mean_s;
      s2 += v * v;
    }
    const double var_s = block_sum(s2, red) / n;                         // np.var: ddof = 0
    if (tid == 0) {
      out[0] = st; out[1] = lt; out[2] = mean_c; out[3] = std_c; out[4] = var_s;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- fast_normcorr
// detex/construct.py:469-483: Pearson correlation of the template t (n samples) with every
// length-n window of s: out[p] = sum_j (t_j - mean t)(s_{p+j} - mean_p) / (n std t std_p),
// population standard deviations.  One thread per window position, float64 accumulation; the
// template's statistics come in pre-reduced (tstat = {mean t, std t}).  A window of zero
// variance gives the reference's division by zero: +-inf / nan, reproduced by the final divide.
__global__ __launch_bounds__(256) void normcorr_kernel(const float* __restrict__ t, int n,
                                                       const float* __restrict__ s, int64_t nout,
                                                       const double* __restrict__ tstat,
                                                       float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= nout) return;
  const double mt = tstat[0], st = tstat[1];
  const float* w = s + p;
  double s1 = 0.0;
  for (int j = 0; j < n; ++j) s1 += (double)w[j];
  const double mw = s1 / n;
  double num = 0.0, ssw = 0.0;
  for (int j = 0; j < n; ++j) {
    const double d = (double)w[j] - mw;
    num += ((double)t[j] - mt) * d;
    ssw += d * d;
  }
  out[p] = (float)(num / ((double)n * st * sqrt(ssw / n)));
}

__global__ __launch_bounds__(256) void normcorr_tstat_kernel(const float* __restrict__ t, int n,
                                                             double* __restrict__ tstat) {
  __shared__ double red[4];
  auto bsum = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  double a = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) a += (double)t[j];
  const double mt = bsum(a) / n;
  a = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double d = (double)t[j] - mt;
    a += d * d;
  }
  const double var = bsum(a) / n;
  if (threadIdx.x == 0) { tstat[0] = mt; tstat[1] = sqrt(var); }
}

}  // namespace

extern "C" int dtx_normcorr(dtx_ctx* ctx, const float* t_dev, int n, const float* s_dev, int64_t m,
                            float* out_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!t_dev || !s_dev || !out_dev || n <= 0 || m < n)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_normcorr: null pointer, n <= 0 or the series is shorter than the template");
  DtxDeviceGuard g(ctx->device);
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_trig_doff, 64))) return rc;
  double* tstat = (double*)ctx->ws_trig_doff.p;
  const int64_t nout = m - n + 1;
  hipLaunchKernelGGL(normcorr_tstat_kernel, dim3(1), dim3(256), 0, ctx->stream, t_dev, n, tstat);
  hipLaunchKernelGGL(normcorr_kernel, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, ctx->stream,
                     t_dev, n, s_dev, nout, (const double*)tstat, out_dev);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

static int trigger_stats_impl(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                              const int64_t* ctab, const float* ds_dev, int64_t P, int64_t ldp,
                              const int32_t* trig_dev, int ntrig, int sta_len, int lta_len,
                              const float* ewf_dev, const int32_t* ewf_off_dev, int emax,
                              double* out_dev) {
  if (!ctx) return DTX_E_ARG;
  if (ctx->tiles.empty()) return dtx_fail(ctx, DTX_E_STATE, "dtx_trigger_stats: no basis uploaded");
  if (ntrig == 0) return DTX_OK;
  if (!x_dev || !ds_dev || !trig_dev || (ewf_dev && !ewf_off_dev) || !out_dev || ntrig < 0 || emax < 0 ||
      sta_len < 0 || lta_len < 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_trigger_stats: null pointer or negative size");
  if (M < ctx->n || P != dtx_num_windows(M, ctx->n, ctx->Nc))
    return dtx_fail(ctx, DTX_E_ARG, "dtx_trigger_stats: P does not belong to blocks of M samples");
  DtxDeviceGuard g(ctx->device);
  int rc;
  const unsigned grid = (unsigned)std::min(ntrig, 1024);
  const int64_t scap = 6 * (int64_t)ctx->n + 1;
  hipStream_t st = ctx->trig_stream ? ctx->trig_stream : ctx->stream;
  if (!ctx->ev_trig) DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_trig, hipEventDisableTiming));
  if (ctx->trig_pending) {
    // the scratch is shared by consecutive launches: one on another stream than the last waits for it (no host
    // wait); growing the scratch (dtx_reserve waits for ctx->stream only) waits for the last launch itself
    if (ctx->trig_last != st) DTX_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_trig, 0));
    if ((size_t)grid * scap * sizeof(float) > ctx->ws_trig.bytes) DTX_HIP(ctx, hipEventSynchronize(ctx->ev_trig));
  }
  if ((rc = dtx_reserve(ctx, ctx->ws_trig, (size_t)grid * scap * sizeof(float)))) return rc;
  TrigArgs a;
  a.x = x_dev; a.M = M; a.stride = chunk_stride; a.ctab = ctab; a.ds = ds_dev; a.P = P; a.ldp = ldp;
  a.trig = trig_dev; a.ntrig = ntrig; a.sta_len = sta_len; a.lta_len = lta_len;
  a.U = (const float*)ctx->U_buf.p; a.doff = (const int32_t*)ctx->doff_all.p;
  a.ewf = ewf_dev; a.eoff = ewf_off_dev; a.out = out_dev;
  a.scratch = (float*)ctx->ws_trig.p; a.scap = scap;
  a.D = ctx->D; a.n = ctx->n; a.Nc = ctx->Nc; a.emax = emax;
  a.full = ewf_dev != nullptr;
  hipLaunchKernelGGL(trig_stats_kernel, dim3(grid, a.full ? 3u : 1u), dim3(kTT), 0, st, a);
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_trig, st));
  ctx->trig_last = st;
  ctx->trig_pending = true;
  return DTX_OK;
}

extern "C" int dtx_trigger_stats(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                                 const float* ds_dev, int64_t P, int64_t ldp,
                                 const int32_t* trig_dev, int ntrig, int sta_len, int lta_len,
                                 const float* ewf_dev, const int32_t* ewf_off_dev, int emax,
                                 double* out_dev) {
  return trigger_stats_impl(ctx, x_dev, M, chunk_stride, nullptr, ds_dev, P, ldp, trig_dev, ntrig, sta_len,
                            lta_len, ewf_dev, ewf_off_dev, emax, out_dev);
}

extern "C" int dtx_trigger_stats_ragged(dtx_ctx* ctx, const float* x_dev, const int64_t* chunk_tab_dev,
                                        int64_t M, const float* ds_dev, int64_t P, int64_t ldp,
                                        const int32_t* trig_dev, int ntrig, int sta_len, int lta_len,
                                        const float* ewf_dev, const int32_t* ewf_off_dev, int emax,
                                        double* out_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!chunk_tab_dev && ntrig > 0) return dtx_fail(ctx, DTX_E_ARG, "dtx_trigger_stats_ragged: no chunk table");
  return trigger_stats_impl(ctx, x_dev, M, 0, chunk_tab_dev, ds_dev, P, ldp, trig_dev, ntrig, sta_len,
                            lta_len, ewf_dev, ewf_off_dev, emax, out_dev);
}

// ---------------------------------------------------------------- FAS screen
// max_i classic_sta_lta(a, nsta, nlta)[i] of a batch of traces (fas.py:175-205 of the reference screens the random
// chunks of _initFAS with obspy.signal.trigger.classic_sta_lta on the Z channel; its documented form:
// sta[i], lta[i] = trailing sums of a^2 over nsta / nlta samples (clipped at the trace start) / nsta, nlta;
// sta[: nlta - 1] = 0; lta floored at the smallest normal double; ratio sta / lta).  A thread takes kStaG
// consecutive samples: the two window sums at its first sample by direct summation in float64 (nlta loads, the
// anchor), then sliding -- no prefix over the trace, no error growing along it.  Trace k: x[tab[2 k] + j stride],
// j < tab[2 k + 1].  out[k] (float64 bits, pre-zeroed) takes an atomicMax: ratios are >= 0, their bit patterns ordered.
constexpr int kStaG = 64;
__global__ __launch_bounds__(256) void stalta_max_kernel(const float* __restrict__ x, const int64_t* __restrict__ tab,
                                                         int stride, int nsta, int nlta,
                                                         unsigned long long* __restrict__ out) {
  const int k = blockIdx.y;
  const int64_t len = tab[2 * k + 1];
  const float* a = x + tab[2 * k];
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * kStaG;
  double best = 0.0;
  if (i0 < len && i0 + kStaG > nlta - 1) {
    auto sq = [&](int64_t j) { const double v = (double)a[j * stride]; return v * v; };
    double ss = 0.0, sl = 0.0;
    for (int64_t j = max((int64_t)0, i0 - nlta + 1); j <= i0; ++j) {
      const double q = sq(j);
      sl += q;
      if (j > i0 - nsta) ss += q;
    }
    const double rs = 1.0 / nsta, rl = 1.0 / nlta, tiny = 2.2250738585072014e-308;
    const int64_t i1 = min(len, i0 + kStaG);
    for (int64_t i = i0;;) {
      if (i >= nlta - 1) best = fmax(best, (ss * rs) / fmax(sl * rl, tiny));
      if (++i >= i1) break;
      const double q = sq(i);
      ss += q - (i - nsta >= 0 ? sq(i - nsta) : 0.0);
      sl += q - (i - nlta >= 0 ? sq(i - nlta) : 0.0);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = best;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + h]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] > 0.0) atomicMax(out + k, (unsigned long long)__double_as_longlong(red[0]));
}

extern "C" int dtx_stalta_max(dtx_ctx* ctx, const float* x_dev, const int64_t* tab_dev, int ntrace, int64_t maxlen,
                              int stride, int nsta, int nlta, double* out_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!x_dev || !tab_dev || !out_dev || ntrace <= 0 || maxlen <= 0 || stride <= 0 || nsta <= 0 || nlta < nsta)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_stalta_max: bad arguments (need 0 < nsta <= nlta)");
  if (ntrace > 65535) return dtx_fail(ctx, DTX_E_ARG, "dtx_stalta_max: more than 65535 traces");
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipMemsetAsync(out_dev, 0, (size_t)ntrace * sizeof(double), ctx->stream));
  const int64_t gx = (maxlen + 256 * kStaG - 1) / (256 * kStaG);
  hipLaunchKernelGGL(stalta_max_kernel, dim3((unsigned)gx, (unsigned)ntrace), dim3(256), 0, ctx->stream, x_dev, tab_dev,
                     stride, nsta, nlta, (unsigned long long*)out_dev);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}
