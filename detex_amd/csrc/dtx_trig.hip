// Trigger-time estimates on the device, three workgroups per trigger (trig_stats_kernel):
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
  float* scratch;        // [gridDim.x][scap] rolling variances of one trigger (when they do not fit the registers)
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

// the workgroup's sums of four values at once; every thread gets them (red: 4 x 8 doubles of LDS)
__device__ inline void block_sum4(double (&v)[4], double* red) {
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = wave_sum(v[j]);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j * (kTT / 64) + (threadIdx.x >> 6)] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kTT / 64; ++w) t += red[j * (kTT / 64) + w];
    v[j] = t;
  }
}

// k-th smallest (0-based) of count non-negative floats, and the (k + 1)-th if next != NULL (the two middle
// values of an even count): bisection on the bit pattern, one bit a pass -- the pattern keeps a bit when at most k
// values lie below it, so it ends as the largest p with |{u < p}| <= k, which is the k-th smallest value.  Two
// vector instructions per value and pass (compare, add with carry), a wave reduction, one LDS atomic per wave and
// one barrier per pass; a first pass takes the minimum and maximum and the bisection starts below the bits they
// share (rolling variances of stationary noise share 10 .. 12 of their 31).  (Rounds 4-5 took 4 bits a pass and
// counted 16 buckets in registers: 37 vector instructions per value and pass, 28 000 per wave for a trigger's
// 36 001 values -- 0.1 ms of the kernel's 0.2; rounds 1-3 counted 8 bits a pass with LDS
// atomics, which a narrow distribution serialises on one bin.  Counting with the wave's ballot and a scalar
// popcount instead of the add with carry measured 3x slower: each scalar read of the compare's mask waits out
// the vector pipeline.)  Returns the value(s) to every thread.  hist: >= 64 words of LDS.
// each(f): calls f(bit pattern) for every value this thread contributes (any partition of the multiset over the
// workgroup's threads will do); 0xffffffff = no value
template <class Each>
__device__ inline float select_rank_impl(Each each, int k, unsigned* hist, float* next) {
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  unsigned lo1 = 0xffffffffu, hi1 = 0u;              // minimum, maximum + 1 (0xffffffff + 1 wraps to "none")
  each([&](unsigned u) {
    lo1 = min(lo1, u);
    hi1 = max(hi1, u + 1u);
  });
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo1 = min(lo1, (unsigned)__shfl_xor(lo1, off, 64));
    hi1 = max(hi1, (unsigned)__shfl_xor(hi1, off, 64));
  }
  __syncthreads();                         // (earlier readers of hist are done)
  if (threadIdx.x < 32) hist[threadIdx.x] = 0u;
  if (ln == 0) { hist[32 + 2 * wv] = lo1; hist[32 + 2 * wv + 1] = hi1; }
  __syncthreads();
  unsigned umin = 0xffffffffu, umax = 0u;
#pragma unroll
  for (int w = 0; w < kTT / 64; ++w) { umin = min(umin, hist[32 + 2 * w]); umax = max(umax, hist[32 + 2 * w + 1]); }
  umax -= 1u;
  unsigned prefix = umin;
  if (umin != umax) {
    const int hb = 31 - __clz(umin ^ umax);          // the highest bit in which two values differ (<= 30)
    prefix = umin & ~((2u << hb) - 1u);
    for (int bit = hb; bit >= 0; --bit) {
      const unsigned pivot = prefix | (1u << bit);
      unsigned c = 0u;
      each([&](unsigned u) { c += u < pivot; });
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
      if (ln == 0) atomicAdd(&hist[bit], c);
      __syncthreads();
      if ((int)hist[bit] <= k) prefix = pivot;
    }
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
    if (ln == 0) { hist[32 + 2 * wv] = le; hist[32 + 2 * wv + 1] = mn; }
    __syncthreads();
    unsigned tle = 0u, tmn = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < kTT / 64; ++w) { tle += hist[32 + 2 * w]; tmn = min(tmn, hist[32 + 2 * w + 1]); }
    *next = (int)tle > k + 1 ? val : __uint_as_float(tmn);
    __syncthreads();
  }
  return val;
}
// ... of count values in memory
__device__ inline float select_rank(const float* __restrict__ v, int count, int k, unsigned* hist,
                                    float* next = nullptr) {
  return select_rank_impl([&](auto f) {
    for (int i = threadIdx.x; i < count; i += kTT) f(__float_as_uint(v[i]));
  }, k, hist, next);
}
// ... of the values the threads hold in registers: r[j], j < NR, 0xffffffff (never a non-negative float's
// pattern: it sorts behind everything and is never reached) where a thread has none
template <int NR>
__device__ inline float select_rank_regs(const unsigned (&r)[NR], int k, unsigned* hist, float* next = nullptr) {
  return select_rank_impl([&](auto f) {
#pragma unroll
    for (int j = 0; j < NR; ++j) f(r[j]);
  }, k, hist, next);
}
constexpr int kRegW = 96;      // rolling variances a thread keeps in registers (fast path of trig_noise)
constexpr int kTile = 8 * kTT; // windows of one tile of trig_noise: eight consecutive ones per thread
constexpr int kWQ = 6;         // 16-byte groups of a length-n window a thread keeps (fast paths: n <= 4 kTT kWQ = 12288)
// four consecutive floats at any 4-byte aligned address in ONE load (global memory takes dword-aligned
// multi-dword loads)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void load4_guarded(const float* __restrict__ p, int idx, int limit, float (&o)[4]) {
  const float* q = p + idx;
  const int m = limit - idx;                 // samples left from idx on
  if (m >= 4) {
    const f4u v = *reinterpret_cast<const f4u*>(q);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = e < m ? q[e] : 0.f;
  }
}

// ---- a length-n window held by the workgroup, n <= 4 kTT kWQ: thread t keeps the 16-byte groups t, t + kTT, ..
// (coalesced loads, ALL of them in flight at once) and one of the n % 4 last samples; 0 beyond n.
// Why registers and few, wide rounds of loads: the kernel runs beside the next scan's preparation -- kernels that
// stream at the memory system's full rate -- and a round trip to memory takes several us there.  Its duration
// in place is (dependent rounds of loads) x (that latency), not its arithmetic: rounds 1-5 walked their data
// in some 50 dependent rounds a trigger and took 0.5 .. 0.7 ms in place against 0.2 ms alone.
struct WinRegs {
  f4u v[kWQ];
  float tail;
};
__device__ inline void load_window(const float* __restrict__ w, int n, int tid, WinRegs& r) {
  const int n4 = n & ~3;
#pragma unroll
  for (int j = 0; j < kWQ; ++j) {
    const int q = 4 * (tid + kTT * j);
    r.v[j] = q < n4 ? *reinterpret_cast<const f4u*>(w + q) : f4u{0.f, 0.f, 0.f, 0.f};
  }
  r.tail = n4 + tid < n ? w[n4 + tid] : 0.f;
}
// sums of d = w - w0 and d^2 over the window's n samples (this thread's share)
__device__ inline void window_moments(const WinRegs& r, int n, int tid, double w0, double& s1, double& s2) {
  const int n4 = n & ~3;
#pragma unroll
  for (int j = 0; j < kWQ; ++j) {
    if (4 * (tid + kTT * j) < n4) {
      const double d0 = (double)r.v[j].x - w0, d1 = (double)r.v[j].y - w0, d2 = (double)r.v[j].z - w0,
                   d3 = (double)r.v[j].w - w0;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  if (n4 + tid < n) {
    const double d = (double)r.tail - w0;
    s1 += d;
    s2 += d * d;
  }
}
// dot products of the held window with k rows of n floats (and the rows' sums if SUM): the workgroup takes four
// rows a round -- 4 kWQ loads in flight per thread --, reduces them over its waves and hands row r's results to
// emit(r, dot, sum), called by thread r % 32 once for every row.  part: 2 x 32 x 8 doubles of LDS
template <bool SUM, class Emit>
__device__ inline void block_rows_dot(const float* __restrict__ rows, int k, int n, int tid, const WinRegs& w,
                                      double* part, Emit emit) {
  const int wv = tid >> 6, ln = tid & 63, n4 = n & ~3;
  for (int r0 = 0; r0 < k; r0 += 32) {
    const int kc = min(32, k - r0);
    for (int rb = 0; rb < kc; rb += 4) {
      double acc[4], su[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* u = rows + (int64_t)(r0 + min(rb + i, kc - 1)) * n;   // (past the last row: it again, dropped)
        double d = 0.0, s = 0.0;
#pragma unroll
        for (int j = 0; j < kWQ; ++j) {
          const int q = 4 * (tid + kTT * j);
          if (q < n4) {
            const f4u a = *reinterpret_cast<const f4u*>(u + q);
            d += ((double)a.x * (double)w.v[j].x + (double)a.y * (double)w.v[j].y) +
                 ((double)a.z * (double)w.v[j].z + (double)a.w * (double)w.v[j].w);
            if (SUM) s += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
          }
        }
        if (n4 + tid < n) {
          const float a = u[n4 + tid];
          d += (double)a * (double)w.tail;
          if (SUM) s += (double)a;
        }
        acc[i] = d;
        su[i] = s;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = wave_sum(acc[i]);
        if (SUM) su[i] = wave_sum(su[i]);
        if (ln == 0 && rb + i < kc) {
          part[(rb + i) * 8 + wv] = acc[i];
          if (SUM) part[256 + (rb + i) * 8 + wv] = su[i];
        }
      }
    }
    __syncthreads();
    if (tid < kc) {
      double d = 0.0, s = 0.0;
#pragma unroll
      for (int x = 0; x < kTT / 64; ++x) {
        d += part[tid * 8 + x];
        if (SUM) s += part[256 + tid * 8 + x];
      }
      emit(r0 + tid, d, s);
    }
    __syncthreads();
  }
}

// dot product of two rows of n floats by one wave (every lane gets it), any n: 16-byte loads, float64
// accumulation, four pairs of loads in flight per lane.  SUM: *usum = the sum of u's n values as well
template <bool SUM = false>
__device__ inline double wave_dot(const float* __restrict__ u, const float* __restrict__ w, int n,
                                  double* usum = nullptr) {
  const int ln = threadIdx.x & 63;
  double d[4] = {0.0, 0.0, 0.0, 0.0}, su = 0.0;
  const int n4 = n & ~3;
#pragma unroll 4
  for (int q = 4 * ln; q < n4; q += 256) {
    const f4u a = *reinterpret_cast<const f4u*>(u + q), b = *reinterpret_cast<const f4u*>(w + q);
    d[0] += (double)a.x * (double)b.x;
    d[1] += (double)a.y * (double)b.y;
    d[2] += (double)a.z * (double)b.z;
    d[3] += (double)a.w * (double)b.w;
    if (SUM) su += ((double)a.x + (double)a.y) + ((double)a.z + (double)a.w);
  }
  if (n4 + ln < n) {
    d[0] += (double)u[n4 + ln] * (double)w[n4 + ln];
    if (SUM) su += (double)u[n4 + ln];
  }
  if (SUM) *usum = wave_sum(su);
  return wave_sum((d[0] + d[1]) + (d[2] + d[3]));
}

// LDS of trig_noise: a tile's leaving and entering samples, the waves' totals of its scan
struct NoiseLds {
  float la[kTile], lb[kTile];
  double wtot[2 * (kTT / 64)];
};

// pre-event noise of one trigger (every thread of the workgroup calls it; thread 0's return value is stored):
// the median of the rolling std (ddof = 1) of 5 n samples before the trigger, or of up to 7 n samples from it
// when there is not enough data before (detect.py:466-472).
// The sums of y = pe - shift and y^2 over window i are those of window 0 plus the running sums of what ENTERS
// (sample i' + n) minus what LEAVES (sample i') over i' < i.  The windows go by in tiles of 8 kTT: the workgroup
// loads a tile's leaving and entering samples with coalesced 16-byte loads (the next tile's are in flight while
// this one is worked on), turns them through LDS so that thread t holds those of ITS eight consecutive windows,
// scans the threads' eight-sample totals over the workgroup and walks its eight windows.  shift = the mean of
// window 0 (the std is shift-invariant: it kills the cancellation); window 0 itself is summed from registers.
// (Rounds 1-5 gave a thread one long run of ~70 windows: every load instruction of a wave was 64 separate cache
// lines, and the sums up to a run's start took a first walk over the same samples.)  The selection runs on the
// VARIANCES rounded to float (the square root is monotone and is taken of the median alone, in float64).
__device__ inline double trig_noise(const TrigArgs& a, const float* xc, int64_t Mc, int64_t a0, float* scr, double* red,
                                    NoiseLds& L, unsigned* hist, int tid) {
  const int wv = tid >> 6, ln = tid & 63, n = a.n;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  int64_t pe0;
  int npe;
  if (a0 > 5 * (int64_t)n) { pe0 = a0 - 5 * (int64_t)n; npe = 5 * n; }
  else { pe0 = a0; npe = (int)min((int64_t)7 * n, Mc - a0); }
  const float* pe = xc + pe0;
  const int count = npe - n + 1;
  if (count < 1 || count > a.scap) return nan;
  const int NS = (count + kTile - 1) / kTile;          // tiles
  const bool in_regs = 8 * NS <= kRegW && n <= 4 * kTT * kWQ;
  // ---- window 0: shift = its mean, then its sums of y and y^2
  double shift, base1 = 0.0, base2 = 0.0;
  if (in_regs) {
    WinRegs f;
    load_window(pe, n, tid, f);
    double s1 = 0.0, s2 = 0.0;
    window_moments(f, n, tid, 0.0, s1, s2);
    shift = block_sum(s1, red) / n;
    window_moments(f, n, tid, shift, base1, base2);
  } else {
    double s = 0.0;
#pragma unroll 4
    for (int q = tid; q < n; q += kTT) s += (double)pe[q];
    shift = block_sum(s, red) / n;
#pragma unroll 4
    for (int q = tid; q < n; q += kTT) {
      const double y = (double)pe[q] - shift;
      base1 += y;
      base2 += y * y;
    }
  }
  base1 = block_sum(base1, red);
  base2 = block_sum(base2, red);
  const double inv_n = 1.0 / n, inv_n1 = 1.0 / (n - 1.0);
  // ---- the tiles
  float ga[2][4], gb[2][4];                            // a tile's samples on their way from memory
  auto fetch = [&](int s) {
    const int b = kTile * s + 4 * tid;
    load4_guarded(pe, b, npe, ga[0]);
    load4_guarded(pe, b + kTile / 2, npe, ga[1]);
    load4_guarded(pe, b + n, npe, gb[0]);
    load4_guarded(pe, b + n + kTile / 2, npe, gb[1]);
  };
  auto step = [&](int s, auto store) {                 // store(e, i, bits): window i = the thread's e-th of tile s
    __syncthreads();                                   // (the last tile's readers are done)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<float4*>(&L.la[h * (kTile / 2) + 4 * tid]) = make_float4(ga[h][0], ga[h][1], ga[h][2], ga[h][3]);
      *reinterpret_cast<float4*>(&L.lb[h * (kTile / 2) + 4 * tid]) = make_float4(gb[h][0], gb[h][1], gb[h][2], gb[h][3]);
    }
    __syncthreads();
    if (s + 1 < NS) fetch(s + 1);
    // what enters minus what leaves, of y and of y^2 = (yb - ya) (yb + ya): the shift drops out of the first
    float xa[8], xb[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 fa = *reinterpret_cast<const float4*>(&L.la[8 * tid + 4 * h]);
      const float4 fb = *reinterpret_cast<const float4*>(&L.lb[8 * tid + 4 * h]);
      xa[4 * h] = fa.x; xa[4 * h + 1] = fa.y; xa[4 * h + 2] = fa.z; xa[4 * h + 3] = fa.w;
      xb[4 * h] = fb.x; xb[4 * h + 1] = fb.y; xb[4 * h + 2] = fb.z; xb[4 * h + 3] = fb.w;
    }
    const int i0 = kTile * s + 8 * tid;
    const double shift2 = 2.0 * shift;
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (i0 + e + n >= npe) xb[e] = xa[e];            // (no sample enters past the last window: nothing it feeds exists)
      const double da = (double)xb[e] - (double)xa[e];
      d1 += da;
      d2 += da * (((double)xb[e] + (double)xa[e]) - shift2);
    }
    // exclusive scan of (d1, d2) over the workgroup: within the wave, then the waves' totals through LDS
    double i1 = d1, i2 = d2;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o1 = __shfl_up(i1, off, 64), o2 = __shfl_up(i2, off, 64);
      if (ln >= off) { i1 += o1; i2 += o2; }
    }
    if (ln == 63) { L.wtot[2 * wv] = i1; L.wtot[2 * wv + 1] = i2; }
    __syncthreads();
    double S1 = base1 + (i1 - d1), S2 = base2 + (i2 - d2);
#pragma unroll
    for (int x = 0; x < kTT / 64; ++x) {
      const double t1 = L.wtot[2 * x], t2 = L.wtot[2 * x + 1];
      if (x < wv) { S1 += t1; S2 += t2; }
      base1 += t1;
      base2 += t2;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double var = (S2 - S1 * S1 * inv_n) * inv_n1;
      store(e, i0 + e, i0 + e < count ? __float_as_uint((float)(var > 0.0 ? var : 0.0)) : 0xffffffffu);
      const double da = (double)xb[e] - (double)xa[e];
      S1 += da;
      S2 += da * (((double)xb[e] + (double)xa[e]) - shift2);
    }
  };
  fetch(0);
  float lo, hi;
  if (in_regs) {
    // the thread's rolling variances stay in its registers: the selection passes read them some 25 times, and
    // from the scratch buffer (written by other waves: L2 round trips, one behind the other) those reads were
    // the kernel's longest chain
    unsigned r[kRegW];
#pragma unroll
    for (int s = 0; s < kRegW / 8; ++s) {
      if (s < NS) {
        step(s, [&](int e, int, unsigned bits) { r[8 * s + e] = bits; });
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[8 * s + e] = 0xffffffffu;
      }
    }
    if (count & 1) lo = hi = select_rank_regs(r, count / 2, hist);
    else lo = select_rank_regs(r, count / 2 - 1, hist, &hi);
  } else {
    for (int s = 0; s < NS; ++s)
      step(s, [&](int, int i, unsigned bits) { if (i < count) scr[i] = __uint_as_float(bits); });
    __threadfence_block();
    __syncthreads();
    if (count & 1) lo = hi = select_rank(scr, count, count / 2, hist);
    else lo = select_rank(scr, count, count / 2 - 1, hist, &hi);
  }
  return 0.5 * (sqrt((double)lo) + sqrt((double)hi));      // (an even count: the mean of the two middle values)
}

__global__ __launch_bounds__(kTT) void trig_stats_kernel(const TrigArgs a) {
  __shared__ double red[4 * (kTT / 64)];
  __shared__ double csh[256], csum[256];   // projections c_k of the detector's rows (<= 208) and the rows' sums
  __shared__ double part[2 * 32 * 8];
  __shared__ NoiseLds nl;
  __shared__ unsigned hist[64];
  const int n = a.n, n4 = n & ~3;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const bool held = n <= 4 * kTT * kWQ;    // the triggering window fits the threads' registers
  float* scr = a.scratch + (int64_t)blockIdx.x * a.scap;
  // blockIdx.y = task: 0 STA / LTA + the window's statistics and projection, 1 the dot products with
  // the training events, 2 the pre-event noise -- three workgroups per trigger instead of one (a
  // station-day has tens of triggers: one workgroup each left most of the chip idle for 0.5 ms)
  const int task = blockIdx.y;
  for (int t = blockIdx.x; t < a.ntrig; t += gridDim.x) {
    // (the thread index made opaque per trigger: the compiler otherwise computes every address offset and
    //  mask of the three tasks' unrolled loads once, ahead of this loop, and spills a hundred of them)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int wv = tid >> 6, ln = tid & 63;
    const int32_t* tr = a.trig + 5 * (int64_t)t;
    const int chunk = tr[0], det = tr[1], p = tr[2], sta_lo = tr[3], lta_lo = tr[4];
    double* out = a.out + (int64_t)t * (kTrigHead + a.emax);
    const float* xc = a.ctab ? a.x + a.ctab[2 * chunk] : a.x + (int64_t)chunk * a.stride;
    const int64_t Mc = a.ctab ? max((int64_t)0, min(a.M, a.ctab[2 * chunk + 1])) : a.M;   // the chunk's samples
    const int64_t a0 = (int64_t)p * a.Nc;
    const float* w = xc + a0;
    if (task == 2) {
      const double noise = trig_noise(a, xc, Mc, a0, scr, red, nl, hist, tid);
      if (tid == 0) out[5] = noise;
      __syncthreads();
      continue;
    }
    WinRegs wr;
    if (a.full && held) load_window(w, n, tid, wr);
    if (task == 1) {
      // ---- dot products with the training events (fast_normcorr(x, ConDat)[0], d1 of 494)
      const int e0 = a.eoff[det], ne = a.eoff[det + 1] - e0;
      if (held) {
        block_rows_dot<false>(a.ewf + (int64_t)e0 * n, ne, n, tid, wr, part,
                              [&](int e, double d, double) { out[kTrigHead + e] = d; });
      } else {
        for (int e = wv; e < ne; e += kTT / 64) {
          const double d = wave_dot(a.ewf + (int64_t)(e0 + e) * n, w, n);
          if (ln == 0) out[kTrigHead + e] = d;
        }
      }
      continue;
    }
    // ---- STA / LTA: means of |DS| over the (edge-adjusted) centred windows; the triggering window ConDat =
    // MPcon[p Nc : p Nc + n] (detect.py:456-457): its mean and std (np.std: ddof = 0) from the sums of d = w - w0
    // and d^2, w0 = the mean of its first samples (one round of loads and ONE reduction for the four sums: they
    // were four loops, each behind the barrier of the one before)
    const float* row = a.ds + ((int64_t)chunk * a.D + det) * a.ldp;
    double ps[4] = {0.0, 0.0, 0.0, 0.0};
    if (sta_lo >= 0) {
#pragma unroll 2
      for (int i = tid; i < a.sta_len; i += kTT) ps[0] += fabs((double)row[sta_lo + i]);
    }
    if (lta_lo >= 0) {
#pragma unroll 4
      for (int i = tid; i < a.lta_len; i += kTT) ps[1] += fabs((double)row[lta_lo + i]);
    }
    double w0 = 0.0;
    if (a.full) {
      const int nh = min(n, 64);
      w0 = wave_sum(ln < nh ? (double)w[ln] : 0.0) / nh;
      if (held) {
        window_moments(wr, n, tid, w0, ps[2], ps[3]);
      } else {
#pragma unroll 4
        for (int q = tid; q < n; q += kTT) {
          const double d = (double)w[q] - w0;
          ps[2] += d;
          ps[3] += d * d;
        }
      }
    }
    block_sum4(ps, red);
    const double st = sta_lo >= 0 ? ps[0] / (double)a.sta_len : nan;
    const double lt = lta_lo >= 0 ? ps[1] / (double)a.lta_len : nan;
    if (!a.full) {
      if (tid == 0) { out[0] = st; out[1] = lt; }
      continue;
    }
    const double md = ps[2] / n, mean_c = w0 + md;
    const double std_c = sqrt(fmax(ps[3] / n - md * md, 0.0));
    // ---- projection into the subspace: c = U w, ssCon = U^T c           (detect.py:460-463); its mean is
    // sum_r c_r (sum of row r) / n -- the row sums ride on the loads of the dot products
    const int r0 = a.doff[det], k = a.doff[det + 1] - r0;
    const float* Ud = a.U + (int64_t)r0 * n;
    if (held) {
      block_rows_dot<true>(Ud, k, n, tid, wr, part, [&](int r, double d, double s) { csh[r] = d; csum[r] = s; });
    } else {
      for (int r = wv; r < k; r += kTT / 64) {
        double su;
        const double d = wave_dot<true>(Ud + (int64_t)r * n, w, n, &su);
        if (ln == 0) { csh[r] = d; csum[r] = su; }
      }
      __syncthreads();
    }
    double mean_s = 0.0;
    for (int r = 0; r < k; ++r) mean_s += csh[r] * csum[r];
    mean_s /= n;
    double s2 = 0.0;
    if (held) {
      // ssCon at the thread's samples, two rows of U a round (2 kWQ loads in flight)
      double v[4 * kWQ], vt = 0.0;
#pragma unroll
      for (int j = 0; j < 4 * kWQ; ++j) v[j] = 0.0;
      for (int r = 0; r < k; r += 2) {
        const double c0 = csh[r], c1 = r + 1 < k ? csh[r + 1] : 0.0;
        const float* u0 = Ud + (int64_t)r * n;
        const float* u1 = Ud + (int64_t)min(r + 1, k - 1) * n;
#pragma unroll
        for (int j = 0; j < kWQ; ++j) {
          const int q = 4 * (tid + kTT * j);
          if (q < n4) {
            const f4u x0 = *reinterpret_cast<const f4u*>(u0 + q), x1 = *reinterpret_cast<const f4u*>(u1 + q);
            v[4 * j] += c0 * (double)x0.x + c1 * (double)x1.x;
            v[4 * j + 1] += c0 * (double)x0.y + c1 * (double)x1.y;
            v[4 * j + 2] += c0 * (double)x0.z + c1 * (double)x1.z;
            v[4 * j + 3] += c0 * (double)x0.w + c1 * (double)x1.w;
          }
        }
        if (n4 + tid < n) vt += c0 * (double)u0[n4 + tid] + c1 * (double)u1[n4 + tid];
      }
#pragma unroll
      for (int j = 0; j < kWQ; ++j) {
        if (4 * (tid + kTT * j) < n4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double d = v[4 * j + e] - mean_s;
            s2 += d * d;
          }
        }
      }
      if (n4 + tid < n) s2 += (vt - mean_s) * (vt - mean_s);
    } else {
#pragma unroll 2
      for (int q = tid; q < n; q += kTT) {
        double v = 0.0;
#pragma unroll 4
        for (int r = 0; r < k; ++r) v += csh[r] * (double)Ud[(int64_t)r * n + q];
        v -= mean_s;
        s2 += v * v;
      }
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
