// Pre-processing of continuous / template data on the device: the numerical part
// of the reference's _applyFilter (detex/construct.py:990-1030): st.detrend('linear'),
// st.filter('bandpass', freqmin, freqmax, corners, zerophase) and the channel
// multiplexing of construct.py:928-987.  ObsPy (third-party, not vendored)
// delegates these to scipy.signal.detrend(type='linear') and to
// sosfilt(zpk2sos(iirfilter(..., 'butter', output='zpk'))) run forward and, for
// zerophase, once more on the reversed output; that arithmetic is restated here
// in float64.  Sorting / merging / trimming of ObsPy streams stays on the host.
//
// An IIR filter is a linear recurrence along time; it is parallelised by
// blocks (memory access coalesced through an LDS tile, see pp_tiled_kernel): (1) every block of kPB samples is run from a zero state and its final
// state kept, (2) the true state entering each block follows from
// S_b = M S_{b-1} + f_{b-1} with M the state transition over one block (a tiny
// sequential pass per series), (3) every block is run again from its true state
// and written out.  Series (chunk x channel) and blocks are independent.
#include <algorithm>

#include "dtx_internal.h"

namespace {

constexpr int kMaxSec = 8;    // biquad sections (bandpass corners <= 8)
constexpr int kPB = 256;   // samples per block (short blocks: more workgroups in flight)
constexpr int kLR = kPB / 64;     // samples per lane of a block in pp_local_kernel
constexpr int kTrendSlices = 16;

struct Sos {
  double b0[kMaxSec], b1[kMaxSec], b2[kMaxSec], a1[kMaxSec], a2[kMaxSec];
  int nsec;
};

// scipy.signal.sosfilt step (direct form II transposed), all sections
__device__ inline double sos_step(const Sos& f, double x, double* z) {
  for (int s = 0; s < f.nsec; ++s) {
    const double y = f.b0[s] * x + z[2 * s];
    z[2 * s] = f.b1[s] * x - f.a1[s] * y + z[2 * s + 1];
    z[2 * s + 1] = f.b2[s] * x - f.a2[s] * y;
    x = y;
  }
  return x;
}

// the same with the section count at compile time: with a runtime count the loop is unrolled to the eight
// sections the state array holds, under predication (a two-section band-pass paid for eight)
template <int NSEC>
__device__ __forceinline__ double sos_step_t(const Sos& f, double x, double* z) {
#pragma unroll
  for (int s = 0; s < NSEC; ++s) {
    const double y = f.b0[s] * x + z[2 * s];
    z[2 * s] = f.b1[s] * x - f.a1[s] * y + z[2 * s + 1];
    z[2 * s + 1] = f.b2[s] * x - f.a2[s] * y;
    x = y;
  }
  return x;
}

// Where series `ser` -- channel ser % Nc of chunk ser / Nc -- starts in the input: chunks `chunk` samples apart, a
// chunk's channels `chan` apart.  A packed [nchunk][Nc][T'] array has chan = T', chunk = Nc T'; chunks cut from one
// [Nc][Tday] record (overlapping or not, dtx_preprocess_planes) have chan = Tday and chunk = their spacing.
struct SerLayout {
  int64_t chunk, chan;
  int Nc;
  __host__ __device__ int64_t off(int ser) const { return (int64_t)(ser / Nc) * chunk + (int64_t)(ser % Nc) * chan; }
};
inline SerLayout ser_uniform(int64_t stride) { return SerLayout{stride, 0, 1}; }   // series s at s * stride

// per series: sums for the least-squares line a + b t (scipy.signal.detrend
// type='linear'); kTrendSlices workgroups per series add their partial sums of
// x and x (t - tc) with float64 atomics; trend_of() turns them into (a, b)
__global__ __launch_bounds__(256) void pp_trend_kernel(const float* __restrict__ raw, int64_t T,
                                                       SerLayout sl,
                                                       double* __restrict__ tsum) {
  __shared__ double sh[8];
  const int ser = blockIdx.y, tid = threadIdx.x;
  const float* x = raw + sl.off(ser);
  const double tc = 0.5 * (double)(T - 1);   // centred abscissa: diagonal normal equations
  const int64_t per = (T + kTrendSlices - 1) / kTrendSlices;
  const int64_t t0 = (int64_t)blockIdx.x * per, t1 = min(t0 + per, T);
  double s0 = 0.0, s1 = 0.0;
  for (int64_t t = t0 + tid; t < t1; t += 256) {
    const double v = (double)x[t];
    s0 += v;
    s1 += v * ((double)t - tc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s1 += __shfl_down(s1, off, 64);
  }
  if ((tid & 63) == 0) { sh[tid >> 6] = s0; sh[4 + (tid >> 6)] = s1; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&tsum[2 * ser], sh[0] + sh[1] + sh[2] + sh[3]);
    atomicAdd(&tsum[2 * ser + 1], sh[4] + sh[5] + sh[6] + sh[7]);
  }
}

// (intercept at t = 0, slope) from the sums
__device__ inline void trend_of(const double* tsum, int64_t T, double* a, double* b) {
  const double n = (double)T, tc = 0.5 * (double)(T - 1);
  const double suu = n * (n * n - 1.0) / 12.0;            // sum (t - tc)^2
  *b = (T > 1) ? tsum[1] / suu : 0.0;
  *a = tsum[0] / n - *b * tc;
}

// input sample i (in filter order) of a series
template <bool FIRST>
__device__ inline double pp_input(const float* raw, const double* mid, const double* trend,
                                  int64_t T, int64_t i, bool rev) {
  const int64_t t = rev ? (T - 1 - i) : i;
  if (FIRST) return (double)raw[t] - (trend[0] + trend[1] * (double)t);
  return mid[t];
}

// passes 1 and 3: a workgroup owns 256 consecutive blocks of one series (thread =
// block).  Samples move through an LDS tile [256 blocks][32 samples]: loaded and
// stored cooperatively, 32 consecutive samples of a block per 16 lanes (full
// 128-byte lines of the float32 input / 256 bytes of the float64 intermediate),
// filtered by the owning thread from LDS.
//   RUN = false: start from a zero state, keep the final state (pass 1)
//   RUN = true : start from the true entry state, write the output (pass 3);
//                LAST selects the float32 result: multiplexed x[Nc t + c] (dec <= 1) or
//                the series' own plane keeping every dec-th sample (dtx_decimate)
constexpr int kSub = 32;
// NSEC (the section count) is a template parameter and the coefficients come from device memory into per-thread
// registers: as a by-value struct of eight-section arrays behind a switch they lived in scalar registers that did
// not fit -- every use inside the sample loop re-read a 16-register tuple from a spill lane by lane (~100
// v_readlane per sample: the filter run was 99 of the forward pass's 167 us per station-day).
template <bool FIRST, bool RUN, bool LAST, int NSEC>
__global__ __launch_bounds__(256) void pp_tiled_kernel(
    const float* __restrict__ raw, SerLayout sl, const double* __restrict__ mid,
    const double* __restrict__ trend, int64_t T, int nblk, const double* __restrict__ coef, bool rev,
    double* __restrict__ fstate, double* __restrict__ out_mid, float* __restrict__ out_mp, int Nc,
    int64_t mp_stride, int dec) {
  __shared__ double tile[256][kSub + 1];
  const int tid = threadIdx.x;
  double cb0[NSEC], cb1[NSEC], cb2[NSEC], ca1[NSEC], ca2[NSEC];
  {
    int opaque;                                   // (keeps the loads per lane: vector registers, not scalar ones)
    asm volatile("v_mov_b32 %0, 0" : "=v"(opaque));
    const double* cv = coef + opaque;
#pragma unroll
    for (int sct = 0; sct < NSEC; ++sct) {
      cb0[sct] = cv[sct]; cb1[sct] = cv[kMaxSec + sct]; cb2[sct] = cv[2 * kMaxSec + sct];
      ca1[sct] = cv[3 * kMaxSec + sct]; ca2[sct] = cv[4 * kMaxSec + sct];
    }
  }
  const int ser = blockIdx.y;
  const int g0 = blockIdx.x * 256;
  const int g = g0 + tid;
  const bool valid = g < nblk;
  const float* r = raw ? raw + sl.off(ser) : nullptr;
  const double* m = mid ? mid + (int64_t)ser * T : nullptr;
  double trc[2] = {0.0, 0.0};
  if (FIRST && trend) trend_of(trend + 2 * ser, T, &trc[0], &trc[1]);
  const double* tr = trc;
  const int chunk = ser / Nc, c = ser - chunk * Nc;
  double z[2 * kMaxSec];
  double* st = fstate + ((int64_t)ser * nblk + (valid ? g : 0)) * (2 * kMaxSec);
#pragma unroll
  for (int k = 0; k < 2 * NSEC; ++k) z[k] = (RUN && valid) ? st[k] : 0.0;

  // The sub-tile's samples are fetched a sub-tile AHEAD, into registers (raw float32 / the float64 intermediate as
  // loaded; trend and conversion when they go to the tile): the loads of sub-tile jj + 1 fly under the filter run
  // and the stores of sub-tile jj.  (Fetched at the top of their own sub-tile, sixteen at a time, their two
  // latencies stood in front of every run: 110 + 134 us for the two runs.)  Loads at a clamped index, then a
  // select: as "in range ? load : 0" every load sat in a branch of its own with its wait inside.
  typedef typename std::conditional<FIRST, float, double>::type tin_t;
  tin_t pre[kSub];
  auto fetch = [&](int jj) {
#pragma unroll
    for (int q = 0; q < kSub; ++q) {
      const int b = q * (256 / kSub) + tid / kSub, j = tid % kSub;
      const int64_t i = min((int64_t)(g0 + b) * kPB + jj + j, T - 1);
      const int64_t t = rev ? (T - 1 - i) : i;
      if constexpr (FIRST) pre[q] = r[t]; else pre[q] = m[t];
    }
  };
  fetch(0);
  for (int jj = 0; jj < kPB; jj += kSub) {
#pragma unroll
    for (int q = 0; q < kSub; ++q) {
      const int b = q * (256 / kSub) + tid / kSub, j = tid % kSub;
      const int64_t i = (int64_t)(g0 + b) * kPB + jj + j;
      double val = (double)pre[q];
      if (FIRST) {
        const int64_t t = rev ? (T - 1 - i) : i;
        val -= tr[0] + tr[1] * (double)t;
      }
      tile[b][j] = (g0 + b < nblk && i < T) ? val : 0.0;
    }
    if (jj + kSub < kPB) fetch(jj + kSub);
    __syncthreads();
    if (valid) {
      const int64_t i0 = (int64_t)g * kPB + jj;
#pragma unroll 4
      for (int j = 0; j < kSub; ++j) {
        if (i0 + j < T) {
          double x = tile[tid][j];
#pragma unroll
          for (int sct = 0; sct < NSEC; ++sct) {         // scipy.signal.sosfilt's step (direct form II transposed)
            const double y = cb0[sct] * x + z[2 * sct];
            z[2 * sct] = cb1[sct] * x - ca1[sct] * y + z[2 * sct + 1];
            z[2 * sct + 1] = cb2[sct] * x - ca2[sct] * y;
            x = y;
          }
          if (RUN) tile[tid][j] = x;
        }
      }
    }
    if (RUN) {
      __syncthreads();
#pragma unroll 4
      for (int mm = 0; mm < kSub; ++mm) {
        const int idx = mm * 256 + tid;
        const int b = idx / kSub, j = idx - b * kSub;
        const int64_t i = (int64_t)(g0 + b) * kPB + jj + j;
        const double yv = tile[b][j];
        if (g0 + b < nblk && i < T) {
          const int64_t t = rev ? (T - 1 - i) : i;
          if (LAST) {
            if (dec <= 1) out_mp[(int64_t)chunk * mp_stride + t * Nc + c] = (float)yv;
            else if (t % dec == 0) out_mp[(int64_t)ser * mp_stride + t / dec] = (float)yv;
          } else {
            out_mid[(int64_t)ser * T + t] = yv;
          }
        }
      }
    }
    __syncthreads();
  }
  if (!RUN && valid) {
#pragma unroll
    for (int k = 0; k < 2 * NSEC; ++k) st[k] = z[k];
  }
}

// pass 2: true entry state of every block, S_b = M S_{b-1} + f_{b-1}, by the same
// three-phase idea one level up.  One wave per series, lane l owns `seg`
// consecutive blocks: (A) walk them from a zero state (entry states relative to
// the segment start, in place) and keep the segment's final state; (B) chain the
// 64 segment states with Mseg = M^seg through shuffles; (C) add the propagated
// segment entry state h (h <- M h per block) to the relative entry states.
__device__ inline void matvec(const double* Ms, int ns, const double* v, double* out) {
  for (int r = 0; r < ns; ++r) {
    double acc = 0.0;
    for (int c = 0; c < ns; ++c) acc += Ms[r * ns + c] * v[c];
    out[r] = acc;
  }
}

__global__ __launch_bounds__(64) void pp_chain_kernel(double* __restrict__ fstate,
                                                      const double* __restrict__ M,
                                                      const double* __restrict__ Mseg, int nblk,
                                                      int seg, int ns) {
  __shared__ double Ms[4 * kMaxSec * kMaxSec], Mg[4 * kMaxSec * kMaxSec];
  const int ser = blockIdx.x, lane = threadIdx.x;
  for (int k = lane; k < ns * ns; k += 64) { Ms[k] = M[k]; Mg[k] = Mseg[k]; }
  __syncthreads();
  double* fs = fstate + (int64_t)ser * nblk * (2 * kMaxSec);
  const int b0 = lane * seg, b1 = min(b0 + seg, nblk);
  double S[2 * kMaxSec], tmp[2 * kMaxSec];
  for (int k = 0; k < ns; ++k) S[k] = 0.0;
  for (int b = b0; b < b1; ++b) {                              // (A)
    double* p = fs + (int64_t)b * (2 * kMaxSec);
    matvec(Ms, ns, S, tmp);
    for (int k = 0; k < ns; ++k) { const double fin = p[k]; p[k] = S[k]; S[k] = tmp[k] + fin; }
  }
  // a lane without blocks has S = 0 and must still pass its entry state on: its own
  // transition is the identity over zero blocks, so only lanes with a full segment apply Mg
  double G[2 * kMaxSec];                                       // (B) entry state of this segment
  for (int k = 0; k < ns; ++k) G[k] = 0.0;
  double carry[2 * kMaxSec];
  for (int k = 0; k < ns; ++k) carry[k] = 0.0;                 // state leaving segment l-1
  for (int l = 0; l < 64; ++l) {
    // carry holds the entry state of segment l (on every lane, kept uniform)
    if (lane == l)
      for (int k = 0; k < ns; ++k) G[k] = carry[k];
    // state leaving segment l = M^(len_l) carry + S_l ; len_l == seg except for the last, whose
    // exit state nobody needs
    matvec(Mg, ns, carry, tmp);
    for (int k = 0; k < ns; ++k) carry[k] = tmp[k] + __shfl(S[k], l, 64);
  }
  double h[2 * kMaxSec];                                       // (C)
  for (int k = 0; k < ns; ++k) h[k] = G[k];
  for (int b = b0; b < b1; ++b) {
    double* p = fs + (int64_t)b * (2 * kMaxSec);
    for (int k = 0; k < ns; ++k) p[k] += h[k];
    matvec(Ms, ns, h, tmp);
    for (int k = 0; k < ns; ++k) h[k] = tmp[k];
  }
}

// The same entry states by a parallel scan: one workgroup of 256 threads per series walks the blocks in tiles of
// 256; inside a tile the recurrence E_{i+1} = M E_i + f_i is an inclusive scan of h (h_0 = f_0 + M carry,
// h_i = f_i) under "shift by 2^k blocks = multiply by M^(2^k)" -- eight steps with the eight matrices
// Mpow[k] = M^(2^k) -- instead of one lane walking its segment block by block (pp_chain_kernel: 110 - 150 us per
// launch whatever the size, two launches per zero-phase filter).
constexpr int kScanT = 256, kScanSteps = 8;
// NS = 2 x sections at compile time: with a runtime count every loop below runs to the 16 states of eight
// sections under predication (78 us per launch for two sections; 150 for the walk)
template <int NS>
__global__ __launch_bounds__(kScanT) void pp_scan_kernel(double* __restrict__ fstate,
                                                         const double* __restrict__ Mpow, int nblk) {
  __shared__ double Ms[kScanSteps][NS * NS];
  __shared__ double P[2][kScanT][NS + 1];
  __shared__ double carry_s[NS];
  const int ser = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < kScanSteps * NS * NS; k += kScanT) Ms[k / (NS * NS)][k % (NS * NS)] = Mpow[k];
  if (tid < NS) carry_s[tid] = 0.0;
  __syncthreads();
  auto mv = [&](const double* Mk, const double (&v)[NS], double (&out)[NS]) {
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < NS; ++c) acc += Mk[r * NS + c] * v[c];
      out[r] = acc;
    }
  };
  double* fs = fstate + (int64_t)ser * nblk * (2 * kMaxSec);
  for (int t0 = 0; t0 < nblk; t0 += kScanT) {
    const int b = t0 + tid;
    const bool valid = b < nblk;
    double h[NS], tmp[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) h[k] = valid ? fs[(int64_t)b * (2 * kMaxSec) + k] : 0.0;
    if (tid == 0) {
      double c[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) c[k] = carry_s[k];
      mv(Ms[0], c, tmp);
#pragma unroll
      for (int k = 0; k < NS; ++k) h[k] += tmp[k];
    }
    int cur = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) P[0][tid][k] = h[k];
    __syncthreads();
#pragma unroll
    for (int step = 0; step < kScanSteps; ++step) {
      const int d = 1 << step;
      if (tid >= d) {
        double nb[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) nb[k] = P[cur][tid - d][k];
        mv(Ms[step], nb, tmp);
#pragma unroll
        for (int k = 0; k < NS; ++k) h[k] += tmp[k];
      }
#pragma unroll
      for (int k = 0; k < NS; ++k) P[cur ^ 1][tid][k] = h[k];
      cur ^= 1;
      __syncthreads();
    }
    // h = inclusive prefix = the entry state of block b + 1; block b's own entry state is its left neighbour's
    if (valid) {
#pragma unroll
      for (int k = 0; k < NS; ++k)
        fs[(int64_t)b * (2 * kMaxSec) + k] = tid == 0 ? carry_s[k] : P[cur][tid - 1][k];
    }
    __syncthreads();
    if (tid == kScanT - 1) {
#pragma unroll
      for (int k = 0; k < NS; ++k) carry_s[k] = h[k];
    }
    __syncthreads();
  }
}

// Pass 1 without a recurrence.  The state a block leaves behind when it starts from zero is LINEAR in its samples:
// sample j of a block of m is followed by m - 1 - j others, so it contributes x_j A^(m-1-j) B = x_j tab[m - 1 - j]
// (A the sections' one-sample state transition, B their input vector; tab is built on the host by running the
// sections over a unit sample).  A dot product per state instead of kPB dependent filter steps per thread: one wave
// per block, four coalesced samples per lane against four table rows held in registers, a wave reduction per state.
// The same pass adds up the series' trend sums (pp_trend_kernel's); the trend is not known yet, so the states are
// those of the RAW samples and pp_detrend_states_kernel takes the line's own response off them afterwards
// (126 + 51 us of pass 1 and the trend pass per station-day -> one pass over the samples).
// values left per lane after the halving steps of wave_reduce_scatter
__host__ __device__ constexpr int reduce_left(int cnt) {
  int off = 32;
  while (cnt % 2 == 0 && off >= 1) { cnt /= 2; off /= 2; }
  return cnt;
}
// acc[0 .. CNT) of every lane -> sums over the wave.  On return acc[0 .. reduce_left) of a lane hold the totals of
// states kbase .., the same in every lane of its group of 64 / (NS / reduce_left) consecutive lanes.
template <int CNT, int OFF, int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&acc)[N], int lane, int& kbase) {
  if constexpr (OFF >= 1 && CNT % 2 == 0) {
    const bool upper = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < CNT / 2; ++i) {
      const double send = upper ? acc[i] : acc[i + CNT / 2];
      const double keep = upper ? acc[i + CNT / 2] : acc[i];
      acc[i] = keep + __shfl_xor(send, OFF, 64);
    }
    if (upper) kbase += CNT / 2;
    wave_reduce_scatter<CNT / 2, OFF / 2>(acc, lane, kbase);
  } else if constexpr (OFF >= 1) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) acc[i] += __shfl_xor(acc[i], OFF, 64);
    wave_reduce_scatter<CNT, OFF / 2>(acc, lane, kbase);
  }
}

// (TIN = double, REV: the same for the backward pass of a zero-phase filter -- the forward output walked from its end)
template <int NS, bool TREND, typename TIN = float, bool REV = false>
__global__ __launch_bounds__(256) void pp_local_kernel(const TIN* __restrict__ raw, SerLayout sl, int64_t T,
                                                       int nblk, const double* __restrict__ tab,
                                                       double* __restrict__ fstate, double* __restrict__ tsum) {
  __shared__ double sh[8];
  const int ser = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TIN* x = raw + sl.off(ser);
  auto at = [&](int64_t i) { return REV ? x[T - 1 - i] : x[i]; };      // sample i in filter order
  double* fs = fstate + (int64_t)ser * nblk * (2 * kMaxSec);
  double tw[kLR][NS];
#pragma unroll
  for (int r = 0; r < kLR; ++r)
#pragma unroll
    for (int k = 0; k < NS; ++k) tw[r][k] = tab[(kPB - 1 - (lane + 64 * r)) * NS + k];
  const double tc = 0.5 * (double)(T - 1);
  double s0 = 0.0, s1 = 0.0;
  const int per = (nblk + (int)gridDim.x - 1) / (int)gridDim.x;
  const int b_lo = (int)blockIdx.x * per, b_hi = min(b_lo + per, nblk);
  for (int b = b_lo + wave; b < b_hi; b += 4) {
    const int64_t t0 = (int64_t)b * kPB;
    const int m = (int)min((int64_t)kPB, T - t0);
    double acc[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) acc[k] = 0.0;
    if (m == kPB) {
      TIN v[kLR];
#pragma unroll
      for (int r = 0; r < kLR; ++r) v[r] = at(t0 + lane + 64 * r);
#pragma unroll
      for (int r = 0; r < kLR; ++r) {
        const double d = (double)v[r];
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] += tw[r][k] * d;
        if (TREND) { s0 += d; s1 += d * ((double)(t0 + lane + 64 * r) - tc); }
      }
    } else {
      for (int r = 0; r < kLR; ++r) {
        const int j = lane + 64 * r;
        if (j < m) {
          const double d = (double)at(t0 + j);
#pragma unroll
          for (int k = 0; k < NS; ++k) acc[k] += tab[(m - 1 - j) * NS + k] * d;
          if (TREND) { s0 += d; s1 += d * ((double)(t0 + j) - tc); }
        }
      }
    }
    // the wave's sums: while the count of values is even the two halves of the lanes trade halves of them (each
    // keeps the half it goes on summing), the rest by plain butterfly steps -- 7 exchanges for four states, not 24
    int kbase = 0;
    wave_reduce_scatter<NS, 32>(acc, lane, kbase);
    constexpr int kLeft = reduce_left(NS), kFree = 64 / (NS / kLeft);      // values a lane ends with; lanes per group
    if ((lane & (kFree - 1)) == 0) {
#pragma unroll
      for (int k = 0; k < kLeft; ++k) fs[(int64_t)b * (2 * kMaxSec) + kbase + k] = acc[k];
    }
  }
  if (TREND) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s0 += __shfl_down(s0, off, 64);
      s1 += __shfl_down(s1, off, 64);
    }
    if (lane == 0) { sh[wave] = s0; sh[4 + wave] = s1; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&tsum[2 * ser], sh[0] + sh[1] + sh[2] + sh[3]);
      atomicAdd(&tsum[2 * ser + 1], sh[4] + sh[5] + sh[6] + sh[7]);
    }
  }
}

// fstate[b] -= the zero-state response of the trend line a + s t over block b: a C1 + s (t0 C1 + C2), with the
// block sums of the impulse table (cfull for whole blocks; the series' last block adds its own up)
__global__ __launch_bounds__(256) void pp_detrend_states_kernel(double* __restrict__ fstate,
                                                                const double* __restrict__ tsum,
                                                                const double* __restrict__ tab,
                                                                const double* __restrict__ cfull, int64_t T,
                                                                int nblk, int ns) {
  const int ser = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int b = idx / ns, k = idx - b * ns;
  if (b >= nblk) return;
  double a, sl;
  trend_of(tsum + 2 * ser, T, &a, &sl);
  const int64_t t0 = (int64_t)b * kPB;
  const int m = (int)min((int64_t)kPB, T - t0);
  double c1, c2;
  if (m == kPB) {
    c1 = cfull[k]; c2 = cfull[ns + k];
  } else {
    c1 = c2 = 0.0;
    for (int j = 0; j < m; ++j) {
      const double w = tab[(m - 1 - j) * ns + k];
      c1 += w;
      c2 += (double)j * w;
    }
  }
  fstate[((int64_t)ser * nblk + b) * (2 * kMaxSec) + k] -= a * c1 + sl * ((double)t0 * c1 + c2);
}

// no filter: detrend (optional) + multiplex
__global__ __launch_bounds__(256) void pp_copy_kernel(const float* __restrict__ raw,
                                                      SerLayout sl,
                                                      const double* __restrict__ trend, int64_t T,
                                                      int Nc, int64_t mp_stride,
                                                      float* __restrict__ out_mp) {
  const int ser = blockIdx.y;
  const int chunk = ser / Nc, c = ser - chunk * Nc;
  const float* r = raw + sl.off(ser);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (int64_t)gridDim.x * 256) {
    double v = (double)r[t];
    if (trend) {
      double a, b;
      trend_of(trend + 2 * ser, T, &a, &b);
      v -= a + b * (double)t;
    }
    out_mp[(int64_t)chunk * mp_stride + t * Nc + c] = (float)v;
  }
}

// The same without a trend for NC channels, 16 bytes per access on both sides: a thread takes four sample frames,
// one 16-byte load from each plane, NC 16-byte stores of the 4 NC consecutive multiplexed floats.  (The kernel
// above scatters 4-byte stores NC floats apart: 95 us per station-day against 41 - 50.)
template <int NC>
__global__ __launch_bounds__(256) void pp_copy_vec_kernel(const float* __restrict__ raw, SerLayout sl,
                                                          int64_t T, int64_t mp_stride,
                                                          float* __restrict__ out_mp) {
  typedef float fv4 __attribute__((ext_vector_type(4)));
  const int chunk = blockIdx.y;
  const float* r = raw + (int64_t)chunk * sl.chunk;
  float* o = out_mp + (int64_t)chunk * mp_stride;
  for (int64_t t = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x); t < T; t += 4 * (int64_t)gridDim.x * 256) {
    float v[4 * NC];
    if (t + 3 < T) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const fv4 w = __builtin_nontemporal_load(reinterpret_cast<const fv4*>(r + (int64_t)c * sl.chan + t));
        v[c] = w.x; v[NC + c] = w.y; v[2 * NC + c] = w.z; v[3 * NC + c] = w.w;
      }
#pragma unroll
      for (int q = 0; q < NC; ++q)
        *reinterpret_cast<float4*>(o + t * NC + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
      for (int k = 0; t + k < T; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) o[(t + k) * NC + c] = r[(int64_t)c * sl.chan + t + k];
    }
  }
}

// host-side preparation shared by dtx_preprocess and dtx_decimate: normalised sections,
// block transition matrix M and M^seg, workspace carving
struct PpPlan {
  Sos f;
  int nser, nblk, ns, seg;
  double *trend, *fstate, *Mdev, *mid;
  // impulse table of the sections (pp_local_kernel): tab[q][k] = state k, q samples after a unit sample entered a
  // zero state; behind it its sums over a full block, C1[k] = sum_q tab[q][k], C2[k] = sum_j j tab[kPB - 1 - j][k]
  const double* tab() const { return Mdev + (size_t)(2 + kScanSteps) * ns * ns; }
  const double* cfull() const { return tab() + (size_t)kPB * ns; }
  // the normalised sections [b0 | b1 | b2 | a1 | a2][kMaxSec] (pp_tiled_kernel reads them into registers)
  const double* coef() const { return cfull() + 2 * (size_t)ns; }
};

int pp_plan(dtx_ctx* ctx, const char* who, const double* sos, int nsec, int nser, int64_t T,
            bool need_mid, PpPlan* pl) {
  if (nsec < 0 || nsec > kMaxSec || (nsec > 0 && !sos))
    return dtx_fail(ctx, DTX_E_ARG, "%s: 0..%d second-order sections supported", who, kMaxSec);
  pl->nser = nser;
  pl->nblk = (int)((T + kPB - 1) / kPB);
  pl->ns = 2 * nsec;
  pl->seg = (pl->nblk + 63) / 64;     // blocks per lane in the chain kernel
  const int ns = pl->ns, nblk = pl->nblk;
  Sos& f = pl->f;
  f.nsec = nsec;
  for (int s = 0; s < nsec; ++s) {
    const double a0 = sos[6 * s + 3];
    if (a0 == 0.0) return dtx_fail(ctx, DTX_E_ARG, "%s: sos a0 == 0", who);
    f.b0[s] = sos[6 * s + 0] / a0; f.b1[s] = sos[6 * s + 1] / a0; f.b2[s] = sos[6 * s + 2] / a0;
    f.a1[s] = sos[6 * s + 4] / a0; f.a2[s] = sos[6 * s + 5] / a0;
  }
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_tr = carve((size_t)nser * 2 * sizeof(double));
  const size_t o_fs = carve((size_t)nser * nblk * 2 * kMaxSec * sizeof(double));
  const size_t o_M = carve((size_t)(2 * ns * ns + 2) * sizeof(double));
  const size_t o_mid = carve(need_mid ? (size_t)nser * T * sizeof(double) : 0);
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc3, off))) return rc;
  char* base = (char*)ctx->ws_misc3.p;
  pl->trend = (double*)(base + o_tr);
  pl->fstate = (double*)(base + o_fs);
  pl->Mdev = (double*)(base + o_M);
  pl->mid = (double*)(base + o_mid);
  if (nsec == 0) return DTX_OK;

  // the transition matrices depend on the sections and on seg only: built once, kept on the device (rounds 1-2
  // rebuilt and copied them per call and synchronised the stream for the copy's host buffer -- a raw-counts
  // SubSpace.detex waited for the previous station's scan at every chunk group)
  std::vector<double> key;
  key.push_back((double)nsec); key.push_back((double)pl->seg); key.push_back((double)kPB);
  key.insert(key.end(), sos, sos + 6 * (size_t)nsec);
  for (int e = 0; e < dtx_ctx::kPpCache; ++e)
    if (ctx->pp_M[e].p != nullptr && key == ctx->pp_key[e]) {
      pl->Mdev = (double*)ctx->pp_M[e].p;
      return DTX_OK;
    }
  const int slot = ctx->pp_next;
  ctx->pp_next = (ctx->pp_next + 1) % dtx_ctx::kPpCache;
  DtxBuf& Mbuf = ctx->pp_M[slot];
  std::vector<double>& Mkey = ctx->pp_key[slot];
  // state transition over one block: column k = free response of unit state k
  std::vector<double> M((size_t)ns * ns, 0.0);
  for (int k = 0; k < ns; ++k) {
    double z[2 * kMaxSec] = {0};
    z[k] = 1.0;
    for (int i = 0; i < kPB; ++i) {
      double x = 0.0;
      for (int s = 0; s < nsec; ++s) {
        const double y = f.b0[s] * x + z[2 * s];
        z[2 * s] = f.b1[s] * x - f.a1[s] * y + z[2 * s + 1];
        z[2 * s + 1] = f.b2[s] * x - f.a2[s] * y;
        x = y;
      }
    }
    for (int r = 0; r < ns; ++r) M[(size_t)r * ns + k] = z[r];
  }
  {  // Mseg = M^seg
    std::vector<double> P((size_t)ns * ns, 0.0), Q((size_t)ns * ns);
    for (int k = 0; k < ns; ++k) P[(size_t)k * ns + k] = 1.0;
    for (int e = 0; e < pl->seg; ++e) {
      for (int r = 0; r < ns; ++r)
        for (int c = 0; c < ns; ++c) {
          double acc = 0.0;
          for (int k = 0; k < ns; ++k) acc += M[(size_t)r * ns + k] * P[(size_t)k * ns + c];
          Q[(size_t)r * ns + c] = acc;
        }
      P.swap(Q);
    }
    M.insert(M.end(), P.begin(), P.end());
  }
  {  // M^(2^k), k = 0 .. kScanSteps - 1, for pp_scan_kernel
    std::vector<double> P(M.begin(), M.begin() + (size_t)ns * ns), Q((size_t)ns * ns);
    for (int e = 0; e < kScanSteps; ++e) {
      M.insert(M.end(), P.begin(), P.end());
      for (int r = 0; r < ns; ++r)
        for (int c = 0; c < ns; ++c) {
          double acc = 0.0;
          for (int k = 0; k < ns; ++k) acc += P[(size_t)r * ns + k] * P[(size_t)k * ns + c];
          Q[(size_t)r * ns + c] = acc;
        }
      P.swap(Q);
    }
  }
  {  // the impulse table and its block sums
    std::vector<double> tab((size_t)kPB * ns), c1(ns, 0.0), c2(ns, 0.0);
    double z[2 * kMaxSec] = {0};
    for (int q = 0; q < kPB; ++q) {
      double x = q == 0 ? 1.0 : 0.0;
      for (int sct = 0; sct < nsec; ++sct) {
        const double y = f.b0[sct] * x + z[2 * sct];
        z[2 * sct] = f.b1[sct] * x - f.a1[sct] * y + z[2 * sct + 1];
        z[2 * sct + 1] = f.b2[sct] * x - f.a2[sct] * y;
        x = y;
      }
      for (int k = 0; k < ns; ++k) tab[(size_t)q * ns + k] = z[k];
    }
    for (int j = 0; j < kPB; ++j)
      for (int k = 0; k < ns; ++k) {
        const double w = tab[(size_t)(kPB - 1 - j) * ns + k];
        c1[k] += w;
        c2[k] += (double)j * w;
      }
    M.insert(M.end(), tab.begin(), tab.end());
    M.insert(M.end(), c1.begin(), c1.end());
    M.insert(M.end(), c2.begin(), c2.end());
    for (const double* arr : {f.b0, f.b1, f.b2, f.a1, f.a2})
      for (int sct = 0; sct < kMaxSec; ++sct) M.push_back(sct < nsec ? arr[sct] : 0.0);
  }
  Mkey.clear();
  // (kernels of earlier calls may still read the old matrices: wait for them before overwriting)
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = dtx_reserve(ctx, Mbuf, ((size_t)2 * kMaxSec * 2 * kMaxSec * (2 + kScanSteps) + (size_t)(kPB + 2) * 2 * kMaxSec + 5 * kMaxSec) *
                                       sizeof(double))))
    return rc;
  DTX_HIP(ctx, hipMemcpy(Mbuf.p, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice));
  Mkey = key;
  pl->Mdev = (double*)Mbuf.p;
  return DTX_OK;
}

// entry states of every block of every series: the scan kernel for its state count (the per-lane walk for an
// odd one)
static void launch_chain(dtx_ctx* ctx, double* fstate, const double* Mdev, int nser, int nblk, int seg, int ns) {
  const dim3 cgrid((unsigned)nser);
  const double* Mp = Mdev + 2 * ns * ns;
  switch (ns) {
#define DTX_SCAN_CASE(NS) case NS: hipLaunchKernelGGL(pp_scan_kernel<NS>, cgrid, dim3(kScanT), 0, ctx->stream, fstate, Mp, nblk); return
    DTX_SCAN_CASE(2); DTX_SCAN_CASE(4); DTX_SCAN_CASE(6); DTX_SCAN_CASE(8);
    DTX_SCAN_CASE(10); DTX_SCAN_CASE(12); DTX_SCAN_CASE(14); DTX_SCAN_CASE(16);
#undef DTX_SCAN_CASE
    default: break;
  }
  hipLaunchKernelGGL(pp_chain_kernel, cgrid, dim3(64), 0, ctx->stream, fstate, Mdev, Mdev + ns * ns, nblk, seg, ns);
}

// pp_tiled_kernel for the plan's section count
template <bool FIRST, bool RUN, bool LAST>
static void launch_tiled(dtx_ctx* ctx, dim3 grid, const PpPlan& pl, const float* raw, SerLayout sl, const double* mid,
                         const double* trend, int64_t T, bool rev, double* out_mid, float* out_mp, int Nc,
                         int64_t mp_stride, int dec) {
#define DTX_TILED_CASE(NSEC)                                                                                      \
  case NSEC:                                                                                                      \
    hipLaunchKernelGGL((pp_tiled_kernel<FIRST, RUN, LAST, NSEC>), grid, dim3(256), 0, ctx->stream, raw, sl,         \
                       mid, trend, T, pl.nblk, pl.coef(), rev, pl.fstate, out_mid, out_mp, Nc, mp_stride, dec);    \
    break
  switch (pl.f.nsec) {
    DTX_TILED_CASE(1); DTX_TILED_CASE(2); DTX_TILED_CASE(3); DTX_TILED_CASE(4);
    DTX_TILED_CASE(5); DTX_TILED_CASE(6); DTX_TILED_CASE(7); DTX_TILED_CASE(8);
    default: break;
  }
#undef DTX_TILED_CASE
}

// pass 1 (+ the trend sums when tsum is given): zero-state exit states of every block of every series
static int launch_local(dtx_ctx* ctx, const float* raw, SerLayout sl, int64_t T, const PpPlan& pl, double* tsum) {
  // ~2000 workgroups whatever the series count; a wave takes every fourth block of its workgroup's run
  const int slices = std::max(1, std::min((pl.nblk + 7) / 8, (2048 + pl.nser - 1) / pl.nser));
  const dim3 grid((unsigned)slices, (unsigned)pl.nser);
#define DTX_LOCAL_CASE(NS)                                                                                          \
  case NS:                                                                                                          \
    if (tsum) hipLaunchKernelGGL((pp_local_kernel<NS, true>), grid, dim3(256), 0, ctx->stream, raw, sl, T,          \
                                 pl.nblk, pl.tab(), pl.fstate, tsum);                                               \
    else hipLaunchKernelGGL((pp_local_kernel<NS, false>), grid, dim3(256), 0, ctx->stream, raw, sl, T,              \
                            pl.nblk, pl.tab(), pl.fstate, tsum);                                                    \
    break
  switch (pl.ns) {
    DTX_LOCAL_CASE(2); DTX_LOCAL_CASE(4); DTX_LOCAL_CASE(6); DTX_LOCAL_CASE(8);
    DTX_LOCAL_CASE(10); DTX_LOCAL_CASE(12); DTX_LOCAL_CASE(14); DTX_LOCAL_CASE(16);
    default: return dtx_fail(ctx, DTX_E_ARG, "pre-processing: %d filter states", pl.ns);
  }
#undef DTX_LOCAL_CASE
  DTX_HIP(ctx, hipGetLastError());
  if (tsum) {
    const dim3 g2((unsigned)(((int64_t)pl.nblk * pl.ns + 255) / 256), (unsigned)pl.nser);
    hipLaunchKernelGGL(pp_detrend_states_kernel, g2, dim3(256), 0, ctx->stream, pl.fstate, (const double*)tsum,
                       pl.tab(), pl.cfull(), T, pl.nblk, pl.ns);
    DTX_HIP(ctx, hipGetLastError());
  }
  return DTX_OK;
}

// the same for the backward pass over the forward output (float64, walked from its end)
static int launch_local_back(dtx_ctx* ctx, const double* mid, int64_t T, const PpPlan& pl) {
  const int slices = std::max(1, std::min((pl.nblk + 7) / 8, (2048 + pl.nser - 1) / pl.nser));
  const dim3 grid((unsigned)slices, (unsigned)pl.nser);
#define DTX_LOCALB_CASE(NS)                                                                                        \
  case NS:                                                                                                         \
    hipLaunchKernelGGL((pp_local_kernel<NS, false, double, true>), grid, dim3(256), 0, ctx->stream, mid, ser_uniform(T), T, \
                       pl.nblk, pl.tab(), pl.fstate, (double*)nullptr);                                            \
    break
  switch (pl.ns) {
    DTX_LOCALB_CASE(2); DTX_LOCALB_CASE(4); DTX_LOCALB_CASE(6); DTX_LOCALB_CASE(8);
    DTX_LOCALB_CASE(10); DTX_LOCALB_CASE(12); DTX_LOCALB_CASE(14); DTX_LOCALB_CASE(16);
    default: return dtx_fail(ctx, DTX_E_ARG, "pre-processing: %d filter states", pl.ns);
  }
#undef DTX_LOCALB_CASE
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

// ---------------------------------------------------------------- raw-count ingest
// The front half of _applyFilter (detex/construct.py:1005-1022): a chunk arrives as FRAGMENTS of
// raw int32 digitiser counts (ObsPy traces of one channel each, with gaps between them).
//   ingest_scatter_kernel   Stream.merge(fill_value=0.0) (construct.py:1038 / 1070): every
//                           fragment converted to float32 at its place in a zeroed [Nc][T] plane
//   ingest_run_kernel       _mergeChannels (1033-1066): the product of the channels is zero where
//                           any channel has a gap (or an exact zero sample -- the reference's
//                           test is on the values); the longest run of non-zero products, the
//                           first one among equals (1059-1063), is what the chunk is trimmed to
struct IngestFrag { int64_t chan, src, dst, len; };

__global__ __launch_bounds__(256) void ingest_scatter_kernel(const int32_t* __restrict__ counts,
                                                             const IngestFrag* __restrict__ frag,
                                                             int64_t T, float* __restrict__ planes) {
  const IngestFrag f = frag[blockIdx.y];
  const int32_t* src = counts + f.src;
  float* dst = planes + f.chan * T + f.dst;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < f.len; i += (int64_t)gridDim.x * 256)
    dst[i] = (float)src[i];
}

// longest run of samples at which every channel is non-zero: each thread summarises a slice
// (length, leading run, trailing run, best run and where it starts), slices are combined in order
struct RunSeg { int64_t len, pre, suf, best, best_at; };
__device__ inline RunSeg run_join(const RunSeg& a, const RunSeg& b, int64_t a0, int64_t b0) {
  // a covers [a0, a0 + a.len), b the samples right behind it starting at b0
  RunSeg r;
  r.len = a.len + b.len;
  r.pre = a.pre == a.len ? a.len + b.pre : a.pre;
  r.suf = b.suf == b.len ? b.len + a.suf : b.suf;
  r.best = a.best; r.best_at = a.best_at;
  const int64_t mid = a.suf + b.pre, mid_at = b0 - a.suf;
  if (mid > r.best) { r.best = mid; r.best_at = mid_at; }          // strictly longer: the first one wins ties
  if (b.best > r.best) { r.best = b.best; r.best_at = b.best_at; }
  return r;
}

// Part g of the record: samples [g * per_blk, (g + 1) * per_blk), 1024 slices, combined in order -> part[g]
// (one workgroup for the whole record took 1.0 ms for an hour of 3-channel data: 363 serial, uncoalesced
// samples per thread on ONE CU; the parts run on as many CUs and a second, tiny launch joins them)
__global__ __launch_bounds__(1024) void ingest_run_kernel(const float* __restrict__ planes, int Nc,
                                                          int64_t T, int64_t stride, int64_t per_blk,
                                                          RunSeg* __restrict__ part,
                                                          int64_t* __restrict__ part0) {
  __shared__ RunSeg seg[1024];
  __shared__ int64_t seg0[1024];
  const int tid = threadIdx.x;
  const int64_t b0 = min((int64_t)blockIdx.x * per_blk, T), b1 = min(b0 + per_blk, T);
  const int64_t per = (b1 - b0 + 1023) / 1024;
  const int64_t t0 = min(b0 + (int64_t)tid * per, b1), t1 = min(t0 + per, b1);
  RunSeg s{t1 - t0, 0, 0, 0, t0};
  int64_t cur = 0;
  bool lead = true;
  for (int64_t t = t0; t < t1; ++t) {
    bool nz = true;
    for (int c = 0; c < Nc; ++c) nz = nz && planes[(int64_t)c * stride + t] != 0.f;
    if (nz) {
      ++cur;
      if (cur > s.best) { s.best = cur; s.best_at = t - cur + 1; }
    } else {
      if (lead) { s.pre = cur; lead = false; }
      cur = 0;
    }
  }
  if (lead) s.pre = cur;
  s.suf = cur;
  seg[tid] = s;
  seg0[tid] = t0;
  __syncthreads();
  for (int stride = 1; stride < 1024; stride <<= 1) {
    if ((tid & (2 * stride - 1)) == 0) seg[tid] = run_join(seg[tid], seg[tid + stride], seg0[tid], seg0[tid + stride]);
    __syncthreads();
  }
  if (tid == 0) {
    part[blockIdx.x] = seg[0];
    part0[blockIdx.x] = b0;
  }
}

// the parts in order -> run = {first sample of the longest run (first among equals), its length}, -1 if none
__global__ void ingest_run_final_kernel(const RunSeg* __restrict__ part, const int64_t* __restrict__ part0,
                                        int nparts, int64_t* __restrict__ run) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  RunSeg acc = part[0];
  const int64_t a0 = part0[0];
  for (int g = 1; g < nparts; ++g) acc = run_join(acc, part[g], a0, part0[g]);
  run[0] = acc.best > 0 ? acc.best_at : -1;
  run[1] = acc.best;
}

// both launches; the parts live in the context's small workspace
int launch_run(dtx_ctx* ctx, const float* planes, int Nc, int64_t T, int64_t stride, int64_t* run_dev) {
  const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>(64, (T + 8191) / 8192));
  const int64_t per_blk = (T + nparts - 1) / nparts;
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_trig_doff, (size_t)nparts * (sizeof(RunSeg) + sizeof(int64_t))))) return rc;
  RunSeg* part = (RunSeg*)ctx->ws_trig_doff.p;
  int64_t* part0 = (int64_t*)(part + nparts);
  hipLaunchKernelGGL(ingest_run_kernel, dim3((unsigned)nparts), dim3(1024), 0, ctx->stream, planes, Nc, T, stride,
                     per_blk, part, part0);
  hipLaunchKernelGGL(ingest_run_final_kernel, dim3(1), dim3(64), 0, ctx->stream, (const RunSeg*)part,
                     (const int64_t*)part0, nparts, run_dev);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

}  // namespace

// dtx_preprocess / dtx_preprocess_planes behind their argument checks
static int preprocess_impl(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc, int64_t T, const SerLayout sl,
                           const double* sos, int nsec, int zerophase, int detrend, float* mp_dev,
                           int64_t mp_stride) {
  DtxDeviceGuard g(ctx->device);
  const int nser = nchunk * Nc;
  PpPlan pl;
  int rc;
  if ((rc = pp_plan(ctx, "dtx_preprocess", sos, nsec, nser, T, zerophase && nsec, &pl))) return rc;
  const int nblk = pl.nblk, ns = pl.ns, seg = pl.seg;
  double* trend = detrend ? pl.trend : nullptr;
  double *fstate = pl.fstate, *Mdev = pl.Mdev, *mid = pl.mid;

  if (detrend) {
    DTX_HIP(ctx, hipMemsetAsync(trend, 0, (size_t)nser * 2 * sizeof(double), ctx->stream));
    if (nsec == 0 || !1) {          // (with a filter the trend sums come out of pass 1)
      hipLaunchKernelGGL(pp_trend_kernel, dim3(kTrendSlices, (unsigned)nser), dim3(256), 0,
                         ctx->stream, raw_dev, T, sl, trend);
      DTX_HIP(ctx, hipGetLastError());
    }
  }
  if (nsec == 0) {
    // no trend, 16-byte aligned planes and output: the vector kernel
    const bool vec = trend == nullptr && (sl.chunk & 3) == 0 && (sl.chan & 3) == 0 && (mp_stride & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(raw_dev) & 15) == 0 && (reinterpret_cast<uintptr_t>(mp_dev) & 15) == 0;
    const dim3 vgrid((unsigned)std::min<int64_t>((T + 1023) / 1024, 4096), (unsigned)(nser / std::max(Nc, 1)));
#define DTX_PP_COPY_VEC(NC)                                                                                  \
    hipLaunchKernelGGL(pp_copy_vec_kernel<NC>, vgrid, dim3(256), 0, ctx->stream, raw_dev, sl, T,             \
                       mp_stride, mp_dev)
    if (vec && (Nc == 1 || Nc == 2 || Nc == 3 || Nc == 4 || Nc == 6)) {
      switch (Nc) {
        case 1: DTX_PP_COPY_VEC(1); break;
        case 2: DTX_PP_COPY_VEC(2); break;
        case 3: DTX_PP_COPY_VEC(3); break;
        case 4: DTX_PP_COPY_VEC(4); break;
        default: DTX_PP_COPY_VEC(6); break;
      }
      DTX_HIP(ctx, hipGetLastError());
      return DTX_OK;
    }
#undef DTX_PP_COPY_VEC
    dim3 grid((unsigned)std::min<int64_t>((T + 255) / 256, 4096), (unsigned)nser);
    hipLaunchKernelGGL(pp_copy_kernel, grid, dim3(256), 0, ctx->stream, raw_dev, sl,
                       (const double*)trend, T, Nc, mp_stride, mp_dev);
    DTX_HIP(ctx, hipGetLastError());
    return DTX_OK;
  }
  dim3 grid((unsigned)((nblk + 255) / 256), (unsigned)nser);
  const double* nod = nullptr;
  const float* nof = nullptr;
  // forward pass: raw - trend -> (mid | output)
  if ((rc = launch_local(ctx, raw_dev, sl, T, pl, trend))) return rc;
  auto chain = [&]() { launch_chain(ctx, fstate, (const double*)Mdev, nser, nblk, seg, ns); };
  chain();
  if (zerophase) {
    launch_tiled<true, true, false>(ctx, grid, pl, raw_dev, sl, nod, (const double*)trend, T, false, mid,
                                    (float*)nullptr, Nc, mp_stride, 0);
    // backward pass over the forward output (obspy: sosfilt(sos, firstpass[::-1])[::-1])
    if ((rc = launch_local_back(ctx, mid, T, pl))) return rc;
    chain();
    // in place: every thread reads its samples of `mid` into LDS before any store of that tile
    launch_tiled<false, true, true>(ctx, grid, pl, nof, sl, (const double*)mid, nod, T, true,
                                    (double*)nullptr, mp_dev, Nc, mp_stride, 0);
  } else {
    launch_tiled<true, true, true>(ctx, grid, pl, raw_dev, sl, nod, (const double*)trend, T, false,
                                   (double*)nullptr, mp_dev, Nc, mp_stride, 0);
  }
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

extern "C" {

int dtx_preprocess(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc, int64_t T,
                   int64_t ser_stride, const double* sos, int nsec, int zerophase, int detrend,
                   float* mp_dev, int64_t mp_stride) {
  if (!ctx) return DTX_E_ARG;
  if (!raw_dev || !mp_dev || nchunk <= 0 || Nc <= 0 || T <= 0 || ser_stride < T ||
      mp_stride < T * Nc)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_preprocess: bad argument");
  return preprocess_impl(ctx, raw_dev, nchunk, Nc, T, SerLayout{(int64_t)Nc * ser_stride, ser_stride, Nc}, sos, nsec,
                         zerophase, detrend, mp_dev, mp_stride);
}

int dtx_preprocess_planes(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc, int64_t T,
                          int64_t chunk_stride, int64_t chan_stride, const double* sos, int nsec, int zerophase,
                          int detrend, float* mp_dev, int64_t mp_stride) {
  if (!ctx) return DTX_E_ARG;
  if (!raw_dev || !mp_dev || nchunk <= 0 || Nc <= 0 || T <= 0 || chunk_stride < 0 || (Nc > 1 && chan_stride < T) ||
      mp_stride < T * Nc)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_preprocess_planes: bad argument");
  return preprocess_impl(ctx, raw_dev, nchunk, Nc, T, SerLayout{chunk_stride, chan_stride, Nc}, sos, nsec, zerophase,
                         detrend, mp_dev, mp_stride);
}

// Trace.decimate of construct.py:1016-1017: causal low-pass (forward sosfilt of the
// caller's sections, no detrend) and every factor-th sample kept, per series plane
int dtx_decimate(dtx_ctx* ctx, const float* raw_dev, int nser, int64_t T, int64_t ser_stride,
                 const double* sos, int nsec, int factor, float* out_dev, int64_t out_stride) {
  if (!ctx) return DTX_E_ARG;
  if (!raw_dev || !out_dev || nser <= 0 || T <= 0 || ser_stride < T || factor < 1 ||
      out_stride < (T + factor - 1) / factor)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_decimate: bad argument");
  if (nsec < 1) return dtx_fail(ctx, DTX_E_ARG, "dtx_decimate: needs >= 1 low-pass section");
  DtxDeviceGuard g(ctx->device);
  PpPlan pl;
  int rc;
  if ((rc = pp_plan(ctx, "dtx_decimate", sos, nsec, nser, T, false, &pl))) return rc;
  dim3 grid((unsigned)((pl.nblk + 255) / 256), (unsigned)nser);
  const double* nod = nullptr;
  if ((rc = launch_local(ctx, raw_dev, ser_uniform(ser_stride), T, pl, nullptr))) return rc;
  launch_chain(ctx, pl.fstate, (const double*)pl.Mdev, nser, pl.nblk, pl.seg, pl.ns);
  // Nc = 1: "multiplexed" and planar coincide, so factor 1 is plain causal filtering
  launch_tiled<true, true, true>(ctx, grid, pl, raw_dev, ser_uniform(ser_stride), nod, nod, T, false, (double*)nullptr, out_dev, 1,
                                 out_stride, factor);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}


int dtx_ingest_counts(dtx_ctx* ctx, const int32_t* counts_dev, const int64_t* frag_host, int nfrag,
                      int Nc, int64_t T, float* planes_dev, int64_t* run_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!counts_dev || !frag_host || !planes_dev || nfrag < 0 || Nc <= 0 || T <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_ingest_counts: null pointer or non-positive size");
  for (int i = 0; i < nfrag; ++i) {
    const int64_t* f = frag_host + 4 * (int64_t)i;
    if (f[0] < 0 || f[0] >= Nc || f[1] < 0 || f[2] < 0 || f[3] < 0 || f[2] + f[3] > T)
      return dtx_fail(ctx, DTX_E_ARG, "dtx_ingest_counts: fragment %d lies outside the %lld-sample chunk",
                      i, (long long)T);
  }
  DtxDeviceGuard g(ctx->device);
  int rc;
  DTX_HIP(ctx, hipMemsetAsync(planes_dev, 0, (size_t)Nc * T * sizeof(float), ctx->stream));
  if (nfrag > 0) {
    if ((rc = dtx_reserve(ctx, ctx->ws_misc2, (size_t)nfrag * sizeof(IngestFrag)))) return rc;
    // a few dozen bytes per fragment; the descriptors are host data of the caller: through the context's
    // staging ring (a stream synchronisation here made every station of a raw-counts SubSpace.detex wait
    // for the scans queued before it)
    if ((rc = dtx_stage_h2d(ctx, ctx->ws_misc2.p, frag_host, (size_t)nfrag * sizeof(IngestFrag)))) return rc;
    int64_t longest = 1;
    for (int i = 0; i < nfrag; ++i) longest = std::max(longest, frag_host[4 * (int64_t)i + 3]);
    const unsigned gx = (unsigned)std::min<int64_t>((longest + 255) / 256, 1024);
    hipLaunchKernelGGL(ingest_scatter_kernel, dim3(gx, (unsigned)nfrag), dim3(256), 0, ctx->stream,
                       counts_dev, (const IngestFrag*)ctx->ws_misc2.p, T, planes_dev);
    DTX_HIP(ctx, hipGetLastError());
  }
  if (run_dev) {
    if ((rc = launch_run(ctx, (const float*)planes_dev, Nc, T, T, run_dev))) return rc;
  }
  return DTX_OK;
}

int dtx_gapfree_run(dtx_ctx* ctx, const float* planes_dev, int Nc, int64_t T, int64_t plane_stride,
                    int64_t* run_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!planes_dev || !run_dev || Nc <= 0 || T <= 0 || plane_stride < T)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_gapfree_run: null pointer or bad size");
  DtxDeviceGuard g(ctx->device);
  return launch_run(ctx, planes_dev, Nc, T, plane_stride, run_dev);
}

}  // extern "C"
