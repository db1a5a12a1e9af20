// Subspace detection statistic on CDNA4 (gfx950): replaces the per-detector
// FFT loop of the reference (detex/detect.py:259-268 -> _MPXDS 559-578, and its
// twin detex/fas.py:120-134) by one implicit-Hankel FP32 MFMA contraction.
//
//   DS[d][p] = sum_{k in d} (U_k . w_p - S_k mu_p)^2 / (n/(n-1) sum_q (w_pq - mu_p)^2)
//   w_p = x[p*Nc : p*Nc + n]                       
//
// With the stream de-multiplexed into channel planes x_c[t] = x[Nc*t + c] and
// the basis permuted the same way, U_k . w_p = sum_c sum_j U_c[k][j] x_c[p + j]:
// a GEMM  Y[K x P] = U[K x n] * H[n x P]  whose right operand is a Hankel matrix
// that is never materialised (dtx_hankel.h).  The normalisation terms (mu_p,
// denominator) come from a float64 sliding-sum kernel; the per-detector sum of
// squares is a second, tiny MFMA with a 0/1 detector-membership matrix, done in
// registers on the accumulator tile (no LDS round trip, no atomics,
// deterministic).
#include <hip/hip_bf16.h>

#include "dtx_hankel.h"
#include "dtx_hankel_bf16.h"
#include "dtx_internal.h"
#include "dtx_plan.h"
#include "dtx_postdev.h"

using namespace dtx;


namespace {

// ------------------------------------------------------------------ block offset, demux
// Block offset.  The statistic is invariant to adding a constant to the data block, but its
// float32 evaluation is not: with a DC offset much larger than the signal and basis rows of
// non-zero mean, U.w and S mu cancel catastrophically.  Removing an offset first (exact in exact
// arithmetic: U.(w - m) - S (mu - m) = U.w - S mu) keeps float32 safe.  Any estimate of the
// block's ordinary level does -- it only conditions the arithmetic -- but it must not follow a
// spike or a burst of any size, nor zero-filled gaps (getdata fillZeros).  A chunk is cut into up
// to 64 slices; each slice takes the float64 mean of its non-zero samples in ONE pass over the
// stream (no atomics); the offset is the MEDIAN of the slice means: a spike or a burst moves only
// the slices it touches.  (Rounds 1-2 took a chunk-wide mean and a 3-sigma clipped second pass:
// two launches, two reads of the stream, float64 atomics -- 39 + 31 us per station-day.)
constexpr int kMeanSlices = 64;                 // one slice per lane of the wave that takes the median
constexpr int kCsum = 3;                       // per slice: {sum, count, sum of squares} of the non-zero samples
constexpr int kOffStride = 8;                  // long slices: one run of 256 samples in eight is read
constexpr double kStatsBig = 1048576.0;        // samples beyond 1024 x the chunk's ordinary scatter are summed apart (win_stats_kernel)
// ctab != NULL (ragged batch): chunk b starts at x + ctab[2 b] and holds min(ctab[2 b + 1], M) samples
__device__ inline const float* chunk_base(const float* x, int64_t chunk_stride, const int64_t* ctab, int b,
                                          int64_t& M) {
  if (ctab == nullptr) return x + (int64_t)b * chunk_stride;
  M = max((int64_t)0, min(M, ctab[2 * b + 1]));
  return x + ctab[2 * b];
}

__global__ __launch_bounds__(256) void chunk_offset_kernel(const float* __restrict__ x, int64_t M,
                                                           int64_t chunk_stride, int nsl,
                                                           double* __restrict__ cslice,
                                                           const int64_t* __restrict__ ctab) {
  __shared__ double sh[3][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float* xb = chunk_base(x, chunk_stride, ctab, b, M);
  const int64_t per = (M + nsl - 1) / nsl;
  const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(i0 + per, M);
  double s0 = 0.0, s2 = 0.0, s3 = 0.0;
  auto take = [&](float vf) {
    if (vf != 0.f) { s0 += (double)vf; s2 += 1.0; s3 += (double)vf * (double)vf; }
  };
  // a slice longer than 8 x 256 samples is sampled: one run of 256 consecutive samples out of every
  // kOffStride runs, spread over the whole slice (gaps and bursts are met in proportion); the level only
  // conditions the float32 arithmetic -- 2000 samples give it to 2 % of the noise sigma
  const int64_t nrun = (i1 - i0 + 255) / 256;
  const int64_t stride = nrun > 8 ? kOffStride : 1;
  int64_t i = i0 + tid;
  for (; i + 7 * 256 * stride < i1; i += 8 * 256 * stride) {       // eight loads in flight
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = xb[i + 256 * stride * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) take(v[k]);
  }
  for (; i < i1; i += 256 * stride) take(xb[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_down(s0, off, 64);
    s2 += __shfl_down(s2, off, 64);
    s3 += __shfl_down(s3, off, 64);
  }
  if ((tid & 63) == 0) { sh[0][tid >> 6] = s0; sh[1][tid >> 6] = s2; sh[2][tid >> 6] = s3; }
  __syncthreads();
  if (tid == 0) {
    double* cs = cslice + ((int64_t)b * nsl + blockIdx.x) * kCsum;
    cs[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    cs[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    cs[2] = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
  }
}

// m0[b] = (lower) median of the means of chunk b's slices that hold any non-zero sample; 0 for an
// all-zero chunk.  One wave per chunk, ranks by counting: deterministic, every consumer reads the same float
// ragged batch: also pb[b] = the chunk's own number of windows (0 when it is not longer than the basis)
// It also clears what later passes of the same call accumulate into (two fill launches less per call): the
// guards' counters (DtxRepairWs::count) and the chunk's row of maxima (zmax: [nchunk][D], may be NULL).
__global__ __launch_bounds__(64) void chunk_offset_final_kernel(const double* __restrict__ cslice, int nsl,
                                                                float* __restrict__ m0, float* __restrict__ tau,
                                                                const int64_t* __restrict__ ctab, int64_t M,
                                                                int n, int Nc, int* __restrict__ pb,
                                                                int* __restrict__ zlist, float* __restrict__ zmax,
                                                                int D, int* __restrict__ zover, int nover) {
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int q = lane; q < nover; q += 64) zover[(int64_t)b * nover + q] = 0;      // win_stats_kernel's flags
  if (zlist != nullptr && b == 0 && lane < 4) zlist[lane] = 0;      // DtxRepairWs::count
  if (zmax != nullptr)
    for (int d = lane; d < D; d += 64) zmax[(int64_t)b * D + d] = 0.f;
  if (ctab != nullptr && lane == 0) {
    const int64_t Mb = max((int64_t)0, min(M, ctab[2 * b + 1]));
    pb[b] = Mb > n ? (int)((Mb - n) / Nc + 1) : 0;
  }
  double v = 0.0, var = 0.0;
  bool ok = false;
  if (lane < nsl) {
    const double* cs = cslice + ((int64_t)b * nsl + lane) * kCsum;
    if (cs[1] >= 0.5) {                                          // (a slice that holds a NaN is left out)
      v = cs[0] / cs[1];
      var = fmax(cs[2] / cs[1] - v * v, 0.0);
      ok = v == v && var == var;
    }
  }
  const unsigned long long mask = __ballot(ok);
  const int nv = __popcll(mask);
  if (nv == 0) {
    if (lane == 0) { m0[b] = 0.f; tau[b] = INFINITY; }
    return;
  }
  int rank = 0, rankv = 0;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const double vj = __shfl(v, j, 64), wj = __shfl(var, j, 64);
    if ((mask >> j) & 1ull) {
      if (vj < v || (vj == v && j < lane)) ++rank;
      if (wj < var || (wj == var && j < lane)) ++rankv;
    }
  }
  if (ok && rank == (nv - 1) / 2) m0[b] = (float)v;
  // the (lower) median of the slices' variances, times kStatsBig: a sample whose square (offset removed) lies above
  // it is summed apart by win_stats_kernel.  Only a conditioning device, like the offset: any value gives the
  // same sums in exact arithmetic.  0 (constant slices) -> +inf: nothing is summed apart
  if (ok && rankv == (nv - 1) / 2) tau[b] = var > 0.0 ? (float)fmin(kStatsBig * var, 3.0e38) : INFINITY;
}

// xp[b][c][t] = x_b[Nc*t + c] - m0[b] for the samples the windows touch, 0 beyond.
__global__ __launch_bounds__(256) void demux_kernel(const float* __restrict__ x, int64_t M,
                                                    int64_t chunk_stride, int Nc, int64_t Tvalid,
                                                    int64_t Tpad, const float* __restrict__ m0f,
                                                    float* __restrict__ xp,
                                                    const int64_t* __restrict__ ctab, int n) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float* xb = chunk_base(x, chunk_stride, ctab, b, M);
  // ragged batch: the samples the chunk's OWN windows touch, exactly as when it is scanned alone
  if (ctab != nullptr) Tvalid = M > n ? (M - n) / Nc + n / Nc : 0;
  float* dst = xp + ((int64_t)b * Nc + c) * Tpad;
  const float m0 = m0f[b];
  // four consecutive samples per thread: four gathers in flight, one 16-byte store where the plane
  // allows it
  const bool vec = ((Tpad & 3) == 0) && ((reinterpret_cast<uintptr_t>(xp) & 15) == 0);
  for (int64_t t = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); t < Tpad;
       t += 4 * (int64_t)gridDim.x * blockDim.x) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t src = (t + k) * Nc + c;
      v[k] = (t + k < Tvalid && src < M) ? xb[src] : m0;
    }
    if (vec && t + 3 < Tpad) {
      *reinterpret_cast<float4*>(dst + t) = make_float4(v[0] - m0, v[1] - m0, v[2] - m0, v[3] - m0);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (t + k < Tpad) dst[t + k] = v[k] - m0;
    }
  }
}

// The same for NC channels with 16-byte accesses on both sides: a thread takes four consecutive sample
// frames (4 NC consecutive floats of the stream: NC 16-byte loads, every byte read once) and stores one
// 16-byte group to each plane.  (The gather above reads every 64-byte line NC times, from NC workgroups:
// 53.8 us per station-day against 48.9.)  Requires a 16-byte aligned chunk base; frames at the ragged end of
// the chunk go element by element.
template <int NC>
__global__ __launch_bounds__(256) void demux_vec_kernel(const float* __restrict__ x, int64_t M,
                                                        int64_t chunk_stride, int64_t Tvalid, int64_t Tpad,
                                                        const float* __restrict__ m0f, float* __restrict__ xp,
                                                        const int64_t* __restrict__ ctab, int n) {
  const int b = blockIdx.z;
  const float* xb = chunk_base(x, chunk_stride, ctab, b, M);
  if (ctab != nullptr) Tvalid = M > n ? (M - n) / NC + n / NC : 0;
  float* dst = xp + (int64_t)b * NC * Tpad;
  const float m0 = m0f[b];
  const bool aligned = (reinterpret_cast<uintptr_t>(xb) & 15) == 0;
  typedef float fv4 __attribute__((ext_vector_type(4)));
  for (int64_t t = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); t < Tpad;
       t += 4 * (int64_t)gridDim.x * blockDim.x) {
    float v[4 * NC];
    if (aligned && t + 3 < Tvalid && (t + 4) * NC <= M) {
      const fv4* src = reinterpret_cast<const fv4*>(xb + t * NC);
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const fv4 w = __builtin_nontemporal_load(src + q);
        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int64_t src = (t + k) * NC + c;
          v[k * NC + c] = (t + k < Tvalid && src < M) ? xb[src] : m0;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      float* d = dst + (int64_t)c * Tpad + t;
      if (t + 3 < Tpad) {
        *reinterpret_cast<float4*>(d) = make_float4(v[c] - m0, v[NC + c] - m0, v[2 * NC + c] - m0, v[3 * NC + c] - m0);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (t + k < Tpad) d[k] = v[k * NC + c] - m0;
      }
    }
  }
}

// split-bf16 engine: 8 shifted hi / lo bf16 copies of every channel plane,
// g[b][c][h][m][v] = 16-byte slot holding x_c[8 v + m .. 8 v + m + 7] (0 beyond Tpad)
__device__ inline unsigned short f32_to_bf16_rne(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);
}
__global__ __launch_bounds__(256) void demux16_kernel(const float* __restrict__ xp, int64_t Tpad,
                                                      int64_t V, i32x4* __restrict__ g) {
  const int plane = blockIdx.y;                 // b * Nc + c
  const float* x = xp + (int64_t)plane * Tpad;
  i32x4* gp = g + (int64_t)plane * 16 * V;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 8 * V;
       e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / V);
    const int64_t v = e - (int64_t)m * V;
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t t = 8 * v + m + j;
      const float val = t < Tpad ? x[t] : 0.f;
      hi[j] = f32_to_bf16_rne(val);
      const float hf = __uint_as_float((unsigned)hi[j] << 16);
      lo[j] = f32_to_bf16_rne(val - hf);
    }
    i32x4 H, L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      H[j] = (int)((unsigned)hi[2 * j] | ((unsigned)hi[2 * j + 1] << 16));
      L[j] = (int)((unsigned)lo[2 * j] | ((unsigned)lo[2 * j + 1] << 16));
    }
    gp[(int64_t)m * V + v] = H;
    gp[(int64_t)(8 + m) * V + v] = L;
  }
}

}  // namespace

// shared with dtx_xcorr.hip: build the shifted bf16 copies of `nplanes` channel planes
int dtx_launch_demux16(dtx_ctx* ctx, const float* xp, int64_t Tpad, int nplanes, void* g16) {
  const int64_t V = Tpad / 8;
  dim3 grid((unsigned)std::min<int64_t>((8 * V + 255) / 256, 8192), (unsigned)nplanes);
  hipLaunchKernelGGL(demux16_kernel, grid, dim3(256), 0, ctx->stream, xp, Tpad, V, (i32x4*)g16);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

namespace {

struct DsArgs {
  const float* xp;      // [nchunk][Nc][Tpad] channel planes (zero padded)
  const float* ut;      // [Nc*Lp][KT] transposed, channel-planar basis tile
  const float* srow;    // [KT] row sums S_k
  const int32_t* doff;  // [nd+1] detector row offsets (local to this K-tile)
  const float* mean;    // [nchunk][Ppad]
  const float* den;     // [nchunk][Ppad]  n/(n-1) * sum (w - mu)^2
  float* ds;            // [nchunk][D][ldp]
  const uint16_t* ut16; // split-bf16 engine: basis image [Nc][Lp/8][2][KT][8]
  const i32x4* g16;     // split-bf16 engine: shifted stream copies [nchunk][Nc][2][8][V]
  int64_t V;
  int64_t Tpad, Ppad, ldp, P;
  int Nc, Lp, D, det0, nd, nptiles;
  int ldk;              // row stride of ut (the K-tile's KT; a repair sub-tile reads a slice of its rows)
  // repair behind the frequency-domain engine: a window is stored only when its energy lies below lkeepv -- that
  // engine's rounding is relative to the block's loudest content, so at the loud windows themselves (a dominant
  // event) its own value is the more accurate of the two; +inf: every window is stored
  float lkeepv;
};

// mean correction, per-detector energy (membership MFMA on the accumulator tile),
// normalisation and store; shared by the float32 and the split-bf16 kernels (same
// accumulator layout).  `lds` must hold 2*KT + 64 floats and be free.
template <int RM, int TN, int NW = 8>
__device__ __forceinline__ void ds_epilogue(f32x4 (&acc)[RM][TN], const DsArgs& a, int chunk,
                                            int64_t p0, float* lds) {
  constexpr int KT = 16 * RM;
  constexpr int kThreads = 64 * NW;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int col0 = wave * 16 * TN;
  float* srow_s = lds;
  int* doff_s = reinterpret_cast<int*>(lds + KT);
  for (int v = tid; v < KT; v += kThreads) srow_s[v] = a.srow[v];
  for (int v = tid; v <= a.nd; v += kThreads) doff_s[v] = a.doff[v];
  __syncthreads();

  float mu[TN], dn[TN];
  const int64_t pcol = p0 + col0 + i;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int64_t p = pcol + 16 * t;  // < Ppad by construction
    mu[t] = a.mean[(int64_t)chunk * a.Ppad + p];
    dn[t] = a.den[(int64_t)chunk * a.Ppad + p];
  }
  float sr[RM][4];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) sr[r][e] = srow_s[16 * r + 4 * kq + e];

  // DS tile of 16 detectors x 16 windows = G[16 x rows] * V[rows x 16] with
  // V = (Y - S mu)^2 taken straight from the accumulator registers: for a fixed
  // register e the four lane groups kq hold rows 16r+4kq+e, which is exactly the
  // k-slab a 16x16x4 MFMA B operand needs.
  const int ndb = (a.nd + 15) >> 4;
  for (int db = 0; db < ndb; ++db) {
    const int d = 16 * db + i;
    const int dlo = (d < a.nd) ? doff_s[d] : 0;
    const int dhi = (d < a.nd) ? doff_s[d + 1] : 0;
    const int rlo = doff_s[16 * db];
    const int rhi = doff_s[min(16 * db + 16, a.nd)];
    f32x4 out[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) out[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      if (16 * r < rhi && 16 * r + 16 > rlo) {  // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * r + 4 * kq + e;
          const float g = (row >= dlo && row < dhi) ? 1.f : 0.f;
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            const float y = acc[r][t][e] - sr[r][e] * mu[t];
            out[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(g, y * y, out[t], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int64_t p = pcol + 16 * t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int det = 16 * db + 4 * kq + e;
        if (det < a.nd && p < a.P && dn[t] < a.lkeepv) {
          float v = (dn[t] > 0.f) ? out[t][e] / dn[t] : 0.f;
          if (__builtin_isinf(v)) v = 0.f;
          a.ds[((int64_t)chunk * a.D + a.det0 + det) * a.ldp + p] = v;
        }
      }
    }
  }
}

template <int RM, int TN, bool TAIL>
__device__ __forceinline__ void ds_mfma_tile(const DsArgs& a, int chunk, int64_t p0, float* lds) {
  constexpr int KT = 16 * RM;
  f32x4 acc[RM][TN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // odd RM: LDS row stride == KT == global row stride -> LDS-DMA staging
  hankel_mainloop<RM, TN, (RM & 1) != 0, TAIL>(acc, a.ut, KT, a.Lp, a.xp + (int64_t)chunk * a.Nc * a.Tpad + p0, a.Tpad,
                          a.Nc, 0, a.Lp, lds);

  ds_epilogue<RM, TN>(acc, a, chunk, p0, lds);
}

template <int RM, int TN, bool TAIL>
__global__ __launch_bounds__(kThreads, 2) void ds_mfma_kernel(const DsArgs a) {
  constexpr int PT = 128 * TN;
  __shared__ __attribute__((aligned(16))) float lds[lds_floats(RM)];
  const int chunk = blockIdx.x / a.nptiles;
  const int pt = blockIdx.x - chunk * a.nptiles;
  ds_mfma_tile<RM, TN, TAIL>(a, chunk, (int64_t)pt * PT, lds);
}

// Repair pass.  What the dynamic-range guards list comes here as TASKS {chunk, column tile of kRepairPT windows,
// the window energy below which a window is stored}; a task's rows are taken in sub-tiles of RM row blocks
// that hold whole detectors (dtxplan::plan_repair_subtiles), one workgroup per (task, sub-tile): one listed
// column tile spreads over nsub workgroups, a listed block over nsub * (tiles with a window in need).  (Round
// 5 walked (listed block) x (22 column tiles, all rows): 22 workgroups busy for 1.6 ms where the chip needs 0.13.)
// The launch is sized for the chip and finds an empty list on ordinary data.
struct RepairArgs {
  DsArgs a;             // the K-tile's images; ut / srow / doff / det0 / nd are set per sub-tile
  const int4* tasks;
  const int* ntask;
  const int4* sub;      // [nsub] {first row block, first detector, detectors, offset into sdoff}
  const int32_t* sdoff;
  int nsub;
};
// contraction indices per step of a task: a task's few MFMAs per step (RM x 1 tiles per wave) hide nothing, every
// step costs a round trip to memory -- a third to a half of the steps with deeper ones (row groups of one or two row
// blocks, i.e. detectors of up to 32 vectors; wider groups stage kBQ rows like the engine)
constexpr int repair_bq(int rm) { return rm == 1 ? 256 : (rm == 2 ? 128 : kBQ); }
constexpr int repair_lds_floats(int rm) {
  const int loop = 2 * (repair_bq(rm) * kts_for(rm) + kRepairPT + repair_bq(rm));
  const int epi = 2 * 16 * rm + 64;
  return loop > epi ? loop : epi;
}
template <int RM>
__global__ __launch_bounds__(kThreads, 2) void ds_mfma_task_kernel(const RepairArgs r) {
  constexpr int TN = kRepairPT / 128;
  __shared__ __attribute__((aligned(16))) float lds[repair_lds_floats(RM)];
  const int nwork = r.ntask[0] * r.nsub;
  for (int e = blockIdx.x; e < nwork; e += gridDim.x) {
    const int4 tk = r.tasks[e / r.nsub];
    const int4 sb = r.sub[e % r.nsub];
    DsArgs b = r.a;
    b.ut += 16 * sb.x; b.srow += 16 * sb.x;
    b.doff = r.sdoff + sb.w; b.det0 += sb.y; b.nd = sb.z;
    b.lkeepv = __int_as_float(tk.z);
    const int chunk = tk.x;
    const int64_t p0 = (int64_t)tk.y * kRepairPT;
    f32x4 acc[RM][TN];
#pragma unroll
    for (int q = 0; q < RM; ++q)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[q][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int BQ = repair_bq(RM);
    const float* xw = b.xp + (int64_t)chunk * b.Nc * b.Tpad + p0;
    const int jdeep = b.Lp / BQ * BQ;                  // Lp is a multiple of kBQ: deep steps, then the rest in kBQ steps
    hankel_mainloop<RM, TN, false, false, BQ>(acc, b.ut, b.ldk, b.Lp, xw, b.Tpad, b.Nc, 0, jdeep, lds);
    if constexpr (BQ != kBQ) {
      if (jdeep < b.Lp) {
        __syncthreads();
        hankel_mainloop<RM, TN, false, false>(acc, b.ut, b.ldk, b.Lp, xw, b.Tpad, b.Nc, jdeep, b.Lp, lds);
      }
    }
    ds_epilogue<RM, TN>(acc, b, chunk, p0, lds);
    __syncthreads();
  }
}

// split-bf16 engine (dtx_hankel_bf16.h): same tile, three bf16 MFMAs per product
template <int RM, int TN, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void ds_mfma_kernel_s(const DsArgs a) {
  constexpr int PT = 16 * TN * NW;
  __shared__ __attribute__((aligned(16))) char lds[lds_bytes_s(RM)];
  const int chunk = blockIdx.x / a.nptiles;
  const int pt = blockIdx.x - chunk * a.nptiles;
  const int64_t p0 = (int64_t)pt * PT;
  f32x4 acc[RM][TN];
#pragma unroll
  for (int r = 0; r < RM; ++r)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  hankel_mainloop_bf16x3<RM, TN, NW>(acc, a.ut16, a.Lp, a.g16 + (int64_t)chunk * a.Nc * 16 * a.V,
                                     a.V, p0 / 8, a.Nc, 0, a.Lp, lds);
  ds_epilogue<RM, TN, NW>(acc, a, chunk, p0, reinterpret_cast<float*>(lds));
}

template <int RM>
int launch_tile_s(dtx_ctx* ctx, const DsArgs& a, int nchunk) {
  constexpr int NW = nw_s(RM);
  constexpr int TN = tnw_s(RM);
  constexpr int PT = 16 * TN * NW;
  DsArgs b = a;
  b.nptiles = (int)((a.P + PT - 1) / PT);
  dim3 grid((unsigned)(b.nptiles * nchunk));
  hipLaunchKernelGGL((ds_mfma_kernel_s<RM, TN, NW>), grid, dim3(64 * NW), 0, ctx->stream, b);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

template <int RM>
int launch_tile(dtx_ctx* ctx, const DsArgs& a, int nchunk, bool tail) {
  if (a.ut16) return launch_tile_s<RM>(ctx, a, nchunk);
  constexpr int TN = tn_for(RM);
  constexpr int PT = 128 * TN;
  DsArgs b = a;
  b.nptiles = (int)((a.P + PT - 1) / PT);
  dim3 grid((unsigned)(b.nptiles * nchunk));
  if constexpr (RM >= 2) {
    if (tail) {
      hipLaunchKernelGGL((ds_mfma_kernel<RM, TN, true>), grid, dim3(kThreads), 0, ctx->stream, b);
      DTX_HIP(ctx, hipGetLastError());
      return DTX_OK;
    }
  }
  hipLaunchKernelGGL((ds_mfma_kernel<RM, TN, false>), grid, dim3(kThreads), 0, ctx->stream, b);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

// rows: real rows of the K-tile; the last row block takes the half-cost 4x4 path
// when it holds at most 8 of them
int launch_tile_rm(dtx_ctx* ctx, int rm, int rows, const DsArgs& a_, int nchunk) {
  const bool tail_ = ((rows - 1) % 16) < 8;
  const DsArgs& a = a_;
  switch (rm) {
    case 1: return launch_tile<1>(ctx, a, nchunk, tail_);
    case 2: return launch_tile<2>(ctx, a, nchunk, tail_);
    case 3: return launch_tile<3>(ctx, a, nchunk, tail_);
    case 4: return launch_tile<4>(ctx, a, nchunk, tail_);
    case 5: return launch_tile<5>(ctx, a, nchunk, tail_);
    case 6: return launch_tile<6>(ctx, a, nchunk, tail_);
    case 7: return launch_tile<7>(ctx, a, nchunk, tail_);
    case 8: return launch_tile<8>(ctx, a, nchunk, tail_);
    case 9: return launch_tile<9>(ctx, a, nchunk, tail_);
    case 10: return launch_tile<10>(ctx, a, nchunk, tail_);
    case 11: return launch_tile<11>(ctx, a, nchunk, tail_);
    case 12: return launch_tile<12>(ctx, a, nchunk, tail_);
    case 13: return launch_tile<13>(ctx, a, nchunk, tail_);
  }
  return dtx_fail(ctx, DTX_E_ARG, "bad row-block count %d", rm);
}

// ---------------------------------------------------------------- basis images (device side)
// S_k = sum_q U[k][q] in float64, stored float32: one workgroup per row
__global__ __launch_bounds__(256) void basis_rowsum_kernel(const float* __restrict__ U, int n,
                                                           float* __restrict__ srow) {
  __shared__ double red[4];
  const float* u = U + (size_t)blockIdx.x * n;
  double s = 0.0;
  for (int q = threadIdx.x; q < n; q += 256) s += (double)u[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) srow[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// One K-tile: ut[(c Lp + j) KT + k] = U[row0 + k][Nc j + c] (0 for k >= rows or j >= L) and the
// split-bf16 image ut16[c][j / 8][h][KT][8] (hi = bf16(v), lo = bf16(v - hi); the plain casts
// round to nearest even and keep a NaN a NaN).  Reads go through an LDS tile so that both the
// row-major source and the k-fastest destination move whole lines.
__global__ __launch_bounds__(256) void basis_tile_kernel(const float* __restrict__ U, int n, int Nc,
                                                         int L, int Lp, int row0, int rows, int KT,
                                                         float* __restrict__ ut,
                                                         uint16_t* __restrict__ ut16) {
  __shared__ float tile[16][65];
  const int c = blockIdx.z, j0 = blockIdx.x * 64, k0 = blockIdx.y * 16;
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int k = e >> 6, j = e & 63;
    float v = 0.f;
    if (k0 + k < rows && j0 + j < L) v = U[(size_t)(row0 + k0 + k) * n + (size_t)(j0 + j) * Nc + c];
    tile[k][j] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 16 * 64; e += 256) {
    const int j = e >> 4, k = e & 15;
    const float v = tile[k][j];
    const size_t jj = (size_t)j0 + j;
    ut[((size_t)c * Lp + jj) * KT + k0 + k] = v;
    const __hip_bfloat16 hi = __float2bfloat16(v);
    const __hip_bfloat16 lo = __float2bfloat16(v - __bfloat162float(hi));
    const size_t qb = (size_t)c * (Lp / 8) + jj / 8;
    ut16[((qb * 2 + 0) * KT + k0 + k) * 8 + jj % 8] = __builtin_bit_cast(uint16_t, hi);
    ut16[((qb * 2 + 1) * KT + k0 + k) * 8 + jj % 8] = __builtin_bit_cast(uint16_t, lo);
  }
}

// cmax = max over detectors d and columns q of sum_{k in d} U[k][q]^2: the largest share of a detector's
// statistic a single sample can carry (split_guard_kernel).  One thread per (column, detector); the maximum of
// non-negative floats is taken on their bit patterns.
__global__ __launch_bounds__(256) void basis_colmax_kernel(const float* __restrict__ U, int n,
                                                           const int32_t* __restrict__ doff,
                                                           float* __restrict__ cmax) {
  const int d = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (q < n)
    for (int k = doff[d]; k < doff[d + 1]; ++k) {
      const float u = U[(size_t)k * n + q];
      s = fmaf(u, u, s);
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = fmaxf(s, __shfl_xor(s, off, 64));
  if ((threadIdx.x & 63) == 0 && s > 0.f) atomicMax(reinterpret_cast<unsigned*>(cmax), __float_as_uint(s));
}

// Guard of the split-bf16 contraction (ds_mfma_kernel_s).  Its products carry a relative error of ~2^-16
// each; over a window they average out -- unless ONE sample carries the window (a spike, a glitch): then
// delta DS ~ 2 * 2^-16 * (sum_k U[k][j]^2) * x_j^2 / den, up to 3e-5 when that sample's share of the statistic is
// of order one.  A block of kSplitGuardV windows is listed for the float32 engine (whose products are exact
// fused multiply-adds) when cmax * max x^2 over the samples its windows read exceeds kSplitGuardTau times the
// smallest live window energy of the block: the listed blocks become repair tasks (ds_mfma_task_kernel) that run
// right behind the split launch, like the frequency-domain engine's.  On ordinary data (no sample above a few sigma, orthonormal
// bases of realistic length: cmax ~ K / n) the list is empty.
__global__ __launch_bounds__(256) void split_guard_kernel(const float* __restrict__ xp, int64_t Tpad, int Nc,
                                                          int L, const float* __restrict__ den, int64_t Ppad,
                                                          int64_t P, int V, int nblk,
                                                          const float* __restrict__ cmax, float tau,
                                                          int* __restrict__ count, int4* __restrict__ tasks,
                                                          const int* __restrict__ pb) {
  __shared__ float red[8];
  const int cb = blockIdx.x, tid = threadIdx.x;
  const int chunk = cb / nblk, blk = cb - chunk * nblk;
  if (pb != nullptr) P = min(P, (int64_t)pb[chunk]);     // ragged batch: the chunk's own windows
  const int64_t w0 = (int64_t)blk * V, w1 = min(w0 + V, P);
  if (w0 >= w1) return;
  const float* d = den + (int64_t)chunk * Ppad;
  float mn = 3.0e38f, mx = 0.f;
  for (int64_t p = w0 + tid; p < w1; p += 256) {
    const float dn = d[p];
    if (dn > 0.f) mn = fminf(mn, dn);
  }
  const int64_t t1 = w1 + L - 1;                         // samples [w0, t1) of every channel plane
  for (int c = 0; c < Nc; ++c) {
    const float* x = xp + ((int64_t)chunk * Nc + c) * Tpad;
    for (int64_t t = w0 + tid; t < t1; t += 256) {
      const float v = x[t];
      mx = fmaxf(mx, v * v);
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
    // (mn stays 3e38 when no window of the block is live: nothing to repair; a NaN sample lists the block)
    if (!(cmax[0] * mx <= tau * mn)) {
      // every column tile of the block, every window stored (the float32 engine's value is the better one throughout)
      atomicAdd(count, 1);
      const int nt = (int)((w1 - w0 + kRepairPT - 1) / kRepairPT);
      const int at = atomicAdd(count + 1, nt);
      for (int q = 0; q < nt; ++q)
        tasks[at + q] = make_int4(chunk, (int)(w0 / kRepairPT) + q, __float_as_int(INFINITY), cb);
    }
  }
}

}  // namespace

template <int RM>
static void launch_repair_rm(dtx_ctx* ctx, const RepairArgs& r, unsigned grid) {
  hipLaunchKernelGGL((ds_mfma_task_kernel<RM>), dim3(grid), dim3(kThreads), 0, ctx->stream, r);
}

int dtx_launch_repair(dtx_ctx* ctx, const DtxRepairWs& ws, const float* xp, int64_t Tpad, const float* mean,
                      const float* den, int64_t Ppad, int64_t P, float* ds, int64_t ldp) {
  for (const DtxKTile& t : ctx->tiles) {
    RepairArgs r;
    DsArgs& a = r.a;
    a.ut16 = nullptr; a.g16 = nullptr; a.V = 0;
    a.xp = xp; a.ut = t.ut_dev; a.srow = t.srow_dev; a.doff = t.doff_dev;
    a.mean = mean; a.den = den; a.ds = ds;
    a.Tpad = Tpad; a.Ppad = Ppad; a.ldp = ldp; a.P = P;
    a.Nc = ctx->Nc; a.Lp = ctx->Lp; a.D = ctx->D; a.det0 = t.det0; a.nd = t.nd; a.nptiles = 0;
    a.ldk = 16 * t.rm; a.lkeepv = INFINITY;
    r.tasks = ws.tasks; r.ntask = ws.count + 1;
    r.sub = (const int4*)t.sub_dev; r.sdoff = t.sdoff_dev; r.nsub = t.nsub;
    // two workgroups per CU are resident; twice that many walk the list (a workgroup's tasks are of equal length)
    const unsigned grid = (unsigned)(4 * ctx->ncu);
    switch (t.rms) {
      case 1: launch_repair_rm<1>(ctx, r, grid); break;
      case 2: launch_repair_rm<2>(ctx, r, grid); break;
      case 3: launch_repair_rm<3>(ctx, r, grid); break;
      case 4: launch_repair_rm<4>(ctx, r, grid); break;
      case 5: launch_repair_rm<5>(ctx, r, grid); break;
      case 6: launch_repair_rm<6>(ctx, r, grid); break;
      case 7: launch_repair_rm<7>(ctx, r, grid); break;
      case 8: launch_repair_rm<8>(ctx, r, grid); break;
      case 9: launch_repair_rm<9>(ctx, r, grid); break;
      case 10: launch_repair_rm<10>(ctx, r, grid); break;
      case 11: launch_repair_rm<11>(ctx, r, grid); break;
      case 12: launch_repair_rm<12>(ctx, r, grid); break;
      case 13: launch_repair_rm<13>(ctx, r, grid); break;
      default: return dtx_fail(ctx, DTX_E_STATE, "repair pass: bad sub-tile height %d", t.rms);
    }
    DTX_HIP(ctx, hipGetLastError());
  }
  return DTX_OK;
}

extern "C" {

int64_t dtx_num_windows(int64_t M, int n, int Nc) {
  if (n <= 0 || Nc <= 0 || M < n) return 0;
  return (M - n) / Nc + 1;
}

// The two halves of dtx_basis_upload.  begin: checks, waits for the old images' users, and puts the copy of the
// stacked basis on the context's stream (asynchronous when U is device memory: a peer copy); finish: every image
// built on the device behind it.  dtx_basis_bcast runs `begin` on every receiver before any `finish`, so that the
// peer copies of all devices are in flight together.
static int basis_upload_begin(dtx_ctx* ctx, const float* U, const int32_t* det_off, int D, int n, int Nc) {
  if (!ctx) return DTX_E_ARG;
  if (!U || !det_off || D <= 0 || n <= 0 || Nc <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_basis_upload: null pointer or non-positive size");
  if (n % Nc != 0)
    return dtx_fail(ctx, DTX_E_ARG, "basis length n=%d is not a multiple of Nc=%d", n, Nc);
  if (n < 2) return dtx_fail(ctx, DTX_E_ARG, "basis length must be >= 2");
  if (D > 65535) return dtx_fail(ctx, DTX_E_ARG, "at most 65535 detectors per upload (got %d)", D);
  if (det_off[0] != 0) return dtx_fail(ctx, DTX_E_ARG, "det_off[0] must be 0");
  for (int d = 0; d < D; ++d) {
    const int k = det_off[d + 1] - det_off[d];
    if (k <= 0) return dtx_fail(ctx, DTX_E_ARG, "detector %d has %d basis vectors", d, k);
    if (k > kKtMax)
      return dtx_fail(ctx, DTX_E_ARG, "detector %d has %d basis vectors (max %d)", d, k, kKtMax);
  }
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the old images may still be in use
  if (ctx->trig_pending) DTX_HIP(ctx, hipEventSynchronize(ctx->ev_trig));   // (a trigger-time launch on its own stream reads U)
  const int Ktot = det_off[D];
  const int L = n / Nc;
  const int Lp = (int)dtx_round_up(L, kBQ);
  int rc;
  dtx_free_tiles_internal(ctx);
  if ((rc = dtx_reserve(ctx, ctx->U_buf, (size_t)Ktot * n * sizeof(float)))) return rc;
  if ((rc = dtx_reserve(ctx, ctx->srow_all, ((size_t)Ktot + 1) * sizeof(float)))) return rc;   // + cmax
  float* Ud = (float*)ctx->U_buf.p;
  hipPointerAttribute_t at;
  const bool on_device = hipPointerGetAttributes(&at, U) == hipSuccess &&
                         at.type == hipMemoryTypeDevice;
  (void)hipGetLastError();                           // a plain host pointer is not an error
  DTX_HIP(ctx, hipMemcpyAsync(Ud, U, (size_t)Ktot * n * sizeof(float), hipMemcpyDefault, ctx->stream));
  if (!on_device) DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the caller may reuse its buffer

  ctx->D = D; ctx->n = n; ctx->Nc = Nc; ctx->L = L; ctx->Lp = Lp; ctx->Ktot = Ktot;
  ctx->hdoff.assign(det_off, det_off + D + 1);
  return DTX_OK;
}

static int basis_upload_finish(dtx_ctx* ctx) {
  DtxDeviceGuard g(ctx->device);
  const int D = ctx->D, n = ctx->n, Nc = ctx->Nc, L = ctx->L, Lp = ctx->Lp, Ktot = ctx->Ktot;
  const int32_t* det_off = ctx->hdoff.data();
  float* Ud = (float*)ctx->U_buf.p;
  float* srow_all = (float*)ctx->srow_all.p;
  {   // detector row offsets on the device (dtx_trigger_stats; the tiles hold them tile-local)
    int rc2;
    if ((rc2 = dtx_reserve(ctx, ctx->doff_all, (size_t)(D + 1) * sizeof(int32_t)))) return rc2;
    DTX_HIP(ctx, hipMemcpy(ctx->doff_all.p, ctx->hdoff.data(), (size_t)(D + 1) * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(basis_rowsum_kernel, dim3((unsigned)Ktot), dim3(256), 0, ctx->stream,
                     (const float*)Ud, n, srow_all);
  DTX_HIP(ctx, hipGetLastError());
  // the split engine's guard constant (split_guard_kernel), behind the row sums in the same buffer
  DTX_HIP(ctx, hipMemsetAsync(srow_all + Ktot, 0, sizeof(float), ctx->stream));
  hipLaunchKernelGGL(basis_colmax_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)D), dim3(256), 0, ctx->stream,
                     (const float*)Ud, n, (const int32_t*)ctx->doff_all.p, srow_all + Ktot);
  DTX_HIP(ctx, hipGetLastError());

  // greedy packing of whole detectors into K-tiles of <= 208 rows; every image is built on the
  // device from the one copy of U (no host pass over the basis, no stream synchronisation)
  int d = 0;
  while (d < D) {
    DtxKTile t;
    t.det0 = d;
    const int row0 = det_off[d];
    int dend = d;
    while (dend < D && det_off[dend + 1] - row0 <= kKtMax) ++dend;
    t.nd = dend - d;
    t.rows = det_off[dend] - row0;
    t.rm = (t.rows + 15) / 16;
    const int KT = 16 * t.rm;
    const size_t nq = (size_t)Nc * Lp;
    std::vector<int32_t> doff(t.nd + 1);
    for (int e = 0; e <= t.nd; ++e) doff[e] = det_off[d + e] - row0;
    const size_t ut16_elems = (size_t)Nc * (Lp / 8) * 2 * KT * 8;
    DTX_HIP(ctx, hipMalloc(&t.ut16_dev, ut16_elems * sizeof(uint16_t)));
    DTX_HIP(ctx, hipMalloc(&t.ut_dev, nq * KT * sizeof(float)));
    DTX_HIP(ctx, hipMalloc(&t.srow_dev, KT * sizeof(float)));
    DTX_HIP(ctx, hipMalloc(&t.doff_dev, doff.size() * sizeof(int32_t)));
    ctx->tiles.push_back(t);  // pushed first so a failed call still gets freed
    hipLaunchKernelGGL(basis_tile_kernel, dim3((unsigned)(Lp / 64), (unsigned)t.rm, (unsigned)Nc),
                       dim3(256), 0, ctx->stream, (const float*)Ud, n, Nc, L, Lp, row0, t.rows, KT,
                       t.ut_dev, t.ut16_dev);
    DTX_HIP(ctx, hipGetLastError());
    DTX_HIP(ctx, hipMemsetAsync(t.srow_dev, 0, KT * sizeof(float), ctx->stream));
    DTX_HIP(ctx, hipMemcpyAsync(t.srow_dev, srow_all + row0, t.rows * sizeof(float),
                                hipMemcpyDeviceToDevice, ctx->stream));
    DTX_HIP(ctx, hipMemcpy(t.doff_dev, doff.data(), doff.size() * sizeof(int32_t),
                           hipMemcpyHostToDevice));   // fresh buffer, a few bytes: synchronous
    {   // the repair pass's row groups (host arithmetic: dtx_plan.h)
      const dtxplan::RepairSubPlan rp = dtxplan::plan_repair_subtiles(doff.data(), t.nd, t.rm);
      static_assert(sizeof(int4) == sizeof(dtxplan::I4), "sub-tile rows are int4");
      DtxKTile& tt = ctx->tiles.back();
      tt.rms = rp.rms; tt.nsub = (int)rp.sub.size();
      DTX_HIP(ctx, hipMalloc(&tt.sub_dev, rp.sub.size() * sizeof(int4)));
      DTX_HIP(ctx, hipMalloc(&tt.sdoff_dev, rp.sdoff.size() * sizeof(int32_t)));
      DTX_HIP(ctx, hipMemcpy(tt.sub_dev, rp.sub.data(), rp.sub.size() * sizeof(int4), hipMemcpyHostToDevice));
      DTX_HIP(ctx, hipMemcpy(tt.sdoff_dev, rp.sdoff.data(), rp.sdoff.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    d = dend;
  }
  return DTX_OK;
}

int dtx_basis_upload(dtx_ctx* ctx, const float* U, const int32_t* det_off, int D, int n,
                     int Nc) {
  const int rc = basis_upload_begin(ctx, U, det_off, D, n, Nc);
  return rc ? rc : basis_upload_finish(ctx);
}

int dtx_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int dtx_device_pci_bus_id(int device, char* buf, int len) {
  if (!buf || len < 16) return DTX_E_ARG;
  buf[0] = 0;
  return hipDeviceGetPCIBusId(buf, len, device) == hipSuccess ? DTX_OK : DTX_E_HIP;
}

int dtx_basis_bcast(dtx_ctx* const* ctxs, int nctx) {
  if (!ctxs || nctx <= 0 || !ctxs[0]) return DTX_E_ARG;
  dtx_ctx* src = ctxs[0];
  if (src->tiles.empty()) return dtx_fail(src, DTX_E_STATE, "dtx_basis_bcast: no basis uploaded on the first context");
  for (int i = 1; i < nctx; ++i)
    if (!ctxs[i]) return dtx_fail(src, DTX_E_ARG, "dtx_basis_bcast: context %d is NULL", i);
  {
    DtxDeviceGuard g(src->device);
    DTX_HIP(src, hipStreamSynchronize(src->stream));     // the source copy of U is complete
  }
  // (1) every receiver's copy of the stacked basis on ITS stream: device-to-device copies (peer copies over xGMI
  // between devices), all in flight together -- xGMI is point to point, the source's links to different peers do not
  // share bandwidth; (2) every receiver's images behind its copy, each on its own device and stream; (3) one
  // wait per receiver at the end.  (Round 5 ran copy, builds and a stream synchronisation device after device.)
  for (int i = 1; i < nctx; ++i) {
    dtx_ctx* c = ctxs[i];
    if (c == src) continue;
    const int rc = basis_upload_begin(c, (const float*)src->U_buf.p, src->hdoff.data(), src->D, src->n, src->Nc);
    if (rc) return dtx_fail(src, rc, "dtx_basis_bcast: context %d: %s", i, c->err.c_str());
  }
  for (int i = 1; i < nctx; ++i) {
    dtx_ctx* c = ctxs[i];
    if (c == src) continue;
    const int rc = basis_upload_finish(c);
    if (rc) return dtx_fail(src, rc, "dtx_basis_bcast: context %d: %s", i, c->err.c_str());
    const int rc2 = dtx_set_engine(c, src->engine);
    if (rc2) return rc2;
  }
  for (int i = 1; i < nctx; ++i) {
    dtx_ctx* c = ctxs[i];
    if (c == src) continue;
    DtxDeviceGuard g(c->device);
    DTX_HIP(c, hipStreamSynchronize(c->stream));
  }
  return DTX_OK;
}

// channel planes of every chunk: the 16-byte kernel where the planes allow it (Tpad % 4 == 0, aligned workspace)
static void launch_demux(dtx_ctx* ctx, hipStream_t st, const float* x_dev, int64_t M, int64_t chunk_stride, int Nc,
                         int64_t Tvalid, int64_t Tpad, const float* m0f, float* xp,
                         const int64_t* ctab, int n, int nchunk) {
  const bool vec = (Tpad & 3) == 0 && (reinterpret_cast<uintptr_t>(xp) & 15) == 0;
  const dim3 vgrid((unsigned)std::min<int64_t>((Tpad + 1023) / 1024, 4096), 1u, (unsigned)nchunk);
#define DTX_DEMUX_VEC(NC)                                                                                       \
  hipLaunchKernelGGL(demux_vec_kernel<NC>, vgrid, dim3(256), 0, st, x_dev, M, chunk_stride, Tvalid, Tpad, m0f,  \
                     xp, ctab, n)
  if (vec && Nc == 1) { DTX_DEMUX_VEC(1); return; }
  if (vec && Nc == 2) { DTX_DEMUX_VEC(2); return; }
  if (vec && Nc == 3) { DTX_DEMUX_VEC(3); return; }
  if (vec && Nc == 4) { DTX_DEMUX_VEC(4); return; }
  if (vec && Nc == 6) { DTX_DEMUX_VEC(6); return; }
#undef DTX_DEMUX_VEC
  const dim3 dgrid((unsigned)std::min<int64_t>((Tpad + 1023) / 1024, 4096), (unsigned)Nc, (unsigned)nchunk);
  hipLaunchKernelGGL(demux_kernel, dgrid, dim3(256), 0, st, x_dev, M, chunk_stride, Nc, Tvalid, Tpad, m0f, xp,
                     ctab, n);
}

// The post pass of a scan: on the context's stream, or -- when the caller handed in a stream for it
// (dtx_ctx_set_post_stream) -- on that stream behind an event, so that the NEXT scan's preparation, which the caller
// has queued meanwhile, runs beside it: the pass reads the whole DS tensor at the memory's rate (150 us at configs[1])
// while offsets, channel planes, block spectra and window statistics (270 us) are chains of latencies.  (The maxima it
// accumulates into are cleared by THIS call's offset pass, in front of it on the context's stream: a caller must not
// hand the next call the result buffers of a call whose results it has not read -- it would lose them either way.)
static int launch_post(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp, int nchunk, const DtxPost& po,
                       const int* pb) {
  if (!ctx->post_stream) {
    ctx->post_on_side = false;
    return dtx_post_launch(ctx, ds_dev, P, ldp, nchunk, ctx->D, po, pb);
  }
  DTX_HIP(ctx, hipEventRecord(ctx->ev_post_ready, ctx->stream));
  DTX_HIP(ctx, hipStreamWaitEvent(ctx->post_stream, ctx->ev_post_ready, 0));
  const int rc = dtx_post_launch(ctx, ds_dev, P, ldp, nchunk, ctx->D, po, pb, ctx->post_stream);
  if (rc) return rc;
  DTX_HIP(ctx, hipEventRecord(ctx->ev_post_done, ctx->post_stream));
  ctx->post_pending = true;
  ctx->post_on_side = true;
  return DTX_OK;
}

// post != NULL: also the histogram / maximum / candidate pass over DS (dtx_ds_post semantics)
// ctab != NULL: a ragged batch (dtx_subspace_scan_ragged): chunk b = ctab[2 b + 1] <= M samples at x_dev + ctab[2 b]
static int subspace_ds_impl(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                            int nchunk, float* ds_dev, int64_t ldp, const DtxPost* post,
                            const int64_t* ctab = nullptr) {
  if (!ctx) return DTX_E_ARG;
  if (ctx->tiles.empty()) return dtx_fail(ctx, DTX_E_STATE, "dtx_subspace_ds: no basis uploaded");
  if (!x_dev || !ds_dev || nchunk <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_subspace_ds: null pointer or nchunk <= 0");
  const int n = ctx->n, Nc = ctx->Nc;
  const int64_t P = dtx_num_windows(M, n, Nc);
  // detect.py:262-266: a block not longer than the template is rejected
  if (M <= n || P <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "data block of %lld samples is not longer than the basis (%d)",
                    (long long)M, n);
  if (ldp < P) return dtx_fail(ctx, DTX_E_ARG, "ldp=%lld < P=%lld", (long long)ldp, (long long)P);
  if (nchunk > 65535) return dtx_fail(ctx, DTX_E_ARG, "nchunk > 65535");
  if (P >= (int64_t)1 << 31) return dtx_fail(ctx, DTX_E_ARG, "too many windows per block");
  DtxDeviceGuard g(ctx->device);

  const int64_t Ppad = dtx_round_up(P, 1024);
  const int64_t Tpad = Ppad + ctx->Lp + 64;
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_xp, (size_t)nchunk * Nc * Tpad * sizeof(float)))) return rc;
  // + DTX_FFT_N: the frequency-domain engine reads whole blocks of window statistics past P
  if ((rc = dtx_reserve(ctx, ctx->ws_mean, ((size_t)nchunk * Ppad + DTX_FFT_N) * sizeof(float)))) return rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_den, ((size_t)nchunk * Ppad + DTX_FFT_N) * sizeof(float)))) return rc;
  float* xp = (float*)ctx->ws_xp.p;
  float* mean = (float*)ctx->ws_mean.p;
  float* den = (float*)ctx->ws_den.p;

  const int64_t Tvalid = P + ctx->L - 1;
  // block offsets (one pass over the stream + a 24-block reduction)
  // (slices of at least 16 samples: a record of a few hundred samples still has enough of them for the median to
  // ignore a spike -- with M / 256 a chunk under 768 samples had one or two slices, and its level followed the spike)
  const int nsl = (int)std::max<int64_t>(1, std::min<int64_t>(kMeanSlices, M / 16));
  const int nsb = (int)((P + kStatsPB - 1) / kStatsPB);     // blocks of the window statistics per chunk (their overflow flags)
  if ((rc = dtx_reserve(ctx, ctx->ws_csum,
                        (size_t)nchunk * (nsl * kCsum * sizeof(double) + 2 * sizeof(float) + sizeof(int) + nsb * sizeof(int)))))
    return rc;
  double* cslice = (double*)ctx->ws_csum.p;
  float* m0f = (float*)(cslice + (size_t)nchunk * nsl * kCsum);
  float* tauf = m0f + nchunk;                 // per chunk: the square from which a sample is summed apart (win_stats_kernel)
  // ragged batch: every chunk's own window count, for the passes that must not look past it
  int* pbuf = (int*)(tauf + nchunk);
  int* pb = ctab ? pbuf : nullptr;
  int* sover = pbuf + nchunk;                  // [nchunk][nsb]: blocks whose big samples overflowed the list (win_stats_kernel)
  const bool use_fft = (ctx->engine == DTX_ENGINE_FFT && dtx_fft_eligible(ctx->L)) ||
                       (ctx->engine == DTX_ENGINE_AUTO && ctx->L >= kFftAutoMinL && ctx->L <= kFftLongMaxL);
  // the guards' lists (DtxRepairWs): the frequency-domain engine lists blocks of its advance V, the split-bf16
  // contraction blocks of kSplitGuardV windows; both end as repair tasks of kRepairPT windows
  const size_t ntile = (size_t)nchunk * (size_t)(Ppad / kRepairPT);
  DtxRepairWs rws{};
  bool guarded = false;
  if (use_fft) {
    if (!ctx->fft.ok &&
        (rc = dtx_fft_basis_build(ctx, ctx->hdoff.data(), ctx->D, n, Nc)))
      return rc;
    const size_t nb = (size_t)nchunk * (size_t)((P + ctx->fft.V - 1) / ctx->fft.V);
    if ((rc = dtx_reserve(ctx, ctx->ws_flag, dtx_repair_ws_bytes(nb, ntile)))) return rc;
    rws = dtx_repair_ws_carve(ctx->ws_flag.p, nb, ntile);
    guarded = true;
  }
  // split-bf16 contraction: blocks of windows a single sample dominates go to the float32 engine
  // (split_guard_kernel); templates shorter than kSplitMinL samples per channel stay on the float32 engine:
  // with so few products per window the split products' 2^-17 errors do not average out (and the padding of
  // L to the 32-sample steps of the bf16 MFMA would dominate anyway)
  const bool split = !use_fft && ctx->engine != DTX_ENGINE_FP32 && ctx->L >= kSplitMinL;
  const int gnblk = split ? (int)((P + kSplitGuardV - 1) / kSplitGuardV) : 0;
  if (split) {
    const size_t nb = (size_t)nchunk * gnblk;
    if ((rc = dtx_reserve(ctx, ctx->ws_flag, dtx_repair_ws_bytes(nb, ntile)))) return rc;
    rws = dtx_repair_ws_carve(ctx->ws_flag.p, nb, ntile);    // (its counters are cleared by chunk_offset_final_kernel)
    guarded = true;
  }
  int* list = guarded ? rws.count : nullptr;
  ctx->guard_ran = guarded;
  hipLaunchKernelGGL(chunk_offset_kernel, dim3((unsigned)nsl, (unsigned)nchunk), dim3(256), 0, ctx->stream,
                     x_dev, M, chunk_stride, nsl, cslice, ctab);
  // (the final pass also zeroes the guard's list counter and the maxima the post pass accumulates into)
  hipLaunchKernelGGL(chunk_offset_final_kernel, dim3((unsigned)nchunk), dim3(64), 0, ctx->stream,
                     (const double*)cslice, nsl, m0f, tauf, ctab, M, n, Nc, pb, list,
                     post ? post->maxds : (float*)nullptr, ctx->D, sover, nsb);
  DTX_HIP(ctx, hipGetLastError());
  // pandas rolling_var(ddof=1) * n = n/(n-1) * sum (w - mu)^2   (detect.py:567-568)
  const double scale = (double)n / ((double)n - 1.0);
  const dim3 sgrid((unsigned)((P + kStatsPB - 1) / kStatsPB), (unsigned)nchunk);

  if (use_fft) {
    // The preparation reads the caller's multiplexed stream directly (rounds 1-2 wrote channel planes
    // first and read them back twice): the window statistics run on a side stream beside the block
    // spectra -- both are chains of memory latencies that leave most of the chip idle on their own.
    // channel planes (main stream), then the window statistics on the side stream beside the block
    // spectra: both are chains of memory latencies that leave most of the chip idle on their own
    launch_demux(ctx, ctx->stream, x_dev, M, chunk_stride, Nc, Tvalid, Tpad, (const float*)m0f, xp, ctab, n, nchunk);
    if ((rc = dtx_aux_fork(ctx))) return rc;
    hipLaunchKernelGGL(win_stats_kernel, sgrid, dim3(256), 0, ctx->aux, (const float*)xp, Tpad,
                       Nc, ctx->L, P, Ppad, scale, mean, den, (const float*)tauf, sover, 0);
    // (the second launch takes the blocks whose outlier list overflowed: none on ordinary data)
    hipLaunchKernelGGL(win_stats_kernel, sgrid, dim3(256), 0, ctx->aux, (const float*)xp, Tpad,
                       Nc, ctx->L, P, Ppad, scale, mean, den, (const float*)tauf, sover, 1);
    DTX_HIP(ctx, hipGetLastError());
    // Dynamic-range guard.  The transforms' float32 rounding is relative to the largest
    // amplitude in an 8192-sample block, not to the window: a block whose loudest window
    // holds more than kFftGuard^2 = 4e4 times the energy of its quietest one (a 200-sigma
    // burst) would put > 4e-6 of error on the quiet windows.  A small kernel lists such blocks
    // (dtx_fft_scan, behind the statistics on the side stream); isolated outlier samples are split off and the
    // block is scanned again without them (their products added directly), and where windows are still out of
    // range the float32 MFMA contraction -- whose error is local to a window -- recomputes them right behind the
    // scan (dtx_launch_repair): launches that find empty lists on ordinary data.
    if ((rc = dtx_fft_scan(ctx, xp, Tpad, mean, den, Ppad, P, nchunk, ds_dev, ldp, rws, pb))) return rc;
    if ((rc = dtx_launch_repair(ctx, rws, xp, Tpad, mean, den, Ppad, P, ds_dev, ldp))) return rc;
    if (post) return launch_post(ctx, ds_dev, P, ldp, nchunk, *post, pb);
    return DTX_OK;
  }

  launch_demux(ctx, ctx->stream, x_dev, M, chunk_stride, Nc, Tvalid, Tpad, (const float*)m0f, xp, ctab, n, nchunk);
  hipLaunchKernelGGL(win_stats_kernel, sgrid, dim3(256), 0, ctx->stream, (const float*)xp, Tpad,
                     Nc, ctx->L, P, Ppad, scale, mean, den, (const float*)tauf, sover, 0);
  hipLaunchKernelGGL(win_stats_kernel, sgrid, dim3(256), 0, ctx->stream, (const float*)xp, Tpad,
                     Nc, ctx->L, P, Ppad, scale, mean, den, (const float*)tauf, sover, 1);
  DTX_HIP(ctx, hipGetLastError());

  i32x4* g16 = nullptr;
  const int64_t V = Tpad / 8;
  // the timed region (dtx_last_kernel_ms, dtx_stream_wait_scan_kernel) of the split engine holds ALL of its passes:
  // the guard's pass over xp / den, the split image, the contraction and the float32 repair of the guard's list
  { const int rck = dtx_kev_next(ctx); if (rck) return rck; }
  if (split) DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
  if (split) {
    hipLaunchKernelGGL(split_guard_kernel, dim3((unsigned)(nchunk * gnblk)), dim3(256), 0, ctx->stream,
                       (const float*)xp, Tpad, Nc, ctx->L, (const float*)den, Ppad, P, kSplitGuardV, gnblk,
                       (const float*)ctx->srow_all.p + ctx->Ktot, kSplitGuardTau, rws.count, rws.tasks,
                       (const int*)pb);
    DTX_HIP(ctx, hipGetLastError());
  }
  if (split) {
    if ((rc = dtx_reserve(ctx, ctx->ws_g, (size_t)nchunk * Nc * 16 * V * sizeof(i32x4)))) return rc;
    g16 = (i32x4*)ctx->ws_g.p;
    if ((rc = dtx_launch_demux16(ctx, xp, Tpad, nchunk * Nc, g16))) return rc;
  }

  if ((rc = dtx_wait_post(ctx))) return rc;      // (an earlier scan's post pass may still be reading this DS tensor)
  {
    char nm[64];
    const int rm0 = ctx->tiles.front().rm;
    if (split) snprintf(nm, sizeof nm, "ds_mfma_kernel_s<%d>", rm0);
    else snprintf(nm, sizeof nm, "ds_mfma_kernel<%d,%d>", rm0, tn_for(rm0));
    ctx->last_kernel = nm;
  }
  if (!split) DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));
  for (const DtxKTile& t : ctx->tiles) {
    DsArgs a;
    a.ut16 = g16 ? t.ut16_dev : nullptr; a.g16 = g16; a.V = V;
    a.xp = xp; a.ut = t.ut_dev; a.srow = t.srow_dev; a.doff = t.doff_dev;
    a.mean = mean; a.den = den; a.ds = ds_dev;
    a.Tpad = Tpad; a.Ppad = Ppad; a.ldp = ldp; a.P = P;
    a.Nc = Nc; a.Lp = ctx->Lp; a.D = ctx->D; a.det0 = t.det0; a.nd = t.nd; a.nptiles = 0;
    a.ldk = 16 * t.rm; a.lkeepv = INFINITY;
    if ((rc = launch_tile_rm(ctx, t.rm, t.rows, a, nchunk))) return rc;
  }
  if (split)      // the guard's blocks again, on the float32 engine (launches that find an empty list on ordinary data)
    if ((rc = dtx_launch_repair(ctx, rws, xp, Tpad, mean, den, Ppad, P, ds_dev, ldp))) return rc;
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  dtx_kev_commit(ctx);
  if (post) return launch_post(ctx, ds_dev, P, ldp, nchunk, *post, pb);
  return DTX_OK;
}

int dtx_subspace_ds(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                    int nchunk, float* ds_dev, int64_t ldp) {
  return subspace_ds_impl(ctx, x_dev, M, chunk_stride, nchunk, ds_dev, ldp, nullptr);
}

int dtx_subspace_scan(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                      int nchunk, float* ds_dev, int64_t ldp, uint32_t* hist_dev,
                      float* maxds_dev, const float* thr_dev, uint64_t* cand_key_dev,
                      float* cand_val_dev, uint32_t cand_cap, uint32_t* ncand_dev) {
  if (!ctx) return DTX_E_ARG;
  if (nchunk > 65535 || ctx->D > 65535) return dtx_fail(ctx, DTX_E_ARG, "nchunk / D > 65535");
  DtxPost po;
  po.hist = hist_dev; po.maxds = maxds_dev; po.thr = thr_dev; po.cand_key = cand_key_dev;
  po.cand_val = cand_val_dev; po.cand_cap = cand_cap; po.ncand = ncand_dev;
  return subspace_ds_impl(ctx, x_dev, M, chunk_stride, nchunk, ds_dev, ldp, &po);
}

int dtx_subspace_scan_ragged(dtx_ctx* ctx, const float* x_dev, const int64_t* chunk_tab_dev, int64_t M,
                             int nchunk, float* ds_dev, int64_t ldp, uint32_t* hist_dev,
                             float* maxds_dev, const float* thr_dev, uint64_t* cand_key_dev,
                             float* cand_val_dev, uint32_t cand_cap, uint32_t* ncand_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!chunk_tab_dev) return dtx_fail(ctx, DTX_E_ARG, "dtx_subspace_scan_ragged: no chunk table");
  if (nchunk > 65535 || ctx->D > 65535) return dtx_fail(ctx, DTX_E_ARG, "nchunk / D > 65535");
  DtxPost po;
  po.hist = hist_dev; po.maxds = maxds_dev; po.thr = thr_dev; po.cand_key = cand_key_dev;
  po.cand_val = cand_val_dev; po.cand_cap = cand_cap; po.ncand = ncand_dev;
  return subspace_ds_impl(ctx, x_dev, M, 0, nchunk, ds_dev, ldp, &po, chunk_tab_dev);
}

const char* dtx_last_kernel_name(dtx_ctx* ctx) { return ctx ? ctx->last_kernel.c_str() : ""; }

int dtx_repair_counts(dtx_ctx* ctx, int32_t counts[4]) {
  if (!ctx) return DTX_E_ARG;
  if (!counts) return dtx_fail(ctx, DTX_E_ARG, "dtx_repair_counts: null pointer");
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  if (!ctx->ws_flag.p || !ctx->guard_ran) return DTX_OK;      // (the float32 engine runs unguarded)
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  DTX_HIP(ctx, hipMemcpy(counts, ctx->ws_flag.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost));
  counts[3] = 0;
  return DTX_OK;
}

int dtx_set_engine(dtx_ctx* ctx, int engine) {
  if (!ctx) return DTX_E_ARG;
  if (engine != DTX_ENGINE_FP32 && engine != DTX_ENGINE_BF16X3 && engine != DTX_ENGINE_FFT &&
      engine != DTX_ENGINE_AUTO)
    return dtx_fail(ctx, DTX_E_ARG, "unknown engine %d", engine);
  ctx->engine = engine;
  return DTX_OK;
}

int dtx_subspace_ds_host(dtx_ctx* ctx, const float* x_host, int64_t M, float* ds_host) {
  if (!ctx) return DTX_E_ARG;
  if (ctx->tiles.empty()) return dtx_fail(ctx, DTX_E_STATE, "no basis uploaded");
  if (!x_host || !ds_host) return dtx_fail(ctx, DTX_E_ARG, "null pointer");
  const int64_t P = dtx_num_windows(M, ctx->n, ctx->Nc);
  if (M <= ctx->n || P <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "data block of %lld samples is not longer than the basis (%d)",
                    (long long)M, ctx->n);
  DtxDeviceGuard g(ctx->device);
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc, (size_t)M * sizeof(float)))) return rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc2, (size_t)ctx->D * P * sizeof(float)))) return rc;
  DTX_HIP(ctx, hipMemcpyAsync(ctx->ws_misc.p, x_host, (size_t)M * sizeof(float),
                              hipMemcpyHostToDevice, ctx->stream));
  if ((rc = dtx_subspace_ds(ctx, (const float*)ctx->ws_misc.p, M, M, 1, (float*)ctx->ws_misc2.p, P)))
    return rc;
  DTX_HIP(ctx, hipMemcpyAsync(ds_host, ctx->ws_misc2.p, (size_t)ctx->D * P * sizeof(float),
                              hipMemcpyDeviceToHost, ctx->stream));
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return DTX_OK;
}

}  // extern "C"
