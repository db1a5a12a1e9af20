// Post-processing of the detection statistic on the device: histogram
// (detex/detect.py:178-181), block maximum (detect.py:275) and the list of
// samples at or above the detector threshold that feeds the reference's
// trigger-picking loop (detect.py:410).  HBM-bound single pass over DS.
#include "dtx_internal.h"
#include "dtx_postdev.h"

namespace {

constexpr int kPostThreads = 256;
typedef float fvec4 __attribute__((ext_vector_type(4)));
constexpr int kPostItems = 64;   // values per thread -> kPostItems KiB of DS per block

__device__ inline int hist_bin(float vf) { return dtx_hist_bin_fast(vf); }
// the float32 images of the float64 edges 1/400 .. 4/400 (dtx_postdev.h: dtx_hist_bin_fast)
constexpr float kT1 = 0x1.47ae16p-9f, kT2 = 0x1.47ae16p-8f, kT3 = 0x1.eb852p-8f, kT4 = 0x1.47ae16p-7f;

template <int ITEMS>
__global__ __launch_bounds__(kPostThreads, 8) void ds_post_kernel(
    const float* __restrict__ ds, int64_t P, int64_t ldp, int D, uint32_t* __restrict__ hist,
    float* __restrict__ maxds, const float* __restrict__ thr, uint64_t* __restrict__ cand_key,
    float* __restrict__ cand_val, uint32_t cand_cap, uint32_t* __restrict__ ncand,
    const int* __restrict__ pb) {
  __shared__ uint32_t sh_hist[DTX_HIST_BINS];
  // the float32 images of the 401 float64 edges i / 400 (for a float v and a float64 edge e, (double)v < e  <=>
  // v < up(e), up(e) the smallest float32 >= e): the general classification is a multiply, two LDS reads and
  // compares instead of a float64 search (646 -> 494 us per station-day when every value lies above bin 3)
  __shared__ float sh_edge[DTX_HIST_BINS + 1];
  __shared__ int sh_max[kPostThreads / 64];
  const int tid = threadIdx.x;
  const int det = blockIdx.y, chunk = blockIdx.z;
  if (pb != nullptr) {        // ragged batch: the chunk's own windows only
    P = min(P, (int64_t)pb[chunk]);
    if ((int64_t)blockIdx.x * (kPostThreads * ITEMS) >= P) return;
  }
  const float* row = ds + ((int64_t)chunk * D + det) * ldp;
  if (hist) {
    for (int b = tid; b < DTX_HIST_BINS; b += kPostThreads) sh_hist[b] = 0u;
    for (int b = tid; b <= DTX_HIST_BINS; b += kPostThreads) {
      const double e = b == DTX_HIST_BINS ? 1.0 : (double)b * (1.0 / 400.0);       // dtx_hist_bin's edges
      float f = (float)e;
      if ((double)f < e) f = __uint_as_float(__float_as_uint(f) + 1u);            // (e > 0 here: next float up)
      sh_edge[b] = f;
    }
  }
  __syncthreads();
  // bin of v by the table: e_i <= v < e_{i+1}, the last bin also v == 1; -1 outside [0, 1] and for NaN
  auto bin_of = [&](float v) -> int {
    if (!(v >= 0.f) || v > 1.f) return -1;
    int i = min((int)(v * 400.f), DTX_HIST_BINS - 1);
    while (i > 0 && v < sh_edge[i]) --i;
    while (i < DTX_HIST_BINS - 1 && v >= sh_edge[i + 1]) ++i;
    return i;
  };
  const float th = thr ? thr[det] : 0.f;
  const bool do_cand = thr && cand_key && cand_val && ncand;
  int vmax = 0;   // bit pattern; DS >= 0 so integer order == float order, NaN sorts highest
  // noise windows all fall into the first bins: keep those counts in registers so the
  // LDS atomics do not serialise on one address
  unsigned low[4] = {0u, 0u, 0u, 0u};
  unsigned wc[4] = {0u, 0u, 0u, 0u};      // wave-wide cumulative counts of the fast path (uniform)
  // 16-byte loads when the row is 16-byte aligned (ldp % 4 == 0 and an aligned base)
  const bool vec = ((ldp & 3) == 0) && ((reinterpret_cast<uintptr_t>(ds) & 15) == 0);
  constexpr int kBatch = 4;      // 16-byte loads in flight per thread
  static_assert(ITEMS / 4 % kBatch == 0, "whole batches");
  const unsigned t4bits = __float_as_uint(kT4);
  const int64_t base = (int64_t)blockIdx.x * (kPostThreads * ITEMS);
  for (int it0 = 0; it0 < ITEMS / 4; it0 += kBatch) {
  float vb[kBatch][4];
  bool whole = vec;                       // every load of this lane's batch is a 16-byte load inside the row
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const int64_t p4 = base + ((int64_t)(it0 + u) * kPostThreads + tid) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) vb[u][j] = 0.f;
    if (vec && p4 + 3 < P) {
      const fvec4 q = __builtin_nontemporal_load(reinterpret_cast<const fvec4*>(row + p4));
      vb[u][0] = q.x; vb[u][1] = q.y; vb[u][2] = q.z; vb[u][3] = q.w;
    } else {
      whole = false;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p4 + j < P) vb[u][j] = row[p4 + j];
    }
  }
  // Nearly every window is noise in the first four bins.  One test per batch: the largest bit pattern of
  // the lane's 16 values, compared as unsigned, is below that of edge 4 exactly when all of them lie in
  // [+0, edge 4) (negative values and NaN have larger patterns) -- and below the threshold when none is a
  // candidate.  When that holds for the whole wave the batch is counted branch-free: three compares per
  // value against the float32 images of the float64 edges (the thresholds dtx_hist_bin_fast uses), scalar
  // population counts, one running maximum.
  unsigned mb = 0u;
#pragma unroll
  for (int u = 0; u < kBatch; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) mb = max(mb, __float_as_uint(vb[u][j]));
  // (without a histogram any finite non-negative value is ordinary)
  const bool quiet = whole && mb < (hist ? t4bits : 0x7f800000u) && !(do_cand && __uint_as_float(mb) >= th);
  if (__builtin_expect(__all(quiet), 1)) {
    if (hist) {
#pragma unroll
      for (int u = 0; u < kBatch; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = vb[u][j];
          wc[0] += (unsigned)__popcll(__ballot(v < kT1));
          wc[1] += (unsigned)__popcll(__ballot(v < kT2));
          wc[2] += (unsigned)__popcll(__ballot(v < kT3));
        }
      wc[3] += 64u * 4u * kBatch;
    }
    vmax = max(vmax, (int)mb);
    continue;
  }
  // anything else (values from bin 4 up, candidates, the row's ragged end, odd values): value by value, read
  // again (cache hits) in a rolled loop -- sixteen unrolled copies of this path cost the common one its
  // registers (74, or 53 spilled at 64)
#pragma unroll 1
  for (int q = 0; q < 4 * kBatch; ++q) {
    const int64_t p = base + ((int64_t)(it0 + (q >> 2)) * kPostThreads + tid) * 4 + (q & 3);
    if (p >= P) continue;
    const float v = row[p];
    int bits = __float_as_int(v);
    if (bits < 0) bits = (v != v) ? 0x7fc00000 : 0;   // negative zero / -NaN
    vmax = max(vmax, bits);
    if (hist) {
      const int b = bin_of(v);
      if (b >= 4) atomicAdd(&sh_hist[b], 1u);
      else if (b >= 0) {
        low[0] += (b == 0); low[1] += (b == 1); low[2] += (b == 2); low[3] += (b == 3);
      }
    }
    if (do_cand && v >= th) {
      const uint32_t slot = atomicAdd(ncand, 1u);
      if (slot < cand_cap) {
        cand_key[slot] = ((uint64_t)chunk << 48) | ((uint64_t)det << 32) | (uint64_t)p;  // dtx_cand_pack
        cand_val[slot] = v;
      }
    }
  }
  }
  if (hist && (tid & 63) == 0) {
    // the wave's cumulative counts -> bins 0 .. 3
    if (wc[0]) atomicAdd(&sh_hist[0], wc[0]);
    if (wc[1] - wc[0]) atomicAdd(&sh_hist[1], wc[1] - wc[0]);
    if (wc[2] - wc[1]) atomicAdd(&sh_hist[2], wc[2] - wc[1]);
    if (wc[3] - wc[2]) atomicAdd(&sh_hist[3], wc[3] - wc[2]);
  }
  if (hist) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned c = low[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
      if ((tid & 63) == 0 && c) atomicAdd(&sh_hist[k], c);
    }
  }
  if (maxds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, __shfl_down(vmax, off, 64));
    if ((tid & 63) == 0) sh_max[tid >> 6] = vmax;
  }
  __syncthreads();
  if (maxds && tid == 0) {
    int m = sh_max[0];
    for (int w = 1; w < kPostThreads / 64; ++w) m = max(m, sh_max[w]);
    atomicMax(reinterpret_cast<int*>(maxds) + (int64_t)chunk * D + det, m);
  }
  if (hist)
    for (int b = tid; b < DTX_HIST_BINS; b += kPostThreads) {
      const uint32_t c = sh_hist[b];
      if (c) atomicAdd(&hist[(int64_t)det * DTX_HIST_BINS + b], c);
    }
}

// False-alarm statistics pass (detex/fas.py:72-86): histogram of DS with uniform
// np.linspace(lo, hi, nbins+1) edges and the sufficient statistics of the beta
// maximum-likelihood fit scipy.stats.beta.fit(dss, floc=0, fscale=1) performs
// (n, sum x, sum x^2, sum log x, sum log1p(-x) over 0 < x < 1), all in float64.
__device__ inline int linspace_bin(double v, double lo, double step, int nbins, double hi) {
  if (!(v >= lo) || v > hi) return -1;
  int i = (int)((v - lo) / step);
  if (i > nbins - 1) i = nbins - 1;
  if (i < 0) i = 0;
  // numpy: edge_i = i * step + lo (separate multiply and add), last edge = hi
  while (i > 0 && v < __dadd_rn(__dmul_rn((double)i, step), lo)) --i;
  while (i < nbins - 1 && v >= __dadd_rn(__dmul_rn((double)(i + 1), step), lo)) ++i;
  return i;
}

__global__ __launch_bounds__(kPostThreads) void ds_moments_kernel(
    const float* __restrict__ ds, int64_t P, int64_t ldp, int D, double lo, double hi, int nbins,
    uint32_t* __restrict__ hist, double* __restrict__ sums) {
  extern __shared__ uint32_t sh_h[];
  __shared__ double sh_s[kPostThreads / 64][5];
  const int tid = threadIdx.x;
  const int det = blockIdx.y, chunk = blockIdx.z;
  const float* row = ds + ((int64_t)chunk * D + det) * ldp;
  if (hist)
    for (int b = tid; b < nbins; b += kPostThreads) sh_h[b] = 0u;
  __syncthreads();
  const double step = (hi - lo) / (double)nbins;
  double a[5] = {0, 0, 0, 0, 0};
  const int64_t base = (int64_t)blockIdx.x * (kPostThreads * kPostItems);
  for (int it = 0; it < kPostItems; ++it) {
    const int64_t p = base + (int64_t)it * kPostThreads + tid;
    if (p < P) {
      const double v = (double)row[p];
      if (hist) {
        const int b = linspace_bin(v, lo, step, nbins, hi);
        if (b >= 0) atomicAdd(&sh_h[b], 1u);
      }
      if (v > 0.0 && v < 1.0) {
        a[0] += 1.0; a[1] += v; a[2] += v * v; a[3] += log(v); a[4] += log1p(-v);
      }
    }
  }
  if (sums) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off, 64);
      if ((tid & 63) == 0) sh_s[tid >> 6][k] = a[k];
    }
  }
  __syncthreads();
  if (sums && tid < 5) {
    double t = 0.0;
    for (int w = 0; w < kPostThreads / 64; ++w) t += sh_s[w][tid];
    atomicAdd(&sums[(int64_t)det * 5 + tid], t);
  }
  if (hist)
    for (int b = tid; b < nbins; b += kPostThreads) {
      const uint32_t c = sh_h[b];
      if (c) atomicAdd(&hist[(int64_t)det * nbins + b], c);
    }
}

}  // namespace

int dtx_post_launch(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp, int nchunk, int D,
                    const DtxPost& po, const int* pb, hipStream_t st) {
  if (st == nullptr) st = ctx->stream;
  // 64 values per thread (64 KB of a row per workgroup) when the launch fills the chip several times over anyway;
  // a small launch (configs[4]: ONE detector x one hour = 110 such workgroups on 256 CUs, 45 us for 7 MB) is cut
  // four times finer
  const int64_t per_block = (int64_t)kPostThreads * kPostItems;
  const int64_t nwg = (P + per_block - 1) / per_block * D * nchunk;
  if (nwg >= 8 * ctx->ncu) {
    dim3 grid((unsigned)((P + per_block - 1) / per_block), (unsigned)D, (unsigned)nchunk);
    hipLaunchKernelGGL((ds_post_kernel<kPostItems>), grid, dim3(kPostThreads), 0, st, ds_dev, P, ldp, D,
                       po.hist, po.maxds, po.thr, po.cand_key, po.cand_val, po.cand_cap, po.ncand, pb);
  } else {
    constexpr int kSmall = 16;
    const int64_t pbs = (int64_t)kPostThreads * kSmall;
    dim3 grid((unsigned)((P + pbs - 1) / pbs), (unsigned)D, (unsigned)nchunk);
    hipLaunchKernelGGL((ds_post_kernel<kSmall>), grid, dim3(kPostThreads), 0, st, ds_dev, P, ldp, D,
                       po.hist, po.maxds, po.thr, po.cand_key, po.cand_val, po.cand_cap, po.ncand, pb);
  }
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

extern "C" {

int dtx_ds_moments(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp, int nchunk, int D,
                   double lo, double hi, int nbins, uint32_t* hist_dev, double* sums_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!ds_dev || P <= 0 || ldp < P || nchunk <= 0 || D <= 0 || nbins <= 0 || nbins > 8192 ||
      !(hi > lo))
    return dtx_fail(ctx, DTX_E_ARG, "dtx_ds_moments: bad argument");
  if (nchunk > 65535 || D > 65535) return dtx_fail(ctx, DTX_E_ARG, "nchunk / D > 65535");
  DtxDeviceGuard g(ctx->device);
  const int64_t per_block = (int64_t)kPostThreads * kPostItems;
  dim3 grid((unsigned)((P + per_block - 1) / per_block), (unsigned)D, (unsigned)nchunk);
  hipLaunchKernelGGL(ds_moments_kernel, grid, dim3(kPostThreads), nbins * sizeof(uint32_t),
                     ctx->stream, ds_dev, P, ldp, D, lo, hi, nbins, hist_dev, sums_dev);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}


int dtx_ds_post(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp, int nchunk, int D,
                uint32_t* hist_dev, float* maxds_dev, const float* thr_dev,
                uint64_t* cand_key_dev, float* cand_val_dev, uint32_t cand_cap,
                uint32_t* ncand_dev) {
  if (!ctx) return DTX_E_ARG;
  if (!ds_dev || P <= 0 || ldp < P || nchunk <= 0 || D <= 0)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_ds_post: bad argument");
  if (nchunk > 65535 || D > 65535) return dtx_fail(ctx, DTX_E_ARG, "nchunk / D > 65535");
  DtxDeviceGuard g(ctx->device);
  if (maxds_dev)
    DTX_HIP(ctx, hipMemsetAsync(maxds_dev, 0, (size_t)nchunk * D * sizeof(float), ctx->stream));
  DtxPost po;
  po.hist = hist_dev; po.maxds = maxds_dev; po.thr = thr_dev; po.cand_key = cand_key_dev;
  po.cand_val = cand_val_dev; po.cand_cap = cand_cap; po.ncand = ncand_dev;
  { const int rcw = dtx_wait_post(ctx); if (rcw) return rcw; }      // (an earlier scan's pass may still be adding to the same histogram)
  ctx->post_on_side = false;
  return dtx_post_launch(ctx, ds_dev, P, ldp, nchunk, D, po);
}

// The small results of a scan handed to the host by a KERNEL on the context's stream: 16-byte stores into
// page-locked host memory (device-accessible under unified addressing).  A pipelined host copied them back on a
// side stream behind an event; with several scans in flight those copies and their barrier packets landed in the
// hardware queue in front of the next scan's first kernel (round 5: ~80 us idle behind every post pass,
// profiles/p7_e2e_timeline.txt).  words4[0 .. head + min(count, cmax)) and keys[0 .. min(count, cmax)) go out, where
// count = words4[cnt_word] (the candidate counter inside the packed results).
namespace {
__global__ __launch_bounds__(256) void export_results_kernel(const uint32_t* __restrict__ words, uint32_t* __restrict__ hwords,
                                                             int64_t head, int64_t cmax, int64_t cnt_word,
                                                             const uint64_t* __restrict__ keys, uint64_t* __restrict__ hkeys) {
  const int64_t n = min((int64_t)words[cnt_word], cmax);
  const int64_t nw = head + n;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
  // (16-byte pieces where both sides allow it: torch's allocations are 256-byte aligned, the tail goes word by word)
  const int64_t nw4 = nw / 4;
  const uint4* s4 = reinterpret_cast<const uint4*>(words);
  uint4* d4 = reinterpret_cast<uint4*>(hwords);
  for (int64_t i = t; i < nw4; i += nt) d4[i] = s4[i];
  for (int64_t i = 4 * nw4 + t; i < nw; i += nt) hwords[i] = words[i];
  const int64_t nk2 = n / 2;
  const uint4* k4 = reinterpret_cast<const uint4*>(keys);
  uint4* h4 = reinterpret_cast<uint4*>(hkeys);
  for (int64_t i = t; i < nk2; i += nt) h4[i] = k4[i];
  for (int64_t i = 2 * nk2 + t; i < n; i += nt) hkeys[i] = keys[i];
}
}  // namespace

int dtx_export_results(dtx_ctx* ctx, const uint32_t* words_dev, uint32_t* words_host, int64_t head_words,
                       int64_t cand_max, int64_t cnt_word, const uint64_t* keys_dev, uint64_t* keys_host) {
  if (!ctx) return DTX_E_ARG;
  if (!words_dev || !words_host || !keys_dev || !keys_host || head_words <= 0 || cand_max < 0 || cnt_word < 0 ||
      cnt_word >= head_words)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_export_results: bad argument");
  if (((uintptr_t)words_dev | (uintptr_t)words_host | (uintptr_t)keys_dev | (uintptr_t)keys_host) & 15)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_export_results: buffers must be 16-byte aligned");
  DtxDeviceGuard g(ctx->device);
  // (behind the post pass that wrote them: on its stream)
  hipLaunchKernelGGL(export_results_kernel, dim3(16), dim3(256), 0,
                     (ctx->post_on_side && ctx->post_stream) ? ctx->post_stream : ctx->stream, words_dev, words_host, head_words,
                     cand_max, cnt_word, keys_dev, keys_host);
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

}  // extern "C"
