// Device-side pieces of the DS post pass (histogram / maximum / threshold candidates),
// shared by the stand-alone pass (dtx_post.hip) and the scan kernels that fold it into
// their epilogue (dtx_fft.hip): the same functions on the same float32 values give the
// same counts, bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/detex_hip.h"

struct DtxPost {
  uint32_t* hist;       // [D][DTX_HIST_BINS], accumulated (may be NULL)
  float* maxds;         // [nchunk][D], zeroed by the host wrapper (may be NULL)
  const float* thr;     // [D] (NULL: no candidates)
  uint64_t* cand_key;   // capacity cand_cap
  float* cand_val;
  uint32_t cand_cap;
  uint32_t* ncand;
};

// np.histogram(v, np.linspace(0, 1, 401)): bin i holds e_i <= v < e_{i+1}, the
// last bin also v == 1; edges are e_i = i * (1/400) in float64 and e_400 = 1.
__device__ inline int dtx_hist_bin(float vf) {
  const double v = (double)vf;
  if (!(v >= 0.0) || v > 1.0) return -1;          // also drops NaN
  const double step = 1.0 / 400.0;
  int i = (int)(v * 400.0);
  if (i > 399) i = 399;
  while (i > 0 && v < (double)i * step) --i;
  while (i < 399 && v >= (double)(i + 1) * step) ++i;
  return i;
}

// the same bin; nearly every window is noise in the first four bins: settle those with three
// compares against the same float64 edges i * (1/400) the general search uses -- no loop, no
// divergence inside a wave
// (in float32: for a float v and a float64 edge e, (double)v < e  <=>  v < up(e), up(e) the
// smallest float32 >= e -- the four edges below, each checked against its neighbour)
__device__ inline int dtx_hist_bin_fast(float vf) {
  constexpr double step = 1.0 / 400.0;
  constexpr float t1 = 0x1.47ae16p-9f, t2 = 0x1.47ae16p-8f, t3 = 0x1.eb852p-8f, t4 = 0x1.47ae16p-7f;
  static_assert((double)t1 >= 1.0 * step && (double)0x1.47ae14p-9f < 1.0 * step, "edge 1");
  static_assert((double)t2 >= 2.0 * step && (double)0x1.47ae14p-8f < 2.0 * step, "edge 2");
  static_assert((double)t3 >= 3.0 * step && (double)0x1.eb851ep-8f < 3.0 * step, "edge 3");
  static_assert((double)t4 >= 4.0 * step && (double)0x1.47ae14p-7f < 4.0 * step, "edge 4");
  if (vf >= 0.f && vf < t4) return vf < t2 ? (vf < t1 ? 0 : 1) : (vf < t3 ? 2 : 3);
  return dtx_hist_bin(vf);
}

// per-thread partial statistics of one (chunk, detector) row; noise windows all fall into the
// first bins, so those counts stay in registers
struct DtxPostAcc {
  int vmax = 0;                      // bit pattern: DS >= 0, so integer order == float order, NaN highest
  unsigned low[4] = {0u, 0u, 0u, 0u};
};
__device__ inline void dtx_post_value(DtxPostAcc& s, const DtxPost& po, float v, int chunk, int det,
                                      int64_t p, float th, bool do_cand) {
  int bits = __float_as_int(v);
  if (bits < 0) bits = (v != v) ? 0x7fc00000 : 0;   // negative zero / -NaN
  s.vmax = max(s.vmax, bits);
  if (po.hist) {
    const int b = dtx_hist_bin_fast(v);
    if (b >= 4) atomicAdd(&po.hist[(int64_t)det * DTX_HIST_BINS + b], 1u);
    else if (b >= 0) {
      s.low[0] += (b == 0); s.low[1] += (b == 1); s.low[2] += (b == 2); s.low[3] += (b == 3);
    }
  }
  if (do_cand && v >= th) {
    const uint32_t slot = atomicAdd(po.ncand, 1u);
    if (slot < po.cand_cap) {
      po.cand_key[slot] = ((uint64_t)chunk << 48) | ((uint64_t)det << 32) | (uint64_t)p;  // dtx_cand_pack
      po.cand_val[slot] = v;
    }
  }
}
// wave-level commit: every lane of the wave calls it
__device__ inline void dtx_post_commit(DtxPostAcc& s, const DtxPost& po, int chunk, int det, int D) {
  const int lane = threadIdx.x & 63;
  if (po.hist) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned c = s.low[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
      if (lane == 0 && c) atomicAdd(&po.hist[(int64_t)det * DTX_HIST_BINS + k], c);
    }
  }
  if (po.maxds) {
    int m = s.vmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
    if (lane == 0) atomicMax(reinterpret_cast<int*>(po.maxds) + (int64_t)chunk * D + det, m);
  }
  s = DtxPostAcc();
}
