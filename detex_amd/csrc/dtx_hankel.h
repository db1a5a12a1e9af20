// Implicit-Hankel FP32 MFMA main loop shared by the subspace detection statistic
// (dtx_subspace.hip) and the all-pairs cross-correlation (dtx_xcorr.hip).
//
//   Y[k][p] += sum_c sum_{j in [jlo, jhi)} A_c[j][k] * x_c[p + j]
//
//   A   "dense" operand, stored transposed and channel-planar in HBM:
//       A_c[j][k] = at[(c*Lp + j)*ldk + k]    (basis vectors / template events)
//   x_c channel planes of the "stream" operand, x_c[t] = xw[c*plane_stride + t]
//       (xw already points at the block's first window), zero padded so that
//       every read below is in bounds.  The Hankel matrix H[j][p] = x_c[p + j]
//       is never built: MFMA B fragments are sliding reads of an LDS copy of x_c.
//
// One block = 8 waves computes a [KT = 16*RM] x [PT = 128*TN] tile; wave w owns
// columns [16*TN*w, 16*TN*(w+1)).  Per step of kBQ = 64 contraction indices the
// block stages A_c[j0:j0+64][0:KT] and x_c[p0+j0 : p0+j0+PT+64] into an LDS
// double buffer: the A tile by LDS-DMA (global_load_lds_dwordx4) when its LDS
// image is linear (template flag GL), else through registers; the loads for
// step s+1 are issued at the top of step s; one barrier per step.  The k-loop
// is hand-scheduled: fragment reads run one k-step ahead of the MFMAs.
//
// v_mfma_f32_16x16x4_f32 fragments (lane = 16*kq + i, kq = 0..3, i = 0..15):
//   A[i][kq] = As[(jj+kq)*KTS + 16 r + i]
//   B[kq][i] = xs[col0 + 16 t + i + jj + kq]
//   C: row 16 r + 4 kq + reg, column 16 t + i
// LDS banks (ds_read_b32: two 32-lane groups, 32 banks): B reads touch 17
// consecutive dwords per group with equal addresses broadcast -> conflict-free.
// A reads touch two rows (kq, kq+1) of 16 consecutive dwords: conflict-free iff
// the row stride KTS % 32 == 16, so KTS = KT (+16 when RM is even).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace dtx {

constexpr int kBQ = 64;        // contraction depth staged per step
constexpr int kThreads = 512;  // 8 waves, 2 per SIMD
constexpr int kRmMax = 13;     // 13 * 16 = 208 rows per K-tile
constexpr int kKtMax = 16 * kRmMax;

__host__ __device__ constexpr int tn_for(int rm) { return rm >= 9 ? 2 : (rm >= 5 ? 4 : 8); }
__host__ __device__ constexpr int pt_for(int rm) { return 128 * tn_for(rm); }
__host__ __device__ constexpr int kts_for(int rm) { return 16 * rm + ((rm & 1) ? 0 : 16); }
// floats of LDS one block needs
// first float of the wave-private patches the TAIL hand-over uses (behind the
// 2*KT + 1 floats the epilogues keep at the start of the buffer)
__host__ __device__ constexpr int patch_base(int rm) { return 2 * 16 * rm + 64; }
__host__ __device__ constexpr int lds_floats(int rm) {
  const int loop = 2 * (kBQ * kts_for(rm) + pt_for(rm) + kBQ);
  const int tail = patch_base(rm) + 8 * 8 * 16 * tn_for(rm);
  return loop > tail ? loop : tail;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ds_read_b32 with a 16-bit immediate byte offset: one base VGPR serves a whole
// 64-row tile, so the fragment reads need no address arithmetic between MFMAs
// (hipcc otherwise merges pairs into ds_read2_b32, whose 8-bit offsets force a
// v_add per pair -- VALU issue next to MFMAs is what costs).
// The result is asynchronous: the caller waits with lds_wait_all() before use.
template <int OFF>
__device__ __forceinline__ float lds_read_imm(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read_b32 offset is 16 bits");
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep MFMAs behind the wait (hipcc would hoist them)
}

// BQ: contraction indices staged per step (kBQ everywhere but in the repair pass's tasks, whose few MFMAs per step
// leave every step's load latency exposed: deeper steps, fewer of them)
template <int RM, int TN, bool GL = false, bool TAIL = false, int BQ = kBQ>
__device__ __forceinline__ void hankel_mainloop(f32x4 (&acc)[RM][TN], const float* __restrict__ at,
                                                int ldk, int Lp, const float* __restrict__ xw,
                                                int64_t plane_stride, int Nc, int jlo, int jhi,
                                                float* lds) {
  constexpr int KT = 16 * RM;
  constexpr int KTS = kts_for(RM);
  constexpr int PT = 128 * TN;
  constexpr int KV = KT / 4;                         // float4 per staged row
  constexpr int UV = BQ * KV;                       // float4 per A tile
  constexpr int NUV = (UV + kThreads - 1) / kThreads;
  constexpr int XV = (PT + BQ) / 4;                 // float4 per x window
  constexpr int BUF = BQ * KTS + PT + BQ;          // floats per LDS buffer
  static_assert(XV <= kThreads, "x window staged in one pass");

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, i = lane & 15, kq = lane >> 4;
  const int col0 = wave * 16 * TN;
  const int spc = (jhi - jlo) / BQ;                 // steps per channel
  const int nsteps = Nc * spc;
  if (nsteps <= 0) return;

  // per-thread staging coordinates (row / float4-column inside the A tile)
  int srow[NUV], scol[NUV];
#pragma unroll
  for (int m = 0; m < NUV; ++m) {
    const int v = tid + m * kThreads;
    srow[m] = v / KV;
    scol[m] = (v - srow[m] * KV) * 4;
  }
  float4 ru[NUV];
  float4 rx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < NUV; ++m) ru[m] = make_float4(0.f, 0.f, 0.f, 0.f);

#define DTX_LOAD_TILE(s_)                                                                  \
  do {                                                                                     \
    const int c_ = (s_) / spc;                                                             \
    const int j0_ = jlo + ((s_) - c_ * spc) * BQ;                                         \
    const float* ab_ = at + ((int64_t)c_ * Lp + j0_) * ldk;                                \
    _Pragma("unroll") for (int m = 0; m < NUV; ++m) {                                      \
      if ((m + 1) * kThreads <= UV || tid + m * kThreads < UV)                             \
        ru[m] = *reinterpret_cast<const float4*>(ab_ + (int64_t)srow[m] * ldk + scol[m]);  \
    }                                                                                      \
    if (tid < XV)                                                                          \
      rx = reinterpret_cast<const float4*>(xw + (int64_t)c_ * plane_stride + j0_)[tid];    \
  } while (0)
#define DTX_STORE_TILE(buf_)                                                               \
  do {                                                                                     \
    float* ud_ = lds + (buf_) * BUF;                                                       \
    _Pragma("unroll") for (int m = 0; m < NUV; ++m) {                                      \
      if ((m + 1) * kThreads <= UV || tid + m * kThreads < UV)                             \
        *reinterpret_cast<float4*>(ud_ + srow[m] * KTS + scol[m]) = ru[m];                 \
    }                                                                                      \
    if (tid < XV) reinterpret_cast<float4*>(ud_ + BQ * KTS)[tid] = rx;                    \
  } while (0)

  // LDS-DMA staging of the A tile (GL): the tile is one linear image when the LDS
  // row stride equals the global one (KTS == KT == ldk), so each wave copies whole
  // 1 KiB units with global_load_lds_dwordx4 -- no staging VGPRs, no ds_write, no
  // address VALU beyond one 64-bit add per unit.
  static_assert(!GL || KTS == KT, "LDS-DMA needs an unpadded LDS image");
  constexpr int UNITS = BQ * KT * 4 / 1024;         // 1 KiB wave-instructions per A tile
  constexpr int NGU = (UNITS + 7) / 8;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define DTX_GLDS_TILE(s_, buf_)                                                            \
  do {                                                                                     \
    const int c_ = (s_) / spc;                                                             \
    const int j0_ = jlo + ((s_) - c_ * spc) * BQ;                                         \
    const float* ab_ = at + ((int64_t)c_ * Lp + j0_) * KT + lane * 4;                      \
    float* ld_ = lds + (buf_) * BUF;                                                       \
    _Pragma("unroll") for (int m = 0; m < NGU; ++m) {                                      \
      const int u_ = wave_u + 8 * m;                                                       \
      if ((m + 1) * 8 <= UNITS || u_ < UNITS)                                              \
        __builtin_amdgcn_global_load_lds(                                                  \
            (const __attribute__((address_space(1))) void*)(ab_ + u_ * 256),               \
            (__attribute__((address_space(3))) void*)(ld_ + u_ * 256), 16, 0, 0);          \
    }                                                                                      \
    if (tid < XV)                                                                          \
      rx = reinterpret_cast<const float4*>(xw + (int64_t)c_ * plane_stride + j0_)[tid];    \
  } while (0)
#define DTX_STORE_X(buf_)                                                                  \
  do {                                                                                     \
    if (tid < XV) reinterpret_cast<float4*>(lds + (buf_) * BUF + BQ * KTS)[tid] = rx;     \
  } while (0)

  const int aoff = kq * KTS + i;
  const int boff = BQ * KTS + col0 + i + kq;

  if constexpr (GL) {
    DTX_GLDS_TILE(0, 0);
    DTX_STORE_X(0);
  } else {
    DTX_LOAD_TILE(0);
    DTX_STORE_TILE(0);
  }
  __syncthreads();

  // Hand-scheduled k-loop: the LDS reads of k-step ks+1 (immediate-offset
  // ds_read_b32 from per-lane base registers) are issued between the MFMAs of
  // k-step ks, one row block at a time; a single lgkmcnt(0) per k-step, long
  // after the reads were issued; sched_barrier pins the order.
  //
  // TAIL: the last row block holds at most 8 real rows.  Instead of a 16x16x4
  // MFMA on a half-empty block it is computed by v_mfma_f32_4x4x1_16B_f32 with
  // CBSZ = 3 (the A rows of block ABID are broadcast to the 8 blocks of each
  // half-wave): one instruction = 8 rows x 32 columns x 1 contraction index in 8
  // cycles, i.e. half the matrix-pipe time of the padded block.  Lane l = 32 g +
  // 4 a + j: A register holds row 4 g + j at contraction index a (a < 4), B
  // registers hold column (l & 31) at index a = 0..3; result registers are rows
  // 4 g + e of column (l & 31).
  const unsigned lds_base = (unsigned)(unsigned long long)(
      (__attribute__((address_space(3))) float*)lds);
  constexpr int KS = BQ / 4;
  constexpr int RF = TAIL ? RM - 1 : RM;          // full 16-row blocks
  constexpr int TH = TN / 2;                      // 32-column groups of the tail
  static_assert(!TAIL || (RM >= 2 && TN % 2 == 0), "tail path needs a full block and even TN");
  const int tg = lane >> 5, ta = (lane >> 2) & 3, tj = lane & 3;
  const int taoff = ta * KTS + 16 * RF + 4 * tg + tj;
  const int tboff = BQ * KTS + col0 + (lane & 31);
  f32x4 tacc[TH > 0 ? TH : 1];
#pragma unroll
  for (int h = 0; h < (TH > 0 ? TH : 1); ++h) tacc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < nsteps; ++s) {
    const unsigned bufb = lds_base + (unsigned)((s & 1) * BUF) * 4u;
    const unsigned abase = bufb + (unsigned)aoff * 4u;
    const unsigned bbase = bufb + (unsigned)boff * 4u;
    const unsigned tabase = bufb + (unsigned)taoff * 4u;
    const unsigned tbbase = bufb + (unsigned)tboff * 4u;
    float av[2][RF], bv[2][TN];
    float tav[2], tbv[2][TH > 0 ? TH : 1][4];
    static_for<0, RF>([&](auto rc) { av[0][rc] = lds_read_imm<(16 * rc) * 4>(abase); });
    static_for<0, TN>([&](auto tc) { bv[0][tc] = lds_read_imm<(16 * tc) * 4>(bbase); });
    if constexpr (TAIL) {
      tav[0] = lds_read_imm<0>(tabase);
      static_for<0, TH>([&](auto hc) {
        static_for<0, 4>([&](auto ac) { tbv[0][hc][ac] = lds_read_imm<(32 * hc + ac) * 4>(tbbase); });
      });
    }
    if (s + 1 < nsteps) {
      if constexpr (GL) DTX_GLDS_TILE(s + 1, (s + 1) & 1);
      else DTX_LOAD_TILE(s + 1);
    }
    lds_wait_all();
    static_for<0, KS>([&](auto ksc) {
      constexpr int ks = ksc, cur = ks & 1, nxt = cur ^ 1;
      static_for<0, RF>([&](auto rc) {
        constexpr int r = rc;
#pragma unroll
        for (int t = 0; t < TN; ++t)
          acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cur][r], bv[cur][t], acc[r][t], 0, 0, 0);
        if constexpr (TAIL) {
          // the four dependent tail MFMAs of each column group are spread over the k-step
          static_for<0, 4>([&](auto ac) {
            constexpr int a = ac;
            if constexpr (r == ((a + 1) * RF + 3) / 4 - 1) {
              static_for<0, TH>([&](auto hc) {
                tacc[hc] = __builtin_amdgcn_mfma_f32_4x4x1f32(tav[cur], tbv[cur][hc][a], tacc[hc],
                                                              3, a, 0);
              });
            }
          });
        }
        if constexpr (ks + 1 < KS) {
          av[nxt][r] = lds_read_imm<((ks + 1) * 4 * KTS + 16 * r) * 4>(abase);
          // the TN stream fragments ride along with the first row blocks
          constexpr int TPR = (TN + RF - 1) / RF;
          static_for<0, TPR>([&](auto qc) {
            constexpr int t = r * TPR + qc;
            if constexpr (t < TN) bv[nxt][t] = lds_read_imm<(16 * t + (ks + 1) * 4) * 4>(bbase);
          });
          if constexpr (TAIL) {
            // 1 + 4*TH tail fragments, spread over the row blocks from the back
            constexpr int NTF = 1 + 4 * TH;
            constexpr int FPR = (NTF + RF - 1) / RF;
            static_for<0, FPR>([&](auto qc) {
              constexpr int f = (RF - 1 - r) * FPR + qc;
              if constexpr (f == 0) {
                tav[nxt] = lds_read_imm<((ks + 1) * 4 * KTS) * 4>(tabase);
              } else if constexpr (f < NTF) {
                constexpr int h = (f - 1) / 4, a = (f - 1) % 4;
                tbv[nxt][h][a] = lds_read_imm<(32 * h + (ks + 1) * 4 + a) * 4>(tbbase);
              }
            });
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (ks + 1 < KS) lds_wait_all();
    });
    if (s + 1 < nsteps) {
      if constexpr (GL) DTX_STORE_X((s + 1) & 1);
      else DTX_STORE_TILE((s + 1) & 1);
    }
    __syncthreads();
  }

  if constexpr (TAIL) {
    // hand the tail rows over in the accumulator layout of a 16x16 block (rows
    // 8..15 zero) through a wave-private LDS patch, so the epilogues stay generic
    float* patch = lds + patch_base(RM) + wave * (8 * 16 * TN);
#pragma unroll
    for (int h = 0; h < TH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) patch[(4 * tg + e) * (16 * TN) + 32 * h + (lane & 31)] = tacc[h][e];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[RM - 1][t][e] = (kq < 2) ? patch[(4 * kq + e) * (16 * TN) + 16 * t + i] : 0.f;
  }
#undef DTX_LOAD_TILE
#undef DTX_GLDS_TILE
#undef DTX_STORE_X
#undef DTX_STORE_TILE
}

// ---------------------------------------------------------------------------
// Sliding window statistics over channel planes, in float64:
//   mean[p] = (1/n) sum_c sum_{j<L} x_c[p+j],  ssd[p] = sum (x - mean)^2 * scale
// (scale < 0: ssd[p] = 1/sqrt(sum (x - mean)^2), or 0 for a zero-variance window)
// One directly summed anchor window per block, then sliding updates combined by
// a block-wide scan (pandas rolling_mean / rolling_var, detect.py:566-568 and
// construct.py:446-447).  planes: [batch][Nc][plane_stride].
// ---------------------------------------------------------------------------
constexpr int kStatsPB = 2048;  // windows per block

// variance below float64 round-off relative to the window power counts as 0
__device__ inline double clamp_var(double s2, double s1, double dn) {
  const double v = s2 - s1 * s1 / dn;
  return (v <= 1e-13 * s2) ? 0.0 : v;
}

__device__ inline float stats_out(double scale, double v) {
  if (scale >= 0.0) return (float)(scale * v);
  return v > 0.0 ? (float)(1.0 / sqrt(v)) : 0.f;
}

__device__ inline double wave_incl_scan(double v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

static __global__ __launch_bounds__(256) void win_stats_kernel(const float* __restrict__ planes,
                                                        int64_t plane_stride, int Nc, int L,
                                                        int64_t P, int64_t Ppad, double scale,
                                                        float* __restrict__ mean,
                                                        float* __restrict__ ssd,
                                                        const float* __restrict__ tau = nullptr,
                                                        int* __restrict__ over = nullptr, int redo = 0) {
  constexpr int E = kStatsPB / 256;  // windows per thread
  __shared__ double sh1[4], sh2[4], sh3[4];
  __shared__ double anchor[3];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  const float* xb = planes + (int64_t)b * Nc * plane_stride;
  const int64_t tlim = plane_stride;                    // samples per channel that may be read
  auto at = [&](int c, int64_t t) -> float { return xb[(int64_t)c * plane_stride + t]; };
  const int64_t p0 = (int64_t)blockIdx.x * kStatsPB;
  if (p0 >= P) return;
  // Samples far beyond the batch element's ordinary scatter (tau[b]: the square from which; none without a tau) are
  // kept OUT of the sliding sums -- they enter as zeros -- and listed (LDS: frame, value); a window's sums get the
  // listed samples it holds added at the end.  A 10^6-sigma glitch then never passes through the float64 sliding
  // sums, and the windows that do not hold it come out as if it were not there: the sums otherwise keep ~1e-16 of
  // the largest power they passed through, and the direct re-summation below, meant for a few windows, met every
  // quiet window of the block (round 5: 10 such glitches took a station-day from 3.3 to 23 ms).  Exact whatever tau
  // is.  More than kBigCap such samples in a block's span (a sustained loud signal): the block raises its flag in
  // `over` ([batch][blocks], cleared by the caller) and the SECOND launch (redo = 1: only flagged blocks run) sums
  // it as one class, round 5's way -- a launch that finds no flag on ordinary data.
  constexpr int kBigCap = 128;
  __shared__ int big_t[kBigCap];
  __shared__ float big_v[kBigCap];
  __shared__ int nbig_s;
  int* myflag = over != nullptr ? over + (int64_t)b * gridDim.x + blockIdx.x : nullptr;
  if (redo && (myflag == nullptr || *myflag == 0)) return;
  const float big = (tau != nullptr && !redo) ? tau[b] : INFINITY;
  // (a sample seen as part of the anchor window, frames [p0, p0 + L), or ENTERING a later window, frames >= p0 + L:
  // every sample of the block's span exactly once)
  auto keep_or_list = [&](float v, int64_t t) -> float {
    if (!(v * v > big)) return v;
    const int k = atomicAdd(&nbig_s, 1);
    if (k < kBigCap) { big_t[k] = (int)(t - p0); big_v[k] = v; }
    return 0.f;
  };
  auto small = [&](float v) -> float { return (v * v > big) ? 0.f : v; };
  // (the per-sample tests sit behind ONE vote per batch of loads -- does any lane of the wave hold such a sample? --
  // so that on ordinary data the loops below are round 5's: with the tests inline the kernel ran 124 us against 92)
  auto any_big = [&](float m) -> bool { return __any(m * m > big) != 0; };

  double l1[kStatsPB / 256], l2[kStatsPB / 256], l3[kStatsPB / 256];
  double off1, off2, off3;
  if (tid == 0) nbig_s = 0;
  __syncthreads();

  // third running quantity: the number of sample-to-sample changes inside the window
  // (per channel).  It is an exact integer, so "the window is constant" -- the
  // zero-variance case the statistic defines as 0 -- is decided exactly, however much
  // round-off the float64 sliding sums have picked up from large values elsewhere.
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  {
  // (twelve loads in flight per pass -- a whole channel of a 30 s template at 100 Hz: taken one at a time
  // this sum was a chain of ~36 exposed memory latencies per block; the previous sample of a lane's
  // is its neighbour's: one extra load per wave instead of one per lane.  Six or twelve in flight measure
  // the same beside the block spectra: 84 - 88 us)
  for (int c = 0; c < Nc; ++c) {
    const float* w0 = xb + (int64_t)c * plane_stride + p0;
    constexpr int kAB = 12;
    for (int j0 = 0; j0 < L; j0 += 256 * kAB) {
      float v[kAB], vl[kAB];
#pragma unroll
      for (int k = 0; k < kAB; ++k) {
        const int j = j0 + 256 * k + tid;
        v[k] = j < L ? w0[j] : 0.f;
        vl[k] = (lane == 0 && j < L && j > 0) ? w0[j - 1] : 0.f;     // (lane 0 only: the wave's left neighbour)
      }
      float vmax = 0.f;
#pragma unroll
      for (int k = 0; k < kAB; ++k) vmax = fmaxf(vmax, fabsf(v[k]));
      const bool hot = any_big(vmax);
#pragma unroll
      for (int k = 0; k < kAB; ++k) {
        const int j = j0 + 256 * k + tid;
        const float up = __shfl_up(v[k], 1, 64);
        const float vp = lane == 0 ? vl[k] : up;
        if (j < L) {
          float vs = v[k];
          if (hot) vs = keep_or_list(vs, p0 + j);
          const double d = (double)vs;
          s1 += d;
          s2 += d * d;
          if (j > 0) s3 += (v[k] != vp) ? 1.0 : 0.0;
        }
      }
    }
  }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
    s3 += __shfl_down(s3, off, 64);
  }
  if (lane == 0) { sh1[wv] = s1; sh2[wv] = s2; sh3[wv] = s3; }
  __syncthreads();
  if (tid == 0) {
    anchor[0] = sh1[0] + sh1[1] + sh1[2] + sh1[3];
    anchor[1] = sh2[0] + sh2[1] + sh2[2] + sh2[3];
    anchor[2] = sh3[0] + sh3[1] + sh3[2] + sh3[3];
  }
  __syncthreads();

  // local element e turns window (p0 + E*tid + e) into the next one: per channel the thread's E
  // steps touch E + 1 consecutive samples at the head and E + 1 at the tail of its windows -- all
  // loaded first (independent loads in flight), each once
  double r1_ = 0.0, r2_ = 0.0, r3_ = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) l1[e] = l2[e] = l3[e] = 0.0;
  const int64_t pt = p0 + (int64_t)E * tid;
  // (the batched loads read up to sample pt + L + E - 1 of the plane whatever P is: only where that
  // stays inside the plane -- the last threads of a chunk's last block take the guarded loop)
  if (pt + L + E <= tlim) {
  for (int c = 0; c < Nc; ++c) {
    float ha[E + 1], tb[E + 1];                                // ha[j] = o[j], tb[j] = o[L - 1 + j]
    {
      const float* o = xb + (int64_t)c * plane_stride + pt;
#pragma unroll
      for (int j = 0; j <= E; ++j) ha[j] = o[j];
#pragma unroll
      for (int j = 0; j <= E; ++j) tb[j] = o[L - 1 + j];
    }
    float hmax = 0.f;
#pragma unroll
    for (int j = 0; j <= E; ++j) hmax = fmaxf(hmax, fmaxf(fabsf(ha[j]), fabsf(tb[j])));
    const bool hot = any_big(hmax);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float fo = ha[e], fn = tb[e + 1];
      if (hot) {
        fo = small(fo);
        fn = (pt + e + 1 < P) ? keep_or_list(fn, pt + L + e) : 0.f;
      }
      const double vo = (double)fo, vn = (double)fn;
      l1[e] += vn - vo;
      l2[e] += vn * vn - vo * vo;
      // changes entering (between samples L-1 and L) and leaving (between 0 and 1)
      if (L > 1) l3[e] += ((tb[e + 1] != tb[e]) ? 1.0 : 0.0) - ((ha[e + 1] != ha[e]) ? 1.0 : 0.0);
    }
  }
  double r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (pt + e + 1 < P) { r1 += l1[e]; r2 += l2[e]; r3 += l3[e]; }
    l1[e] = r1;
    l2[e] = r2;
    l3[e] = r3;
  }
  r1_ = r1; r2_ = r2; r3_ = r3;
  } else {
  double r1 = 0.0, r2 = 0.0, r3 = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int64_t pw = p0 + (int64_t)E * tid + e;
    if (pw + 1 < P) {
      for (int c = 0; c < Nc; ++c) {
        const float o0 = at(c, pw), o1 = at(c, pw + 1), oL1 = at(c, pw + L - 1), oL = at(c, pw + L);
        const double vo = (double)small(o0), vn = (double)keep_or_list(oL, pw + L);
        r1 += vn - vo;
        r2 += vn * vn - vo * vo;
        // changes entering (between samples L-1 and L) and leaving (between 0 and 1)
        if (L > 1) r3 += ((oL != oL1) ? 1.0 : 0.0) - ((o1 != o0) ? 1.0 : 0.0);
      }
    }
    l1[e] = r1;
    l2[e] = r2;
    l3[e] = r3;
  }
  r1_ = r1; r2_ = r2; r3_ = r3;
  }
  const double r1 = r1_, r2 = r2_, r3 = r3_;
  const double i1 = wave_incl_scan(r1, lane), i2 = wave_incl_scan(r2, lane),
               i3 = wave_incl_scan(r3, lane);
  __syncthreads();
  if (lane == 63) { sh1[wv] = i1; sh2[wv] = i2; sh3[wv] = i3; }
  __syncthreads();
  off1 = i1 - r1; off2 = i2 - r2; off3 = i3 - r3;  // exclusive within the wave
  for (int w = 0; w < wv; ++w) { off1 += sh1[w]; off2 += sh2[w]; off3 += sh3[w]; }
  // (every append lies in front of the barriers above)
  if (nbig_s > kBigCap) {                            // too many for the list: the second launch takes the block
    if (tid == 0 && myflag != nullptr) *myflag = 1;
    return;
  }
  const int nbig = nbig_s;
  // what window (p0 + q) holds of the listed samples: frames q .. q + L - 1 of the block's span
  auto listed = [&](int q, double& S1, double& S2) {
    for (int k = 0; k < nbig; ++k) {
      const int dt = big_t[k] - q;
      if (dt >= 0 && dt < L) {
        const double d = (double)big_v[k];
        S1 += d;
        S2 += d * d;
      }
    }
  };
  const double a1 = anchor[0], a2 = anchor[1], a3 = anchor[2];
  const double dn = (double)Nc * (double)L;
  float* mo = mean + (int64_t)b * Ppad;
  float* so = ssd + (int64_t)b * Ppad;
  // a window is constant iff no sample changes inside any channel and all channels
  // start from the same value
  auto constant = [&](int64_t pw, double nchg) {
    if (nchg > 0.5) return false;
    const float v0 = at(0, pw);
    for (int c = 1; c < Nc; ++c)
      if (at(c, pw) != v0) return false;
    return true;
  };
  if (tid == 0) {
    double S1 = a1, S2 = a2;
    listed(0, S1, S2);
    mo[p0] = (float)(S1 / dn);
    so[p0] = constant(p0, a3) ? stats_out(scale, 0.0) : stats_out(scale, clamp_var(S2, S1, dn));
  }
  // The sliding sums of a block pass through the largest window power of the block, and what float64 rounds
  // off there (~1e-16 of it) stays in every later window's sums: a 10^4-sigma spike leaves 4e-9 behind, 2e-3 of
  // the sum of squares of a window that holds ONE live sample at the edge of a dead stretch (found by a soak run of
  // test_random_scan: DS off by 2.5e-4 there).  A window whose variance is below kStatsRecheck of the block's
  // largest power is therefore summed again directly (mean first, then the squares of the residuals): rare, exact.
  // (The power is that of the sliding sums, which the listed samples never enter: with a tau it is at most n tau =
  // 2^20 n x the ordinary scatter, and an ordinary full window, n x the scatter, never qualifies.)
  __shared__ double shm[4];
  __shared__ double blockmax;
  {
    double m2 = a2;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (p0 + (int64_t)E * tid + e + 1 < P) m2 = fmax(m2, a2 + off2 + l2[e]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m2 = fmax(m2, __shfl_xor(m2, off, 64));
    if (lane == 0) shm[wv] = m2;
    __syncthreads();
    if (tid == 0) blockmax = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
    __syncthreads();
  }
  // 1e-9: what the sums round off is a few 1e-16 of the largest power they hold (one rounding of anchor + prefix;
  // outliers never enter them), i.e. below 1e-6 of such a window's variance.  (Round 5 used 1e-7: a record riding a
  // level of 5000 x its scatter -- a step no offset removes -- then sent EVERY window through the serial
  // re-summation: 35 ms per station-day instead of 3.3, profiles/r6_dirty_cost.txt (e))
  constexpr double kStatsRecheck = 1e-9;
  const double recheck = kStatsRecheck * blockmax;
  auto direct = [&](int64_t pw, double& S1, double& var) {
    double s = 0.0;
    for (int c = 0; c < Nc; ++c)
      for (int j = 0; j < L; ++j) s += (double)at(c, pw + j);
    const double mu = s / dn;
    double q = 0.0;
    for (int c = 0; c < Nc; ++c)
      for (int j = 0; j < L; ++j) {
        const double d = (double)at(c, pw + j) - mu;
        q += d * d;
      }
    S1 = s;
    var = q;
  };
  // a thread's eight results are eight consecutive windows: stored directly, a wave instruction
  // writes 64 four-byte pieces 32 bytes apart.  Through LDS (slot q + (q >> 5): the strided writes are
  // conflict-free) each instruction stores 256 consecutive bytes.
  __shared__ float stm[kStatsPB + kStatsPB / 32], sts[kStatsPB + kStatsPB / 32];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int64_t pw = p0 + (int64_t)E * tid + e + 1;
    float m = 0.f, sd = 0.f;
    if (pw < P && pw < p0 + kStatsPB) {
      double S1 = a1 + off1 + l1[e];
      double S2 = a2 + off2 + l2[e];
      const double S3 = a3 + off3 + l3[e];
      const bool flat = constant(pw, S3);
      const bool quiet = !flat && S2 - S1 * S1 / dn < recheck;      // (judged on the sliding sums alone)
      listed((int)(pw - p0), S1, S2);
      double var = flat ? 0.0 : clamp_var(S2, S1, dn);
      if (quiet) direct(pw, S1, var);
      m = (float)(S1 / dn);
      sd = stats_out(scale, var);
    }
    const int q = E * tid + e;                           // window p0 + 1 + q
    stm[q + (q >> 5)] = m;
    sts[q + (q >> 5)] = sd;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int q = 256 * e + tid;
    const int64_t pw = p0 + 1 + q;
    if (pw < P && pw < p0 + kStatsPB) {
      mo[pw] = stm[q + (q >> 5)];
      so[pw] = sts[q + (q >> 5)];
    }
  }
}

}  // namespace dtx
