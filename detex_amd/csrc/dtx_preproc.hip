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
// blocks: (1) every block of kPB samples is run from a zero state and its final
// state kept, (2) the true state entering each block follows from
// S_b = M S_{b-1} + f_{b-1} with M the state transition over one block (a tiny
// sequential pass per series), (3) every block is run again from its true state
// and written out.  Series (chunk x channel) and blocks are independent.
#include "dtx_internal.h"

namespace {

constexpr int kMaxSec = 8;    // biquad sections (bandpass corners <= 8)
constexpr int kPB = 1024;     // samples per block

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

// per series: least-squares line a + b t (scipy.signal.detrend type='linear')
__global__ __launch_bounds__(256) void pp_trend_kernel(const float* __restrict__ raw, int64_t T,
                                                       int64_t ser_stride,
                                                       double* __restrict__ trend) {
  __shared__ double sh[8];
  const int ser = blockIdx.x, tid = threadIdx.x;
  const float* x = raw + (int64_t)ser * ser_stride;
  // centred abscissa u = t - (T-1)/2 makes the normal equations diagonal
  const double tc = 0.5 * (double)(T - 1);
  double s0 = 0.0, s1 = 0.0;
  for (int64_t t = tid; t < T; t += 256) {
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
    s0 = sh[0] + sh[1] + sh[2] + sh[3];
    s1 = sh[4] + sh[5] + sh[6] + sh[7];
    const double n = (double)T;
    const double suu = n * (n * n - 1.0) / 12.0;          // sum (t - tc)^2
    const double slope = (T > 1) ? s1 / suu : 0.0;
    trend[2 * ser] = s0 / n - slope * tc;                 // intercept at t = 0
    trend[2 * ser + 1] = slope;
  }
}

// input sample i (in filter order) of a series
template <bool FIRST>
__device__ inline double pp_input(const float* raw, const double* mid, const double* trend,
                                  int64_t T, int64_t i, bool rev) {
  const int64_t t = rev ? (T - 1 - i) : i;
  if (FIRST) return (double)raw[t] - (trend ? trend[0] + trend[1] * (double)t : 0.0);
  return mid[t];
}

// pass 1: run every block from a zero state, keep its final state
template <bool FIRST>
__global__ __launch_bounds__(256) void pp_block_state_kernel(
    const float* __restrict__ raw, int64_t ser_stride, const double* __restrict__ mid,
    const double* __restrict__ trend, int64_t T, int nblk, Sos f, bool rev,
    double* __restrict__ fstate) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ser = blockIdx.y;
  if (g >= nblk) return;
  const float* r = raw ? raw + (int64_t)ser * ser_stride : nullptr;
  const double* m = mid ? mid + (int64_t)ser * T : nullptr;
  const double* tr = trend ? trend + 2 * ser : nullptr;
  double z[2 * kMaxSec];
  for (int k = 0; k < 2 * f.nsec; ++k) z[k] = 0.0;
  const int64_t i0 = g * kPB, i1 = min(i0 + (int64_t)kPB, T);
  for (int64_t i = i0; i < i1; ++i) sos_step(f, pp_input<FIRST>(r, m, tr, T, i, rev), z);
  double* out = fstate + ((int64_t)ser * nblk + g) * (2 * kMaxSec);
  for (int k = 0; k < 2 * f.nsec; ++k) out[k] = z[k];
}

// pass 2: true entry state of every block, sequential over blocks (one thread per series)
__global__ void pp_chain_kernel(double* __restrict__ fstate, const double* __restrict__ M, int nblk,
                                int ns, int nser) {
  const int ser = blockIdx.x * blockDim.x + threadIdx.x;
  if (ser >= nser) return;
  double S[2 * kMaxSec];
  for (int k = 0; k < ns; ++k) S[k] = 0.0;
  double* fs = fstate + (int64_t)ser * nblk * (2 * kMaxSec);
  for (int b = 0; b < nblk; ++b) {
    double fin[2 * kMaxSec], nxt[2 * kMaxSec];
    for (int k = 0; k < ns; ++k) { fin[k] = fs[b * (2 * kMaxSec) + k]; fs[b * (2 * kMaxSec) + k] = S[k]; }
    for (int r = 0; r < ns; ++r) {
      double acc = fin[r];
      for (int c = 0; c < ns; ++c) acc += M[r * ns + c] * S[c];
      nxt[r] = acc;
    }
    for (int k = 0; k < ns; ++k) S[k] = nxt[k];
  }
}

// pass 3: run every block from its true state; write float64 (intermediate) or the
// multiplexed float32 output x[Nc t + c]
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256) void pp_block_run_kernel(
    const float* __restrict__ raw, int64_t ser_stride, const double* __restrict__ mid,
    const double* __restrict__ trend, int64_t T, int nblk, Sos f, bool rev,
    const double* __restrict__ fstate, double* __restrict__ out_mid, float* __restrict__ out_mp,
    int Nc, int64_t mp_stride) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int ser = blockIdx.y;
  if (g >= nblk) return;
  const float* r = raw ? raw + (int64_t)ser * ser_stride : nullptr;
  const double* m = mid ? mid + (int64_t)ser * T : nullptr;
  const double* tr = trend ? trend + 2 * ser : nullptr;
  double z[2 * kMaxSec];
  const double* st = fstate + ((int64_t)ser * nblk + g) * (2 * kMaxSec);
  for (int k = 0; k < 2 * f.nsec; ++k) z[k] = st[k];
  const int64_t i0 = g * kPB, i1 = min(i0 + (int64_t)kPB, T);
  const int chunk = ser / Nc, c = ser - chunk * Nc;
  for (int64_t i = i0; i < i1; ++i) {
    const double y = sos_step(f, pp_input<FIRST>(r, m, tr, T, i, rev), z);
    const int64_t t = rev ? (T - 1 - i) : i;
    if (LAST) out_mp[(int64_t)chunk * mp_stride + t * Nc + c] = (float)y;
    else out_mid[(int64_t)ser * T + t] = y;
  }
}

// no filter: detrend (optional) + multiplex
__global__ __launch_bounds__(256) void pp_copy_kernel(const float* __restrict__ raw,
                                                      int64_t ser_stride,
                                                      const double* __restrict__ trend, int64_t T,
                                                      int Nc, int64_t mp_stride,
                                                      float* __restrict__ out_mp) {
  const int ser = blockIdx.y;
  const int chunk = ser / Nc, c = ser - chunk * Nc;
  const float* r = raw + (int64_t)ser * ser_stride;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < T; t += (int64_t)gridDim.x * 256) {
    double v = (double)r[t];
    if (trend) v -= trend[2 * ser] + trend[2 * ser + 1] * (double)t;
    out_mp[(int64_t)chunk * mp_stride + t * Nc + c] = (float)v;
  }
}

}  // namespace

extern "C" {

int dtx_preprocess(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc, int64_t T,
                   int64_t ser_stride, const double* sos, int nsec, int zerophase, int detrend,
                   float* mp_dev, int64_t mp_stride) {
  if (!ctx) return DTX_E_ARG;
  if (!raw_dev || !mp_dev || nchunk <= 0 || Nc <= 0 || T <= 0 || ser_stride < T ||
      mp_stride < T * Nc)
    return dtx_fail(ctx, DTX_E_ARG, "dtx_preprocess: bad argument");
  if (nsec < 0 || nsec > kMaxSec || (nsec > 0 && !sos))
    return dtx_fail(ctx, DTX_E_ARG, "dtx_preprocess: 0..%d second-order sections supported", kMaxSec);
  DtxDeviceGuard g(ctx->device);
  const int nser = nchunk * Nc;
  const int nblk = (int)((T + kPB - 1) / kPB);
  const int ns = 2 * nsec;

  Sos f;
  f.nsec = nsec;
  for (int s = 0; s < nsec; ++s) {
    const double a0 = sos[6 * s + 3];
    if (a0 == 0.0) return dtx_fail(ctx, DTX_E_ARG, "dtx_preprocess: sos a0 == 0");
    f.b0[s] = sos[6 * s + 0] / a0; f.b1[s] = sos[6 * s + 1] / a0; f.b2[s] = sos[6 * s + 2] / a0;
    f.a1[s] = sos[6 * s + 4] / a0; f.a2[s] = sos[6 * s + 5] / a0;
  }
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

  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
  const size_t o_tr = carve((size_t)nser * 2 * sizeof(double));
  const size_t o_fs = carve((size_t)nser * nblk * 2 * kMaxSec * sizeof(double));
  const size_t o_M = carve((size_t)(ns * ns + 1) * sizeof(double));
  const size_t o_mid = carve(zerophase && nsec ? (size_t)nser * T * sizeof(double) : 0);
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_misc3, off))) return rc;
  char* base = (char*)ctx->ws_misc3.p;
  double* trend = detrend ? (double*)(base + o_tr) : nullptr;
  double* fstate = (double*)(base + o_fs);
  double* Mdev = (double*)(base + o_M);
  double* mid = (double*)(base + o_mid);

  if (detrend) {
    hipLaunchKernelGGL(pp_trend_kernel, dim3((unsigned)nser), dim3(256), 0, ctx->stream, raw_dev, T,
                       ser_stride, trend);
    DTX_HIP(ctx, hipGetLastError());
  }
  if (nsec == 0) {
    dim3 grid((unsigned)std::min<int64_t>((T + 255) / 256, 4096), (unsigned)nser);
    hipLaunchKernelGGL(pp_copy_kernel, grid, dim3(256), 0, ctx->stream, raw_dev, ser_stride,
                       (const double*)trend, T, Nc, mp_stride, mp_dev);
    DTX_HIP(ctx, hipGetLastError());
    return DTX_OK;
  }
  DTX_HIP(ctx, hipMemcpyAsync(Mdev, M.data(), (size_t)ns * ns * sizeof(double), hipMemcpyHostToDevice,
                              ctx->stream));
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // M is a stack-lifetime host buffer
  dim3 grid((unsigned)((nblk + 255) / 256), (unsigned)nser);
  dim3 cgrid((unsigned)((nser + 63) / 64));
  // forward pass: raw - trend -> (mid | output)
  hipLaunchKernelGGL((pp_block_state_kernel<true>), grid, dim3(256), 0, ctx->stream, raw_dev,
                     ser_stride, (const double*)nullptr, (const double*)trend, T, nblk, f, false, fstate);
  hipLaunchKernelGGL(pp_chain_kernel, cgrid, dim3(64), 0, ctx->stream, fstate, (const double*)Mdev,
                     nblk, ns, nser);
  if (zerophase) {
    hipLaunchKernelGGL((pp_block_run_kernel<true, false>), grid, dim3(256), 0, ctx->stream, raw_dev,
                       ser_stride, (const double*)nullptr, (const double*)trend, T, nblk, f, false,
                       (const double*)fstate, mid, (float*)nullptr, Nc, mp_stride);
    // backward pass over the forward output (obspy: sosfilt(sos, firstpass[::-1])[::-1])
    hipLaunchKernelGGL((pp_block_state_kernel<false>), grid, dim3(256), 0, ctx->stream,
                       (const float*)nullptr, ser_stride, (const double*)mid, (const double*)nullptr,
                       T, nblk, f, true, fstate);
    hipLaunchKernelGGL(pp_chain_kernel, cgrid, dim3(64), 0, ctx->stream, fstate, (const double*)Mdev,
                       nblk, ns, nser);
    hipLaunchKernelGGL((pp_block_run_kernel<false, true>), grid, dim3(256), 0, ctx->stream,
                       (const float*)nullptr, ser_stride, (const double*)mid, (const double*)nullptr,
                       T, nblk, f, true, (const double*)fstate, (double*)nullptr, mp_dev, Nc,
                       mp_stride);
  } else {
    hipLaunchKernelGGL((pp_block_run_kernel<true, true>), grid, dim3(256), 0, ctx->stream, raw_dev,
                       ser_stride, (const double*)nullptr, (const double*)trend, T, nblk, f, false,
                       (const double*)fstate, (double*)nullptr, mp_dev, Nc, mp_stride);
  }
  DTX_HIP(ctx, hipGetLastError());
  return DTX_OK;
}

}  // extern "C"
