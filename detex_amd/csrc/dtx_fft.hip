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
//     16 frequencies per thread x Nc complex MACs straight from L2, an in-LDS 8192-point
//     FFT, and the epilogue (Y - S_k mu_p)^2 accumulated in registers; DS leaves the
//     chip once, already normalised -- the K x P matrix of projections never exists.
//
// The FFT: Stockham autosort, radices 16 x 8 x 8 x 8, 16 points per thread in registers,
// three exchanges through a 66 KB LDS image (two workgroups per CU).  Every pass reads
// point (tid + 512 m), m = 0..15, so the first pass is fed from registers (the MAC
// output) and the last pass leaves window (tid + 512 m) in register m: coalesced DS
// stores, and only m < V / 512 is kept.  Twiddles w^1, w^2, w^4 per pass live in
// registers (the other powers are products), loaded once per workgroup from a kN-entry
// float64-built table.
//
// Work: per (pair, block) 13 * 8192 / 2 butterflies instead of 2 * 2 * Nc * L * V MACs --
// ~150x fewer flops than the direct contraction; float32 error grows with log kN
// instead of sqrt(n) (measured 2e-7 at DS = 1 against 2e-6 for the MFMA engines).
#include <complex>

#include "dtx_hankel.h"
#include "dtx_internal.h"

namespace {

constexpr int kN = DTX_FFT_N;        // transform length (complex points)
constexpr int kT = 512;              // threads per workgroup, 16 points each
constexpr int kLdsC = kN + kN / 32;  // complex slots of the padded LDS image
static_assert(kN == 8192, "radix plan 16 x 8 x 8 x 8 is for 8192 points");

typedef float2 c32;

__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
// a * (SG * i)
template <int SG>
__device__ __forceinline__ c32 muli(c32 a) {
  return SG > 0 ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

// natural-order DFTs with kernel e^{SG 2 pi i s q / R}
template <int SG>
__device__ __forceinline__ void dft4(c32& u0, c32& u1, c32& u2, c32& u3) {
  const c32 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = muli<SG>(csub(u1, u3));
  u0 = cadd(a0, a2); u1 = cadd(a1, a3); u2 = csub(a0, a2); u3 = csub(a1, a3);
}
template <int SG>
__device__ __forceinline__ void dft8(c32 (&u)[8]) {
  c32 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6], o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
  dft4<SG>(e0, e1, e2, e3);
  dft4<SG>(o0, o1, o2, o3);
  constexpr float h = 0.70710678118654752440f;
  const float sg = (float)SG;
  o1 = make_float2((o1.x - sg * o1.y) * h, (o1.y + sg * o1.x) * h);     // * (1 + SG i) / sqrt 2
  o2 = muli<SG>(o2);
  o3 = make_float2((-o3.x - sg * o3.y) * h, (-o3.y + sg * o3.x) * h);   // * (-1 + SG i) / sqrt 2
  u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
  u[1] = cadd(e1, o1); u[5] = csub(e1, o1);
  u[2] = cadd(e2, o2); u[6] = csub(e2, o2);
  u[3] = cadd(e3, o3); u[7] = csub(e3, o3);
}
template <int SG>
__device__ __forceinline__ void dft16(c32 (&u)[16]) {
  c32 e[8], o[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { e[q] = u[2 * q]; o[q] = u[2 * q + 1]; }
  dft8<SG>(e);
  dft8<SG>(o);
  // cos / sin (pi q / 8)
  constexpr float cs[8] = {1.f, 0.92387953251128675613f, 0.70710678118654752440f,
                           0.38268343236508977173f, 0.f, -0.38268343236508977173f,
                           -0.70710678118654752440f, -0.92387953251128675613f};
  constexpr float sn[8] = {0.f, 0.38268343236508977173f, 0.70710678118654752440f,
                           0.92387953251128675613f, 1.f, 0.92387953251128675613f,
                           0.70710678118654752440f, 0.38268343236508977173f};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const c32 t = cmul(o[q], make_float2(cs[q], (float)SG * sn[q]));
    u[q] = cadd(e[q], t);
    u[q + 8] = csub(e[q], t);
  }
}

__device__ __forceinline__ int lpad(int a) { return a + (a >> 5); }

// twiddles of one workgroup thread: w^1, w^2, w^4 of the three radix-8 passes (the last
// pass has different ones for its two butterflies), already conjugated for SG < 0
struct Twid {
  c32 p2[3], p3[3], p4[2][3];
};
template <int SG>
__device__ __forceinline__ void load_twiddles(Twid& w, const c32* __restrict__ tab, int tid) {
  auto ld = [&](int idx) {
    c32 t = tab[idx];
    if (SG < 0) t.y = -t.y;
    return t;
  };
  const int k2 = (tid & 15) * (kN / 128), k3 = (tid & 127) * (kN / 1024);
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    w.p2[e] = ld(k2 << e);
    w.p3[e] = ld(k3 << e);
    w.p4[0][e] = ld(tid << e);
    w.p4[1][e] = ld((tid + 512) << e);
  }
}

// one radix-8 pass over the thread's two butterflies (points 2 s + h, s = 0..7)
template <int SG>
__device__ __forceinline__ void pass8(c32 (&v)[16], const c32 (&wa)[3], const c32 (&wb)[3]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const c32 w1 = h ? wb[0] : wa[0], w2 = h ? wb[1] : wa[1], w4 = h ? wb[2] : wa[2];
    const c32 w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
    c32 u[8];
    u[0] = v[h];
    u[1] = cmul(v[2 + h], w1);
    u[2] = cmul(v[4 + h], w2);
    u[3] = cmul(v[6 + h], w3);
    u[4] = cmul(v[8 + h], w4);
    u[5] = cmul(v[10 + h], w5);
    u[6] = cmul(v[12 + h], w6);
    u[7] = cmul(v[14 + h], w7);
    dft8<SG>(u);
#pragma unroll
    for (int s = 0; s < 8; ++s) v[2 * s + h] = u[s];
  }
}

__device__ __forceinline__ void gather(c32 (&v)[16], const c32* lds, int tid) {
#pragma unroll
  for (int m = 0; m < 16; ++m) v[m] = lds[lpad(tid + 512 * m)];
}
// scatter of a radix-8 pass whose finished sub-transform length is p (16 or 128)
template <int P>
__device__ __forceinline__ void scatter8(const c32 (&v)[16], c32* lds, int tid) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = tid + 512 * h, k = i & (P - 1), j = (i - k) * 8 + k;
#pragma unroll
    for (int s = 0; s < 8; ++s) lds[lpad(j + s * P)] = v[2 * s + h];
  }
}

// v[m] = point (tid + 512 m) in, natural-order transform point (tid + 512 m) out
template <int SG>
__device__ __forceinline__ void fft8192(c32 (&v)[16], c32* lds, int tid, const Twid& w) {
  dft16<SG>(v);                                   // pass 1: radix 16, no twiddles
#pragma unroll
  for (int s = 0; s < 16; ++s) lds[lpad(16 * tid + s)] = v[s];
  __syncthreads();
  gather(v, lds, tid);
  __syncthreads();
  pass8<SG>(v, w.p2, w.p2);                       // pass 2: sub-transforms of 16 -> 128
  scatter8<16>(v, lds, tid);
  __syncthreads();
  gather(v, lds, tid);
  __syncthreads();
  pass8<SG>(v, w.p3, w.p3);                       // pass 3: 128 -> 1024
  scatter8<128>(v, lds, tid);
  __syncthreads();
  gather(v, lds, tid);
  __syncthreads();
  pass8<SG>(v, w.p4[0], w.p4[1]);                 // pass 4: 1024 -> 8192, output in place
}

// ---------------------------------------------------------------- block spectra
// Xh[chunk][blk][c][f] = sum_t xp[chunk][c][blk V + t] e^{-2 pi i t f / kN}
__global__ __launch_bounds__(kT) void xhat_kernel(const float* __restrict__ xp, int64_t Tpad,
                                                  int Nc, int V, int nblk,
                                                  const c32* __restrict__ tab,
                                                  c32* __restrict__ xh) {
  __shared__ c32 lds[kLdsC];
  const int tid = threadIdx.x, blk = blockIdx.x, c = blockIdx.y, chunk = blockIdx.z;
  Twid w;
  load_twiddles<-1>(w, tab, tid);
  const float* x = xp + ((int64_t)chunk * Nc + c) * Tpad;
  const int64_t t0 = (int64_t)blk * V;
  c32 v[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int64_t t = t0 + tid + 512 * m;
    v[m] = make_float2(t < Tpad ? x[t] : 0.f, 0.f);
  }
  fft8192<-1>(v, lds, tid, w);
  c32* dst = xh + (((int64_t)chunk * nblk + blk) * Nc + c) * kN;
#pragma unroll
  for (int m = 0; m < 16; ++m) dst[tid + 512 * m] = v[m];
}

// ---------------------------------------------------------------- the scan
struct FftArgs {
  const c32* xh;        // [nchunk][nblk][Nc][kN]
  const c32* W;         // [npairs][Nc][kN]
  const c32* tab;       // [kN] e^{+2 pi i m / kN}
  const int4* unit;     // [nunits] {first pair, pairs, detector of slot 0, detector of slot 1 or -1}
  const float4* pinfo;  // [npairs] {S_k, S_k+1, slot of the imaginary part (0 / 1), 0}
  const float* mean;    // [nchunk][Ppad]
  const float* den;     // [nchunk][Ppad]
  float* ds;            // [nchunk][D][ldp]
  int64_t Ppad, ldp, P;
  int Nc, D, V, nblk, nchunk;
};

template <int VM>   // VM = V / 512 outputs kept per thread
__global__ __launch_bounds__(kT, 2) void ds_fft_kernel(const FftArgs a) {
  __shared__ c32 lds[kLdsC];
  const int tid = threadIdx.x;
  const int per_unit = a.nchunk * a.nblk;
  const int unit = blockIdx.x / per_unit;
  const int rem = blockIdx.x - unit * per_unit;
  const int chunk = rem / a.nblk, blk = rem - chunk * a.nblk;
  const int4 ui = a.unit[unit];
  const int64_t p0 = (int64_t)blk * a.V;
  if (p0 >= a.P) return;

  Twid w;
  load_twiddles<1>(w, a.tab, tid);
  float mu[VM], dn[VM], acc0[VM], acc1[VM];
#pragma unroll
  for (int m = 0; m < VM; ++m) {
    const int64_t p = p0 + tid + 512 * m;
    const bool ok = p < a.P;
    mu[m] = ok ? a.mean[(int64_t)chunk * a.Ppad + p] : 0.f;
    dn[m] = ok ? a.den[(int64_t)chunk * a.Ppad + p] : 0.f;
    acc0[m] = 0.f;
    acc1[m] = 0.f;
  }
  const c32* xh = a.xh + ((int64_t)chunk * a.nblk + blk) * a.Nc * kN + tid;
  for (int pr = 0; pr < ui.y; ++pr) {
    const c32* wp = a.W + (int64_t)(ui.x + pr) * a.Nc * kN + tid;
    c32 z[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) z[m] = make_float2(0.f, 0.f);
    for (int c = 0; c < a.Nc; ++c) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const c32 ww = wp[(int64_t)c * kN + 512 * m], xx = xh[(int64_t)c * kN + 512 * m];
        z[m].x += ww.x * xx.x - ww.y * xx.y;
        z[m].y += ww.x * xx.y + ww.y * xx.x;
      }
    }
    fft8192<1>(z, lds, tid, w);
    const float4 pi = a.pinfo[ui.x + pr];
    const bool two = pi.z != 0.f;
#pragma unroll
    for (int m = 0; m < VM; ++m) {
      const float re = z[m].x - pi.x * mu[m], im = z[m].y - pi.y * mu[m];
      acc0[m] += re * re + (two ? 0.f : im * im);
      acc1[m] += two ? im * im : 0.f;
    }
    __syncthreads();   // the LDS image is reused by the next pair's first scatter
  }
#pragma unroll
  for (int m = 0; m < VM; ++m) {
    const int64_t p = p0 + tid + 512 * m;
    if (p < a.P) {
      float v0 = (dn[m] > 0.f) ? acc0[m] / dn[m] : 0.f;
      if (__builtin_isinf(v0)) v0 = 0.f;
      a.ds[((int64_t)chunk * a.D + ui.z) * a.ldp + p] = v0;
      if (ui.w >= 0) {
        float v1 = (dn[m] > 0.f) ? acc1[m] / dn[m] : 0.f;
        if (__builtin_isinf(v1)) v1 = 0.f;
        a.ds[((int64_t)chunk * a.D + ui.w) * a.ldp + p] = v1;
      }
    }
  }
}

template <int VM>
void launch_fft(dtx_ctx* ctx, const FftArgs& a, unsigned grid) {
  hipLaunchKernelGGL((ds_fft_kernel<VM>), dim3(grid), dim3(kT), 0, ctx->stream, a);
}

// float64 radix-2 transform on the host, kernel e^{+2 pi i j f / n}, unnormalised
void host_fft_pos(std::vector<std::complex<double>>& a) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  const double pi = 3.14159265358979323846264338327950288;
  for (size_t len = 2; len <= n; len <<= 1) {
    std::vector<std::complex<double>> tw(len / 2);
    for (size_t k = 0; k < len / 2; ++k)
      tw[k] = std::complex<double>(std::cos(2 * pi * k / len), std::sin(2 * pi * k / len));
    for (size_t i = 0; i < n; i += len)
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * tw[k];
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
  }
}

}  // namespace

void dtx_fft_free(dtx_ctx* ctx) {
  DtxFft& f = ctx->fft;
  if (f.W_dev) (void)hipFree(f.W_dev);
  if (f.tab_dev) (void)hipFree(f.tab_dev);
  if (f.unit_dev) (void)hipFree(f.unit_dev);
  if (f.pinfo_dev) (void)hipFree(f.pinfo_dev);
  f = DtxFft();
}

bool dtx_fft_eligible(int L) { return L >= kFftMinL && L <= kFftMaxL; }

// hU: the stacked basis [Ktot][n] on the host (multiplexed rows)
int dtx_fft_basis_build(dtx_ctx* ctx, const float* hU, const int32_t* det_off, int D, int n,
                        int Nc) {
  dtx_fft_free(ctx);
  const int L = n / Nc;
  if (!dtx_fft_eligible(L)) return DTX_OK;
  DtxFft& f = ctx->fft;
  f.V = (kN - L + 1) / 512 * 512;

  // units: one detector each; two consecutive single-vector detectors share a pair
  std::vector<int4> units;
  std::vector<float4> pinfo;
  std::vector<int> prow;   // per pair: row of the real part, row of the imaginary part (-1: none)
  auto rowsum = [&](int r) {
    double s = 0.0;
    for (int q = 0; q < n; ++q) s += (double)hU[(size_t)r * n + q];
    return (float)s;
  };
  for (int d = 0; d < D;) {
    const int r0 = det_off[d], k = det_off[d + 1] - r0;
    int4 u;
    u.x = (int)pinfo.size();
    if (k == 1 && d + 1 < D && det_off[d + 2] - det_off[d + 1] == 1) {
      u.y = 1; u.z = d; u.w = d + 1;
      pinfo.push_back(make_float4(rowsum(r0), rowsum(r0 + 1), 1.f, 0.f));
      prow.push_back(r0); prow.push_back(r0 + 1);
      d += 2;
    } else {
      u.y = (k + 1) / 2; u.z = d; u.w = -1;
      for (int q = 0; q < k; q += 2) {
        const bool has2 = q + 1 < k;
        pinfo.push_back(make_float4(rowsum(r0 + q), has2 ? rowsum(r0 + q + 1) : 0.f, 0.f, 0.f));
        prow.push_back(r0 + q); prow.push_back(has2 ? r0 + q + 1 : -1);
      }
      d += 1;
    }
    units.push_back(u);
  }
  f.nunits = (int)units.size();
  f.npairs = (int)pinfo.size();

  std::vector<float2> W((size_t)f.npairs * Nc * kN);
  std::vector<std::complex<double>> z(kN);
  for (int pr = 0; pr < f.npairs; ++pr)
    for (int c = 0; c < Nc; ++c) {
      const int ra = prow[2 * pr], rb = prow[2 * pr + 1];
      for (int j = 0; j < kN; ++j) z[j] = 0.0;
      for (int j = 0; j < L; ++j)
        z[j] = std::complex<double>((double)hU[(size_t)ra * n + (size_t)j * Nc + c],
                                    rb >= 0 ? (double)hU[(size_t)rb * n + (size_t)j * Nc + c] : 0.0);
      host_fft_pos(z);
      float2* dst = W.data() + ((size_t)pr * Nc + c) * kN;
      for (int q = 0; q < kN; ++q)
        dst[q] = make_float2((float)(z[q].real() / kN), (float)(z[q].imag() / kN));
    }
  std::vector<float2> tab(kN);
  const double pi = 3.14159265358979323846264338327950288;
  for (int m = 0; m < kN; ++m)
    tab[m] = make_float2((float)std::cos(2 * pi * m / kN), (float)std::sin(2 * pi * m / kN));

  DTX_HIP(ctx, hipMalloc(&f.W_dev, W.size() * sizeof(float2)));
  DTX_HIP(ctx, hipMalloc(&f.tab_dev, tab.size() * sizeof(float2)));
  DTX_HIP(ctx, hipMalloc(&f.unit_dev, units.size() * sizeof(int4)));
  DTX_HIP(ctx, hipMalloc(&f.pinfo_dev, pinfo.size() * sizeof(float4)));
  DTX_HIP(ctx, hipMemcpy(f.W_dev, W.data(), W.size() * sizeof(float2), hipMemcpyHostToDevice));
  DTX_HIP(ctx, hipMemcpy(f.tab_dev, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
  DTX_HIP(ctx, hipMemcpy(f.unit_dev, units.data(), units.size() * sizeof(int4), hipMemcpyHostToDevice));
  DTX_HIP(ctx, hipMemcpy(f.pinfo_dev, pinfo.data(), pinfo.size() * sizeof(float4),
                         hipMemcpyHostToDevice));
  f.ok = true;
  return DTX_OK;
}

// xp / mean / den: the channel planes and window statistics dtx_subspace_ds prepared
int dtx_fft_scan(dtx_ctx* ctx, const float* xp, int64_t Tpad, const float* mean, const float* den,
                 int64_t Ppad, int64_t P, int nchunk, float* ds, int64_t ldp) {
  DtxFft& f = ctx->fft;
  if (!f.ok) return dtx_fail(ctx, DTX_E_STATE, "FFT engine: no spectra for this basis");
  const int Nc = ctx->Nc, V = f.V;
  const int nblk = (int)((P + V - 1) / V);
  const int64_t grid = (int64_t)f.nunits * nchunk * nblk;
  if (grid >= ((int64_t)1 << 31)) return dtx_fail(ctx, DTX_E_ARG, "FFT engine: grid too large");
  int rc;
  if ((rc = dtx_reserve(ctx, ctx->ws_xhat, (size_t)nchunk * nblk * Nc * kN * sizeof(float2)))) return rc;
  c32* xh = (c32*)ctx->ws_xhat.p;
  hipLaunchKernelGGL(xhat_kernel, dim3((unsigned)nblk, (unsigned)Nc, (unsigned)nchunk), dim3(kT), 0,
                     ctx->stream, xp, Tpad, Nc, V, nblk, (const c32*)f.tab_dev, xh);
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k0, ctx->stream));   // times the dominant kernel only
  FftArgs a;
  a.xh = xh; a.W = (const c32*)f.W_dev; a.tab = (const c32*)f.tab_dev;
  a.unit = (const int4*)f.unit_dev; a.pinfo = (const float4*)f.pinfo_dev;
  a.mean = mean; a.den = den; a.ds = ds; a.Ppad = Ppad; a.ldp = ldp; a.P = P;
  a.Nc = Nc; a.D = ctx->D; a.V = V; a.nblk = nblk; a.nchunk = nchunk;
  switch (V / 512) {
    case 1: launch_fft<1>(ctx, a, (unsigned)grid); break;
    case 2: launch_fft<2>(ctx, a, (unsigned)grid); break;
    case 3: launch_fft<3>(ctx, a, (unsigned)grid); break;
    case 4: launch_fft<4>(ctx, a, (unsigned)grid); break;
    case 5: launch_fft<5>(ctx, a, (unsigned)grid); break;
    case 6: launch_fft<6>(ctx, a, (unsigned)grid); break;
    case 7: launch_fft<7>(ctx, a, (unsigned)grid); break;
    case 8: launch_fft<8>(ctx, a, (unsigned)grid); break;
    case 9: launch_fft<9>(ctx, a, (unsigned)grid); break;
    case 10: launch_fft<10>(ctx, a, (unsigned)grid); break;
    case 11: launch_fft<11>(ctx, a, (unsigned)grid); break;
    case 12: launch_fft<12>(ctx, a, (unsigned)grid); break;
    case 13: launch_fft<13>(ctx, a, (unsigned)grid); break;
    case 14: launch_fft<14>(ctx, a, (unsigned)grid); break;
    case 15: launch_fft<15>(ctx, a, (unsigned)grid); break;
    default: return dtx_fail(ctx, DTX_E_STATE, "FFT engine: bad block advance %d", V);
  }
  DTX_HIP(ctx, hipGetLastError());
  DTX_HIP(ctx, hipEventRecord(ctx->ev_k1, ctx->stream));
  ctx->kernel_timed = true;
  return DTX_OK;
}
