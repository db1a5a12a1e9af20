// How fast can a kernel read 886 MB on this chip?  Read-only sweeps over block size, loads in flight,
// grid shape (one tile per workgroup / persistent grid-stride) and cache hints.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/detex_hip.h"

typedef float fvec4 __attribute__((ext_vector_type(4)));

template <int T, int U, int NT>
__global__ __launch_bounds__(T) void rd(const fvec4* __restrict__ x, long n4, float* __restrict__ out, int persistent) {
  float acc = 0.f;
  const long tile = (long)T * U;
  const long ntile = (n4 + tile - 1) / tile;
  for (long t = blockIdx.x; t < ntile; t += persistent ? gridDim.x : ntile) {
    fvec4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = t * tile + (long)u * T + threadIdx.x;
      if (i < n4) v[u] = NT ? __builtin_nontemporal_load(x + i) : x[i];
      else v[u] = (fvec4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
  }
  if (acc == 123.456f) out[0] = acc;
}

template <int T, int U, int NT>
void run(const fvec4* x, long n4, float* out, int persistent, int gmul) {
  const long tile = (long)T * U;
  const long ntile = (n4 + tile - 1) / tile;
  const unsigned grid = persistent ? 256u * gmul : (unsigned)ntile;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rd<T, U, NT>), dim3(grid), dim3(T), 0, 0, x, n4, out, persistent);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms / 10);
  }
  printf("T=%4d U=%2d nt=%d %s grid=%7u : %7.1f us  %.2f TB/s\n", T, U, NT, persistent ? "persist" : "tiles  ", grid,
         best * 1e3, n4 * 16.0 / best / 1e9);
}

// the post pass's own loop structure: rows of P floats at pitch ldp, grid (tiles, D, nchunk), ITEMS floats per thread in
// batches of B 16-byte loads; MODE 0: maximum only, branch-free; 1: + the wave-uniform "ordinary" test and 64-bit indices
template <int ITEMS, int B, int NT, int MODE>
__global__ __launch_bounds__(256) void rows(const float* __restrict__ ds, long P, long ldp, int D, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const float* row = ds + ((long)blockIdx.z * D + blockIdx.y) * ldp;
  const long base = (long)blockIdx.x * (256 * ITEMS);
  int vmax = 0;
  unsigned wc = 0;
  for (int it0 = 0; it0 < ITEMS / 4; it0 += B) {
    fvec4 v[B];
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const long p4 = base + ((long)(it0 + u) * 256 + tid) * 4;
      if (p4 + 3 < P) v[u] = NT ? __builtin_nontemporal_load((const fvec4*)(row + p4)) : *(const fvec4*)(row + p4);
      else v[u] = (fvec4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < B; ++u) {
      const float q[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 1) {
          const long p = base + ((long)(it0 + u) * 256 + tid) * 4 + j;
          const bool ordinary = p < P && q[j] >= 0.f && q[j] < 0.01f;
          if (__all(ordinary)) { wc += (unsigned)__popcll(__ballot(q[j] < 0.0025f)); }
          else if (p < P) wc += 1000;
        }
        vmax = max(vmax, __float_as_int(q[j]));
      }
    }
  }
  if (vmax == 12345 || wc == 77) out[0] = (float)vmax;
}

template <int ITEMS, int B, int NT, int MODE>
void run_rows(const float* x, float* out) {
  const long P = 369001, ldp = 369024;
  const int D = 25, nchunk = 24;
  const long per = 256L * ITEMS;
  dim3 grid((unsigned)((P + per - 1) / per), D, nchunk);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((rows<ITEMS, B, NT, MODE>), grid, dim3(256), 0, 0, x, P, ldp, D, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms / 10);
  }
  printf("rows ITEMS=%3d B=%2d nt=%d mode=%d grid.x=%3u : %7.1f us  %.2f TB/s\n", ITEMS, B, NT, MODE, grid.x, best * 1e3,
         (double)P * D * nchunk * 4 / best / 1e9);
}

int main() {
  const long n4 = 886L * 1000 * 1000 / 16;
  fvec4* x; float* out;
  hipMalloc(&x, n4 * 16); hipMalloc(&out, 4);
  hipMemset(x, 0, n4 * 16);
  run<256, 4, 0>(x, n4, out, 0, 0);
  run<256, 8, 0>(x, n4, out, 0, 0);
  run<256, 16, 0>(x, n4, out, 0, 0);
  run<512, 8, 0>(x, n4, out, 0, 0);
  run<1024, 4, 0>(x, n4, out, 0, 0);
  run<1024, 8, 0>(x, n4, out, 0, 0);
  run<256, 8, 1>(x, n4, out, 0, 0);
  run<256, 16, 1>(x, n4, out, 0, 0);
  run<1024, 8, 1>(x, n4, out, 0, 0);
  for (int g : {4, 8, 16}) {
    run<256, 4, 0>(x, n4, out, 1, g);
    run<256, 8, 0>(x, n4, out, 1, g);
    run<256, 8, 1>(x, n4, out, 1, g);
    run<512, 8, 0>(x, n4, out, 1, g);
  }
  run<1024, 8, 0>(x, n4, out, 1, 2);
  run<1024, 4, 0>(x, n4, out, 1, 2);
  const float* xf = (const float*)x;
  {
    // the library's own post pass on the same buffer: zeros, then noise-like values
    dtx_ctx* ctx = nullptr;
    if (dtx_ctx_create(0, nullptr, 0, &ctx) != 0) { printf("ctx: %s\n", dtx_last_error(nullptr)); return 1; }
    uint32_t* hist; float* maxds; float* thr; uint64_t* key; float* val; uint32_t* cnt;
    hipMalloc(&hist, 25 * 400 * 4); hipMalloc(&maxds, 24 * 25 * 4); hipMalloc(&thr, 25 * 4);
    hipMalloc(&key, 8 << 20); hipMalloc(&val, 4 << 20); hipMalloc(&cnt, 4);
    hipMemset(hist, 0, 25 * 400 * 4); hipMemset(cnt, 0, 4);
    std::vector<float> th(25, 0.3f);
    hipMemcpy(thr, th.data(), 100, hipMemcpyHostToDevice);
    for (int data = 0; data < 3; ++data) {
      if (data >= 1) {
        std::vector<float> h(1 << 22);
        unsigned r = 12345u;
        for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (r >> 8) * ((data == 1 ? 0.009f : 0.5f) / 16777216.f); }
        for (long o = 0; o < n4 * 4; o += (long)h.size())
          hipMemcpy((float*)x + o, h.data(), (size_t)std::min<long>((long)h.size(), n4 * 4 - o) * 4, hipMemcpyHostToDevice);
      }
      for (int mode = 0; mode < 3; ++mode) {
        float ms = 0.f, best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          dtx_timer_start(ctx);
          for (int i = 0; i < 10; ++i)
            dtx_ds_post(ctx, xf, 369001, 369024, 24, 25, mode >= 1 ? hist : nullptr, maxds, mode == 2 ? thr : nullptr,
                        mode == 2 ? key : nullptr, mode == 2 ? val : nullptr, 1 << 20, mode == 2 ? cnt : nullptr);
          dtx_timer_stop(ctx, &ms);
          best = fminf(best, ms / 10);
        }
        printf("dtx_ds_post data=%s mode=%d : %7.1f us  %.2f TB/s\n", data == 0 ? "zeros" : data == 1 ? "noise" : "loud ", mode, best * 1e3,
               369001.0 * 25 * 24 * 4 / best / 1e9);
      }
      run_rows<64, 4, 0, 1>(xf, out);
      run_rows<64, 4, 1, 1>(xf, out);
    }
  }
  run_rows<64, 4, 0, 0>(xf, out);
  run_rows<64, 4, 1, 0>(xf, out);
  run_rows<64, 16, 0, 0>(xf, out);
  run_rows<64, 16, 1, 0>(xf, out);
  run_rows<16, 4, 0, 0>(xf, out);
  run_rows<16, 4, 1, 0>(xf, out);
  run_rows<32, 8, 1, 0>(xf, out);
  run_rows<64, 4, 0, 1>(xf, out);
  run_rows<64, 4, 1, 1>(xf, out);
  run_rows<16, 4, 1, 1>(xf, out);
  run_rows<32, 8, 1, 1>(xf, out);
  run_rows<64, 16, 1, 1>(xf, out);
  return 0;
}
