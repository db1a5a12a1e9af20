// Sustained matrix-pipe rate of this MI355X, registers only (no LDS, no memory in the
// loop): the number the split-bf16 engine's issued MFMA rate has to be read against.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
// Prints, per (instruction, operand data, waves per SIMD): TFLOP/s over a >= 100 ms launch
// and the shader clock it implies (cycles per MFMA are fixed: 16 for 16x16x32 bf16,
// 32 for 16x16x4 f32).  Random operands toggle the multipliers; zeros do not -- the gap
// between the two is the power limit, not the kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kAcc = 12;  // independent accumulator tiles per wave

template <bool BF16>
__global__ __launch_bounds__(512, 2) void spin(const i32x4* __restrict__ src, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  i32x4 a0 = src[lane], a1 = src[64 + lane], b0 = src[128 + lane], b1 = src[192 + lane];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      if constexpr (BF16) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            __builtin_bit_cast(bf16x8, (i & 1) ? a1 : a0), __builtin_bit_cast(bf16x8, (i & 2) ? b1 : b0),
            acc[i], 0, 0, 0);
      } else {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, (i & 1) ? a1[0] : a0[0]),
                                                      __builtin_bit_cast(float, (i & 2) ? b1[0] : b0[0]),
                                                      acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;  // keep the loop alive
}

static uint16_t bf16_of(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool BF16>
static void run(const char* label, const i32x4* src, float* out, int blocks, int threads, double flop_per_mfma,
                double cyc_per_mfma) {
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {   // grow the launch until it lasts >= 150 ms
    hipEventRecord(e0);
    hipLaunchKernelGGL((spin<BF16>), dim3(blocks), dim3(threads), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (ms >= 150.f) break;
    iters = (int)(iters * 200.0 / (ms > 0.05f ? ms : 0.05f));
  }
  const double waves = (double)blocks * threads / 64.0;
  const double mfma = waves * (double)iters * kAcc;
  const double tflops = mfma * flop_per_mfma / (ms * 1e-3) / 1e12;
  // every SIMD runs the MFMAs of its resident waves back to back
  const double simds = 256.0 * 4.0;
  const double ghz = mfma * cyc_per_mfma / simds / (ms * 1e-3) / 1e9;
  printf("%-44s %8.1f ms  %8.1f TFLOP/s  pipe-busy-equivalent clock %.3f GHz\n", label, ms, tflops, ghz);
}

int main() {
  std::vector<uint16_t> h(4 * 64 * 8);
  srand(1);
  i32x4* src;
  float* out;
  hipMalloc(&src, h.size() * 2);
  hipMalloc(&out, 64);
  for (int mode = 0; mode < 3; ++mode) {
    // 0: zeros, 1: random normal-ish values ("hi" parts), 2: random values 2^-9 smaller ("lo" parts mixed in)
    for (size_t i = 0; i < h.size(); ++i) {
      float v = (float)rand() / RAND_MAX * 2.f - 1.f;
      if (mode == 0) v = 0.f;
      if (mode == 2 && (i / 512) % 2 == 1) v *= 1.0f / 512.f;
      h[i] = bf16_of(v);
    }
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const char* dn = mode == 0 ? "zeros" : (mode == 1 ? "random" : "random hi x lo");
    char label[128];
    snprintf(label, sizeof label, "bf16 16x16x32, %s, 2 wave/SIMD", dn);
    run<true>(label, src, out, 256 * 8, 512, 2.0 * 16 * 16 * 32, 16.0);
    snprintf(label, sizeof label, "f32 16x16x4, %s, 2 wave/SIMD", dn);
    run<false>(label, src, out, 256 * 8, 512, 2.0 * 16 * 16 * 4, 32.0);
  }
  return 0;
}
