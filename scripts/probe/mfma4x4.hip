// Probe of v_mfma_f32_4x4x1_16B_f32 operand / result layout and CBSZ / ABID broadcast.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* c) {
  int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) c[l * 4 + i] = acc[i];
}
int main() {
  float ha[64], hb[64], hc[256];
  float *da, *db, *dc;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dc, 1024);
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 * (1 + l); }
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  auto show = [&](const char* t) {
    hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
    printf("%s\n", t);
    for (int l : {0, 1, 3, 4, 5, 17, 32, 33, 63}) {
      printf(" lane %2d:", l);
      for (int i = 0; i < 4; ++i) printf(" %10.0f (a=%g b=%g)", hc[l * 4 + i], hc[l*4+i] / hb[l], hc[l*4+i] / 100.0f / (1+l) );
      printf("\n");
    }
  };
  hipLaunchKernelGGL((k<0, 0>), 1, 64, 0, 0, da, db, dc); show("cbsz0");
  hipLaunchKernelGGL((k<3, 0>), 1, 64, 0, 0, da, db, dc); show("cbsz3 abid0");
  hipLaunchKernelGGL((k<3, 2>), 1, 64, 0, 0, da, db, dc); show("cbsz3 abid2");
  hipLaunchKernelGGL((k<4, 5>), 1, 64, 0, 0, da, db, dc); show("cbsz4 abid5");
  return 0;
}
