/* Plain-C user of the boundary (include/detex_hip.h): what a cgo / JNI / FFI binding
 * of another host language would do.  Scores one synthetic data block against two
 * detectors and prints the best window of each.
 *
 *   gcc -std=c99 -Iinclude examples/ds_host.c -Ldetex_amd -ldetexhip -lm \
 *       -Wl,-rpath,$PWD/detex_amd -o ds_host && ./ds_host
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "detex_hip.h"

static float frand(unsigned* s) { /* uniform in [-1, 1) */
  *s = *s * 1664525u + 1013904223u;
  return (float)((*s >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

int main(void) {
  enum { Nc = 3, L = 200, n = Nc * L, K = 5, D = 2, T = 6000, M = Nc * T };
  const int32_t det_off[D + 1] = {0, 3, 5};
  float* U = malloc(sizeof(float) * K * n);
  float* x = malloc(sizeof(float) * M);
  unsigned seed = 7;
  for (int k = 0; k < K; ++k) {           /* unit-norm random rows */
    double ss = 0;
    for (int q = 0; q < n; ++q) { U[k * n + q] = frand(&seed); ss += (double)U[k * n + q] * U[k * n + q]; }
    for (int q = 0; q < n; ++q) U[k * n + q] /= (float)sqrt(ss);
  }
  for (int i = 0; i < M; ++i) x[i] = frand(&seed);
  const int plant[D] = {1500, 4200};      /* plant row 0 of each detector */
  for (int d = 0; d < D; ++d)
    for (int q = 0; q < n; ++q) x[Nc * plant[d] + q] += 40.0f * U[det_off[d] * n + q];

  dtx_ctx* ctx = NULL;
  if (dtx_ctx_create(0, NULL, 1, &ctx)) { fprintf(stderr, "%s\n", dtx_last_error(NULL)); return 2; }
  if (dtx_basis_upload(ctx, U, det_off, D, n, Nc)) { fprintf(stderr, "%s\n", dtx_last_error(ctx)); return 3; }
  const int64_t P = dtx_num_windows(M, n, Nc);
  float* ds = malloc(sizeof(float) * D * P);
  if (dtx_subspace_ds_host(ctx, x, M, ds)) { fprintf(stderr, "%s\n", dtx_last_error(ctx)); return 4; }
  int ok = 1;
  for (int d = 0; d < D; ++d) {
    int64_t best = 0;
    for (int64_t p = 1; p < P; ++p) if (ds[d * P + p] > ds[d * P + best]) best = p;
    printf("detector %d: max DS %.4f at window %lld (planted at %d)\n", d, ds[d * P + best],
           (long long)best, plant[d]);
    ok = ok && best == plant[d] && ds[d * P + best] > 0.5f;
  }
  /* Several GPUs from one process: one context per device, the basis replicated with
   * dtx_basis_bcast, the data blocks dealt round-robin -- days shard embarrassingly, nothing else is
   * exchanged.  (With a single device the second context simply shares it.) */
  {
    enum { NB = 4 };
    const int ndev = dtx_device_count();
    const int nctx = ndev > 1 ? ndev : 2;
    dtx_ctx** cs = calloc((size_t)nctx, sizeof(dtx_ctx*));
    cs[0] = ctx;
    for (int i = 1; i < nctx; ++i)
      if (dtx_ctx_create(i % ndev, NULL, 1, &cs[i])) { fprintf(stderr, "%s\n", dtx_last_error(NULL)); return 5; }
    if (dtx_basis_bcast(cs, nctx)) { fprintf(stderr, "%s\n", dtx_last_error(ctx)); return 6; }
    for (int b = 0; b < NB; ++b) {            /* block b = the same stream shifted by b samples */
      dtx_ctx* c = cs[b % nctx];
      if (dtx_subspace_ds_host(c, x + Nc * b, M - Nc * NB, ds)) { fprintf(stderr, "%s\n", dtx_last_error(c)); return 7; }
      const int64_t Pb = dtx_num_windows(M - Nc * NB, n, Nc);
      int64_t best = 0;
      for (int64_t p = 1; p < Pb; ++p) if (ds[p] > ds[best]) best = p;
      printf("block %d on context %d: detector 0 peaks at window %lld\n", b, b % nctx, (long long)best);
      ok = ok && best == plant[0] - b;
    }
    for (int i = 1; i < nctx; ++i) dtx_ctx_destroy(cs[i]);
    free(cs);
  }
  dtx_ctx_destroy(ctx);
  free(U); free(x); free(ds);
  return ok ? 0 : 1;
}
