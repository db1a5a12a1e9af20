// Context, error reporting, workspaces and HIP-event timers of libdetexhip.so.
#include "dtx_internal.h"
#include "dtx_plan.h"

static thread_local std::string g_create_err;

int dtx_fail(dtx_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  else g_create_err = buf;
  return code;
}

int dtx_reserve(dtx_ctx* ctx, DtxBuf& b, size_t bytes) {
  if (bytes <= b.bytes) return DTX_OK;
  if (b.p) {
    // the old buffer may still be in use by work queued on the stream
    DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DTX_HIP(ctx, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  size_t want = dtxplan::reserve_capacity(bytes);
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    return dtx_fail(ctx, DTX_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  }
  b.bytes = want;
  return DTX_OK;
}

void dtx_free_tiles_internal(dtx_ctx* c) {
  for (auto& t : c->tiles) {
    if (t.ut_dev) (void)hipFree(t.ut_dev);
    if (t.ut16_dev) (void)hipFree(t.ut16_dev);
    if (t.srow_dev) (void)hipFree(t.srow_dev);
    if (t.doff_dev) (void)hipFree(t.doff_dev);
    if (t.sub_dev) (void)hipFree(t.sub_dev);
    if (t.sdoff_dev) (void)hipFree(t.sdoff_dev);
  }
  c->tiles.clear();
  dtx_fft_free(c);
}

int dtx_stage_h2d(dtx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return DTX_OK;
  const int k = ctx->stage_next;
  ctx->stage_next = (k + 1) % dtx_ctx::kStage;
  if (!ctx->stage_ev[k]) DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
  else DTX_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k]));      // (eight copies ago: long done)
  if (ctx->stage_cap[k] < bytes) {
    if (ctx->stage_buf[k]) DTX_HIP(ctx, hipHostFree(ctx->stage_buf[k]));
    ctx->stage_buf[k] = nullptr;
    ctx->stage_cap[k] = 0;
    const size_t want = dtxplan::stage_capacity(bytes);
    DTX_HIP(ctx, hipHostMalloc(&ctx->stage_buf[k], want, hipHostMallocDefault));
    ctx->stage_cap[k] = want;
  }
  memcpy(ctx->stage_buf[k], src_host, bytes);
  DTX_HIP(ctx, hipMemcpyAsync(dst_dev, ctx->stage_buf[k], bytes, hipMemcpyHostToDevice, ctx->stream));
  DTX_HIP(ctx, hipEventRecord(ctx->stage_ev[k], ctx->stream));
  return DTX_OK;
}

int dtx_kev_next(dtx_ctx* ctx) {
  const int slot = dtxplan::ring_next_slot(ctx->kev_count, dtx_ctx::kKev);
  // (each event of the pair on its own: a create that failed half-way is completed by the next call)
  if (!ctx->kev0[slot]) DTX_HIP(ctx, hipEventCreate(&ctx->kev0[slot]));
  if (!ctx->kev1[slot]) DTX_HIP(ctx, hipEventCreate(&ctx->kev1[slot]));
  ctx->ev_k0 = ctx->kev0[slot];
  ctx->ev_k1 = ctx->kev1[slot];
  return DTX_OK;
}

// The pair handed out by dtx_kev_next has been recorded on both sides of the kernel: only now does the
// slot count (a launch path that failed in between leaves the ring as it was -- the slot is reused).
void dtx_kev_commit(dtx_ctx* ctx) {
  ++ctx->kev_count;
  ctx->kernel_timed = true;
}

int dtx_aux_fork(dtx_ctx* ctx) {
  if (!ctx->aux) {
    DTX_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  if (ctx->aux_is_main) return DTX_OK;
  DTX_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  DTX_HIP(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
  return DTX_OK;
}

int dtx_aux_join(dtx_ctx* ctx) {
  if (ctx->aux_is_main) return DTX_OK;
  DTX_HIP(ctx, hipEventRecord(ctx->ev_join, ctx->aux));
  DTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return DTX_OK;
}

int dtx_wait_post(dtx_ctx* ctx) {
  if (ctx->post_pending) {
    DTX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_post_done, 0));
    ctx->post_pending = false;
  }
  return DTX_OK;
}

extern "C" {

int dtx_version(void) { return DTX_ABI_VERSION; }

int dtx_chunk_table_check(const int64_t* chunk_tab_host, int nchunk, int64_t M, int64_t numel) {
  if (!chunk_tab_host || nchunk <= 0 || M < 0 || numel < 0) return 1;
  return dtxplan::check_chunk_table(chunk_tab_host, nchunk, M, numel);
}

const char* dtx_last_error(dtx_ctx* ctx) {
  return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

int dtx_ctx_create(int device, void* stream, int own_stream, dtx_ctx** out) {
  if (!out) return dtx_fail(nullptr, DTX_E_ARG, "dtx_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return dtx_fail(nullptr, DTX_E_HIP, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev)
    return dtx_fail(nullptr, DTX_E_ARG, "device %d out of range [0,%d)", device, ndev);
  DtxDeviceGuard g(device);
  dtx_ctx* c = new dtx_ctx();
  c->device = device;
  if (hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->ncu <= 0)
    c->ncu = 256;
  if (!own_stream) {
    c->stream = (hipStream_t)stream;   // NULL = the default stream
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return dtx_fail(nullptr, DTX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  hipEvent_t* evs[2] = {&c->ev_t0, &c->ev_t1};
  for (auto ev : evs) {
    e = hipEventCreate(ev);
    if (e != hipSuccess) {
      (void)dtx_ctx_destroy(c);          // releases the stream and the events made so far
      return dtx_fail(nullptr, DTX_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
    }
  }
  *out = c;
  return DTX_OK;
}

int dtx_ctx_destroy(dtx_ctx* ctx) {
  if (!ctx) return DTX_OK;
  DtxDeviceGuard g(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->post_stream && ctx->ev_post_done && ctx->post_on_side) (void)hipEventSynchronize(ctx->ev_post_done);
  if (ctx->aux && !ctx->aux_is_main) (void)hipStreamSynchronize(ctx->aux);
  if (ctx->ev_trig) {
    if (ctx->trig_pending) (void)hipEventSynchronize(ctx->ev_trig);
    (void)hipEventDestroy(ctx->ev_trig);
  }
  dtx_free_tiles_internal(ctx);
  DtxBuf* bufs[] = {&ctx->ws_xp, &ctx->ws_mean, &ctx->ws_den,
                    &ctx->ws_misc, &ctx->ws_misc2, &ctx->ws_misc3, &ctx->ws_g, &ctx->ws_csum,
                    &ctx->ws_xhat, &ctx->ws_flag, &ctx->U_buf, &ctx->srow_all, &ctx->ws_trig,
                    &ctx->ws_trig_doff, &ctx->doff_all};
  for (auto b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto& b : ctx->pp_M)
    if (b.p) (void)hipFree(b.p);
  for (int k = 0; k < dtx_ctx::kStage; ++k) {
    if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
    if (ctx->stage_buf[k]) (void)hipHostFree(ctx->stage_buf[k]);
  }
  hipEvent_t evs[4] = {ctx->ev_t0, ctx->ev_t1, ctx->ev_post_ready, ctx->ev_post_done};
  for (auto ev : evs)
    if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < dtx_ctx::kKev; ++i) {
    if (ctx->kev0[i]) (void)hipEventDestroy(ctx->kev0[i]);
    if (ctx->kev1[i]) (void)hipEventDestroy(ctx->kev1[i]);
  }
  if (ctx->aux && !ctx->aux_is_main) {
    (void)hipStreamSynchronize(ctx->aux);
    (void)hipStreamDestroy(ctx->aux);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return DTX_OK;
}

int dtx_ctx_sync(dtx_ctx* ctx) {
  if (!ctx) return DTX_E_ARG;
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->post_stream && ctx->post_on_side) DTX_HIP(ctx, hipStreamSynchronize(ctx->post_stream));   // (a scan's post pass)
  ctx->post_pending = false;
  return DTX_OK;
}

void* dtx_ctx_stream(dtx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int dtx_ctx_set_trigger_stream(dtx_ctx* ctx, void* stream) {
  if (!ctx) return DTX_E_ARG;
  ctx->trig_stream = (hipStream_t)stream;     // (launches on different streams are ordered by event: dtx_trig.hip)
  return DTX_OK;
}

int dtx_ctx_set_post_stream(dtx_ctx* ctx, void* stream) {
  if (!ctx) return DTX_E_ARG;
  DtxDeviceGuard g(ctx->device);
  { const int rc = dtx_wait_post(ctx); if (rc) return rc; }
  ctx->post_stream = (hipStream_t)stream;
  if (stream && !ctx->ev_post_ready) {
    DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_post_ready, hipEventDisableTiming));
    DTX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_post_done, hipEventDisableTiming));
  }
  return DTX_OK;
}

int dtx_stream_wait_post(dtx_ctx* ctx, void* stream) {
  if (!ctx) return DTX_E_ARG;
  if (!ctx->post_on_side || !ctx->ev_post_done) return DTX_OK;
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_post_done, 0));
  if ((hipStream_t)stream == ctx->stream) ctx->post_pending = false;
  return DTX_OK;
}

int dtx_stream_wait_scan_kernel(dtx_ctx* ctx, void* stream) {
  if (!ctx) return DTX_E_ARG;
  if (!ctx->kernel_timed || !ctx->ev_k1) return DTX_OK;
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, ctx->ev_k1, 0));
  return DTX_OK;
}

int dtx_timer_start(dtx_ctx* ctx) {
  if (!ctx) return DTX_E_ARG;
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));
  return DTX_OK;
}

int dtx_timer_stop(dtx_ctx* ctx, float* ms) {
  if (!ctx || !ms) return DTX_E_ARG;
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
  DTX_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
  DTX_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_t0, ctx->ev_t1));
  return DTX_OK;
}

int dtx_kernel_ms_mean(dtx_ctx* ctx, int n, float* ms) {
  if (!ctx || !ms || n <= 0) return DTX_E_ARG;
  if (!ctx->kernel_timed) return dtx_fail(ctx, DTX_E_STATE, "no timed kernel has run yet");
  int slots[dtx_ctx::kKev];
  if (!dtxplan::ring_last_slots(ctx->kev_count, dtx_ctx::kKev, n, slots))
    return dtx_fail(ctx, DTX_E_ARG, "dtx_kernel_ms_mean: only the last %d launches are kept (%lld timed so far)",
                    dtx_ctx::kKev, ctx->kev_count);
  DtxDeviceGuard g(ctx->device);
  double sum = 0.0;
  for (int q = 0; q < n; ++q) {
    const int slot = slots[q];
    float t = 0.f;
    DTX_HIP(ctx, hipEventSynchronize(ctx->kev1[slot]));
    DTX_HIP(ctx, hipEventElapsedTime(&t, ctx->kev0[slot], ctx->kev1[slot]));
    sum += t;
  }
  *ms = (float)(sum / n);
  return DTX_OK;
}

int dtx_kernel_ms_list(dtx_ctx* ctx, int n, float* ms) {
  if (!ctx || !ms || n <= 0) return DTX_E_ARG;
  if (!ctx->kernel_timed) return dtx_fail(ctx, DTX_E_STATE, "no timed kernel has run yet");
  int slots[dtx_ctx::kKev];
  if (!dtxplan::ring_last_slots(ctx->kev_count, dtx_ctx::kKev, n, slots))
    return dtx_fail(ctx, DTX_E_ARG, "dtx_kernel_ms_list: only the last %d launches are kept (%lld timed so far)",
                    dtx_ctx::kKev, ctx->kev_count);
  DtxDeviceGuard g(ctx->device);
  for (int q = 0; q < n; ++q) {
    DTX_HIP(ctx, hipEventSynchronize(ctx->kev1[slots[q]]));
    DTX_HIP(ctx, hipEventElapsedTime(&ms[q], ctx->kev0[slots[q]], ctx->kev1[slots[q]]));
  }
  return DTX_OK;
}

int dtx_last_kernel_ms(dtx_ctx* ctx, float* ms) {
  if (!ctx || !ms) return DTX_E_ARG;
  if (!ctx->kernel_timed) return dtx_fail(ctx, DTX_E_STATE, "no timed kernel has run yet");
  DtxDeviceGuard g(ctx->device);
  DTX_HIP(ctx, hipEventSynchronize(ctx->ev_k1));
  DTX_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_k0, ctx->ev_k1));
  return DTX_OK;
}

}  // extern "C"
