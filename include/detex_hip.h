/* detex_hip.h -- C ABI of libdetexhip.so: the MI355X (gfx950) hot path of Detex.
 *
 * The reference (d-chambers/Detex v1.0.9) is pure Python and has no FFI; the
 * seam this library replaces is the set of plain-Python calls between its
 * orchestration layer and its NumPy/SciPy numerics:
 *
 *   detex/construct.py:148   _makeDFcclags(eventList, row)       -> dtx_xcorr_allpairs
 *   detex/construct.py:389   _CCX2(fd1, fd2, td1, td2, Nc1, Nc2) -> (one pair of the above)
 *   detex/detect.py:259-268  per-detector loop calling _MPXDS    -> dtx_subspace_ds
 *   detex/detect.py:559-578  _SSDetex._MPXDS(MPcon, reqlen, ssTD, ssFD, Nc, MPconFD)
 *   detex/fas.py:120-134     _MPXSSCorr(MPcon, reqlen, ssTD, ssFD, Nc)  (same statistic)
 *   detex/detect.py:319-388  _loadMPSubSpace (stack basis vectors)      -> dtx_basis_upload
 *   detex/detect.py:178-181  np.histogram(DS, linspace(0,1,401))  -> dtx_ds_post (hist)
 *   detex/detect.py:275-281  MaxDS / inf scrub                    -> dtx_ds_post (maxds)
 *   detex/detect.py:410      candidates for the trigger loop      -> dtx_ds_post (cand)
 *   detex/detect.py:447-499  _estMag at every trigger (+ 501-524 STA/LTA there) -> dtx_trigger_stats
 *   detex/construct.py:469-483 fast_normcorr(t, s)               -> dtx_normcorr
 *   detex/construct.py:1005-1022, 1033-1071 _applyFilter merge / longest gap-free run -> dtx_ingest_counts
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ / torch types.
 *   - every function returns 0 on success, <0 on error (DTX_E_*); the message
 *     is available from dtx_last_error(ctx) (ctx may be NULL for creation
 *     errors).  No exceptions cross the ABI, no global mutable state.
 *   - one context = one HIP device + one stream.  All "_dev" pointers are
 *     device pointers valid on that device; work is enqueued on the context's
 *     stream and is asynchronous unless stated; dtx_ctx_sync() waits for it.
 *   - float data is IEEE float32; the detection statistic and CC coefficients
 *     match the reference's float64 NumPy path to 1e-5 absolute.
 *   - "multiplexed" = channels interleaved sample-wise exactly as
 *     detex/construct.py:928-987 (multiplex): x[Nc*t + c].
 */
#ifndef DETEX_HIP_H
#define DETEX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTX_OK 0
#define DTX_E_ARG (-1)     /* bad argument */
#define DTX_E_HIP (-2)     /* a HIP runtime call failed */
#define DTX_E_STATE (-3)   /* call sequence error (e.g. no basis uploaded) */
#define DTX_E_NOMEM (-4)

#define DTX_HIST_BINS 400  /* detect.py:80: np.linspace(0, 1, num=401) */

typedef struct dtx_ctx dtx_ctx;

/* Library ABI version (bumped on any signature change). */
int dtx_version(void);

/* Create a context on HIP device `device`.  With own_stream == 0 the context
 * enqueues on `stream`, an existing hipStream_t (e.g. torch's current stream);
 * NULL is HIP's default (null) stream.  With own_stream != 0 `stream` is
 * ignored and the context creates and owns a non-blocking stream. */
int dtx_ctx_create(int device, void* stream, int own_stream, dtx_ctx** out);
int dtx_ctx_destroy(dtx_ctx* ctx);
/* Block until everything enqueued on the context's stream has finished. */
int dtx_ctx_sync(dtx_ctx* ctx);
/* The hipStream_t the context enqueues on. */
void* dtx_ctx_stream(dtx_ctx* ctx);
const char* dtx_last_error(dtx_ctx* ctx);

/* ------------------------------------------------------------------------
 * Subspace detection statistic            (detect.py:559-578, fas.py:120-134)
 * ------------------------------------------------------------------------ */

/* Stack the basis vectors of all detectors acting on one station
 * (detect.py:319-388 builds ssTD[name] = U per detector; here they are one
 * [Ktot][n] row-major matrix, detector d owning rows det_off[d]..det_off[d+1]).
 * U may be a host or a device pointer; det_off is a host array of D+1 ints.
 * n must be a multiple of Nc (templates are multiplexed streams) and equal for
 * all detectors of the station.  Synchronous. */
int dtx_basis_upload(dtx_ctx* ctx, const float* U, const int32_t* det_off,
                     int D, int n, int Nc);

/* Multi-device hosts (one process driving several GPUs): number of HIP devices, and the
 * basis of ctxs[0] replicated into ctxs[1 .. nctx-1] -- the contexts may sit on different devices: the
 * stacked basis travels device to device (peer copies over xGMI where the devices are linked: issued to all
 * receivers' streams before anything else, so that they run together -- xGMI is point to point), every
 * image (row sums, MFMA tiles, pair spectra) is then built on the receiving device as in
 * dtx_basis_upload, each device on its own stream, and the call waits for all of them once at the end.
 * Continuous-data days shard embarrassingly: after this call each context scans its own share with
 * dtx_subspace_scan.  THE PRODUCT'S BROADCAST IS NOT THIS CALL: one process per GPU (torchrun) broadcasts
 * the bases with RCCL through the process group (detex_amd.parallel.broadcast_banks -> torch.distributed
 * backend 'nccl' over xGMI) and every rank calls dtx_basis_upload on its own device; this entry point serves
 * a host that drives several GPUs from ONE process through the C ABI.  Synchronous.
 * dtx_device_pci_bus_id: "domain:bus:device.function" of a HIP device (hipDeviceGetPCIBusId), the key under
 * /sys/bus/pci/devices/ for its NUMA node and local CPUs (detex_amd.parallel.bind_to_device). */
int dtx_device_count(void);
int dtx_basis_bcast(dtx_ctx* const* ctxs, int nctx);
int dtx_device_pci_bus_id(int device, char* buf, int len);

/* Arithmetic engine of dtx_subspace_ds (default DTX_ENGINE_AUTO):
 *   DTX_ENGINE_AUTO    DTX_ENGINE_FFT for templates of 256 .. 32768 samples per channel
 *                      (where it is faster: 8x at 3000 samples, 200 basis vectors),
 *                      DTX_ENGINE_BF16X3 otherwise.
 *   DTX_ENGINE_FP32    v_mfma_f32_16x16x4_f32: exact float32 products, bitwise an
 *                      fma chain.
 *   DTX_ENGINE_BF16X3  every float32 operand split into two bfloat16 parts,
 *                      a*b ~= ahi*bhi + ahi*blo + alo*bhi on v_mfma_f32_16x16x32_bf16
 *                      with float32 accumulation: float32-level accuracy (the
 *                      dropped alo*blo term is 2^-18 relative; DS agrees with the
 *                      float64 reference as closely as the FP32 engine does: both
 *                      are limited by float32 accumulation, ~2e-6 at DS = 1, 1e-8
 *                      on noise), 3.6x faster.  Templates shorter than 64 samples
 *                      per channel are always computed by the FP32 engine.
 *   DTX_ENGINE_FFT     frequency-domain (overlap-save) evaluation of the same statistic,
 *                      the reference's own route (detect.py:559-578: ifft(ssFD * xFD)):
 *                      8192-point float32 transforms in LDS, two basis vectors per
 *                      complex transform, ~150x fewer flops than the contraction.
 *                      Float32 error grows with log(8192) instead of sqrt(n): ~2e-7 at
 *                      DS = 1.  Templates longer than 5633 samples per channel are cut
 *                      into segments of 4096 that reuse the block spectra; longer than
 *                      32768 fall back to DTX_ENGINE_BF16X3. */
#define DTX_ENGINE_FP32 0
#define DTX_ENGINE_BF16X3 1
#define DTX_ENGINE_FFT 2
#define DTX_ENGINE_AUTO 3
int dtx_set_engine(dtx_ctx* ctx, int engine);

/* Number of detection windows for a data block of M multiplexed samples:
 * len(_MPXDS(...)) = (M - n)//Nc + 1 (detect.py:577), or 0 if M < n. */
int64_t dtx_num_windows(int64_t M, int n, int Nc);

/* Detection statistic of every uploaded detector over `nchunk` data blocks.
 *   x_dev    multiplexed data, block b starts at x_dev + b*chunk_stride and
 *            holds M samples (blocks may overlap: chunk_stride < M is fine,
 *            cf. the 3600 s + 120 s chunks of getdata.py:509-518)
 *   ds_dev   out, [nchunk][D][ldp] float32, ldp >= P = dtx_num_windows(M,n,Nc):
 *            ds[b][d][p] = sum_k (U_k.w - S_k mu)^2 / (n/(n-1) sum (w-mu)^2),
 *            w = x_b[p*Nc : p*Nc+n]            
 * Deliberate, documented deviations from the reference's round-off artefacts:
 * a window of exactly zero variance scores 0 (the reference divides round-off
 * by ~0), and +inf is stored as 0 (the reference zeroes infs whenever one is
 * present, detect.py:278-281).
 * Dirty records (zero-filled gaps, glitches of any size, loud onsets) are held to the same 1e-5: under the
 * frequency-domain engine a guard lists the blocks whose dynamic range float32 transforms cannot carry, isolated
 * outlier samples are taken out of the transforms and their products added back directly, and the windows still out
 * of range are recomputed by the float32 contraction (dtx_repair_counts says what was listed).  The reference's
 * float64 transforms need none of this.  Data are expected detrended / band-passed as the reference's pipeline
 * leaves them (construct.py:1005-1028): a record riding a level >~ 1000 x its scatter behind a step inside a block
 * is held to ~1e-8 x level / scatter only.  Asynchronous. */
int dtx_subspace_ds(dtx_ctx* ctx, const float* x_dev, int64_t M,
                    int64_t chunk_stride, int nchunk, float* ds_dev,
                    int64_t ldp);

/* Post-processing of a DS tensor produced by dtx_subspace_ds (any of the
 * outputs may be NULL to skip it).
 *   hist_dev   [D][DTX_HIST_BINS] uint32, ACCUMULATED over the nchunk blocks
 *              with np.histogram(DS, np.linspace(0,1,401)) semantics
 *              (detect.py:178-181); caller zeroes it before the first call
 *   maxds_dev  [nchunk][D] float32 = DS.max() (detect.py:275); NaN propagates
 *   thr_dev    [D] float32 thresholds; entries with ds >= thr (detect.py:410)
 *              are appended to the candidate list
 *   cand_key_dev / cand_val_dev  capacity `cand_cap` entries; key packs
 *              chunk << 48 | det << 32 | window index -- see dtx_cand_unpack
 *   ncand_dev  uint32 counter (caller zeroes it); may exceed cand_cap, in which
 *              case only cand_cap entries were stored (order unspecified)
 * Asynchronous. */
int dtx_ds_post(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp,
                int nchunk, int D, uint32_t* hist_dev, float* maxds_dev,
                const float* thr_dev, uint64_t* cand_key_dev,
                float* cand_val_dev, uint32_t cand_cap, uint32_t* ncand_dev);

/* A stream for the post pass (histogram / maxima / candidates) of dtx_subspace_scan(_ragged); NULL (default): the
 * context's own.  With one, the pass runs there behind the scan, BESIDE whatever the caller queues next on the
 * context's stream -- in a pipelined loop the next scan's preparation -- and the library orders the next scan's
 * kernel (which overwrites DS) behind it.  Its results (hist / maxds / candidates) are then complete on that stream,
 * not on the context's: wait with dtx_ctx_sync, make a stream wait with dtx_stream_wait_post, or read them through
 * dtx_export_results (which follows the pass onto its stream).  DS itself is complete on the context's stream as
 * before.  The result buffers of a call (hist / maxds / candidates) must not be handed to the next call before they
 * were read.  No reference counterpart (a host loop: detect.py:259-281). */
int dtx_ctx_set_post_stream(dtx_ctx* ctx, void* stream);
int dtx_stream_wait_post(dtx_ctx* ctx, void* stream);

/* The small results of a scan handed to the host by a kernel on the context's stream (no copy engine, no second
 * stream): words_dev = a caller's packed 4-byte results [head_words | candidate values ...] whose word `cnt_word`
 * (< head_words) is the candidate counter, keys_dev = the candidate keys; words_host / keys_host = page-locked host
 * memory (hipHostMalloc: device-accessible).  Copies words[0 .. head_words + min(count, cand_max)) and
 * keys[0 .. min(count, cand_max)); all four pointers 16-byte aligned.  The host reads them after an event recorded
 * behind this call.  No reference counterpart (its arrays never leave the host).  Asynchronous. */
int dtx_export_results(dtx_ctx* ctx, const uint32_t* words_dev, uint32_t* words_host, int64_t head_words,
                       int64_t cand_max, int64_t cnt_word, const uint64_t* keys_dev, uint64_t* keys_host);

/* dtx_subspace_ds followed by dtx_ds_post in one call (same arguments, same results).  The
 * reference does exactly this per chunk (detect.py:259-281, 178-181, 410): detection statistic,
 * then its histogram, its maximum and the samples at or above the threshold.
 * maxds_dev is zeroed here; hist_dev / ncand_dev are the caller's
 * to zero, as for dtx_ds_post.  Asynchronous. */
int dtx_subspace_scan(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                      int nchunk, float* ds_dev, int64_t ldp, uint32_t* hist_dev,
                      float* maxds_dev, const float* thr_dev, uint64_t* cand_key_dev,
                      float* cand_val_dev, uint32_t cand_cap, uint32_t* ncand_dev);

/* The same for a RAGGED batch: chunks of different lengths at arbitrary positions in one launch.
 * The reference cuts a station's record at multiples of conDatDuration (detex/getdata.py:869-886
 * _divideIntoChunks, 494-524 getConData: the first and the last chunk of a request hold less data
 * than the others), and _applyFilter trims chunks to their data (construct.py:1003-1023);
 * equal-length batches then cost one set of launches per distinct length.
 *   chunk_tab_dev  [nchunk][2] int64 on the device: {offset of chunk b's first sample from x_dev,
 *                  its samples M_b}; M_b is clamped to [0, M]
 *   M              the longest chunk of the batch: sizes the workspaces and the DS rows exactly as a
 *                  uniform batch of M-sample chunks does (ds_dev [nchunk][D][ldp], ldp >= P(M))
 * Chunk b has P_b = (M_b - n) / Nc + 1 windows (none if M_b <= n): DS[b][d][p] for p < P_b is what
 * dtx_subspace_scan returns for that chunk alone; entries at p >= P_b are unspecified.  Histogram,
 * maximum and candidates cover p < P_b only.  Asynchronous. */
int dtx_subspace_scan_ragged(dtx_ctx* ctx, const float* x_dev, const int64_t* chunk_tab_dev,
                             int64_t M, int nchunk, float* ds_dev, int64_t ldp, uint32_t* hist_dev,
                             float* maxds_dev, const float* thr_dev, uint64_t* cand_key_dev,
                             float* cand_val_dev, uint32_t cand_cap, uint32_t* ncand_dev);

/* Host-side check of the chunk table of a ragged batch BEFORE it goes to the device (the device passes clamp
 * M_b to [0, M] but trust the offsets): chunk_tab_host [nchunk][2] int64 as for dtx_subspace_scan_ragged,
 * numel = samples in the flat buffer behind x_dev.  Returns 0 when every chunk lies inside the buffer and holds
 * at most M samples, else 1 + the index of the first chunk that does not.  No context, no device work. */
int dtx_chunk_table_check(const int64_t* chunk_tab_host, int nchunk, int64_t M, int64_t numel);

/* False-alarm statistics of a DS tensor (detex/fas.py:72-86: np.histogram(dss,
 * np.linspace(-.01, 1, 401)) and scipy.stats.beta.fit(dss, floc=0, fscale=1)).
 *   hist_dev  [D][nbins] uint32, accumulated, uniform np.linspace(lo, hi, nbins+1)
 *             edges with np.histogram semantics (may be NULL)
 *   sums_dev  [D][5] float64, accumulated: n, sum x, sum x^2, sum log x,
 *             sum log1p(-x) over the samples with 0 < x < 1 -- the sufficient
 *             statistics of the beta MLE (may be NULL).  Float64 atomics: the
 *             last bits depend on arrival order.
 * Caller zeroes both before the first call.  Asynchronous. */
int dtx_ds_moments(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp,
                   int nchunk, int D, double lo, double hi, int nbins,
                   uint32_t* hist_dev, double* sums_dev);

/* ------------------------------------------------------------------------
 * Trigger-time estimates                (detect.py:447-499, 501-524, 637-664)
 * ------------------------------------------------------------------------ */

/* For each of `ntrig` triggers (rows of trig_dev: chunk, detector, window index p, first DS
 * sample of the STA window or -1, first DS sample of the LTA window or -1 -- int32 x 5), in
 * float64, three workgroups per trigger (window statistics, event dot products, noise):
 *   out[0] mean(|DS[sta_lo : sta_lo + sta_len]|)  (NaN if sta_lo < 0)   detect.py:505-511
 *   out[1] mean(|DS[lta_lo : lta_lo + lta_len]|)  (NaN if lta_lo < 0)
 *   out[2], out[3]  mean and std (ddof 0) of ConDat = x_chunk[p*Nc : p*Nc + n]   detect.py:456
 *   out[4] var(U^T (U ConDat)), U = the rows of the trigger's detector (the reference forms
 *          the n x n matrix UtU = U^T U for this, detect.py:367, 460-463)
 *   out[5] median of the rolling std (window n, ddof 1) of the 5 n samples before the
 *          trigger, or of up to 7 n samples from it if there are fewer before   detect.py:466-472
 *   out[6 + e] ConDat . ewf[ewf_off[det] + e]  for the detector's training events: the lag-0
 *          numerator of fast_normcorr (construct.py:469-483, detect.py:484) and d1 of 494
 * x_dev / M / chunk_stride / ds_dev / P / ldp as in dtx_subspace_scan (the same buffers);
 * ewf_dev [Etot][n] float32 training events, ewf_off_dev [D+1] int32 (device), emax = the
 * largest event count of a detector; out_dev [ntrig][DTX_TRIG_HEAD + emax] float64.
 * ewf_dev == NULL: only out[0], out[1] are computed (STA / LTA without magnitude estimates,
 * detect.py:289-295 with estimateMags=False); the other entries keep what the caller put there.
 * Uses the basis of the last dtx_basis_upload.  Asynchronous. */
#define DTX_TRIG_HEAD 6
int dtx_trigger_stats(dtx_ctx* ctx, const float* x_dev, int64_t M, int64_t chunk_stride,
                      const float* ds_dev, int64_t P, int64_t ldp, const int32_t* trig_dev,
                      int ntrig, int sta_len, int lta_len, const float* ewf_dev,
                      const int32_t* ewf_off_dev, int emax, double* out_dev);
/* ... for the triggers of a dtx_subspace_scan_ragged call: chunk_tab_dev and M as there (a trigger's
 * data windows stay inside its own chunk's M_b samples). */
int dtx_trigger_stats_ragged(dtx_ctx* ctx, const float* x_dev, const int64_t* chunk_tab_dev, int64_t M,
                             const float* ds_dev, int64_t P, int64_t ldp, const int32_t* trig_dev,
                             int ntrig, int sta_len, int lta_len, const float* ewf_dev,
                             const int32_t* ewf_off_dev, int emax, double* out_dev);

/* The stream dtx_trigger_stats / dtx_trigger_stats_ragged launch on: NULL (default) = the context's own.
 * The reference estimates magnitudes inside its chunk loop (detect.py:422-429); a host that keeps several
 * scans queued on the context's stream can hand in a second stream so that the trigger-time kernel of one
 * batch runs beside the scan of the next.  ORDERING IS THE CALLER'S: the stream must wait for the scan that
 * produced ds_dev / wrote x_dev (an event) before dtx_trigger_stats is called, and trig_dev / out_dev /
 * ewf_dev must be valid on it.  Consecutive launches on different streams are ordered among themselves by
 * the library (they share a scratch buffer); dtx_basis_upload and dtx_ctx_destroy wait for the last one. */
int dtx_ctx_set_trigger_stream(dtx_ctx* ctx, void* stream);
/* Makes `stream` wait for the END OF THE SCAN KERNEL of the context's last dtx_subspace_ds / dtx_subspace_scan(_ragged)
 * call (not for the post pass behind it): what a host uses to start one batch's trigger-time kernel when the NEXT
 * batch's scan kernel has finished -- beside that batch's post pass and the preparation of the one after, which
 * leave room on the CUs, instead of beside a scan kernel, which does not (its workgroups hold every vector register
 * of a CU: a trigger-time workgroup on a CU keeps the scan off it for its whole run).  No-op before the first scan. */
int dtx_stream_wait_scan_kernel(dtx_ctx* ctx, void* stream);

/* fast_normcorr(t, s) (construct.py:469-483; callers detect.py:484, subspace.py:765): Pearson
 * correlation of the n-sample template t with every length-n window of the m-sample series s
 * (m >= n), out[p], p = 0 .. m - n; float64 accumulation, float32 in / out.  A window of zero
 * variance divides by zero as in the reference (inf / nan).  Asynchronous. */
int dtx_normcorr(dtx_ctx* ctx, const float* t_dev, int n, const float* s_dev, int64_t m,
                 float* out_dev);

/* The screen of fas._checkSTALTA (fas.py:175-205): out[k] = max_i obspy.signal.trigger.classic_sta_lta(a_k, nsta, nlta)[i]
 * for ntrace traces a_k[j] = x[tab[2 k] + j * stride], j < tab[2 k + 1] (device table; stride 1: channel planes,
 * stride Nc: one channel of a multiplexed chunk); maxlen = the longest trace.  float64 window sums, out float64.
 * Asynchronous.  The caller applies the reference's length rule (len <= nlta: rejected) and the limit. */
int dtx_stalta_max(dtx_ctx* ctx, const float* x_dev, const int64_t* tab_dev, int ntrace, int64_t maxlen,
                   int stride, int nsta, int nlta, double* out_dev);

static inline uint64_t dtx_cand_pack(uint32_t chunk, uint32_t det, uint32_t p) {
  return ((uint64_t)chunk << 48) | ((uint64_t)det << 32) | (uint64_t)p;
}
static inline void dtx_cand_unpack(uint64_t key, uint32_t* chunk, uint32_t* det,
                                   uint32_t* p) {
  *chunk = (uint32_t)(key >> 48);
  *det = (uint32_t)((key >> 32) & 0xFFFFu);
  *p = (uint32_t)(key & 0xFFFFFFFFu);
}

/* Host-pointer convenience: one data block in host memory -> DS in host memory
 * (what a single reference _MPXDS loop over detectors returns).  Synchronous;
 * includes the PCIe transfers. */
int dtx_subspace_ds_host(dtx_ctx* ctx, const float* x_host, int64_t M,
                         float* ds_host /* [D][P] */);

/* ------------------------------------------------------------------------
 * All-pairs normalised cross-correlation          (construct.py:369-466)
 * ------------------------------------------------------------------------ */

/* Events: N multiplexed templates of equal length n = Nc*L, row-major
 * td_dev[N][n] (device).  For every pair b < c (reference loop order,
 * construct.py:380-381) computes what _CCX2 returns:
 *   cc      max over the channel-aligned lags in the central half of the lag
 *           range of the Pearson CC of x_b against the zero-padded, shifted x_c
 *   lag     (maxind + 1 + trunc)*Nc - n, in multiplexed samples (int32)
 *   subsamp cosine-fit sub-sample shift (construct.py:397-422)
 * written at the SciPy condensed index of (b, c):
 *   idx = N*b - b*(b+1)/2 + (c - b - 1)
 * Only pairs with pair_begin <= idx < pair_end are computed (multi-GPU
 * sharding); outputs are full-size [N*(N-1)/2] arrays, device pointers.
 * Asynchronous. */
/* (Engine: under DTX_ENGINE_AUTO / DTX_ENGINE_FFT events of 1024 .. 32768 samples per channel use
 * the frequency-domain lag search; DTX_ENGINE_BF16X3 / DTX_ENGINE_FP32 force the MFMA contraction.) */
int dtx_xcorr_allpairs(dtx_ctx* ctx, const float* td_dev, int N, int n, int Nc,
                       float* cc_dev, int32_t* lag_dev, float* subsamp_dev,
                       int64_t pair_begin, int64_t pair_end);

/* Host-pointer convenience for small problems.  Synchronous. */
int dtx_xcorr_allpairs_host(dtx_ctx* ctx, const float* td_host, int N, int n,
                            int Nc, float* cc_host, int32_t* lag_host,
                            float* subsamp_host);

/* ------------------------------------------------------------------------
 * Pre-processing                              (construct.py:990-1030, 928-987)
 * ------------------------------------------------------------------------ */

/* The numerical part of _applyFilter on the device, in float64: per series
 * st.detrend('linear') (= scipy.signal.detrend), st.filter('bandpass', ...)
 * (= sosfilt of the Butterworth band-pass second-order sections, run forward
 * and, if zerophase, again over the reversed result), then multiplexing
 * (construct.py:977-978).  The caller designs the filter (host, e.g.
 * scipy.signal.iirfilter(corners, [lo, hi], btype='band', ftype='butter',
 * output='sos')) and passes the sections as sos[nsec][6] = b0 b1 b2 a0 a1 a2
 * (host pointer); nsec = 0 skips the filter.
 *   raw_dev  [nchunk*Nc] series of T float32 samples, series s at
 *            raw_dev + s*ser_stride; series s is channel s % Nc of chunk s / Nc
 *   mp_dev   out: chunk b multiplexed at mp_dev + b*mp_stride, x[Nc*t + c]
 * Asynchronous. */
int dtx_preprocess(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc,
                   int64_t T, int64_t ser_stride, const double* sos, int nsec,
                   int zerophase, int detrend, float* mp_dev, int64_t mp_stride);
/* The same for chunks that are cut from longer channel planes where they lie -- the reference's chunks
 * overlap (conDatDuration + conBuff, getdata.py:869-886), so a day's [Nc][Tday] record holds them as
 * views, not as a packed array: channel c of chunk b at raw_dev + b*chunk_stride + c*chan_stride
 * (chunk_stride >= 0: overlapping chunks are fine, the input is only read; chan_stride >= T).  Saves the
 * caller a packed copy of every chunk (0.14 ms per station-day of SubSpace.detex with an unfiltered
 * fetcher).  dtx_preprocess is this with chunk_stride = Nc*ser_stride, chan_stride = ser_stride. */
int dtx_preprocess_planes(dtx_ctx* ctx, const float* raw_dev, int nchunk, int Nc, int64_t T,
                          int64_t chunk_stride, int64_t chan_stride, const double* sos, int nsec,
                          int zerophase, int detrend, float* mp_dev, int64_t mp_stride);

/* The front half of _applyFilter (construct.py:1005-1022) for one chunk that arrives as
 * fragments of raw int32 digitiser counts (one ObsPy trace each):
 *   - Stream.merge(fill_value=0.0) (construct.py:1038, 1070): the fragments are converted to
 *     float32 at their place in planes_dev [Nc][T], gaps are zero (fillZeros=True stops here);
 *   - _mergeChannels (1033-1066): run_dev[0], run_dev[1] = first sample and length of the longest
 *     run in which no channel is zero -- the product test of construct.py:1047-1050, so an exact
 *     zero sample counts as a gap, as there; the first run among equals (1062); (-1, 0) if
 *     none.  The caller trims the chunk to it before dtx_decimate / dtx_preprocess.
 * counts_dev: the fragments' samples, concatenated; frag_host: [nfrag][4] int64 host array of
 * (channel, offset into counts_dev, first sample in the chunk, length); fragments must not
 * overlap (ObsPy resolves overlaps by its merge method; not restated).  run_dev may be NULL.
 * Asynchronous apart from staging the descriptors. */
int dtx_ingest_counts(dtx_ctx* ctx, const int32_t* counts_dev, const int64_t* frag_host, int nfrag,
                      int Nc, int64_t T, float* planes_dev, int64_t* run_dev);

/* The _mergeChannels step alone (construct.py:1033-1066) on zero-filled planes that are already
 * on the device, e.g. one chunk's columns of a whole day: channel c at planes_dev + c*plane_stride,
 * T samples each; run_dev[0..1] as for dtx_ingest_counts.  Asynchronous. */
int dtx_gapfree_run(dtx_ctx* ctx, const float* planes_dev, int Nc, int64_t T, int64_t plane_stride,
                    int64_t* run_dev);

/* Trace.decimate(factor) of _applyFilter (construct.py:1016-1017), which runs
 * BEFORE the detrend / band-pass above: ObsPy (third-party, not vendored)
 * low-passes causally with a Chebyshev-II filter (sosfilt, forward only) and
 * keeps every factor-th sample, data[::factor].  The caller designs the
 * low-pass (detex_amd.hip.decimate_sos restates ObsPy's lowpass_cheby_2
 * design) and passes sos[nsec][6] as for dtx_preprocess; float64 arithmetic.
 *   raw_dev  nser series of T float32 samples, series s at raw_dev + s*ser_stride
 *   out_dev  out: series s at out_dev + s*out_stride, ceil(T/factor) samples
 * Asynchronous. */
int dtx_decimate(dtx_ctx* ctx, const float* raw_dev, int nser, int64_t T,
                 int64_t ser_stride, const double* sos, int nsec, int factor,
                 float* out_dev, int64_t out_stride);

/* ------------------------------------------------------------------------
 * Timing helper: HIP events on the context's stream (bench.py roofline).
 * ------------------------------------------------------------------------ */
int dtx_timer_start(dtx_ctx* ctx);
/* Records the stop event, waits for it, returns elapsed milliseconds. */
int dtx_timer_stop(dtx_ctx* ctx, float* ms);
/* Milliseconds spent inside the dominant kernel of the last dtx_subspace_ds /
 * dtx_xcorr_allpairs call (events recorded around that kernel only; valid
 * after dtx_ctx_sync). */
int dtx_last_kernel_ms(dtx_ctx* ctx, float* ms);
/* Mean of that over the last n such calls (1 <= n <= 64; waits for the newest): a caller that runs several
 * calls ahead of the device reads the durations afterwards instead of synchronising after each. */
int dtx_kernel_ms_mean(dtx_ctx* ctx, int n, float* ms);
/* The same n durations one by one (ms[0 .. n), in the order dtxplan::ring_last_slots lists them: newest first):
 * bench.py's per-step spread (kernel_ms_min / median / max) and its settled-clock pre-warm criterion. */
int dtx_kernel_ms_list(dtx_ctx* ctx, int n, float* ms);
/* What the dynamic-range guards of the last dtx_subspace_ds / dtx_subspace_scan(_ragged) call listed (waits for
 * the stream): counts[0] = blocks listed, counts[1] = column tiles of 128 windows handed to the float32
 * contraction again, counts[2] = blocks scanned again with outlier samples split off, counts[3] = 0.  All zero on
 * ordinary data; the reference has no counterpart (its float64 transforms need no guard: detect.py:559-578). */
int dtx_repair_counts(dtx_ctx* ctx, int32_t counts[4]);
/* Name of the dominant kernel the context's last scan / cross-correlation call dispatched (the instantiation the
 * library chose: e.g. "ds_fft_pp_kernel<10,false,3>", "xc_fft22_kernel<9>"); bench.py labels its roofline with it. */
const char* dtx_last_kernel_name(dtx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DETEX_HIP_H */
