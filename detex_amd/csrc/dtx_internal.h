// Internal definitions shared by the translation units of libdetexhip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/detex_hip.h"

#define DTX_ABI_VERSION 1

// One K-tile of the stacked basis: up to KT_MAX rows (whole detectors only),
// stored transposed and channel-planar for the MFMA kernel (see dtx_subspace.hip).
struct DtxKTile {
  int rm = 0;             // row blocks of 16 (KT = 16*rm rows incl. zero padding)
  int rows = 0;           // real rows
  int det0 = 0;           // first global detector in this tile
  int nd = 0;             // detectors in this tile
  float* ut_dev = nullptr;      // [Nc*Lp][KT] transposed, channel-planar basis
  uint16_t* ut16_dev = nullptr; // split-bf16 image [Nc][Lp/8][2][KT][8]
  float* srow_dev = nullptr;    // [KT] row sums S_k
  int32_t* doff_dev = nullptr;  // [nd+1] detector row offsets local to the tile
  // repair sub-tiles (dtxplan::plan_repair_subtiles): runs of whole detectors inside `rms` row blocks of this tile's
  // images -- the row groups of the repair pass's tasks (ds_mfma_task_kernel)
  int rms = 0, nsub = 0;
  void* sub_dev = nullptr;      // [nsub] int4 {first row block, first detector (tile-local), detectors, offset into sdoff}
  int32_t* sdoff_dev = nullptr; // the sub-tiles' detector row offsets, relative to their first row block
};

// Frequency-domain engine (dtx_fft.hip): pair spectra of the basis, built on first use
#ifndef DTX_FFT_N
#define DTX_FFT_N 8192
#endif
struct DtxFft {
  bool ok = false;
  bool two = false;            // some unit pairs two single-vector detectors
  bool pingpong = false;       // every unit splits between two half-workgroups: ds_fft_pp_kernel
  int V = 0;                   // windows per block (block advance), a multiple of 512
  int nseg = 1;                // template segments (> 1: templates longer than kFftMaxL)
  int nunits = 0, npairs = 0;
  void* W_dev = nullptr;       // [npairs][Nc][DTX_FFT_N] complex64
  void* tab_dev = nullptr;     // [DTX_FFT_N] twiddle table
  void* unit_dev = nullptr;    // [nunits] int4
  void* pinfo_dev = nullptr;   // [npairs] float4
  void* pdet_dev = nullptr;    // [npairs] int2
  void* prow_dev = nullptr;    // [npairs] int2: basis rows of the real / imaginary part
  void* upre_dev = nullptr;    // [nunits + 1] int2: pairs of half 0 / half 1 before each unit (ping-pong kernel)
  void* runs_dev = nullptr;    // [nruns] int2 {first pair, pairs}: runs of whole detectors (the re-scan's work items)
  int nruns = 0;
  std::vector<int> unit_pairs; // pairs per unit (host copy: dtx_fft_scan groups units per launch)
};
// templates (samples per channel) the frequency-domain engine takes; others fall back
constexpr int kFftMinL = 2;
// measured (profiles/attic/seg_ab.sh, 200 basis vectors): two segments of 4096 take 12.8 ms per
// station-day; one segment takes 11.3 ms at V = 2560 and 13.6 ms at V = 2048
#ifndef DTX_FFT_SEG_L
#define DTX_FFT_SEG_L (DTX_FFT_N - 2560 + 1)
#endif
constexpr int kFftMaxL = DTX_FFT_SEG_L;            // longest single-segment template
constexpr int kFftLongMaxL = 8 * (DTX_FFT_N / 2);  // longest template at all (8 segments of 4096)
// DTX_ENGINE_AUTO: shortest template it hands to the frequency-domain engine (measured
// crossover with the split-bf16 MFMA engine at 200 basis vectors, Nc = 3: L ~ 220;
// profiles/f3_crossover.txt)
constexpr int kFftAutoMinL = 256;

struct DtxBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct dtx_ctx {
  int device = 0;
  int ncu = 256;                // compute units (grid shaping)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // side stream of the scan's preparation (window statistics run beside the block spectra) and the
  // events that fork it from / join it to `stream`; created on first use
  hipStream_t aux = nullptr;
  bool aux_is_main = false;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // stream of the trigger-time kernel (dtx_trigger_stats): NULL = `stream`.  A caller that keeps several scans
  // queued hands in a stream of its own (dtx_ctx_set_trigger_stream): the kernel -- a few dozen workgroups of
  // dependent passes, 0.2 .. 0.4 ms -- then runs beside the next scan instead of in front of it
  hipStream_t trig_stream = nullptr;
  // stream of a scan's post pass (histogram / maxima / candidates): NULL = `stream`.  A caller that keeps several scans
  // queued hands in a stream of its own (dtx_ctx_set_post_stream): the pass -- a read of the whole DS tensor at the
  // memory's rate -- then runs beside the NEXT scan's preparation (offsets, channel planes, block spectra, window
  // statistics: chains of latencies that leave the memory system idle) instead of in front of it
  hipStream_t post_stream = nullptr;
  hipEvent_t ev_post_ready = nullptr, ev_post_done = nullptr;
  bool post_pending = false;           // a post pass on post_stream that nothing on `stream` has waited for yet
  bool post_on_side = false;           // the last scan's post pass ran on post_stream (dtx_export_results follows it there)
  hipStream_t trig_last = nullptr;     // stream of the last trigger-time launch ...
  hipEvent_t ev_trig = nullptr;        // ... and the event recorded behind it (orders a launch on another stream, basis uploads, destroy)
  bool trig_pending = false;
  std::string err;

  // basis
  int D = 0, n = 0, Nc = 0, L = 0, Lp = 0, Ktot = 0;
  std::vector<DtxKTile> tiles;
  DtxBuf U_buf;                 // device copy of the stacked basis [Ktot][n] (images are built from it on the device)
  DtxBuf srow_all;              // [Ktot] row sums S_k (float64 sums, stored float32)
  DtxBuf doff_all;              // [D+1] int32 copy of hdoff on the device
  std::vector<int32_t> hdoff;   // [D+1]
  DtxFft fft;

  // workspaces (grown on demand, never shrunk)
  DtxBuf ws_xp, ws_mean, ws_den, ws_misc, ws_misc2, ws_misc3, ws_g, ws_csum, ws_xhat, ws_flag, ws_trig, ws_trig_doff;
  // pre-processing: the filter's block-transition matrices on the device, kept for the filter / geometry they
  // were built for (pp_key) -- rebuilding them per call cost a stream synchronisation per call
  static constexpr int kPpCache = 8;   // (a station's chunk groups differ in length: one entry per geometry)
  DtxBuf pp_M[kPpCache];
  std::vector<double> pp_key[kPpCache];
  int pp_next = 0;                     // round-robin replacement
  int engine = DTX_ENGINE_AUTO;
  std::string last_kernel;             // the dominant kernel the last call dispatched (dtx_last_kernel_name)
  bool guard_ran = false;              // the last scan filled DtxRepairWs::count (dtx_repair_counts)

  // small host tables on their way to the device without a stream synchronisation: a ring of page-locked
  // staging buffers, each guarded by the event recorded behind its last copy (dtx_stage_h2d)
  static constexpr int kStage = 8;
  void* stage_buf[kStage] = {};
  size_t stage_cap[kStage] = {};
  hipEvent_t stage_ev[kStage] = {};
  int stage_next = 0;

  // timing
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;   // user timer
  // dominant-kernel timer: a ring of event pairs, one per timed launch (ev_k0 / ev_k1 = the current pair), so
  // that a caller can run several calls ahead and read the durations afterwards (dtx_kernel_ms_mean)
  static constexpr int kKev = 64;
  hipEvent_t kev0[kKev] = {}, kev1[kKev] = {};
  long long kev_count = 0;                        // timed launches so far
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
  bool kernel_timed = false;
};

// ---- repair of what the dynamic-range guards list (dtx_subspace.hip, dtx_fft.hip)
// An outlier sample the frequency-domain engine takes out of a listed block's transforms and adds back directly
// (spike_split_kernel): frame t of its chunk's channel plane c, value v (block offset already removed).
struct DtxSpike { int t, c; float v; };
constexpr int kMaxSpk = 48;            // outliers per block; more: the block keeps its transforms and goes to the contraction
// windows per repair task (column tile of the float32 contraction).  A task is a chain of 3 L / 64 steps whose
// loads nothing hides on an otherwise idle CU: 128 against 256 windows, profiles/r6_dirty_cost.txt (c): config2_dirty
// 3.53 / 3.57 ms per step, a loud onset (39 / 20 tiles) 3.47 / 3.60, one tile 3.44 / 3.46 (clean: 3.21 - 3.31)
constexpr int kRepairPT = 128;
// One workspace (ctx->ws_flag) for a call's guard lists; count[] is cleared by chunk_offset_final_kernel.
struct DtxRepairWs {
  int* count;        // [0] listed blocks, [1] repair tasks, [2] blocks re-scanned with outliers split off, [3] -
  int4* tasks;       // [ntile] {chunk, first window / kRepairPT, bits of the window energy below which a window is stored, block}
  int* list;         // [nb] listed blocks (chunk * nblk + blk), in the order found
  int* slist;        // [nb] blocks with outliers split off
  float* gmn;        // [nb] by block: quietest live own window energy / loudest window energy (guard_kernel)
  float* gmx;
  int* nspk;         // [nb] by block: outliers split off
  DtxSpike* spk;     // [nb][kMaxSpk]
};
inline size_t dtx_repair_ws_bytes(size_t nb, size_t ntile) {
  return 16 + ntile * 16 + nb * (5 * 4 + kMaxSpk * sizeof(DtxSpike));
}
inline DtxRepairWs dtx_repair_ws_carve(void* base, size_t nb, size_t ntile) {
  DtxRepairWs w;
  char* p = (char*)base;
  w.count = (int*)p; p += 16;
  w.tasks = (int4*)p; p += ntile * 16;
  w.list = (int*)p; p += nb * 4;
  w.slist = (int*)p; p += nb * 4;
  w.gmn = (float*)p; p += nb * 4;
  w.gmx = (float*)p; p += nb * 4;
  w.nspk = (int*)p; p += nb * 4;
  w.spk = (DtxSpike*)p;
  return w;
}
// the listed tasks again on the float32 contraction (dtx_subspace.hip); every launch finds an empty list on ordinary data
int dtx_launch_repair(dtx_ctx* ctx, const DtxRepairWs& ws, const float* xp, int64_t Tpad, const float* mean,
                      const float* den, int64_t Ppad, int64_t P, float* ds, int64_t ldp);
int dtx_fail(dtx_ctx* ctx, int code, const char* fmt, ...);
int dtx_reserve(dtx_ctx* ctx, DtxBuf& b, size_t bytes);
void dtx_free_tiles_internal(dtx_ctx* c);
// shortest template (samples per channel) the split-bf16 engine is used for
constexpr int kSplitMinL = 64;
struct DtxPost;
void dtx_fft_free(dtx_ctx* ctx);
bool dtx_fft_eligible(int L);
int dtx_fft_basis_build(dtx_ctx* ctx, const int32_t* det_off, int D, int n, int Nc);
// xp: the channel planes; mean / den: the window statistics, being computed on the context's side
// stream (forked by the caller): the guard pass follows them there, the block spectra are taken on the
// main stream meanwhile, and the two streams meet in front of the scan kernel
// pb != NULL (ragged batch): blocks wholly past their chunk's own pb[chunk] windows are skipped
int dtx_fft_scan(dtx_ctx* ctx, const float* xp, int64_t Tpad, const float* mean, const float* den,
                 int64_t Ppad, int64_t P, int nchunk, float* ds, int64_t ldp, const DtxRepairWs& ws,
                 const int* pb = nullptr);
// dst_dev <- bytes of src_host, asynchronously on the context's stream; src_host may be reused at once
int dtx_stage_h2d(dtx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int dtx_kev_next(dtx_ctx* ctx);      // ev_k0 / ev_k1 := the next pair of the ring
void dtx_kev_commit(dtx_ctx* ctx);   // ... both recorded: the pair counts (dtx_kernel_ms_mean)
int dtx_aux_fork(dtx_ctx* ctx);      // aux waits for what is queued on stream
int dtx_aux_join(dtx_ctx* ctx);      // stream waits for what is queued on aux
// a block is handed to the float32 MFMA engine when max / min window energy exceeds this squared.
// Calibration (tests/test_gpu_engines.py, profiles/attic/sweep_more.py): the error of a quiet window
// that shares a block with a loud one is ~1.8e-8 * sqrt(DS) * sqrt(energy ratio); 200 keeps it
// under 4e-6 even for detectors that span half the window's dimensions (DS ~ 0.5 on noise)
constexpr float kFftGuard = 200.f;
// ... and the repair leaves alone the windows that hold at least 1 / kFftKeep of the loudest window's energy
// (1.8e-8 * sqrt(256) = 2.9e-7: the frequency-domain engine's value is the better one there)
constexpr float kFftKeep = 256.f;
// split-bf16 engine: windows per guard block, and the share of a window's energy one sample (times the basis'
// largest column energy) may hold before the block goes to the float32 engine (dtx_subspace.hip)
constexpr int kSplitGuardV = 1024;
constexpr float kSplitGuardTau = 0.1f;
// pb != NULL (ragged batch): chunk b's own window count; the pass covers p < min(P, pb[b])
int dtx_post_launch(dtx_ctx* ctx, const float* ds_dev, int64_t P, int64_t ldp, int nchunk, int D,
                    const DtxPost& po, const int* pb = nullptr, hipStream_t st = nullptr);
// `stream` waits for the post pass still running on post_stream (before anything overwrites the DS tensor it reads)
int dtx_wait_post(dtx_ctx* ctx);
// shortest events (samples per channel) the all-pairs cross-correlation sends through the
// frequency domain under DTX_ENGINE_AUTO / DTX_ENGINE_FFT (dtx_xcfft.hip)
constexpr int kXcFftMinL = 1024;
int dtx_xcfft_run(dtx_ctx* ctx, const float* td_dev, int N, int n, int Nc, int L, int Pz,
                  int64_t Ppad, int64_t Tz, const float* z, const float* fsum, const float* frs,
                  const float* mean_w, const float* rsw, unsigned long long* keys,
                  int64_t pair_begin, int64_t pair_end, int64_t b_begin, int64_t b_end);
int dtx_launch_demux16(dtx_ctx* ctx, const float* xp, int64_t Tpad, int nplanes, void* g16);

#define DTX_HIP(ctx, call)                                                    \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
      return dtx_fail((ctx), DTX_E_HIP, "%s failed: %s (%s:%d)", #call,       \
                      hipGetErrorString(e_), __FILE__, __LINE__);             \
  } while (0)

struct DtxDeviceGuard {
  int prev = -1;
  explicit DtxDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DtxDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static inline int64_t dtx_round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
