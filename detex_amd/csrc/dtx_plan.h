// Host-side planning of libdetexhip.so: pure arithmetic on sizes and index tables, no HIP.
//
// Everything a launch path decides on the host before it touches the device lives here -- the pairs /
// units / half split of the frequency-domain scan (dtx_fft_basis_build), the group split of its left-over
// blocks (dtx_fft_scan), the segment / lag-block choice and the work list of the all-pairs lag search
// (dtx_xcfft_run), the staging and timing rings of the context -- so that the same code compiles in a
// plain C++17 translation unit under -fsanitize=address,undefined (tests/csrc/plan_check.cpp, run by
// tests/test_plan_sanitizer.py on the CPU box; GPU sanitizers are not available on this pool).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace dtxplan {

// layout-compatible with HIP's int2 / int4 (the tables are copied to the device as they are)
struct I2 { int x, y; };
struct I4 { int x, y, z, w; };
static_assert(sizeof(I2) == 8 && sizeof(I4) == 16, "table rows are packed ints");

// ------------------------------------------------------------------------------------------------
// Frequency-domain scan: pairs, units, half split  (dtx_fft.hip, dtx_fft_basis_build)
// ------------------------------------------------------------------------------------------------
struct FftBasisPlan {
  int nseg = 1;               // template segments (> 1: templates longer than max_single_L)
  int V = 0;                  // windows per block (block advance), a multiple of 512
  bool two = false;           // some pair holds two single-vector detectors
  int nunits = 0, npairs = 0;
  int cum0 = 0, cum1 = 0;     // pairs of half 0 / half 1 of the ping-pong kernel
  bool balanced = false;      // the halves are balanced well enough for the ping-pong kernel
  std::vector<I4> units;      // {first pair, pairs, pairs of half 0, 0}: runs of whole detectors
  std::vector<I2> pdet;       // per pair: {detector closed by this pair or -1, second detector or -1}
  std::vector<I2> prow;       // per pair: {basis row of the real part, of the imaginary part or -1}
  std::vector<I2> upre;       // [nunits + 1]: pairs of half 0 / half 1 before each unit
  std::vector<int> unit_pairs;
};

// det_off[D + 1]: detector d owns basis rows det_off[d] .. det_off[d + 1] - 1 (at least one each).
// L: samples per channel; N: transform length; max_single_L: longest single-segment template;
// unit_cap: pairs a unit is closed at.
// Pairs: consecutive vectors of a detector (an odd one out pairs with a zero row); two consecutive
// single-vector detectors share a pair.  Units: runs of whole detectors' pairs, closed once they hold
// unit_cap (a detector with more pairs is a unit of its own).  Pairs [0, z) of a unit go to half 0 of
// the ping-pong kernel: the detector boundary that leaves the two halves' running totals closest.
inline FftBasisPlan plan_fft_basis(const int32_t* det_off, int D, int L, int N, int max_single_L, int unit_cap) {
  FftBasisPlan f;
  const int seg = N / 2;
  f.nseg = L <= max_single_L ? 1 : (L + seg - 1) / seg;
  f.V = f.nseg == 1 ? (N - L + 1) / 512 * 512 : seg;
  int ufirst = 0;
  auto close_unit = [&]() {
    const int np = (int)f.prow.size() - ufirst;
    if (np > 0) {
      int best = np, bestd = 1 << 30;
      for (int q = 0; q <= np; ++q) {
        if (q > 0 && f.pdet[ufirst + q - 1].x < 0) continue;        // inside a detector
        const int dd = std::abs((f.cum0 + q) - (f.cum1 + np - q));
        if (dd < bestd) { best = q; bestd = dd; }
      }
      f.cum0 += best; f.cum1 += np - best;
      f.units.push_back(I4{ufirst, np, best, 0});
    }
    ufirst = (int)f.prow.size();
  };
  for (int d = 0; d < D;) {
    const int r0 = det_off[d], k = det_off[d + 1] - r0;
    const bool shared = k == 1 && d + 1 < D && det_off[d + 2] - det_off[d + 1] == 1;
    const int np = shared ? 1 : (k + 1) / 2;
    if ((int)f.prow.size() - ufirst + np > unit_cap) close_unit();
    if (shared) {
      f.two = true;
      f.pdet.push_back(I2{d, d + 1});
      f.prow.push_back(I2{r0, r0 + 1});
      d += 2;
    } else {
      for (int q = 0; q < k; q += 2) {
        const bool has2 = q + 1 < k;
        f.pdet.push_back(I2{q + 2 >= k ? d : -1, -1});
        f.prow.push_back(I2{r0 + q, has2 ? r0 + q + 1 : -1});
      }
      d += 1;
    }
  }
  close_unit();
  f.nunits = (int)f.units.size();
  f.npairs = (int)f.prow.size();
  f.upre.assign(f.units.size() + 1, I2{0, 0});
  for (size_t u = 0; u < f.units.size(); ++u)
    f.upre[u + 1] = I2{f.upre[u].x + f.units[u].z, f.upre[u].y + f.units[u].y - f.units[u].z};
  // the ping-pong kernel is worth it when the halves are balanced: at least 0.8 pairs per round
  const int pp_rounds = std::max(2 * f.cum0 + 1, 2 * f.cum1 + 2);
  f.balanced = 5 * f.npairs >= 4 * pp_rounds;
  f.unit_pairs.resize(f.units.size());
  for (size_t u = 0; u < f.units.size(); ++u) f.unit_pairs[u] = f.units[u].y;
  return f;
}

// The pair tables in the order the ping-pong kernel walks them: half 0's pairs of every unit first, then
// half 1's -- each half-workgroup walks a contiguous run.
inline void reorder_pairs_for_halves(FftBasisPlan& f) {
  std::vector<I2> pdet2, prow2;
  pdet2.reserve(f.pdet.size());
  prow2.reserve(f.prow.size());
  for (int h = 0; h < 2; ++h)
    for (const I4& un : f.units)
      for (int q = h ? un.z : 0; q < (h ? un.y : un.z); ++q) {
        pdet2.push_back(f.pdet[un.x + q]);
        prow2.push_back(f.prow[un.x + q]);
      }
  f.pdet.swap(pdet2);
  f.prow.swap(prow2);
}

// Runs of pairs that close whole detectors, in the order the pair tables are stored: the work items of the
// re-scan of a block whose outliers were split off (ds_fft_rescan_kernel).  after_reorder: the tables went
// through reorder_pairs_for_halves (a unit's pairs then lie in two runs, one per half).
inline std::vector<I2> plan_fft_runs(const FftBasisPlan& f, bool after_reorder) {
  std::vector<I2> runs;
  if (!after_reorder) {
    for (const I4& un : f.units) runs.push_back(I2{un.x, un.y});
    return runs;
  }
  const int h0 = f.upre.back().x;
  for (size_t u = 0; u + 1 < f.upre.size(); ++u) {
    const int n0 = f.upre[u + 1].x - f.upre[u].x, n1 = f.upre[u + 1].y - f.upre[u].y;
    if (n0 > 0) runs.push_back(I2{f.upre[u].x, n0});
    if (n1 > 0) runs.push_back(I2{h0 + f.upre[u].y, n1});
  }
  return runs;
}

// ------------------------------------------------------------------------------------------------
// Float32 contraction: row groups of the repair pass  (dtx_subspace.hip, dtx_basis_upload)
// ------------------------------------------------------------------------------------------------
// A K-tile holds nd whole detectors in rm row blocks of 16 (doff[nd + 1]: tile-local row offsets).  The repair
// pass recomputes a few windows for ALL rows: its tasks take the rows in sub-tiles of `rms` row blocks, each
// holding whole detectors (the per-detector sums are formed inside a workgroup), so that one listed column tile
// spreads over nsub workgroups.  rms = the most row blocks any single detector touches; a sub-tile starts at the
// row block its first detector starts in (moved down where it would reach past the tile: the window then covers
// rows of earlier detectors, which its detector table ignores).  sdoff: the sub-tiles' detector offsets
// relative to their first row block, concatenated (sub.w = offset of the sub-tile's nd + 1 entries).
struct RepairSubPlan {
  int rms = 1;
  std::vector<I4> sub;        // {first row block, first detector, detectors, offset into sdoff}
  std::vector<int32_t> sdoff;
};
inline RepairSubPlan plan_repair_subtiles(const int32_t* doff, int nd, int rm) {
  RepairSubPlan p;
  for (int d = 0; d < nd; ++d) p.rms = std::max(p.rms, (doff[d + 1] + 15) / 16 - doff[d] / 16);
  p.rms = std::min(p.rms, std::max(rm, 1));
  for (int d = 0; d < nd;) {
    const int rb = doff[d] / 16;
    int e = d + 1;
    while (e < nd && (doff[e + 1] + 15) / 16 - rb <= p.rms) ++e;
    const int rb0 = std::max(0, std::min(rb, rm - p.rms));
    p.sub.push_back(I4{rb0, d, e - d, (int)p.sdoff.size()});
    for (int q = d; q <= e; ++q) p.sdoff.push_back(doff[q] - 16 * rb0);
    d = e;
  }
  return p;
}

// ------------------------------------------------------------------------------------------------
// Frequency-domain scan: grid of the ping-pong kernel  (dtx_fft.hip, dtx_fft_scan)
// ------------------------------------------------------------------------------------------------
constexpr int kPlanMaxGroups = 32;
struct PpGridPlan {
  int nfull = 0;              // blocks run by one workgroup over every unit
  int nrem = 0;               // left-over blocks, each split over ng groups of units
  int ng = 1;
  short gb[kPlanMaxGroups + 1] = {0};   // unit borders of the groups: group g = units gb[g] .. gb[g + 1] - 1
  int64_t grid = 0;
};

// One workgroup per block runs every unit for as many blocks as fill the ncu CUs a whole number of times;
// the remaining nrem < ncu blocks are split over ng groups of units each, ng chosen so that nrem * ng
// short workgroups again fill the CUs evenly.  force > 0 (experiments): that many groups for every block.
inline PpGridPlan plan_pp_grid(int64_t nblocks, int ncu, int nunits, int npairs, const int* unit_pairs, int force) {
  PpGridPlan g;
  ncu = std::max(1, ncu);
  g.nfull = force > 0 ? 0 : (int)(nblocks / ncu * ncu);
  g.nrem = (int)(nblocks - g.nfull);
  double best = 1e300;
  for (int k = 1; k <= std::min(nunits, kPlanMaxGroups) && g.nrem > 0; ++k) {
    if (force > 0 && k != std::min(std::min(force, nunits), kPlanMaxGroups)) continue;
    short b[kPlanMaxGroups + 1];
    int u = 0, acc = 0, maxp = 0, last = 0, n = 0;
    b[0] = 0;
    for (int j = 1; j <= k; ++j) {
      // unit borders nearest to j / k of the pairs
      const int target = (int)((int64_t)npairs * j / k);
      while (u < nunits && (j == k || acc + unit_pairs[u] - target <= target - acc)) acc += unit_pairs[u++];
      if (u > b[n]) { b[++n] = (short)u; maxp = std::max(maxp, acc - last); last = acc; }
    }
    const double cost = (double)(((int64_t)g.nrem * n + ncu - 1) / ncu) * (maxp + 3.0);
    if (cost < best) { best = cost; g.ng = n; std::copy(b, b + n + 1, g.gb); }
  }
  g.grid = (int64_t)g.nfull + (int64_t)g.nrem * g.ng;
  return g;
}

// ------------------------------------------------------------------------------------------------
// All-pairs lag search: segments, lag blocks, work list  (dtx_xcfft.hip, dtx_xcfft_run)
// ------------------------------------------------------------------------------------------------
struct XcSegPlan {
  int nseg = 1, seglen = 0, V = 0, nblk = 0;
};

// Events up to max_single_L samples: one segment.  Longer ones are cut into nseg segments of seglen samples;
// a lag block advances by V = 512 floor((N - seglen + 1) / 512) lags.  nseg is chosen by cost: one
// transform + epilogue per lag block (~7.5 MAC steps' worth) and one MAC step per (block, segment, channel)
// that sees data.  Pz = 2 S + 1 lags are searched.
inline XcSegPlan plan_xc_segments(int L, int Pz, int Nc, int N, int max_single_L) {
  XcSegPlan p;
  const int S = (Pz - 1) / 2;
  p.seglen = L;
  p.V = (N - L + 1) / 512 * 512;
  if (L > max_single_L) {
    double best = 1e300;
    for (int ns = 2; ns <= 16; ++ns) {
      const int sl0 = (L + ns - 1) / ns;                    // shortest segments that cover the event
      const int v = std::min(15 * 512, (N - sl0 + 1) / 512 * 512);
      if (v < 4 * 512) continue;
      // the longest segments that advance allows (fewer (block, segment) products see data), every
      // segment non-empty
      const int sl = std::min(N - v + 1, (L + ns - 2) / (ns - 1) - 1);
      if (sl < sl0) continue;
      const int nb = (Pz + v - 1) / v;
      double cost = 0;
      for (int b = 0; b < nb; ++b) cost += 7.5 + Nc * std::min(ns, std::max(1, (S + L - b * v + sl - 1) / sl));
      if (cost < best) { best = cost; p.nseg = ns; p.seglen = sl; p.V = v; }
    }
  }
  p.nblk = p.V > 0 ? (Pz + p.V - 1) / p.V : 0;
  return p;
}

// Work items {later event c (and c + 1 when bit 30 of .y is set), first of TWO pair spectra}: `group` units (of two
// pair spectra) are walked together, `ctile` (even) later events at a time, the later events taken in twos inside a
// tile -- the workgroups of one later event share its block spectra in L2, those of one unit its pair spectra.  Rows b
// in [b_begin, b_end] hold the pairs of the caller's range; a unit is listed for the two later events when it holds a
// row below the larger of them.
inline std::vector<I2> plan_xc_items22(int N, int64_t b_begin, int64_t b_end, int group, int ctile) {
  std::vector<I2> items;
  const int unit = 2, npr = (N + 1) / 2;
  for (int g0 = 0; g0 < npr; g0 += unit * group) {
    const int64_t g_lo = 2 * (int64_t)g0, g_hi = 2 * (int64_t)(g0 + unit * group) - 1;
    if (g_hi < b_begin || g_lo > b_end) continue;
    for (int c0 = (int)std::max<int64_t>(g_lo + 1, 1); c0 < N; c0 += ctile)
      for (int pr0 = g0; pr0 < std::min(npr, g0 + unit * group); pr0 += unit) {
        const int64_t r_lo = 2 * (int64_t)pr0, r_hi = 2 * (int64_t)(pr0 + unit) - 1;
        if (r_hi < b_begin || r_lo > b_end) continue;
        const int c_end = std::min(N, c0 + ctile);
        for (int c = c0; c < c_end; c += 2) {
          const int has1 = c + 1 < c_end ? 1 : 0;
          if (c + has1 > r_lo) items.push_back(I2{c, pr0 | (has1 << 30)});
        }
      }
  }
  return items;
}

// ------------------------------------------------------------------------------------------------
// Rings of the context  (dtx_ctx.hip)
// ------------------------------------------------------------------------------------------------
// Timing ring: slot of the next pair; the slots dtx_kernel_ms_mean reads for the last n committed launches.
inline int ring_next_slot(long long committed, int size) { return (int)(committed % size); }
inline bool ring_last_slots(long long committed, int size, int n, int* slots) {
  if (n <= 0 || n > size || n > committed) return false;
  for (int q = 0; q < n; ++q) slots[q] = (int)((committed - 1 - q) % size);
  return true;
}
// Staging ring: bytes to allocate for a buffer that must hold `bytes`
inline size_t stage_capacity(size_t bytes) { return std::max<size_t>(bytes + bytes / 2, 4096); }
// Workspace growth (dtx_reserve)
inline size_t reserve_capacity(size_t bytes) { return bytes + bytes / 8 + 256; }

// ------------------------------------------------------------------------------------------------
// Ragged batches: host-side validation of a chunk table  (dtx_chunk_table_check)
// ------------------------------------------------------------------------------------------------
// tab[2 b] = offset of chunk b's first sample in the flat buffer, tab[2 b + 1] = its samples.  Returns 0 when
// every chunk lies inside a buffer of `numel` samples and is at most M long, else 1 + the first bad chunk.
inline int check_chunk_table(const int64_t* tab, int nchunk, int64_t M, int64_t numel) {
  for (int b = 0; b < nchunk; ++b) {
    const int64_t a = tab[2 * b], m = tab[2 * b + 1];
    if (a < 0 || m < 0 || m > M || a > numel || m > numel - a) return 1 + b;
  }
  return 0;
}

}  // namespace dtxplan
