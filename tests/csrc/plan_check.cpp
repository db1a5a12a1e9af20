// CPU sanitizer target of the host-side planning code (detex_amd/csrc/dtx_plan.h): compiled with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_plan_sanitizer.py and run over the shapes of the GPU sweeps (tests/test_gpu_random_sweep.py:
// L = 4 .. 16384, Nc in {1, 2, 3, 4, 6}, 1 .. 450 detectors of 1 .. 100 vectors, 1 .. 24 chunks) and of the
// BASELINE configurations.  Every table a launch path uploads is checked for the invariants its kernel relies
// on; any out-of-bounds access, signed overflow or misaligned read inside the planners aborts the run.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>

#include "../../detex_amd/csrc/dtx_plan.h"

using namespace dtxplan;

static long g_checks = 0;
#define CHECK(cond)                                                                   \
  do {                                                                                \
    ++g_checks;                                                                       \
    if (!(cond)) {                                                                    \
      std::fprintf(stderr, "plan_check: %s failed at %s:%d\n", #cond, __FILE__, __LINE__); \
      std::abort();                                                                   \
    }                                                                                 \
  } while (0)

constexpr int kN = 8192, kMaxSingle = kN - 2560 + 1, kUnitCap = 8;

static void check_basis(const std::vector<int32_t>& off, int L) {
  const int D = (int)off.size() - 1;
  FftBasisPlan f = plan_fft_basis(off.data(), D, L, kN, kMaxSingle, kUnitCap);
  CHECK(f.V % 512 == 0 && f.V >= 512 && f.V <= kN);
  if (f.nseg == 1) CHECK(f.V + L - 1 <= kN);
  else CHECK(f.nseg * (kN / 2) >= L && (f.nseg - 1) * (kN / 2) < L && f.V == kN / 2);
  CHECK((int)f.prow.size() == f.npairs && (int)f.pdet.size() == f.npairs && (int)f.units.size() == f.nunits);
  // every basis row exactly once
  std::vector<int> seen(off[D], 0);
  for (const I2& r : f.prow) {
    CHECK(r.x >= 0 && r.x < off[D]);
    ++seen[r.x];
    if (r.y >= 0) { CHECK(r.y < off[D]); ++seen[r.y]; }
  }
  for (int v : seen) CHECK(v == 1);
  // a detector's pairs are contiguous and closed by its last pair; shared pairs hold two single-vector detectors
  int p = 0;
  bool two = false;
  for (int d = 0; d < D;) {
    const int k = off[d + 1] - off[d];
    if (k == 1 && d + 1 < D && off[d + 2] - off[d + 1] == 1) {
      CHECK(f.pdet[p].x == d && f.pdet[p].y == d + 1 && f.prow[p].x == off[d] && f.prow[p].y == off[d] + 1);
      two = true;
      ++p; d += 2;
      continue;
    }
    for (int q = 0; q < k; q += 2, ++p) {
      CHECK(f.prow[p].x == off[d] + q && f.prow[p].y == (q + 1 < k ? off[d] + q + 1 : -1));
      CHECK(f.pdet[p].x == (q + 2 >= k ? d : -1) && f.pdet[p].y == -1);
    }
    ++d;
  }
  CHECK(p == f.npairs && two == f.two);
  // units tile the pairs, hold whole detectors, respect the cap unless one detector is larger, and split at a
  // detector boundary
  int at = 0, c0 = 0, c1 = 0;
  for (size_t u = 0; u < f.units.size(); ++u) {
    const I4& un = f.units[u];
    CHECK(un.x == at && un.y > 0 && un.z >= 0 && un.z <= un.y);
    CHECK(f.pdet[un.x + un.y - 1].x >= 0);                       // ends at a detector boundary
    if (un.z > 0) CHECK(f.pdet[un.x + un.z - 1].x >= 0);
    if (un.y > kUnitCap) {                                        // one detector of its own
      for (int q = 0; q + 1 < un.y; ++q) CHECK(f.pdet[un.x + q].x < 0);
    }
    CHECK(f.upre[u].x == c0 && f.upre[u].y == c1 && f.unit_pairs[u] == un.y);
    c0 += un.z; c1 += un.y - un.z;
    at += un.y;
  }
  CHECK(at == f.npairs && c0 == f.cum0 && c1 == f.cum1 && f.cum0 + f.cum1 == f.npairs);
  CHECK(f.upre.back().x == c0 && f.upre.back().y == c1);
  // the half order is a permutation: half 0's pairs of every unit first
  FftBasisPlan g = f;
  reorder_pairs_for_halves(g);
  CHECK(g.prow.size() == f.prow.size());
  std::multiset<std::pair<int, int>> a, b;
  for (const I2& r : f.prow) a.insert({r.x, r.y});
  for (const I2& r : g.prow) b.insert({r.x, r.y});
  CHECK(a == b);
  for (size_t u = 0; u < f.units.size(); ++u) {
    const I4& un = f.units[u];
    for (int q = 0; q < un.z; ++q) CHECK(g.prow[f.upre[u].x + q].x == f.prow[un.x + q].x);
    for (int q = un.z; q < un.y; ++q) CHECK(g.prow[f.cum0 + f.upre[u].y + q - un.z].x == f.prow[un.x + q].x);
  }
  // the re-scan's runs (plain and half order): every pair in exactly one run, every run ends at a detector border
  for (int re = 0; re < 2; ++re) {
    const FftBasisPlan& t = re ? g : f;
    const std::vector<I2> runs = plan_fft_runs(t, re != 0);
    std::vector<int> cover(t.npairs, 0);
    for (const I2& r : runs) {
      CHECK(r.x >= 0 && r.y > 0 && r.x + r.y <= t.npairs);
      CHECK(t.pdet[r.x + r.y - 1].x >= 0);
      if (r.x > 0) CHECK(t.pdet[r.x - 1].x >= 0);
      for (int q = 0; q < r.y; ++q) ++cover[r.x + q];
    }
    for (int v : cover) CHECK(v == 1);
  }
  // the repair pass's row groups over K-tiles packed like dtx_basis_upload packs them (whole detectors, <= 208 rows)
  for (int d = 0; d < D;) {
    int dend = d;
    while (dend < D && off[dend + 1] - off[d] <= 208) ++dend;
    if (dend == d) break;                                        // (a detector of more than 208 vectors is refused at upload)
    const int nd = dend - d, rows = off[dend] - off[d], rm = (rows + 15) / 16;
    std::vector<int32_t> doff(nd + 1);
    for (int e = 0; e <= nd; ++e) doff[e] = off[d + e] - off[d];
    const RepairSubPlan rp = plan_repair_subtiles(doff.data(), nd, rm);
    CHECK(rp.rms >= 1 && rp.rms <= rm);
    int dat = 0;
    for (const I4& sb : rp.sub) {
      CHECK(sb.y == dat && sb.z >= 1 && sb.x >= 0 && sb.x + rp.rms <= rm);
      CHECK(sb.w >= 0 && sb.w + sb.z + 1 <= (int)rp.sdoff.size());
      for (int q = 0; q <= sb.z; ++q) {
        const int v = rp.sdoff[sb.w + q];
        CHECK(v == doff[sb.y + q] - 16 * sb.x && v >= 0 && v <= 16 * rp.rms);   // rows lie inside the sub-tile's blocks
      }
      dat += sb.z;
    }
    CHECK(dat == nd);
    d = dend;
  }
  // the ping-pong grid over this basis for a few block counts / CU counts
  for (int64_t nb : {1, 7, 255, 256, 257, 1752, 5000}) {
    for (int ncu : {1, 64, 256, 304}) {
      for (int force : {0, 3}) {
        PpGridPlan gp = plan_pp_grid(nb, ncu, f.nunits, f.npairs, f.unit_pairs.data(), force);
        CHECK(gp.nfull + gp.nrem == nb && gp.nfull % ncu == 0 && gp.nrem >= 0);
        if (force == 0) CHECK(gp.nrem < ncu);
        CHECK(gp.grid == (int64_t)gp.nfull + (int64_t)gp.nrem * gp.ng);
        if (gp.nrem > 0) {
          CHECK(gp.ng >= 1 && gp.ng <= std::min(f.nunits, kPlanMaxGroups));
          CHECK(gp.gb[0] == 0 && gp.gb[gp.ng] == f.nunits);
          for (int k = 0; k < gp.ng; ++k) CHECK(gp.gb[k] < gp.gb[k + 1]);
        }
      }
    }
  }
}

static void check_xc(int N, int L, int Nc) {
  const int trunc = L / 2 - 1 < 0 ? 0 : L / 2 - 1;              // (n // (2 Nc) - 1 with n = Nc L)
  const int S = (L - 1) - trunc, Pz = 2 * S + 1;
  XcSegPlan sp = plan_xc_segments(L, Pz, Nc, kN, kMaxSingle);
  CHECK(sp.V % 512 == 0 && sp.V >= 512 && sp.V <= 15 * 512 + 512);
  CHECK(sp.seglen >= 1 && sp.seglen + sp.V - 1 <= kN);
  CHECK((int64_t)sp.nseg * sp.seglen >= L && (int64_t)(sp.nseg - 1) * sp.seglen < L);
  CHECK(sp.nblk == (Pz + sp.V - 1) / sp.V && sp.nblk >= 1);
  if (L > kMaxSingle) CHECK(sp.V >= 4 * 512 && sp.nseg >= 2 && sp.nseg <= 16);
  const int64_t npairs = (int64_t)N * (N - 1) / 2;
  auto row_of = [&](int64_t idx) {                             // row b of condensed index idx
    int64_t b = 0;
    while ((b + 1) * N - (b + 1) * (b + 2) / 2 <= idx) ++b;
    return b;
  };
  const int group = 8, ctile = 4;                                // the shipped work list (dtx_xcfft.hip)
  std::vector<std::pair<int64_t, int64_t>> ranges = {{0, npairs}};
  if (npairs > 3) { ranges.push_back({1, npairs / 2}); ranges.push_back({npairs / 2, npairs}); ranges.push_back({npairs - 1, npairs}); }
  for (auto [pb, pe] : ranges) {
    const int64_t bb = row_of(pb), be = row_of(pe - 1);
    // units of two pair spectra against one or two later events; every
    // (later event, unit) of the range exactly once, nothing listed that holds no pair
    for (int ct : {ctile, 2, 6}) {
      std::vector<I2> items = plan_xc_items22(N, bb, be, group, ct);
      std::set<std::pair<int, int>> seen;
      const int npr = (N + 1) / 2;
      for (const I2& it : items) {
        const int has1 = (it.y >> 30) & 1, pr0 = it.y & 0x3fffffff;
        CHECK(it.x >= 1 && it.x + has1 < N && pr0 >= 0 && pr0 < npr && pr0 % 2 == 0);
        CHECK(it.x + has1 > 2 * pr0);
        CHECK(seen.insert({it.x, pr0}).second);
        if (has1) CHECK(seen.insert({it.x + 1, pr0}).second);
      }
      for (int64_t b = bb; b <= be; ++b)
        for (int c = (int)b + 1; c < N; ++c) CHECK(seen.count({c, (int)(b / 4 * 4 / 2)}) == 1);
    }
  }
}

int main() {
  std::mt19937 rng(12345);
  auto rnd = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
  const int ks[] = {1, 1, 2, 3, 5, 8, 13, 16, 17, 40, 100};
  // the shapes of test_random_scan / test_random_scan_long + the BASELINE banks
  std::vector<int> Ls = {4, 63, 64, 255, 256, 300, 700, 1100, 2100, 3000, 5000, 5633, 5634, 7681, 9000, 13000, 16384, 32768};
  for (int rep = 0; rep < 400; ++rep) {
    std::vector<int32_t> off = {0};
    const int nd = rep % 5 == 0 ? rnd(200, 450) : rnd(1, 7);
    for (int d = 0; d < nd; ++d) off.push_back(off.back() + (rep % 5 == 0 ? 1 + (rnd(0, 9) == 0) : ks[rnd(0, 10)]));
    check_basis(off, rep < (int)Ls.size() ? Ls[rep] : rnd(2, 32768));
  }
  {
    std::vector<int32_t> off;                                    // configs[1]: 25 x 8; configs[4]: 1 x 64; 200 singles
    for (int d = 0; d <= 25; ++d) off.push_back(8 * d);
    check_basis(off, 3000);
    check_basis({0, 64}, 1000);
    off.clear();
    for (int d = 0; d <= 200; ++d) off.push_back(d);
    check_basis(off, 3000);
    check_basis({0, 1}, 2);
  }
  for (int L : {1024, 1500, 2500, 3000, 5200, 5633, 5634, 9000, 13000, 20000, 32768})
    for (int Nc : {1, 3, 6})
      for (int N : {2, 3, 9, 23, 64, 209}) check_xc(N, L, Nc);
  check_xc(2000 > 400 ? 400 : 2000, 13000, 3);                  // configs[2] geometry at a coverage-checkable N
  // rings and capacities
  for (long long committed : {0LL, 1LL, 5LL, 63LL, 64LL, 65LL, 1000003LL}) {
    int slots[64];
    for (int n : {0, 1, 2, 64, 65}) {
      const bool ok = ring_last_slots(committed, 64, n, slots);
      CHECK(ok == (n >= 1 && n <= 64 && n <= committed));
      if (ok) {
        CHECK(slots[0] == (ring_next_slot(committed, 64) + 63) % 64);
        for (int q = 0; q < n; ++q) CHECK(slots[q] >= 0 && slots[q] < 64);
      }
    }
  }
  for (size_t b : {(size_t)0, (size_t)1, (size_t)4096, (size_t)1 << 33}) {
    CHECK(stage_capacity(b) >= b && stage_capacity(b) >= 4096);
    CHECK(reserve_capacity(b) > b);
  }
  // chunk tables (ragged batches)
  {
    const int64_t good[] = {0, 100, 50, 120, 170, 0};
    CHECK(check_chunk_table(good, 3, 120, 170) == 0);
    const int64_t past[] = {0, 100, 80, 100};
    CHECK(check_chunk_table(past, 2, 100, 170) == 2);
    const int64_t neg[] = {-1, 10};
    CHECK(check_chunk_table(neg, 1, 100, 170) == 1);
    const int64_t longc[] = {0, 101};
    CHECK(check_chunk_table(longc, 1, 100, 170) == 1);
    const int64_t huge[] = {INT64_MAX - 5, 10};                  // no signed overflow in the bound test
    CHECK(check_chunk_table(huge, 1, 100, INT64_MAX) == 1);
  }
  std::printf("plan_check ok: %ld checks\n", g_checks);
  return 0;
}
