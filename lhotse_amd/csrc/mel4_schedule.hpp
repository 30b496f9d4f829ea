// Host-side schedule of the mel filterbank for v_mfma_f32_4x4x1_16B_f32 (kernel_fft512c.hpp).  Pure C++ (no HIP): also
// compiled by tests/ on the CPU.
//
// One MFMA instruction = 16 independent blocks ("slots"); a slot multiplies 4 frames x 1 bin by 1 bin x 4 consecutive
// filters and accumulates, so over `steps` instructions it walks `steps` consecutive bins of the band of its filter group.
// An accumulator set = 16 slots = 4 rows of 4 slots (a row = 16 lanes = one DPP row).  A group whose (4-aligned) band is
// longer than the set's step count is split over 2, 3 or 4 adjacent slots of one row; the kernel adds the partial sums
// into the group's last slot with two row_shr DPP multiply-adds gated by the per-lane masks m4 / m8:
//     v += m4 * v[lane - 4];  v += m8 * v[lane - 8]
//   2 slots (positions 0,1 or 2,3): m4 on the second;  3 slots (positions 0-2): m4 and m8 on the third;
//   4 slots: m4 on the second and fourth, m8 on the fourth.
// Every slot starts at a bin that is a multiple of 4, so that a lane's operands of 4 consecutive steps are one 16-byte
// LDS read.  The builder tries every combination of per-set step counts (multiples of 4) and keeps the one with the
// fewest total steps.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

namespace hipfeat {

// "no output from this lane": a column no filterbank has.  (Not -1: the integer tables live in LDS as float words, and
// 0xFFFFFFFF is a NaN that a later kernel could pick up from never-written LDS padding.)
constexpr int kMel4NoColumn = 1 << 20;

struct Mel4Schedule {
  int nsets = 0;
  int steps[4] = {0, 0, 0, 0};  // MFMA steps per set (multiples of 4)
  int step0[4] = {0, 0, 0, 0};  // first step of the set in the weight table
  std::vector<float> wtab;      // [total steps / 4][64 lanes][4 steps]: B operands
  std::vector<float> ltab;      // [nsets][64 lanes][4]: power-row offset (int bits), output column (int bits, kMel4NoColumn = none), m4, m8
};

// h_mel: [K][M] row-major filterbank (bin x filter).  prow_stride: floats between the power rows of consecutive frames.
inline bool build_mel4_schedule(const float* h_mel, int M, int K, int prow_stride, int max_sets, int max_steps, Mel4Schedule& out) {
  const int ng = (M + 3) / 4;
  std::vector<int> lo(ng, 0), alen(ng, 4);  // 4-aligned band start, length from there to the last non-zero bin
  for (int g = 0; g < ng; ++g) {
    int l = K, h = 0;
    for (int k = 0; k < K; ++k)
      for (int j = 4 * g; j < std::min(M, 4 * g + 4); ++j)
        if (h_mel[(size_t)k * M + j] != 0.0f) {
          l = std::min(l, k);
          h = std::max(h, k + 1);
        }
    if (h > 0) lo[g] = l & ~3, alen[g] = h - (l & ~3);
  }
  // pstride: bins between the starts of consecutive pieces of a split group; shift: bins (multiple of 4) the first piece starts below lo
  struct Place { int set, row, pos, cnt, pstride, shift; };
  struct Plan { int total = 1 << 30; std::vector<int> T; std::vector<Place> place; };
  Plan best;
  std::vector<int> order(ng);
  for (int g = 0; g < ng; ++g) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return alen[a] > alen[b]; });

  // enumerate non-increasing step counts T[0] >= T[1] >= ... (multiples of 4) for 1 .. max_sets sets
  std::vector<int> T;
  auto try_combo = [&]() {
    const int ns = (int)T.size();
    int total = 0;
    for (int t : T) total += t;
    if (total >= best.total) return;
    // free[s][r][pos]; quads[s][r][q]: slots of DPP row r whose reads start on bank quad q (mod 16 dwords).  A 16-lane group of the
    // kernels' ds_read_b128 is conflict-free only if its four slots start on four different quads (tools/ubench/lds_rate.hip: 3.9 /
    // 7.1 / 12.1 clk per instruction for 1 / 2 / 4 slots per quad), so placement minimises the worst multiplicity of the row.
    std::vector<std::vector<std::vector<char>>> used(ns, std::vector<std::vector<char>>(4, std::vector<char>(4, 0)));
    std::vector<std::vector<std::vector<int>>> quads(ns, std::vector<std::vector<int>>(4, std::vector<int>(4, 0)));
    std::vector<Place> place(ng);
    auto fit = [&](int g, int s, int c, Place& pl) {
      const int Ts = T[s];
      // pieces of a split group: a stride that is an odd number of quads puts them on different quads by itself; otherwise stagger
      // them by one quad less (Ts - 4) when the band still fits
      int pstride = Ts;
      if (c > 1 && ((Ts / 4) & 1) == 0 && (c > 2 || ((Ts / 4) & 3) == 0) && (c - 1) * (Ts - 4) + Ts >= alen[g]) pstride = Ts - 4;
      const int cap = (c - 1) * pstride + Ts;
      int best_cost = 1 << 30;
      for (int r = 0; r < 4; ++r) {
        const int step = c == 2 ? 2 : (c == 1 ? 1 : 4);
        for (int pos = 0; pos + c <= 4; pos += step) {
          bool fr = true;
          for (int k = 0; k < c; ++k) fr = fr && !used[s][r][pos + k];
          if (!fr) continue;
          for (int shift = 0; shift <= 12 && shift <= lo[g] && alen[g] + shift <= cap; shift += 4) {
            int q[4] = {quads[s][r][0], quads[s][r][1], quads[s][r][2], quads[s][r][3]};
            for (int k = 0; k < c; ++k) ++q[((lo[g] - shift + k * pstride) / 4) & 3];
            const int mult = std::max(std::max(q[0], q[1]), std::max(q[2], q[3]));
            const int cost = mult * 64 + r * 4 + pos + (shift ? 16 : 0);  // fewest conflicts, then first fit, unshifted preferred
            if (cost < best_cost) {
              best_cost = cost;
              pl = Place{s, r, pos, c, pstride, shift};
            }
          }
          break;  // positions further right in the same row are equivalent for the row's quad census
        }
      }
      return best_cost != (1 << 30);
    };
    for (int g : order) {
      // candidate sets ordered by slots needed (fewest first), then by smaller step count
      std::vector<int> cand;
      for (int s = 0; s < ns; ++s)
        if ((alen[g] + T[s] - 1) / T[s] <= 4) cand.push_back(s);
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) {
        const int ca = (alen[g] + T[a] - 1) / T[a], cb = (alen[g] + T[b] - 1) / T[b];
        return ca != cb ? ca < cb : T[a] < T[b];
      });
      bool done = false;
      for (int s : cand) {
        Place pl;
        if (fit(g, s, (alen[g] + T[s] - 1) / T[s], pl)) {
          for (int k = 0; k < pl.cnt; ++k) {
            used[s][pl.row][pl.pos + k] = 1;
            ++quads[s][pl.row][((lo[g] - pl.shift + k * pl.pstride) / 4) & 3];
          }
          place[g] = pl;
          done = true;
          break;
        }
      }
      if (!done) return;
    }
    best.total = total;
    best.T = T;
    best.place = place;
  };
  for (int ns = 1; ns <= max_sets; ++ns) {
    T.assign(ns, 4);
    // odometer over non-increasing sequences
    std::vector<int> idx(ns, 0);
    const int nopt = max_steps / 4;
    while (true) {
      bool mono = true;
      for (int s = 1; s < ns; ++s) mono = mono && idx[s] <= idx[s - 1];
      if (mono) {
        for (int s = 0; s < ns; ++s) T[s] = 4 * (idx[s] + 1);
        try_combo();
      }
      int d = ns - 1;
      while (d >= 0 && ++idx[d] == nopt) idx[d--] = 0;
      if (d < 0) break;
    }
  }
  if (best.total == (1 << 30)) return false;

  const int ns = (int)best.T.size();
  out.nsets = ns;
  int total = 0;
  for (int s = 0; s < ns; ++s) {
    out.steps[s] = best.T[s];
    out.step0[s] = total;
    total += best.T[s];
  }
  out.wtab.assign((size_t)total * 64, 0.0f);
  out.ltab.assign((size_t)ns * 256, 0.0f);
  auto bits = [](int v) { float f; std::memcpy(&f, &v, 4); return f; };
  for (int s = 0; s < ns; ++s)
    for (int lane = 0; lane < 64; ++lane) {  // unused slots: offset 0, no output
      float* lt = out.ltab.data() + ((size_t)s * 64 + lane) * 4;
      lt[0] = bits((lane & 3) * prow_stride);
      lt[1] = bits(kMel4NoColumn);
    }
  for (int g = 0; g < ng; ++g) {
    const Place& pl = best.place[g];
    const int Ts = best.T[pl.set];
    for (int k = 0; k < pl.cnt; ++k) {
      const int b = 4 * pl.row + pl.pos + k;  // slot index inside the set
      int bin0 = lo[g] - pl.shift + k * pl.pstride;
      const int slo = bin0, shi = std::min(lo[g] + alen[g], k + 1 < pl.cnt ? bin0 + pl.pstride : bin0 + Ts);
      bin0 = std::max(0, std::min(bin0, (prow_stride - Ts) & ~3));
      for (int t = 0; t < Ts; ++t) {
        const int bin = bin0 + t;
        if (bin < slo || bin >= shi || bin >= K) continue;
        const int step = out.step0[pl.set] + t;
        for (int j = 0; j < 4; ++j)
          if (4 * g + j < M) out.wtab[(((size_t)step / 4) * 64 + 4 * b + j) * 4 + (step & 3)] = h_mel[(size_t)bin * M + 4 * g + j];
      }
      const bool last = k == pl.cnt - 1;
      float m4 = 0.f, m8 = 0.f;
      if (pl.cnt == 2 && k == 1) m4 = 1.f;
      if (pl.cnt == 3 && k == 2) m4 = 1.f, m8 = 1.f;
      if (pl.cnt == 4 && (k == 1 || k == 3)) m4 = 1.f;
      if (pl.cnt == 4 && k == 3) m8 = 1.f;
      for (int i = 0; i < 4; ++i) {
        float* lt = out.ltab.data() + ((size_t)pl.set * 64 + 4 * b + i) * 4;
        lt[0] = bits(i * prow_stride + bin0);
        lt[1] = bits(last && 4 * g + i < M ? 4 * g + i : kMel4NoColumn);
        lt[2] = m4;
        lt[3] = m8;
      }
    }
  }
  return true;
}

}  // namespace hipfeat
