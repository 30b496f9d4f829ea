// Test shim (CPU only): C entry point around lhotse_amd/csrc/mel4_schedule.hpp so that tests/test_mel4_schedule.py can
// emulate the kernel's matrix-core mel phase from the very tables the plan uploads.
#include "../../lhotse_amd/csrc/mel4_schedule.hpp"
extern "C" int mel4_build(const float* h_mel, int M, int K, int prow_stride, int max_sets, int max_steps, int* nsets, int* steps, int* step0,
                          float* wtab, int wtab_cap, float* ltab, int ltab_cap) {
  hipfeat::Mel4Schedule s;
  if (!hipfeat::build_mel4_schedule(h_mel, M, K, prow_stride, max_sets, max_steps, s)) return 0;
  if ((int)s.wtab.size() > wtab_cap || (int)s.ltab.size() > ltab_cap) return -1;
  *nsets = s.nsets;
  for (int i = 0; i < 4; ++i) steps[i] = s.steps[i], step0[i] = s.step0[i];
  std::memcpy(wtab, s.wtab.data(), s.wtab.size() * sizeof(float));
  std::memcpy(ltab, s.ltab.data(), s.ltab.size() * sizeof(float));
  return (int)s.wtab.size();
}
