// Whisper log-mel fast path (frame 400, any shift): the 400-point real DFT as small f32 GEMMs on the matrix cores.
//
// Reference: log_mel_spectrogram (lhotse/features/whisper_fbank.py:62-68): torch.stft(n_fft=400, hop=160, hann window,
// center/reflect) -> |.|^2 -> filters @ magnitudes -> log10(clamp(1e-10)).  400 = 2^4 * 5^2 has no cheap radix-16
// FFT on 16-lane groups, and a direct DFT costs 160 k MACs per frame; instead, with y the windowed frame:
//
//   one radix-2 decimation-in-frequency step      a[n] = y[n] + y[n+200],  b[n] = y[n] - y[n+200]      (n < 200)
//     even bins  X[2m]   = DFT200(a)[m]           odd bins  X[2m+1] = sum_n b[n] W400^(n(2m+1))
//   the real-input symmetry of both halves        a[n] +- a[200-n],  b[n] -+ b[200-n]                  (0 < n < 100)
//
// leaves four real GEMMs with K ~ 100:  Re/Im of the 101 even bins and of the 100 odd bins -- 40 k MACs per frame.
// One workgroup (2 waves) = one tile of 16 frames: each wave gathers the samples of 8 frames and writes their four
// symmetrised vectors to the LDS rows; then wave 0 takes the even bins and wave 1 the odd bins: it keeps the B
// operands (the 16 frames' vectors of its parity) in registers and streams the cos/sin matrices (A operands,
// precomputed on the host in MFMA lane order, L2 resident, double buffered in registers) through
// v_mfma_f32_16x16x4_f32: D[bin][frame] += C[bin][n] * V[n][frame].  The power |X|^2 of a 16-bin tile is formed in
// the accumulator registers, which ARE the B operand layout of the next GEMM (k order = accumulator row order), so the
// mel filterbank follows as further MFMAs without touching LDS: D[mel][frame] += W[mel][bin] * P[bin][frame], only for
// the (bin tile, mel tile) pairs that hold non-zero weights.  The two waves exchange their partial mel sums through LDS
// (each finishes half of the mel tiles).  Epilogue: log10(max(., 1e-10)), 16-byte stores; the per-cut normalisation is
// whisper_norm_kernel (kernel_generic.hpp).  26.9 KB LDS per workgroup -> 5 workgroups = 10 waves per CU, so one
// workgroup's sample gather (HBM latency) overlaps the others' matrix-core phase.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

constexpr int kWhN = 400;
constexpr int kWhBinTiles = 14;   // 7 tiles of 16 even bins (2m, m <= 100), then 7 tiles of 16 odd bins (2m+1, m <= 99)
constexpr int kWhCosSteps = 26;   // k-steps of 4 over the cos vectors (101 / 100 entries)
constexpr int kWhSinSteps = 25;   // k-steps of 4 over the sin vectors (99 / 100 entries)
constexpr int kWhSteps = kWhCosSteps + kWhSinSteps;
constexpr int kWhRowStride = 420;  // floats per frame row in LDS (== 4 mod 32: conflict-free operand reads)
constexpr int kWhOffCosE = 0, kWhOffSinE = 104, kWhOffCosO = 204, kWhOffSinO = 304;
constexpr int kWhMaxMelTiles = 8;  // num_filters <= 128
constexpr int kWhSlots = 4;        // mel tiles a 16-bin tile may feed (consecutive); mel_a holds kWhSlots per bin tile

struct WhisperParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;  // [400]
  const float* dft_a;   // [14 bin tiles][51 steps][64 lanes]
  const float* mel_a;   // [14 bin tiles][kWhSlots][4 k-steps][64 lanes]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, shift, M;
  float mel_floor;
  int32_t ablate;  // experiments only (HIPFEAT_WH_ABLATE): 1 skips phase 1, 2 skips phase 2
  int32_t mt_lo[kWhBinTiles];    // first mel tile with non-zero weights in bin tile bt
  int32_t mt_cnt[kWhBinTiles];   // number of consecutive mel tiles from mt_lo (<= kWhSlots)
};

template <int NMT>
__global__ __launch_bounds__(128) void whisper_kernel(const WhisperParams p) {
  __shared__ __attribute__((aligned(16))) float rows[16 * kWhRowStride];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = blockIdx.x;
  int cut, fb;
  if (p.uniform_bpc > 0) {
    cut = blk / p.uniform_bpc;
    fb = blk - cut * p.uniform_bpc;
  } else {
    cut = find_cut(p.cuts, p.num_cuts, blk);
    fb = blk - p.cuts[cut].first_block;
  }
  const CutDesc cd = p.cuts[cut];
  const int f0 = fb * 16;
  const int nf = min(16, cd.num_frames - f0);
  const float* __restrict__ w = p.wave + cd.wave_off;
  const float* __restrict__ win = p.window;

  // ---- phase 1: windowed frames -> symmetrised vectors in LDS (centred frames, "reflect" edges) --------------------
  // lane <-> n (two passes: n = lane and n = 64 + lane <= 100), frames unrolled so that many loads are in flight
  const int S = cd.num_samples;
  const int jt = (f0 * p.shift) - kWhN / 2;                   // first sample of the tile (may be negative)
  const bool interior = jt >= 0 && jt + 15 * p.shift + kWhN <= S;  // uniform: no reflection anywhere in the tile
  auto sample = [&](int j) -> float {
    if (!interior) {
      if (j < 0) j = -j;
      if (j >= S) j = 2 * S - 2 - j;
      return ((unsigned)j < (unsigned)S) ? w[j] : 0.0f;
    }
    return w[j];
  };
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = 64 * half + lane;
    if (n <= 100 && !(p.ablate & 1)) {
      const float w0 = win[n], w2 = win[n + 200], w1 = win[200 - n], w3 = win[(400 - n) % 400];
#pragma unroll 4
      for (int ff = 0; ff < 8; ++ff) {
        const int f = 8 * wv + ff;
        const int j = jt + f * p.shift;
        float* row = rows + f * kWhRowStride;
        const float y0 = w0 * sample(j + n);
        const float y2 = w2 * sample(j + n + 200);
        const float y1 = w1 * sample(j + 200 - n);
        const float y3 = w3 * sample(j + (400 - n) % 400);  // n == 0: unused
        const float an = y0 + y2, bn = y0 - y2;
        const float am = y1 + y3, bm = y1 - y3;
        if (n == 0) {
          row[kWhOffCosE] = an;
          row[kWhOffCosO] = bn;
        } else if (n == 100) {
          row[kWhOffCosE + 100] = an;
          row[kWhOffSinO + 99] = bn;
        } else {
          row[kWhOffCosE + n] = an + am;
          row[kWhOffSinE + n - 1] = an - am;
          row[kWhOffCosO + n] = bn - bm;
          row[kWhOffSinO + n - 1] = bn + bm;
        }
      }
    }
  }
  if (wv != 0) {
  } else if (lane < 48) {  // zero the k padding of the even-cos vector (entries 101 .. 103)
    const int f = lane / 3, e = lane - 3 * f;
    rows[f * kWhRowStride + kWhOffCosE + 101 + e] = 0.f;
  } else {  // and the unused last entry of the even-sin vector (its coefficient is 0, but 0 * garbage may be NaN)
    rows[(lane - 48) * kWhRowStride + kWhOffSinE + 99] = 0.f;
  }
  __syncthreads();

  // ---- phase 2: DFT GEMMs -> power -> mel GEMM, all on the matrix cores --------------------------------------------
  const int fr = lane & 15, g = lane >> 4;
  f32x4 macc[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) macc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ da = p.dft_a + lane;
  const float* __restrict__ ma = p.mel_a + lane;

  // A operands are double buffered in registers: tile bt+1's cos/sin rows and mel weights are requested before tile
  // bt's MFMAs start, so one L2 round trip hides behind ~55 MFMAs (a wave is alone on its SIMD most of the time).
  float bc[kWhCosSteps], bs[kWhSinSteps];
  auto load_b = [&](int par) {
    const float* vc = rows + fr * kWhRowStride + (par ? kWhOffCosO : kWhOffCosE) + g;
    const float* vs = rows + fr * kWhRowStride + (par ? kWhOffSinO : kWhOffSinE) + g;
#pragma unroll
    for (int s = 0; s < kWhCosSteps; ++s) bc[s] = vc[4 * s];
#pragma unroll
    for (int s = 0; s < kWhSinSteps; ++s) bs[s] = vs[4 * s];
  };
  auto load_a = [&](int bt, float (&A)[kWhSteps], float (&MA)[kWhSlots * 4]) {
    const float* __restrict__ a = da + (size_t)bt * kWhSteps * 64;
#pragma unroll
    for (int s = 0; s < kWhSteps; ++s) A[s] = a[s * 64];
    const float* __restrict__ m = ma + (size_t)bt * kWhSlots * 256;
#pragma unroll
    for (int i = 0; i < kWhSlots * 4; ++i) MA[i] = m[i * 64];
  };
  auto compute = [&](int bt, const float (&A)[kWhSteps], const float (&MA)[kWhSlots * 4]) {
    f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kWhSinSteps; ++s) {  // two independent accumulation chains, interleaved
      re = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s], bc[s], re, 0, 0, 0);
      im = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kWhCosSteps + s], bs[s], im, 0, 0, 0);
    }
    re = __builtin_amdgcn_mfma_f32_16x16x4f32(A[kWhCosSteps - 1], bc[kWhCosSteps - 1], re, 0, 0, 0);
    const f32x4 pw = re * re + im * im;  // |X|^2 of bins (tile rows 4g + r), frame fr
    const int lo = p.mt_lo[bt], cnt = p.mt_cnt[bt];
#pragma unroll
    for (int i = 0; i < kWhSlots; ++i) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (i < cnt && mt == lo + i) {  // uniform
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(MA[4 * i + 0], pw.x, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(MA[4 * i + 1], pw.y, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(MA[4 * i + 2], pw.z, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(MA[4 * i + 3], pw.w, macc[mt], 0, 0, 0);
        }
      }
    }
  };
  float A0[kWhSteps], A1[kWhSteps], MA0[kWhSlots * 4], MA1[kWhSlots * 4];
  const int bt0 = 7 * wv;  // wave 0: bin tiles 0 .. 6 (even bins), wave 1: 7 .. 13 (odd bins)
  load_a(bt0, A0, MA0);
  load_b(wv);
  if (!(p.ablate & 2)) {
#pragma unroll 1
    for (int t = 0; t < 6; t += 2) {
      load_a(bt0 + t + 1, A1, MA1);
      compute(bt0 + t, A0, MA0);
      load_a(bt0 + t + 2, A0, MA0);
      compute(bt0 + t + 1, A1, MA1);
    }
    compute(bt0 + 6, A0, MA0);
  }
  // exchange partial mel sums: a wave finishes the mel tiles of its own parity (mt & 1 == wv) and hands over the others
  __syncthreads();  // every B operand has been read: the rows are dead
  float* xch = rows;  // [NMT][4][64] partials of the tiles the OTHER wave finishes
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
    if ((mt & 1) != wv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) xch[(mt * 4 + r) * 64 + lane] = macc[mt][r];
    }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
    if ((mt & 1) == wv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) macc[mt][r] += xch[(mt * 4 + r) * 64 + lane];
    }

  // ---- epilogue: log10(max(mel, floor)); lane holds mels 16 mt + 4 g + r of frame fr --------------------------------
  if (fr < nf) {
    float* orow = p.out + (cd.out_row + f0 + fr) * p.out_stride;
    const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      if ((mt & 1) != wv) continue;  // uniform
      const int m0 = 16 * mt + 4 * g;
      f32x4 v;
      v.x = log10f(fmaxf(macc[mt].x, p.mel_floor));
      v.y = log10f(fmaxf(macc[mt].y, p.mel_floor));
      v.z = log10f(fmaxf(macc[mt].z, p.mel_floor));
      v.w = log10f(fmaxf(macc[mt].w, p.mel_floor));
      if (vec_ok && m0 + 3 < p.M) {
        *reinterpret_cast<f32x4*>(orow + m0) = v;
      } else {
        if (m0 + 0 < p.M) orow[m0 + 0] = v.x;
        if (m0 + 1 < p.M) orow[m0 + 1] = v.y;
        if (m0 + 2 < p.M) orow[m0 + 2] = v.z;
        if (m0 + 3 < p.M) orow[m0 + 3] = v.w;
      }
    }
  }
}

}  // namespace hipfeat
