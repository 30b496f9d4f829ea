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
// One wave = one tile of 16 frames: it gathers the samples, writes the four symmetrised vectors of every frame to its
// LDS rows, then for each parity keeps the B operands (its frames' vectors) in registers and streams the cos/sin
// matrices (A operands, precomputed on the host in MFMA lane order, L2/L1 resident) through
// v_mfma_f32_16x16x4_f32: D[bin][frame] += C[bin][n] * V[n][frame].  The power |X|^2 of a 16-bin tile is formed in
// the accumulator registers, which ARE the B operand layout of the next GEMM (k order = accumulator row order), so the
// mel filterbank follows as further MFMAs without touching LDS: D[mel][frame] += W[mel][bin] * P[bin][frame], only for
// the (bin tile, mel tile) pairs that hold non-zero weights.  Epilogue: log10(max(., 1e-10)), 16-byte stores; the
// per-cut normalisation is whisper_norm_kernel (kernel_generic.hpp).
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

struct WhisperParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;  // [400]
  const float* dft_a;   // [14 bin tiles][51 steps][64 lanes]
  const float* mel_a;   // [pairs][4 k-steps][64 lanes]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, shift, M;
  float mel_floor;
  uint32_t mel_mask[kWhBinTiles];   // bit mt set: bin tile bt holds non-zero weights of mel tile mt
  int32_t pair_base[kWhBinTiles];   // index of the first (bt, mt) pair of bin tile bt in mel_a
};

template <int NMT>
__global__ __launch_bounds__(64) void whisper_kernel(const WhisperParams p) {
  __shared__ __attribute__((aligned(16))) float rows[16 * kWhRowStride];
  const int lane = threadIdx.x;
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
  for (int it = lane; it < 16 * 101; it += 64) {
    const int f = it / 101, n = it - f * 101;  // n = 0 .. 100
    const int64_t j = (int64_t)(f0 + f) * p.shift - kWhN / 2;
    float* row = rows + f * kWhRowStride;
    const float y0 = win[n] * load_sample_center(w, j + n, cd.num_samples);
    const float y2 = win[n + 200] * load_sample_center(w, j + n + 200, cd.num_samples);
    const float an = y0 + y2, bn = y0 - y2;
    if (n == 0) {
      row[kWhOffCosE] = an;
      row[kWhOffCosO] = bn;
    } else if (n == 100) {
      row[kWhOffCosE + 100] = an;
      row[kWhOffSinO + 99] = bn;
    } else {
      const float y1 = win[200 - n] * load_sample_center(w, j + 200 - n, cd.num_samples);
      const float y3 = win[400 - n] * load_sample_center(w, j + 400 - n, cd.num_samples);
      const float am = y1 + y3, bm = y1 - y3;
      row[kWhOffCosE + n] = an + am;
      row[kWhOffSinE + n - 1] = an - am;
      row[kWhOffCosO + n] = bn - bm;
      row[kWhOffSinO + n - 1] = bn + bm;
    }
  }
  if (lane < 48) {  // zero the k padding of the even-cos vector (entries 101 .. 103)
    const int f = lane / 3, e = lane - 3 * f;
    rows[f * kWhRowStride + kWhOffCosE + 101 + e] = 0.f;
  } else {  // and the unused last entry of the even-sin vector (its coefficient is 0, but 0 * garbage may be NaN)
    rows[(lane - 48) * kWhRowStride + kWhOffSinE + 99] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- phase 2: DFT GEMMs -> power -> mel GEMM, all on the matrix cores --------------------------------------------
  const int fr = lane & 15, g = lane >> 4;
  f32x4 macc[NMT];
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt) macc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ da = p.dft_a + lane;
  const float* __restrict__ ma = p.mel_a + lane;

#pragma unroll 1
  for (int par = 0; par < 2; ++par) {
    const float* vc = rows + fr * kWhRowStride + (par ? kWhOffCosO : kWhOffCosE) + g;
    const float* vs = rows + fr * kWhRowStride + (par ? kWhOffSinO : kWhOffSinE) + g;
    float bc[kWhCosSteps], bs[kWhSinSteps];
#pragma unroll
    for (int s = 0; s < kWhCosSteps; ++s) bc[s] = vc[4 * s];
#pragma unroll
    for (int s = 0; s < kWhSinSteps; ++s) bs[s] = vs[4 * s];
#pragma unroll 1
    for (int t = 0; t < 7; ++t) {
      const int bt = par * 7 + t;
      const float* __restrict__ a = da + (size_t)bt * kWhSteps * 64;
      f32x4 re = {0.f, 0.f, 0.f, 0.f}, im = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < kWhSinSteps; ++s) {  // two independent accumulation chains, interleaved
        re = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s * 64], bc[s], re, 0, 0, 0);
        im = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(kWhCosSteps + s) * 64], bs[s], im, 0, 0, 0);
      }
      re = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(kWhCosSteps - 1) * 64], bc[kWhCosSteps - 1], re, 0, 0, 0);
      const f32x4 pw = re * re + im * im;  // |X|^2 of bins (tile rows 4g + r), frame fr
      const uint32_t mask = p.mel_mask[bt];
      int pair = p.pair_base[bt];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (mask & (1u << mt)) {  // uniform
          const float* __restrict__ m = ma + (size_t)pair * 256;
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[0], pw.x, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[64], pw.y, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[128], pw.z, macc[mt], 0, 0, 0);
          macc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[192], pw.w, macc[mt], 0, 0, 0);
          ++pair;
        }
      }
    }
  }

  // ---- epilogue: log10(max(mel, floor)); lane holds mels 16 mt + 4 g + r of frame fr --------------------------------
  if (fr < nf) {
    float* orow = p.out + (cd.out_row + f0 + fr) * p.out_stride;
    const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
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
