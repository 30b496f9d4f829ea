// Whisper log-mel, second fast path (frame 400, hop 160): the 400-point real DFT as a mixed-radix FFT on the vector ALUs.
//
// Reference: log_mel_spectrogram (lhotse/features/whisper_fbank.py:62-68).  With y the windowed frame, n = 16 j + l
// (l < 16, j < 25) and k = k2 + 25 k1 (k2 < 25, k1 < 16):
//
//     X[k2 + 25 k1] = sum_l W16^(l k1) * W400^(l k2) * Y_l[k2],      Y_l[k2] = sum_j y[16 j + l] W25^(j k2)
//
// One frame lives on 16 lanes (a DPP row; 4 frames per wave, 16 per workgroup = one matrix-core tile):
//   1. lane l loads its 25 decimated samples straight from global memory (64-byte coalesced per frame; frames touching a
//      cut edge go through torch.stft's reflection rule), times the window (LDS);
//   2. 25-point real DFT in registers, only k2 = 0..12 (the other half is its conjugate): with a_j = y_j + y_(25-j),
//      b_j = y_j - y_(25-j):  Re Y[k2] = y_0 + sum_j a_j cos(2 pi j k2 / 25),  Im Y[k2] = -sum_j b_j sin(2 pi j k2 / 25)
//      -- 288 multiply-adds per lane whose coefficients are wave-uniform and arrive as scalar operands;
//   3. twiddle W400^(l k2) (LDS table), transpose inside the 16-lane group through a wave-private, padded LDS buffer
//      (row stride 34 floats: 8-byte accesses conflict free both ways), so that lane k2 holds Y'_l[k2] for all l;
//   4. fft16 in registers over l -> X[k2 + 25 k1]; |X|^2 goes to the power tile P[16 frames][202]: lane k2 in 1..12
//      delivers the bins k2 + 25 k1 (k1 < 8) and, through X[400 - k] = conj X[k], the bins 25 (16 - k1) - k2
//      (k1 >= 8); lane 0 the bins 25 k1 (k1 <= 8); lanes 13..15 idle in steps 3-4;
//   5. banded mel GEMM on the matrix cores from the power tile (v_mfma_f32_16x16x4_f32, A operands prearranged on the host,
//      the mel tiles are dealt to the four waves by the host so that their k-steps balance), log10, 16-byte stores.
// The per-cut normalisation is whisper_norm_kernel (kernel_generic.hpp), as for the first fast path.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

constexpr int kW2N = 400, kW2Shift = 160;
constexpr int kW2PStride = 202;                 // power tile row stride: == 10 mod 32, every bank is hit exactly twice by a B-operand read
constexpr int kW2TStride = 34;                  // transpose row: 16 complex + 1 complex of padding
constexpr int kW2TFrame = 13 * kW2TStride;      // floats per frame in the transpose buffer
constexpr int kW2TilesPerBlock = 8;             // 128 frames per workgroup
constexpr int kW2MaxMelTiles = 8;               // num_filters <= 128

struct Whisper2Params {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;  // [400]
  const float* cs;      // [12 rows j = 1..12][12 k = 1..12][2]: cos(2 pi j k / 25), -sin(2 pi j k / 25)
  const float* tw;      // [13][16] complex W400^(l k2): row k2, column l
  const float* mel_a;   // [chunks][64 lanes][4 k-steps] A operands of the mel GEMM in lane order (bands padded to whole chunks)
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, M;
  float mel_floor;
  const int32_t* sched;  // [4 waves][2 slots][4]: mel tile (-1: none), first bin (multiple of 4), chunks of 4 k-steps, offset (in chunks)
};

#ifndef HIPFEAT_W2_OCC
#define HIPFEAT_W2_OCC 3  // 4 fits too (128 VGPRs, 36 spilled) and runs at the same speed
#endif
__global__ __launch_bounds__(256, HIPFEAT_W2_OCC) void whisper2_kernel(const Whisper2Params p) {
  // transpose buffer [16 frames][13][34]; once a group has read its frame back, the same region receives the frame's power row
  // (row stride 442 == 26 mod 32: the B-operand reads of the mel GEMM hit every bank exactly twice)
  __shared__ __attribute__((aligned(16))) float tbuf[16 * kW2TFrame + 24];
  HF_POISON_ARRAY(tbuf, 16 * kW2TFrame + 24);
  __shared__ __attribute__((aligned(16))) float winl[kW2N];
  __shared__ __attribute__((aligned(16))) v2 twl[13 * 16];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x;
  int cut, fb;
  if (p.uniform_bpc > 0) {
    cut = blk / p.uniform_bpc;
    fb = blk - cut * p.uniform_bpc;
  } else {
    cut = p.uniform_bpc < 0 ? block_cut_map(p.cuts, p.num_cuts)[blk] : find_cut(p.cuts, p.num_cuts, blk);
    fb = blk - p.cuts[cut].first_block;
  }
  const CutDesc cd = p.cuts[cut];
  const float* __restrict__ w = p.wave + cd.wave_off;
  const int S = cd.num_samples;
  for (int i = tid; i < kW2N; i += 256) winl[i] = p.window[i];
  for (int i = tid; i < 13 * 16; i += 256) twl[i] = reinterpret_cast<const v2*>(p.tw)[i];
  const int q = lane & 15, g = lane >> 4;
  const int fi = 4 * wv + g;  // frame of the tile this 16-lane group owns
  float* tb = tbuf + fi * kW2TFrame;
  float* prow = tb;  // power row of this frame: bins 0 .. 200, then zeros up to 215 (they meet zero weights)
  const int first_tile = fb * kW2TilesPerBlock;
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < kW2TilesPerBlock; ++t) {
    const int f0 = (first_tile + t) * 16;
    if (f0 >= cd.num_frames) break;
    const int nf = min(16, cd.num_frames - f0);

    // ---- 1. decimated, windowed samples of this group's frame: lane l takes the samples 16 j + l (64-byte coalesced) -------
    float s[25];
    {
      const int64_t j0 = (int64_t)(f0 + fi) * kW2Shift - kW2N / 2;
      const bool live = fi < nf;
      const bool inside = j0 >= 0 && j0 + kW2N <= (int64_t)S;
#pragma unroll
      for (int j = 0; j < 25; ++j) {
        const int idx = 16 * j + q;
        float x = 0.f;
        if (live) x = inside ? w[j0 + idx] : load_sample_center(w, j0 + idx, S);
        s[j] = x * winl[idx];
      }
    }
    // ---- 2. 25-point real DFT, k2 = 0 .. 12 -----------------------------------------------------------------------------
    v2 Y[13];
    {
      float a[12], b[12];
      float sum = s[0];
#pragma unroll
      for (int j = 1; j <= 12; ++j) {
        a[j - 1] = s[j] + s[25 - j];
        b[j - 1] = s[j] - s[25 - j];
        sum += a[j - 1];
      }
      Y[0] = v2{sum, 0.f};
      // j outermost: the 12 (cos, -sin) coefficient pairs of one j (consecutive in memory, fetched through the scalar cache and used as
      // SGPR-pair operands of v_pk_fma_f32) update 12 independent (Re, Im) accumulators -- no dependent chains.
      // The opaque pointer keeps hipcc from hoisting all 288 loop-invariant loads out of the tile loop (they do not fit the
      // SGPR file and would come back as spilled VGPRs).
#pragma unroll
      for (int k = 0; k < 12; ++k) Y[1 + k] = v2{s[0], 0.f};
#pragma unroll
      for (int j = 1; j <= 12; ++j) {
        // constant address space: the loads stay scalar (s_load) after the opaque copy
        const __attribute__((address_space(4))) float* cj = (const __attribute__((address_space(4))) float*)(p.cs) + (j - 1) * 24;
        asm volatile("" : "+s"(cj), "+v"(Y[1]));  // ... and after the previous j: at most two rows of coefficients in flight
        const v2 ab = v2{a[j - 1], b[j - 1]};
#pragma unroll
        for (int k = 0; k < 12; ++k) Y[1 + k] = ab * v2{cj[2 * k], cj[2 * k + 1]} + Y[1 + k];
      }
    }
    // ---- 3. twiddle, transpose inside the group ---------------------------------------------------------------------------
    *reinterpret_cast<v2*>(tb + 2 * q) = Y[0];
#pragma unroll
    for (int k = 1; k <= 12; ++k) *reinterpret_cast<v2*>(tb + k * kW2TStride + 2 * q) = cmul(Y[k], twl[k * 16 + q]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v2 xin[16], X[16];
    {
      const float* src = tb + min(q, 12) * kW2TStride;
#pragma unroll
      for (int l = 0; l < 16; ++l) xin[l] = *reinterpret_cast<const v2*>(src + 2 * l);
    }
    // ---- 4. 16-point FFT over l, power, scatter into the power tile ---------------------------------------------------------
    fft16(xin, X);
    prow[201 + q] = 0.f;  // the last 16-bin chunk of a band may reach up to 15 entries past bin 200 (zero weights): keep them finite
    if (q == 0) {
#pragma unroll
      for (int k1 = 0; k1 <= 8; ++k1) prow[25 * k1] = X[k1].x * X[k1].x + X[k1].y * X[k1].y;
    } else if (q <= 12) {
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) {
        const int bin = k1 < 8 ? q + 25 * k1 : 25 * (16 - k1) - q;
        prow[bin] = X[k1].x * X[k1].x + X[k1].y * X[k1].y;
      }
    }
    __syncthreads();  // the power rows are complete

    // ---- 5. banded mel GEMM: D[mel][frame] += W[mel][bin] P[bin][frame] ---------------------------------------------------------
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
      const int4 sc = *reinterpret_cast<const int4*>(p.sched + (wv * 2 + slot) * 4);  // uniform: one scalar load
      const int mt = sc.x;
      if (mt < 0) continue;
      const int k0 = sc.y, chunks = sc.z;
      const f32x4* __restrict__ ma = reinterpret_cast<const f32x4*>(p.mel_a) + (size_t)sc.w * 64 + lane;
      const float* pb = tbuf + q * kW2TFrame + k0 + g;  // B operand: frame q, bin k0 + 4 s + g
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < chunks; ++i) {
        const f32x4 a4 = ma[(size_t)i * 64];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, pb[16 * i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, pb[16 * i + 4], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, pb[16 * i + 8], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, pb[16 * i + 12], acc1, 0, 0, 0);
      }
      const f32x4 acc = acc0 + acc1;
      // lane holds mels 16 mt + 4 g + r of frame q
      if (q < nf) {
        float* orow = p.out + (cd.out_row + f0 + q) * p.out_stride;
        const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
        const int m0 = 16 * mt + 4 * g;
        f32x4 v;
        v.x = fast_log(fmaxf(acc.x, p.mel_floor)) * 0.4342944819032518f;
        v.y = fast_log(fmaxf(acc.y, p.mel_floor)) * 0.4342944819032518f;
        v.z = fast_log(fmaxf(acc.z, p.mel_floor)) * 0.4342944819032518f;
        v.w = fast_log(fmaxf(acc.w, p.mel_floor)) * 0.4342944819032518f;
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
    __syncthreads();  // every B operand has been read: the power tile may be overwritten
  }
}

}  // namespace hipfeat
