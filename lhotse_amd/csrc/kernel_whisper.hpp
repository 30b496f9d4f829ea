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
// One workgroup (2 waves) walks over kWhTilesPerBlock consecutive tiles of 16 frames of one cut.  Per tile: the sample
// span (15*160 + 400 floats) arrives in LDS by LDS-DMA (global_load_lds_dwordx4), requested one tile ahead so the HBM
// latency hides behind the previous tile's matrix-core phase; each wave turns 8 frames of it into their four
// symmetrised vectors (LDS rows); then wave 0 takes the even bins and wave 1 the odd bins: it keeps the B
// operands (the 16 frames' vectors of its parity) in registers and streams the cos/sin matrices (A operands,
// precomputed on the host in MFMA lane order, L2 resident, double buffered in registers) through
// v_mfma_f32_16x16x4_f32: D[bin][frame] += C[bin][n] * V[n][frame].  The power |X|^2 of a 16-bin tile is formed in
// the accumulator registers, which ARE the B operand layout of the next GEMM (k order = accumulator row order), so the
// mel filterbank follows as further MFMAs without touching LDS: D[mel][frame] += W[mel][bin] * P[bin][frame], only for
// the (bin tile, mel tile) pairs that hold non-zero weights.  The two waves exchange their partial mel sums through LDS
// (each finishes half of the mel tiles).  Epilogue: log10(max(., 1e-10)), 16-byte stores; the per-cut normalisation is
// whisper_norm_kernel (kernel_generic.hpp).  38 KB LDS and ~250 registers per lane -> 4 workgroups = 8 waves per CU.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

constexpr int kWhN = 400;
constexpr int kWhBinTiles = 14;   // 7 tiles of 16 even bins (2m, m <= 100), then 7 tiles of 16 odd bins (2m+1, m <= 99)
constexpr int kWhCosSteps = 26;   // k-steps of 4 over the cos vectors (101 / 100 entries)
constexpr int kWhSinSteps = 25;   // k-steps of 4 over the sin vectors (99 / 100 entries)
constexpr int kWhSteps = kWhCosSteps + kWhSinSteps;
constexpr int kWhChunks = (kWhSteps + 3) / 4;  // 16-byte operand loads: 4 k-steps per lane per load
constexpr int kWhRowStride = 420;  // floats per frame row in LDS (== 4 mod 32: conflict-free operand reads)
constexpr int kWhOffCosE = 0, kWhOffSinE = 104, kWhOffCosO = 204, kWhOffSinO = 304;
constexpr int kWhMaxMelTiles = 8;  // num_filters <= 128
constexpr int kWhTilesPerBlock = 8;  // 128 frames per workgroup
constexpr int kWhShift = 160;       // the fast path is specialised for Whisper's hop
constexpr int kWhSpan = 15 * kWhShift + kWhN;  // 2800 samples per tile
constexpr int kWhSpanChunks = (kWhSpan + 255) / 256;  // 1 KiB LDS-DMA chunks
constexpr int kWhSlots = 4;        // mel tiles a 16-bin tile may feed (consecutive); mel_a holds kWhSlots per bin tile

struct WhisperParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;  // [400]
  const float* dft_a;   // [14 bin tiles][13 chunks][64 lanes][4 k-steps]  (step 51 is padding)
  const float* mel_a;   // [14 bin tiles][kWhSlots][64 lanes][4 k-steps]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, shift, M;
  float mel_floor;
  int32_t ablate;  // experiments only (HIPFEAT_WH_ABLATE): 1 skips phase 1, 2 skips phase 2
  int32_t mt_lo[kWhBinTiles];    // first mel tile with non-zero weights in bin tile bt
  int32_t mt_cnt[kWhBinTiles];   // number of consecutive mel tiles from mt_lo (<= kWhSlots)
};

template <int NMT>
__global__ __launch_bounds__(128, 2) void whisper_kernel(const WhisperParams p) {
  __shared__ __attribute__((aligned(16))) float rows[16 * kWhRowStride];
  __shared__ __attribute__((aligned(16))) float xs[kWhSpanChunks * 256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const float* __restrict__ w = p.wave + cd.wave_off;
  const int S = cd.num_samples;
  const int fr = lane & 15, g = lane >> 4;
  const f32x4* __restrict__ da = reinterpret_cast<const f32x4*>(p.dft_a) + lane;
  const f32x4* __restrict__ ma = reinterpret_cast<const f32x4*>(p.mel_a) + lane;
  const int bt0 = 7 * wv;  // wave 0: bin tiles 0 .. 6 (even bins), wave 1: 7 .. 13 (odd bins)

  // window taps of this lane's two n values (n = lane, 64 + lane <= 100): w[n], w[n+200], w[200-n], w[400-n]
  float wn[2][4];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int n = min(64 * half + lane, 100);
    wn[half][0] = p.window[n], wn[half][1] = p.window[n + 200], wn[half][2] = p.window[200 - n], wn[half][3] = p.window[(400 - n) % 400];
  }

  // Stage the sample span of the tile starting at frame f0 into xs.  Interior tiles: LDS-DMA (1 KiB per wave
  // instruction, dword-aligned global addresses are enough).  Tiles touching a cut edge: scalar loads with the
  // torch.stft "reflect" rule.
  auto stage_span = [&](int f0) {
    const int jt = f0 * kWhShift - kWhN / 2;
    if (jt >= 0 && jt + kWhSpanChunks * 256 <= S) {
      const char* src = reinterpret_cast<const char*>(w + jt);
      for (int ch = wv; ch < kWhSpanChunks; ch += 2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)ch * 1024u + (unsigned)lane * 16u)),
                                         (__attribute__((address_space(3))) void*)(xs + ch * 256), 16, 0, 0);
    } else {
      for (int i = tid; i < kWhSpan; i += 128) xs[i] = load_sample_center(w, (int64_t)jt + i, S);
    }
  };

  float bc[kWhCosSteps], bs[kWhSinSteps];
  auto load_b = [&](int par) {
    const float* vc = rows + fr * kWhRowStride + (par ? kWhOffCosO : kWhOffCosE) + g;
    const float* vs = rows + fr * kWhRowStride + (par ? kWhOffSinO : kWhOffSinE) + g;
#pragma unroll
    for (int s = 0; s < kWhCosSteps; ++s) bc[s] = vc[4 * s];
#pragma unroll
    for (int s = 0; s < kWhSinSteps; ++s) bs[s] = vs[4 * s];
  };
  // one global_load_dwordx4 per 4 k-steps
  int opaque0 = 0;  // re-defined (opaquely) every tile iteration: keeps LICM from hoisting the whole operand stream
  auto load_a = [&](int bt, float (&A)[kWhChunks * 4], float (&MA)[kWhSlots * 4]) {
    bt += opaque0;
    const f32x4* __restrict__ a = da + (size_t)bt * kWhChunks * 64;
#pragma unroll
    for (int c = 0; c < kWhChunks; ++c) {
      const f32x4 v = a[c * 64];
      A[4 * c] = v.x, A[4 * c + 1] = v.y, A[4 * c + 2] = v.z, A[4 * c + 3] = v.w;
    }
    const f32x4* __restrict__ m = ma + (size_t)bt * kWhSlots * 64;
#pragma unroll
    for (int i = 0; i < kWhSlots; ++i) {
      const f32x4 v = m[i * 64];
      MA[4 * i] = v.x, MA[4 * i + 1] = v.y, MA[4 * i + 2] = v.z, MA[4 * i + 3] = v.w;
    }
  };

  const int first_tile = fb * kWhTilesPerBlock;
  if (first_tile * 16 < cd.num_frames) stage_span(first_tile * 16);

#pragma unroll 1
  for (int t = 0; t < kWhTilesPerBlock; ++t) {
    const int f0 = (first_tile + t) * 16;
    if (f0 >= cd.num_frames) break;
    const int nf = min(16, cd.num_frames - f0);
    asm volatile("" : "+s"(opaque0));
    f32x4 macc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) macc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: span (LDS) -> windowed, symmetrised vectors of 8 frames per wave (LDS rows) ----------------------
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's DMA chunks (and the previous tile's stores) are done
    __syncthreads();                      // everyone's chunks are in; the previous tile's exchange buffer is dead
    if (!(p.ablate & 1)) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int n = 64 * half + lane;
        if (n <= 100) {
#pragma unroll 4
          for (int ff = 0; ff < 8; ++ff) {
            const int f = 8 * wv + ff;
            const float* x = xs + f * kWhShift;
            float* row = rows + f * kWhRowStride;
            const float y0 = wn[half][0] * x[n];
            const float y2 = wn[half][1] * x[n + 200];
            const float y1 = wn[half][2] * x[200 - n];
            const float y3 = wn[half][3] * x[(400 - n) % 400];  // n == 0: unused
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
    }
    if (wv != 0) {
    } else if (lane < 48) {  // zero the k padding of the even-cos vector (entries 101 .. 103)
      const int f = lane / 3, e = lane - 3 * f;
      rows[f * kWhRowStride + kWhOffCosE + 101 + e] = 0.f;
    } else {  // and the unused last entry of the even-sin vector (its coefficient is 0, but 0 * garbage may be NaN)
      rows[(lane - 48) * kWhRowStride + kWhOffSinE + 99] = 0.f;
    }
    __syncthreads();  // rows complete; the span buffer is dead

    // ---- phase 2: DFT GEMMs -> power -> mel GEMM on the matrix cores -----------------------------------------------
    // The first two operand tiles are requested BEFORE the next span's DMA: gfx950 retires vector-memory operations
    // in order, so anything queued behind the DMA waits for HBM.
    float A0[kWhChunks * 4], A1[kWhChunks * 4], MA0[kWhSlots * 4], MA1[kWhSlots * 4];
    load_a(bt0, A0, MA0);
    load_a(bt0 + 1, A1, MA1);
    load_b(wv);
    {
      const int fn = f0 + 16;
      if (t + 1 < kWhTilesPerBlock && fn < cd.num_frames) stage_span(fn);
    }
    auto compute = [&](int bt, const float (&A)[kWhChunks * 4], const float (&MA)[kWhSlots * 4]) {
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
    if (!(p.ablate & 2)) {
      compute(bt0 + 0, A0, MA0);
      load_a(bt0 + 2, A0, MA0);
      compute(bt0 + 1, A1, MA1);
      load_a(bt0 + 3, A1, MA1);
      compute(bt0 + 2, A0, MA0);
      load_a(bt0 + 4, A0, MA0);
      compute(bt0 + 3, A1, MA1);
      load_a(bt0 + 5, A1, MA1);
      compute(bt0 + 4, A0, MA0);
      load_a(bt0 + 6, A0, MA0);
      compute(bt0 + 5, A1, MA1);
      compute(bt0 + 6, A0, MA0);
    }
    // exchange partial mel sums: a wave finishes the mel tiles of its own parity (mt & 1 == wv), hands over the others
    __syncthreads();  // every B operand has been read: the rows are dead
    float* xch = rows;  // [NMT][4][64]
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

    // ---- epilogue: log10(max(mel, floor)); lane holds mels 16 mt + 4 g + r of frame fr ------------------------------
    if (fr < nf) {
      float* orow = p.out + (cd.out_row + f0 + fr) * p.out_stride;
      const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if ((mt & 1) != wv) continue;  // uniform
        const int m0 = 16 * mt + 4 * g;
        f32x4 v;
        v.x = fast_log(fmaxf(macc[mt].x, p.mel_floor)) * 0.4342944819032518f;
        v.y = fast_log(fmaxf(macc[mt].y, p.mel_floor)) * 0.4342944819032518f;
        v.z = fast_log(fmaxf(macc[mt].z, p.mel_floor)) * 0.4342944819032518f;
        v.w = fast_log(fmaxf(macc[mt].w, p.mel_floor)) * 0.4342944819032518f;
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
    // the loop-top wait + barrier separates this tile's exchange reads from the next tile's row writes
  }
}

}  // namespace hipfeat
