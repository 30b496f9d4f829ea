// fft2048 fast path, wave-autonomous (44.1 / 48 kHz Kaldi log-mel filterbanks, Wav2LogFilterBank layers.py:565-578, and librosa-style
// log-mel with n_fft 2048): the organisation of kernel_fft512c.hpp / kernel_fft1024c.hpp -- a wave owns its frames from the samples in
// HBM to the stored log-mel rows, private LDS-DMA span one round ahead, no workgroup barrier in the steady state -- with 32 lanes per
// frame and two frames per wave.
//
//   real FFT(2048) = complex FFT(1024) of z[n] = y[2n] + i y[2n+1], 1024 = 32 x 32 on 32 lanes per frame:
//     pass 1   lane q holds z[32 n1 + q], n1 = 0..31 (rows >= ceil(N/64) are zero padding) -> 32-point FFT in registers, times
//              W_1024^(q k1) (table in global memory, L1-resident: the only constants that do not fit the LDS budget of 8 waves);
//     exchange through the wave's LDS region in two halves of 17 rows: rows k1 = 0..16, then 16..31 and row 0 again;
//     pass 2   a 32-point FFT per row, split between TWO lanes: lane l computes the EVEN outputs (k2 = 2 j) of row l and the ODD
//              outputs (k2 = 2 j + 1) of the mirror row (32 - l) % 32 -- one radix-2 butterfly stage + a 16-point FFT each, one per
//              half: lanes <= 16 find their own row in the first half and the mirror row in the second, lanes >= 17 the other way
//              round.  With R0 / R1 the results of the two halves, R0[s] and R1[15 - s] are mirror bins (k, 1024 - k) in EVERY lane:
//     split    X[k] = E[k] + W_2048^k O[k] on bin pairs (k, 1024 - k) without any cross-lane traffic.  Lane 0 is the exception
//              (row 0 is its own mirror: even pairs with even, odd with odd: 17 pairs instead of 16); it follows the common schedule
//              through per-step register selects and one extra step, exactly as in kernel_fft1024c.hpp;
//     |X|^2 (or |X|) -> two power rows of 1040 floats in the same LDS region.
//   mel filterbank on the matrix cores with v_mfma_f32_4x4x1_16B_f32 as in kernel_fft512c.hpp (mel4_schedule.hpp), up to four
//   accumulator sets of up to 64 steps; of the four frame rows of a block only two carry frames (the other two repeat them).
#pragma once
#include "common.hpp"
#include "fft_common.hpp"
#include "kernel_fft1024c.hpp"  // sel64; HFC_SEP, mul24, phase-timer macros via kernel_fft512c.hpp

namespace hipfeat {

constexpr int kXExRowStride = 66;                        // dwords per exchange row (32 complex + 2 pad: rows 2 banks apart)
constexpr int kXExRows = 17;                             // rows per exchange half
constexpr int kXExFrameStride = kXExRows * kXExRowStride;  // 1122
constexpr int kXPRowStride = 1040;                       // dwords per power row (1025 bins + pad; == 16 mod 64)
constexpr int kXRegion = 2 * kXExFrameStride;            // 2244 dwords per wave; the 2 power rows (2080) alias it
constexpr int kXMaxSets = 4;                             // accumulator sets (16 slots each)
constexpr int kXMaxSteps = 64;                           // MFMA steps per set
constexpr int kXMaxWaves = 8;                            // waves per workgroup (fewer when the span buffers are long)
constexpr int kXSplitSteps = 17;                         // bin-pair steps of the split (16 + lane 0's extra one)

struct Fft2048cParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  // shared LDS image, copied once per workgroup: [nrows][32] v2 window/2 | [17][32] v2 split twiddles -i W_2048^k(step, lane) |
  // [2][16] v2 butterfly twiddles of pass 2 (ones | W_32^n) | weight table [total steps / 4][64 lanes][4 steps] | lane table [sets][64 lanes][4]
  const float* shared_consts;
  const float* twp;  // [32][32] v2 W_1024^(q k1), row k1, column q (global memory)
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc;
  int32_t frames_per_block, rounds, waves;  // rounds of 2 frames per wave; frames_per_block = waves * rounds * 2
  int32_t N, shift, npad_left, M, flags;
  float preemph, mel_floor;
  int32_t shared_floats, tws_off, tw32_off, wtab_off, ltab_off;
  int32_t xs_floats;  // floats of one wave's sample-span buffer (multiple of 4)
  int32_t nsets, steps[kXMaxSets], step0[kXMaxSets];  // accumulator sets: MFMA steps (multiples of 4) and first step in the weight table
};

// One half of the 32-point FFT of a row: u[n] = (x[n] + sg x[n + 16]) t[n], then a 16-point FFT.  sg = +1, t = 1: the even outputs
// X[2 j]; sg = -1, t[n] = W_32^n: the odd outputs X[2 j + 1].  Which of the two a lane computes is data (sg, the table row), not code.
__device__ __forceinline__ void half_fft32(const v2 (&x)[32], v2 sg, const v2* t, v2 (&out)[16]) {
  v2 u[16], tw[16];
#pragma unroll
  for (int n = 1; n < 16; ++n) {
    tw[n] = t[n];
    HFC_SEP();
  }
#pragma unroll
  for (int n = 0; n < 16; ++n) u[n] = x[n + 16] * sg + x[n];
#pragma unroll
  for (int n = 1; n < 16; ++n) u[n] = cmul2(u[n], tw[n]);
  fft16(u, out);
}

// S0 / S1 / S2 != 0: the step counts of a three-set mel schedule as compile-time constants (the 80-filter Kaldi default at 44.1 / 48 kHz:
// 52, 28, 16), as in kernel_fft1024c.hpp; any other filterbank runs the generic <NROWS, ODD, 0, 0, 0>.
// W12: 12 waves per workgroup = 3 waves/SIMD without a span prefetch, the span buffer aliasing the exchange / power region, as in
// kernel_fft1024c.hpp; the pass twiddles are then requested AFTER the 32-point FFT, in two bursts (133-147 VGPRs).  Same-box A/B against the
// 8-wave fixed-schedule instances: 44.1 kHz (odd hop) + 4 %, 48 kHz - 3 % -- so only the 44.1 kHz default uses it.
constexpr int kXWavesFixed = 12;
template <int NROWS, bool ODD, int S0 = 0, int S1 = 0, int S2 = 0, bool W12 = false>
__global__ __launch_bounds__(64 * (W12 ? kXWavesFixed : kXMaxWaves), (W12 ? 3 : 2)) void fft2048c_kernel(const Fft2048cParams p) {
  constexpr bool kFixed = S0 != 0;
  constexpr bool kPrefetch = !W12;  // span of round r + 1 requested during round r (needs a span buffer of its own)
  constexpr bool kTwLate = NROWS == 32 || !kPrefetch;  // pass twiddles requested after the 32-point FFT
  constexpr int kFixSteps[4] = {S0, S1, S2, 0}, kFixStep0[4] = {0, S0, S0 + S1, S0 + S1 + S2};
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  const v2* cwin = reinterpret_cast<const v2*>(smem);                 // [NROWS][32]
  const v2* ctws = reinterpret_cast<const v2*>(smem + p.tws_off);     // [17][32] split twiddles per (step, lane)
  const v2* ctw32 = reinterpret_cast<const v2*>(smem + p.tw32_off);   // [2][16]
  const float* wtab = smem + p.wtab_off;
  const float* ltab = smem + p.ltab_off;

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
  // round 5, experiment 2: the fixed-schedule instances are the 80-filter Kaldi defaults at 44.1 kHz (ODD: 1102-sample frames, hop 441)
  // and 48 kHz (1200 / 480) and nothing else (hipfeat.hip selects them on exactly these numbers): frame geometry, span length, waves
  // per workgroup and filter count as compile-time constants free ~20 scalar registers (the kernel spills 44) and the scalar
  // arithmetic that derives addresses from them every round
#ifdef HIPFEAT_ABL_RUNTIME_GEOMETRY
  constexpr bool kGeo = false;
#else
  constexpr bool kGeo = kFixed;
#endif
  const int N = kGeo ? (ODD ? 1102 : 1200) : p.N, shift = kGeo ? (ODD ? 441 : 480) : p.shift;
  const int npad_left = kGeo ? (ODD ? 330 : 360) : p.npad_left;
  const int xs_floats = kGeo ? (((ODD ? 441 : 480) + 64 * NROWS + 3) & ~3) : p.xs_floats;
  const int nwaves = kGeo ? (W12 ? kXWavesFixed : kXMaxWaves) : p.waves;
  const int M = kGeo ? 80 : p.M;

  for (int i = tid; i < p.shared_floats; i += 64 * nwaves) smem[i] = p.shared_consts[i];
  float* xs = smem + p.shared_floats + wv * (kPrefetch ? xs_floats + kXRegion : kXRegion);
  float* myreg = kPrefetch ? xs + xs_floats : xs;  // (no prefetch: the span is dead once the samples are in registers, before the exchange)
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  // the fixed-schedule instances serve Kaldi plans only (hipfeat.hip: never librosa): |X|^2, natural log and Kaldi's edge rule are
  // compile-time facts there -- 17 uniform branches around v_sqrt per round less (round 5, experiment 1)
#ifdef HIPFEAT_ABL_RUNTIME_MAG
  const bool mag = (p.flags & F_FFT_MAG) != 0;
#else
  const bool mag = kFixed ? false : (p.flags & F_FFT_MAG) != 0;
#endif
  const float log_scale = (!kFixed && (p.flags & F_LOG10)) ? 0.30102999566398120f : 0.69314718055994531f;  // log2 -> log10 / ln
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;

  auto stage_span = [&](int f0, unsigned lane4) {
    const int64_t j0 = (int64_t)f0 * shift - npad_left;
    if (j0 >= 0 && j0 + xs_floats <= cd.num_samples) {
      const char* src = reinterpret_cast<const char*>(w + j0);  // uniform
      const int nfull = xs_floats >> 8;
#pragma unroll
      for (int ch = 0; ch < 10; ++ch) {
        if (ch < nfull)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)ch * 1024u + 4u * lane4)),
                                           (__attribute__((address_space(3))) void*)(xs + ch * 256), 16, 0, 0);
      }
      if ((unsigned)nfull * 256u + lane4 < (unsigned)xs_floats)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)nfull * 1024u + 4u * lane4)),
                                         (__attribute__((address_space(3))) void*)(xs + nfull * 256), 16, 0, 0);
    } else {
      if (!kFixed && (p.flags & F_CENTER))  // torch.stft / librosa "reflect" padding (the edge sample is not repeated)
        for (int i = (int)(lane4 >> 2); i < xs_floats; i += 64) xs[i] = load_sample_center(w, j0 + i, cd.num_samples);
      else
        for (int i = (int)(lane4 >> 2); i < xs_floats; i += 64) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
    }
  };

  const int first_frame = fb * p.frames_per_block + 2 * wv;  // the waves take the frame pairs round-robin
  __syncthreads();  // the constant tables are in place (the only workgroup barrier of the kernel)
  if (kPrefetch && first_frame < cd.num_frames) stage_span(first_frame, (unsigned)lane * 4u);

#ifdef HIPFEAT_PHASE_TIMERS
  unsigned long long hfc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hfc_last = __builtin_readcyclecounter();
#endif
  for (int r = 0; r < p.rounds; ++r) {
    const int f0 = first_frame + 2 * nwaves * r;
    if (f0 >= cd.num_frames) break;
    const int nf = min(2, cd.num_frames - f0);

    int lane_o = lane;  // opaque copy: keeps LICM from pinning per-lane addresses in VGPRs for the whole kernel
    asm volatile("" : "+v"(lane_o));
    if (!kPrefetch) stage_span(f0, (unsigned)lane_o * 4u);  // (the previous round's power-row reads were consumed by its MFMAs: the region is free)
    if (r == 0 || !kPrefetch) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); with prefetch, later rounds waited before their predecessor's stores
    const int q = lane_o & 31, g = lane_o >> 5;
    const unsigned long long q0 = __builtin_amdgcn_ballot_w64(q == 0);  // lanes 0 and 32

    v2 R0[16], R1[16];
    {
      v2 a[32];
      {
        const float* x = xs + mul24(g, shift) + 2 * q;
        v2 z[32];
        // with all 32 input rows live the window (64 more VGPRs) put 8 registers into scratch (VERDICT r5 task 7): there it is fetched in
        // two bursts of 16 rows after the mean, in front of the rows it multiplies -- same products, bit-identical
        constexpr bool kWinLate = NROWS == 32;
        v2 win[kWinLate ? 16 : NROWS];
        float pv[NROWS];  // left neighbour of each pair's first sample (the frame's first sample replicates itself, layers.py:166)
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          if (ODD) z[n1] = v2{x[64 * n1], x[64 * n1 + 1]};  // odd hop: the second frame's pairs sit on odd float offsets
          else z[n1] = *reinterpret_cast<const v2*>(x + 64 * n1);
          HFC_SEP();
        }
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) pv[n1] = n1 == 0 ? x[q == 0 ? 0 : -1] : x[64 * n1 - 1];
        if (!kWinLate) {
#pragma unroll
          for (int n1 = 0; n1 < NROWS; ++n1) {
            win[n1] = cwin[n1 * 32 + q];
            HFC_SEP();
          }
        }
        // the pass twiddles W_1024^(q k1) come from global memory (L1): requested BEFORE the next span, so that the wait for them
        // (vmcnt is in order) does not include the span's trip to HBM
        // (with all 32 input rows live -- NROWS == 32 -- that request would spill: there both the twiddle request and, after it, the
        // span request follow the 32-point FFT)
        v2 twp[32];
        const v2* gt = reinterpret_cast<const v2*>(p.twp) + q;
        if (!kTwLate) {
#pragma unroll
          for (int k1 = 1; k1 < 32; ++k1) twp[k1] = gt[k1 * 32];
        }
        // the samples are in flight to registers; once they have arrived the buffer is free for the next round's span
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        HFC_T(0);  // sample, neighbour and window reads
        if (kPrefetch && NROWS < 32 && r + 1 < p.rounds && f0 + 2 * nwaves < cd.num_frames) stage_span(f0 + 2 * nwaves, (unsigned)lane_o * 4u);
        HFC_T(1);  // span request (LDS-DMA issue)

#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          if (64 * (n1 + 1) > N) {  // samples at or beyond N are not part of the frame
            const int m0 = 64 * n1 + 2 * q;
            if (m0 >= N) z[n1].x = 0.f;
            if (m0 + 1 >= N) z[n1].y = 0.f;
          }
        }
        float mu = 0.f;
        if (dc) {
          v2 sa = z[0], sb = z[1], sc = z[2], sd = z[3];
#pragma unroll
          for (int n1 = 4; n1 < NROWS; ++n1) {
            if ((n1 & 3) == 0) sa += z[n1];
            if ((n1 & 3) == 1) sb += z[n1];
            if ((n1 & 3) == 2) sc += z[n1];
            if ((n1 & 3) == 3) sd += z[n1];
          }
          const v2 sum2 = (sa + sb) + (sc + sd);
          float t = row16_sum(sum2.x + sum2.y);
          t += __shfl_xor(t, 16, 64);  // the frame's other DPP row
          mu = t * inv_n;
        }
        // y[n] = (x[n] - mu) - c (x[n-1] - mu) = x[n] - c x[n-1] - (1 - c) mu, times the window
        {
          const float nc = -c, mu1 = (1.0f - c) * mu;
          if (kWinLate) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                win[i] = cwin[(16 * h + i) * 32 + q];
                HFC_SEP();
              }
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                const int n1 = 16 * h + i;
                v2 t;
                t.x = fmaf(nc, pv[n1 < NROWS ? n1 : 0], z[n1].x);
                t.y = fmaf(nc, z[n1].x, z[n1].y);
                z[n1] = (t - v2{mu1, mu1}) * win[i];
              }
            }
          } else {
#pragma unroll
            for (int n1 = 0; n1 < NROWS; ++n1) {
              v2 t;
              t.x = fmaf(nc, pv[n1], z[n1].x);
              t.y = fmaf(nc, z[n1].x, z[n1].y);
              z[n1] = (t - v2{mu1, mu1}) * win[n1];
            }
          }
        }
#pragma unroll
        for (int n1 = NROWS; n1 < 32; ++n1) z[n1] = v2{0.f, 0.f};
        fft32<NROWS>(z, a);
        if (kTwLate) {  // two bursts of 16: request, multiply, request, multiply
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            asm volatile("" : "+v"(a[16 * h].x), "+v"(a[16 * h + 15].y) : : "memory");
#pragma unroll
            for (int k1 = 16 * h + (h == 0 ? 1 : 0); k1 < 16 * h + 16; ++k1) twp[k1] = gt[k1 * 32];
#pragma unroll
            for (int k1 = 16 * h + (h == 0 ? 1 : 0); k1 < 16 * h + 16; ++k1) a[k1] = cmul2(a[k1], twp[k1]);
          }
          if (kPrefetch && r + 1 < p.rounds && f0 + 2 * nwaves < cd.num_frames) stage_span(f0 + 2 * nwaves, (unsigned)lane_o * 4u);
        } else {
#pragma unroll
          for (int k1 = 1; k1 < 32; ++k1) a[k1] = cmul2(a[k1], twp[k1]);
        }
      }
      HFC_T(2);  // mean, prolog, pass 1, twiddles
      float* exf = myreg + mul24(g, kXExFrameStride);
      // ---- half 0: rows 0..16.  Lanes <= 16 read their own row (even outputs), lanes >= 17 the mirror row 32 - l (odd outputs) ----
      {
#pragma unroll
        for (int rr = 0; rr < kXExRows; ++rr) *reinterpret_cast<v2*>(exf + rr * kXExRowStride + 2 * q) = a[rr];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool odd = q >= 17;
        const float* src = exf + mul24(odd ? 32 - q : q, kXExRowStride);
        v2 x32[32];
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) {
          x32[n2] = *reinterpret_cast<const v2*>(src + 2 * n2);
          HFC_SEP();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float sgn = odd ? -1.0f : 1.0f;
        half_fft32(x32, v2{sgn, sgn}, ctw32 + (odd ? 16 : 0), R0);
      }
      // ---- half 1: rows 16..31 in slots 0..15, row 0 in slot 16.  Lanes <= 16: mirror row (odd outputs); lanes >= 17: own row ----
      {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) *reinterpret_cast<v2*>(exf + rr * kXExRowStride + 2 * q) = a[16 + rr];
        *reinterpret_cast<v2*>(exf + 16 * kXExRowStride + 2 * q) = a[0];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool odd = q <= 16;
        const int slot = q == 0 ? 16 : (q <= 16 ? 16 - q : q - 16);
        const float* src = exf + mul24(slot, kXExRowStride);
        v2 x32[32];
#pragma unroll
        for (int n2 = 0; n2 < 32; ++n2) {
          x32[n2] = *reinterpret_cast<const v2*>(src + 2 * n2);
          HFC_SEP();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float sgn = odd ? -1.0f : 1.0f;
        half_fft32(x32, v2{sgn, sgn}, ctw32 + (odd ? 16 : 0), R1);
      }
      HFC_T(3);  // exchange + pass 2
    }

    {
      float* prow = myreg + mul24(g, kXPRowStride);
      if (q < 15) prow[1025 + q] = 0.f;  // the padding a slot may read past bin 1024 (weight 0) must be finite
      // R0[s] is bin kb + 64 s with kb = l (lanes <= 16: even outputs of row l) or 64 - l (lanes >= 17: odd outputs of row 32 - l);
      // R1[15 - s] is its mirror 1024 - kb - 64 s.  Lane 0: R0 = bins 64 j, R1 = bins 32 + 64 j; steps 9..15 and 16 pair R1 with itself.
      const int kb = q <= 16 ? q : 64 - q;
      float* pA = prow + kb;
      float* pB = prow + 1024 - kb;
      float* pA2 = prow + __builtin_bit_cast(int, sel64(q0, __builtin_bit_cast(float, -544), __builtin_bit_cast(float, kb)));
      float* pB2 = prow + __builtin_bit_cast(int, sel64(q0, __builtin_bit_cast(float, 1568), __builtin_bit_cast(float, 1024 - kb)));
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        v2 tw[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          tw[rr] = ctws[(4 * h + rr) * 32 + q];
          HFC_SEP();
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int s = 4 * h + rr;
          v2 zk, m;
          if (s <= 8) {
            zk = R0[s];
            m = sel64(q0, R0[(16 - s) & 15], R1[15 - s]);
          } else {
            zk = sel64(q0, R1[s - 9], R0[s]);
            m = sel64(q0, R1[24 - s], R1[15 - s]);
          }
          const v2 sp = m * HF_CJ + zk;
          const v2 dm = m * HF_NCJ + zk;
          const v2 tt = cmul2(dm, tw[rr]);
          const v2 xp = sp + tt, xm = sp - tt;
          float va = xp.x * xp.x + xp.y * xp.y, vb = xm.x * xm.x + xm.y * xm.y;
          if (mag) va = __builtin_amdgcn_sqrtf(va), vb = __builtin_amdgcn_sqrtf(vb);  // |X| (librosa-style filterbanks)
          if (s <= 8) {
            pA[64 * s] = va;
            pB[-64 * s] = vb;
          } else {
            pA2[64 * s] = va;
            pB2[-64 * s] = vb;
          }
        }
      }
      {  // step 16: lane 0's last pair, bins 480 and 544 (odd outputs 7 and 8 of row 0)
        const v2 tw = ctws[16 * 32 + q];
        const v2 zk = R1[7], m = R1[8];
        const v2 sp = m * HF_CJ + zk;
        const v2 dm = m * HF_NCJ + zk;
        const v2 tt = cmul2(dm, tw);
        const v2 xp = sp + tt, xm = sp - tt;
        float va = xp.x * xp.x + xp.y * xp.y, vb = xm.x * xm.x + xm.y * xm.y;
        if (mag) va = __builtin_amdgcn_sqrtf(va), vb = __builtin_amdgcn_sqrtf(vb);
        if (q == 0) {
          prow[480] = va;
          prow[544] = vb;
        }
      }
    }
    HFC_T(4);  // split step, power rows
    // the wave's two power rows are complete once its own (in-order) LDS queue has drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // the next round's span (requested at the start of this round) must have landed before this round's stores join the same
    // in-order vmcnt queue: waiting here instead of at the top of the next round never waits for the stores
    if (kPrefetch) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    HFC_T(5);  // wait for the next span
    // ---- mel filterbank on the matrix cores: one accumulator set after the other, 32 steps at a time, two accumulation chains ----
    float* orow = p.out + (cd.out_row + f0) * p.out_stride;
#pragma unroll
    for (int s = 0; s < kXMaxSets; ++s) {
      if (kFixed ? s < 3 : s < p.nsets) {  // uniform
        const float* lt = ltab + s * 256 + 4 * lane_o;
        const int poff = __builtin_bit_cast(int, lt[0]);
        const int col = __builtin_bit_cast(int, lt[1]);
        const float m4 = lt[2], m8 = lt[3];
        const float* pa = myreg + poff;
        const float* wb = wtab + (kFixed ? kFixStep0[s < 4 ? s : 3] : p.step0[s]) * 64 + 4 * lane_o;
        // opaque per round: otherwise hipcc evaluates every "chunk c4 exists" test once per kernel, runs out of SGPRs for the results
        // and fetches them back with v_readlane in front of every chunk
        int nsteps = kFixSteps[s < 4 ? s : 3];
        if (!kFixed) {
          nsteps = p.steps[s];
          asm volatile("" : "+s"(nsteps));
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk8 = 0; blk8 < kXMaxSteps / 32; ++blk8) {
          if (32 * blk8 < nsteps) {  // uniform
            f32x4 av[8], bv[8];
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
              if (32 * blk8 + 4 * c4 < nsteps) {  // uniform
                av[c4] = *reinterpret_cast<const f32x4*>(pa + 32 * blk8 + 4 * c4);
                bv[c4] = *reinterpret_cast<const f32x4*>(wb + (8 * blk8 + c4) * 256);
              }
            }
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) {
              if (32 * blk8 + 4 * c4 < nsteps) {  // uniform
#pragma unroll
                for (int i = 0; i < 4; i += 2) {  // two accumulation chains, alternating: no back-to-back dependent MFMAs
                  acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c4][i], bv[c4][i], acc0, 0, 0, 0);
                  acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c4][i + 1], bv[c4][i + 1], acc1, 0, 0, 0);
                }
              }
            }
          }
        }
        const f32x4 acc = acc0 + acc1;
        float val[4];  // rows 2, 3 of a block repeat the two frames
        mel4_reduce_floor(acc, m4, m8, p.mel_floor, val);  // fft_common.hpp: row_shr:4 / row_shr:8 multiply-adds, floor
        val[0] = __builtin_amdgcn_logf(val[0]) * log_scale;
        val[1] = __builtin_amdgcn_logf(val[1]) * log_scale;
        if (col < M) mel4_store_saddr<2>(orow, (unsigned)col, p.out_stride, nf, val);
      }
    }
    HFC_T(6);  // mel phase: operand reads, MFMAs, reduction, log, stores
#ifdef HIPFEAT_PHASE_TIMERS
    hfc_acc[7] += 1;
#endif
  }
#ifdef HIPFEAT_PHASE_TIMERS
  if (lane == 0 && g_phase_buf) {
    unsigned long long* o = g_phase_buf + ((size_t)blockIdx.x * kXMaxWaves + wv) * 8;
    for (int i = 0; i < 8; ++i) o[i] = hfc_acc[i];
  }
#endif
}

}  // namespace hipfeat
