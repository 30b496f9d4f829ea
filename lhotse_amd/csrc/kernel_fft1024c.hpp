// fft1024 fast path, wave-autonomous (22.05 / 24 / 32 kHz Kaldi log-mel filterbanks, Wav2LogFilterBank layers.py:565-578):
// the organisation of kernel_fft512c.hpp -- a wave owns its four frames from the samples in HBM to the stored log-mel rows, no
// workgroup barrier in the steady state -- with twice the FFT per frame.
//
//   real FFT(1024) = complex FFT(512) of z[n] = y[2n] + i y[2n+1], 512 = 32 x 16 on 16 lanes per frame:
//     pass 1   lane q holds z[16 n1 + q], n1 = 0..31 (rows >= ceil(N/32) are zero padding) -> 32-point FFT in registers
//              (one radix-2 stage + two 16-point FFTs), times W_512^(q k1);
//     exchange through the wave's LDS region in two halves of 16 rows (all lanes read in both halves: no masked reads);
//     pass 2   lane q owns the rows k1 = q and 32 - q (lane 0: rows 0 and 16) -> two 16-point FFTs: Za[k2] = Z[q + 32 k2],
//              Zb[k2] = Z[(32 - q) + 32 k2];
//     split    X[k] = E[k] + W_1024^k O[k] on bin pairs (k, 512 - k): the partner of Za[s] is Zb[15 - s] -- the SAME lane, so the
//              step needs no cross-lane traffic at all.  Lane 0 is the exception: its two rows are their own mirrors (17 pairs
//              instead of 16); it follows the common schedule through per-step register selects and one extra step;
//     |X|^2 -> four power rows of 528 floats in the same LDS region.
//   mel filterbank on the matrix cores with v_mfma_f32_4x4x1_16B_f32 exactly as in kernel_fft512c.hpp (mel4_schedule.hpp), with up to
//   four accumulator sets of up to 32 steps (513 bins: the bands are twice as long), processed one set after the other.
// Workgroup = 8 waves (one per CU: 14 KB of LDS per wave + 25 KB of shared tables), <= 256 VGPRs.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"
#include "kernel_fft512c.hpp"  // HFC_SEP, mul24, phase-timer macros

namespace hipfeat {

constexpr int kWExRowStride = 34;                        // dwords per exchange row (16 complex + 2 pad)
constexpr int kWExFrameStride = 16 * kWExRowStride;      // 544 (== 32 mod 64): 16 rows per half
constexpr int kWPRowStride = 528;                        // dwords per power row (513 bins + pad; == 16 mod 64)
constexpr int kWRegion = 4 * kWExFrameStride + 16;       // 2192 dwords per wave; the 4 power rows (2112) alias it
constexpr int kWMaxSets = 4;                             // accumulator sets (16 slots each)
constexpr int kWMaxSteps = 32;                           // MFMA steps per set
constexpr int kWWaves = 8;                               // waves per workgroup (generic instances: 2 waves/SIMD)
constexpr int kWWavesFixed = 12;                         // fixed-schedule instances: 3 waves/SIMD (137 VGPRs; the span buffer aliases the region)
constexpr int kWSplitSteps = 17;                         // bin-pair steps of the split (16 + lane 0's extra one)

struct Fft1024cParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  // shared LDS image, copied once per workgroup: [nrows][16] v2 window/2 | [32][16] v2 W_512^(q k1) | [17][16] v2 split twiddles
  // -i W_1024^k(step, lane) | weight table [total steps / 4][64 lanes][4 steps] | lane table [sets][64 lanes][4]
  const float* shared_consts;
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc;
  int32_t frames_per_block, rounds;  // rounds of 4 frames per wave; frames_per_block = 8 waves * rounds * 4
  int32_t N, shift, npad_left, M, flags;
  float preemph, mel_floor;
  int32_t shared_floats, wtab_off, ltab_off;
  int32_t xs_floats;  // floats of one wave's sample-span buffer (multiple of 4)
  int32_t nsets, steps[kWMaxSets], step0[kWMaxSets];  // accumulator sets: MFMA steps (multiples of 4) and first step in the weight table
};

// r = m ? a : b per lane with the mask in an SGPR pair (VOP3 encoding: the VOP2 / VCC form of v_cndmask issues ~5x slower on gfx950)
__device__ __forceinline__ float sel64(unsigned long long m, float a, float b) {
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ v2 sel64(unsigned long long m, v2 a, v2 b) { return v2{sel64(m, a.x, b.x), sel64(m, a.y, b.y)}; }

// S0 / S1 / S2 != 0: the (padded) step counts of a three-set schedule as compile-time constants -- the 80-filter Kaldi defaults at 24 / 32 kHz
// (24, 16, 8) and 22.05 kHz (24, 24, 8).  Without the "does this chunk exist" tests the mel phase needs 100 VGPRs, ~350 scalar instructions
// and 20 SGPR spills less per round (+2 % at 24 kHz, +4 % at 32 kHz); any other filterbank runs the generic <NROWS, 0, 0, 0>.
//
// The fixed-schedule instances run 12 waves per workgroup = 3 waves/SIMD (the power probe of round 3 shows these kernels at the full clock under the
// power cap: latency-limited at 2 waves/SIMD).  The LDS for that comes from giving up the span prefetch: the wave's span buffer ALIASES its
// exchange / power region (8.8 KB per wave instead of 14.2), the span of a round is requested at the top of that round and waited for at
// once -- the third wave per SIMD covers that latency and more.
//
// PLAIN: no DC removal and no pre-emphasis (the librosa-style front end, lhotse/features/librosa_fbank.py:66-137: y = x * window): no
// left-neighbour reads, no mean, and -- what makes the librosa default <32, 16, 16, 8, true> fit the 168 registers of 3 waves/SIMD with
// all 32 input rows live, where round 3's attempt spilled 17 -- the window goes onto the samples as they arrive.
template <int NROWS, int S0 = 0, int S1 = 0, int S2 = 0, bool PLAIN = false>
__global__ __launch_bounds__(64 * (S0 != 0 ? kWWavesFixed : kWWaves), (S0 != 0 ? 3 : 2)) void fft1024c_kernel(const Fft1024cParams p) {
  constexpr bool kFixed = S0 != 0;
  constexpr int kWv = kFixed ? kWWavesFixed : kWWaves;  // waves per workgroup
  constexpr bool kPrefetch = !kFixed;                   // span of round r + 1 requested during round r (needs a span buffer of its own)
  constexpr int kFixSteps[4] = {S0, S1, S2, 0}, kFixStep0[4] = {0, S0, S0 + S1, S0 + S1 + S2};
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  const v2* cwin = reinterpret_cast<const v2*>(smem);  // [NROWS][16]
  const v2* ctwp = cwin + NROWS * 16;                  // [32][16] row k1, column q
  const v2* ctws = ctwp + 512;                         // [17][16] split twiddles per (step, lane)
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
  // round 5, experiment 2 (see kernel_fft2048c.hpp): the fixed-schedule Kaldi instances are the 80-filter defaults at 24 kHz
  // <20, 24,16,8> (600-sample frames, hop 240), 32 kHz <26, 24,16,8> (800 / 320) and 22.05 kHz <20, 24,24,8> (551 / 220) and nothing
  // else (hipfeat.hip selects them on exactly these numbers): frame geometry, span length and filter count are compile-time constants
#ifdef HIPFEAT_ABL_RUNTIME_GEOMETRY
  constexpr bool kGeo = false;
#else
  constexpr bool kGeo = kFixed && !PLAIN;
#endif
  constexpr int kGeoN = NROWS == 26 ? 800 : (S1 == 24 ? 551 : 600), kGeoShift = NROWS == 26 ? 320 : (S1 == 24 ? 220 : 240);
  const int N = kGeo ? kGeoN : p.N, shift = kGeo ? kGeoShift : p.shift;
  const int npad_left = kGeo ? (kGeoN - kGeoShift) / 2 : p.npad_left;
  const int xs_floats = kGeo ? ((3 * kGeoShift + 32 * NROWS + 3) & ~3) : p.xs_floats;
  const int M = kGeo ? 80 : p.M;

  for (int i = tid; i < p.shared_floats; i += 64 * kWv) smem[i] = p.shared_consts[i];
  float* xs = smem + p.shared_floats + wv * (kPrefetch ? xs_floats + kWRegion : kWRegion);
  float* myreg = kPrefetch ? xs + xs_floats : xs;  // (no prefetch: the span is dead once the samples are in registers, before the exchange)
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  // the fixed-schedule Kaldi instances (not PLAIN = the librosa default) serve Kaldi plans only: |X|^2 and the natural log are
  // compile-time facts there (round 5, experiment 1; see kernel_fft2048c.hpp)
  constexpr bool kKaldiOnly = kFixed && !PLAIN;
#ifdef HIPFEAT_ABL_RUNTIME_MAG
  const bool mag = (p.flags & F_FFT_MAG) != 0;
#else
  const bool mag = kKaldiOnly ? false : (p.flags & F_FFT_MAG) != 0;
#endif
  const float log_scale = (!kKaldiOnly && (p.flags & F_LOG10)) ? 0.30102999566398120f : 0.69314718055994531f;  // log2 -> log10 / ln
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
      if (!kKaldiOnly && (p.flags & F_CENTER))  // torch.stft / librosa "reflect" padding (the edge sample is not repeated)
        for (int i = (int)(lane4 >> 2); i < xs_floats; i += 64) xs[i] = load_sample_center(w, j0 + i, cd.num_samples);
      else
        for (int i = (int)(lane4 >> 2); i < xs_floats; i += 64) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
    }
  };

  const int first_frame = fb * p.frames_per_block + 4 * wv;  // the waves take the frame quads round-robin
  __syncthreads();  // the constant tables are in place (the only workgroup barrier of the kernel)
  if (kPrefetch && first_frame < cd.num_frames) stage_span(first_frame, (unsigned)lane * 4u);

#ifdef HIPFEAT_PHASE_TIMERS
  unsigned long long hfc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hfc_last = __builtin_readcyclecounter();
#endif
  for (int r = 0; r < p.rounds; ++r) {
    const int f0 = first_frame + 4 * kWv * r;
    if (f0 >= cd.num_frames) break;
    const int nf = min(4, cd.num_frames - f0);

    int lane_o = lane;  // opaque copy: keeps LICM from pinning per-lane addresses in VGPRs for the whole kernel
    asm volatile("" : "+v"(lane_o));
    if (!kPrefetch) stage_span(f0, (unsigned)lane_o * 4u);  // (the previous round's power-row reads were consumed by its MFMAs: the region is free)
    if (r == 0 || !kPrefetch) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); with prefetch, later rounds waited before their predecessor's stores
    const int q = lane_o & 15, g = lane_o >> 4;
    const unsigned long long q0 = __builtin_amdgcn_ballot_w64(q == 0);  // lanes 0, 16, 32, 48

    v2 Za[16], Zb[16];
    {
      const float* x = xs + mul24(g, shift) + 2 * q;
      v2 z[32];
      v2 win[NROWS];
      float pv[PLAIN ? 1 : NROWS];  // left neighbour of each pair's first sample (the frame's first sample replicates itself, layers.py:166)
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        z[n1] = *reinterpret_cast<const v2*>(x + 32 * n1);
        HFC_SEP();
      }
      if (!PLAIN) {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) pv[n1] = n1 == 0 ? x[q == 0 ? 0 : -1] : x[32 * n1 - 1];
      }
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        win[n1] = cwin[n1 * 16 + q];
        HFC_SEP();
      }
      // the samples are in flight to registers; once they have arrived the buffer is free for the next round's span
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HFC_T(0);  // sample, neighbour and window reads
      if (kPrefetch && r + 1 < p.rounds && f0 + 4 * kWv < cd.num_frames) stage_span(f0 + 4 * kWv, (unsigned)lane_o * 4u);
      HFC_T(1);  // span request (LDS-DMA issue)

#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        if (32 * (n1 + 1) > N) {  // samples at or beyond N are not part of the frame
          const int m0 = 32 * n1 + 2 * q;
          if (m0 >= N) z[n1].x = 0.f;
          if (m0 + 1 >= N) z[n1].y = 0.f;
        }
      }
      if (PLAIN) {
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) z[n1] = z[n1] * win[n1];
      } else {
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
          mu = row16_sum(sum2.x + sum2.y) * inv_n;
        }
        // y[n] = (x[n] - mu) - c (x[n-1] - mu) = x[n] - c x[n-1] - (1 - c) mu, times the window
        const float nc = -c, mu1 = (1.0f - c) * mu;
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          v2 t;
          t.x = fmaf(nc, pv[n1], z[n1].x);
          t.y = fmaf(nc, z[n1].x, z[n1].y);
          z[n1] = (t - v2{mu1, mu1}) * win[n1];
        }
      }
#pragma unroll
      for (int n1 = NROWS; n1 < 32; ++n1) z[n1] = v2{0.f, 0.f};
      // the pass twiddles W_512^(q k1) are requested before the 32-point FFT (62 registers that a 2-waves-per-SIMD kernel has) --
      // except when all 32 input rows are live (NROWS == 32: that request would spill): then in two bursts of 16 after it
      v2 twp[32];
      if (NROWS < 32) {
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) {
          twp[k1] = ctwp[k1 * 16 + q];
          HFC_SEP();
        }
      }
      v2 a[32];
      fft32<NROWS>(z, a);
      if (NROWS < 32) {
#pragma unroll
        for (int k1 = 1; k1 < 32; ++k1) a[k1] = cmul2(a[k1], twp[k1]);
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int k1 = 16 * h + (h == 0 ? 1 : 0); k1 < 16 * h + 16; ++k1) {
            twp[k1] = ctwp[k1 * 16 + q];
            HFC_SEP();
          }
#pragma unroll
          for (int k1 = 16 * h + (h == 0 ? 1 : 0); k1 < 16 * h + 16; ++k1) a[k1] = cmul2(a[k1], twp[k1]);
        }
      }
      HFC_T(2);  // mean, prolog, pass 1, twiddles
      // exchange in two halves of 16 rows: half 0 carries every lane's first row (k1 = q), half 1 its second one
      // (k1 = 32 - q, i.e. slot (16 - q) % 16 of the half; lane 0: k1 = 16, slot 0)
      float* exf = myreg + mul24(g, kWExFrameStride);
      v2 b0[16], b1[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) *reinterpret_cast<v2*>(exf + rr * kWExRowStride + 2 * q) = a[16 * h + rr];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float* src = exf + mul24(h == 0 ? q : ((16 - q) & 15), kWExRowStride);
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
          if (h == 0) b0[n2] = *reinterpret_cast<const v2*>(src + 2 * n2);
          else b1[n2] = *reinterpret_cast<const v2*>(src + 2 * n2);
          HFC_SEP();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      HFC_T(3);  // exchange
      fft16(b0, Za);
      fft16(b1, Zb);
    }

    {
      float* prow = myreg + mul24(g, kWPRowStride);
      if (q < 15) prow[513 + q] = 0.f;  // the padding a slot may read past bin 512 (weight 0) must be finite
      // write addresses: bin kA = q + 32 s and its mirror 512 - kA; lane 0's second row (steps 9..15) sits elsewhere
      float* pA = prow + q;
      float* pB = prow + 512 - q;
      float* pA2 = prow + __builtin_bit_cast(int, sel64(q0, __builtin_bit_cast(float, -272), __builtin_bit_cast(float, q)));
      float* pB2 = prow + __builtin_bit_cast(int, sel64(q0, __builtin_bit_cast(float, 784), __builtin_bit_cast(float, 512 - q)));
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        v2 tw[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          tw[rr] = ctws[(4 * h + rr) * 16 + q];
          HFC_SEP();
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int s = 4 * h + rr;
          v2 zk, m;
          if (s <= 8) {
            zk = Za[s];
            m = sel64(q0, Za[(16 - s) & 15], Zb[15 - s]);
          } else {
            zk = sel64(q0, Zb[s - 9], Za[s]);
            m = sel64(q0, Zb[24 - s], Zb[15 - s]);
          }
          const v2 sp = m * HF_CJ + zk;
          const v2 dm = m * HF_NCJ + zk;
          const v2 tt = cmul2(dm, tw[rr]);
          const v2 xp = sp + tt, xm = sp - tt;
          float va = xp.x * xp.x + xp.y * xp.y, vb = xm.x * xm.x + xm.y * xm.y;
          if (mag) va = __builtin_amdgcn_sqrtf(va), vb = __builtin_amdgcn_sqrtf(vb);  // |X| (librosa-style filterbanks)
          if (s <= 8) {
            pA[32 * s] = va;
            pB[-32 * s] = vb;
          } else {
            pA2[32 * s] = va;
            pB2[-32 * s] = vb;
          }
        }
      }
      {  // step 16: lane 0's last pair, bins 240 and 272 (its row k1 = 16, k2 = 7 and 8)
        const v2 tw = ctws[16 * 16 + q];
        const v2 zk = Zb[7], m = Zb[8];
        const v2 sp = m * HF_CJ + zk;
        const v2 dm = m * HF_NCJ + zk;
        const v2 tt = cmul2(dm, tw);
        const v2 xp = sp + tt, xm = sp - tt;
        float va = xp.x * xp.x + xp.y * xp.y, vb = xm.x * xm.x + xm.y * xm.y;
        if (mag) va = __builtin_amdgcn_sqrtf(va), vb = __builtin_amdgcn_sqrtf(vb);
        if (q == 0) {
          prow[240] = va;
          prow[272] = vb;
        }
      }
    }
    HFC_T(4);  // pass 2, split step, power rows
    // the wave's four power rows are complete once its own (in-order) LDS queue has drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // the next round's span (requested at the start of this round) must have landed before this round's stores join the
    // same in-order vmcnt queue: waiting here instead of at the top of the next round never waits for the stores
    if (kPrefetch) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    HFC_T(5);  // wait for the next span
    // ---- mel filterbank on the matrix cores: one accumulator set after the other (kernel_fft512c.hpp, mel4_schedule.hpp) ----
    float* orow = p.out + (cd.out_row + f0) * p.out_stride;
    // one accumulator set after the other; inside a set two interleaved accumulation chains (even / odd chunks of 4 steps)
#pragma unroll
    for (int s = 0; s < kWMaxSets; ++s) {
      if (kFixed ? s < 3 : s < p.nsets) {  // uniform
        const float* lt = ltab + s * 256 + 4 * lane_o;
        const int poff = __builtin_bit_cast(int, lt[0]);
        const int col = __builtin_bit_cast(int, lt[1]);
        const float m4 = lt[2], m8 = lt[3];
        const float* pa = myreg + poff;
        const float* wb = wtab + (kFixed ? kFixStep0[s] : p.step0[s]) * 64 + 4 * lane_o;
        int nsteps = kFixSteps[s];
        if (!kFixed) {
          nsteps = p.steps[s];  // opaque per round: keeps hipcc from hoisting (and then spilling) every "chunk exists" test
          asm volatile("" : "+s"(nsteps));
        }
        f32x4 av[kWMaxSteps / 4], bv[kWMaxSteps / 4];
#pragma unroll
        for (int c4 = 0; c4 < kWMaxSteps / 4; ++c4) {
          if (4 * c4 < nsteps) {  // uniform
            av[c4] = *reinterpret_cast<const f32x4*>(pa + 4 * c4);
            bv[c4] = *reinterpret_cast<const f32x4*>(wb + c4 * 256);
          }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < kWMaxSteps / 4; c4 += 2) {
          if (4 * c4 < nsteps) {  // uniform; steps are padded to multiples of 8 on the host (zero weights)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c4][i], bv[c4][i], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c4 + 1][i], bv[c4 + 1][i], acc1, 0, 0, 0);
            }
          }
        }
        const f32x4 acc = acc0 + acc1;
        float val[4];
        mel4_reduce_floor(acc, m4, m8, p.mel_floor, val);  // fft_common.hpp: row_shr:4 / row_shr:8 multiply-adds, floor
#pragma unroll
        for (int i = 0; i < 4; ++i) val[i] = __builtin_amdgcn_logf(val[i]) * log_scale;
        if (col < M) mel4_store_saddr<4>(orow, (unsigned)col, p.out_stride, nf, val);
      }
    }
    HFC_T(6);  // mel phase: operand reads, MFMAs, reduction, log, stores
#ifdef HIPFEAT_PHASE_TIMERS
    hfc_acc[7] += 1;
#endif
  }
#ifdef HIPFEAT_PHASE_TIMERS
  if (lane == 0 && g_phase_buf) {
    unsigned long long* o = g_phase_buf + ((size_t)blockIdx.x * kWv + wv) * 8;
    for (int i = 0; i < 8; ++i) o[i] = hfc_acc[i];
  }
#endif
}

}  // namespace hipfeat
