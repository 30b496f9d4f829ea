// fft512 fast path, wave-autonomous variant ("c"): log-mel filterbank (Wav2LogFilterBank, layers.py:565-578) for
// fft 512 with NO workgroup barrier in the steady state.
//
// The "b" kernel couples the four waves of a workgroup twice per 16-frame tile: the 16x16x4 matrix-core GEMM of the
// mel stage needs the power rows of all 16 frames, i.e. of all four waves (barrier), and the shared sample span may only
// be overwritten once everybody has framed it (barrier).  Measured on MI355X (phase timers, round 2): of 13.9 k clk per
// wave-tile 3.3 k were the mel stage (five dependent LDS round trips + the re-fetch of 48 KB of filter weights per tile
// from L2) and 1.3 k barrier waits, while VALU, LDS and matrix pipes each sat half idle.
//
// Here a wave owns its four frames from the samples to the stored log-mel rows:
//   * its own sample span (3 shift + N floats) comes by LDS-DMA into a wave-private buffer, requested one round ahead
//     (as soon as the current round's samples sit in registers);
//   * S3 (framing, DC, pre-emphasis, window, 512-point real FFT on 16 lanes per frame, |X|^2) is the "b" kernel's, with
//     single ds_read_b64 (hipcc's merged ds_read2_b64 runs at half the LDS rate on gfx950: tools/ubench/lds_rate.hip)
//     and two-instruction complex multiplies;
//   * the mel filterbank runs on the matrix cores with v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 blocks per
//     instruction, block = (4 frames of THIS wave) x (4 consecutive mel filters) x (1 bin), so a block "slot" walks the
//     band of its filter group one bin per instruction.  Wide bands are split over 2 or 4 adjacent slots (summed with
//     two row_shr DPP multiply-adds), which levels the 298 (group, bin) pairs of the 80-filter bank to ~10 steps on each
//     of 2 accumulator sets.  A operand = one power value per lane (ds_read_b32 from the wave's own rows), B operand =
//     one weight per lane from a 5 KB table in LDS: no vector-memory traffic for weights at all;
//   * log and store straight from the accumulators (lane = 4 slot + filter, register = frame).
// One __syncthreads() at kernel start (constant tables), none afterwards.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

constexpr int kCExRowStride = 34;                          // dwords per exchange row (16 complex + 2 pad)
constexpr int kCExFrameStride = 8 * kCExRowStride + 16;    // 288 (== 32 mod 64): 8 rows per half
constexpr int kCPRowStride = 272;                          // dwords per power row (== 16 mod 64: the 4 frames of a slot hit disjoint bank quads)
constexpr int kCRegion = 4 * kCExFrameStride + 16;         // 1168 dwords per wave; the 4 power rows (1088) alias it
constexpr int kCMaxSets = 2;                               // accumulator sets (16 slots each)
constexpr int kCMaxSteps = 16;                             // MFMA steps per set
constexpr int kCWaves = 8;                                 // waves per workgroup
constexpr int kCLmStride = 48;                             // MFCC: floats per log-mel row in LDS (<= 40 filters; rows 3 bank quads apart)
constexpr int kCDctChunks = 10;                            // MFCC: DCT steps / 4 (<= 40 filters), operands resident in registers
constexpr int kCDctChunksSmall = 6;                        // MFCC with <= 24 filters (mode 3): leaves room for the split-step twiddles

struct Fft512cParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  // shared LDS image, copied once per workgroup: [nrows][16] v2 window/2 | [16][16] v2 W_256^(q k1) | [8][16] v2 -i W_512^(q+16 k2)
  // | weight table [2 sets][16 steps / 4][64 lanes][4 steps] (zero beyond a set's steps) | lane table [2 sets][64 lanes][4]
  // (power-row offset (int), output column (int, >= M = none), m4, m8)
  const float* shared_consts;
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc;
  int32_t frames_per_block, rounds;  // rounds of 4 frames per wave; frames_per_block = 8 waves * rounds * 4
  int32_t N, shift, npad_left, M, flags;
  float preemph, mel_floor;
  int32_t shared_floats;  // floats of the shared image
  int32_t wtab_off, ltab_off;  // float offsets of the weight / lane tables inside the image
  int32_t xs_floats;      // floats of one wave's sample-span buffer (multiple of 4)
  // MFCC (MODE 2): [kCDctChunks][64 lanes][4] DCT operands (lane = cepstral coefficient, step = filter; layers.py:697-706), then [64] lifter
  const float* dct_tab;
  int32_t C;
  int32_t total_quads;  // FLAT instances: frame quads (4 frames of one cut) of the whole batch; CutDesc::first_block then is a cut's first quad
};

#ifdef HIPFEAT_PHASE_TIMERS
#define HFC_T(i) { const unsigned long long tt_ = __builtin_readcyclecounter(); hfc_acc[i] += tt_ - hfc_last; hfc_last = tt_; }
#else
#define HFC_T(i)
#endif
#define HFC_SEP() asm volatile("")
// lane-index multiplies: v_mul_u32_u24 issues at the full VALU rate, v_mul_lo_u32 at a quarter of it
__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }

// NROWS: pass-1 rows that can hold samples; NFULL: rows known to lie entirely inside the frame (N >= 32 NFULL): no length masks there;
// MODE 0: log-mel filterbank on 2 accumulator sets x 16 steps (many narrow filters: the 80-filter default); 1: log-mel on 1 set x 32
// steps (few, wide filters: 23 / 40); 2: MFCC = mode 1 + the DCT as a second run of 4 x 4 x 1 blocks (Wav2MFCC, layers.py:708-724);
// 3: MFCC with <= 24 filters (the 23-filter default): 6 instead of 10 chunks of DCT operands, and the split-step twiddles in registers
//
// FLAT (round 4; ragged batches): the waves of the launch take the frame QUADS of the whole batch round-robin -- workgroup g owns quads
// [g, g + 1) x 8 rounds, wave wv of it quad 8 r + wv in round r -- instead of every cut having workgroups of its own.  A cut's last
// workgroup used to be half empty on average and short workgroups paid the start-up (constant image, first span) for a few rounds of
// work: 17 % of a LibriSpeech-like batch.  A wave keeps the descriptor of the cut it is in and steps to the next one when its quad index
// passes the cut's last quad (CutDesc::first_block = first quad of the cut; one wave-uniform compare per round, a descriptor load per
// crossing); the workgroup's first cut comes from the workgroup -> cut map (common.hpp).  The four frames of a quad always belong to ONE cut.
template <int NROWS, int NFULL, int MODE, bool FLAT = false>
__global__ __launch_bounds__(64 * kCWaves, 4) void fft512c_kernel(const Fft512cParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  const v2* cwin = reinterpret_cast<const v2*>(smem);  // [NROWS][16]
  const v2* ctwp = cwin + NROWS * 16;                  // [16][16] row k1, column q
  const v2* ctws = ctwp + 256;                         // [8][16] w = -i W_512^(q+16 k2)
  const float* wtab = smem + p.wtab_off;
  const float* ltab = smem + p.ltab_off;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int blk = blockIdx.x;
  int cut, fb;
  if (FLAT) {
    cut = __builtin_amdgcn_readfirstlane(block_cut_map(p.cuts, p.num_cuts)[blk]);
    fb = 0;
  } else if (p.uniform_bpc > 0) {
    cut = blk / p.uniform_bpc;
    fb = blk - cut * p.uniform_bpc;
  } else {
    cut = p.uniform_bpc < 0 ? block_cut_map(p.cuts, p.num_cuts)[blk] : find_cut(p.cuts, p.num_cuts, blk);
    fb = blk - p.cuts[cut].first_block;
  }
  CutDesc cd = FLAT ? load_cut_uniform(p.cuts + cut) : p.cuts[cut];  // (FLAT: the cut of the CURRENT round; it changes as the wave walks through the batch)
  const int N = p.N, shift = p.shift;

  for (int i = tid; i < p.shared_floats; i += 64 * kCWaves) smem[i] = p.shared_consts[i];
  float* xs = smem + p.shared_floats + wv * (p.xs_floats + kCRegion);
  float* myreg = xs + p.xs_floats;
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;

  // Stage the span of the four frames starting at f0 into this wave's buffer.  Interior rounds: LDS-DMA (lane i supplies
  // the global address of its 16 bytes, the hardware writes piece base + 16 i).  Rounds touching a cut edge (reflection,
  // zero padding of a batch row): per-lane loads through the edge rule.
  auto stage_span = [&](const CutDesc& cd, int f0, unsigned lane4) {  // (the descriptor of the round that is being staged)
    const float* __restrict__ w = p.wave + cd.wave_off;
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    if (j0 >= 0 && j0 + p.xs_floats <= cd.num_samples) {
      const char* src = reinterpret_cast<const char*>(w + j0);  // uniform
      // whole 1 KiB pieces without a lane mask, then the (shorter) last piece: straight-line code, one exec mask
      const int nfull = p.xs_floats >> 8;
#pragma unroll
      for (int ch = 0; ch < 6; ++ch) {
        if (ch < nfull)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)ch * 1024u + 4u * lane4)),
                                           (__attribute__((address_space(3))) void*)(xs + ch * 256), 16, 0, 0);
      }
      if ((unsigned)nfull * 256u + lane4 < (unsigned)p.xs_floats)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)nfull * 1024u + 4u * lane4)),
                                         (__attribute__((address_space(3))) void*)(xs + nfull * 256), 16, 0, 0);
    } else {
      for (int i = (int)(lane4 >> 2); i < p.xs_floats; i += 64) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);  // the whole buffer: rows past N are read (and met by a zero window) too
    }
  };

  // the waves of a workgroup take the frame quads round-robin, so that a short cut still spreads over all of them
  const int first_frame = fb * p.frames_per_block + 4 * wv;
  // FLAT: this wave's quad of round 0, the cut it lies in (a workgroup may span several short cuts), and the same for the round after
  // the current one (its span is requested a round ahead)
  int quad = 0, next_first = 0, f0_flat = 0;
  int cut_n = 0, quad_n = 0, next_first_n = 0, f0_n = 0;
  CutDesc cd_n = cd;
  auto advance = [&](int q, int& c, CutDesc& d, int& nfirst) {  // the cut of quad q >= the current one (wave-uniform)
    while (c + 1 < p.num_cuts && q >= nfirst) {
      ++c;
      d = load_cut_uniform(p.cuts + c);
      nfirst = c + 1 < p.num_cuts ? __builtin_amdgcn_readfirstlane(p.cuts[c + 1].first_block) : p.total_quads;
    }
  };
  if (FLAT) {
    quad = blk * (p.frames_per_block >> 2) + wv;
    next_first = cut + 1 < p.num_cuts ? __builtin_amdgcn_readfirstlane(p.cuts[cut + 1].first_block) : p.total_quads;
    advance(quad, cut, cd, next_first);
    f0_flat = 4 * (quad - cd.first_block);
    cut_n = cut, cd_n = cd, next_first_n = next_first, quad_n = quad + kCWaves;
    advance(quad_n, cut_n, cd_n, next_first_n);
    f0_n = 4 * (quad_n - cd_n.first_block);
  }
  __syncthreads();  // the constant tables are in place (the only workgroup barrier of the kernel)
  if (FLAT ? quad < p.total_quads : first_frame < cd.num_frames) stage_span(cd, FLAT ? f0_flat : first_frame, (unsigned)lane * 4u);

  // Both twiddle tables of this lane (15 + 8 complex values) live in registers for the whole kernel where the register budget of 4 waves
  // per SIMD allows it (not in MFCC mode, whose DCT operands take that room): 24 LDS reads less per round, + 4.9 % (same-call A/B).  The
  // window on top of that does not fit (10 spilled registers, - 2.5 %).
  constexpr bool kMfcc = MODE == 2 || MODE == 3;
  constexpr int DCH = MODE == 3 ? kCDctChunksSmall : kCDctChunks;
#ifdef HIPFEAT_ABL_MFCC_TWS  // round-6 experiment (VERDICT r5 task 4): MODE 2 keeps the split-step twiddles in registers as MODE 3 does, paid for
  // by reading the last four chunks of DCT operands from memory every round (tools/r6_mfcc_ab.sh; result in DESIGN section 4.1)
  constexpr bool kTwsExp = MODE == 2 && NROWS <= 13;
#else
  constexpr bool kTwsExp = false;
#endif
  // chunks of DCT operands resident in registers: all of them, except where they do not fit -- with 16 live input rows (32 ms frames) MODE 2
  // spilled 17 VGPRs into scratch (VERDICT r5 task 7: no instance with scratch); there 5 of the 10 chunks stay resident and the other 5 are
  // re-read from memory (L1 / L2 hits, 5 x 16 B per lane) while the filterbank runs.  Same operands in the same order: bit-identical.
  constexpr int DREG = kTwsExp ? 6 : (MODE == 2 && NROWS > 13) ? 5 : DCH;
  constexpr bool kDctLate = DREG < DCH;
  constexpr bool kRegTw = !kMfcc && NROWS <= 13;  // (16 live input rows leave no room either: 24 spilled registers)
  constexpr bool kRegTws = kRegTw || (MODE == 3 && NROWS <= 13) || kTwsExp;  // the split-step twiddles alone
  v2 twpreg[kRegTw ? 16 : 1], twsreg[kRegTws ? 8 : 1];
  if (kRegTw) {
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) twpreg[k1] = ctwp[k1 * 16 + (lane & 15)];
  }
  if (kRegTws) {
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) twsreg[k2] = ctws[k2 * 16 + (lane & 15)];
  }
  // MFCC: the DCT operands of this lane (its cepstral coefficient x every filter) and its lifter value stay in registers
  f32x4 dw[kMfcc ? DREG : 1];
  float lift = 1.0f;
  if (kMfcc) {
#pragma unroll
    for (int c4 = 0; c4 < DREG; ++c4) dw[c4] = *reinterpret_cast<const f32x4*>(p.dct_tab + (c4 * 64 + lane) * 4);
    lift = p.dct_tab[DCH * 256 + lane];
  }
#ifdef HIPFEAT_PHASE_TIMERS
  unsigned long long hfc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hfc_last = __builtin_readcyclecounter();
#endif
  for (int r = 0; r < p.rounds; ++r) {
    const int f0 = FLAT ? f0_flat : first_frame + 4 * kCWaves * r;
    if (FLAT ? quad >= p.total_quads : f0 >= cd.num_frames) break;
    const int nf = min(4, cd.num_frames - f0);

    // this round's span was requested a round ago by this very wave: its own vmcnt covers the LDS-DMA, the in-order LDS
    // queue covers the per-lane stores of an edge round
    if (r == 0)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    HFC_T(0);  // wait for the span
    // Re-derive every per-lane address inside the loop from an opaque copy of the lane id (LICM would otherwise pin ~25
    // loop-invariant addresses in VGPRs for the whole kernel and cost a wave of occupancy).
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int q = lane_o & 15, g = lane_o >> 4;

    int early_poff[2] = {0, 0};
    v2 Z[16];
    {
      const float* x = xs + mul24(g, shift) + 2 * q;
      v2 z[16];
      v2 win[NROWS];
      v2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        z[n1] = *reinterpret_cast<const v2*>(x + 32 * n1);
        HFC_SEP();
      }
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        win[n1] = cwin[n1 * 16 + q];
        HFC_SEP();
      }
      float pv[NROWS];  // left neighbour of each pair's first sample (the frame's first sample replicates itself)
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) pv[n1] = n1 == 0 ? x[q == 0 ? 0 : -1] : x[32 * n1 - 1];
      // the samples are in flight to registers; once they have arrived the buffer is free for the next round's span
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HFC_T(1);  // sample + window reads
      if (FLAT) {
        if (r + 1 < p.rounds && quad_n < p.total_quads) stage_span(cd_n, f0_n, (unsigned)lane_o * 4u);
      } else if (r + 1 < p.rounds && f0 + 4 * kCWaves < cd.num_frames) {
        stage_span(cd, f0 + 4 * kCWaves, (unsigned)lane_o * 4u);
      }

      // samples at or beyond N (the frame length) are not part of the frame (uniform test per row, lane mask only in
      // the boundary rows)
#pragma unroll
      for (int n1 = NFULL; n1 < NROWS; ++n1) {
        if (32 * (n1 + 1) > N) {
          const int m0 = 32 * n1 + 2 * q;
          if (m0 >= N) z[n1].x = 0.f;
          if (m0 + 1 >= N) z[n1].y = 0.f;
        }
      }
      {  // four partial sums: the dependent chain is 4 + 2 adds instead of NROWS
        v2 sa = z[0], sb = z[1], sc = z[2], sd = z[3];
#pragma unroll
        for (int n1 = 4; n1 < NROWS; ++n1) {
          if ((n1 & 3) == 0) sa += z[n1];
          if ((n1 & 3) == 1) sb += z[n1];
          if ((n1 & 3) == 2) sc += z[n1];
          if ((n1 & 3) == 3) sd += z[n1];
        }
        sum2 = (sa + sb) + (sc + sd);
      }
      float mu = 0.f;
      if (dc) mu = row16_sum(sum2.x + sum2.y) * inv_n;
      // pre-emphasis on the DC-free frame: y[n] = (x[n] - mu) - c (x[n-1] - mu) = x[n] - c x[n-1] - (1 - c) mu, the very first
      // sample of the frame replicating itself (layers.py:166).  The left neighbour of a pair's first element comes from the
      // span (one ds_read_b32 per row) instead of a cross-lane DPP chain.
      {
        const float nc = -c, mu1 = (1.0f - c) * mu;
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          v2 t;
          // two plain v_fma_f32 (inline asm): hipcc's SLP vectoriser otherwise builds the (x[2m-1], x[2m]) pair with two v_mov per row to
          // feed one v_pk_fma_f32 -- 13 instructions per round more, + 1.2 ... 3.5 % in same-call A/Bs
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "s"(nc), "v"(pv[n1]), "v"(z[n1].x));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "s"(nc), "v"(z[n1].x), "v"(z[n1].y));
          z[n1] = (t - v2{mu1, mu1}) * win[n1];
        }
      }
#pragma unroll
      for (int n1 = NROWS; n1 < 16; ++n1) z[n1] = v2{0.f, 0.f};
      v2 a[16];
      fft16(z, a);
      // pass twiddles W_256^(q k1): fetched from LDS in two bursts of 8 (one latency exposure each)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2 tw[8];
        if (kRegTw) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) tw[rr] = twpreg[8 * h + rr];
        } else {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) {
            tw[rr] = ctwp[(8 * h + rr) * 16 + q];
            HFC_SEP();
          }
        }
#pragma unroll
        for (int rr = (h == 0 ? 1 : 0); rr < 8; ++rr) a[8 * h + rr] = cmul2(a[8 * h + rr], tw[rr]);
      }

      HFC_T(2);  // DMA issue, prolog, pass 1, twiddles
      // exchange in two halves: rows k1 = 8h .. 8h+7 go through an 8-row block; lanes with (q >> 3) == h then read
      // "their" row (all n2) back
      float* exf = myreg + mul24(g, kCExFrameStride);
      v2 b[16];
#ifdef HIPFEAT_ABL_NO_EXCHANGE  // experiment builds (round 4: profiles/r04_lds_conflicts.txt): which phase owns the LDS bank conflicts?  (wrong results)
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) b[n2] = a[n2];
#else
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) *reinterpret_cast<v2*>(exf + rr * kCExRowStride + 2 * q) = a[8 * h + rr];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (h == 0 || q >= 8) {  // half 0: every lane reads (b is never undefined: nothing for hipcc to carry around the round loop); half 1: lanes 8.. replace it
#pragma unroll
          for (int n2 = 0; n2 < 16; ++n2) {
            b[n2] = *reinterpret_cast<const v2*>(exf + (q % 8) * kCExRowStride + 2 * n2);
            HFC_SEP();
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#endif
      HFC_T(3);  // exchange
      // the power-row offsets of this lane's filterbank slots are requested here, a phase early: the operand reads of the mel phase then
      // start without a dependent LDS look-up in front of them (+ 0.6 %)
      early_poff[0] = __builtin_bit_cast(int, ltab[4 * lane_o]);
      if (MODE == 0) early_poff[1] = __builtin_bit_cast(int, ltab[256 + 4 * lane_o]);
      fft16(b, Z);
    }

    {
      float* prow = myreg + mul24(g, kCPRowStride);
      float* pown = prow + q;
      float* ppar = prow + ((16 - q) & 15) + (q == 0 ? 16 : 0);
      if (q < kCPRowStride - 257) prow[257 + q] = 0.f;  // the padding a slot may read past bin 256 (weight 0) must be finite
      float t1[16];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].x);
        t1[2 * k2 + 1] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].y);
      }
      // second half of the lane map l -> (16 - l) % 16: shift right by one; lane 0 has no source and keeps its own
      // register (16 - k2) % 16 instead (its mirror partner is itself)
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_shr1_keep(Z[(16 - k2) & 15].x, t1[2 * k2]);
        t1[2 * k2 + 1] = dpp_shr1_keep(Z[(16 - k2) & 15].y, t1[2 * k2 + 1]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2 tw[4];  // split-step twiddles of 4 bin pairs per burst
        if (kRegTws) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tw[rr] = twsreg[4 * h + rr];
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            tw[rr] = ctws[(4 * h + rr) * 16 + q];
            HFC_SEP();
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int k2 = 4 * h + rr;
          const v2 m = v2{t1[2 * k2], t1[2 * k2 + 1]};
          const v2 sp = m * HF_CJ + Z[k2];
          const v2 dm = m * HF_NCJ + Z[k2];
          const v2 tt = cmul2(dm, tw[rr]);
          // the bin pair (k, 256 - k) side by side: re2 = (Re X[k], Re X[256-k]) = sp.x +- tt.x, im2 likewise; |X|^2 of both bins in two
          // packed instructions (the broadcasts are op_sel modifiers) instead of two multiplies and two multiply-adds
          v2 re2, im2, pw;
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(re2) : "v"(tt), "v"(HF_CJ), "v"(sp));
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(im2) : "v"(tt), "v"(HF_CJ), "v"(sp));
          pw = re2 * re2;
          pw = im2 * im2 + pw;
#ifdef HIPFEAT_ABL_NO_POWER
          asm volatile("" : : "v"(pw.x), "v"(pw.y));
#else
          pown[16 * k2] = pw.x;
          ppar[16 * (15 - k2)] = pw.y;
#endif
        }
      }
      if (q == 0) prow[128] = 4.f * (Z[8].x * Z[8].x + Z[8].y * Z[8].y);
    }
    HFC_T(4);  // pass 2, split step, power rows
    // the wave's four power rows are complete once its own (in-order) LDS queue has drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- mel filterbank on the matrix cores, 4 frames x 4 filters x 1 bin per block, 16 blocks per instruction ----
    // All operands of the phase are requested before the first MFMA (one LDS round trip; the FFT registers are dead):
    // A = four consecutive power values per 16-byte read (lane = slot b, frame i: P[i][bin0(b) + 4 c ..]), B = four
    // consecutive steps of the weight table per 16-byte read (lane = slot b, filter j).
    // the next round's span (requested at the start of this round) must have landed before this round's stores join the
    // same in-order vmcnt queue: waiting here instead of at the top of the next round never waits for the stores
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    f32x4 dwx[kDctLate ? DCH - DREG : 1];  // the non-resident DCT operand chunks of this round: in flight during the filterbank
    if (kDctLate) {
#pragma unroll
      for (int c4 = DREG; c4 < DCH; ++c4) dwx[c4 - DREG] = *reinterpret_cast<const f32x4*>(p.dct_tab + (c4 * 64 + lane_o) * 4);
    }
    float* orow = p.out + (cd.out_row + f0) * p.out_stride;
    constexpr int S = MODE == 0 ? kCMaxSets : 1, T = MODE == 0 ? kCMaxSteps : 2 * kCMaxSteps;  // S x T = 32 steps either way
    int lt_poff[S], lt_col[S];
    float lt_m4[S], lt_m8[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* lt = ltab + s * 256 + 4 * lane_o;
      lt_poff[s] = early_poff[s];
      lt_col[s] = __builtin_bit_cast(int, lt[1]);
      lt_m4[s] = lt[2];
      lt_m8[s] = lt[3];
    }
    f32x4 av[S][T / 4], bv[S][T / 4];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float* pa = myreg + lt_poff[s];
      const float* wb = wtab + s * (T * 64) + 4 * lane_o;
#pragma unroll
      for (int c4 = 0; c4 < T / 4; ++c4) {
#if defined(HIPFEAT_ABL_NO_MELOPS) || defined(HIPFEAT_ABL_NO_MEL_A)
        av[s][c4] = f32x4{1.f, 2.f, 3.f, 4.f} * __builtin_bit_cast(float, lane_o);
#else
        av[s][c4] = *reinterpret_cast<const f32x4*>(pa + 4 * c4);
#endif
#if defined(HIPFEAT_ABL_NO_MELOPS) || defined(HIPFEAT_ABL_NO_MEL_B)
        bv[s][c4] = f32x4{1.f, 2.f, 3.f, 4.f} * __builtin_bit_cast(float, lane_o + c4);
#else
        bv[s][c4] = *reinterpret_cast<const f32x4*>(wb + c4 * 256);
#endif
      }
    }
    HFC_T(5);  // operand reads of the mel phase issued (not yet waited for)
    float lm[4];  // MODE 2: the lane's log-mel values of the four frames
    {
      // Two independent accumulation chains, issued alternately: a 4 x 4 x 1 block issues in 8 clk but its result is ready for the
      // next link of the SAME chain only after 16 -- set after set (what the per-set loop compiled to) ran at half the matrix-core rate.
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int c4 = 0; c4 < (S * T) / 8; ++c4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (MODE == 0) {
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0][c4][i], bv[0][c4][i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[S - 1][c4][i], bv[S - 1][c4][i], acc[1], 0, 0, 0);
          } else {  // one set of 32 steps: even chunks on chain 0, odd chunks on chain 1
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0][2 * c4][i], bv[0][2 * c4][i], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0][2 * c4 + 1][i], bv[0][2 * c4 + 1][i], acc[1], 0, 0, 0);
          }
        }
      }
      if (MODE != 0) acc[0] += acc[1];
      // Reduction of the pieces of a split filter group, floor, log (fft_common.hpp::mel4_reduce_floor): every value of the round first,
      // then the stores of a set under ONE lane mask (the per-value `if` of the first version cost eight exec-mask round trips per round).
      float val[S][4];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        mel4_reduce_floor(acc[s], lt_m4[s], lt_m8[s], p.mel_floor, val[s]);
#pragma unroll
        for (int i = 0; i < 4; ++i) val[s][i] = fast_log(val[s][i]);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int col = lt_col[s];
        if (kMfcc) {
#pragma unroll
          for (int i = 0; i < 4; ++i) lm[i] = val[s][i];
        } else if (col < p.M) {
          mel4_store_saddr<4>(orow, (unsigned)col, p.out_stride, nf, val[s]);
        }
      }
    }
    if (kMfcc) {
      // ---- DCT on the matrix cores: block = (4 frames) x (4 cepstral coefficients) x (1 filter); all 16 blocks walk the filters together.
      // The power rows are dead (every operand of the filterbank sits in registers): the four log-mel rows overwrite their start.
      float* lmrow = myreg;
      const int col = lt_col[0];
      if (col < p.M) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lmrow[i * kCLmStride + col] = lm[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      f32x4 al[DCH];  // A operand: lane (block b, frame i) = log-mel of frame i, four filters per read (the same for every block)
#pragma unroll
      for (int c4 = 0; c4 < DCH; ++c4) al[c4] = *reinterpret_cast<const f32x4*>(lmrow + (lane_o & 3) * kCLmStride + 4 * c4);
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < DCH; ++c4) {
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          const f32x4 w4 = (kDctLate && c4 >= DREG) ? dwx[c4 >= DREG ? c4 - DREG : 0] : dw[c4 < DREG ? c4 : 0];
          d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(al[c4][i], w4[i], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(al[c4][i + 1], w4[i + 1], d1, 0, 0, 0);
        }
      }
      d0 += d1;
      if (lane_o < p.C) {  // lane = cepstral coefficient, register = frame
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < nf) orow[i * p.out_stride + lane_o] = d0[i] * lift;
      }
      // the next round's exchange writes follow these reads in the wave's own LDS queue (in order)
    }
    HFC_T(6);  // MFMAs, reduction, log, stores
#ifdef HIPFEAT_PHASE_TIMERS
    hfc_acc[7] += 1;
#endif
    // the next round's exchange writes follow this round's power-row reads in the wave's own LDS queue (in order)
    if (FLAT) {  // the staged round becomes the current one; locate the one after it
      quad = quad_n, cut = cut_n, cd = cd_n, next_first = next_first_n, f0_flat = f0_n;
      quad_n += kCWaves;
      advance(quad_n, cut_n, cd_n, next_first_n);
      f0_n = 4 * (quad_n - cd_n.first_block);
    }
  }
#ifdef HIPFEAT_PHASE_TIMERS
  if (lane == 0 && g_phase_buf) {
    unsigned long long* o = g_phase_buf + ((size_t)blockIdx.x * kCWaves + wv) * 8;
    for (int i = 0; i < 8; ++i) o[i] = hfc_acc[i];
  }
#endif
}

}  // namespace hipfeat
