// fft512 fast path with 16-frame tiles ("b"): MFCC (DCT as a second matrix-core GEMM), (log-)spectrogram, and log-mel
// filterbanks outside the static schedule of the wave-autonomous kernel (kernel_fft512c.hpp).  Workgroup = 4 waves, tile = 16
// consecutive frames of one cut, FOUR workgroups (16 waves) per CU:
//   S1  the tile's sample span (15 shift + N floats) goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4), requested one phase
//       ahead (during S5 of the previous tile, when the span buffer is dead); tiles touching a cut edge use per-lane loads
//       through the reflect / zero-pad rule;
//   S3  one frame per 16-lane group: DC removal, pre-emphasis, window, real FFT(512) = complex FFT(256) = 16 x 16 in registers
//       (packed (re, im) VGPR pairs), exchange in two halves through a 4.6 KB wave-private LDS region, split step on bin pairs
//       with the mirror operand fetched by DPP, |X|^2 -> LDS power tile P[16][260];
//   S5  mel = banded f32 GEMM on the matrix cores (v_mfma_f32_16x16x4_f32) over each 16-mel tile's non-zero band only; the
//       wave's filter weights (A operands) are re-fetched from L2 before the barrier; log epilogue, 16-byte stores;
//   S6  (MFCC) cepstra = DCT^T x log-mel as a second, dense but tiny MFMA GEMM from LDS, lifter, stores.
// <= 128 VGPRs and 34.7 KB LDS per workgroup.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"
#include "fft512_common.hpp"  // Fft512Params, WaveWork, tile constants

namespace hipfeat {

constexpr int kBExRowStride = 34;                        // dwords per exchange row (16 complex + 2 pad)
constexpr int kBExParts = 2;
constexpr int kBExFrameStride = 8 * kBExRowStride + 16;  // 288 (== 32 mod 64): 8 rows per half
constexpr int kBWaveRegion = 4 * kBExFrameStride + 16;   // 1168 dwords per wave (== 16 mod 64)
constexpr int kBMelVec = kMelARegs / 4;                  // 16-byte loads of filter weights per lane per tile


constexpr int kMaxDctGroups = 10;
#define HF_SEP() asm volatile("")  // keeps hipcc from merging two ds_read_b64 into one half-rate ds_read2_b64 (tools/ubench/lds_rate.hip)

// OUT = 0: log-mel filterbank (Wav2LogFilterBank).  OUT = 1: MFCC -- the log-mel tile goes to LDS instead of
// HBM and a second (dense, tiny) MFMA GEMM applies the DCT (layers.py:716), then the lifter (:717-718).
// OUT = 2: (log-)spectrogram (Wav2Spec / Wav2LogSpec, layers.py:392-402, :461-473) -- every wave streams the
// power rows of its own four frames from LDS to HBM; no matrix-core stage.
template <int NROWS, int OUT>
__global__ __launch_bounds__(256, 4) void fft512b_kernel(const Fft512Params p) {
  constexpr bool MFCC = OUT == 1, SPEC = OUT == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  float* xs = smem;
  const v2* cwin = reinterpret_cast<const v2*>(smem + p.xs_floats);  // [NROWS][16]
  const v2* ctwp = cwin + NROWS * 16;                                // [16][16] row k1, column q
  const v2* ctws = ctwp + 256;                                       // [8][16] w = -i W_512^(q+16 k2)
  float* regions = smem + p.xs_floats + p.const_floats;

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
  const int N = p.N, shift = p.shift;
  const int nchunks = (p.xs_floats + 255) >> 8;  // 1 KiB LDS-DMA chunks covering the span buffer

  for (int i = tid; i < p.const_floats; i += 256) smem[p.xs_floats + i] = p.lds_consts[i];
  float* lm = regions + 4 * kBWaveRegion;            // [16 frames][lm_stride] log-mel tile (MFCC only)
  float* dctl = lm + kTileFrames * p.lm_stride;      // DCT A operands (MFCC only)
  if (MFCC)
    for (int i = tid; i < p.dct_floats; i += 256) dctl[i] = p.dct_consts[i];
  const WaveWork ww = p.work[wv];
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;
  // uniform (SGPR) base + 32-bit lane offset keeps global addresses out of the VGPR file
  const char* __restrict__ mel_base = reinterpret_cast<const char*>(p.mel_a) + (size_t)wv * kBMelVec * 64 * 16;

  // Stage the sample span of the tile starting at frame f0 into xs.  Interior tiles: LDS-DMA, each wave
  // moves 1 KiB chunks (lane i supplies the global address of its 16 bytes; the hardware writes
  // chunk base + 16 i).  The global side only needs dword alignment (measured: cuts packed back to
  // back at odd sample offsets run at the same rate and give identical results), so any packing works.
  // Tiles touching a cut edge (reflection / zero padding): scalar loads.
  auto stage_span = [&](int f0, unsigned lane16) {
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    if (j0 >= 0 && j0 + (int64_t)nchunks * 256 <= cd.num_samples) {
      const char* src = reinterpret_cast<const char*>(w + j0);  // uniform
      for (int ch = wv; ch < nchunks; ch += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)ch * 1024u + lane16)),
                                         (__attribute__((address_space(3))) void*)(xs + ch * 256), 16, 0, 0);
    } else {
      for (int i = tid; i < p.xs_floats; i += 256) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);  // the whole buffer: rows past N are read (and masked) too
    }
  };

  const int first_tile = fb * p.tiles_per_block;
  if (first_tile * kTileFrames < cd.num_frames) stage_span(first_tile * kTileFrames, (unsigned)lane * 16u);

  // (log-)spectrograms: both twiddle tables of this lane stay in registers for the whole kernel (as in kernel_fft512c.hpp; the other
  // output stages have no room for them under the 4-waves-per-SIMD register budget); read from the global copy of the tables: the LDS
  // copy is not yet visible here
  constexpr bool kRegTw = SPEC && NROWS <= 13;
  v2 twpreg[kRegTw ? 16 : 1], twsreg[kRegTw ? 8 : 1];
  if (kRegTw) {
    const v2* gtw = reinterpret_cast<const v2*>(p.lds_consts) + NROWS * 16 + (lane & 15);
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1) twpreg[k1] = gtw[k1 * 16];
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) twsreg[k2] = gtw[256 + k2 * 16];
  }
#ifdef HIPFEAT_PHASE_TIMERS
  unsigned long long hf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int t = 0; t < p.tiles_per_block; ++t) {
    const int f0 = (first_tile + t) * kTileFrames;
    if (f0 >= cd.num_frames) break;
    const int nf = min(kTileFrames, cd.num_frames - f0);

    // ---- S1: the span was requested one phase ago; wait for this wave's DMA, then for everyone's
    HF_T(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    HF_T(1);
    __syncthreads();
    HF_T(2);
    // Re-derive every per-lane address inside the loop from an opaque copy of the lane id: LICM
    // otherwise hoists ~25 loop-invariant LDS/global addresses into VGPRs for the whole kernel,
    // which costs a wave of occupancy; recomputing them is a handful of VALU ops per tile.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int q = lane_o & 15, g = lane_o >> 4;
    const unsigned lane16 = (unsigned)lane_o * 16u;
    float* myreg = regions + wv * kBWaveRegion;

    // ---- S3 ---------------------------------------------------------------------------------
    {
      const float* x = xs + (4 * wv + g) * shift + 2 * q;
      v2 z[16];
      v2 win[NROWS];
      v2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) { z[n1] = *reinterpret_cast<const v2*>(x + 32 * n1); HF_SEP(); }
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) { win[n1] = cwin[n1 * 16 + q]; HF_SEP(); }
      // samples at or beyond N (the frame length) are not part of the frame: the template instance may carry up to three
      // rows more than ceil(N / 32), so every row is checked (uniform test per row, lane mask only in the boundary rows)
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        if (32 * (n1 + 1) > N) {
          const int m0 = 32 * n1 + 2 * q;
          if (m0 >= N) z[n1].x = 0.f;
          if (m0 + 1 >= N) z[n1].y = 0.f;
        }
      }
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) sum2 += z[n1];
      float mu = 0.f;
      if (dc) mu = row16_sum(sum2.x + sum2.y) * inv_n;
      // previous sample of the first element of each pair: lane q-1's second element; for lane 0 it is
      // lane 15's second element of the previous row (fetched one row earlier with row_ror:1), and the
      // very first sample of the frame replicates itself (layers.py:166)
      float wrap = 0.f;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const v2 d = z[n1] - v2{mu, mu};
        const float dp = dpp_shr1_keep(n1 == 0 ? d.x : wrap, d.y);
        if (n1 + 1 < NROWS) wrap = dpp_mov<DPP_ROW_ROR1>(d.y);
        z[n1] = (d - v2{c, c} * v2{dp, d.x}) * win[n1];
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
          for (int r = 0; r < 8; ++r) tw[r] = twpreg[8 * h + r];
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) { tw[r] = ctwp[(8 * h + r) * 16 + q]; HF_SEP(); }
        }
#pragma unroll
        for (int r = (h == 0 ? 1 : 0); r < 8; ++r) a[8 * h + r] = cmul2(a[8 * h + r], tw[r]);
      }

      // exchange in two halves: rows k1 = 8h .. 8h+7 go through an 8-row block; lanes with
      // (q >> 3) == h then read "their" row (all n2) back
      float* exf = myreg + g * kBExFrameStride;
      v2 b[16];
      constexpr int RPP = 16 / kBExParts;  // rows per part
#pragma unroll
      for (int h = 0; h < kBExParts; ++h) {
#pragma unroll
        for (int r = 0; r < RPP; ++r) *reinterpret_cast<v2*>(exf + r * kBExRowStride + 2 * q) = a[RPP * h + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (h == 0 || q / RPP == h) {  // part 0: every lane reads (b is never undefined: nothing for hipcc to carry around the tile loop)
#pragma unroll
          for (int n2 = 0; n2 < 16; ++n2) { b[n2] = *reinterpret_cast<const v2*>(exf + (q % RPP) * kBExRowStride + 2 * n2); HF_SEP(); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      v2 Z[16];
      fft16(b, Z);

      float* prow = myreg + g * kPRowStride;
      float* pown = prow + q;
      float* ppar = prow + ((16 - q) & 15) + (q == 0 ? 16 : 0);
      if (q < 3) prow[257 + q] = 0.f;
      float t1[16];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].x);
        t1[2 * k2 + 1] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].y);
      }
      // second half of the lane map l -> (16 - l) % 16: shift right by one; lane 0 has no source and
      // keeps its own register (16 - k2) % 16 instead (its mirror partner is itself)
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_shr1_keep(Z[(16 - k2) & 15].x, t1[2 * k2]);
        t1[2 * k2 + 1] = dpp_shr1_keep(Z[(16 - k2) & 15].y, t1[2 * k2 + 1]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2 tw[4];  // split-step twiddles of 4 bin pairs per burst
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kRegTw) tw[r] = twsreg[4 * h + r];
          else {
            tw[r] = ctws[(4 * h + r) * 16 + q];
            HF_SEP();
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k2 = 4 * h + r;
          const v2 m = v2{t1[2 * k2], t1[2 * k2 + 1]};
          const v2 sp = m * HF_CJ + Z[k2];
          const v2 dm = m * HF_NCJ + Z[k2];
          const v2 tt = cmul2(dm, tw[r]);
          const v2 xp = sp + tt, xm = sp - tt;
          pown[16 * k2] = xp.x * xp.x + xp.y * xp.y;
          ppar[16 * (15 - k2)] = xm.x * xm.x + xm.y * xm.y;
        }
      }
      if (q == 0) prow[128] = 4.f * (Z[8].x * Z[8].x + Z[8].y * Z[8].y);
    }
    // the filter weights of this wave's band: requested now (the FFT registers are dead), so the L2
    // latency hides behind the barrier and the start of S5
    f32x4 ma[kBMelVec];
    if (!SPEC) {
#pragma unroll
      for (int i = 0; i < kBMelVec; ++i) ma[i] = *reinterpret_cast<const f32x4*>(mel_base + ((unsigned)i * 1024u + lane16));
    }
    HF_T(3);
    __syncthreads();
    HF_T(4);

    // ---- S5 ---------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);  // waves in the short MFMA/epilogue phase go first: +1.4 % (measured)
    if (SPEC) {
      {
        const int fn = f0 + kTileFrames;
        if (t + 1 < p.tiles_per_block && fn < cd.num_frames) stage_span(fn, lane16);
      }
      const int K = 257;
#pragma unroll
      for (int fr = 0; fr < 4; ++fr) {
        const int f = 4 * wv + fr;
        if (f < nf) {  // uniform
          const float* prow = myreg + fr * kPRowStride;
          float* orow = p.out + (cd.out_row + f0 + f) * p.out_stride;
          for (int col = lane_o; col < K; col += 64) {
            float v = prow[col];
            if (p.flags & F_FFT_MAG) v = sqrtf(v);
            if (p.flags & F_LOG_SPEC) v = fast_log(v + p.log_offset);
            orow[col] = v;
          }
        }
      }
    } else {
      const int j = lane_o & 15, kk = lane_o >> 4;
      const float* pb = regions + (j >> 2) * kBWaveRegion + (j & 3) * kPRowStride + 2 * kk;
      // P values of the first segment: ALL chunks are requested unconditionally (clamped, always valid
      // offsets), one chunk ahead of the MFMAs that use them; the first request goes out before anything
      // else in this phase so that its LDS latency overlaps the weight hand-over and the DMA issue.
      constexpr int CH = 4;
      constexpr int NCH = (kMaxGroups0 + CH - 1) / CH;
      v2 pv[NCH][CH];
      auto load_chunk = [&](int ci) {
#pragma unroll
        for (int i = 0; i < CH; ++i) { pv[ci][i] = *reinterpret_cast<const v2*>(pb + min(ww.bin0 + 8 * (ci * CH + i), kPRowStride - 8)); HF_SEP(); }
      };
      load_chunk(0);
      // gfx950 has ONE in-order counter for all vector-memory operations: take delivery of the weights
      // (requested before the barrier) BEFORE the span DMA is issued, otherwise their first use would
      // have to wait for the much slower HBM transfer queued behind them.
#pragma unroll
      for (int i = 0; i < kBMelVec; ++i) asm volatile("" : "+v"(ma[i]));
      // xs is dead until the next tile: stage the next span now (lands during the mel GEMM)
      {
        const int fn = f0 + kTileFrames;
        if (t + 1 < p.tiles_per_block && fn < cd.num_frames) stage_span(fn, lane16);
      }
      float* orow = p.out + (cd.out_row + f0 + j) * p.out_stride;
      const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
      auto epilogue = [&](const f32x4 acc, int tile) {
        const int m0 = tile * 16 + 4 * kk;
        f32x4 v;
        v.x = fast_log(fmaxf(acc.x, p.mel_floor));
        v.y = fast_log(fmaxf(acc.y, p.mel_floor));
        v.z = fast_log(fmaxf(acc.z, p.mel_floor));
        v.w = fast_log(fmaxf(acc.w, p.mel_floor));
        if (MFCC) {
          *reinterpret_cast<f32x4*>(lm + j * p.lm_stride + m0) = v;  // every column < 16 * tiles gets a finite value
        } else if (j < nf) {
          if (vec_ok && m0 + 3 < p.M) {
            *reinterpret_cast<f32x4*>(orow + m0) = v;
          } else {
            if (m0 + 0 < p.M) orow[m0 + 0] = v.x;
            if (m0 + 1 < p.M) orow[m0 + 1] = v.y;
            if (m0 + 2 < p.M) orow[m0 + 2] = v.z;
            if (m0 + 3 < p.M) orow[m0 + 3] = v.w;
          }
        }
      };
      auto wgt = [&](int step) -> float { return ma[step >> 2][step & 3]; };
      if (ww.ngroups0 > 0) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < NCH; ++ci) {
          if (ci + 1 < NCH) load_chunk(ci + 1);
          if (ci * CH < ww.ngroups0) {
#pragma unroll
            for (int i = 0; i < CH; ++i)
              if (ci * CH + i < kMaxGroups0) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (ci * CH + i)), pv[ci][i].x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (ci * CH + i) + 1), pv[ci][i].y, acc2, 0, 0, 0);
              }
          }
        }
        epilogue(acc + acc2, ww.tile0);
      }
      if (ww.ngroups1 > 0) {
        v2 pv[kMaxGroups1];
#pragma unroll
        for (int gi = 0; gi < kMaxGroups1; ++gi) pv[gi] = *reinterpret_cast<const v2*>(pb + min(ww.bin1 + 8 * gi, kPRowStride - 8));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gi = 0; gi < kMaxGroups1; ++gi) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (kMaxGroups0 + gi)), pv[gi].x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (kMaxGroups0 + gi) + 1), pv[gi].y, acc2, 0, 0, 0);
        }
        epilogue(acc + acc2, ww.tile1);
      }
      if (MFCC) {
        // ---- S6: cepstra = DCT^T x log-mel on the matrix cores, lifter, store -------------------
        __syncthreads();
        const int nct = (p.C + 15) >> 4;
        const float* lmb = lm + j * p.lm_stride + 2 * kk;  // B operand: frame j, mel slot kk
        for (int ct = wv; ct < nct; ct += 4) {
          const v2* da = reinterpret_cast<const v2*>(dctl) + (size_t)ct * p.dct_groups * 64 + lane_o;
          v2 av[kMaxDctGroups], bv[kMaxDctGroups];
#pragma unroll
          for (int gi = 0; gi < kMaxDctGroups; ++gi) {
            const int ge = min(gi, p.dct_groups - 1);
            av[gi] = da[ge * 64];
            bv[gi] = *reinterpret_cast<const v2*>(lmb + 8 * ge);
          }
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gi = 0; gi < kMaxDctGroups; ++gi) {
            if (gi < p.dct_groups) {
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].x, bv[gi].x, acc, 0, 0, 0);
              acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[gi].y, bv[gi].y, acc2, 0, 0, 0);
            }
          }
          const f32x4 r4 = acc + acc2;
          const int c0 = 16 * ct + 4 * kk;  // D[row = ceps 4*(lane>>4)+r][col = frame lane&15]
          if (j < nf) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (c0 + r < p.C) {
                // lifter values sit behind the DCT operands in LDS (ones when liftering is off)
                orow[c0 + r] = r4[r] * dctl[p.dct_floats - 64 + c0 + r];
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    HF_T(5);
    HF_ACC(0, t0, t1);  // wait for DMA / stores (vmcnt)
    HF_ACC(1, t1, t2);  // barrier 1
    HF_ACC(2, t2, t3);  // S3
    HF_ACC(3, t3, t4);  // barrier 2
    HF_ACC(4, t4, t5);  // S5
    HF_ACC(5, t0, t0 + 1);
    // the loop-top wait + barrier separates this tile's P reads from the next tile's exchange writes
  }
#ifdef HIPFEAT_PHASE_TIMERS
  if (lane == 0 && g_phase_buf) {
    unsigned long long* o = g_phase_buf + ((size_t)blockIdx.x * 4 + wv) * 8;
    for (int i = 0; i < 8; ++i) o[i] = hf_acc[i];
  }
#endif
}

}  // namespace hipfeat
