// fft256 fast path, wave-autonomous (8 kHz telephone audio: 25 ms / 10 ms frames = 200 / 80 samples; also <= 16 ms frames at 16 kHz):
// log-mel filterbank (Wav2LogFilterBank, layers.py:565-578) in the organisation of kernel_fft512c.hpp -- a wave owns its frames from the
// samples in HBM to the stored rows, private LDS-DMA span one round ahead, constants of the critical path in registers, mel filterbank on
// v_mfma_f32_4x4x1_16B_f32, no workgroup barrier in the steady state -- with the FFT of kernel_fft256.hpp:
//
//   real FFT(256) = complex FFT(128) = 16 x 8 on EIGHT lanes per frame, eight frames per wave: lane q holds z[8 n1 + q], n1 = 0..15
//   (pass 1: fft16 in registers), times W_128^(q k1); after the LDS exchange (two halves of 8 rows) lane q holds the rows k1 = q and
//   q + 8 (pass 2: two fft8); its 16 outputs are the bins q + 8 j.  The two frames of a 16-lane DPP row are interleaved (even lanes frame
//   A, odd lanes frame B) so that the split-step partner map q -> (8 - q) % 8 is row_mirror, quad_perm xor 1, row_shr:2 with the two
//   lanes that have no shift source being exactly the self-partnered q = 0 lanes.
//   Eight power rows of 144 floats (129 bins) per wave; the 4 x 4 x 1 blocks take them four frames at a time: every filterbank step is
//   two matrix-core instructions with the same weights (frames 0..3, frames 4..7); 2 accumulator sets x 8 steps.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"
#include "kernel_fft256.hpp"   // row8i_sum, row8i_negate_index, exchange strides
#include "kernel_fft512c.hpp"  // Fft512cParams, HFC_SEP, mul24

namespace hipfeat {

constexpr int kDPRowStride = 144;                 // dwords per power row: 129 bins + pad (== 16 mod 64, as mel4_schedule.hpp assumes)
constexpr int kDRegion = 8 * kDPRowStride;        // 1152 dwords per wave = 8 exchange blocks of 8 x 18 = 8 power rows
constexpr int kDSets = 2, kDSteps = 8;            // accumulator sets x MFMA steps per set
constexpr int kDWaves = 8;                        // waves per workgroup

// NROWS: pass-1 rows (of 16 samples) that can hold samples; NFULL: rows known to lie entirely inside the frame (N >= 16 NFULL): no length masks there
template <int NROWS, int NFULL>
__global__ __launch_bounds__(64 * kDWaves, 4) void fft256c_kernel(const Fft512cParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  const v2* cwin = reinterpret_cast<const v2*>(smem);  // [NROWS][8]
  const v2* ctwp = cwin + NROWS * 8;                   // [16][8] row k1, column q: W_128^(q k1)
  const v2* ctws = ctwp + 128;                         // [8][8] w = -i W_256^(q + 8 j)
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
  const int N = p.N, shift = p.shift;

  for (int i = tid; i < p.shared_floats; i += 64 * kDWaves) smem[i] = p.shared_consts[i];
  float* xs = smem + p.shared_floats + wv * (p.xs_floats + kDRegion);
  float* myreg = xs + p.xs_floats;
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;

  auto stage_span = [&](int f0, unsigned lane4) {
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    if (j0 >= 0 && j0 + p.xs_floats <= cd.num_samples) {
      const char* src = reinterpret_cast<const char*>(w + j0);  // uniform
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
      for (int i = (int)(lane4 >> 2); i < p.xs_floats; i += 64) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
    }
  };

  const int first_frame = fb * p.frames_per_block + 8 * wv;  // the waves take the frame octets round-robin
  __syncthreads();  // the constant tables are in place (the only workgroup barrier of the kernel)
  if (first_frame < cd.num_frames) stage_span(first_frame, (unsigned)lane * 4u);

  // both twiddle tables of this lane live in registers for the whole kernel (kernel_fft512c.hpp: no LDS round trip in front of the
  // twiddle multiply and of the bursts of the split step)
  // (the pass twiddles only where they fit: with them the NROWS = 13 instance spills 4 registers, the NROWS = 16 one 17)
  constexpr bool kRegTwp = false;
  v2 twpreg[kRegTwp ? 16 : 1], twsreg[8];
  {
    const int q0 = (lane & 15) >> 1;
    if (kRegTwp) {
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) twpreg[k1] = ctwp[k1 * 8 + q0];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) twsreg[j] = ctws[j * 8 + q0];
  }

  for (int r = 0; r < p.rounds; ++r) {
    const int f0 = first_frame + 8 * kDWaves * r;
    if (f0 >= cd.num_frames) break;
    const int nf = min(8, cd.num_frames - f0);

    if (r == 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); later rounds waited before their predecessor's stores
    int lane_o = lane;  // opaque copy: keeps LICM from pinning per-lane addresses in VGPRs for the whole kernel
    asm volatile("" : "+v"(lane_o));
    const int q = (lane_o & 15) >> 1, g = 2 * (lane_o >> 4) + (lane_o & 1);  // two frames interleaved per 16-lane row
    int early_poff[kDSets];

    {
      v2 Z1[8], Z2[8];  // rows k1 = q and q + 8: bins q + 16 k2 and q + 8 + 16 k2
      {
        const float* x = xs + mul24(g, shift) + 2 * q;
        v2 z[16];
        v2 win[NROWS];
        float pv[NROWS];  // left neighbour of each pair's first sample (the frame's first sample replicates itself, layers.py:166)
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          z[n1] = *reinterpret_cast<const v2*>(x + 16 * n1);
          HFC_SEP();
        }
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) {
          win[n1] = cwin[n1 * 8 + q];
          HFC_SEP();
        }
#pragma unroll
        for (int n1 = 0; n1 < NROWS; ++n1) pv[n1] = n1 == 0 ? x[q == 0 ? 0 : -1] : x[16 * n1 - 1];
        // the samples are in flight to registers; once they have arrived the buffer is free for the next round's span
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (r + 1 < p.rounds && f0 + 8 * kDWaves < cd.num_frames) stage_span(f0 + 8 * kDWaves, (unsigned)lane_o * 4u);

#pragma unroll
        for (int n1 = NFULL; n1 < NROWS; ++n1) {
          if (16 * (n1 + 1) > N) {  // samples at or beyond N are not part of the frame
            const int m0 = 16 * n1 + 2 * q;
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
          mu = row8i_sum(sum2.x + sum2.y) * inv_n;
        }
        {  // y[n] = (x[n] - mu) - c (x[n-1] - mu) = x[n] - c x[n-1] - (1 - c) mu, times the window
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
        for (int n1 = NROWS; n1 < 16; ++n1) z[n1] = v2{0.f, 0.f};
        v2 a[16];
        fft16(z, a);
        if (kRegTwp) {
#pragma unroll
          for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul2(a[k1], twpreg[k1]);
        } else {  // from LDS in two bursts of 8 (one latency exposure each)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            v2 tw[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
              tw[rr] = ctwp[(8 * h + rr) * 8 + q];
              HFC_SEP();
            }
#pragma unroll
            for (int rr = (h == 0 ? 1 : 0); rr < 8; ++rr) a[8 * h + rr] = cmul2(a[8 * h + rr], tw[rr]);
          }
        }
        // exchange in two halves: rows k1 = 8h .. 8h+7 through an 8-row block per frame; lane q reads row q of each half
        float* exf = myreg + mul24(g, k256ExFrameStride);
        v2 b[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) *reinterpret_cast<v2*>(exf + rr * k256ExRowStride + 2 * q) = a[8 * h + rr];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int n2 = 0; n2 < 8; ++n2) {
            b[8 * h + n2] = *reinterpret_cast<const v2*>(exf + mul24(q, k256ExRowStride) + 2 * n2);
            HFC_SEP();
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        // the power-row offsets of this lane's filterbank slots, a phase early (no dependent LDS look-up in front of the operand reads)
#pragma unroll
        for (int s = 0; s < kDSets; ++s) early_poff[s] = __builtin_bit_cast(int, ltab[s * 256 + 4 * lane_o]);
        fft8(b, Z1);
        fft8(b + 8, Z2);
      }
      // ZZ[j] = bin q + 8 j: ZZ[2 k2] = Z1[k2], ZZ[2 k2 + 1] = Z2[k2]
      auto ZZ = [&](int j) -> v2 { return (j & 1) ? Z2[j >> 1] : Z1[j >> 1]; };
      float* prow = myreg + mul24(g, kDPRowStride);
      float* pown = prow + q;
      float* ppar = prow + ((8 - q) & 7) + (q == 0 ? 8 : 0);
      prow[129 + q] = 0.f;  // the padding a slot may read past bin 128 (weight 0) must be finite: bins 129 .. 143
      if (q < 7) prow[137 + q] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // partner bin 128 - k: lane (8 - q) % 8, register 15 - j; lane 0 of a frame: itself, register (16 - j) % 16
        const v2 src = ZZ(15 - j), own = ZZ((16 - j) & 15);
        const v2 m = v2{row8i_negate_index(own.x, src.x), row8i_negate_index(own.y, src.y)};
        const v2 zk = ZZ(j);
        const v2 sp = m * HF_CJ + zk;
        const v2 dm = m * HF_NCJ + zk;
        const v2 tt = cmul2(dm, twsreg[j]);
        const v2 xp = sp + tt, xm = sp - tt;
        pown[8 * j] = xp.x * xp.x + xp.y * xp.y;
        ppar[8 * (15 - j)] = xm.x * xm.x + xm.y * xm.y;
      }
      if (q == 0) prow[64] = 4.f * (Z1[4].x * Z1[4].x + Z1[4].y * Z1[4].y);
    }
    // the wave's eight power rows are complete once its own (in-order) LDS queue has drained
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // the next round's span (requested at the start of this round) must have landed before this round's stores join the same
    // in-order vmcnt queue: waiting here instead of at the top of the next round never waits for the stores
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    // ---- mel filterbank on the matrix cores: 4 frames x 4 filters x 1 bin per block, frames 0..3 and 4..7 with the same weights ----
    float* orow = p.out + (cd.out_row + f0) * p.out_stride;
    f32x4 av[kDSets][2][kDSteps / 4], bv[kDSets][kDSteps / 4];
#pragma unroll
    for (int s = 0; s < kDSets; ++s) {
      const float* pa = myreg + early_poff[s];
      const float* wb = wtab + s * (kDSteps * 64) + 4 * lane_o;
#pragma unroll
      for (int c4 = 0; c4 < kDSteps / 4; ++c4) {
        av[s][0][c4] = *reinterpret_cast<const f32x4*>(pa + 4 * c4);
        av[s][1][c4] = *reinterpret_cast<const f32x4*>(pa + 4 * kDPRowStride + 4 * c4);
        bv[s][c4] = *reinterpret_cast<const f32x4*>(wb + c4 * 256);
      }
    }
#pragma unroll
    for (int s = 0; s < kDSets; ++s) {
      const float* lt = ltab + s * 256 + 4 * lane_o;
      const int col = __builtin_bit_cast(int, lt[1]);
      const float m4 = lt[2], m8 = lt[3];
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < kDSteps / 4; ++c4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s][0][c4][i], bv[s][c4][i], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[s][1][c4][i], bv[s][c4][i], acc1, 0, 0, 0);
        }
      }
      // (fft_common.hpp::mel4_reduce_floor / mel4_store: fused DPP multiply-adds, one lane mask per set instead of one per value)
      float val[8];
      {
        float v4[4];
        mel4_reduce_floor(acc0, m4, m8, p.mel_floor, v4);
#pragma unroll
        for (int i = 0; i < 4; ++i) val[i] = fast_log(v4[i]);
        mel4_reduce_floor(acc1, m4, m8, p.mel_floor, v4);
#pragma unroll
        for (int i = 0; i < 4; ++i) val[4 + i] = fast_log(v4[i]);
      }
      if (col < p.M) mel4_store_saddr<8>(orow, (unsigned)col, p.out_stride, nf, val);
    }
    // the next round's exchange writes follow this round's power-row reads in the wave's own LDS queue (in order)
  }
}

}  // namespace hipfeat
