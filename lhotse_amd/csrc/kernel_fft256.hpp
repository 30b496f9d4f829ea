// fft256 fast path (8 kHz telephone audio: 25 ms / 10 ms frames = 200 / 80 samples; also <= 16 ms frames at 16 kHz).
// Same organisation as kernel_fft512b.hpp -- LDS-DMA span staging one tile ahead, register-resident FFT, power tile in
// LDS, banded f32 MFMA mel GEMM (+ DCT GEMM for MFCC), 4 workgroups per CU -- with the real FFT(256) = complex
// FFT(128) = 16 x 8 mapped on EIGHT lanes per frame:
//
//   lane q (0..7) of a frame holds column n2 = q: 16 complex points z[8 n1 + q], n1 = 0..15      (pass 1: fft16 in
//   registers, as in the 512 kernel); after the twiddle W_128^(q k1) and the LDS exchange lane q holds the TWO rows
//   k1 = q and k1 = q + 8 (8 points each)                                                          (pass 2: 2 x fft8);
//   its 16 outputs are the bins k = q + 8 j, j = 0..15 (j = 2 k2 + row).  The split step pairs bin k with 128 - k,
//   which lives in lane (8 - q) % 8, register 15 - j.  The two frames of a 16-lane DPP row are INTERLEAVED (even lanes
//   frame A, odd lanes frame B): the lane map q -> (8 - q) % 8 is then row_mirror, quad_perm xor 1, row_shr:2, and the
//   two lanes without a shift source (bound_ctrl off keeps `old`) are exactly the q = 0 lanes whose partner is
//   themselves -- no select instruction.
//
// A wave carries 8 frames, a workgroup tile 32 frames; the mel GEMM runs over two 16-frame MFMA column tiles.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"
#include "fft512_common.hpp"   // Fft512Params, WaveWork, kMaxGroups*, kMelARegs
#include "kernel_fft512b.hpp"  // kBMelVec, kMaxDctGroups

namespace hipfeat {

constexpr int k256TileFrames = 32;
constexpr int k256ExRowStride = 18;                          // dwords per exchange row (8 complex + 2 pad)
constexpr int k256ExFrameStride = 8 * k256ExRowStride;       // 144 (== 16 mod 64): 8 rows per half
constexpr int k256PRowStride = 132;                          // dwords per power row: 129 bins + pad (== 4 mod 64)
constexpr int k256WaveRegion = 8 * k256ExFrameStride;        // 1152 dwords per wave (>= 8 * 132 power rows); 32.6 KB per workgroup at 8 kHz -> 5 per CU

constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128, DPP_ROW_SHR2 = 0x112;
// frames interleaved in a 16-lane row (lane = 2 q + f): total over the 8 lanes of the same parity, in every one of them
__device__ __forceinline__ float row8i_sum(float v) {
  v += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(v);  // q ^ 1
  v += dpp_mov<DPP_ROW_ROR4>(v);          // q - 2
  v += dpp_mov<DPP_ROW_ROR8>(v);          // q - 4
  return v;
}
// lane (q, f) <- lane ((8 - q) % 8, f) of `v`; the q = 0 lanes (no source in the last move) keep `keep`
__device__ __forceinline__ float row8i_negate_index(float keep, float v) {
  const float m = dpp_mov<DPP_QUAD(1, 0, 3, 2)>(dpp_mov<DPP_ROW_MIRROR>(v));  // q -> 7 - q, same frame
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, m), DPP_ROW_SHR2, 0xF, 0xF, false));
}

// OUT = 0 fbank, 1 MFCC, 2 (log-)spectrogram -- see kernel_fft512b.hpp
template <int NROWS, int OUT>
// (16 live input rows: 2-3 VGPRs went to scratch under the 5-blocks register cap; those instances take 4 blocks per CU -- VERDICT r5 task 7)
__global__ __launch_bounds__(256, (NROWS > 13 ? 4 : 5)) void fft256_kernel(const Fft512Params p) {
  constexpr bool MFCC = OUT == 1, SPEC = OUT == 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  float* xs = smem;
  const v2* cwin = reinterpret_cast<const v2*>(smem + p.xs_floats);  // [NROWS][8]
  const v2* ctwp = cwin + NROWS * 8;                                 // [16][8] row k1, column q: W_128^(q k1)
  const v2* ctws = ctwp + 128;                                       // [8][8] w = -i W_256^(q + 8 j)
  const v2* ctwsp = ctws + 64;                                       // [8][8] (-w.y, w.x)
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
  const int nchunks = (p.xs_floats + 255) >> 8;

  for (int i = tid; i < p.const_floats; i += 256) smem[p.xs_floats + i] = p.lds_consts[i];
  float* lm = regions + 4 * k256WaveRegion;        // [32 frames][lm_stride] log-mel tile (MFCC only)
  float* dctl = lm + k256TileFrames * p.lm_stride;  // DCT A operands (MFCC only)
  if (MFCC)
    for (int i = tid; i < p.dct_floats; i += 256) dctl[i] = p.dct_consts[i];
  const WaveWork ww = p.work[wv];
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;
  const char* __restrict__ mel_base = reinterpret_cast<const char*>(p.mel_a) + (size_t)wv * kBMelVec * 64 * 16;

  auto stage_span = [&](int f0, unsigned lane16) {
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    if (j0 >= 0 && j0 + (int64_t)nchunks * 256 <= cd.num_samples) {
      const char* src = reinterpret_cast<const char*>(w + j0);  // uniform
      for (int ch = wv; ch < nchunks; ch += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((unsigned)ch * 1024u + lane16)),
                                         (__attribute__((address_space(3))) void*)(xs + ch * 256), 16, 0, 0);
    } else {
      for (int i = tid; i < p.xs_floats; i += 256) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);  // the whole buffer: rows past N are read (and masked or met by a zero window) too
    }
  };

  const int first_tile = fb * p.tiles_per_block;
  if (first_tile * k256TileFrames < cd.num_frames) stage_span(first_tile * k256TileFrames, (unsigned)lane * 16u);

  for (int t = 0; t < p.tiles_per_block; ++t) {
    const int f0 = (first_tile + t) * k256TileFrames;
    if (f0 >= cd.num_frames) break;
    const int nf = min(k256TileFrames, cd.num_frames - f0);

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's DMA chunks and the previous tile's stores
    __syncthreads();
    int lane_o = lane;  // opaque copy: keeps LICM from hoisting ~25 per-lane addresses into VGPRs (see kernel_fft512b.hpp)
    asm volatile("" : "+v"(lane_o));
    const int q = (lane_o & 15) >> 1, g = 2 * (lane_o >> 4) + (lane_o & 1);  // two frames interleaved per 16-lane row
    const unsigned lane16 = (unsigned)lane_o * 16u;
    float* myreg = regions + wv * k256WaveRegion;

    // ---- S3: 8 frames per wave ------------------------------------------------------------------------------------
    {
      const float* x = xs + (8 * wv + g) * shift + 2 * q;
      v2 z[16];
      v2 win[NROWS];
      float pv[NROWS];
      v2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) z[n1] = *reinterpret_cast<const v2*>(x + 16 * n1);
      // previous sample of the first element of each pair; the very first sample of the frame replicates itself
      // (layers.py:166).  Read from the span instead of a cross-lane move: 8-lane groups have no DPP rotate.
      pv[0] = x[q == 0 ? 0 : -1];
#pragma unroll
      for (int n1 = 1; n1 < NROWS; ++n1) pv[n1] = x[16 * n1 - 1];
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) win[n1] = cwin[n1 * 8 + q];
      // samples at or beyond N (the frame length) are not part of the frame: the template instance may carry up to three
      // rows more than ceil(N / 16), so every row is checked (uniform test per row, lane mask only in the boundary rows)
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        if (16 * (n1 + 1) > N) {
          const int m0 = 16 * n1 + 2 * q;
          if (m0 >= N) z[n1].x = 0.f;
          if (m0 + 1 >= N) z[n1].y = 0.f;
        }
      }
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) sum2 += z[n1];
      float mu = 0.f;
      if (dc) mu = row8i_sum(sum2.x + sum2.y) * inv_n;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const v2 d = z[n1] - v2{mu, mu};
        const float dp = pv[n1] - mu;
        z[n1] = (d - v2{c, c} * v2{dp, d.x}) * win[n1];
      }
#pragma unroll
      for (int n1 = NROWS; n1 < 16; ++n1) z[n1] = v2{0.f, 0.f};
      v2 a[16];
      fft16(z, a);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2 tw[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) tw[r] = ctwp[(8 * h + r) * 8 + q];
#pragma unroll
        for (int r = (h == 0 ? 1 : 0); r < 8; ++r) a[8 * h + r] = cmul(a[8 * h + r], tw[r]);
      }
      // exchange in two halves: rows k1 = 8h .. 8h+7 through an 8-row block per frame; lane q reads row q of each half
      float* exf = myreg + g * k256ExFrameStride;
      v2 b[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 8; ++r) *reinterpret_cast<v2*>(exf + r * k256ExRowStride + 2 * q) = a[8 * h + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) b[8 * h + n2] = *reinterpret_cast<const v2*>(exf + q * k256ExRowStride + 2 * n2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      v2 Z1[8], Z2[8];  // rows k1 = q and q + 8: bins q + 16 k2 and q + 8 + 16 k2
      fft8(b, Z1);
      fft8(b + 8, Z2);
      // ZZ[j] = bin q + 8 j: ZZ[2 k2] = Z1[k2], ZZ[2 k2 + 1] = Z2[k2]
      auto ZZ = [&](int j) -> v2 { return (j & 1) ? Z2[j >> 1] : Z1[j >> 1]; };

      float* prow = myreg + g * k256PRowStride;
      float* pown = prow + q;
      float* ppar = prow + ((8 - q) & 7) + (q == 0 ? 8 : 0);
      if (q < 3) prow[129 + q] = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        v2 tw[4], twq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tw[r] = ctws[(4 * h + r) * 8 + q];
          twq[r] = ctwsp[(4 * h + r) * 8 + q];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 4 * h + r;
          // partner bin 128 - k: lane (8 - q) % 8, register 15 - j; lane 0 of a frame: itself, register (16 - j) % 16
          const v2 src = ZZ(15 - j), own = ZZ((16 - j) & 15);
          const v2 m = v2{row8i_negate_index(own.x, src.x), row8i_negate_index(own.y, src.y)};
          const v2 zk = ZZ(j);
          const v2 sp = m * HF_CJ + zk;
          const v2 dm = m * HF_NCJ + zk;
          const v2 tt = cmulc(dm, tw[r], twq[r]);
          const v2 xp = sp + tt, xm = sp - tt;
          pown[8 * j] = xp.x * xp.x + xp.y * xp.y;
          ppar[8 * (15 - j)] = xm.x * xm.x + xm.y * xm.y;
        }
      }
      if (q == 0) prow[64] = 4.f * (Z1[4].x * Z1[4].x + Z1[4].y * Z1[4].y);
    }
    f32x4 ma[kBMelVec];
    if (!SPEC) {
#pragma unroll
      for (int i = 0; i < kBMelVec; ++i) ma[i] = *reinterpret_cast<const f32x4*>(mel_base + ((unsigned)i * 1024u + lane16));
    }
    __syncthreads();

    // ---- S5 -------------------------------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(1);
    if (SPEC) {
      {
        const int fn = f0 + k256TileFrames;
        if (t + 1 < p.tiles_per_block && fn < cd.num_frames) stage_span(fn, lane16);
      }
      const int K = 129;
#pragma unroll
      for (int fr = 0; fr < 8; ++fr) {
        const int f = 8 * wv + fr;
        if (f < nf) {  // uniform
          const float* prow = myreg + fr * k256PRowStride;
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
#pragma unroll
      for (int i = 0; i < kBMelVec; ++i) asm volatile("" : "+v"(ma[i]));  // weights delivered before the DMA is queued
      {
        const int fn = f0 + k256TileFrames;
        if (t + 1 < p.tiles_per_block && fn < cd.num_frames) stage_span(fn, lane16);
      }
      const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
      auto wgt = [&](int step) -> float { return ma[step >> 2][step & 3]; };
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {  // two MFMA column tiles of 16 frames
        const int fr = 16 * nt + j;      // frame of this lane's column
        const float* pb = regions + (fr >> 3) * k256WaveRegion + (fr & 7) * k256PRowStride + 2 * kk;
        float* orow = p.out + (cd.out_row + f0 + fr) * p.out_stride;
        auto epilogue = [&](const f32x4 acc, int tile) {
          const int m0 = tile * 16 + 4 * kk;
          f32x4 v;
          v.x = fast_log(fmaxf(acc.x, p.mel_floor));
          v.y = fast_log(fmaxf(acc.y, p.mel_floor));
          v.z = fast_log(fmaxf(acc.z, p.mel_floor));
          v.w = fast_log(fmaxf(acc.w, p.mel_floor));
          if (MFCC) {
            *reinterpret_cast<f32x4*>(lm + fr * p.lm_stride + m0) = v;
          } else if (fr < nf) {
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
        if (ww.ngroups0 > 0) {
          constexpr int CH = 4;
          constexpr int NCH = (kMaxGroups0 + CH - 1) / CH;
          v2 pv[NCH][CH];
          auto load_chunk = [&](int ci) {
#pragma unroll
            for (int i = 0; i < CH; ++i) pv[ci][i] = *reinterpret_cast<const v2*>(pb + min(ww.bin0 + 8 * (ci * CH + i), k256PRowStride - 8));
          };
          load_chunk(0);
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
          for (int gi = 0; gi < kMaxGroups1; ++gi) pv[gi] = *reinterpret_cast<const v2*>(pb + min(ww.bin1 + 8 * gi, k256PRowStride - 8));
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int gi = 0; gi < kMaxGroups1; ++gi) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (kMaxGroups0 + gi)), pv[gi].x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wgt(2 * (kMaxGroups0 + gi) + 1), pv[gi].y, acc2, 0, 0, 0);
          }
          epilogue(acc + acc2, ww.tile1);
        }
      }
      if (MFCC) {
        // ---- S6: cepstra = DCT^T x log-mel on the matrix cores, lifter, store ---------------------------------------
        __syncthreads();
        const int nct = (p.C + 15) >> 4;
        for (int wi = wv; wi < 2 * nct; wi += 4) {  // work item = (cepstral tile, frame column tile)
          const int ct = wi >> 1, nt = wi & 1;
          const int fr = 16 * nt + j;
          const float* lmb = lm + fr * p.lm_stride + 2 * kk;
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
          const int c0 = 16 * ct + 4 * kk;
          if (fr < nf) {
            float* orow = p.out + (cd.out_row + f0 + fr) * p.out_stride;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (c0 + r < p.C) orow[c0 + r] = r4[r] * dctl[p.dct_floats - 64 + c0 + r];
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
}

}  // namespace hipfeat
