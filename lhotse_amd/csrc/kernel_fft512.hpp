// Fast path: fft_length == 512 (16 kHz / 25 ms and friends), fbank output.
//
// Work decomposition (wave64, 4 waves per workgroup, one tile = 16 consecutive frames of one cut):
//
//   S1  all 256 lanes: coalesced load of the tile's sample span (15*shift + N floats, reflected
//       at the cut edges) HBM -> LDS.  Each sample is read from HBM once per tile (frames overlap
//       60 %, so the LDS copy is what the 16 frames share).
//   S3  every 16-lane group owns ONE frame (4 frames per wave, 16 per workgroup) and keeps its
//       whole 512-point real FFT in registers:
//         real FFT(512) = complex FFT(256) of z[n] = y[2n] + i y[2n+1], 256 = 16 x 16
//         pass 1: lane q holds z[16 n1 + q] (n1 = 0..15) -> 16-point FFT over n1 in registers,
//                 times W_256^(q k1)
//         exchange through a wave-private LDS region (the only LDS round trip of the FFT)
//         pass 2: lane k1 holds all n2 -> 16-point FFT in registers -> Z[k1 + 16 k2]
//         split step X[k] = E[k] + W_512^k O[k]: the mirror bin Z[256-k] lives in lane 16-k1 and is
//                 fetched with two DPP moves (row_mirror, row_ror:1) -- no LDS
//         |X|^2 -> LDS power tile P[16 frames][257 bins]
//       DC removal, pre-emphasis (previous sample via DPP row_ror) and the window are applied on
//       the way in (layers.py:155-170); the window is pre-scaled by 1/2 on the host, which makes
//       the split step's E/O halving exact and free.
//   S5  mel filterbank as a banded f32 MFMA GEMM (v_mfma_f32_16x16x4_f32): D[mel][frame] +=
//       W^T[mel][bin] * P^T[bin][frame] over the non-zero band of each 16-mel tile only
//       (72 k-steps instead of 5 x 65 dense for 80 mels); log(max(.,eps)) epilogue and 16-byte stores.
//
// f32 MFMA is a k-ordered fmaf chain, so the mel sums are bit-identical to the generic kernel's
// banded dot products when the bins are visited in ascending order.
#pragma once
#include "common.hpp"

namespace hipfeat {

constexpr int kTileFrames = 16;
constexpr int kExRowStride = 36;                   // dwords per exchange row (16 complex + 4 pad)
constexpr int kExFrameStride = 16 * kExRowStride;  // 576
constexpr int kWaveRegion = 4 * kExFrameStride + 16;  // 2320 dwords per wave (== 16 mod 64)
constexpr int kPRowStride = 260;                   // dwords per power row (== 4 mod 64)
constexpr int kMaxGroups = 16;                     // 8-bin groups of MFMA work per wave

struct WaveWork {  // mel work of one wave: up to two (tile, band) segments
  int32_t tile0, bin0, ngroups0, tile1, bin1, ngroups1, pad0, pad1;
};

struct Fft512Params {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window_half;  // [512]  0.5 * window, zero beyond N
  const float2* tw_pass;     // [16][16]  W_256^(q*k1)
  const float2* tw_split;    // [256]     -i * W_512^k
  const float* mel_a;        // [4 waves][2*kMaxGroups steps][64 lanes] MFMA A operands
  const WaveWork* work;      // [4]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, tiles_per_block;
  int32_t N, shift, npad_left, M, flags;
  float preemph, mel_floor;
  int32_t xs_floats;  // LDS floats reserved for the sample span
};

// ---- DPP helpers (row = 16 lanes) -------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int DPP_ROW_ROR1 = 0x121;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;

__device__ __forceinline__ float row16_sum(float v) {  // every lane of the row gets the same total
  v += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(v);
  v += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}
// lane l <- lane (16 - l) % 16 of the same row
__device__ __forceinline__ float row16_negate_index(float v) { return dpp_mov<DPP_ROW_ROR1>(dpp_mov<DPP_ROW_MIRROR>(v)); }

// ---- 16-point complex FFT in registers (radix-4 x radix-4, natural order in and out) -----------
__device__ __forceinline__ void cmul(float& r, float& i, float wr, float wi) {
  const float t = r * wr - i * wi;
  i = r * wi + i * wr;
  r = t;
}

__device__ __forceinline__ void fft16(const float (&xr)[16], const float (&xi)[16], float (&Xr)[16], float (&Xi)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  float yr[16], yi[16];  // y[4*m + n]
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const float s0r = xr[n] + xr[n + 8], s0i = xi[n] + xi[n + 8];
    const float s1r = xr[n] - xr[n + 8], s1i = xi[n] - xi[n + 8];
    const float s2r = xr[n + 4] + xr[n + 12], s2i = xi[n + 4] + xi[n + 12];
    const float s3r = xr[n + 4] - xr[n + 12], s3i = xi[n + 4] - xi[n + 12];
    yr[n] = s0r + s2r;
    yi[n] = s0i + s2i;
    yr[8 + n] = s0r - s2r;
    yi[8 + n] = s0i - s2i;
    yr[4 + n] = s1r + s3i;  // s1 - i*s3
    yi[4 + n] = s1i - s3r;
    yr[12 + n] = s1r - s3i;  // s1 + i*s3
    yi[12 + n] = s1i + s3r;
  }
  // twiddles W16^(n*m)
  cmul(yr[4 + 1], yi[4 + 1], C1, -S1);    // n=1 m=1: W^1
  cmul(yr[8 + 1], yi[8 + 1], R2, -R2);    // n=1 m=2: W^2
  cmul(yr[12 + 1], yi[12 + 1], S1, -C1);  // n=1 m=3: W^3
  cmul(yr[4 + 2], yi[4 + 2], R2, -R2);    // n=2 m=1: W^2
  {                                        // n=2 m=2: W^4 = -i
    const float t = yr[8 + 2];
    yr[8 + 2] = yi[8 + 2];
    yi[8 + 2] = -t;
  }
  cmul(yr[12 + 2], yi[12 + 2], -R2, -R2);  // n=2 m=3: W^6
  cmul(yr[4 + 3], yi[4 + 3], S1, -C1);     // n=3 m=1: W^3
  cmul(yr[8 + 3], yi[8 + 3], -R2, -R2);    // n=3 m=2: W^6
  cmul(yr[12 + 3], yi[12 + 3], -C1, S1);   // n=3 m=3: W^9
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float ar = yr[4 * m], ai = yi[4 * m], br = yr[4 * m + 1], bi = yi[4 * m + 1];
    const float cr = yr[4 * m + 2], ci = yi[4 * m + 2], dr = yr[4 * m + 3], di = yi[4 * m + 3];
    const float s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
    const float s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
    Xr[m] = s0r + s2r;  // k' = 0
    Xi[m] = s0i + s2i;
    Xr[m + 4] = s1r + s3i;  // k' = 1
    Xi[m + 4] = s1i - s3r;
    Xr[m + 8] = s0r - s2r;  // k' = 2
    Xi[m + 8] = s0i - s2i;
    Xr[m + 12] = s1r - s3i;  // k' = 3
    Xi[m + 12] = s1i + s3r;
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NROWS = ceil(N / 32): rows n1 >= NROWS of pass 1 are structurally zero (zero padding to 512).
#ifndef HIPFEAT_FFT512_WAVES_PER_SIMD
#define HIPFEAT_FFT512_WAVES_PER_SIMD 2
#endif
template <int NROWS>
__global__ __launch_bounds__(256, HIPFEAT_FFT512_WAVES_PER_SIMD) void fft512_fbank_kernel(const Fft512Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* regions = smem + p.xs_floats;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave index, provably uniform
  const int q = lane & 15, g = lane >> 4;

  // ---- which cut / which frames ---------------------------------------------------------
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
  const int N = p.N, shift = p.shift;
  const int span = (kTileFrames - 1) * shift + N;

  // ---- per-lane constants, resident in registers for the whole workgroup -----------------
  float win0[NROWS], win1[NROWS];
#pragma unroll
  for (int n1 = 0; n1 < NROWS; ++n1) {
    const float2 t = *reinterpret_cast<const float2*>(p.window_half + 32 * n1 + 2 * q);
    win0[n1] = t.x;
    win1[n1] = t.y;
  }
  float twr[16], twi[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) {
    const float2 t = p.tw_pass[q * 16 + k1];
    twr[k1] = t.x;
    twi[k1] = t.y;
  }
  float wsr[16], wsi[16];
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) {
    const float2 t = p.tw_split[q + 16 * k2];
    wsr[k2] = t.x;
    wsi[k2] = t.y;
  }
  const WaveWork ww = p.work[wv];
  float mela[2 * kMaxGroups];
#pragma unroll
  for (int s = 0; s < 2 * kMaxGroups; ++s) mela[s] = p.mel_a[(wv * 2 * kMaxGroups + s) * 64 + lane];

  float* myreg = regions + wv * kWaveRegion;
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n_is_div = (float)N;
  const float c = p.preemph;

  for (int t = 0; t < p.tiles_per_block; ++t) {
    const int f0 = (fb * p.tiles_per_block + t) * kTileFrames;
    if (f0 >= cd.num_frames) break;  // uniform across the workgroup
    const int nf = min(kTileFrames, cd.num_frames - f0);

    // ---- S1: sample span -> LDS ---------------------------------------------------------
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    if (j0 >= 0 && j0 + span <= cd.num_samples) {
      for (int i = tid; i < span; i += 256) xs[i] = w[j0 + i];
    } else {
      for (int i = tid; i < span; i += 256) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
    }
    __syncthreads();

    // ---- S3: one frame per 16 lanes -------------------------------------------------------
    {
      const float* x = xs + (4 * wv + g) * shift + 2 * q;
      float zr[16], zi[16];
      float s = 0.f;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        float2 v = *reinterpret_cast<const float2*>(x + 32 * n1);
        const int m0 = 32 * n1 + 2 * q;
        if (n1 == NROWS - 1) {  // only the last row can cross N
          if (m0 >= N) v.x = 0.f;
          if (m0 + 1 >= N) v.y = 0.f;
        }
        zr[n1] = v.x;
        zi[n1] = v.y;
        s += v.x + v.y;
      }
      float mu = 0.f;
      if (dc) mu = row16_sum(s) / inv_n_is_div;
      float tprev = 0.f;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const float d0 = zr[n1] - mu, d1 = zi[n1] - mu;
        const float tcur = dpp_mov<DPP_ROW_ROR1>(d1);  // lane q <- lane q-1 (lane 0 <- lane 15)
        // previous sample of the first element: lane q-1's second element; for lane 0 it is
        // lane 15's second element of the previous row; the very first sample replicates itself
        const float dp = (q == 0) ? (n1 == 0 ? d0 : tprev) : tcur;
        tprev = tcur;
        zr[n1] = (d0 - c * dp) * win0[n1];
        zi[n1] = (d1 - c * d0) * win1[n1];
      }
#pragma unroll
      for (int n1 = NROWS; n1 < 16; ++n1) {
        zr[n1] = 0.f;
        zi[n1] = 0.f;
      }
      float ar[16], ai[16];
      fft16(zr, zi, ar, ai);
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) cmul(ar[k1], ai[k1], twr[k1], twi[k1]);
      // exchange: row k1 of this frame's block receives this lane's A[k1] at column q
      float* exf = myreg + g * kExFrameStride;
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) *reinterpret_cast<float2*>(exf + k1 * kExRowStride + 2 * q) = make_float2(ar[k1], ai[k1]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float br[16], bi[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(exf + q * kExRowStride + 4 * j);
        br[2 * j] = v.x;
        bi[2 * j] = v.y;
        br[2 * j + 1] = v.z;
        bi[2 * j + 1] = v.w;
      }
      float Zr[16], Zi[16];
      fft16(br, bi, Zr, Zi);
      // all lanes must have finished reading the exchange rows before the power rows overwrite them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // mirror bins: Z[256 - k] for k = q + 16 k2 is lane (16-q)%16, register 15-k2 (q != 0)
      float mr[16], mi[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        mr[k2] = row16_negate_index(Zr[15 - k2]);
        mi[k2] = row16_negate_index(Zi[15 - k2]);
      }
      float* prow = myreg + g * kPRowStride + q;
      if (q < 3) prow[257] = 0.f;  // pad columns 257..259 are read (with zero weight) by the last k-group
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        // lane 0 pairs bin 16 k2 with bin 16 (16 - k2): its own register (16 - k2) % 16
        const float b_r = (q == 0) ? Zr[(16 - k2) & 15] : mr[k2];
        const float b_i = (q == 0) ? Zi[(16 - k2) & 15] : mi[k2];
        const float sr = Zr[k2] + b_r, si = Zi[k2] - b_i;  // a + conj(b)
        const float dr = Zr[k2] - b_r, di = Zi[k2] + b_i;  // a - conj(b)
        const float tr = wsr[k2] * dr - wsi[k2] * di, ti = wsr[k2] * di + wsi[k2] * dr;
        const float xr_ = sr + tr, xi_ = si + ti;
        prow[16 * k2] = xr_ * xr_ + xi_ * xi_;
        if (k2 == 0 && q == 0) {
          const float nr = sr - tr, ni = si - ti;  // Nyquist bin 256
          prow[256] = nr * nr + ni * ni;
        }
      }
    }
    __syncthreads();

    // ---- S5: banded mel GEMM on the matrix cores + log + store ---------------------------
    {
      const int j = lane & 15, kk = lane >> 4;  // B operand: frame j, bin slot kk
      const float* pb = regions + (j >> 2) * kWaveRegion + (j & 3) * kPRowStride + 2 * kk;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int gi = 0; gi < kMaxGroups; ++gi) {
        if (gi < ww.ngroups0) {
          const float2 pv = *reinterpret_cast<const float2*>(pb + ww.bin0 + 8 * gi);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * gi], pv.x, acc0, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * gi + 1], pv.y, acc0, 0, 0, 0);
        } else if (gi < ww.ngroups0 + ww.ngroups1) {
          const float2 pv = *reinterpret_cast<const float2*>(pb + ww.bin1 + 8 * (gi - ww.ngroups0));
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * gi], pv.x, acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * gi + 1], pv.y, acc1, 0, 0, 0);
        }
      }
      // D[row = mel 4*(lane>>4)+r][col = frame lane&15]
      if (j < nf) {
        float* orow = p.out + (cd.out_row + f0 + j) * p.out_stride;
        const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm) {
          const int ng = sgm == 0 ? ww.ngroups0 : ww.ngroups1;
          if (ng <= 0) continue;
          const int m0 = (sgm == 0 ? ww.tile0 : ww.tile1) * 16 + 4 * kk;
          const f32x4 a = sgm == 0 ? acc0 : acc1;
          f32x4 v;
          v.x = logf(fmaxf(a.x, p.mel_floor));
          v.y = logf(fmaxf(a.y, p.mel_floor));
          v.z = logf(fmaxf(a.z, p.mel_floor));
          v.w = logf(fmaxf(a.w, p.mel_floor));
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
    // the next tile's S1 only writes xs; its first barrier separates this tile's P reads from
    // the next tile's exchange writes
  }
}

}  // namespace hipfeat
