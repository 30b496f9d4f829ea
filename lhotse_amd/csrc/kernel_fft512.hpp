// Fast path: fft_length == 512 (16 kHz / 25 ms and friends), fbank output.
//
// Work decomposition (wave64, 4 waves per workgroup, one tile = 16 consecutive frames of one cut):
//
//   S1  all 256 lanes: the tile's sample span (15*shift + N floats, reflected at the cut edges)
//       goes HBM -> registers -> LDS with 16-byte loads; the NEXT tile's span is requested right
//       after, so its HBM latency hides under this tile's arithmetic.  Each sample is read from
//       HBM once per tile (frames overlap 60 %: the LDS copy is what the 16 frames share).
//   S3  every 16-lane group owns ONE frame (4 frames per wave, 16 per workgroup) and keeps its
//       whole 512-point real FFT in registers, as packed (re, im) pairs (v_pk_* arithmetic):
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
//       the split step's E/O halving exact and free.  Window and twiddles live in LDS.
//   S5  mel filterbank as a banded f32 MFMA GEMM (v_mfma_f32_16x16x4_f32): D[mel][frame] +=
//       W^T[mel][bin] * P^T[bin][frame] over the non-zero band of each 16-mel tile only
//       (72 k-steps instead of 5 x 65 dense for 80 mels); log(max(.,eps)) epilogue and 16-byte stores.
//       The A operands (filter weights) of a wave's band stay in its registers for the whole launch.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

constexpr int kTileFrames = 16;
// Exchange rows: 16 complex + 2 pad dwords.  With a row stride of 34 and a frame stride of 544
// (== 32 mod 64) the 8-byte writes (16 lanes contiguous) and the 8-byte reads (lane q walks row q;
// 2 x 16 lanes per LDS cycle) are both bank-conflict free.
constexpr int kExRowStride = 34;
constexpr int kExFrameStride = 16 * kExRowStride;     // 544
#ifdef HIPFEAT_EXPERIMENT_REGION  // perf-only experiment (results are garbage): smaller LDS footprint
constexpr int kWaveRegion = HIPFEAT_EXPERIMENT_REGION;
#else
constexpr int kWaveRegion = 4 * kExFrameStride + 16;  // 2192 dwords per wave (== 16 mod 64)
#endif
constexpr int kPRowStride = 260;                      // dwords per power row (== 4 mod 64)
#ifdef HF_X_MAXG0
constexpr int kMaxGroups0 = HF_X_MAXG0;
#else
constexpr int kMaxGroups0 = 20;
#endif                       // 8-bin MFMA groups of a wave's first / second mel tile
constexpr int kMaxGroups1 = 4;
constexpr int kMelARegs = 2 * (kMaxGroups0 + kMaxGroups1);
constexpr int kPrefetch = 3;                          // float4 per lane per tile (256 * 3 * 4 >= span)

struct WaveWork {  // mel work of one wave: up to two (tile, band) segments
  int32_t tile0, bin0, ngroups0, tile1, bin1, ngroups1, pad0, pad1;
};

struct Fft512Params {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* lds_consts;  // [nrows][16] v2 window/2 | [16][16] v2 W_256^(q k1) (row k1) | [16][16] v2 -i W_512^(q+16 k2) (row k2)
  const float* mel_a;       // [4 waves][kMelARegs steps][64 lanes] MFMA A operands
  const WaveWork* work;     // [4]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, tiles_per_block;
  int32_t N, shift, npad_left, M, flags;
  float preemph, mel_floor, log_offset;
  int32_t xs_floats;     // LDS floats reserved for the sample span
  int32_t const_floats;  // LDS floats of the constant block
  // MFCC stage (kernel b only): DCT as a second MFMA GEMM over the log-mel tile
  const float* dct_consts;  // [ceps tiles][mel groups of 8][64 lanes][2] MFMA A operands (DCT^T) | [64] lifter; copied to LDS
  int32_t C, lm_stride, dct_groups, dct_floats;
};

// Optional phase timers (experiment builds only): per-phase shader-clock totals over all waves.
#ifdef HIPFEAT_PHASE_TIMERS
__device__ unsigned long long* g_phase_buf;  // [grid * 4 waves][8], written once per wave (no atomics)
#define HF_T(i) const unsigned long long t##i = __builtin_readcyclecounter()
#define HF_ACC(slot, a, b) hf_acc[slot] += (unsigned long long)((b) - (a))
#ifdef HIPFEAT_PHASE_TIMERS2
#define HF_U(i) u##i = __builtin_readcyclecounter()
#else
#define HF_U(i)
#endif
#else
#define HF_T(i)
#define HF_U(i)
#define HF_ACC(slot, a, b)
#endif

#ifndef HIPFEAT_HOIST_TW
#define HIPFEAT_HOIST_TW 0
#endif
#ifndef HIPFEAT_S5_CHUNK
#define HIPFEAT_S5_CHUNK 4
#endif
#ifndef HIPFEAT_FFT512_WAVES_PER_SIMD
#define HIPFEAT_FFT512_WAVES_PER_SIMD 3
#endif

// NROWS = ceil(N / 32): rows n1 >= NROWS of pass 1 are structurally zero (zero padding to 512).
template <int NROWS>
__global__ __launch_bounds__(256, HIPFEAT_FFT512_WAVES_PER_SIMD) void fft512_fbank_kernel(const Fft512Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  const v2* cwin = reinterpret_cast<const v2*>(smem + p.xs_floats);  // [NROWS][16]
  const v2* ctwp = cwin + NROWS * 16;                                // [16][16] row k1, column q
  const v2* ctws = ctwp + 256;                                       // [8][16] row k2 < 8, column q: w = -i W_512^(q+16 k2)
  const v2* ctwsp = ctws + 128;                                      // [8][16] (-w.y, w.x)
  float* regions = smem + p.xs_floats + p.const_floats;

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
  // 16-byte path: every tile start is 16-byte aligned and the span fits the prefetch registers
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(w) & 15) == 0) && ((shift & 3) == 0) && ((p.npad_left & 3) == 0) &&
                         (p.xs_floats <= kPrefetch * 1024);

  // ---- constants: FFT tables -> LDS, this wave's mel weights -> registers -----------------
  for (int i = tid; i < p.const_floats; i += 256) smem[p.xs_floats + i] = p.lds_consts[i];
  const WaveWork ww = p.work[wv];
  float mela[kMelARegs];
#pragma unroll
  for (int s = 0; s < kMelARegs; ++s) mela[s] = p.mel_a[(wv * kMelARegs + s) * 64 + lane];

  float* myreg = regions + wv * kWaveRegion;
  const bool dc = (p.flags & F_REMOVE_DC) != 0;
  const float inv_n = 1.0f / (float)N;
  const float c = p.preemph;

  // ---- tile loop with register prefetch of the next span ---------------------------------
  const int first_tile = fb * p.tiles_per_block;
  f32x4 pre[kPrefetch];
  auto tile_is_interior = [&](int f0) -> bool {
    const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
    return aligned16 && j0 >= 0 && j0 + kPrefetch * 1024 <= cd.num_samples;
  };
  auto issue_prefetch = [&](int f0) {
    const float* src = w + ((int64_t)f0 * shift - p.npad_left);
#pragma unroll
    for (int i = 0; i < kPrefetch; ++i) pre[i] = *reinterpret_cast<const f32x4*>(src + 4 * (tid + 256 * i));
  };
  bool have_pre = false;
  if (first_tile * kTileFrames < cd.num_frames && tile_is_interior(first_tile * kTileFrames)) {
    issue_prefetch(first_tile * kTileFrames);
    have_pre = true;
  }

#ifdef HIPFEAT_PHASE_TIMERS
  unsigned long long hf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int t = 0; t < p.tiles_per_block; ++t) {
    const int f0 = (first_tile + t) * kTileFrames;
    if (f0 >= cd.num_frames) break;  // uniform across the workgroup
    const int nf = min(kTileFrames, cd.num_frames - f0);

    // ---- S1: sample span -> LDS ---------------------------------------------------------
    HF_T(0);
    if (have_pre) {
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) {
        const int e = 4 * (tid + 256 * i);
        if (e < p.xs_floats) *reinterpret_cast<f32x4*>(xs + e) = pre[i];
      }
    } else {
      const int64_t j0 = (int64_t)f0 * shift - p.npad_left;
      for (int i = tid; i < span; i += 256) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
    }
    {  // request the next tile's span now; it lands while this tile is being computed
      const int fn = f0 + kTileFrames;
      have_pre = (t + 1 < p.tiles_per_block) && fn < cd.num_frames && tile_is_interior(fn);
      if (have_pre) issue_prefetch(fn);
    }
    HF_T(1);
    __syncthreads();
    HF_T(2);

    // ---- S3: one frame per 16 lanes -------------------------------------------------------
#ifdef HIPFEAT_PHASE_TIMERS2
    unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0;
#endif
    {
      const float* x = xs + (4 * wv + g) * shift + 2 * q;
      v2 z[16];
      v2 win[NROWS];
      v2 sum2 = {0.f, 0.f};
      // the sample rows and the window are requested together up front
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) z[n1] = *reinterpret_cast<const v2*>(x + 32 * n1);
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) win[n1] = cwin[n1 * 16 + q];
#if HIPFEAT_HOIST_TW
      v2 twp[16];  // pass twiddles: in flight while the window / first FFT run
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) twp[k1] = ctwp[k1 * 16 + q];
#endif
      // samples at or beyond N are not part of the frame (the template instance may carry up to three rows more than
      // ceil(N / 32)): uniform test per row, lane mask only in the boundary rows
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
      HF_U(0);
      float mu = 0.f;
      if (dc) mu = row16_sum(sum2.x + sum2.y) * inv_n;
      float tprev = 0.f;
#pragma unroll
      for (int n1 = 0; n1 < NROWS; ++n1) {
        const v2 d = z[n1] - v2{mu, mu};
        const float tcur = dpp_mov<DPP_ROW_ROR1>(d.y);  // lane q <- lane q-1 (lane 0 <- lane 15)
        // previous sample of the first element: lane q-1's second element; for lane 0 it is
        // lane 15's second element of the previous row; the very first sample replicates itself
        const float dp = (q == 0) ? (n1 == 0 ? d.x : tprev) : tcur;
        tprev = tcur;
        z[n1] = (d - v2{c, c} * v2{dp, d.x}) * win[n1];
      }
#pragma unroll
      for (int n1 = NROWS; n1 < 16; ++n1) z[n1] = v2{0.f, 0.f};
      HF_U(1);
      v2 a[16];
      fft16(z, a);
      HF_U(2);
#pragma unroll
#if HIPFEAT_HOIST_TW
      for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul(a[k1], twp[k1]);
#else
      for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul(a[k1], ctwp[k1 * 16 + q]);
#endif
      // exchange: row k1 of this frame's block receives this lane's A[k1] at column q
      float* exf = myreg + g * kExFrameStride;
#pragma unroll
      for (int k1 = 0; k1 < 16; ++k1) *reinterpret_cast<v2*>(exf + k1 * kExRowStride + 2 * q) = a[k1];
      HF_U(3);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      v2 b[16];
#pragma unroll
      for (int n2 = 0; n2 < 16; ++n2) b[n2] = *reinterpret_cast<const v2*>(exf + q * kExRowStride + 2 * n2);
      HF_U(4);
#if HIPFEAT_HOIST_TW
      v2 tsw[8], tswp[8];  // split-step twiddles: in flight while the second FFT runs
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        tsw[k2] = ctws[k2 * 16 + q];
        tswp[k2] = ctwsp[k2 * 16 + q];
      }
#endif
      v2 Z[16];
      fft16(b, Z);
      HF_U(5);
      // all lanes must have finished reading the exchange rows before the power rows overwrite them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // Split step on bin PAIRS.  For k = q + 16 k2 (k2 < 8) the mirror bin 256-k is register 15-k2
      // of lane (16-q)%16; with a = Z[k], b = Z[256-k], s = a + conj(b), d = a - conj(b), t = (-i W^k) d:
      //   X[k] = s + t   and   X[256-k] = conj(s - t),
      // so each lane evaluates 8 pairs and stores 16 power values (8 in its own column, 8 in its
      // partner's).  Lane 0 (column 0) is its own partner with registers (16-k2)%16; its k2 = 0 pair
      // yields DC and Nyquist, and bin 128 (register 8, self-paired, W = -i) is added separately.
      float* prow = myreg + g * kPRowStride;
      float* pown = prow + q;
      float* ppar = prow + ((16 - q) & 15) + (q == 0 ? 16 : 0);
      if (q < 3) prow[257 + q] = 0.f;  // pad columns 257..259 are read (with zero weight) by the last k-group
      float t1[16];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].x);
        t1[2 * k2 + 1] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].y);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) t1[i] = dpp_mov<DPP_ROW_ROR1>(t1[i]);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        v2 m = v2{t1[2 * k2], t1[2 * k2 + 1]};
        if (q == 0) m = Z[(16 - k2) & 15];
        const v2 sp = m * HF_CJ + Z[k2];   // a + conj(b)
        const v2 dm = m * HF_NCJ + Z[k2];  // a - conj(b)
#if HIPFEAT_HOIST_TW
        const v2 tt = cmulc(dm, tsw[k2], tswp[k2]);
#else
        const v2 tt = cmulc(dm, ctws[k2 * 16 + q], ctwsp[k2 * 16 + q]);
#endif
        const v2 xp = sp + tt, xm = sp - tt;
        pown[16 * k2] = xp.x * xp.x + xp.y * xp.y;
        ppar[16 * (15 - k2)] = xm.x * xm.x + xm.y * xm.y;
      }
      if (q == 0) prow[128] = 4.f * (Z[8].x * Z[8].x + Z[8].y * Z[8].y);
    }
    HF_T(3);
    __syncthreads();
    HF_T(4);

    // ---- S5: banded mel GEMM on the matrix cores + log + store ---------------------------
    {
      const int j = lane & 15, kk = lane >> 4;  // B operand: frame j, bin slot kk
      const float* pb = regions + (j >> 2) * kWaveRegion + (j & 3) * kPRowStride + 2 * kk;
      float* orow = p.out + (cd.out_row + f0 + j) * p.out_stride;
      const bool vec_ok = ((p.out_stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.out) & 15) == 0);
      auto epilogue = [&](const f32x4 acc, int tile) {
        const int m0 = tile * 16 + 4 * kk;
        f32x4 v;
        v.x = fast_log(fmaxf(acc.x, p.mel_floor));
        v.y = fast_log(fmaxf(acc.y, p.mel_floor));
        v.z = fast_log(fmaxf(acc.z, p.mel_floor));
        v.w = fast_log(fmaxf(acc.w, p.mel_floor));
        if (j < nf) {
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
      // gfx950 counts loads and stores in ONE in-order counter (vmcnt).  Consume the prefetched
      // span registers here, before this tile's output stores are issued: the loads were requested
      // a whole S3 ago, so this wait is free, and the next S1 then never has to wait behind the
      // (slow to retire) stores.
#pragma unroll
      for (int i = 0; i < kPrefetch; ++i) asm volatile("" : "+v"(pre[i]));
      // Each segment is processed in chunks of HIPFEAT_S5_CHUNK groups: one burst of LDS reads per
      // chunk (a single latency exposure), then its MFMAs; a chunk runs when the band reaches it
      // (groups past the band inside a chunk carry zero weights and a clamped, valid bin offset).
      // Two accumulators hide the 40-cycle dependent-MFMA latency behind the 32-cycle issue interval.
      constexpr int CH = HIPFEAT_S5_CHUNK;
      if (ww.ngroups0 > 0) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c0 = 0; c0 < kMaxGroups0; c0 += CH) {
          if (c0 < ww.ngroups0) {
            v2 pv[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i)
              if (c0 + i < kMaxGroups0) pv[i] = *reinterpret_cast<const v2*>(pb + min(ww.bin0 + 8 * (c0 + i), kPRowStride - 8));
#pragma unroll
            for (int i = 0; i < CH; ++i)
              if (c0 + i < kMaxGroups0) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * (c0 + i)], pv[i].x, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * (c0 + i) + 1], pv[i].y, acc2, 0, 0, 0);
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
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * (kMaxGroups0 + gi)], pv[gi].x, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(mela[2 * (kMaxGroups0 + gi) + 1], pv[gi].y, acc2, 0, 0, 0);
        }
        epilogue(acc + acc2, ww.tile1);
      }
    }
    HF_T(5);
    HF_ACC(0, t0, t1);  // S1 work
    HF_ACC(1, t1, t2);  // barrier 1
    HF_ACC(2, t2, t3);  // S3
    HF_ACC(3, t3, t4);  // barrier 2
    HF_ACC(4, t4, t5);  // S5
    HF_ACC(5, t0, t0 + 1);  // tile count
#ifdef HIPFEAT_PHASE_TIMERS2
    hf_acc[0] = hf_acc[0] - (t1 - t0) + (u0 - t2);  // slot0: reads + sum
    hf_acc[1] = hf_acc[1] - (t2 - t1) + (u1 - u0);  // slot1: mean + preprocess
    hf_acc[2] = hf_acc[2] - (t3 - t2) + (u2 - u1);  // slot2: fft1
    hf_acc[3] = hf_acc[3] - (t4 - t3) + (u3 - u2);  // slot3: twiddle + exchange write
    hf_acc[4] = hf_acc[4] - (t5 - t4) + (u4 - u3);  // slot4: exchange read
    hf_acc[6] += (u5 - u4);                          // slot6: fft2
    hf_acc[7] += (t3 - u5);                          // slot7: mirror + split + P write
#endif
    // the next tile's S1 only writes xs; its barrier separates this tile's P reads from the next
    // tile's exchange writes
  }
#ifdef HIPFEAT_PHASE_TIMERS
  if (lane == 0 && g_phase_buf) {
    unsigned long long* o = g_phase_buf + ((size_t)blockIdx.x * 4 + wv) * 8;
    for (int i = 0; i < 8; ++i) o[i] = hf_acc[i];
  }
#endif
}

}  // namespace hipfeat
