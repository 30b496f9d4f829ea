// "Wave per frame" kernel: every power-of-two FFT size from 512 to 2048 (complex size H = 256 .. 1024) that has no
// specialised kernel -- 22.05 / 24 / 32 kHz (fft 1024) and 44.1 / 48 kHz (fft 2048) with 25 ms frames, and any option the
// fast paths do not take (energy columns, magnitude spectra ...).  All four feature kinds.
//
// One WAVE owns one frame at a time and never synchronises with the rest of its workgroup (the only __syncthreads is
// after the twiddle table load): samples come straight from global memory (coalesced, reflected at the cut edges,
// overlapping frames hit L1/L2), the DC mean / log-energy are wave reductions, the windowed frame goes to a wave-private
// LDS buffer as H complex numbers, the complex FFT runs as three register passes H = N1 x 8 x 8 (N1 = H/64 = 2..16,
// two transposes through the same buffer with wave-level barriers), the split step and |X|^2 overwrite the buffer with
// the power row, and the mel filterbank / DCT are lane-per-output dot products over a COMPACT band table
// (mel_t[t][j] = weight of filter j at bin lo_j + t: coalesced across lanes for every t).
//
// Replaces, for these sizes, the workgroup-wide radix-2 passes of kernel_generic.hpp (11 barrier-separated LDS sweeps at
// fft 2048, one workgroup per CU): measured in DESIGN.md section 4.2.
#pragma once
#include "common.hpp"
#include "fft_common.hpp"

namespace hipfeat {

struct WaveParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;    // [N]
  const float2* tw;       // W_2H^k, k < H
  const float* mel_blob;  // [M] int4 {first bin rounded down to 4, offset into the weights, float4 groups, 0}, then the compact
                          // zero-padded weights: copied to LDS by every workgroup (~1100 + 6 M floats)
  const float* dct;       // [M][C]  (copied to LDS when dct_in_lds)
  const float* lifter;    // [C]
  int64_t out_stride;
  int32_t num_cuts, uniform_bpc, frames_per_wave;
  int32_t N, shift, H, K, M, C, mel_blob_floats, dct_in_lds;
  int32_t kind, flags, npad_left;
  float preemph, log_energy_floor, mel_floor, log_offset;
};

__device__ __forceinline__ v2 twiddle_h(const float2* __restrict__ tw, int m, int H) {  // W_H^m, 0 <= m < H, from W_2H^k (k < H)
  int t = 2 * m;
  const bool neg = t >= H;
  t -= neg ? H : 0;
  const float2 w = tw[t];
  return neg ? v2{-w.x, -w.y} : v2{w.x, w.y};
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_sum(float v) {  // every lane gets the total: 4 DPP adds inside the rows of 16, then the 4 row totals
  v = row16_sum(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}

template <int N1>
__device__ __forceinline__ void small_fft(const v2* x, v2* X) {
  if constexpr (N1 == 16) {
    v2 a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x[i];
    fft16(a, b);
#pragma unroll
    for (int i = 0; i < 16; ++i) X[i] = b[i];
  } else if constexpr (N1 == 8) {
    fft8(x, X);
  } else if constexpr (N1 == 4) {
    const v2 s0 = x[0] + x[2], s1 = x[0] - x[2], s2 = x[1] + x[3], u3 = swap2(x[1] - x[3]);
    X[0] = s0 + s2, X[2] = s0 - s2, X[1] = u3 * HF_CJ + s1, X[3] = u3 * HF_NCJ + s1;
  } else {
    X[0] = x[0] + x[1], X[1] = x[0] - x[1];
  }
}

// Complex FFT of size H = N1 * 8 * 8 of one frame by one wave, in place in LDS, natural order in and out:
//   n = 64 n1 + 8 n2 + n3,  k = k1 + N1 k2 + 8 N1 k3
//   pass A  FFT_N1 over n1 (lane = (n2, n3)),      twiddle W_H^(8 n2 k1)
//   pass B  FFT_8  over n2 (item = (k1, n3)),      twiddle W_H^(n3 (k1 + N1 k2))
//   pass C  FFT_8  over n3 (item = (k1, k2))
// While the three passes run, element e = 64 n1 + 8 n2 + n3 lives at e + (e >> 3) = 72 n1 + 9 n2 + n3: with these strides
// the 8-byte accesses of passes B and C are bank-conflict free (unpadded, eight lanes share a bank pair: measured 45 % of
// the LDS cycles were conflicts).  Pass A takes its input from registers (lane l holds the elements l + 64 n1, which is how the
// front end produces them); the result comes back in natural order in LDS.
__device__ __forceinline__ int fft3_pad(int e) { return e + (e >> 3); }

template <int N1>
__device__ __forceinline__ void fft3_frame(v2* zf, const v2* __restrict__ twh, int lane, const v2* x) {  // twh[m] = W_H^m, x[n1] = element lane + 64 n1
  const int lp = lane + (lane >> 3);  // 9 n2 + n3
  {
    v2 A[N1];
    small_fft<N1>(x, A);
    const int n2 = lane >> 3;
#pragma unroll
    for (int k1 = 1; k1 < N1; ++k1) A[k1] = cmul(A[k1], twh[8 * n2 * k1]);
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) zf[72 * k1 + lp] = A[k1];
  }
  wave_lds_sync();
  constexpr int ITEMS = 8 * N1, ROUNDS = (ITEMS + 63) / 64;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // pass B, in place (an item re-writes exactly the column it read)
    const int it = lane + 64 * r;
    if (it < ITEMS) {
      const int k1 = it >> 3, n3 = it & 7;
      v2 x[8], B[8];
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) x[n2] = zf[72 * k1 + 9 * n2 + n3];
      fft8(x, B);
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        B[k2] = cmul(B[k2], twh[n3 * (k1 + N1 * k2)]);
        zf[72 * k1 + 9 * k2 + n3] = B[k2];
      }
    }
  }
  wave_lds_sync();
  {
    v2 X[ROUNDS][8];  // pass C: all reads, barrier, then natural-order writes
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = lane + 64 * r;
      if (it < ITEMS) {
        const int k1 = it >> 3, k2 = it & 7;
        v2 x[8];
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) x[n3] = zf[72 * k1 + 9 * k2 + n3];
        fft8(x, X[r]);
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int it = lane + 64 * r;
      if (it < ITEMS) {
        const int k1 = it >> 3, k2 = it & 7;
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) zf[k1 + N1 * k2 + 8 * N1 * k3] = X[r][k3];
      }
    }
  }
  wave_lds_sync();
}


#ifndef HIPFEAT_WAVE_OCC
#define HIPFEAT_WAVE_OCC 4
#endif
template <int N1>
__global__ __launch_bounds__(256, (N1 == 16 ? 2 : HIPFEAT_WAVE_OCC)) void wave_kernel(const WaveParams p) {
  constexpr int H = 64 * N1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  float2* tw = reinterpret_cast<float2*>(smem);                  // [H]
  float* winl = smem + 2 * H + 4 * (144 * N1 + 8);               // [N] window, shared by the four waves
  v2* twh = reinterpret_cast<v2*>(winl + ((p.N + 3) & ~3));       // [H] W_H^m for the FFT passes
  float* melb = reinterpret_cast<float*>(twh + H);                // filterbank descriptors + compact weights
  float* dctl = melb + ((p.mel_blob_floats + 3) & ~3);            // [M][C] DCT matrix when it fits
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* buf = smem + 2 * H + wv * (144 * N1 + 8);               // wave-private: 72 N1 complex (padded FFT buffer), later the power row
  v2* zf = reinterpret_cast<v2*>(buf);

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
  for (int i = tid; i < H; i += 256) {
    tw[i] = p.tw[i];
    twh[i] = twiddle_h(p.tw, i, H);
  }
  for (int i = tid; i < p.N; i += 256) winl[i] = p.window[i];
  for (int i = tid; i < p.mel_blob_floats; i += 256) melb[i] = p.mel_blob[i];
  if (p.dct_in_lds)
    for (int i = tid; i < p.M * p.C; i += 256) dctl[i] = p.dct[i];
  __syncthreads();
  const int4* meld = reinterpret_cast<const int4*>(melb);
  const float* melw = melb + 4 * p.M;
  const float* __restrict__ dctp = p.dct_in_lds ? dctl : p.dct;

  const int N = p.N, K = p.K, M = p.M;
  const bool use_e = (p.flags & F_USE_ENERGY) != 0;
  const int fbase = fb * 4 * p.frames_per_wave;
#pragma unroll 1
  for (int i = 0; i < p.frames_per_wave; ++i) {
    const int f = fbase + 4 * i + wv;
    if (f >= cd.num_frames) break;
    const int64_t j0 = (int64_t)f * p.shift - p.npad_left;

    // ---- samples, DC mean, raw log-energy (layers.py:155-162), in FFT order: lane l holds the sample pairs ----------
    // (x[2n], x[2n+1]) for n = l + 64 q -- the complex elements z[n] of the half-size FFT, exactly what pass A wants.
    // Frames that lie inside the cut (almost all) use one 8-byte load per pair; edge frames go sample by sample.
    v2 xp[N1];
    float s = 0.f;
    const bool inside = j0 >= 0 && j0 + ((N + 1) & ~1) <= (int64_t)cd.num_samples;
#pragma unroll
    for (int q = 0; q < N1; ++q) {
      const int m0 = 2 * (lane + 64 * q);
      v2 v = {0.f, 0.f};
      if (m0 < N) {
        if (inside) {
          __builtin_memcpy(&v, w + j0 + m0, sizeof(v2));  // 4-byte aligned 8-byte load
          if (m0 + 1 >= N) v.y = 0.f;
        } else if (p.flags & F_CENTER) {
          v.x = load_sample_center(w, j0 + m0, cd.num_samples);
          if (m0 + 1 < N) v.y = load_sample_center(w, j0 + m0 + 1, cd.num_samples);
        } else {
          v.x = load_sample(w, j0 + m0, cd.num_samples, cd.padded_len);
          if (m0 + 1 < N) v.y = load_sample(w, j0 + m0 + 1, cd.num_samples, cd.padded_len);
        }
      }
      xp[q] = v;
      s += v.x + v.y;
    }
    float mean = 0.f;
    if (p.flags & F_REMOVE_DC) mean = wave_sum(s) / (float)N;
    float log_e = 0.f;
    if (use_e && (p.flags & F_RAW_ENERGY)) {
      float e = 0.f;
#pragma unroll
      for (int q = 0; q < N1; ++q) {
        const int m0 = 2 * (lane + 64 * q);
        const float d0 = m0 < N ? xp[q].x - mean : 0.f, d1 = m0 + 1 < N ? xp[q].y - mean : 0.f;
        e = fmaf(d0, d0, fmaf(d1, d1, e));
      }
      log_e = fmaxf(logf(wave_sum(e) + 1e-15f), p.log_energy_floor);
    }
    // y[m] = (d[m] - c d[max(m-1, 0)]) w[m] with d = x - mean, zero padded to 2H; d[2n-1] is the previous lane's second sample
    v2 y[N1];
    {
      float e = 0.f;
#pragma unroll
      for (int q = 0; q < N1; ++q) {
        const int m0 = 2 * (lane + 64 * q);
        const float d0 = xp[q].x - mean, d1 = xp[q].y - mean;
        // lane l >= 1 wants lane l-1's second sample of this register, lane 0 wants lane 63's of the previous register:
        // lane 63 offers that one (nobody reads its current one here), so a single wrap-around ds_bpermute serves both
        const float offer = (q > 0 && lane == 63) ? xp[q > 0 ? q - 1 : 0].y - mean : d1;
        float dm = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(4 * ((lane - 1) & 63), __builtin_bit_cast(int, offer)));
        if (q == 0 && lane == 0) dm = d0;  // d[max(m - 1, 0)] at the start of the frame
        v2 v = {0.f, 0.f};
        if (m0 < N) {
          const v2 wn = *reinterpret_cast<const v2*>(winl + m0);
          v.x = (d0 - p.preemph * dm) * wn.x;
          if (m0 + 1 < N) v.y = (d1 - p.preemph * d0) * wn.y;
        }
        y[q] = v;
        e = fmaf(v.x, v.x, fmaf(v.y, v.y, e));
      }
      if (use_e && !(p.flags & F_RAW_ENERGY)) log_e = fmaxf(logf(wave_sum(e) + 1e-15f), p.log_energy_floor);  // layers.py:183-185
    }
    wave_lds_sync();  // the previous frame's readers of this wave's buffer are done

    // ---- complex FFT, split step X[k] = E[k] + W_2H^k O[k], power (layers.py:32-42) --------------------------------
    fft3_frame<N1>(zf, twh, lane, y);
    {
      // bins k and H - k come from the same two FFT outputs: with a = Z[k], b = Z[H-k], E = (a + conj b)/2, O = -i (a - conj b)/2,
      // T = W_2H^k O:  X[k] = E + T and X[H-k] = conj(E - T).  Lane l takes k = l + 64 q <= H/2; k = 0 yields P[0] and P[H].
      float pa[N1 / 2 + 1], pb[N1 / 2 + 1];
#pragma unroll
      for (int q = 0; q <= N1 / 2; ++q) {
        const int k = lane + 64 * q;
        pa[q] = pb[q] = 0.f;
        if (k <= H / 2) {
          const v2 a = zf[k], b = zf[(H - k) & (H - 1)];
          const float ex = 0.5f * (a.x + b.x), ey = 0.5f * (a.y - b.y);
          const float ox = 0.5f * (a.y + b.y), oy = -0.5f * (a.x - b.x);
          const float2 wk = tw[k];
          const float tx = wk.x * ox - wk.y * oy, ty = wk.x * oy + wk.y * ox;
          const float xr = ex + tx, xi = ey + ty, yr = ex - tx, yi = ey - ty;
          float v = xr * xr + xi * xi, u = yr * yr + yi * yi;
          if (p.flags & F_FFT_MAG) { v = sqrtf(v); u = sqrtf(u); }
          pa[q] = v;
          pb[q] = u;
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int q = 0; q <= N1 / 2; ++q) {
        const int k = lane + 64 * q;
        if (k <= H / 2) {
          buf[k] = pa[q];
          if (k < H / 2) buf[H - k] = pb[q];
        }
      }
      if (lane < 3) buf[K + lane] = 0.f;  // the float4 reads of the mel stage run up to 3 entries past the row (zero weights)
    }
    wave_lds_sync();

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    float* __restrict__ orow = p.out + (cd.out_row + f) * p.out_stride;
    if (p.kind == 0 || p.kind == 1) {
      for (int k = lane; k < K; k += 64) {
        float v = buf[k];
        if (p.kind == 1) v = fast_log(v + p.log_offset);
        if (use_e && k == 0) v = log_e;
        orow[k] = v;
      }
    } else {
      const int ecol = (p.kind == 2 && use_e) ? 1 : 0;
      // mel: FOUR lanes per filter (16 consecutive filters per round, similar band widths).  Power row, filter
      // descriptors and the compact weights are all in LDS; a filter's taps start at a multiple of four bins (zero
      // weights in front and behind), so lane (jj, sub) takes the float4 groups sub, sub + 4, ... with two 16-byte
      // LDS reads per four multiply-adds; two DPP adds combine the quarters.
      const int jj = lane >> 2, sub = lane & 3;
      for (int j0m = 0; j0m < M; j0m += 16) {
        const int j = j0m + jj;
        float acc = 0.f;
        if (j < M) {
          const int4 dsc = meld[j];
          const float4* pb4 = reinterpret_cast<const float4*>(buf + dsc.x);
          const float4* mw4 = reinterpret_cast<const float4*>(melw + dsc.y);
          for (int g = sub; g < dsc.z; g += 4) {
            const float4 a = pb4[g], b = mw4[g];
            acc = fmaf(a.x, b.x, acc);
            acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc);
            acc = fmaf(a.w, b.w, acc);
          }
        }
        acc += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(acc);
        acc += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(acc);
        if (j < M && sub == 0) buf[2 * H - 120 + j] = acc;  // mel energies stashed behind the power row (K + 3 <= 2H - 120, M <= 128)
      }
      wave_lds_sync();
      // one lane per filter for the logarithm, and one coalesced store of the row
      for (int j = lane; j < M; j += 64) {
        const float acc = buf[2 * H - 120 + j];
        // v_log_f32 (log2, 1 ulp) times a constant, as in the fft512 kernels: the argument is a normal positive float
        const float v = __builtin_amdgcn_logf(fmaxf(acc, p.mel_floor)) * ((p.flags & F_LOG10) ? 0.30102999566398120f : 0.69314718055994531f);
        if (p.kind == 2) orow[ecol + j] = v;
        else buf[2 * H - 120 + j] = v;  // MFCC: log-mel vector for the DCT
      }
      if (p.kind == 2) {
        if (ecol && lane == 0) orow[0] = log_e;
      } else {
        wave_lds_sync();
        const float* lmv = buf + 2 * H - 120;  // log-mel vector [M <= 128]
        const int cc = lane >> 2;              // DCT: four lanes per cepstrum as well
        for (int c0 = 0; c0 < p.C; c0 += 16) {
          const int c = c0 + cc;
          float acc = 0.f;
          if (c < p.C) {
#pragma unroll 4
            for (int m = sub; m < M; m += 4) acc = fmaf(lmv[m], dctp[(size_t)m * p.C + c], acc);
          }
          acc += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(acc);
          acc += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(acc);
          if (c < p.C && sub == 0) {
            if (p.flags & F_LIFTER) acc *= p.lifter[c];
            if (use_e && c == 0) acc = log_e;
            orow[c] = acc;
          }
        }
      }
    }
    wave_lds_sync();  // the buffer is reused by the next frame
  }
}

}  // namespace hipfeat
