// Generic feature kernel: any frame length / shift / fft size / feature kind.
//
// One workgroup (256 threads) = `fpb` consecutive frames of one cut.  Everything between the
// coalesced load of the sample span and the coalesced store of the feature rows stays in LDS:
//
//   HBM --(span of (fpb-1)*shift+N samples, reflected at the cut edges)--> LDS xs
//   xs  -> per-frame DC mean / raw log-energy            (layers.py:155-162)
//       -> pre-emphasis, window, zero pad                 (layers.py:165-181)
//       -> real FFT: even/odd samples packed into a complex FFT of size fft/2, radix-2 DIF in LDS,
//          then the split step X[k] = E[k] + W^k O[k]     (replaces torch.fft.rfft, layers.py:32-36)
//       -> |X|^2 or |X|                                   (layers.py:38-42)
//       -> mel filterbank (banded dot products), log      (layers.py:571-572)
//       -> DCT + lifter for MFCC                          (layers.py:716-718)
//   LDS --> HBM rows of the output matrix
//
// Non power-of-two fft sizes (round_to_power_of_two=False) use a direct DFT instead of the FFT.
// This kernel is the correctness baseline and the fallback for every configuration the
// specialised kernels do not cover.
#pragma once
#include "common.hpp"

namespace hipfeat {

__device__ __forceinline__ float group32_sum(float v) {
  // sum over the 32 consecutive lanes of a half wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
  return v;
}

__global__ __launch_bounds__(256) void generic_kernel(const GenericParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HF_POISON_LDS(smem);
  float* xs = smem;
  float2* z = reinterpret_cast<float2*>(smem + p.off_z);
  float* zr = smem + p.off_z;
  float* P = smem + p.off_p;
  float2* tw = reinterpret_cast<float2*>(smem + p.off_tw);
  float* stat = smem + p.off_stat;
  float* melb = smem + p.off_mel;

  const int tid = threadIdx.x;
  constexpr int T = 256;
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
  const int f0 = fb * p.fpb;
  const int nf = min(p.fpb, cd.num_frames - f0);
  const float* __restrict__ w = p.wave + cd.wave_off;
  const int64_t j0 = (int64_t)f0 * p.shift - p.npad_left;
  const bool pow2 = (p.flags & F_POW2) != 0;
  const int N = p.N, H = p.H, K = p.K, fft = p.fft;

  // ---- phase 1: sample span + twiddles into LDS ------------------------------------
  if (p.flags & F_CENTER) {
    for (int i = tid; i < p.span; i += T) xs[i] = load_sample_center(w, j0 + i, cd.num_samples);
  } else {
    for (int i = tid; i < p.span; i += T) xs[i] = load_sample(w, j0 + i, cd.num_samples, cd.padded_len);
  }
  if (pow2)
    for (int i = tid; i < H; i += T) tw[i] = p.tw[i];
  __syncthreads();

  // ---- phase 2: per-frame mean and raw log-energy (32 threads per frame) ------------
  {
    const int f = tid >> 5, l = tid & 31;
    if (f < p.fpb) {
      const float* x = xs + f * p.shift;
      float mean = 0.f;
      if (p.flags & F_REMOVE_DC) {
        float s = 0.f;
        for (int m = l; m < N; m += 32) s += x[m];
        mean = group32_sum(s) / (float)N;
      }
      float le = 0.f;
      if ((p.flags & F_USE_ENERGY) && (p.flags & F_RAW_ENERGY)) {
        float s = 0.f;
        for (int m = l; m < N; m += 32) {
          float d = x[m] - mean;
          s = fmaf(d, d, s);
        }
        le = fmaxf(logf(group32_sum(s) + 1e-15f), p.log_energy_floor);
      }
      if (l == 0) {
        stat[2 * f] = mean;
        stat[2 * f + 1] = le;
      }
    }
  }
  __syncthreads();

  // ---- phase 3: DC removal, pre-emphasis, window, zero pad --------------------------
  auto yval = [&](int f, int m) -> float {
    if (m >= N) return 0.f;
    const float* x = xs + f * p.shift;
    const float mean = stat[2 * f];
    const float d = x[m] - mean;
    const float dp = x[m > 0 ? m - 1 : 0] - mean;  // replicate pad (layers.py:166)
    return (d - p.preemph * dp) * p.window[m];
  };
  if (pow2) {
    for (int idx = tid; idx < p.fpb * H; idx += T) {
      const int f = idx >> p.log2H, n = idx & (H - 1);
      z[idx] = make_float2(yval(f, 2 * n), yval(f, 2 * n + 1));
    }
  } else {
    for (int idx = tid; idx < p.fpb * fft; idx += T) {
      const int f = idx / fft, m = idx - f * fft;
      zr[idx] = yval(f, m);
    }
  }
  __syncthreads();
  if ((p.flags & F_USE_ENERGY) && !(p.flags & F_RAW_ENERGY)) {
    // energy of the windowed frame (layers.py:183-185)
    const int f = tid >> 5, l = tid & 31;
    if (f < p.fpb) {
      float s = 0.f;
      const float* yr = zr + (size_t)f * fft;  // pow2: float2[H] == float[fft], same layout
      for (int m = l; m < fft; m += 32) s = fmaf(yr[m], yr[m], s);
      const float le = fmaxf(logf(group32_sum(s) + 1e-15f), p.log_energy_floor);
      if (l == 0) stat[2 * f + 1] = le;
    }
    __syncthreads();
  }

  // ---- phase 4/5: spectrum ------------------------------------------------------------
  if (pow2) {
    // radix-2 decimation-in-frequency, natural order in, bit-reversed order out
    const int halfH = H >> 1;
    for (int s = p.log2H - 1; s >= 0; --s) {
      const int half = 1 << s;
      for (int b = tid; b < p.fpb * halfH; b += T) {
        const int f = b >> (p.log2H - 1), bb = b & (halfH - 1);  // halfH is a power of two
        const int j = bb & (half - 1);
        const int i0 = ((bb >> s) << (s + 1)) | j;
        const int i1 = i0 + half;
        float2* zf = z + (size_t)f * H;
        const float2 u = zf[i0], v = zf[i1];
        const float2 tws = tw[j << (p.log2H - s)];  // W_{2*half}^j = W_fft^{j*H/half}
        const float dx = u.x - v.x, dy = u.y - v.y;
        zf[i0] = make_float2(u.x + v.x, u.y + v.y);
        zf[i1] = make_float2(dx * tws.x - dy * tws.y, dx * tws.y + dy * tws.x);
      }
      __syncthreads();
    }
    // split step + power
    const int rs = 32 - p.log2H;
    for (int idx = tid; idx < p.fpb * K; idx += T) {
      const int f = idx / K, k = idx - f * K;
      const int kz = k & (H - 1), kc = (H - k) & (H - 1);
      const int bz = p.log2H ? (int)(__brev((unsigned)kz) >> rs) : 0;
      const int bc = p.log2H ? (int)(__brev((unsigned)kc) >> rs) : 0;
      const float2 a = z[(size_t)f * H + bz], b = z[(size_t)f * H + bc];
      const float ex = 0.5f * (a.x + b.x), ey = 0.5f * (a.y - b.y);
      const float ox = 0.5f * (a.y + b.y), oy = -0.5f * (a.x - b.x);
      float2 wk = (k < H) ? tw[k] : make_float2(-1.f, 0.f);
      const float xr = ex + (wk.x * ox - wk.y * oy);
      const float xi = ey + (wk.x * oy + wk.y * ox);
      float pw = xr * xr + xi * xi;
      if (p.flags & F_FFT_MAG) pw = sqrtf(pw);
      P[idx] = pw;
    }
  } else {
    // direct DFT (non power-of-two fft): X[k] = sum_n y[n] W_fft^{nk}
    for (int idx = tid; idx < p.fpb * K; idx += T) {
      const int f = idx / K, k = idx - f * K;
      const float* yr = zr + (size_t)f * fft;
      float re = 0.f, im = 0.f;
      int ph = 0;
      for (int n = 0; n < N; ++n) {
        const float2 t = p.tw[ph];
        re = fmaf(yr[n], t.x, re);
        im = fmaf(yr[n], t.y, im);
        ph += k;
        if (ph >= fft) ph -= fft;
      }
      float pw = re * re + im * im;
      if (p.flags & F_FFT_MAG) pw = sqrtf(pw);
      P[idx] = pw;
    }
  }
  __syncthreads();

  // ---- phase 6: feature-specific epilogue ---------------------------------------------
  float* __restrict__ out = p.out + (cd.out_row + f0) * p.out_stride;
  const bool use_e = (p.flags & F_USE_ENERGY) != 0;
  if (p.kind == 0 || p.kind == 1) {
    for (int idx = tid; idx < nf * K; idx += T) {
      const int f = idx / K, k = idx - f * K;
      float v = P[f * K + k];
      if (p.kind == 1) v = logf(v + p.log_offset);
      if (use_e && k == 0) v = stat[2 * f + 1];
      out[(int64_t)f * p.out_stride + k] = v;
    }
    return;
  }
  const int M = p.M;
  const int ecol = (p.kind == 2 && use_e) ? 1 : 0;
  for (int idx = tid; idx < p.fpb * M; idx += T) {
    const int f = idx / M, j = idx - f * M;
    const int2 r = p.mel_range[j];
    const float* pf = P + f * K;
    float acc = 0.f;
    for (int k = r.x; k < r.y; ++k) acc = fmaf(pf[k], p.mel[(size_t)k * M + j], acc);
    const float v = (p.flags & F_LOG10) ? log10f(fmaxf(acc, p.mel_floor)) : logf(fmaxf(acc, p.mel_floor));
    if (p.kind == 2) {
      if (f < nf) {
        out[(int64_t)f * p.out_stride + ecol + j] = v;
        if (ecol && j == 0) out[(int64_t)f * p.out_stride] = stat[2 * f + 1];
      }
    } else {
      melb[idx] = v;
    }
  }
  if (p.kind == 3) {
    __syncthreads();
    const int C = p.C;
    for (int idx = tid; idx < nf * C; idx += T) {
      const int f = idx / C, c = idx - f * C;
      const float* mf = melb + f * M;
      float acc = 0.f;
      for (int m = 0; m < M; ++m) acc = fmaf(mf[m], p.dct[(size_t)m * C + c], acc);
      if (p.flags & F_LIFTER) acc *= p.lifter[c];
      if (use_e && c == 0) acc = stat[2 * f + 1];  // the log-energy replaces C0 (Kaldi; the intent of layers.py:721-722)
      out[(int64_t)f * p.out_stride + c] = acc;
    }
  }
}

// Whisper post-pass (whisper_fbank.py:67-80) over one cut per workgroup: the main kernel left log10(max(mel, 1e-10)) in
// the cut's rows; here: per-cut maximum over the S / shift computed frames, clamp to (max - 8) in log10 units, the
// affine map (x + 4) / 4, and zeros in the padding row.
__global__ __launch_bounds__(1024) void whisper_norm_kernel(const CutDesc* __restrict__ cuts, float* __restrict__ out, int64_t stride,
                                                            int32_t M, int32_t shift) {
  __shared__ float red[16];
  const CutDesc cd = cuts[blockIdx.x];
  const int valid = min(cd.num_samples / shift, cd.num_frames);
  float* __restrict__ base = out + cd.out_row * stride;
  const int64_t n = (int64_t)valid * M, nt = (int64_t)cd.num_frames * M;
  // dense rows (stride == M) on a 16-byte boundary: one linear float4 sweep per pass
  const bool dense = stride == M && ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
  float mx = -INFINITY;
  // cuts of up to ~12 s (24 float4 per lane) stay in registers between the two passes: one read instead of two
  constexpr int kKeep = 24;
  const int64_t n4all = n >> 2;
  const bool keep = dense && n4all <= (int64_t)kKeep * 1024;
  float4 held[kKeep];
  if (keep) {
    const float4* b4 = reinterpret_cast<const float4*>(base);
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
      const int64_t i = threadIdx.x + 1024 * k;
      if (i < n4all) {
        held[k] = b4[i];
        mx = fmaxf(fmaxf(mx, fmaxf(held[k].x, held[k].y)), fmaxf(held[k].z, held[k].w));
      }
    }
    for (int64_t i = (n4all << 2) + threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, base[i]);
  } else if (dense) {
    const float4* b4 = reinterpret_cast<const float4*>(base);
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
      const float4 v = b4[i];
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 1024) mx = fmaxf(mx, base[i]);
  } else {
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
      const int64_t r = i / M;
      mx = fmaxf(mx, base[r * stride + (i - r * M)]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
  const float lo = mx - 8.0f;
  if (keep) {
    float4* b4 = reinterpret_cast<float4*>(base);
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
      const int64_t i = threadIdx.x + 1024 * k;
      if (i < n4all) {
        float4 v = held[k];
        v.x = (fmaxf(v.x, lo) + 4.0f) * 0.25f;
        v.y = (fmaxf(v.y, lo) + 4.0f) * 0.25f;
        v.z = (fmaxf(v.z, lo) + 4.0f) * 0.25f;
        v.w = (fmaxf(v.w, lo) + 4.0f) * 0.25f;
        b4[i] = v;
      }
    }
    for (int64_t i = (n4all << 2) + threadIdx.x; i < nt; i += 1024) base[i] = (i < n) ? (fmaxf(base[i], lo) + 4.0f) * 0.25f : 0.0f;
  } else if (dense) {
    float4* b4 = reinterpret_cast<float4*>(base);
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += 1024) {
      float4 v = b4[i];
      v.x = (fmaxf(v.x, lo) + 4.0f) * 0.25f;
      v.y = (fmaxf(v.y, lo) + 4.0f) * 0.25f;
      v.z = (fmaxf(v.z, lo) + 4.0f) * 0.25f;
      v.w = (fmaxf(v.w, lo) + 4.0f) * 0.25f;
      b4[i] = v;
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < nt; i += 1024) base[i] = (i < n) ? (fmaxf(base[i], lo) + 4.0f) * 0.25f : 0.0f;
  } else {
    for (int64_t i = threadIdx.x; i < nt; i += 1024) {
      const int64_t r = i / M;
      float* q = base + r * stride + (i - r * M);
      *q = (r < valid) ? (fmaxf(*q, lo) + 4.0f) * 0.25f : 0.0f;
    }
  }
}

}  // namespace hipfeat
