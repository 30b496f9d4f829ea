// Post-feature transforms on a collated (B, T, F) float32 batch:
//   * GlobalMVN.forward / inverse  (lhotse/dataset/signal_transforms.py:50-60)
//   * SpecAugment                  (lhotse/dataset/signal_transforms.py:121-371): time warp = two bicubic
//     F.interpolate calls (:338-371), frequency / time masks filled with the sequence mean (:239-266, :297-335)
// The random choices are made by the host mirror with the reference's own RNG calls; the kernels apply them.
// Both are pure HBM streams: one read + one write of the batch for the warp/clone pass (with a deterministic
// per-tile partial sum for the means), and a second pass that writes only the masked elements.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace hipfeat {

struct WarpSeg {  // rows [start, start + len) of sequence seq: [0, center) -> [0, warped), [center, len) -> [warped, len)
  int32_t seq, start, len, center, warped;
};

struct SpecAugParams {
  const float* in;
  float* out;
  int32_t B, T, F;
  const int32_t* seg_off;   // [B + 1]  segments of sequence b: segs[seg_off[b] .. seg_off[b + 1])
  const WarpSeg* segs;
  const int32_t* mask_off;  // [B + 1]  masks of sequence b: masks[3 * mask_off[b] ..)
  const int32_t* masks;     // [n][3]   axis (1 = time, 2 = feature), begin, end
  float* partials;          // [B][tiles] sums of the output tiles
  int32_t tiles, rows_per_tile;
};

// torch's cubic convolution coefficients, A = -0.75 (ATen/native/UpSample.h: get_cubic_upsample_coefficients), with the
// multiply-adds fused like the reference's compiled kernels do (measured against F.interpolate: oracle/specaug_ref.py)
__device__ __forceinline__ float cubic1(float x) {  // ((A + 2) x - (A + 3)) x x + 1
  const float A = -0.75f;
  return fmaf(__fmul_rn(fmaf(A + 2.0f, x, -(A + 3.0f)), x), x, 1.0f);
}
__device__ __forceinline__ float cubic2(float x) {  // ((A x - 5A) x + 8A) x - 4A
  const float A = -0.75f;
  return fmaf(fmaf(fmaf(A, x, -5.0f * A), x, 8.0f * A), x, -4.0f * A);
}
__device__ __forceinline__ void cubic_coeffs(float t, float w[4]) {
  w[0] = cubic2(__fadd_rn(t, 1.0f));
  w[1] = cubic1(t);
  w[2] = cubic1(__fsub_rn(1.0f, t));
  w[3] = cubic2(__fsub_rn(2.0f, t));
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// pass 1: out = in with every warp segment resampled; partials[b][tile] = sum of the tile's output.
// V = 4: feature_dim % 4 == 0 and 16-byte aligned buffers -> one float4 per lane and source row.
template <int V>
__global__ __launch_bounds__(256) void specaug_warp_kernel(SpecAugParams p) {
  __shared__ float red[4];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int r0 = tile * p.rows_per_tile;
  const int nrows = min(p.rows_per_tile, p.T - r0);
  const int s0 = p.seg_off[b], s1 = p.seg_off[b + 1];
  const int F = p.F, FV = F / V;
  const float* __restrict__ src = p.in + (int64_t)b * p.T * F;
  float* __restrict__ dst = p.out + (int64_t)b * p.T * F;
  float acc = 0.f;
  const int n = nrows * FV;
  const int drr = 256 / FV, dcc = 256 - drr * FV;  // (row, column) of element i + 256 from those of element i: no division in the loop
  int rr = threadIdx.x / FV, cc = threadIdx.x - rr * FV;
  for (int i = threadIdx.x; i < n; i += 256, rr += drr, cc += dcc) {
    if (cc >= FV) { cc -= FV; ++rr; }
    const int col = cc * V;
    const int row = r0 + rr;
    float v[V];
    int hit = -1;
    for (int s = s0; s < s1; ++s) {
      const int st = p.segs[s].start;
      if (row >= st && row < st + p.segs[s].len) hit = s;
    }
    if (hit < 0) {
      if (V == 4) {
        const float4 q = *reinterpret_cast<const float4*>(src + (int64_t)row * F + col);
        v[0] = q.x; v[V > 1 ? 1 : 0] = q.y; v[V > 2 ? 2 : 0] = q.z; v[V > 3 ? 3 : 0] = q.w;
      } else {
        v[0] = src[(int64_t)row * F + col];
      }
    } else {
      const WarpSeg g = p.segs[hit];
      const int r = row - g.start;
      int in_len, out_len, base, d;
      if (r < g.warped) { in_len = g.center; out_len = g.warped; base = g.start; d = r; }
      else { in_len = g.len - g.center; out_len = g.len - g.warped; base = g.start + g.center; d = r - g.warped; }
      // area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=true) in float32, multiply-add fused
      const float scale = (float)in_len / (float)out_len;
      const float real = fmaf(scale, (float)d + 0.5f, -0.5f);
      const float fl = floorf(real);
      const int idx = (int)fl;
      float w[4];
      cubic_coeffs(real - fl, w);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rj = min(max(idx - 1 + j, 0), in_len - 1) + base;
        if (V == 4) {
          const float4 q = *reinterpret_cast<const float4*>(src + (int64_t)rj * F + col);
          v[0] = fmaf(w[j], q.x, v[0]); v[V > 1 ? 1 : 0] = fmaf(w[j], q.y, v[V > 1 ? 1 : 0]);
          v[V > 2 ? 2 : 0] = fmaf(w[j], q.z, v[V > 2 ? 2 : 0]); v[V > 3 ? 3 : 0] = fmaf(w[j], q.w, v[V > 3 ? 3 : 0]);
        } else {
          v[0] = fmaf(w[j], src[(int64_t)rj * F + col], v[0]);
        }
      }
    }
    if (V == 4) {
      *reinterpret_cast<float4*>(dst + (int64_t)row * F + col) = make_float4(v[0], v[V > 1 ? 1 : 0], v[V > 2 ? 2 : 0], v[V > 3 ? 3 : 0]);
      acc += (v[0] + v[V > 1 ? 1 : 0]) + (v[V > 2 ? 2 : 0] + v[V > 3 ? 3 : 0]);
    } else {
      dst[(int64_t)row * F + col] = v[0];
      acc += v[0];
    }
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) p.partials[(int64_t)b * p.tiles + tile] = s;
}

// pass 2: masked elements <- mean of the (warped) sequence
__global__ __launch_bounds__(256) void specaug_mask_kernel(SpecAugParams p) {
  extern __shared__ unsigned char flags[];  // [F] column flags, then [rows_per_tile] row flags
  __shared__ double dred[4];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int m0 = p.mask_off[b], m1 = p.mask_off[b + 1];
  if (m0 == m1) return;
  const int F = p.F;
  const int r0 = tile * p.rows_per_tile;
  const int nrows = min(p.rows_per_tile, p.T - r0);
  unsigned char* colf = flags;
  unsigned char* rowf = flags + F;
  for (int i = threadIdx.x; i < F + p.rows_per_tile; i += 256) flags[i] = 0;
  double s = 0.0;
  for (int i = threadIdx.x; i < p.tiles; i += 256) s += (double)p.partials[(int64_t)b * p.tiles + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (float)(((dred[0] + dred[1]) + (dred[2] + dred[3])) / ((double)p.T * (double)F));
  bool any = false;
  for (int m = m0; m < m1; ++m) {
    const int axis = p.masks[3 * m], lo = max(p.masks[3 * m + 1], 0), hi = p.masks[3 * m + 2];
    if (axis == 2) {
      for (int c = lo + threadIdx.x; c < min(hi, F); c += 256) colf[c] = 1;
      any = any || hi > lo;
    } else {
      const int a = max(lo, r0), e = min(hi, r0 + nrows);
      for (int r = a + threadIdx.x; r < e; r += 256) rowf[r - r0] = 1;
      any = any || e > a;
    }
  }
  if (!any) return;  // uniform across the block
  __syncthreads();
  float* __restrict__ dst = p.out + ((int64_t)b * p.T + r0) * F;
  const int lane = threadIdx.x & 63;
  for (int rr = threadIdx.x >> 6; rr < nrows; rr += 4) {  // a wave per row: no index arithmetic, 256-byte store bursts
    float* __restrict__ row = dst + (int64_t)rr * F;
    if (rowf[rr]) {
      for (int c = lane; c < F; c += 64) row[c] = mean;
    } else {
      for (int c = lane; c < F; c += 64)
        if (colf[c]) row[c] = mean;
    }
  }
}

// GlobalMVN: (x - mean) / std, or x * std + mean (two roundings, like the two torch ops)
__global__ __launch_bounds__(256) void global_mvn_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ means,
                                                         const float* __restrict__ stds, int64_t n, int32_t F, int32_t inverse) {
#pragma clang fp contract(off)  // two roundings, like the reference's two torch ops
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % F);
    const float x = in[i];
    const float prod = x * stds[c], diff = x - means[c];  // plain operators: the pragma covers this block, not HIP's inline helpers
    out[i] = inverse ? prod + means[c] : diff / stds[c];
  }
}

}  // namespace hipfeat
