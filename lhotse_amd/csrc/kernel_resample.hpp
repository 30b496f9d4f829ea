// Polyphase sinc resampler (speed perturbation), SURVEY.md section 8f row 1.
//
// Reference: ResampleTensor = _get_sinc_resample_kernel + _apply_sinc_resample_kernel
// (lhotse/augmentation/resample.py:184-315), reached from Speed.__call__ (lhotse/augmentation/torchaudio.py:37-42).
// The reference zero-pads the waveform by (width, width + orig) and runs conv1d(stride = orig) with `new` filters
// of 2*width + orig taps:  y[j*new + ph] = sum_i xpad[j*orig + i] * K[ph][i].
//
// Here: one workgroup = `outs_per_block` consecutive output samples of one cut.  The input span those outputs
// touch is staged in LDS once (coalesced, zero outside the cut), the filter bank too when it is small (speed
// factors 0.9 / 1.1 give 10 x 23 / 10 x 25 taps); each lane then accumulates one output at a time as an
// ascending-tap fmaf chain and stores it coalesced.
#pragma once
#include "common.hpp"

namespace hipfeat {

struct ResCut {
  int64_t in_off, out_off;
  int32_t in_len, out_len;
  int32_t first_block, pad;
};

struct ResampleParams {
  const float* in;
  float* out;
  const ResCut* cuts;
  const float* kernel;  // [nw][kw]
  int32_t num_cuts, orig, nw, kw, width, outs_per_block, span_floats, kernel_in_lds;
};

__device__ __forceinline__ int find_res_cut(const ResCut* __restrict__ cuts, int num_cuts, int blk) {
  int lo = 0, hi = num_cuts - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (cuts[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void resample_kernel(const ResampleParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                  // [span_floats]
  float* kl = smem + p.span_floats;  // [nw * kw] when kernel_in_lds
  const int tid = threadIdx.x;
  const int cut = find_res_cut(p.cuts, p.num_cuts, blockIdx.x);
  const ResCut cd = p.cuts[cut];
  const int o0 = (blockIdx.x - cd.first_block) * p.outs_per_block;
  const int o1 = min(o0 + p.outs_per_block, cd.out_len);
  const int j0 = o0 / p.nw, j1 = (o1 - 1) / p.nw;
  const int span = (j1 - j0) * p.orig + p.kw;
  const int64_t x0 = (int64_t)j0 * p.orig - p.width;  // input index of xs[0]
  const float* __restrict__ x = p.in + cd.in_off;
  for (int i = tid; i < span; i += 256) {
    const int64_t s = x0 + i;
    xs[i] = (s >= 0 && s < cd.in_len) ? x[s] : 0.0f;
  }
  if (p.kernel_in_lds)
    for (int i = tid; i < p.nw * p.kw; i += 256) kl[i] = p.kernel[i];
  __syncthreads();
  const float* __restrict__ kt = p.kernel_in_lds ? kl : p.kernel;
  float* __restrict__ y = p.out + cd.out_off;
  for (int o = o0 + tid; o < o1; o += 256) {
    const int j = o / p.nw, ph = o - j * p.nw;
    const float* xr = xs + (j - j0) * p.orig;
    const float* kr = kt + ph * p.kw;
    float acc = 0.f;
    for (int i = 0; i < p.kw; ++i) acc = fmaf(xr[i], kr[i], acc);
    y[o] = acc;
  }
}

// --------------------------------------------------------------------------------------
// Fast path for the ratios speed perturbation actually uses (compile-time ORIG / NEW / WIDTH).
//
// One lane = one input hop j (ORIG input samples -> NEW output samples): it reads its KW = 2*WIDTH + ORIG taps of
// x from LDS once and feeds NEW independent accumulators, so one LDS read serves NEW FMAs.  The filter bank is
// wave-uniform: it is read through the scalar cache (`kt` = transposed bank [KW][NEWP], one row per tap) and the
// FMAs take it as an SGPR operand -- no LDS or VGPR traffic for coefficients.  Per accumulator the order is the
// same ascending-tap fmaf chain as the generic kernel (results are bit-identical between the two).
// The NEW results per lane are transposed through LDS (re-using the input span) so the stores are coalesced.
// --------------------------------------------------------------------------------------
template <int ORIG, int NEW, int WIDTH>
struct ResampleFast {
  static constexpr int KW = 2 * WIDTH + ORIG;
  static constexpr int NEWP = (NEW + 3) & ~3;       // row pitch of the transposed bank
  static constexpr int HOPS = 256;                  // hops per block (one per lane)
  static constexpr int SPAN = HOPS * ORIG + KW - ORIG;  // input samples the block touches
  static constexpr int OUTS = HOPS * NEW;
  static constexpr int LDS_FLOATS = SPAN > OUTS ? SPAN : OUTS;
};

// The work of one workgroup: hops [j0, j0 + 256) of cut `cd`.  `xs` = G::LDS_FLOATS floats of LDS.
template <int ORIG, int NEW, int WIDTH>
__device__ __forceinline__ void resample_fast_block(const float* __restrict__ in, float* __restrict__ out, const ResCut& cd, int block_in_cut,
                                                    const float* __restrict__ kt, float* xs) {
  using G = ResampleFast<ORIG, NEW, WIDTH>;
  const int tid = threadIdx.x;
  const int j0 = block_in_cut * G::HOPS;
  const int x0 = j0 * ORIG - WIDTH;  // j0 * ORIG <= in_len + ORIG < 2^31
  const float* __restrict__ x = in + cd.in_off;
  {
    constexpr int N = (G::SPAN + 255) / 256;
    float v[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {  // all loads in flight before the first LDS write
      const int s = x0 + tid + 256 * k;
      v[k] = ((unsigned)s < (unsigned)cd.in_len) ? x[s] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      if (tid + 256 * k < G::SPAN) xs[tid + 256 * k] = v[k];
  }
  __syncthreads();
  float acc[NEW];
#pragma unroll
  for (int ph = 0; ph < NEW; ++ph) acc[ph] = 0.f;
  const float* xr = xs + tid * ORIG;
  // the bank through the constant address space: wave-uniform, read-only for the whole launch -> scalar loads, SGPR operands.  (As a plain
  // pointer that came out of a struct the compiler has to assume that the kernel's own stores may alias it, fetches the 250 coefficients
  // with vector loads into 250 registers per lane -- and the mixed launch ran at half the rate of the per-factor ones.)
  const __attribute__((address_space(4))) float* ktc = (const __attribute__((address_space(4))) float*)kt;
#pragma unroll
  for (int i = 0; i < G::KW; ++i) {
    const float xv = xr[i];
#pragma unroll
    for (int ph = 0; ph < NEW; ++ph) acc[ph] = fmaf(xv, ktc[i * G::NEWP + ph], acc[ph]);
  }
  __syncthreads();
#pragma unroll
  for (int ph = 0; ph < NEW; ++ph) xs[tid * NEW + ph] = acc[ph];
  __syncthreads();
  const int o0 = j0 * NEW;
  const int n = min(G::OUTS, cd.out_len - o0);
  float* __restrict__ y = out + cd.out_off + o0;
#pragma unroll
  for (int k = 0; k < NEW; ++k)
    if (tid + 256 * k < n) y[tid + 256 * k] = xs[tid + 256 * k];
}

template <int ORIG, int NEW, int WIDTH>
__global__ __launch_bounds__(256) void resample_fast_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const ResCut* __restrict__ cuts, const float* __restrict__ kt,
                                                            int num_cuts) {
  using G = ResampleFast<ORIG, NEW, WIDTH>;
  __shared__ float xs[G::LDS_FLOATS];
  const int cut = find_res_cut(cuts, num_cuts, blockIdx.x);
  const ResCut cd = cuts[cut];
  resample_fast_block<ORIG, NEW, WIDTH>(in, out, cd, blockIdx.x - cd.first_block, kt, xs);
}

}  // namespace hipfeat
