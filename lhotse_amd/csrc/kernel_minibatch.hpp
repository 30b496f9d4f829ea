// On-the-fly mini-batch (BASELINE configs[4]; lhotse/dataset/input_strategies.py:410-462 with PerturbSpeed-ed cuts,
// lhotse/dataset/cut_transforms/perturb_speed.py:8-47): everything that has to happen in front of the feature launch, in ONE launch.
//
//   * mixed-factor speed perturbation: every perturbed cut carries the index of its polyphase bank (ResCut::pad = kind), the banks of all
//     kinds are resident, a workgroup = 256 hops of one cut through resample_fast_block<ORIG, NEW, WIDTH> of its kind (bit-identical to
//     the per-factor launches of hipfeat_resample);
//   * the padding rows [T_b, rows_per_cut) of the collated (B, rows_per_cut, F) tensor (collate_matrices' padding_value,
//     lhotse/dataset/collation.py:506-535): they do not depend on the features, so they are written here, next to the resampler's
//     traffic, instead of in a third launch behind the feature kernel;
//   * the descriptor table of the feature launch.  For mini-batches of up to ~100 cuts BOTH descriptor tables travel in the kernel
//     arguments (3.3 KB of the 4 KB a launch may carry, hidden arguments included): workgroup 0 copies the CutDesc table to HBM, where the feature kernel -- stream
//     order -- finds it; no staging buffer, no host -> device copy and none of its latency in front of the launch.  Larger mini-batches
//     stage the tables through pinned memory as hipfeat_extract does.
#pragma once
#include "common.hpp"
#include "kernel_resample.hpp"

namespace hipfeat {

constexpr int kMbKinds = 4;            // <9,10,7> <11,10,7> <19,20,7> <21,20,7>
constexpr int kMbInlineBytes = 3328;   // descriptor bytes a launch carries in its kernel arguments
constexpr int kMbFillBlocks = 8;       // padding workgroups per cut

struct MbHeader {
  float* arena;              // inputs in front, resampled cuts behind them (ResCut offsets are arena offsets)
  float* out;                // (B, rows_per_cut, F) dense
  CutDesc* cuts_dst;         // HBM descriptor table of the feature launch (written here when the tables are inline)
  const ResCut* res_src;     // staged tables (nullptr = inline)
  const float* kt[kMbKinds]; // transposed banks [KW][NEWP] per kind
  int32_t num_cuts, num_res, res_blocks, copy_descs;
  int32_t rows_per_cut, feature_dim;
  float pad_value;
  int32_t pad_;
};
struct MbInlineArgs {
  MbHeader h;
  unsigned char blob[kMbInlineBytes];  // ResCut[num_res], then CutDesc[num_cuts]
};


typedef int mb_i4 __attribute__((ext_vector_type(4)));

// Where the two descriptor tables are read from: HBM (staged) ...
template <typename T>
__device__ __forceinline__ T mb_uniform(const T& v) {  // every field of a 32-byte descriptor through v_readfirstlane: the index was workgroup-uniform
  static_assert(sizeof(T) == 32, "descriptor size");
  union {
    int w[8];
    T t;
  } u;
  u.t = v;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.w[i] = __builtin_amdgcn_readfirstlane(u.w[i]);
  return u.t;
}
struct MbTablesGlobal {
  const ResCut* res;
  const CutDesc* cds;
  __device__ __forceinline__ ResCut res_at(int i) const { return mb_uniform(res[i]); }
  __device__ __forceinline__ int res_first_block(int i) const { return __builtin_amdgcn_readfirstlane(res[i].first_block); }
  __device__ __forceinline__ CutDesc cut_at(int i) const { return mb_uniform(cds[i]); }
  __device__ __forceinline__ mb_i4 cut_words(int k) const { return reinterpret_cast<const mb_i4*>(cds)[k]; }
};
// ... or the kernel-argument segment (constant address space: wave-uniform indices become scalar loads).  Going through the by-value
// parameter with a run-time index would have hipcc copy all 3.5 KB to scratch first.
typedef const __attribute__((address_space(4))) mb_i4* MbConstWords;
struct MbTablesKernarg {
  MbConstWords res4, cds4;  // 2 x int4 per descriptor (both are 32 bytes)
  template <typename T>
  __device__ __forceinline__ static T pair(MbConstWords p, int i) {
    static_assert(sizeof(T) == 32, "descriptor size");
    union {
      mb_i4 w[2];
      T t;
    } u;
    u.w[0] = p[2 * i];
    u.w[1] = p[2 * i + 1];
    return u.t;
  }
  __device__ __forceinline__ ResCut res_at(int i) const { return pair<ResCut>(res4, i); }
  __device__ __forceinline__ int res_first_block(int i) const { return res4[2 * i + 1].z; }  // ResCut::first_block = dword 6
  __device__ __forceinline__ CutDesc cut_at(int i) const { return pair<CutDesc>(cds4, i); }
  __device__ __forceinline__ mb_i4 cut_words(int k) const { return cds4[k]; }
};
static_assert(offsetof(ResCut, first_block) == 24 && sizeof(ResCut) == 32 && sizeof(CutDesc) == 32, "descriptor layout");

template <typename Tables>
__device__ __forceinline__ void minibatch_prep_body(const MbHeader& h, const Tables& tb, float* xs) {
  const int blk = blockIdx.x;
  if (blk < h.res_blocks) {
    int lo = 0, hi = h.num_res - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tb.res_first_block(mid) <= blk) lo = mid; else hi = mid - 1;
    }
    const ResCut cd = tb.res_at(lo);
    const int b = blk - cd.first_block;
    switch (cd.pad) {  // workgroup-uniform
      case 0: resample_fast_block<9, 10, 7>(h.arena, h.arena, cd, b, h.kt[0], xs); break;
      case 1: resample_fast_block<11, 10, 7>(h.arena, h.arena, cd, b, h.kt[1], xs); break;
      case 2: resample_fast_block<19, 20, 7>(h.arena, h.arena, cd, b, h.kt[2], xs); break;
      default: resample_fast_block<21, 20, 7>(h.arena, h.arena, cd, b, h.kt[3], xs); break;
    }
    return;
  }
  const int f = blk - h.res_blocks;
  const int cut = f / kMbFillBlocks, part = f - cut * kMbFillBlocks;
  if (cut >= h.num_cuts) return;
  const CutDesc cd = tb.cut_at(cut);
  if (h.copy_descs && f == 0) {  // one workgroup publishes the feature launch's table (2 x 16 bytes per cut)
    const int n4 = h.num_cuts * (int)(sizeof(CutDesc) / 16);
    mb_i4* dst = reinterpret_cast<mb_i4*>(h.cuts_dst);
    for (int k = threadIdx.x; k < n4; k += 256) dst[k] = tb.cut_words(k);
  }
  // padding rows of this cut: contiguous (the collated tensor is dense), 16-byte stores where the start allows it
  const int64_t n = (int64_t)(h.rows_per_cut - cd.num_frames) * h.feature_dim;
  if (n <= 0) return;
  float* __restrict__ base = h.out + (cd.out_row + cd.num_frames) * (int64_t)h.feature_dim;
  const int64_t lead = min<int64_t>(n, (int64_t)((4 - ((reinterpret_cast<uintptr_t>(base) >> 2) & 3)) & 3));
  const int64_t n4 = (n - lead) >> 2;
  const float v = h.pad_value;
  if (part == 0) {
    if ((int64_t)threadIdx.x < lead) base[threadIdx.x] = v;
    const int64_t tail0 = lead + 4 * n4;
    if (tail0 + threadIdx.x < n) base[tail0 + threadIdx.x] = v;
  }
  float4* __restrict__ b4 = reinterpret_cast<float4*>(base + lead);
  const float4 v4 = make_float4(v, v, v, v);
  for (int64_t k = (int64_t)part * 256 + threadIdx.x; k < n4; k += kMbFillBlocks * 256) b4[k] = v4;
}

// LDS: dynamic, sized by the widest kind the bank holds (11.3 KB for 0.9 / 1.1 only, 21.6 KB with 0.95 / 1.05); six workgroups per CU
__global__ __launch_bounds__(256, 6) void minibatch_prep_inline_kernel(const MbInlineArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const __attribute__((address_space(4))) unsigned char* ka = (const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
  MbTablesKernarg tb;
  tb.res4 = (MbConstWords)(ka + offsetof(MbInlineArgs, blob));
  tb.cds4 = (MbConstWords)(ka + offsetof(MbInlineArgs, blob) + (size_t)a.h.num_res * sizeof(ResCut));
  minibatch_prep_body(a.h, tb, xs);
}

__global__ __launch_bounds__(256, 6) void minibatch_prep_kernel(const MbHeader h) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  MbTablesGlobal tb{h.res_src, h.cuts_dst};
  minibatch_prep_body(h, tb, xs);
}

}  // namespace hipfeat
