// On-the-fly mini-batch (BASELINE configs[4]; lhotse/dataset/input_strategies.py:410-462 with PerturbSpeed-ed cuts,
// lhotse/dataset/cut_transforms/perturb_speed.py:8-47): everything that has to happen in front of the feature launch, in ONE launch.
//
//   * mixed-factor speed perturbation: every perturbed cut carries the index of its polyphase bank (ResCut::pad = kind), the banks of
//     both kinds are resident, a work item = 256 hops of one cut through resample_fast_block<ORIG, NEW, WIDTH> of its kind (bit-identical
//     to the per-factor launches of hipfeat_resample);
//   * the padding rows [T_b, rows_per_cut) of the collated (B, rows_per_cut, F) tensor (collate_matrices' padding_value,
//     lhotse/dataset/collation.py:506-535) -- of each of the K tensors when the launch pair serves K mini-batches at once (a prefetching
//     loader: fewer, larger launches): they do not depend on the features, so they are written here, next to the resampler's traffic,
//     instead of in a third launch behind the feature kernel; a work item = 16 KB of one cut's padding;
//   * the descriptor table of the feature launch.  For mini-batches of up to ~60 cuts ALL tables travel in the kernel arguments (3.3 KB
//     of the 4 KB a launch may carry, hidden arguments included): one workgroup copies the CutDesc table to HBM, where the feature kernel
//     -- stream order -- finds it; no staging buffer, no host -> device copy and none of its latency in front of the launch.  Larger
//     mini-batches stage the tables through pinned memory as hipfeat_extract does.
//
// Work distribution (third version; the two before it are in DESIGN.md section 4.6).  The launch is ONE list of items -- the padding items
// first (pure stores: they overlap with the resampling instead of trailing it), then the resampler's -- over a grid of a few
// workgroups per CU that take items round-robin, so that a 30 s cut's 200 items spread over the whole chip like a 1 s cut's 7.  Every
// workgroup first copies the tables (<= 24 KB) from the kernel-argument segment / HBM to LDS -- one round trip -- and then locates each
// of its items by bisection over prefix sums in LDS.  (Version 1 bisected over the table in the kernel-argument segment: six dependent
// scalar loads from there cost more than the resampling.  Version 2 gave every cut its own grid row: no search, but the workgroups of a
// long cut's row worked through 3-4 items each while those of the short cuts' rows had none.)
#pragma once
#include "common.hpp"
#include "kernel_resample.hpp"

namespace hipfeat {

constexpr int kMbKinds = 2;            // <9,10,7> <11,10,7>: speed 0.9 / 1.1 (wider banks -- 0.95 / 1.05 -- would set the register and LDS budget of
                                       // every workgroup of the launch; they keep their per-factor launches)
constexpr int kMbInlineBytes = 3328;   // table bytes a launch carries in its kernel arguments
constexpr int kMbLdsTableBytes = 24576;  // tables up to this size are searched in LDS, larger ones where they are (HBM)
constexpr int kMbFillFloats = 4096;    // floats per padding item (256 lanes x 4 x float4)
constexpr int kMbXsFloats = ResampleFast<11, 10, 7>::LDS_FLOATS > ResampleFast<9, 10, 7>::LDS_FLOATS ? ResampleFast<11, 10, 7>::LDS_FLOATS
                                                                                                        : ResampleFast<9, 10, 7>::LDS_FLOATS;

// Table blob (kernel arguments, or staged in HBM): ResCut[num_res] (first_block = exclusive prefix sum of the cuts' resampler items) |
// CutDesc[num_cuts] | int32 fill_first[num_cuts + 1] (prefix sum of the cuts' padding items) | int32 rows_per_cut[num_cuts]
struct MbHeader {
  float* arena;              // inputs in front, resampled cuts behind them (ResCut offsets are arena offsets)
  float* out;                // the collated tensor(s), dense
  CutDesc* cuts_dst;         // HBM descriptor table of the feature launch (written here when the tables are inline)
  const unsigned char* tables;  // staged blob (nullptr = inline)
  const float* kt[kMbKinds]; // transposed banks [KW][NEWP] per kind
  int32_t num_cuts, num_res, fill_items, res_items;
  int32_t table_bytes, copy_descs, feature_dim;
  float pad_value;
  int32_t feature_blocks, quads_per_wg;  // workgroups of the feature launch (its workgroup -> cut map is written next to the descriptor table);
                                         // > 0: the launch is laid out by frame quads (fft512c FLAT), CutDesc::first_block = a cut's first quad
};
struct MbInlineArgs {
  MbHeader h;
  alignas(16) unsigned char blob[kMbInlineBytes];  // (read back 16 bytes per lane)
};
static_assert(offsetof(MbInlineArgs, blob) % 16 == 0 && sizeof(MbInlineArgs) <= 3584, "kernel-argument layout");

typedef int mb_i4 __attribute__((ext_vector_type(4)));

// first index i in [0, n) with prefix[i * stride] > v, minus one (prefix[0] == 0 <= v): the owner of item v.  Wave-uniform.
__device__ __forceinline__ int mb_owner(const int32_t* prefix, int stride, int n, int v) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (prefix[mid * stride] <= v) lo = mid; else hi = mid - 1;
  }
  return __builtin_amdgcn_readfirstlane(lo);
}

template <typename T>
__device__ __forceinline__ T mb_uniform32(const T* p) {  // a 32-byte descriptor, every dword through v_readfirstlane (the index was workgroup-uniform)
  static_assert(sizeof(T) == 32, "descriptor size");
  const int* w = reinterpret_cast<const int*>(p);
  union {
    int w[8];
    T t;
  } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.w[i] = __builtin_amdgcn_readfirstlane(w[i]);
  return u.t;
}

// tb: the blob (LDS or HBM, generic pointer); xs: kMbXsFloats floats of LDS
__device__ __forceinline__ void minibatch_prep_body(const MbHeader& h, const unsigned char* tb, float* xs) {
  const ResCut* res = reinterpret_cast<const ResCut*>(tb);
  const CutDesc* cds = reinterpret_cast<const CutDesc*>(tb + (size_t)h.num_res * sizeof(ResCut));
  const int32_t* fill_first = reinterpret_cast<const int32_t*>(tb + (size_t)h.num_res * sizeof(ResCut) + (size_t)h.num_cuts * sizeof(CutDesc));
  const int32_t* rows = fill_first + h.num_cuts + 1;
  if (h.copy_descs && blockIdx.x == 0) {  // one workgroup publishes the feature launch's table (2 x 16 bytes per cut)
    const int n4 = h.num_cuts * (int)(sizeof(CutDesc) / 16);
    const mb_i4* src = reinterpret_cast<const mb_i4*>(cds);
    mb_i4* dst = reinterpret_cast<mb_i4*>(h.cuts_dst);
    for (int k = threadIdx.x; k < n4; k += 256) dst[k] = src[k];
    // ... and its workgroup -> cut map behind it (common.hpp::block_cut_map): one lane per cut writes the cut's run of workgroups
    int32_t* map = reinterpret_cast<int32_t*>(h.cuts_dst + h.num_cuts);
    if (h.quads_per_wg > 0) {  // one lane per workgroup: the cut that holds its first quad (bisection over the cuts' first quads)
      for (int g = threadIdx.x; g < h.feature_blocks; g += 256) {
        const int q = g * h.quads_per_wg;
        int lo = 0, hi = h.num_cuts - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (cds[mid].first_block <= q) lo = mid; else hi = mid - 1;
        }
        map[g] = lo;
      }
    } else {
      for (int c = threadIdx.x; c < h.num_cuts; c += 256) {
        const int b0 = cds[c].first_block, b1 = c + 1 < h.num_cuts ? cds[c + 1].first_block : h.feature_blocks;
        for (int b = b0; b < b1; ++b) map[b] = c;
      }
    }
  }
  const int total = h.fill_items + h.res_items;
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    if (item < h.fill_items) {
      const int cut = mb_owner(fill_first, 1, h.num_cuts, item);
      const CutDesc cd = mb_uniform32(cds + cut);
      const int chunk = item - __builtin_amdgcn_readfirstlane(fill_first[cut]);
      const int64_t n = (int64_t)(__builtin_amdgcn_readfirstlane(rows[cut]) - cd.num_frames) * h.feature_dim;  // floats of padding of this cut
      float* __restrict__ base = h.out + (cd.out_row + cd.num_frames) * (int64_t)h.feature_dim + (int64_t)chunk * kMbFillFloats;
      const int m = (int)min<int64_t>(kMbFillFloats, n - (int64_t)chunk * kMbFillFloats);
      const float v = h.pad_value;
      if ((reinterpret_cast<uintptr_t>(base) & 15) == 0) {
        const float4 v4 = make_float4(v, v, v, v);
        float4* __restrict__ b4 = reinterpret_cast<float4*>(base);
#pragma unroll
        for (int k = 0; k < kMbFillFloats / 1024; ++k)
          if (4 * (threadIdx.x + 256 * k) + 3 < m) b4[threadIdx.x + 256 * k] = v4;
        const int tail0 = m & ~3;
        if (tail0 + (int)threadIdx.x < m) base[tail0 + threadIdx.x] = v;
      } else {
        for (int k = threadIdx.x; k < m; k += 256) base[k] = v;
      }
    } else {
      const int p = item - h.fill_items;
      const int i = mb_owner(&res[0].first_block, (int)(sizeof(ResCut) / sizeof(int32_t)), h.num_res, p);
      const ResCut cd = mb_uniform32(res + i);
      const int b = p - cd.first_block;
      if (cd.pad == 0) resample_fast_block<9, 10, 7>(h.arena, h.arena, cd, b, h.kt[0], xs);
      else resample_fast_block<11, 10, 7>(h.arena, h.arena, cd, b, h.kt[1], xs);
      __syncthreads();  // the item's results have left LDS before the next item's samples arrive
    }
  }
}

// LDS: the resampler's span buffer, then the tables
__global__ __launch_bounds__(256) void minibatch_prep_inline_kernel(const MbInlineArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[kMbXsFloats];
  __shared__ __attribute__((aligned(16))) unsigned char tb[kMbInlineBytes];
  // the tables are read where the launch put them: the kernel-argument segment (constant address space), 16 bytes per lane, once
  const __attribute__((address_space(4))) mb_i4* src =
      (const __attribute__((address_space(4))) mb_i4*)((const __attribute__((address_space(4))) unsigned char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                       offsetof(MbInlineArgs, blob));
  for (int k = threadIdx.x; 16 * k < a.h.table_bytes; k += 256) reinterpret_cast<mb_i4*>(tb)[k] = src[k];
  __syncthreads();
  minibatch_prep_body(a.h, tb, xs);
}

__global__ __launch_bounds__(256) void minibatch_prep_kernel(const MbHeader h) {
  __shared__ __attribute__((aligned(16))) float xs[kMbXsFloats];
  extern __shared__ __attribute__((aligned(16))) unsigned char tb_dyn[];  // table_bytes when they fit kMbLdsTableBytes, else nothing
  const unsigned char* tb = h.tables;
  if (h.table_bytes <= kMbLdsTableBytes) {
    for (int k = threadIdx.x; 16 * k < h.table_bytes; k += 256) reinterpret_cast<mb_i4*>(tb_dyn)[k] = reinterpret_cast<const mb_i4*>(h.tables)[k];
    __syncthreads();
    tb = tb_dyn;
  }
  minibatch_prep_body(h, tb, xs);
}

}  // namespace hipfeat
