// Shared device/host definitions for libhipfeat (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hipfeat {

// Experiment builds (-DHIPFEAT_LDS_POISON): every feature kernel fills its dynamic LDS with NaN before it starts, so that
// a read of a location the kernel never wrote (harmless-looking when it meets a zero weight and the leftover bits of the
// previous kernel happen to be finite) shows up as NaN in the GPU tests.  tools/lds_poison.sh runs the suite on such a build.
#ifdef HIPFEAT_LDS_POISON
#ifndef HIPFEAT_LDS_POISON_FROM
#define HIPFEAT_LDS_POISON_FROM 0  // bisecting aid: poison only [FROM, TO) floats
#define HIPFEAT_LDS_POISON_TO (1 << 30)
#endif
__device__ int g_lds_poison_floats;
#define HF_POISON_LDS(smem_)                                                                                           \
  do {                                                                                                                 \
    for (int i_ = threadIdx.x + HIPFEAT_LDS_POISON_FROM; i_ < min(g_lds_poison_floats, HIPFEAT_LDS_POISON_TO); i_ += blockDim.x)   \
      (smem_)[i_] = __builtin_nanf("");                                                                                \
    __syncthreads();                                                                                                   \
  } while (0)
#define HF_POISON_ARRAY(arr_, n_)                                                                                      \
  do {                                                                                                                 \
    for (int i_ = threadIdx.x; i_ < (int)(n_); i_ += blockDim.x) (arr_)[i_] = __builtin_nanf("");                       \
    __syncthreads();                                                                                                   \
  } while (0)
#else
#define HF_POISON_LDS(smem_)
#define HF_POISON_ARRAY(arr_, n_)
#endif

// One cut of a batch, as the kernels see it (32 bytes, device resident).
struct CutDesc {
  int64_t wave_off;     // first sample of the cut in the waveform buffer (elements)
  int64_t out_row;      // first output row of the cut
  int32_t num_samples;  // S
  int32_t padded_len;   // P: row length the edge reflection is taken on (== S unless batch_zero_pad)
  int32_t num_frames;   // rows to produce
  int32_t first_block;  // index of the cut's first workgroup in the grid (exclusive prefix sum)
};

enum : int32_t {
  F_REMOVE_DC = 1 << 0,
  F_USE_ENERGY = 1 << 1,
  F_RAW_ENERGY = 1 << 2,
  F_FFT_MAG = 1 << 3,
  F_LIFTER = 1 << 4,
  F_POW2 = 1 << 5,
  F_LOG_SPEC = 1 << 6,
  F_LOG10 = 1 << 8,   // mel epilogue in log10 (Whisper)
  F_CENTER = 1 << 7,  // torch.stft(center=True, pad_mode="reflect") framing (Whisper)
};

// Kernel arguments of the generic kernel (passed by value; lives in the kernarg segment).
struct GenericParams {
  const float* wave;
  float* out;
  const CutDesc* cuts;
  const float* window;     // [N]
  const float2* tw;        // pow2: W_fft^k, k < fft/2 ; otherwise W_fft^k, k < fft
  const float* mel;        // [K][M]
  const int2* mel_range;   // [M] first bin, one-past-last bin with a non-zero weight
  const float* dct;        // [M][C]
  const float* lifter;     // [C]
  int64_t out_stride;      // floats per output row
  int32_t num_cuts;
  int32_t uniform_bpc;     // >0: every cut has exactly this many workgroups (no search needed)
  int32_t N, shift, fft, H, log2H, K, M, C;
  int32_t kind, flags, fpb, npad_left;
  float preemph, log_energy_floor, mel_floor, log_offset;
  // LDS carve-up (float offsets)
  int32_t span, off_z, off_p, off_tw, off_stat, off_mel;
};

__device__ __forceinline__ float load_sample(const float* __restrict__ w, int64_t j, int32_t S, int32_t P) {
  // Index restatement of the flip/cat of layers.py:756-764: frame t, tap i reads j = shift*t - npad_left + i,
  // j<0 -> -j-1, j>=P -> 2P-1-j; indices in [S, P) are the zero padding of a batch row.
  if (j < 0) j = -j - 1;
  if (j >= P) j = 2 * (int64_t)P - 1 - j;
  return (j >= 0 && j < S) ? w[j] : 0.0f;
}

// torch.stft "reflect" padding: the edge sample is not repeated (j<0 -> -j, j>=S -> 2S-2-j).
__device__ __forceinline__ float load_sample_center(const float* __restrict__ w, int64_t j, int32_t S) {
  if (j < 0) j = -j;
  if (j >= S) j = 2 * (int64_t)S - 2 - j;
  return (j >= 0 && j < S) ? w[j] : 0.0f;
}

// Ragged batches: the host puts a workgroup -> cut map (int32 per workgroup) behind the descriptor table and says so with
// uniform_bpc = -1: ONE dependent load instead of log2(cuts) of them in front of every workgroup (13 for a LibriSpeech-like batch of
// 8000 cuts -- a few microseconds of a workgroup that lives for ~40).
__device__ __forceinline__ const int32_t* block_cut_map(const CutDesc* __restrict__ cuts, int num_cuts) {
  return reinterpret_cast<const int32_t*>(cuts + num_cuts);
}

// A descriptor whose index is wave-uniform, every dword through v_readfirstlane: the fields then live in SGPRs (addresses, loop bounds and
// the "interior span" test of the kernels stay scalar) even where the compiler cannot prove the index uniform.
__device__ __forceinline__ CutDesc load_cut_uniform(const CutDesc* __restrict__ p) {
  static_assert(sizeof(CutDesc) == 32, "descriptor size");
  const int* w = reinterpret_cast<const int*>(p);
  union {
    int w[8];
    CutDesc d;
  } u;
#pragma unroll
  for (int i = 0; i < 8; ++i) u.w[i] = __builtin_amdgcn_readfirstlane(w[i]);
  return u.d;
}

// Locate the cut that owns workgroup `blk` (binary search over first_block): layouts without the map.
__device__ __forceinline__ int find_cut(const CutDesc* __restrict__ cuts, int num_cuts, int blk) {
  int lo = 0, hi = num_cuts - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (cuts[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

}  // namespace hipfeat
