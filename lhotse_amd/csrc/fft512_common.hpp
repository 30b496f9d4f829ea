// Shared definitions of the 16-frame-tile fast paths (kernel_fft512b.hpp, kernel_fft256.hpp): tile constants, the static
// schedule of the banded 16x16x4 matrix-core mel GEMM, kernel parameters, optional phase timers (experiment builds).
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
constexpr int kWaveRegion = 4 * kExFrameStride + 16;  // 2192 dwords per wave (== 16 mod 64)
constexpr int kPRowStride = 260;                      // dwords per power row (== 4 mod 64)
constexpr int kMaxGroups0 = 20;                       // 8-bin MFMA groups of a wave's first / second mel tile
constexpr int kMaxGroups1 = 4;
constexpr int kMelARegs = 2 * (kMaxGroups0 + kMaxGroups1);

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
#define HF_U(i)
#else
#define HF_T(i)
#define HF_U(i)
#define HF_ACC(slot, a, b)
#endif


}  // namespace hipfeat
