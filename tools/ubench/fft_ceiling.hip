// FFT-only ceiling of the fft512c design on this chip (task 1a).
//
// The product kernel (lhotse_amd/csrc/kernel_fft512c.hpp) is limited by the package power cap, not by HBM: what bounds it is the
// energy of one 512-point real FFT done the way that kernel does it (16 lanes per frame, complex 16 x 16 in registers with one LDS
// exchange, 128 VGPRs -> 4 waves per SIMD, 2 workgroups of 8 waves per CU).  This program runs exactly that arithmetic with the same
// occupancy and the same wave-private LDS footprint, but WITHOUT any HBM traffic, and in nested levels, so that the frames/s of each
// level under the power cap is a hard ceiling for every kernel of this design that contains it:
//
//   level 0  samples (noise, resident in the wave's LDS span buffer) -> window -> pass 1 -> twiddle -> LDS exchange -> pass 2
//            (the complex 256-point FFT alone; result folded into one register)
//   level 1  + split step -> |X|^2 -> power rows in LDS                     (= "the real FFT + power spectrum")
//   level 2  + DC removal and pre-emphasis in front of the window           (= S3 of the product kernel)
//   level 3  + the mel filterbank on v_mfma_f32_4x4x1 (2 sets x 16 steps) + log, result folded into one register (no stores)
//
// frames/s x 960 B / 8 TB/s is what `roofline.frac` of bench.py could reach if everything the level leaves out were free.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I lhotse_amd/csrc tools/ubench/fft_ceiling.hip -o tools/ubench/fft_ceiling
// Run (GPU box): tools/ubench/fft_ceiling <level> <seconds> [zeros]   -- prints one JSON line; tools/fft_ceiling.sh samples rocm-smi around it.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "fft_common.hpp"

using namespace hipfeat;

constexpr int kRowStride = 34, kFrameStride = 8 * kRowStride + 16, kPRow = 272, kRegion = 4 * kFrameStride + 16;
constexpr int kWaves = 8, kShift = 160, kN = 400, kNRows = 13, kNFull = 12;
constexpr int kSpan = 3 * kShift + 32 * kNRows;  // 896 floats
constexpr int kShared = kNRows * 32 + 512 + 256 + 2 * 16 * 64 + 2 * 256;  // window | pass twiddles | split twiddles | weights | lane table

__device__ __forceinline__ int mul24(int a, int b) { return (int)__umul24((unsigned)a, (unsigned)b); }
__device__ __forceinline__ float hash01(unsigned x) {
  x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;  // [-1, 1)
}

template <int LEVEL>
__global__ __launch_bounds__(64 * kWaves, 4) void ceiling_kernel(float* sink, int rounds, int zeros) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  v2* cwin = reinterpret_cast<v2*>(smem);
  v2* ctwp = cwin + kNRows * 16;
  v2* ctws = ctwp + 256;
  float* wtab = smem + kNRows * 32 + 512 + 256;
  float* ltab = wtab + 2 * 16 * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // constants: a povey-like window / 2, the two twiddle tables of the product kernel, a synthetic band schedule
  for (int i = tid; i < kNRows * 32; i += 64 * kWaves) smem[i] = i < kN ? 0.5f * powf(0.5f - 0.5f * cosf(6.2831853f * i / (kN - 1)), 0.85f) : 0.f;
  for (int i = tid; i < 256; i += 64 * kWaves) {
    const int k1 = i >> 4, qq = i & 15;
    float s, c;
    sincosf(-6.283185307179586f * (float)(qq * k1) / 256.f, &s, &c);
    ctwp[i] = v2{c, s};
  }
  for (int i = tid; i < 128; i += 64 * kWaves) {
    const int k2 = i >> 4, qq = i & 15;
    float s, c;
    sincosf(-6.283185307179586f * (float)(qq + 16 * k2) / 512.f, &s, &c);
    ctws[i] = v2{s, -c};  // -i W
  }
  for (int i = tid; i < 2 * 16 * 64; i += 64 * kWaves) wtab[i] = 0.25f + 0.5f * fabsf(hash01(i));
  for (int i = tid; i < 2 * 256; i += 64 * kWaves) {
    const int l = (i >> 2) & 63, s = i >> 8, f = i & 3;
    // slot = l >> 2, frame = l & 3: power-row offset of a 16-bin band (bank quads staggered like the product's schedule)
    const int poff = (l & 3) * kPRow + ((l >> 2) * 15 + s * 4) % 240;
    ltab[i] = f == 0 ? __builtin_bit_cast(float, poff & ~3) : f == 1 ? __builtin_bit_cast(float, l) : 0.f;
  }
  float* xs = smem + kShared + wv * (kSpan + kRegion);
  float* myreg = xs + kSpan;
  for (int i = lane; i < kSpan; i += 64) xs[i] = zeros ? 0.f : 0.5f * hash01((blockIdx.x * kWaves + wv) * 1024u + i);
  for (int i = lane; i < kRegion; i += 64) myreg[i] = 0.f;
  __syncthreads();
  v2 twpreg[16], twsreg[8];
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) twpreg[k1] = ctwp[k1 * 16 + (lane & 15)];
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) twsreg[k2] = ctws[k2 * 16 + (lane & 15)];
  const float inv_n = 1.0f / kN, c = 0.97f;
  float fold = 0.f;

  for (int r = 0; r < rounds; ++r) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int q = lane_o & 15, g = lane_o >> 4;
    v2 Z[16];
    {
      const float* x = xs + mul24(g, kShift) + 2 * q;
      v2 z[16], win[kNRows];
#pragma unroll
      for (int n1 = 0; n1 < kNRows; ++n1) {
        z[n1] = *reinterpret_cast<const v2*>(x + 32 * n1);
        asm volatile("");
      }
#pragma unroll
      for (int n1 = 0; n1 < kNRows; ++n1) {
        win[n1] = cwin[n1 * 16 + q];
        asm volatile("");
      }
      float pv[kNRows];
      if (LEVEL >= 2) {
#pragma unroll
        for (int n1 = 0; n1 < kNRows; ++n1) pv[n1] = n1 == 0 ? x[q == 0 ? 0 : -1] : x[32 * n1 - 1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (LEVEL >= 2) {
#pragma unroll
        for (int n1 = kNFull; n1 < kNRows; ++n1) {
          const int m0 = 32 * n1 + 2 * q;
          if (m0 >= kN) z[n1].x = 0.f;
          if (m0 + 1 >= kN) z[n1].y = 0.f;
        }
        v2 sa = z[0], sb = z[1], sc = z[2], sd = z[3];
#pragma unroll
        for (int n1 = 4; n1 < kNRows; ++n1) {
          if ((n1 & 3) == 0) sa += z[n1];
          if ((n1 & 3) == 1) sb += z[n1];
          if ((n1 & 3) == 2) sc += z[n1];
          if ((n1 & 3) == 3) sd += z[n1];
        }
        const v2 sum2 = (sa + sb) + (sc + sd);
        const float mu = row16_sum(sum2.x + sum2.y) * inv_n;
        const float nc = -c, mu1 = (1.0f - c) * mu;
#pragma unroll
        for (int n1 = 0; n1 < kNRows; ++n1) {
          v2 t;  // two plain v_fma_f32, as the product (kernel_fft512c.hpp)
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.x) : "s"(nc), "v"(pv[n1]), "v"(z[n1].x));
          asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t.y) : "s"(nc), "v"(z[n1].x), "v"(z[n1].y));
          z[n1] = (t - v2{mu1, mu1}) * win[n1];
        }
      } else {
#pragma unroll
        for (int n1 = 0; n1 < kNRows; ++n1) z[n1] = z[n1] * win[n1];
      }
#pragma unroll
      for (int n1 = kNRows; n1 < 16; ++n1) z[n1] = v2{0.f, 0.f};
      v2 a[16];
      fft16(z, a);
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) a[k1] = cmul2(a[k1], twpreg[k1]);
      float* exf = myreg + mul24(g, kFrameStride);
      v2 b[16];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) *reinterpret_cast<v2*>(exf + rr * kRowStride + 2 * q) = a[8 * h + rr];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (h == 0 || q >= 8) {
#pragma unroll
          for (int n2 = 0; n2 < 16; ++n2) {
            b[n2] = *reinterpret_cast<const v2*>(exf + (q % 8) * kRowStride + 2 * n2);
            asm volatile("");
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      fft16(b, Z);
    }
    if (LEVEL == 0) {
      v2 s = Z[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) s += Z[k];  // 15 packed adds stand in for nothing the product does: keeps all 16 outputs alive
      fold += s.x + s.y;
      continue;
    }
    {
      float* prow = myreg + mul24(g, kPRow);
      float* pown = prow + q;
      float* ppar = prow + ((16 - q) & 15) + (q == 0 ? 16 : 0);
      if (q < kPRow - 257) prow[257 + q] = 0.f;
      float t1[16];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].x);
        t1[2 * k2 + 1] = dpp_mov<DPP_ROW_MIRROR>(Z[15 - k2].y);
      }
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        t1[2 * k2] = dpp_shr1_keep(Z[(16 - k2) & 15].x, t1[2 * k2]);
        t1[2 * k2 + 1] = dpp_shr1_keep(Z[(16 - k2) & 15].y, t1[2 * k2 + 1]);
      }
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        const v2 m = v2{t1[2 * k2], t1[2 * k2 + 1]};
        const v2 sp = m * HF_CJ + Z[k2];
        const v2 dm = m * HF_NCJ + Z[k2];
        const v2 tt = cmul2(dm, twsreg[k2]);
        v2 re2, im2, pw;  // the bin pair (k, 256 - k) side by side, as in the product
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(re2) : "v"(tt), "v"(HF_CJ), "v"(sp));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,1,1]" : "=v"(im2) : "v"(tt), "v"(HF_CJ), "v"(sp));
        pw = re2 * re2;
        pw = im2 * im2 + pw;
        pown[16 * k2] = pw.x;
        ppar[16 * (15 - k2)] = pw.y;
      }
      if (q == 0) prow[128] = 4.f * (Z[8].x * Z[8].x + Z[8].y * Z[8].y);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (LEVEL >= 3) {
      f32x4 av[2][4], bv[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float* pa = myreg + __builtin_bit_cast(int, ltab[s * 256 + 4 * lane_o]);
        const float* wb = wtab + s * (16 * 64) + 4 * lane_o;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
          av[s][c4] = *reinterpret_cast<const f32x4*>(pa + 4 * c4);
          bv[s][c4] = *reinterpret_cast<const f32x4*>(wb + c4 * 256);
        }
      }
      // as the product since round 3: the two sets as two interleaved accumulation chains, then mel4_reduce_floor (fft_common.hpp)
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[0][c4][i], bv[0][c4][i], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[1][c4][i], bv[1][c4][i], acc[1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float val[4];
        mel4_reduce_floor(acc[s], ltab[s * 256 + 4 * lane_o + 2], ltab[s * 256 + 4 * lane_o + 3], 1.1920929e-07f, val);
#pragma unroll
        for (int i = 0; i < 4; ++i) fold += fast_log(val[i]);
      }
    } else {
      fold += myreg[lane_o];  // one LDS read keeps the power rows observable
    }
  }
  if (fold == 123.456f) sink[blockIdx.x * 64 * kWaves + tid] = fold;  // never true for this data; keeps the arithmetic live
}

template <int LEVEL>
static double run(double seconds, int zeros, int* out_rounds, int* out_blocks) {
  const int blocks = 256 * 2 * 4;  // four workgroups per resident slot, like the product's layouts
  const int rounds = 16;           // 16 rounds of 4 frames per wave, like the product's 10 000-cut launch
  const size_t lds = (size_t)(kShared + kWaves * (kSpan + kRegion)) * 4;
  float* sink;
  hipMalloc(&sink, (size_t)blocks * 64 * kWaves * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ceiling_kernel<LEVEL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // one launch = blocks x 8 waves x rounds x 4 frames; the product's launch runs 10 000 x 1000 frames: 39 launches here are one of those
  const int per = 39;
  for (int i = 0; i < per; ++i) ceiling_kernel<LEVEL><<<blocks, 64 * kWaves, lds>>>(sink, rounds, zeros);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  double frames = 0, ms_total = 0;
  while (ms_total < seconds * 1000.0) {
    hipEventRecord(a);
    for (int rep = 0; rep < 20; ++rep)
      for (int i = 0; i < per; ++i) ceiling_kernel<LEVEL><<<blocks, 64 * kWaves, lds>>>(sink, rounds, zeros);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms_total += ms;
    frames += 20.0 * per * (double)blocks * kWaves * rounds * 4;
  }
  hipFree(sink);
  *out_rounds = rounds, *out_blocks = blocks;
  return frames / (ms_total * 1e-3);
}

int main(int argc, char** argv) {
  const int level = argc > 1 ? atoi(argv[1]) : 1;
  const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
  const int zeros = argc > 3 && !strcmp(argv[3], "zeros");
  int rounds = 0, blocks = 0;
  double fps = 0;
  switch (level) {
    case 0: fps = run<0>(seconds, zeros, &rounds, &blocks); break;
    case 1: fps = run<1>(seconds, zeros, &rounds, &blocks); break;
    case 2: fps = run<2>(seconds, zeros, &rounds, &blocks); break;
    default: fps = run<3>(seconds, zeros, &rounds, &blocks); break;
  }
  printf("{\"level\": %d, \"input\": \"%s\", \"frames_per_s\": %.4g, \"cut_equiv_per_s\": %.4g, \"frac_of_hbm_if_rest_free\": %.4f, \"workgroups\": %d, \"rounds\": %d}\n",
         level, zeros ? "zeros" : "noise", fps, fps / 1000.0, fps * 960.0 / 8e12, blocks, rounds);
  return 0;
}
