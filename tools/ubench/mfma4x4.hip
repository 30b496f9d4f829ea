// v_mfma_f32_4x4x1_16B_f32 on gfx950: operand layout probe + issue / dependent-accumulator timing.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void layout(float* out) {
  const int l = threadIdx.x;
  // A = 100*block + row i ; B = 1000 * (1 + col j) ... encode so that D = A*B identifies (block, i, j)
  const float a = (float)(1 + (l >> 2)) + 0.25f * (l & 3);   // block+1 . i/4
  const float b = (float)(1 << (l & 3)) * (1 + 16 * (l >> 2));  // 2^j * (1 + 16 block)
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <int CHAINS>
__global__ void rate(float* out, int iters, unsigned long long* clk) {
  const float a = threadIdx.x * 0.001f, b = 1.0f;
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
      if (CHAINS > 1) c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
      if (CHAINS > 2) c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
      if (CHAINS > 3) c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int CHAINS>
void run() {
  const int iters = 1000, blocks = 256;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 4 * 8);
  rate<CHAINS><<<blocks, 256>>>(out, iters, clk);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 1024; ++i) avg += h[i]; avg /= 1024;
  printf("mfma_f32_4x4x1_16B chains=%d: %.2f clk per MFMA (1 wave/SIMD)\n", CHAINS, avg / (iters * 8.0 * CHAINS));
  hipFree(out); hipFree(clk);
}
int main() {
  float* out; hipMalloc(&out, 64 * 4 * 4);
  layout<<<1, 64>>>(out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  // expected if A: lane = 4*blk + i, B: lane = 4*blk + j, D: vgpr i, lane 4*blk + j:  D = (blk+1+i/4) * 2^j*(1+16 blk)
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l >> 2, j = l & 3, i = r;
      const float want = ((float)(1 + blk) + 0.25f * i) * ((float)(1 << j) * (1 + 16 * blk));
      if (h[l * 4 + r] != want) { ok = 0; if (l < 8) printf("lane %d vgpr %d: got %g want %g\n", l, r, h[l * 4 + r], want); }
    }
  printf("layout A[lane=4b+i], B[lane=4b+j], D[vgpr i][lane 4b+j]: %s\n", ok ? "CONFIRMED" : "MISMATCH");
  run<1>(); run<2>(); run<4>();
  return 0;
}
