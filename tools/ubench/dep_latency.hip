// Micro-benchmark: dependent-issue latency of VALU instructions (one wave per SIMD, chains of length 1/2/4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE, int CH>
__global__ void k(float* out, int iters, unsigned long long* clk) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2};
  const float c = 1.0001f; const v2 c2 = {1.0001f, 0.9999f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // v_pk_add_f32
      if (CH == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n" : "+v"(p0) : "v"(c2));) }
      if (CH == 2) { REP16(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n" : "+v"(p0), "+v"(p1) : "v"(c2));) }
      if (CH == 4) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2));) }
    } else if (MODE == 1) {  // v_add_f32
      if (CH == 1) { REP16(asm volatile("v_add_f32 %0, %0, %1\n" : "+v"(a0) : "v"(c));) }
      if (CH == 2) { REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n" : "+v"(a0), "+v"(a1) : "v"(c));) }
      if (CH == 4) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c));) }
    } else if (MODE == 2) {  // v_pk_fma_f32
      if (CH == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n" : "+v"(p0) : "v"(c2));) }
      if (CH == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1\n" : "+v"(p0), "+v"(p1) : "v"(c2));) }
      if (CH == 4) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c2));) }
    } else if (MODE == 3) {  // v_mov_b32_dpp dependent on previous
      if (CH == 1) { REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n" : "+v"(a0));) }
      if (CH == 2) { REP16(asm volatile("s_nop 0\n v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n" : "+v"(a0), "+v"(a1));) }
      if (CH == 4) { REP16(asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}
template <int MODE, int CH>
void run(const char* name) {
  const int iters = 500, blocks = 256;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 4 * 8);
  k<MODE, CH><<<blocks, 256>>>(out, iters, clk);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 1024; ++i) avg += h[i]; avg /= 1024;
  printf("%-16s chains=%d  %.2f clk per instruction (1 wave/SIMD)\n", name, CH, avg / (iters * 16.0 * CH));
  hipFree(out); hipFree(clk);
}
int main() {
  run<0,1>("v_pk_add_f32"); run<0,2>("v_pk_add_f32"); run<0,4>("v_pk_add_f32");
  run<1,1>("v_add_f32"); run<1,2>("v_add_f32"); run<1,4>("v_add_f32");
  run<2,1>("v_pk_fma_f32"); run<2,2>("v_pk_fma_f32"); run<2,4>("v_pk_fma_f32");
  run<3,1>("v_mov_dpp"); run<3,2>("v_mov_dpp"); run<3,4>("v_mov_dpp");
  return 0;
}
