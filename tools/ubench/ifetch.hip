// Micro-benchmark: does instruction fetch bound long straight-line VALU code on gfx950?  The same packed-FMA stream
// (8-byte VOP3P encodings, 8 independent chains) as a short loop body (lives in the wave's instruction buffer / a few
// cache lines) and as long unrolled bodies of 4 / 16 / 64 KB, at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
#define PK8 "v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n" \
            "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
#define A8 "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n" \
           "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
#define R4(x) x x x x
#define R8(x) x x x x x x x x
// BODY = number of 8-instruction groups per loop iteration (64 B of code each for VOP3P, 32 B for VOP2)
template <int BODY, int KIND>
__global__ void k(float* out, int groups, unsigned long long* clk) {
  v2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, p4 = {2, 1}, p5 = {4, 3}, p6 = {6, 5}, p7 = {8, 7};
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
  const v2 c2 = {1.0001f, 0.9999f};
  const float c = 1.0001f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < groups / BODY; ++i) {
    if (KIND == 0) {
      if (BODY == 8) { R8(asm volatile(PK8 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));) }
      if (BODY == 64) { R8(R8(asm volatile(PK8 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)) }
      if (BODY == 256) { R4(R8(R8(asm volatile(PK8 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));))) }
      if (BODY == 1024) { R4(R4(R8(R8(asm volatile(PK8 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)))) }
    } else {
      if (BODY == 8) { R8(asm volatile(A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
      if (BODY == 64) { R8(R8(asm volatile(A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)) }
      if (BODY == 256) { R4(R8(R8(asm volatile(A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));))) }
      if (BODY == 1024) { R4(R4(R8(R8(asm volatile(A8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)))) }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && clk) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int BODY, int KIND>
void run(int wps) {
  const int groups = 1024 * 16;  // 131072 instructions per wave
  const int blocks = 256 * wps;
  float* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 4 * 8);
  k<BODY, KIND><<<blocks, 256>>>(out, BODY, nullptr);
  hipDeviceSynchronize();
  k<BODY, KIND><<<blocks, 256>>>(out, groups, clk);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)blocks * 4);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
  const double instr = (double)groups * 8;
  printf("%-10s body %6d B  waves/SIMD=%d : %.2f clk per instruction per SIMD (%.2f per wave)\n", KIND ? "v_add_f32" : "v_pk_fma", BODY * (KIND ? 32 : 64), wps,
         avg / (instr * wps), avg / instr);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int wps : {1, 2, 4}) {
    run<8, 0>(wps); run<64, 0>(wps); run<256, 0>(wps); run<1024, 0>(wps);
    run<8, 1>(wps); run<64, 1>(wps); run<256, 1>(wps); run<1024, 1>(wps);
  }
  return 0;
}
