// Micro-benchmark: LDS instruction throughput on gfx950 by instruction form (the forms hipcc picks for the
// fft512 kernels: ds_read_b64 vs the merged ds_read2_b64, ds_read2_b32, b128; b32/b64/b128 writes).
// Build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // byte address: conflict-free consecutive 8-byte (or 16-byte) slots per lane, a private 2 KiB window per wave
  const unsigned a8 = (unsigned)(wv * 4096 + lane * 8);
  const unsigned a16 = (unsigned)(wv * 4096 + lane * 16);
  const unsigned a4 = (unsigned)(wv * 4096 + lane * 4);
  v2 r0 = {0, 0}, r1 = {0, 0}, r2 = {0, 0}, r3 = {0, 0};
  v4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
  float s0 = 0, s1 = 0;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 4 x ds_read_b64
      REP4(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n s_waitcnt lgkmcnt(0)"
                        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a8));)
    } else if (MODE == 1) {  // 2 x ds_read2_b64 (same bytes as MODE 0)
      REP4(asm volatile("ds_read2_b64 %0, %2 offset1:64\n ds_read2_b64 %1, %2 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)"
                        : "=v"(q0), "=v"(q1) : "v"(a8));)
    } else if (MODE == 2) {  // 2 x ds_read_b128
      REP4(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(a16));)
    } else if (MODE == 3) {  // 4 x ds_read2_b32 at odd dword offsets (the "unaligned pair" read)
      REP4(asm volatile("ds_read2_b32 %0, %4 offset0:1 offset1:2\n ds_read2_b32 %1, %4 offset0:129 offset1:130\n ds_read2_b32 %2, %4 offset0:65 offset1:66\n ds_read2_b32 %3, %4 offset0:193 offset1:194\n s_waitcnt lgkmcnt(0)"
                        : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a8));)
    } else if (MODE == 4) {  // 4 x ds_read_b32
      REP4(asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:256\n ds_read_b32 %0, %2 offset:512\n ds_read_b32 %1, %2 offset:768\n s_waitcnt lgkmcnt(0)"
                        : "=v"(s0), "=v"(s1) : "v"(a4));)
    } else if (MODE == 5) {  // 4 x ds_write_b32
      REP4(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:256\n ds_write_b32 %0, %1 offset:512\n ds_write_b32 %0, %1 offset:768\n s_waitcnt lgkmcnt(0)" ::"v"(a4), "v"(s0));)
    } else if (MODE == 6) {  // 4 x ds_write_b64
      REP4(asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n s_waitcnt lgkmcnt(0)" ::"v"(a8), "v"(r0));)
    } else if (MODE == 7) {  // 2 x ds_write2_b64
      REP4(asm volatile("ds_write2_b64 %0, %1, %1 offset1:64\n ds_write2_b64 %0, %1, %1 offset0:128 offset1:192\n s_waitcnt lgkmcnt(0)" ::"v"(a8), "v"(r0));)
    } else if (MODE == 8) {  // 2 x ds_write_b128
      REP4(asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)" ::"v"(a16), "v"(q0));)
    } else if (MODE == 9) {  // 4 x ds_bpermute_b32
      REP4(asm volatile("ds_bpermute_b32 %0, %2, %3\n ds_bpermute_b32 %1, %2, %3\n ds_bpermute_b32 %0, %2, %3\n ds_bpermute_b32 %1, %2, %3\n s_waitcnt lgkmcnt(0)"
                        : "=v"(s0), "=v"(s1) : "v"(a4), "v"((float)lane));)
    } else if (MODE >= 12 && MODE <= 14) {
      // 2 x ds_read_b128 with the A-operand pattern of the 4x4x1 mel phase: lane = 4 slot + frame, frame rows 272 dwords apart
      // (16 bank quads), slot start = 4 s dwords with s = slot % 4 (MODE 12: the four slots of a 16-lane group cover all 64 banks),
      // (slot % 4) / 2 (MODE 13: two slots per 16 banks), 0 (MODE 14: all four slots on the same 16 banks)
      const int slot = lane >> 2, fr = lane & 3;
      const int sv = MODE == 12 ? (slot & 3) : (MODE == 13 ? ((slot & 3) >> 1) : 0);
      const unsigned ap = (unsigned)(wv * 8192 + (fr * 272 + 4 * sv + 16 * (slot & 3) + 64 * (slot >> 2)) * 4);  // distinct addresses in every lane
      REP4(asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:16\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1) : "v"(ap));)
    } else if (MODE == 10) {  // 16 x ds_read_b64, one wait (burst as the kernels issue them)
      asm volatile(
          "ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
          "ds_read_b64 %0, %4 offset:8\n ds_read_b64 %1, %4 offset:520\n ds_read_b64 %2, %4 offset:1032\n ds_read_b64 %3, %4 offset:1544\n"
          "ds_read_b64 %0, %4 offset:16\n ds_read_b64 %1, %4 offset:528\n ds_read_b64 %2, %4 offset:1040\n ds_read_b64 %3, %4 offset:1552\n"
          "ds_read_b64 %0, %4 offset:24\n ds_read_b64 %1, %4 offset:536\n ds_read_b64 %2, %4 offset:1048\n ds_read_b64 %3, %4 offset:1560\n s_waitcnt lgkmcnt(0)"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(a8));
    } else if (MODE == 11) {  // 8 x ds_read2_b64, one wait (same bytes as MODE 10)
      asm volatile(
          "ds_read2_b64 %0, %2 offset1:64\n ds_read2_b64 %1, %2 offset0:128 offset1:192\n"
          "ds_read2_b64 %0, %2 offset0:1 offset1:65\n ds_read2_b64 %1, %2 offset0:129 offset1:193\n"
          "ds_read2_b64 %0, %2 offset0:2 offset1:66\n ds_read2_b64 %1, %2 offset0:130 offset1:194\n"
          "ds_read2_b64 %0, %2 offset0:3 offset1:67\n ds_read2_b64 %1, %2 offset0:131 offset1:195\n s_waitcnt lgkmcnt(0)"
          : "=v"(q0), "=v"(q1) : "v"(a8));
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && clk) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0.x + r1.y + r2.x + r3.y + q0.x + q1.w + s0 + s1;
}
template <int MODE>
void run(int wg_per_cu, const char* name, double bytes_per_wave_iter, int instr_per_iter) {
  const int iters = 2000;
  const int blocks = 256 * wg_per_cu;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  unsigned long long* clk;
  hipMalloc(&clk, (size_t)blocks * 4 * 8);
  const size_t lds = 32768;
  k<MODE><<<blocks, 256, lds>>>(out, 10, nullptr);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipEventRecord(a);
  k<MODE><<<blocks, 256, lds>>>(out, iters, clk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> hc((size_t)blocks * 4);
  hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : hc) avg += (double)v;
  avg /= hc.size();
  const double waves_per_cu = 4.0 * wg_per_cu;
  const double bytes_per_cu = bytes_per_wave_iter * iters * waves_per_cu;
  const double instr_per_cu = (double)instr_per_iter * iters * waves_per_cu;
  printf("%-28s waves/CU=%2.0f  %.3f ms  %.1f B/clk/CU  %.2f clk/instr/CU  (wave lifetime %.0f clk)\n", name, waves_per_cu, ms, bytes_per_cu / avg,
         avg / instr_per_cu, avg);
  hipFree(clk);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>(w, "ds_read_b64 x4 /wait", 4 * 4 * 512.0, 16);
    run<1>(w, "ds_read2_b64 x2 /wait", 4 * 4 * 512.0, 8);
    run<10>(w, "ds_read_b64 x16 /wait", 16 * 512.0, 16);
    run<11>(w, "ds_read2_b64 x8 /wait", 16 * 512.0, 8);
    run<2>(w, "ds_read_b128 x2 /wait", 4 * 2 * 1024.0, 8);
    run<3>(w, "ds_read2_b32(odd) x4 /wait", 4 * 4 * 512.0, 16);
    run<4>(w, "ds_read_b32 x4 /wait", 4 * 4 * 256.0, 16);
    run<5>(w, "ds_write_b32 x4 /wait", 4 * 4 * 256.0, 16);
    run<6>(w, "ds_write_b64 x4 /wait", 4 * 4 * 512.0, 16);
    run<7>(w, "ds_write2_b64 x2 /wait", 4 * 4 * 512.0, 8);
    run<8>(w, "ds_write_b128 x2 /wait", 4 * 2 * 1024.0, 8);
    run<9>(w, "ds_bpermute_b32 x4 /wait", 4 * 4 * 256.0, 16);
    run<12>(w, "b128 mel-A, 4 slots x 16 banks", 4 * 2 * 1024.0, 8);
    run<13>(w, "b128 mel-A, 2 slots share", 4 * 2 * 1024.0, 8);
    run<14>(w, "b128 mel-A, 4 slots share", 4 * 2 * 1024.0, 8);
  }
  return 0;
}
