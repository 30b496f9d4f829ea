// Micro-benchmark: issue rate of scalar vs packed f32 VALU instructions on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* clk) {
  const unsigned long long c0 = __builtin_readcyclecounter();
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0001f;
  const v2 c2 = {1.0001f, 0.9999f};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // v_fma_f32, 8 independent chains
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                        "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 1) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
    } else if (MODE == 2) {  // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
    } else if (MODE == 3) {  // v_add_f32
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
    } else if (MODE == 4) {  // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
    } else if (MODE == 5) {  // v_mov_b32 dpp row_mirror (independent)
      REP8(asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %5 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_mirror row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %7 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (MODE == 6) {  // v_pk_add_f32 with op_sel swap + neg
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                        "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                        "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                        "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
    } else if (MODE == 7) {  // v_cndmask_b32
      REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");)
    }
    else if (MODE == 8) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "s20","s21");) }
    else if (MODE == 9) { REP8(asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 10) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 11) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 12) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 13) { REP8(asm volatile("v_add_f32_dpp %0, %0, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 14) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 15) { REP8(asm volatile("v_pk_fma_f32 %0, %8, %8, %0\n v_pk_fma_f32 %1, %8, %8, %1\n v_pk_fma_f32 %2, %8, %8, %2\n v_pk_fma_f32 %3, %8, %8, %3\n v_pk_fma_f32 %4, %8, %8, %4\n v_pk_fma_f32 %5, %8, %8, %5\n v_pk_fma_f32 %6, %8, %8, %6\n v_pk_fma_f32 %7, %8, %8, %7\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));) }
    else if (MODE == 16) { REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
    else if (MODE == 17) { REP8(asm volatile("v_fmaak_f32 %0, %0, %8, 0x3f800000\n v_fmaak_f32 %1, %1, %8, 0x3f800000\n v_fmaak_f32 %2, %2, %8, 0x3f800000\n v_fmaak_f32 %3, %3, %8, 0x3f800000\n v_fmaak_f32 %4, %4, %8, 0x3f800000\n v_fmaak_f32 %5, %5, %8, 0x3f800000\n v_fmaak_f32 %6, %6, %8, 0x3f800000\n v_fmaak_f32 %7, %7, %8, 0x3f800000\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && clk) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  if (MODE == 18 || MODE == 19) {
    asm volatile("v_cmp_gt_u32 vcc, 32, %0" :: "v"(threadIdx.x & 63) : "vcc");
    for (int i = 0; i < iters; ++i) {
      if (MODE == 18) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                        "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
      else { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                        "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
double run(int waves_per_simd, const char* name, double flops_per_lane_instr) {
  const int iters = 2000;
  const int blocks = 256 * waves_per_simd;  // 256-thread blocks = 1 wave per SIMD each
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  unsigned long long* clk; hipMalloc(&clk, (size_t)blocks * 4 * 8);
  k<MODE><<<blocks, 256>>>(out, 10, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<MODE><<<blocks, 256>>>(out, iters, clk);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_wave = (double)iters * 64;
  const double per_simd = instr_per_wave * waves_per_simd;  // instr issued per SIMD
  const double ns_per_instr = ms * 1e6 / per_simd;
  const double tflops = flops_per_lane_instr * 64 * instr_per_wave * (double)blocks * 4 / (ms * 1e-3) / 1e12;
  std::vector<unsigned long long> hc((size_t)blocks * 4);
  hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : hc) avg += (double)v; avg /= hc.size();
  const double clk_per_instr_simd = avg / per_simd;  // wave lifetime / instructions issued by all waves of the SIMD
  hipFree(clk);
  printf("[%.2f clk/instr/SIMD by s_memtime, eff clock %.2f GHz] ", clk_per_instr_simd, avg / (ms * 1e6));
  printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns/instr/SIMD (= %.2f clk @2.4GHz)  %.1f TFLOP/s\n", name, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, tflops);
  hipFree(out);
  return ns_per_instr;
}
int main() {
  for (int w : {1, 3}) {
    run<0>(w, "v_fma_f32", 2);
    run<1>(w, "v_pk_fma_f32", 4);
    run<2>(w, "v_pk_add_f32", 2);
    run<3>(w, "v_add_f32", 1);
    run<4>(w, "v_pk_mul_f32", 2);
    run<5>(w, "v_mov_b32_dpp", 0);
    run<6>(w, "v_pk_add_f32 op_sel", 2);
    run<7>(w, "v_cndmask_b32", 0);
    run<18>(w, "v_cndmask_b32 e32 vcc(def)", 0);
    run<19>(w, "v_cndmask_b32_e64 vcc", 0);
    run<8>(w, "v_cndmask_b32_e64 sgpr", 0);
    run<9>(w, "v_fmac_f32", 2);
    run<10>(w, "v_mul_f32", 1);
    run<11>(w, "v_xor_b32", 0);
    run<12>(w, "v_mov_b32", 0);
    run<13>(w, "v_add_f32_dpp ror1", 1);
    run<14>(w, "v_fma_f32 (2 src regs)", 2);
    run<15>(w, "v_pk_fma_f32 (acc form)", 4);
    run<16>(w, "v_sub_f32", 1);
    run<17>(w, "v_mad? v_fmaak", 2);
  }
  return 0;
}
