// Shared device primitives of the fft512 kernels: packed complex arithmetic, row-of-16 DPP moves,
// the in-register 16-point FFT, fast log.
#pragma once
#include "common.hpp"

namespace hipfeat {

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- DPP helpers (row = 16 lanes) -------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// lane l >= 1 of a row receives lane l-1's `v`; lane 0 (no source lane, bound_ctrl off) keeps `keep`.
// One DPP move instead of a DPP move plus a v_cndmask: on gfx950 the VOP2 form of v_cndmask_b32
// (implicit VCC), which hipcc selects for lane-predicated selects, issues ~4x slower than any other VALU op
// (tools/ubench/valu_rate.hip).
__device__ __forceinline__ float dpp_shr1_keep(float keep, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false));
}
constexpr int DPP_QUAD(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_ROR1 = 0x121;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_ROW_HALF_MIRROR = 0x141;

__device__ __forceinline__ float row16_sum(float v) {  // every lane of the row gets the same total
  v += dpp_mov<DPP_QUAD(1, 0, 3, 2)>(v);
  v += dpp_mov<DPP_QUAD(2, 3, 0, 1)>(v);
  v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov<DPP_ROW_MIRROR>(v);
  return v;
}
// lane l <- lane (16 - l) % 16 of the same row
__device__ __forceinline__ float row16_negate_index(float v) { return dpp_mov<DPP_ROW_ROR1>(dpp_mov<DPP_ROW_MIRROR>(v)); }

// ---- packed complex arithmetic ---------------------------------------------------------------
// hipcc folds whole-register swaps / broadcasts of packed f32 operands into op_sel modifiers but not
// a sign flip of ONE half, so conjugation and multiplication by +-i are written as a packed multiply
// by the constant (1,-1) / (-1,1) (exact) feeding an op_sel-swapped add.
#define HF_CJ (v2{1.f, -1.f})
#define HF_NCJ (v2{-1.f, 1.f})
__device__ __forceinline__ v2 swap2(v2 a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ v2 rot_mi(v2 a) { return swap2(a * HF_NCJ); }  // a * (-i) = (a.y, -a.x)
__device__ __forceinline__ v2 cmulc(v2 a, v2 w, v2 wp) {  // a * w with wp = (-w.y, w.x) precomputed
  return v2{a.x, a.x} * w + v2{a.y, a.y} * wp;
}
__device__ __forceinline__ v2 cmul(v2 a, v2 w) {  // a * w, only w itself available: a*w.x + (i a)*w.y
  return a * v2{w.x, w.x} + swap2(a * HF_CJ) * v2{w.y, w.y};
}

// a * w in two packed instructions from w alone: (a.x, a.y) * w.x, then (-a.y, a.x) * w.y added -- the half-negation
// rides on the neg_lo modifier of VOP3P, which hipcc does not form from source-level negations.
__device__ __forceinline__ v2 cmul2(v2 a, v2 w) {
  v2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}

// 16-point complex FFT in registers (radix-4 x radix-4, natural order in and out)
__device__ __forceinline__ void fft16(const v2 (&x)[16], v2 (&X)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  v2 y[16];  // y[4*m + n]
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const v2 s0 = x[n] + x[n + 8], s1 = x[n] - x[n + 8];
    const v2 s2 = x[n + 4] + x[n + 12], u3 = swap2(x[n + 4] - x[n + 12]);
    y[n] = s0 + s2;
    y[8 + n] = s0 - s2;
    y[4 + n] = u3 * HF_CJ + s1;    // s1 + (-i) t : one packed fma, the swap folds into op_sel
    y[12 + n] = u3 * HF_NCJ + s1;  // s1 - (-i) t
  }
  // twiddles W16^(n*m): w = (c, -s), wp = (s, c)
  y[4 + 1] = cmulc(y[4 + 1], v2{C1, -S1}, v2{S1, C1});      // n=1 m=1: W^1
  y[8 + 1] = cmulc(y[8 + 1], v2{R2, -R2}, v2{R2, R2});      // n=1 m=2: W^2
  y[12 + 1] = cmulc(y[12 + 1], v2{S1, -C1}, v2{C1, S1});    // n=1 m=3: W^3
  y[4 + 2] = cmulc(y[4 + 2], v2{R2, -R2}, v2{R2, R2});      // n=2 m=1: W^2
  y[8 + 2] = rot_mi(y[8 + 2]);                              // n=2 m=2: W^4 = -i
  y[12 + 2] = cmulc(y[12 + 2], v2{-R2, -R2}, v2{R2, -R2});  // n=2 m=3: W^6
  y[4 + 3] = cmulc(y[4 + 3], v2{S1, -C1}, v2{C1, S1});      // n=3 m=1: W^3
  y[8 + 3] = cmulc(y[8 + 3], v2{-R2, -R2}, v2{R2, -R2});    // n=3 m=2: W^6
  y[12 + 3] = cmulc(y[12 + 3], v2{-C1, S1}, v2{-S1, -C1});  // n=3 m=3: W^9
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const v2 s0 = y[4 * m] + y[4 * m + 2], s1 = y[4 * m] - y[4 * m + 2];
    const v2 s2 = y[4 * m + 1] + y[4 * m + 3], u3 = swap2(y[4 * m + 1] - y[4 * m + 3]);
    X[m] = s0 + s2;                // k' = 0
    X[m + 4] = u3 * HF_CJ + s1;    // k' = 1
    X[m + 8] = s0 - s2;            // k' = 2
    X[m + 12] = u3 * HF_NCJ + s1;  // k' = 3
  }
}

// 32-point complex FFT in registers: one radix-2 decimation-in-frequency stage, then two 16-point FFTs
//   X[2 k]     = FFT16(x[n] + x[n + 16])[k]
//   X[2 k + 1] = FFT16((x[n] - x[n + 16]) W_32^n)[k]
// Rows n >= NZ are structurally zero (zero padding of the frame): their adds / subtracts are not issued at all.
template <int NZ>
__device__ __forceinline__ void fft32(const v2 (&x)[32], v2 (&X)[32]) {
  constexpr float C[16] = {1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f, 0.55557023301960222f,
                           0.38268343236508977f, 0.19509032201612827f, 0.0f, -0.19509032201612827f, -0.38268343236508977f, -0.55557023301960222f,
                           -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323044f};
  constexpr float S[16] = {0.0f, 0.19509032201612827f, 0.38268343236508977f, 0.55557023301960222f, 0.70710678118654752f, 0.83146961230254524f,
                           0.92387953251128674f, 0.98078528040323044f, 1.0f, 0.98078528040323044f, 0.92387953251128674f, 0.83146961230254524f,
                           0.70710678118654752f, 0.55557023301960222f, 0.38268343236508977f, 0.19509032201612827f};
  v2 e[16], o[16], E[16], O[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const bool hi = n + 16 < NZ;  // x[n + 16] may be non-zero
    e[n] = hi ? x[n] + x[n + 16] : x[n];
    const v2 d = hi ? x[n] - x[n + 16] : x[n];
    // W_32^n = (cos, -sin): w = (c, -s), wp = (s, c)
    if (n == 0) o[n] = d;
    else if (n == 8) o[n] = rot_mi(d);
    else o[n] = cmulc(d, v2{C[n], -S[n]}, v2{S[n], C[n]});
  }
  fft16(e, E);
  fft16(o, O);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    X[2 * k] = E[k];
    X[2 * k + 1] = O[k];
  }
}

// 8-point complex FFT in registers (radix-4 x radix-2, natural order in and out)
__device__ __forceinline__ void fft8(const v2* x, v2* X) {
  constexpr float R2 = 0.70710678118654752f;
  v2 e[4], o[4];
  {
    const v2 s0 = x[0] + x[4], s1 = x[0] - x[4], s2 = x[2] + x[6], u3 = swap2(x[2] - x[6]);
    e[0] = s0 + s2, e[2] = s0 - s2, e[1] = u3 * HF_CJ + s1, e[3] = u3 * HF_NCJ + s1;
  }
  {
    const v2 s0 = x[1] + x[5], s1 = x[1] - x[5], s2 = x[3] + x[7], u3 = swap2(x[3] - x[7]);
    o[0] = s0 + s2, o[2] = s0 - s2, o[1] = u3 * HF_CJ + s1, o[3] = u3 * HF_NCJ + s1;
  }
  o[1] = cmulc(o[1], v2{R2, -R2}, v2{R2, R2});    // W8^1
  o[2] = rot_mi(o[2]);                            // W8^2 = -i
  o[3] = cmulc(o[3], v2{-R2, -R2}, v2{R2, -R2});  // W8^3
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    X[k] = e[k] + o[k];
    X[k + 4] = e[k] - o[k];
  }
}

// Epilogue of a 4 x 4 x 1 filterbank accumulator (lane = 4 slot + filter, register = frame): the pieces of a filter group that the
// schedule split over adjacent slots are summed across the DPP row (v += m4 * v[lane - 4], then v += m8 * v[lane - 8], zeros shifted in)
// and the result is floored.  The row_shr adds are v_fmac_f32 with a DPP source operand (hipcc emits a v_mov_b32_dpp plus a v_fma for the
// intrinsic form), the floor is a bare v_max (no canonicalising v_max(x, x) in front); `s_nop 7` covers the matrix-core -> VALU read
// hazard and the VALU-write -> DPP-read hazard of whatever precedes the block, which hipcc cannot see through inline asm; inside the
// block a register is read through DPP no sooner than three instructions after it was written.
__device__ __forceinline__ void mel4_reduce_floor(const f32x4 acc, float m4, float m8, float floor_, float (&v)[4]) {
  float v0 = acc[0], v1 = acc[1], v2_ = acc[2], v3 = acc[3];
  asm volatile(
      "s_nop 7\n\t"
      "v_fmac_f32_dpp %0, %0, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %2, %2, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %3, %3, %4 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %0, %0, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %1, %1, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %2, %2, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_fmac_f32_dpp %3, %3, %5 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_max_f32 %0, %0, %6\n\tv_max_f32 %1, %1, %6\n\tv_max_f32 %2, %2, %6\n\tv_max_f32 %3, %3, %6"
      : "+v"(v0), "+v"(v1), "+v"(v2_), "+v"(v3)
      : "v"(m4), "v"(m8), "v"(floor_));
  v[0] = v0, v[1] = v1, v[2] = v2_, v[3] = v3;
}
// the rows of one accumulator set under ONE lane mask (the caller tests `col < M` once); nf < ROWS only in the last round of a cut
template <int ROWS>
__device__ __forceinline__ void mel4_store(float* o, int64_t stride, int nf, const float* v) {
  if (nf >= ROWS) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) o[i * stride] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < ROWS - 1; ++i)
      if (i < nf) o[i * stride] = v[i];
  }
}
// the same with the row base wave-uniform and the column as a 32-bit lane offset: global_store_dword voffset, vdata, s[base:base+1]
template <int ROWS>
__device__ __forceinline__ void mel4_store_saddr(float* row0, unsigned col, int64_t stride, int nf, const float* v) {
  const unsigned off = col * 4u;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    if (i < nf) {  // nf is wave-uniform
      const float* base = row0 + i * stride;
      asm volatile("global_store_dword %0, %1, %2" : : "v"(off), "v"(v[i]), "s"(base) : "memory");
    }
  }
}

// natural log of a normal positive float: v_log_f32 (log2, 1 ulp) times ln 2.  The argument is
// >= mel_floor (1.19e-7), so the denormal path of the library logf is never needed.
__device__ __forceinline__ float fast_log(float x) {
#ifdef HIPFEAT_ACCURATE_LOG
  return logf(x);
#else
  return __builtin_amdgcn_logf(x) * 0.69314718055994531f;
#endif
}

}  // namespace hipfeat
