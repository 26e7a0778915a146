// A 32 x 32 x 16 fp32 product on the bf16 matrix cores: six terms of an exact three-way split of BOTH operands, made on the fly from fp32
// fragments in LDS — the drop-in for eight consecutive v_mfma_f32_32x32x2f32 of the general kernels (k_rows_gemm, k_ffn_fused, k_dw_gemm), whose
// operands are not prepared planes (gathered rows, sums of rows, transposed weights of the backward, any width).
//
// Why: round 5 blamed v_mfma_f32_32x32x2f32 for wrong results beside another queue's bf16 matrix kernel and asked for it to leave every default path;
// round 6 did that with this header — and found the instruction innocent (the site was an LDS read consumed too early in the LayerNorm-on-load branch:
// profiles/r06_overlap_hazard.log).  The six-term form stays the default because it is FASTER: 6 x 8-pass instead of 8 x 16-pass matrix instructions
// per 16-step (config 4 -4 %, forward + backward of a GNCore 15.9 -> 13.1 ms); the fp32 forms run where a call's flags ask (GNX_FLAG_FP32_MFMA).
//
// Arithmetic (as gnx_ffn_x6.hip / gnx_edge_x6.hip): x = h + m + l exactly, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (8 + 8 + 8 mantissa
// bits, round to nearest each); a·b ~ hh + hm + mh + hl + lh + mm, the three dropped terms are <= 2^-24 |a b| together; fp32 accumulation in the
// matrix pipe, small terms first.  Measured against float64 the error is at or below the fp32 instruction's (tests/test_gpu_x6_stress.py).
//
// Layout: lane (c = lane & 31, hi = lane >> 5) holds, for BOTH operands, the eight reduction indices 8 hi .. 8 hi + 7 of the 16-step (any
// bijection works as long as A and B agree); A: row c of the 32-row block, B: column c of the 32-column block; the 32 x 32 accumulator has the
// C/D layout of every 32 x 32 matrix instruction (row (q & 3) + 8 (q >> 2) + 4 hi, column c) — exactly what the fp32 form leaves.
#pragma once
#include <hip/hip_runtime.h>

namespace gnx {

typedef float x6_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 x6_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned x6_u32x4 __attribute__((ext_vector_type(4)));

struct X6Frag {
  x6_bf16x8 h, m, l;
};

__device__ __forceinline__ unsigned x6_cvt2(float x0, float x1) {
  typedef float f2_ __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_{x0, x1}, b2_));
}

// eight fp32 values (reduction order) -> their three bf16 parts
__device__ __forceinline__ X6Frag x6_split8(const float (&v)[8]) {
  x6_u32x4 h, m, l;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = v[2 * p], x1 = v[2 * p + 1];
    h[p] = x6_cvt2(x0, x1);
    const float r0 = x0 - __uint_as_float(h[p] << 16), r1 = x1 - __uint_as_float(h[p] & 0xffff0000u);
    m[p] = x6_cvt2(r0, r1);
    l[p] = x6_cvt2(r0 - __uint_as_float(m[p] << 16), r1 - __uint_as_float(m[p] & 0xffff0000u));
  }
  X6Frag f;
  f.h = __builtin_bit_cast(x6_bf16x8, h);
  f.m = __builtin_bit_cast(x6_bf16x8, m);
  f.l = __builtin_bit_cast(x6_bf16x8, l);
  return f;
}

// eight fp32 values `stride` floats apart, starting at p (LDS): p[0], p[stride], ...
__device__ __forceinline__ X6Frag x6_frag(const float* p, int stride) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p[j * stride];
  return x6_split8(v);
}

// acc += A(32 x 16) * B(16 x 32), six bf16 terms, small terms first
__device__ __forceinline__ x6_f32x16 x6_mma(const X6Frag& a, const X6Frag& b, x6_f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.m, b.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, acc, 0, 0, 0);
  return acc;
}

}  // namespace gnx
