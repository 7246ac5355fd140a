// Development experiment, not part of the shipped library: included by uglad_amd/csrc/glad_device.h only under -DUGLAD_BF16X3
// (EXTRA="-DUGLAD_BF16X3=1" bash scripts/dev_build.sh bf16).  Result: profiles/r03_bf16x3_experiment.txt.
#pragma once
// EXPERIMENT (-DUGLAD_BF16X3, never in the shipped library): BASELINE config 3's "bf16 x 3" contraction.  Every fp32 operand is split
// into three bf16 terms (8 + 8 + 8 mantissa bits) on the fly and the product is six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, hl, lh, mm)
// instead of eight v_mfma_f32_32x32x2_f32 per 16 values of k.  Measured: profiles/r03_bf16x3_experiment.txt.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_bf16x3(float x, short& h, short& m, short& l) {
  auto rne = [](float v) -> unsigned {
    unsigned u = __float_as_uint(v);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  };
  const unsigned uh = rne(x);
  const float r1 = x - __uint_as_float(uh << 16);
  const unsigned um = rne(r1);
  const float r2 = r1 - __uint_as_float(um << 16);
  h = (short)uh;
  m = (short)um;
  l = (short)rne(r2);
}
__device__ __forceinline__ void mfma_tile_bf16x3(const float* __restrict__ Ap, int a_si, int a_sk, const float* __restrict__ Bp, int b_sk,
                                                 int b_sj, int K, f32x16& acc) {
  const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
  const float* a = Ap + li * a_si + 8 * kh * a_sk;  // lane l: A[row li][k = 8 kh + j], B[k = 8 kh + j][col li]
  const float* b = Bp + li * b_sj + 8 * kh * b_sk;
  for (int k = 0; k < K; k += 16) {
    s16x8 ah, am, al, bh, bm, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      short h, m, l;
      split_bf16x3(a[(k + j) * a_sk], h, m, l);
      ah[j] = h; am[j] = m; al[j] = l;
      split_bf16x3(b[(k + j) * b_sk], h, m, l);
      bh[j] = h; bm[j] = m; bl[j] = l;
    }
    auto bf = [](const s16x8& v) { return __builtin_bit_cast(bf16x8, v); };
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(am), bf(bm), acc, 0, 0, 0);  // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(ah), bf(bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(al), bf(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(ah), bf(bm), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(am), bf(bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf(ah), bf(bh), acc, 0, 0, 0);
  }
}
