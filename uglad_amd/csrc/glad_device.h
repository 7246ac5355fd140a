// Device-side building blocks of the GLAD cell for gfx950 (wave64, f32 MFMA, LDS-resident matrices).
// One workgroup of 512 threads (8 waves, two per SIMD) owns one D x D matrix; DP = D rounded up to 32.
// Matrices live in LDS with row stride LD = DP + 1 so that row-wise, column-wise and MFMA-operand accesses
// (ds_read_b32, 32 banks) are all conflict-free.
#pragma once
#include <hip/hip_runtime.h>

namespace uglad {

#ifndef UGLAD_THREADS
#define UGLAD_THREADS 512
#endif
constexpr int kThreads = UGLAD_THREADS;  // 512 = two waves per SIMD (measured 17 % faster end to end than 256 = one)
constexpr int kWaves = kThreads / 64;
constexpr int kNsIters = 10;  // reference: torch_sqrtm.py:14,33

// ---- parameter vector offsets (include/uglad_hip.h)
constexpr int P_T = 0, P_RW1 = 1, P_RB1 = 10, P_RW2 = 13, P_RB2 = 22, P_RW3 = 25, P_RB3 = 28;
constexpr int P_LW1 = 29, P_LB1 = 35, P_LW2 = 38, P_LB2 = 41;
constexpr int kNParam = 42, kNRho = 28;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Ordering point between two phases that ONE wave runs on LDS data of its own (no other wave touches it): the DS unit
// processes the instructions of a wave in issue order, so on the hardware only the compiler has to be kept from moving
// accesses across; the emulator runs the lanes of a wave as separate fibers and needs a real rendezvous.
#ifdef UGLAD_SIMT_EMUL
#define UGLAD_WAVE_SYNC() simt::wave_sync_point()
#else
#define UGLAD_WAVE_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")  // (the data may sit in LDS or in global memory)
#endif

// ------------------------------------------------------------------------------------------ reductions
// Cross-lane moves through the DPP path of the vector ALU (a few cycles each) instead of ds_bpermute (an LDS round trip per
// step): quad_perm / row_ror inside a row of 16 lanes, row_bcast:15/31 across the four rows of the wave.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}

// value of the neighbouring lane (lane ^ 1)
__device__ __forceinline__ float lane_xor1(float v) { return dpp_move<0xb1>(v); }  // quad_perm:[1,0,3,2]

// v + (the value of lane ^ 32), in every lane: v_permlane32_swap exchanges the upper half of its first operand with the lower half of
// its second (gfx950), no LDS round trip.
__device__ __forceinline__ float sum_halves(float v) {
#ifdef UGLAD_SIMT_EMUL
  return v + __shfl_xor(v, 32);
#else
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

// the same value, but opaque to the optimiser from here on: stops loop-invariant code motion from hoisting everything derived from
// it (per-thread entry lists, address tables) out of a long loop and keeping it live in registers across the whole body
__device__ __forceinline__ int opaque_v(int v) {
#ifndef UGLAD_SIMT_EMUL
  asm volatile("" : "+v"(v));
#endif
  return v;
}

// The lane index from the hardware (two instructions, no input register), opaque to the optimiser: where a kernel is held to a register
// budget, a `lane` that stays live across a long loop body costs a register -- or, spilled, a scratch reload per iteration.
__device__ __forceinline__ int lane_now() {
#ifdef UGLAD_SIMT_EMUL
  return threadIdx.x & 63;
#else
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
#endif
}

// value of lane `src` (wave-uniform index) in every lane: v_readlane_b32, no LDS round trip
__device__ __forceinline__ float bcast_lane(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Sum over the 64 lanes, result in every lane (and the same bits on every lane).
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xb1>(v);        // quad_perm:[1,0,3,2]
  v += dpp_move<0x4e>(v);        // quad_perm:[2,3,0,1]
  v += dpp_move<0x124>(v);       // row_ror:4
  v += dpp_move<0x128>(v);       // row_ror:8   -> every lane holds the sum of its row of 16
  v += dpp_move<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Sum over the workgroup, result in every thread.  s_red: >= kWaves floats of LDS.  Two barriers.
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) t += s_red[i];
  __syncthreads();
  return t;
}

// ------------------------------------------------------------------------------------------ rhoNN (glad_params.py:38-49,61-81)
// Activations from one correctly-rounded expf and one division each.  The library tanhf (~150 instructions) was 90 % of the
// entrywise phases; the raw hardware exp (__expf) is fast but its ~3e-7 relative error showed up as a 1 % error in the most
// ill-conditioned gradient (theta_init_offset at D=64), so the accurate expf stays.
// exp(x) for x <= ~0 on the hardware exp2 with the rounding error of x*log2(e) carried along (2^(t+e) = 2^t (1 + e ln 2)),
// and a division as rcp + one Newton step: both within ~1 ulp at a third of the library cost.
__device__ __forceinline__ float exp_acc(float x) {
  const float t = x * 1.44269504f;
  float e = fmaf(x, 1.44269504f, -t);
  e = fmaf(x, 1.925963033e-8f, e);
  const float r = __builtin_amdgcn_exp2f(t);
  return fmaf(r, e * 0.69314718f, r);
}

__device__ __forceinline__ float div_acc(float a, float b) {
  float q = __builtin_amdgcn_rcpf(b);
  q = fmaf(q, fmaf(-b, q, 1.0f), q);
  return a * q;
}

__device__ __forceinline__ float sigmoidf_(float x) {
  // 1/(1+e^-x), evaluated from e^{-|x|} so the exponential never overflows
  const float t = exp_acc(-fabsf(x));
  const float s = div_acc(1.0f, 1.0f + t);
  return (x >= 0.f) ? s : 1.0f - s;
}

__device__ __forceinline__ float tanhf_(float x) {
  const float t = exp_acc(-2.0f * fabsf(x));
  return copysignf(div_acc(1.0f - t, 1.0f + t), x);
}

struct RhoAct {
  float h1[3], h2[3], rho;
};

__device__ __forceinline__ void rho_forward(const float* __restrict__ p, float x1, float x2, float x3, RhoAct& a) {
#pragma unroll
  for (int o = 0; o < 3; ++o)
    a.h1[o] = tanhf_(fmaf(p[P_RW1 + 3 * o], x1, fmaf(p[P_RW1 + 3 * o + 1], x2, fmaf(p[P_RW1 + 3 * o + 2], x3, p[P_RB1 + o]))));
#pragma unroll
  for (int o = 0; o < 3; ++o)
    a.h2[o] = tanhf_(fmaf(p[P_RW2 + 3 * o], a.h1[0], fmaf(p[P_RW2 + 3 * o + 1], a.h1[1], fmaf(p[P_RW2 + 3 * o + 2], a.h1[2], p[P_RB2 + o]))));
  a.rho = sigmoidf_(fmaf(p[P_RW3], a.h2[0], fmaf(p[P_RW3 + 1], a.h2[1], fmaf(p[P_RW3 + 2], a.h2[2], p[P_RB3]))));
}

// Two entries at a time on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: one issue slot for two lanes'
// worth of work; only exp2 and rcp stay scalar).  Same arithmetic, operation by operation, as the scalar versions above.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float a) { return (v2f){a, a}; }
__device__ __forceinline__ v2f exp_acc2(v2f x) {
  const v2f t = x * 1.44269504f;
  v2f e = fma2(x, splat2(1.44269504f), -t);
  e = fma2(x, splat2(1.925963033e-8f), e);
  const v2f r = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  return fma2(r, e * 0.69314718f, r);
}
__device__ __forceinline__ v2f div_acc2(v2f a, v2f b) {
  v2f q = {__builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y)};
  q = fma2(q, fma2(-b, q, splat2(1.0f)), q);
  return a * q;
}
__device__ __forceinline__ v2f tanh2(v2f x) {
  const v2f t = exp_acc2(-2.0f * __builtin_elementwise_abs(x));
  const v2f r = div_acc2(1.0f - t, 1.0f + t);
  return (v2f){copysignf(r.x, x.x), copysignf(r.y, x.y)};
}
__device__ __forceinline__ v2f sigmoid2(v2f x) {
  const v2f t = exp_acc2(-__builtin_elementwise_abs(x));
  const v2f s = div_acc2(splat2(1.0f), 1.0f + t);
  return (v2f){(x.x >= 0.f) ? s.x : 1.0f - s.x, (x.y >= 0.f) ? s.y : 1.0f - s.y};
}
struct RhoAct2 {
  v2f h1[3], h2[3], rho;
  __device__ __forceinline__ RhoAct half(int c) const {
    RhoAct a;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      a.h1[o] = c ? h1[o].y : h1[o].x;
      a.h2[o] = c ? h2[o].y : h2[o].x;
    }
    a.rho = c ? rho.y : rho.x;
    return a;
  }
};
__device__ __forceinline__ void rho_forward2(const float* __restrict__ p, v2f x1, v2f x2, v2f x3, RhoAct2& a) {
#pragma unroll
  for (int o = 0; o < 3; ++o)
    a.h1[o] = tanh2(fma2(splat2(p[P_RW1 + 3 * o]), x1,
                         fma2(splat2(p[P_RW1 + 3 * o + 1]), x2, fma2(splat2(p[P_RW1 + 3 * o + 2]), x3, splat2(p[P_RB1 + o])))));
#pragma unroll
  for (int o = 0; o < 3; ++o)
    a.h2[o] = tanh2(fma2(splat2(p[P_RW2 + 3 * o]), a.h1[0],
                         fma2(splat2(p[P_RW2 + 3 * o + 1]), a.h1[1], fma2(splat2(p[P_RW2 + 3 * o + 2]), a.h1[2], splat2(p[P_RB2 + o])))));
  a.rho = sigmoid2(fma2(splat2(p[P_RW3]), a.h2[0], fma2(splat2(p[P_RW3 + 1]), a.h2[1], fma2(splat2(p[P_RW3 + 2]), a.h2[2], splat2(p[P_RB3])))));
}

// sign(x) * max(0, |x| - rho)   (glad_params.py:81)
__device__ __forceinline__ float soft_threshold(float x, float rho) {
  const float m = fabsf(x) - rho;
  return m > 0.f ? copysignf(m, x) : 0.f;
}

// Backward of rho = rhoNN(x1,x2,x3) for upstream g_rho, weight w (1 or 2: an off-diagonal entry stands for (i,j) and (j,i)).
// Accumulates the 28 parameter gradients into g[0..27] (layout = params[1..28]) and returns d/dx1, d/dx3.
__device__ __forceinline__ void rho_backward(const float* __restrict__ p, float x1, float x2, float x3, const RhoAct& a,
                                             float g_rho, float w, float* g, float& gx1, float& gx3) {
  const float go = g_rho * a.rho * (1.f - a.rho);
  const float gow = go * w;
  float ga2[3], ga1[3];
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    g[(P_RW3 - 1) + h] += gow * a.h2[h];
    ga2[h] = go * p[P_RW3 + h] * (1.f - a.h2[h] * a.h2[h]);
  }
  g[P_RB3 - 1] += gow;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float t = ga2[o] * w;
#pragma unroll
    for (int h = 0; h < 3; ++h) g[(P_RW2 - 1) + 3 * o + h] += t * a.h1[h];
    g[(P_RB2 - 1) + o] += t;
  }
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const float s = ga2[0] * p[P_RW2 + h] + ga2[1] * p[P_RW2 + 3 + h] + ga2[2] * p[P_RW2 + 6 + h];
    ga1[h] = s * (1.f - a.h1[h] * a.h1[h]);
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float t = ga1[o] * w;
    g[(P_RW1 - 1) + 3 * o + 0] += t * x1;
    g[(P_RW1 - 1) + 3 * o + 1] += t * x2;
    g[(P_RW1 - 1) + 3 * o + 2] += t * x3;
    g[(P_RB1 - 1) + o] += t;
  }
  gx1 = ga1[0] * p[P_RW1 + 0] + ga1[1] * p[P_RW1 + 3] + ga1[2] * p[P_RW1 + 6];
  gx3 = ga1[0] * p[P_RW1 + 2] + ga1[1] * p[P_RW1 + 5] + ga1[2] * p[P_RW1 + 8];
}

// rho_backward for two entries at once (packed fp32 pipe); g2[q].x + g2[q].y is what the scalar version accumulates in g[q].
__device__ __forceinline__ void rho_backward2(const float* __restrict__ p, v2f x1, v2f x2, v2f x3, const RhoAct2& a, v2f g_rho,
                                              v2f w, v2f* g2, v2f& gx1, v2f& gx3) {
  const v2f go = g_rho * a.rho * (1.f - a.rho);
  const v2f gow = go * w;
  v2f ga2[3], ga1[3];
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    g2[(P_RW3 - 1) + h] = fma2(gow, a.h2[h], g2[(P_RW3 - 1) + h]);
    ga2[h] = go * p[P_RW3 + h] * (1.f - a.h2[h] * a.h2[h]);
  }
  g2[P_RB3 - 1] += gow;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const v2f t = ga2[o] * w;
#pragma unroll
    for (int h = 0; h < 3; ++h) g2[(P_RW2 - 1) + 3 * o + h] = fma2(t, a.h1[h], g2[(P_RW2 - 1) + 3 * o + h]);
    g2[(P_RB2 - 1) + o] += t;
  }
#pragma unroll
  for (int h = 0; h < 3; ++h) {
    const v2f s = ga2[0] * p[P_RW2 + h] + ga2[1] * p[P_RW2 + 3 + h] + ga2[2] * p[P_RW2 + 6 + h];
    ga1[h] = s * (1.f - a.h1[h] * a.h1[h]);
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const v2f t = ga1[o] * w;
    g2[(P_RW1 - 1) + 3 * o + 0] = fma2(t, x1, g2[(P_RW1 - 1) + 3 * o + 0]);
    g2[(P_RW1 - 1) + 3 * o + 1] = fma2(t, x2, g2[(P_RW1 - 1) + 3 * o + 1]);
    g2[(P_RW1 - 1) + 3 * o + 2] = fma2(t, x3, g2[(P_RW1 - 1) + 3 * o + 2]);
    g2[(P_RB1 - 1) + o] += t;
  }
  gx1 = ga1[0] * p[P_RW1 + 0] + ga1[1] * p[P_RW1 + 3] + ga1[2] * p[P_RW1 + 6];
  gx3 = ga1[0] * p[P_RW1 + 2] + ga1[1] * p[P_RW1 + 5] + ga1[2] * p[P_RW1 + 8];
}

// ------------------------------------------------------------------------------------------ LambdaNN (glad_params.py:51-59,83-95)
__device__ __forceinline__ float lambda_forward(const float* __restrict__ p, float n, float lam_prev) {
  float o = p[P_LB2];
#pragma unroll
  for (int h = 0; h < 3; ++h)
    o = fmaf(p[P_LW2 + h], tanhf(fmaf(p[P_LW1 + 2 * h], n, fmaf(p[P_LW1 + 2 * h + 1], lam_prev, p[P_LB1 + h]))), o);
  return sigmoidf_(o);
}

// ------------------------------------------------------------------------------------------ spectral square root
// r_i = what the chosen evaluation of (b^T b + c I)^{1/2} returns on eigenvalue beta_i of b.  nrmA = ||b^T b + cI||_F
// (only used by NS10): the reference's coupled Newton-Schulz iteration acts on each eigenvalue independently.
__device__ __forceinline__ float sqrt_spectrum(float beta, float c, float nrmA, int mode) {
  const float alpha = fmaf(beta, beta, c);
  if (mode == 0) return sqrtf(alpha);
  float y = alpha / nrmA, z = 1.f;
#pragma unroll
  for (int it = 0; it < kNsIters; ++it) {
    const float T = 0.5f * (3.f - z * y);
    y = y * T;
    z = T * z;
  }
  return y * sqrtf(nrmA);
}

// ------------------------------------------------------------------------------------------ shifted spectral form
// theta_half = f(b) is NOT assembled as U diag(phi(beta)) U^T: with trained parameters b = S/lam - Z is negative definite with
// |beta| ~ 20..300 while r(beta) - |beta| ~ 0.1..1, i.e. phi(beta) = -beta + (small), and the fp32 eigensolver's ~1e-6 ||b|| error in
// (U, beta) lands in full on theta_half -- 1e-6 per step, which the 42 gradients amplify ~100 x (measured: DESIGN.md section 2).
// Instead        theta_half = -alpha b + U diag(psi) U^T,      psi_i = phi(beta_i) + alpha beta_i,
// an identity for every alpha; -alpha b is formed entrywise from S and Z (one rounding), only the remainder goes through the
// eigenvectors, and d psi / d beta = phi' + alpha is ~0.01 where phi' ~ -1.  alpha = clamp(-sum phi beta / sum beta^2, 0, 1) minimises
// ||psi||_2 (phi of the exact square root is good enough for choosing it).  psi is evaluated in fp64 from the fp32 eigenvalues (O(D)
// work): phi + alpha beta cancels to ~1 % of its terms, and the Newton-Schulz recurrence of NS10 is then the reference's spectral
// function to 1e-16, not to 1e-6.
// Call with every thread of the workgroup; thread tid < D passes its eigenvalue.  s3: 3 * D + 80 doubles of LDS scratch.
// Returns psi of this thread's eigenvalue (0 for tid >= D); alpha is the same in every thread, and so is
// cond = cond_2(b^T b + 4/lam I) = (max beta^2 + 4/lam) / (min beta^2 + 4/lam): what the reference's 10 Newton-Schulz steps
// (torch_sqrtm.py:13-29) depend on -- the regime diagnostic of SURVEY.md section 7, hard part 1.
__device__ __forceinline__ float shifted_spectrum(float be, int D, float lam, int mode, double* __restrict__ s3, float& alpha,
                                                  float& cond) {
  const int tid = threadIdx.x;
  const double c4 = 4.0 / (double)lam, b = (double)be;
  const double al = b * b + c4;
  const double r_exact = sqrt(al);
  if (tid < D) {
    s3[tid] = al * al;
    s3[D + tid] = 0.5 * (r_exact - b) * b;
    s3[2 * D + tid] = b * b;
  }
  __syncthreads();
  double* part = s3 + 3 * D;  // the three sums, then max and min of beta^2
  if ((tid >> 6) == 0) {  // wave 0: lane t < 48 sums entries t % 16, t % 16 + 16, ... of array t / 16, then a butterfly over each 16 lanes
    const double* a = s3 + ((tid >> 4) < 3 ? (tid >> 4) : 2) * D;
    double s = 0.0, mx = 0.0, mn = 1e300;
    for (int i = tid & 15; i < D; i += 16) {
      const double v = a[i];
      s += v;
      mx = v > mx ? v : mx;
      mn = v < mn ? v : mn;
    }
#pragma unroll
    for (int o = 1; o < 16; o *= 2) {
      s += __shfl_xor(s, o);
      const double mxo = __shfl_xor(mx, o), mno = __shfl_xor(mn, o);
      mx = mxo > mx ? mxo : mx;
      mn = mno < mn ? mno : mn;
    }
    if ((tid & 15) == 0 && tid < 48) part[tid >> 4] = s;
    if (tid == 32) {  // (the beta^2 array)
      part[3] = mx;
      part[4] = mn;
    }
  }
  __syncthreads();
  // (read as five values: the 80 partial results of an earlier version, fetched by every thread at once, were 160 registers and spills)
  const double n2 = part[0], pb = part[1], bb = part[2], b2max = part[3], b2min = part[4];
  cond = (float)((b2max + c4) / (b2min + c4));
  double a_opt = (bb > 0.0) ? -pb / bb : 0.0;
  a_opt = a_opt < 0.0 ? 0.0 : (a_opt > 1.0 ? 1.0 : a_opt);
  alpha = (float)a_opt;
  double r = r_exact;
  if (mode != 0) {  // NS10: the reference's coupled Newton-Schulz iteration acts on every eigenvalue independently (torch_sqrtm.py:13-29)
    const double nrmA = sqrt(n2);
    double y = al / nrmA, z = 1.0;
#pragma unroll
    for (int it = 0; it < kNsIters; ++it) {
      const double T = 0.5 * (3.0 - z * y);
      y = y * T;
      z = T * z;
    }
    r = y * sqrt(nrmA);
  }
  const double psi = 0.5 * (r - b) + (double)alpha * b;
  __syncthreads();  // (s3 may be reused by the caller)
  return (tid < D) ? (float)psi : 0.f;
}

// ------------------------------------------------------------------------------------------ Jacobi eigensolver
// Two-sided cyclic Jacobi with the round-robin parallel ordering: in each of the n-1 rounds of a sweep the n/2 disjoint
// pairs are rotated at once, A <- J^T A J on 2x2 blocks, V <- V J.  On return diag(A) = eigenvalues, columns of V =
// eigenvectors.  Rows/columns >= D of a padded matrix carry exact zeros off the diagonal and are never rotated.
template <int DP>
__device__ __forceinline__ void pair_of(int r, int k, int& i1, int& i2) {
  constexpr int m = DP - 1;
  if (k == 0) {
    i1 = m;
    i2 = r;
  } else {
    i1 = r + k;
    if (i1 >= m) i1 -= m;
    i2 = r + m - k;
    if (i2 >= m) i2 -= m;
  }
}

// Rotations are applied in Rutishauser's form  x' = x - s (y + tau x),  y' = y + s (x - tau y),  tau = s / (1 + c):
// the correction is O(s), so the many near-identity rotations of the late sweeps add no rounding error of the size of
// ulp(x) -- with the plain  c x - s y  form fp32 eigenvectors lose orthogonality at the 1e-5 level.
template <int DP>
__device__ void jacobi_eig(float* __restrict__ A, float* __restrict__ V, float* s_t, float* s_s, float* s_h, float* s_red,
                           int* s_flag) {
  constexpr int LD = DP + 1, H = DP / 2;
  const int tid = threadIdx.x;
  float fro = 0.f;
  for (int idx = tid; idx < DP * DP; idx += kThreads) {
    const float v = A[(idx / DP) * LD + (idx % DP)];
    fro = fmaf(v, v, fro);
  }
  fro = sqrtf(block_sum(fro, s_red));
  const float thresh = 1e-8f * fro;
  for (int sweep = 0; sweep < 16; ++sweep) {
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    for (int r = 0; r < DP - 1; ++r) {
      if (tid < H) {
        int p, q;
        pair_of<DP>(r, tid, p, q);
        const float apq = A[p * LD + q];
        float t = 0.f, s = 0.f, h = 0.f;
        if (fabsf(apq) > thresh) {
          const float app = A[p * LD + p], aqq = A[q * LD + q];
          const float tau = (aqq - app) / (2.f * apq);
          t = copysignf(1.f, tau) / (fabsf(tau) + sqrtf(fmaf(tau, tau, 1.f)));
          const float c = 1.f / sqrtf(fmaf(t, t, 1.f));
          s = t * c;
          h = s / (1.f + c);
          *s_flag = 1;
        }
        s_t[tid] = t;
        s_s[tid] = s;
        s_h[tid] = h;
      }
      __syncthreads();
      for (int b = tid; b < H * H; b += kThreads) {
        const int kr = b / H, kc = b - kr * H;
        int r1, r2, c1, c2;
        pair_of<DP>(r, kr, r1, r2);
        pair_of<DP>(r, kc, c1, c2);
        const float x11 = A[r1 * LD + c1], x12 = A[r1 * LD + c2], x21 = A[r2 * LD + c1], x22 = A[r2 * LD + c2];
        float z11, z12, z21, z22;
        if (kr == kc) {
          // the rotated 2x2 pivot block: a_pp' = a_pp - t a_pq, a_qq' = a_qq + t a_pq, a_pq' = 0
          const float t = s_t[kr];
          z11 = x11 - t * x12;
          z22 = x22 + t * x12;
          z12 = (t != 0.f) ? 0.f : x12;
          z21 = (t != 0.f) ? 0.f : x21;
        } else {
          const float sr = s_s[kr], hr = s_h[kr], sc = s_s[kc], hc = s_h[kc];
          const float y11 = x11 - sr * (x21 + hr * x11), y21 = x21 + sr * (x11 - hr * x21);
          const float y12 = x12 - sr * (x22 + hr * x12), y22 = x22 + sr * (x12 - hr * x22);
          z11 = y11 - sc * (y12 + hc * y11);
          z12 = y12 + sc * (y11 - hc * y12);
          z21 = y21 - sc * (y22 + hc * y21);
          z22 = y22 + sc * (y21 - hc * y22);
        }
        A[r1 * LD + c1] = z11;
        A[r1 * LD + c2] = z12;
        A[r2 * LD + c1] = z21;
        A[r2 * LD + c2] = z22;
      }
      for (int b = tid; b < DP * H; b += kThreads) {
        const int i = b / H, kc = b - i * H;
        int c1, c2;
        pair_of<DP>(r, kc, c1, c2);
        const float sc = s_s[kc], hc = s_h[kc];
        const float v1 = V[i * LD + c1], v2 = V[i * LD + c2];
        V[i * LD + c1] = v1 - sc * (v2 + hc * v1);
        V[i * LD + c2] = v2 + sc * (v1 - hc * v2);
      }
      __syncthreads();
    }
    if (*s_flag == 0) break;
    __syncthreads();
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------ f32 MFMA GEMM on LDS operands
// C = op(X) * op(Y), all DP x DP, operands in LDS (stride LD), result left in registers as 32x32 tiles.
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the accumulator register
// `reg` of lane l is C[(reg&3) + 8*(reg>>2) + 4*(l>>5)][l&31].  Exact fp32 (a k-ordered fmaf chain).
template <int NT, bool UPPER>
struct Tiles {
  static constexpr int kCount = UPPER ? NT * (NT + 1) / 2 : NT * NT;
  static constexpr int kPerWave = (kCount + kWaves - 1) / kWaves;
  // tile t -> (I, J); for UPPER the enumeration is row-major over I <= J
  __device__ static __forceinline__ void ij(int t, int& I, int& J) {
    if (UPPER) {
      int i = 0, rem = t;
      while (rem >= NT - i) {
        rem -= NT - i;
        ++i;
      }
      I = i;
      J = i + rem;
    } else {
      I = t / NT;
      J = t - I * NT;
    }
  }
};

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// acc += sum_k A[i][k] B[k][j] for one 32x32 tile; A[i][k] at Ap[i*a_si + k*a_sk], B[k][j] at Bp[k*b_sk + j*b_sj]; K a multiple
// of 16.  Operands are fetched 8 k-steps (16 values of k) ahead of the MFMAs that consume them, so the LDS latency of
// the next chunk hides behind the 8 x 64 matrix-pipe cycles of the current one.
__device__ __forceinline__ void mfma_tile(const float* __restrict__ Ap, int a_si, int a_sk, const float* __restrict__ Bp,
                                          int b_sk, int b_sj, int K, f32x16& acc) {
  // (opaque: a caller that loops over several products would otherwise keep every product's lane-derived base address live across its
  // whole loop body -- in cell_bwd_kernel they were the spilled registers)
  const int lane = opaque_v(threadIdx.x) & 63, li = lane & 31, kh = lane >> 5;
  const float* a = Ap + li * a_si + kh * a_sk;
  const float* b = Bp + li * b_sj + kh * b_sk;
  float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    a0[u] = a[2 * u * a_sk];
    b0[u] = b[2 * u * b_sk];
  }
  for (int k = 0; k < K; k += 16) {
    const bool more = k + 16 < K;
    const float* an = a + (more ? 16 * a_sk : 0);
    const float* bn = b + (more ? 16 * b_sk : 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a1[u] = an[2 * u * a_sk];
      b1[u] = bn[2 * u * b_sk];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0[u] = a1[u];
      b0[u] = b1[u];
    }
    a = an;
    b = bn;
  }
}

#if defined(UGLAD_BF16X3) && !defined(UGLAD_SIMT_EMUL)
#include "../../scripts/retired/bf16x3_tile.h"  // development experiment (bf16 x 3 products): profiles/r03_bf16x3_experiment.txt
#define UGLAD_GEMM_TILE mfma_tile_bf16x3
#else
#define UGLAD_GEMM_TILE mfma_tile
#endif

template <int NT, bool TA, bool TB, bool UPPER>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ X, const float* __restrict__ Y,
                                         f32x16 (&acc)[Tiles<NT, UPPER>::kPerWave]) {
  constexpr int DP = NT * 32, LD = DP + 1;
  using T = Tiles<NT, UPPER>;
  const int w = __builtin_amdgcn_readfirstlane(opaque_v(threadIdx.x) >> 6);  // (scalar: tile indices and operand bases in SGPRs)
#pragma unroll
  for (int n = 0; n < T::kPerWave; ++n) {
    const int t = w + kWaves * n;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    if (t < T::kCount) {
      int I, J;
      T::ij(t, I, J);
      // A[i][k] = TA ? X[k][I*32+i] : X[I*32+i][k] ;  B[k][j] = TB ? Y[J*32+j][k] : Y[k][J*32+j]
      UGLAD_GEMM_TILE(TA ? X + I * 32 : X + I * 32 * LD, TA ? 1 : LD, TA ? LD : 1, TB ? Y + J * 32 * LD : Y + J * 32,
                      TB ? 1 : LD, TB ? LD : 1, DP, acc[n]);
    }
  }
}

// Store register tiles into an LDS matrix (full enumeration only).
template <int NT>
__device__ __forceinline__ void store_tiles(float* __restrict__ Y, const f32x16 (&acc)[Tiles<NT, false>::kPerWave]) {
  constexpr int LD = NT * 32 + 1;
  using T = Tiles<NT, false>;
  const int tid = opaque_v(threadIdx.x), lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
  for (int n = 0; n < T::kPerWave; ++n) {
    const int t = w + kWaves * n;
    if (t < T::kCount) {
      int I, J;
      T::ij(t, I, J);
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[(I * 32 + acc_row(e, lane)) * LD + J * 32 + (lane & 31)] = acc[n][e];
    }
  }
}

}  // namespace uglad
