// EXPERIMENT, measured and rejected in round 2 (profiles/r02_bwd_lean_experiment.txt): 0.44 ms per launch at
// M = 1024, D = 128 against the shipped cell_bwd_kernel's 0.29 ms.  Every phase of the backward cell is throughput-bound (the GEMMs
// run at 91 % of the MFMA peak, the rhoNN backward is VALU-bound), so a second workgroup on the CU only halves each one's speed,
// and this form does more work (full instead of symmetric products) through slower memory.  Built only with
// -DUGLAD_EXP_BWD_LEAN in round 2; round 3 moved it out of uglad_amd/csrc and removed its hooks from glad_kernels.hip (the include after
// cell_bwd_kernel, bwd_lean_floats() in uglad_workspace_floats, the dispatch in uglad_cell_bwd): commit e6d941b still has them wired.
// Kept for the record only; not compiled by anything.
//
// Backward cell for D <= 128 on ONE LDS matrix (the eigenvectors U): 77 KB of LDS and <= 128 registers, so that two
// workgroups share a CU.  Replaces autograd through glad.py:139-144, torch_sqrtm.py:32-46, glad_params.py:61-81 (SURVEY.md
// Appendix B), like cell_bwd_kernel, whose arithmetic it reproduces operation by operation in the entrywise parts.
//
// The chain  G_half -> C = U^T G_half U -> Y = C o F -> G_B = U Y U^T  multiplies by U from both sides.  The working matrix lives
// in ACCUMULATOR REGISTERS as strips of 16 columns or 16 rows (8 tiles of v_mfma_f32_16x16x4_f32 per wave); a product from
// the left keeps column strips, a product from the right keeps row strips, and between the two the matrix changes hands
// through a scratch slab in global memory (written by the workgroup itself a moment ago: L2), each wave fetching just its own
// strip -- 8 KB, all of it requested at once -- as the operand that does not come from LDS:
//
//   phase A   entrywise rhoNN / threshold backward on the evenly dealt upper triangle (as in the forward epilogue)
//             -> G_half (both triangles) to slab X0, direct part of dL/dZ to slab X2, 28 parameter-gradient partials
//   R   = U^T G_half      column strips:  A = U^T from LDS,  B = G_half(:, strip) from X0       -> X1
//   C   = R U             row strips:     A = R(strip, :) from X1,  B = U from LDS;  Y = C o F in registers  -> X0
//   T2  = U Y             column strips:  A = U from LDS,  B = Y(:, strip) from X0              -> X1
//   G_B = T2 U^T          row strips:     A = T2(strip, :) from X1,  B = U^T from LDS
//   G_out = direct part - G_B (row strips, 16 rows x 512 bytes per wave);  dL/dlambda partial
//
// The k order of an MFMA chain is free: lane group g = lane >> 4 takes k = c0 + 16 (g & 1) + 8 (g >> 1) + 4 p + s inside a
// chunk of 32 (16-byte pieces of a row strip; the two groups of a half-wave read LDS rows 16 apart: no bank conflicts).
#pragma once
#include "../eig_lean.h"

namespace uglad {

// scratch floats per matrix in the caller's workspace: three D x D slabs
__host__ __device__ constexpr int bwd_lean_floats(int D) { return 3 * D * D; }

#ifdef UGLAD_BWD_STAGGER
__device__ int g_bwd_slot[8 * 256];
#endif

template <int NT>
__global__ __launch_bounds__(kThreads, 4) void cell_bwd_lean_kernel(
    const float* __restrict__ Gnext, const float* __restrict__ S, const float* __restrict__ Zin,
    const float* __restrict__ half, const float* __restrict__ U, const float* __restrict__ beta,
    const float* __restrict__ lam_ptr, const float* __restrict__ params, float* __restrict__ Gout,
    float* __restrict__ grad_rho_partial, float* __restrict__ glam_partial, float* __restrict__ xws, int D, int mode, int gs) {
  constexpr int DP = NT * 32, LD = DP + 1, NQ = DP / 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float sU[DP * LD];
  __shared__ float s_beta[DP], s_r[DP];
  __shared__ __attribute__((aligned(16))) float s_a[kNsIters][DP];  // NS10: a_i^(t) ...
  __shared__ __attribute__((aligned(16))) float s_q[kNsIters][DP];  // ... and its square
  __shared__ float s_red[8];
  __shared__ float s_g[kWaves][kNRho + 1];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int goff = 16 * (g & 1) + 8 * (g >> 1);
  const size_t base = (size_t)blockIdx.x * D * D;
  const float* Sm = S + base;
  const float* Zm = Zin + base;
  const float* Hm = half + base;
  const float* Gm = Gnext + base;
  float* Go = Gout + base;
  float* X0 = xws + (size_t)blockIdx.x * bwd_lean_floats(D);
  float* X1 = X0 + (size_t)D * D;
  float* X2 = X1 + (size_t)D * D;
  const int grp = blockIdx.x / gs;
  params += (size_t)grp * kNParam;
  const float lam = lam_ptr[grp];
  const float c4 = 4.0f / lam, inv_lam2 = 1.0f / (lam * lam);
  const bool vec = ((D & 3) == 0) && ((reinterpret_cast<size_t>(xws) & 15) == 0);

  KSTAMP(0);
#ifdef UGLAD_BWD_STAGGER
  // experiment: the second workgroup of a CU's first pair starts UGLAD_BWD_STAGGER ticks late, so that its entrywise (VALU) phases
  // meet the other one's GEMM (MFMA) phases instead of marching in step with them
  if (blockIdx.x < 512) {
    __shared__ int s_slot;
    if (tid == 0) {
      const unsigned cu = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
      s_slot = atomicAdd(&g_bwd_slot[(xcc & 7) * 256 + (cu & 255)], 1);
    }
    __syncthreads();
    if (s_slot & 1) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)(UGLAD_BWD_STAGGER)) __builtin_amdgcn_s_sleep(64);
    }
  }
#endif
  // ---- U -> LDS (zero padding), spectrum
  for (int idx0 = 0; idx0 < DP * DP; idx0 += 8 * kThreads) {
    float u[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = idx0 + q * kThreads + tid;
      const int i = idx / DP, k = idx - i * DP;
      u[q] = ((idx < DP * DP) && i < D && k < D) ? U[base + i * D + k] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int idx = idx0 + q * kThreads + tid;
      if (idx < DP * DP) sU[(idx / DP) * LD + (idx % DP)] = u[q];
    }
  }
  float a2 = 0.f;
  if (tid < D) {
    const float be = beta[(size_t)blockIdx.x * D + tid];
    const float al = fmaf(be, be, c4);
    a2 = al * al;
  }
  const float nrmA = sqrtf(block_sum(a2, s_red));
  float r2 = 0.f;
  if (tid < DP) {
    float be = 0.f, r = 1.f;
    if (tid < D) {
      be = beta[(size_t)blockIdx.x * D + tid];
      r = sqrt_spectrum(be, c4, nrmA, mode);
      r2 = r * r;
    }
    s_beta[tid] = be;
    s_r[tid] = r;
  }
  const float nrmR = sqrtf(block_sum(r2, s_red));
  if (mode == UGLAD_SQRT_NS10 && tid < DP) {
    float a = s_r[tid] / nrmR;
#pragma unroll
    for (int it = 0; it < kNsIters; ++it) {
      s_a[it][tid] = a;
      s_q[it][tid] = a * a;
      a = 0.5f * a * (3.f - a * a);
    }
  }
  KSTAMP(1);
  // ---- phase A: rhoNN + threshold backward on the upper triangle, entry e = tid + kThreads q (see cell_fwd_kernel)
  {
    float gacc[kNRho];
#pragma unroll
    for (int q = 0; q < kNRho; ++q) gacc[q] = 0.f;
    constexpr int kMaxQ = ((DP / 2) * (DP + 1) + kThreads - 1) / kThreads;
    constexpr int kQ = kMaxQ < 4 ? kMaxQ : 4;
    const int D1 = D + 1, total = ((D + 1) / 2) * D1;
    const int sp = kThreads / D1, sc = kThreads - sp * D1;
    auto entry = [&](int e, int p, int c) -> int {
      if (e >= total) return -1;
      if (c < D - p) return (p << 16) | (p + c);
      const int i = D - 1 - p;
      return (i == p) ? -1 : ((i << 16) | (i + (c - (D - p))));
    };
    int p = tid / D1, c = tid - p * D1;
    for (int q0 = 0; q0 < kMaxQ; q0 += kQ) {
      int pk[kQ];
      float hx[kQ], zz[kQ], sv[kQ], gn[kQ];
#pragma unroll
      for (int u = 0; u < kQ; ++u) {
        pk[u] = (q0 + u < kMaxQ) ? entry(tid + kThreads * (q0 + u), p, c) : -1;
        c += sc;
        p += sp;
        if (c >= D1) {
          c -= D1;
          ++p;
        }
        const int i = pk[u] >> 16, j = pk[u] & 0xffff;
        const bool in = pk[u] >= 0;
        hx[u] = in ? Hm[i * D + j] : 0.f;
        zz[u] = in ? Zm[i * D + j] : 0.f;
        sv[u] = in ? Sm[i * D + j] : 0.f;
        gn[u] = in ? ((i == j) ? Gm[i * D + j] : 0.5f * (Gm[i * D + j] + Gm[j * D + i])) : 0.f;
      }
      constexpr int kQ2 = (kQ + 1) / 2;
#pragma unroll
      for (int h2 = 0; h2 < kQ2; ++h2) {
        const int u0 = 2 * h2, u1 = (2 * h2 + 1 < kQ) ? 2 * h2 + 1 : 2 * h2;
        const bool has1 = 2 * h2 + 1 < kQ;
        RhoAct2 act2;
        rho_forward2(params, (v2f){hx[u0], has1 ? hx[u1] : 0.f}, (v2f){sv[u0], has1 ? sv[u1] : 0.f},
                     (v2f){zz[u0], has1 ? zz[u1] : 0.f}, act2);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          const int u = c2 ? u1 : u0;
          if ((c2 == 0 || has1) && pk[u] >= 0) {
            const int i = pk[u] >> 16, j = pk[u] & 0xffff;
            const RhoAct act = act2.half(c2);
            const float x = hx[u];
            const bool active = fabsf(x) > act.rho;
            const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
            const float g_rho = active ? -sgn * gn[u] : 0.f;
            float gx1, gx3;
            rho_backward(params, x, sv[u], zz[u], act, g_rho, (i == j) ? 1.f : 2.f, gacc, gx1, gx3);
            const float gh = (active ? gn[u] : 0.f) + gx1;
            X0[i * D + j] = gh;
            X0[j * D + i] = gh;
            X2[i * D + j] = gx3;  // direct part of dL/dZ_in (upper triangle)
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kNRho; ++q) {
      const float v = wave_sum(gacc[q]);
      if (lane == 0) s_g[w][q] = v;
    }
  }
  __syncthreads();  // (U, spectrum and the slabs are in place)
  KSTAMP(2);

  const bool active = 16 * w < DP;
  const int strip = 16 * w;
  // one strip of a slab as MFMA operand: col strips feed B (k = row of the slab), row strips feed A (k = column of the slab)
  auto load_col_strip = [&](const float* X, float (&b)[NQ][4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c0 = 32 * (q >> 1), pp = q & 1;
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int k = c0 + goff + 4 * pp + ss, j = strip + l16;
        b[q][ss] = (k < D && j < D) ? X[k * D + j] : 0.f;
      }
    }
  };
  auto load_row_strip = [&](const float* X, float (&a)[NQ][4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int c0 = 32 * (q >> 1), pp = q & 1;
      const int k = c0 + goff + 4 * pp, i = strip + l16;
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < D && k < D) {
        const float* pp_ = X + (size_t)i * D + k;
        if (vec) {
          v = *reinterpret_cast<const f4*>(pp_);
        } else {
          v.x = pp_[0];
          if (k + 1 < D) v.y = pp_[1];
          if (k + 2 < D) v.z = pp_[2];
          if (k + 3 < D) v.w = pp_[3];
        }
      }
      a[q][0] = v.x;
      a[q][1] = v.y;
      a[q][2] = v.z;
      a[q][3] = v.w;
    }
  };
  // acc[t] (t = 16-row tile of the column strip): C[16 t + 4 g + r][strip + l16];  A[i][k] from LDS via `lds_a(t, k)`
  // acc[t] (t = 16-column tile of the row strip): C[strip + 4 g + r][16 t + l16];  B[k][j] from LDS via `lds_b(t, k)`
  float op[NQ][4];
  f32x4 acc[NQ];

  // ---- R = U^T G_half (column strips):  A[i][k] = U[k][16 t + i]
  if (active) {
    load_col_strip(X0, op);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = 32 * (q >> 1) + goff + 4 * (q & 1);
        const float* a = sU + k * LD + 16 * t + l16;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], op[q][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[LD], op[q][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2 * LD], op[q][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3 * LD], op[q][3], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * t + 4 * g + r, j = strip + l16;
        if (i < D && j < D) X1[i * D + j] = acc[t][r];
      }
    }
  }
  __syncthreads();
  KSTAMP(3);
  // ---- C = R U (row strips):  B[k][j] = U[k][16 t + j];  Y = C o F;  diagonal term of dL/dlam
  float glam = 0.f;
  if (active) {
    load_row_strip(X1, op);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = 32 * (q >> 1) + goff + 4 * (q & 1);
        const float* b = sU + k * LD + 16 * t + l16;
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][0], b[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][1], b[LD], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][2], b[2 * LD], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][3], b[3 * LD], acc[t], 0, 0, 0);
      }
    }
    KSTAMP(4);
    // rows i = strip + 4 g + r of this lane (four consecutive: one 16-byte LDS read per Newton-Schulz iterate)
    const int i0 = strip + 4 * g;
    const f4 bi4 = *reinterpret_cast<const f4*>(&s_beta[i0]);
    const f4 ri4 = *reinterpret_cast<const f4*>(&s_r[i0]);
    const float bi[4] = {bi4.x, bi4.y, bi4.z, bi4.w};
    const float ri[4] = {ri4.x, ri4.y, ri4.z, ri4.w};
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      const int j = 16 * t + l16;
      const float bj = s_beta[j], rj = s_r[j];
      float Kr[4];
      if (mode == UGLAD_SQRT_EXACT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Kr[r] = 1.0f / (ri[r] + rj);
      } else {
        float P[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int it = 0; it < kNsIters; ++it) {
          const f4 a4 = *reinterpret_cast<const f4*>(&s_a[it][i0]);
          const f4 q4 = *reinterpret_cast<const f4*>(&s_q[it][i0]);
          const float aj = s_a[it][j], qj = s_q[it][j];
          P[0] *= 0.5f * (3.f - q4.x - qj + a4.x * aj);
          P[1] *= 0.5f * (3.f - q4.y - qj + a4.y * aj);
          P[2] *= 0.5f * (3.f - q4.z - qj + a4.z * aj);
          P[3] *= 0.5f * (3.f - q4.w - qj + a4.w * aj);
        }
        const float sc = 1.0f / (2.f * nrmR);
#pragma unroll
        for (int r = 0; r < 4; ++r) Kr[r] = P[r] * sc;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + r;
        float v = 0.f;
        if (i < D && j < D) {
          const float cij = acc[t][r];
          if (i == j) glam = fmaf(cij, -2.f * Kr[r] * inv_lam2, glam);
          v = cij * 0.5f * fmaf(bi[r] + bj, Kr[r], -1.f);
          X0[i * D + j] = v;
        }
      }
    }
  }
  __syncthreads();
  KSTAMP(5);
  // ---- T2 = U Y (column strips):  A[i][k] = U[16 t + i][k]
  if (active) {
    load_col_strip(X0, op);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* arow = sU + (16 * t + l16) * LD;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float* a = arow + 32 * (q >> 1) + goff + 4 * (q & 1);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], op[q][0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], op[q][1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], op[q][2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], op[q][3], acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * t + 4 * g + r, j = strip + l16;
        if (i < D && j < D) X1[i * D + j] = acc[t][r];
      }
    }
  }
  __syncthreads();
  KSTAMP(6);
  // ---- G_B = T2 U^T (row strips):  B[k][j] = U[16 t + j][k];  G_out = direct part - G_B;  dL/dlam -= <S, G_B> / lam^2
  if (active) {
    load_row_strip(X1, op);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* brow = sU + (16 * t + l16) * LD;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float* b = brow + 32 * (q >> 1) + goff + 4 * (q & 1);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][0], b[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][1], b[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][2], b[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[q][3], b[3], acc[t], 0, 0, 0);
      }
    }
    KSTAMP(7);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = strip + 4 * g + r, j = 16 * t + l16;
        if (i < D && j < D) {
          const float gb = acc[t][r];
          const float gz = (i <= j) ? X2[i * D + j] : X2[j * D + i];
          Go[i * D + j] = gz - gb;
          glam = fmaf(-Sm[i * D + j] * inv_lam2, gb, glam);
        }
      }
    }
  }
  KSTAMP(8);
  // ---- reductions: 28 rhoNN gradients (summed per wave after phase A) + dL/dlam
  {
    const float v = wave_sum(glam);
    if (lane == 0) s_g[w][kNRho] = v;
  }
  __syncthreads();
  if (tid <= kNRho) {
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < kWaves; ++ww) v += s_g[ww][tid];
    if (tid < kNRho)
      grad_rho_partial[(size_t)blockIdx.x * kNRho + tid] += v;
    else
      glam_partial[blockIdx.x] = v;
  }
  KSTAMP(9);
}

}  // namespace uglad
