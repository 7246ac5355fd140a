// Backward cell for FEW, LARGE matrices (D > 128 and a batch that cannot fill the chip with one workgroup per matrix: BASELINE
// config 5 puts ONE 256 x 256 matrix on each GPU): the same arithmetic as cell_bwd_kernel (glad.py:139-144, torch_sqrtm.py:32-46,
// glad_params.py:61-81), split over many workgroups per matrix.  A launch boundary is the only grid-wide
// barrier the products need, so the cell becomes six short launches instead of one long-lived workgroup per matrix:
//
//   wide_phase_a     entrywise rhoNN / threshold backward, four workgroups per upper 64 x 64 tile: G_half -> X0 (both triangles),
//                    direct part of dL/dZ -> G_out (both triangles), 28 gradient partials per workgroup
//   wide_gemm<TN>    R  = U^T X0                 -> X1      one workgroup per 64 x 64 output tile, operands staged through LDS in
//   wide_gemm<NN>    Y  = (R U) o F              -> X0      k chunks of 64 with the next chunk prefetched into registers;
//   wide_gemm<NN>    T2 = U Y                    -> X1      epilogues: the Newton-Schulz divided differences (Y) and
//   wide_gemm<NT>    G_out -= T2 U^T                        G_out = direct part - G_B with the dL/dlambda partial
//   wide_reduce      partials -> grad_rho_partial (+=), glam_partial (=), in fixed order (deterministic)
//
// The same tile product with other epilogues serves the forward cell (theta_half + rhoNN: kEpiThetaHalf) and the two explicit
// inverses of a pass (Theta_0, the loss's Theta_L^-1: kEpiInverse / kEpiResidual / kEpiNewton); wide_fwd.h holds the last merge of
// the divide & conquer.
//
// X0, X1: the two D x D slabs per matrix the single-workgroup kernel keeps in the workspace for D > 128; the partial sums live in
// the region the forward uses for (d, e, tau) and the T factors.
#pragma once
#include "glad_device.h"
#include "tridiag.h"  // f4

namespace uglad {

constexpr int kWT = 64;         // output tile of a workgroup (four waves, one 32 x 32 MFMA tile each)
constexpr int kWK = 64;         // k chunk staged in LDS (four round trips to L2 per product at D = 256; 32 measured the same: launch-bound)
constexpr int kWThreads = 256;
constexpr int kWLd = kWT + 1;   // LDS row stride of a staged chunk ([k][x]: conflict-free operand reads and scatter stores)
constexpr int kWMaxD = 256;
constexpr int kWQ = 4;          // phase A: workgroups per tile

__host__ __device__ constexpr int wide_tiles(int D) { return (D + kWT - 1) / kWT; }
// floats of per-matrix partial sums: [kWQ x upper tiles][28] + [tiles^2] (Y epilogue) + [tiles^2] (G_out epilogue)
__host__ __device__ constexpr int wide_partial_floats(int D) {
  return kWQ * (wide_tiles(D) * (wide_tiles(D) + 1) / 2) * kNRho + 2 * wide_tiles(D) * wide_tiles(D);
}

__device__ __forceinline__ float wide_block_sum(float v, float* s4) {  // 256 threads; s4: 4 floats of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ---- phase A: kWQ workgroups per upper tile (I <= J) of one matrix, 64 / kWQ rows of the tile each
__global__ __launch_bounds__(kWThreads) void wide_phase_a_kernel(
    const float* __restrict__ Gnext, const float* __restrict__ S, const float* __restrict__ Zin, const float* __restrict__ half,
    const float* __restrict__ params, float* __restrict__ X0, float* __restrict__ Gout, float* __restrict__ partial, int D, int gs,
    size_t slab_stride, size_t partial_stride) {
  __shared__ float s_g[4][kNRho];
  const int m = blockIdx.y, nt = wide_tiles(D);
  const int rq = blockIdx.x % kWQ;  // which rows of the tile
  int t = blockIdx.x / kWQ, I = 0;
  while (t >= nt - I) {  // upper tiles row by row: row I has nt - I of them
    t -= nt - I;
    ++I;
  }
  const int J = I + t;
  const size_t base = (size_t)m * D * D;
  const float* Gm = Gnext + base;
  const float* Sm = S + base;
  const float* Zm = Zin + base;
  const float* Hm = half + base;
  float* Xm = X0 + (size_t)m * slab_stride;
  float* Go = Gout + base;
  params += (size_t)(m / gs) * kNParam;
  float g[kNRho];
#pragma unroll
  for (int q = 0; q < kNRho; ++q) g[q] = 0.f;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // two entries per thread and pass (rows r0 + w, r0 + w + 4 of the tile: 64 consecutive columns per wave and row)
  for (int r0 = rq * (kWT / kWQ); r0 < (rq + 1) * (kWT / kWQ); r0 += 8) {
    const int j = J * kWT + lane;
    int iv[2];
    float hx[2], zz[2], sv[2], gn[2];
    bool in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      iv[u] = I * kWT + r0 + w + 4 * u;
      in[u] = iv[u] < D && j < D && iv[u] <= j;
      const int i = iv[u];
      hx[u] = in[u] ? Hm[(size_t)i * D + j] : 0.f;
      zz[u] = in[u] ? Zm[(size_t)i * D + j] : 0.f;
      sv[u] = in[u] ? Sm[(size_t)i * D + j] : 0.f;
      gn[u] = in[u] ? ((i == j) ? Gm[(size_t)i * D + j] : 0.5f * (Gm[(size_t)i * D + j] + Gm[(size_t)j * D + i])) : 0.f;
    }
    RhoAct2 act2;
    rho_forward2(params, (v2f){hx[0], hx[1]}, (v2f){sv[0], sv[1]}, (v2f){zz[0], zz[1]}, act2);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (in[u]) {
        const int i = iv[u];
        const RhoAct act = act2.half(u);
        const float x = hx[u];
        const bool active = fabsf(x) > act.rho;
        const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
        const float g_rho = active ? -sgn * gn[u] : 0.f;
        float gx1, gx3;
        rho_backward(params, x, sv[u], zz[u], act, g_rho, (i == j) ? 1.f : 2.f, g, gx1, gx3);
        const float gh = (active ? gn[u] : 0.f) + gx1;
        Xm[(size_t)i * D + j] = gh;
        Xm[(size_t)j * D + i] = gh;
        Go[(size_t)i * D + j] = gx3;  // direct part of dL/dZ_in, finished by the last product's epilogue
        Go[(size_t)j * D + i] = gx3;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < kNRho; ++q) {
    const float v = wave_sum(g[q]);
    if (lane == 0) s_g[w][q] = v;
  }
  __syncthreads();
  if (tid < kNRho) partial[(size_t)m * partial_stride + (size_t)blockIdx.x * kNRho + tid] = (s_g[0][tid] + s_g[1][tid]) + (s_g[2][tid] + s_g[3][tid]);
}

// ---- tile GEMM.  C(i, j) = sum_k A(i, k) B(k, j) on the 64 x 64 tile (blockIdx.y, blockIdx.x) of matrix blockIdx.z, where
// A(i, k) = TA ? Ag[k][i] : Ag[i][k] and B(k, j) = TB ? Bg[j][k] : Bg[k][j] (row-major D x D, per-matrix strides as given).
enum { kEpiStore = 0, kEpiDivDiff = 1, kEpiGout = 2, kEpiThetaHalf = 3, kEpiInverse = 4, kEpiResidual = 5, kEpiNewton = 6, kEpiDotT = 7 };
//   kEpiDotT      no store: partial = <X^T, acc> over the tile (X: WideFwd's extra operand; the gradient of Theta_0's shift beyond the
//                 eigensolver's size, wide_ns.h)
// The inverse of a symmetric matrix A + shift I from its eigen-decomposition, with one Newton step (spectral_to_global's arithmetic):
//   kEpiInverse   X0 = (U diag(1 / (beta + shift))) U^T, upper tiles, stored symmetrically        (beta: head of the partial record)
//   kEpiResidual  E  = I - (A + shift I) X0          = I - acc - shift X0[i][j]                     (X0 = the B operand)
//   kEpiNewton    X  = X0 + X0 E                     = X0[i][j] + acc, upper tiles, stored symmetrically (X0 = the A operand)
// `shift` travels in lam_ptr's place as a device scalar per group (theta_init_offset), or nullptr for 0.

// Extra operands of the forward epilogue (kEpiThetaHalf): theta_half = (U diag(phi)) U^T on the upper tiles, then rhoNN + soft
// threshold, both triangles of Z (and theta_half when training) stored from the one computed value, ||Z - theta_half||^2 per tile.
struct WideFwd {
  const float* Zin;
  const float* params;
  float* half_out;  // may be null (inference)
  float* cond_max;  // may be null: running maximum of cond(b^T b + 4/lam I) per matrix (uglad_cell_fwd)
  const float* X;   // kEpiDotT: the matrix of the inner product, row stride D, x_stride floats per matrix
  size_t x_stride;
};

template <bool TA, bool TB, int EPI>
__global__ __launch_bounds__(kWThreads) void wide_gemm_kernel(
    const float* __restrict__ Ag, size_t a_stride, const float* __restrict__ Bg, size_t b_stride, float* __restrict__ Cg,
    size_t c_stride, const float* __restrict__ S, const float* __restrict__ beta, const float* __restrict__ lam_ptr,
    float* __restrict__ partial, size_t partial_stride, int partial_off, int D, int mode, int gs, int lda, int ldb, int ldc,
    WideFwd fw) {
  __shared__ float sA[kWK * kWLd], sB[kWK * kWLd];
  __shared__ float s4[4];
  __shared__ float s_phi[(EPI == kEpiThetaHalf || EPI == kEpiInverse) ? kWMaxD : 1];
  __shared__ double s_dbl[EPI == kEpiThetaHalf ? 3 * kWMaxD + 80 : 1];
  if ((EPI == kEpiThetaHalf || EPI == kEpiInverse || EPI == kEpiNewton) && blockIdx.y > blockIdx.x) return;  // (symmetric: upper tiles only; uniform per workgroup)
  __shared__ float s_beta[EPI == kEpiDivDiff ? kWMaxD : 1], s_r[EPI == kEpiDivDiff ? kWMaxD : 1];
  __shared__ float s_a[EPI == kEpiDivDiff ? kNsIters : 1][EPI == kEpiDivDiff ? 2 * kWT : 1];  // NS10: a^(t) of [0..63] rows, [64..127] columns
  __shared__ float s_q[EPI == kEpiDivDiff ? kNsIters : 1][EPI == kEpiDivDiff ? 2 * kWT : 1];  // ... and its square
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = blockIdx.z, I = blockIdx.y, J = blockIdx.x;
  const float* A = Ag + (size_t)m * a_stride;
  const float* B = Bg + (size_t)m * b_stride;
  float* C = Cg + (size_t)m * c_stride;
  const int i0 = I * kWT, j0 = J * kWT;
  constexpr bool kInv = EPI == kEpiInverse || EPI == kEpiResidual || EPI == kEpiNewton;
  const float lam = kInv ? 1.f : ((EPI != kEpiStore && EPI != kEpiDotT) ? lam_ptr[m / gs] : 1.f);
  const float shift = (kInv && lam_ptr) ? lam_ptr[(size_t)(m / gs) * partial_off] : 0.f;  // (partial_off: stride between the groups' scalars)
  const float c4 = 4.0f / lam, inv_lam2 = 1.0f / (lam * lam);
  float nrmR = 1.f;

  float alpha = 0.f;
  if (EPI == kEpiThetaHalf) {  // psi(beta) = phi(beta) + alpha beta of the shifted form theta_half = -alpha b + U diag(psi) U^T (glad_device.h)
    const float* bm = beta + (size_t)m * partial_stride;  // (beta sits at the head of the matrix's partial-sum region)
    const float be = (tid < D) ? bm[tid] : 0.f;
    float cond;
    s_phi[tid] = shifted_spectrum(be, D, lam, mode, s_dbl, alpha, cond);
    if (fw.cond_max && tid == 0 && I == 0 && J == 0) fw.cond_max[m] = fmaxf(fw.cond_max[m], cond);
  }
  if (EPI == kEpiInverse) {  // f = 1 / (beta + shift)
    const float* bm = beta + (size_t)m * partial_stride;
    s_phi[tid] = (tid < D) ? 1.0f / (bm[tid] + shift) : 0.f;
  }
  if (EPI == kEpiDivDiff) {  // spectrum of this matrix: r_i (sqrt_spectrum), and for NS10 the iterates a_i^(t) of the rows / columns here
    const float* bm = beta + (size_t)m * D;
    const float be = (tid < D) ? bm[tid] : 0.f;
    float a2 = 0.f;
    if (tid < D) {
      const float al = fmaf(be, be, c4);
      a2 = al * al;
    }
    const float nrmA = sqrtf(wide_block_sum(a2, s4));
    float r = 1.f, r2 = 0.f;
    if (tid < D) {
      r = sqrt_spectrum(be, c4, nrmA, mode);
      r2 = r * r;
    }
    s_beta[tid] = be;
    s_r[tid] = r;
    nrmR = sqrtf(wide_block_sum(r2, s4));
    if (mode == UGLAD_SQRT_NS10 && tid < 2 * kWT) {
      const int idx = (tid < kWT) ? i0 + tid : j0 + tid - kWT;
      float a = ((idx < D) ? s_r[idx] : 1.f) / nrmR;
#pragma unroll
      for (int it = 0; it < kNsIters; ++it) {
        s_a[it][tid] = a;
        s_q[it][tid] = a * a;
        a = 0.5f * a * (3.f - a * a);
      }
    }
  }

  // staging: thread -> (x, k) of the two operand chunks; a chunk is 64 x 64 floats = 16 per thread, four runs of four consecutive
  // source elements (one 16-byte load each when the layout allows)
  //   source contiguous in k (A not transposed / B transposed): x = tid / 16 + 16 p, k = 4 (tid % 16) .. + 3
  //   source contiguous in x (A transposed / B not transposed): k = tid / 16 + 16 p, x = 4 (tid % 16) .. + 3
  constexpr int kPF = kWT * kWK / kWThreads;  // 16
  float pa[kPF], pb[kPF];
  auto fetch = [&](const float* __restrict__ src, int ld, bool contig_k, int x0, int k0, float (&p)[kPF]) {
    const bool vec = ((ld & 3) == 0) && ((D & 3) == 0) && ((reinterpret_cast<size_t>(src) & 15) == 0);
#pragma unroll
    for (int pp = 0; pp < kPF / 4; ++pp) {
      const int r = (tid >> 4) + 16 * pp, c0 = 4 * (tid & 15);
      const int row = contig_k ? x0 + r : k0 + r;      // the source row this run lies in
      const int col = contig_k ? k0 + c0 : x0 + c0;    // first source column of the run
      if (vec) {  // (D % 4 == 0: a run is inside the matrix or outside as a whole)
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < D && col < D) v = *reinterpret_cast<const f4*>(src + (size_t)row * ld + col);
        p[4 * pp] = v.x;
        p[4 * pp + 1] = v.y;
        p[4 * pp + 2] = v.z;
        p[4 * pp + 3] = v.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) p[4 * pp + c] = (row < D && col + c < D) ? src[(size_t)row * ld + col + c] : 0.f;
      }
    }
  };
  auto stash = [&](float* dst, bool contig_k, const float (&p)[kPF], int k0, bool scale) {
#pragma unroll
    for (int pp = 0; pp < kPF / 4; ++pp) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = (tid >> 4) + 16 * pp, cc = 4 * (tid & 15) + c;
        const int k = contig_k ? cc : r, x = contig_k ? r : cc;
        dst[k * kWLd + x] = ((EPI == kEpiThetaHalf || EPI == kEpiInverse) && scale) ? p[4 * pp + c] * s_phi[(k0 + k < kWMaxD) ? k0 + k : 0] : p[4 * pp + c];
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32, li = lane & 31, kh = lane >> 5;
  fetch(A, lda, !TA, i0, 0, pa);
  fetch(B, ldb, TB, j0, 0, pb);
  for (int k0 = 0; k0 < D; k0 += kWK) {
    __syncthreads();  // (the previous chunk has been consumed; first pass: the prologue's LDS arrays are complete)
    stash(sA, !TA, pa, k0, true);
    stash(sB, TB, pb, k0, false);
    __syncthreads();
    if (k0 + kWK < D) {
      fetch(A, lda, !TA, i0, k0 + kWK, pa);
      fetch(B, ldb, TB, j0, k0 + kWK, pb);
    }
#pragma unroll
    for (int u = 0; u < kWK / 2; ++u) {
      const int k = 2 * u + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[k * kWLd + wi + li], sB[k * kWLd + wj + li], acc, 0, 0, 0);
    }
  }

  // ---- epilogue: accumulator entry e of lane l = C[i0 + wi + acc_row(e, l)][j0 + wj + (l & 31)]
  const int j = j0 + wj + li;
  float glam = 0.f;
  if (EPI == kEpiStore) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = i0 + wi + acc_row(e, lane);
      if (i < D && j < D) C[(size_t)i * ldc + j] = acc[e];
    }
  } else if (EPI == kEpiDotT) {
    const float* Xm = fw.X + (size_t)m * fw.x_stride;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = i0 + wi + acc_row(e, lane);
      if (i < D && j < D) glam = fmaf(Xm[(size_t)j * D + i], acc[e], glam);
    }
  } else if (EPI == kEpiDivDiff) {
    const int jc = (j < D) ? j : 0;
    const float bj = s_beta[jc], rj = s_r[jc];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int il = wi + acc_row(e, lane), i = i0 + il;
      if (i < D && j < D) {
        float K;
        if (mode == UGLAD_SQRT_EXACT) {
          K = 1.0f / (s_r[i] + rj);
        } else {
          float P = 1.f;
#pragma unroll
          for (int it = 0; it < kNsIters; ++it) {
            const float ai = s_a[it][il], aj = s_a[it][kWT + wj + li];
            P *= 0.5f * (3.f - s_q[it][il] - s_q[it][kWT + wj + li] + ai * aj);
          }
          K = P * (1.0f / (2.f * nrmR));
        }
        const float cij = acc[e];
        if (i == j) glam = fmaf(cij, -2.f * K * inv_lam2, glam);
        C[(size_t)i * ldc + j] = cij * 0.5f * fmaf(s_beta[i] + bj, K, -1.f);
      }
    }
  } else if (EPI == kEpiInverse || EPI == kEpiNewton) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = i0 + wi + acc_row(e, lane);
      if (i < D && j < D && i <= j) {
        const float v = (EPI == kEpiNewton) ? A[(size_t)i * lda + j] + acc[e] : acc[e];  // (Newton: the A operand is X0)
        C[(size_t)i * ldc + j] = v;
        C[(size_t)j * ldc + i] = v;
      }
    }
  } else if (EPI == kEpiResidual) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = i0 + wi + acc_row(e, lane);
      if (i < D && j < D) C[(size_t)i * ldc + j] = ((i == j) ? 1.f : 0.f) - acc[e] - shift * B[(size_t)i * ldb + j];  // (the B operand is X0)
    }
  } else if (EPI == kEpiThetaHalf) {
    const float* Sm = S + (size_t)m * D * D;
    const float* Zm = fw.Zin + (size_t)m * D * D;
    float* Hm = fw.half_out ? fw.half_out + (size_t)m * D * D : nullptr;
    const float* prm = fw.params + (size_t)(m / gs) * kNParam;
    const float inv_lam = 1.0f / lam;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {  // two entries per pass on the packed pipe
      int iv[2];
      float xv[2], sv[2], zv[2];
      bool in[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        iv[u] = i0 + wi + acc_row(e + u, lane);
        in[u] = iv[u] < D && j < D && iv[u] <= j;
        sv[u] = in[u] ? Sm[(size_t)iv[u] * D + j] : 0.f;
        zv[u] = in[u] ? Zm[(size_t)iv[u] * D + j] : 0.f;
        xv[u] = in[u] ? fmaf(-alpha, fmaf(inv_lam, sv[u], -zv[u]), acc[e + u]) : 0.f;  // b = S/lam - Z with tridiag_kernel's rounding
      }
      RhoAct2 act;
      rho_forward2(prm, (v2f){xv[0], xv[1]}, (v2f){sv[0], sv[1]}, (v2f){zv[0], zv[1]}, act);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (in[u]) {
          const int i = iv[u];
          const float zn = soft_threshold(xv[u], u ? act.rho.y : act.rho.x);
          const float d = zn - xv[u];
          glam = fmaf((i == j) ? 1.f : 2.f, d * d, glam);
          C[(size_t)i * ldc + j] = zn;
          C[(size_t)j * ldc + i] = zn;
          if (Hm) {
            Hm[(size_t)i * D + j] = xv[u];
            Hm[(size_t)j * D + i] = xv[u];
          }
        }
      }
    }
  } else {
    const float* Sm = S + (size_t)m * D * D;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int i = i0 + wi + acc_row(e, lane);
      if (i < D && j < D) {
        const float gb = acc[e];
        C[(size_t)i * ldc + j] -= gb;  // (holds the direct part of dL/dZ_in since phase A)
        glam = fmaf(-Sm[(size_t)i * D + j] * inv_lam2, gb, glam);
      }
    }
  }
  if (EPI != kEpiStore && !kInv) {
    const float v = wide_block_sum(glam, s4);
    if (tid == 0) partial[(size_t)m * partial_stride + partial_off + I * gridDim.x + J] = v;
  }
}

// ---- forward: per-tile ||Z - theta_half||^2 -> normF_partial[m], in a fixed order
__global__ void wide_norm_reduce_kernel(const float* __restrict__ partial, size_t partial_stride, int partial_off, float* __restrict__ normF_partial,
                                        int M, int D) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x, nt = wide_tiles(D);
  if (m >= M) return;
  float v = 0.f;
  for (int I = 0; I < nt; ++I)
    for (int J = I; J < nt; ++J) v += partial[(size_t)m * partial_stride + partial_off + I * nt + J];
  normF_partial[m] = v;
}

// ---- loss of few large matrices: tr(S Theta) (+ the log-cosh structure penalty) per 64-row block -> partial; then, with the spectrum
// of Theta from the eigen-decomposition, loss_partial[m] = -logdet + trace term (loss_fwd_kernel's arithmetic and NaN / -inf rules)
__device__ __forceinline__ float wide_log_cosh(float x) {
  const float a = fabsf(x);
  return a + log1pf(expf(-2.f * a)) - 0.69314718056f;
}
__global__ __launch_bounds__(kWThreads) void wide_loss_trace_kernel(const float* __restrict__ theta, const float* __restrict__ S, int s_batch,
                                                                    const float* __restrict__ struct_theta, float* __restrict__ partial,
                                                                    size_t partial_stride, int partial_off, int D) {
  __shared__ float s4[4];
  const int m = blockIdx.y, tid = threadIdx.x;
  const size_t base = (size_t)m * D * D, sbase = (size_t)(m % s_batch) * D * D;
  float tr = 0.f;
  const int r0 = blockIdx.x * kWT, r1 = (r0 + kWT < D) ? r0 + kWT : D;
  for (int idx = r0 * D + tid; idx < r1 * D; idx += kWThreads) {
    const int i = idx / D, j = idx - i * D;
    const float th = theta[base + idx];
    tr = fmaf(S[sbase + (size_t)j * D + i], th, tr);
    if (struct_theta) {
      const float mask = (1.f - struct_theta[sbase + idx]) - ((i == j) ? 1.f : 0.f);
      tr += wide_log_cosh(th * mask);
    }
  }
  tr = wide_block_sum(tr, s4);
  if (tid == 0) partial[(size_t)m * partial_stride + partial_off + blockIdx.x] = tr;
}
__global__ void wide_loss_finish_kernel(const float* __restrict__ partial, size_t partial_stride, int partial_off, float* __restrict__ loss_partial,
                                        int M, int D) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* rec = partial + (size_t)m * partial_stride;
  float tr = 0.f;
  for (int t = 0; t < wide_tiles(D); ++t) tr += rec[partial_off + t];
  float lad = 0.f;
  int neg = 0, zero = 0;
  for (int i = 0; i < D; ++i) {  // beta: head of the record
    const float be = rec[i];
    lad += logf(fabsf(be));
    neg += (be < 0.f) ? 1 : 0;
    zero += (be == 0.f) ? 1 : 0;
  }
  float logdet = lad;
  if (neg & 1) logdet = __builtin_nanf("");
  if (zero > 0) logdet = -__builtin_inff();
  loss_partial[m] = -logdet + tr;
}

// ---- partial sums -> the cell's outputs, in a fixed order: one wave per (matrix, output), lane t takes partial t, DPP tree
__global__ __launch_bounds__(64) void wide_reduce_kernel(const float* __restrict__ partial, size_t partial_stride, float* __restrict__ grad_rho_partial,
                                                         float* __restrict__ glam_partial, int D) {
  const int m = blockIdx.x, q = blockIdx.y, lane = threadIdx.x, nt = wide_tiles(D), nup = kWQ * (nt * (nt + 1) / 2);
  const float* p = partial + (size_t)m * partial_stride;
  float v = 0.f;
  if (q < kNRho) {
    for (int t = lane; t < nup; t += 64) v += p[t * kNRho + q];
  } else {
    for (int t = lane; t < 2 * nt * nt; t += 64) v += p[nup * kNRho + t];
  }
  v = wave_sum(v);
  if (lane == 0) {
    if (q < kNRho) grad_rho_partial[(size_t)m * kNRho + q] += v;
    else glam_partial[m] = v;
  }
}

}  // namespace uglad
