// Few large matrices (see wide_bwd.h): the LAST merge of the divide & conquer -- 256 poles at D = 256, a third of the second stage
// when one workgroup does it -- as two launches with many workgroups per matrix.  The roots of the secular equation are
// independent of each other, and so are the 64 x 64 tiles of the eigenvector update Q <- (Q W') diag(1 / ||W'_j||).
//
//   wide_secular_kernel   32 roots per workgroup, eight lanes per root with 32 poles each in registers (secular_root_reg)
//   wide_merge_kernel     one workgroup per 64 x 64 tile of the new Q: Gu-Eisenstat zhat for all poles and the norms of the tile's
//                         columns in the prologue (recomputed per workgroup: 256 x 256 terms, cheaper than another launch), then
//                         the product with W'[k][j] = zhat_k / ((d_k - dk_j) - mu_j) generated while the B operand is staged.
//                         Q before the merge is block diagonal (two halves), so a row tile only meets the k range of its half.
//
// Input: the record dc_tridiagonal_lean(..., last) leaves per matrix (eig_lean.h); the secular launch appends dk, mu at
// [5 DP, 7 DP) and writes the merged eigenvalues over d in the matrix's (d, e, tau) record.
#pragma once
#include "eig_lean.h"
#include "wide_bwd.h"

namespace uglad {

__global__ __launch_bounds__(kWThreads) void wide_secular_kernel(float* __restrict__ last_base, size_t last_stride, float* __restrict__ tri_base,
                                                                 size_t tri_stride, int n, int DP) {
  __shared__ float s_ds[kWMaxD], s_zh[kWMaxD];
  const int tid = threadIdx.x, m = blockIdx.y;
  float* last = last_base + (size_t)m * last_stride;
  s_ds[tid] = (tid < n) ? last[tid] : 3.0e38f;
  s_zh[tid] = (tid < n) ? last[2 * DP + tid] : 0.f;
  __syncthreads();
  const float rho = last[4 * DP];
  const int skip = reinterpret_cast<const int*>(last)[4 * DP + 1];
  const int root = blockIdx.x * (kWThreads / 8) + tid / 8, sub = tid % 8;
  if (root >= n) return;  // (whole groups of eight)
  int K = root;
  float mu = 0.f;
  if (!skip) (void)secular_root_reg<8, kWMaxD / 8>(s_ds, s_zh, rho, n, root, sub, K, mu);
  if (sub == 0) {
    const float dK = s_ds[K];
    last[5 * DP + root] = dK;
    last[6 * DP + root] = mu;
    tri_base[(size_t)m * tri_stride + root] = dK + mu;  // the merged eigenvalue, ascending in `root`
  }
}

// Qold, Qnew: the matrix's two slabs (row stride ld = DP + 1); every entry of the DP x DP result is written (identity on the padding)
__global__ __launch_bounds__(kWThreads) void wide_merge_kernel(const float* __restrict__ Qold_base, float* __restrict__ Qnew_base,
                                                               size_t slab_stride, const float* __restrict__ last_base, size_t last_stride,
                                                               int n, int DP, int ld) {
  __shared__ float sA[kWK * kWLd], sB[kWK * kWLd];
  __shared__ float s_ds[kWMaxD], s_dk[kWMaxD], s_mu[kWMaxD];
  __shared__ float s_dso[kWMaxD], s_invo[kWMaxD];  // poles / signed zhat in ORIGINAL column order (the k index of the product)
  __shared__ int s_perm[kWMaxD];
  __shared__ float s_sc[kWT], s_part[4][kWT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = blockIdx.z, I = blockIdx.y, J = blockIdx.x;
  const int i0 = I * kWT, j0 = J * kWT;
  const float* last = last_base + (size_t)m * last_stride;
  const float* Qo = Qold_base + (size_t)m * slab_stride;
  float* Qn = Qnew_base + (size_t)m * slab_stride;
  const int skip = reinterpret_cast<const int*>(last)[4 * DP + 1];
  const int h = 128;  // the merge joins [0, 128) and [128, n)

  // ---- prologue: vectors of the merge; zhat_j (thread j = sorted position of the pole)
  float zs = 0.f;
  if (tid < n) {
    s_ds[tid] = last[tid];
    zs = last[DP + tid];
    s_perm[tid] = reinterpret_cast<const int*>(last)[3 * DP + tid];
    s_dk[tid] = last[5 * DP + tid];
    s_mu[tid] = last[6 * DP + tid];
  } else {
    s_ds[tid] = 3.0e38f;
    s_perm[tid] = tid;
    s_dk[tid] = 0.f;
    s_mu[tid] = 0.f;
  }
  s_dso[tid] = 3.0e38f;  // (a pole at infinity with weight zero: contributes an exact zero to every column)
  s_invo[tid] = 0.f;
  __syncthreads();
  if (tid < n && !skip) {
    const float dj = s_ds[tid];
    float prod = 1.f;
    for (int i = 0; i < n; ++i) {
      const float num = (s_dk[i] - dj) + s_mu[i];
      const float den = (i == tid) ? 1.f : s_ds[i] - dj;
      prod *= num * fast_rcp(den);
    }
    const float zhat = sqrtf(fmaxf(prod, 0.f));
    const int g = s_perm[tid];
    s_invo[g] = (zs < 0.f) ? -zhat : zhat;
    s_dso[g] = dj;
  }
  __syncthreads();
  // B(k, j): column j = sorted position of the root, k = original column
  auto wgen = [&](int k, int j) -> float {
    if (j >= n || k >= n) return (k == j) ? 1.f : 0.f;                      // padding: identity
    if (skip) return (k == s_perm[j]) ? 1.f : 0.f;                          // uncoupled halves: the merge only sorts
    return s_invo[k] * fast_rcp((s_dso[k] - s_dk[j]) - s_mu[j]);
  };
  {  // norms of this tile's 64 columns: thread (c = tid % 64, part = tid / 64) sums a quarter of the k range
    const int c = tid & 63, part = tid >> 6, j = j0 + c;
    float s2 = 0.f;
    if (j < n && !skip) {
      const int kq = (n + 3) / 4;
      for (int k = part * kq; k < (part + 1) * kq && k < n; ++k) {
        const float v = wgen(k, j);
        s2 = fmaf(v, v, s2);
      }
    }
    s_part[part][c] = s2;
    __syncthreads();
    if (tid < kWT) {
      const float t = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
      s_sc[tid] = (j0 + tid < n && !skip) ? 1.0f / sqrtf(t) : 1.f;
    }
  }
  __syncthreads();

  // ---- the product on this tile.  k range of the row tile: rows of the first half meet k in [0, h), the others [h, DP)
  const int kbeg = (i0 < h) ? 0 : h, kend = (i0 < h) ? h : DP;  // (h and the tiles are multiples of 64)
  constexpr int kPF = kWT * kWK / kWThreads;
  float pa[kPF];
  auto fetch_a = [&](int k0) {  // A(i, k) = Qold[i][k], contiguous in k (row stride ld: odd, scalar loads)
#pragma unroll
    for (int pp = 0; pp < kPF / 4; ++pp) {
      const int i = i0 + (tid >> 4) + 16 * pp, kc = k0 + 4 * (tid & 15);
#pragma unroll
      for (int c = 0; c < 4; ++c) pa[4 * pp + c] = (i < DP && kc + c < DP) ? Qo[(size_t)i * ld + kc + c] : 0.f;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int wi = (w >> 1) * 32, wj = (w & 1) * 32, li = lane & 31, kh = lane >> 5;
  fetch_a(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += kWK) {
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < kPF / 4; ++pp) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = (tid >> 4) + 16 * pp, cc = 4 * (tid & 15) + c;
        sA[cc * kWLd + r] = pa[4 * pp + c];                        // [k][i]
        sB[r * kWLd + cc] = wgen(k0 + r, j0 + cc) * s_sc[cc];      // [k][j], k = r, j = cc
      }
    }
    __syncthreads();
    if (k0 + kWK < kend) fetch_a(k0 + kWK);
#pragma unroll
    for (int u = 0; u < kWK / 2; ++u) {
      const int k = 2 * u + kh;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[k * kWLd + wi + li], sB[k * kWLd + wj + li], acc, 0, 0, 0);
    }
  }
  const int j = j0 + wj + li;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = i0 + wi + acc_row(e, lane);
    if (i < DP && j < DP) Qn[(size_t)i * ld + j] = acc[e];
  }
}

}  // namespace uglad
