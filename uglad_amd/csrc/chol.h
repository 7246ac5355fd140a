// Inverse and log-determinant of a symmetric matrix by blocked factorisation, for the two places where the reference calls an LU-based
// primitive: Theta_0 = torch.inverse(S + t I) (glad.py:115) and torch.logdet(Theta_L) with its backward Theta_L^-T (main.py:307) --
// SURVEY.md section 7, kernels K5 / K6.  Rounds 1-2 ran the path's eigensolver there (0.54 + 0.57 ms per pass at M = 1024, D = 128: its
// latency-bound tridiagonalisation and divide & conquer for a result that needs neither eigenvalues nor eigenvectors).  Here
//   A = L L^T          32 x 32 diagonal blocks by ONE wave in registers (lane = row, pivots and multipliers broadcast with v_readlane),
//                      the block's inverse T = L_jj^-1 right behind it (lane = column); panels L_ij = A_ij T^T and the symmetric
//                      trailing update on v_mfma_f32_32x32x2_f32, one tile per wave;
//   W = L^-1           block forward substitution, tile products on MFMA;
//   X = W^T W          upper tiles on MFMA, mirrored: A^-1, exactly symmetric;
//   logdet A = sum log(pivot).
// chol_inverse_packed (D <= 128, LDS): a pivot that is not > 0 (the matrix is indefinite, singular or holds a NaN) makes it return false;
// the caller then flags the matrix and the eigen path computes it (torch.logdet's NaN / -inf rules live there).
// ldl_inverse (D > 256, workspace slabs: wide_ns.h): the same scheme as A = L D L^T, any signs.
#pragma once
#include "glad_device.h"

namespace uglad {

// ---------------------------------------------------------------------------------------------------------------- packed tiles
// The same factorisation on PACKED tile storage, so that two workgroups share a CU (round 3, late): only the lower tiles of A / L live in
// LDS (slot i (i + 1) / 2 + j for tile (i, j), i >= j, 32 rows of stride 33), the diagonal block's inverse T_jj overwrites A_jj in place
// (L_jj itself is never needed again), and only the off-diagonal tiles of W = L^-1 get storage of their own (slot i (i - 1) / 2 + j,
// i > j): 16 tiles = 67.6 KB at NT = 4 instead of two full matrices = 132 KB.  X = W^T W is formed in registers (at most two tiles per
// wave), written over the dead L tiles (tile (I, J), I <= J, into slot (J, I)) and copied out mirrored -- exactly symmetric.
constexpr int kTS = 33;          // row stride of a packed tile
constexpr int kTF = 32 * kTS;    // floats per packed tile
__host__ __device__ constexpr int chol_lower_tiles(int NT) { return NT * (NT + 1) / 2; }
__host__ __device__ constexpr int chol_offdiag_tiles(int NT) { return NT * (NT - 1) / 2; }
__device__ __forceinline__ int chol_slot(int i, int j) { return i * (i + 1) / 2 + j; }      // i >= j
__device__ __forceinline__ int chol_wslot(int i, int j) { return i * (i - 1) / 2 + j; }     // i > j

// One 32 x 32 diagonal block (packed tile at `a`) by ONE wave: in: A_jj (lower triangle read); out, in place: T = L_jj^-1 (lower
// triangular, zeros above).  Returns false on a pivot that is not > 0.
__device__ __forceinline__ bool chol_diag_tile(float* __restrict__ a, float& logsum, float& pmin, float& pmax) {
  const int lane = threadIdx.x & 63, r = lane & 31;
  float b[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) b[c] = a[r * kTS + c];
  bool ok = true;
  float invd[32];
  float log2sum = 0.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float piv = bcast_lane(b[c], c);
    ok = ok && (piv > 0.f);
    pmin = fminf(pmin, piv);
    pmax = fmaxf(pmax, piv);
    const float inv = __builtin_amdgcn_rsqf(piv), s = piv * inv;
    invd[c] = inv;
    log2sum += __builtin_amdgcn_logf(piv);
    b[c] = (r == c) ? s : ((r > c) ? b[c] * inv : 0.f);
#pragma unroll
    for (int c2 = c + 1; c2 < 32; ++c2) b[c2] = fmaf(-b[c], bcast_lane(b[c], c2), b[c2]);
  }
  float t[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    float acc = (i == r) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < i; ++k) acc = fmaf(-bcast_lane(b[k], i), t[k], acc);
    t[i] = acc * invd[i];
  }
  logsum += 0.69314718056f * log2sum;
  if (lane < 32) {
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i * kTS + r] = t[i];
  }
  return ok;
}

// sP: the lower tiles of A (identity on the padding of the diagonal tiles, zeros elsewhere) -> overwritten; on success the upper tiles of
// X = A^-1 sit in sP (tile (I, J), I <= J, in slot (J, I); chol_packed_at reads entry (i, j) of X).  sQ: chol_offdiag_tiles(NT) tiles of
// scratch.  All kThreads threads call; result, logdet and pivot_ratio are uniform.  s_flag: one int, s_log: three floats of LDS.
template <int NT>
__device__ __forceinline__ bool chol_inverse_packed(float* __restrict__ sP, float* __restrict__ sQ, float& logdet, float& pivot_ratio,
                                                    int* __restrict__ s_flag, float* __restrict__ s_log) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31;
  if (tid == 0) {
    *s_flag = 1;
    s_log[0] = 0.f;
    s_log[1] = 3.0e38f;
    s_log[2] = 0.f;
  }
  __syncthreads();
  auto P = [&](int i, int j) { return sP + chol_slot(i, j) * kTF; };
  auto Q = [&](int i, int j) { return sQ + chol_wslot(i, j) * kTF; };
  auto W = [&](int i, int j) { return (i == j) ? P(i, i) : Q(i, j); };  // W_jj = T_jj lives in the diagonal slot of sP
  auto store_tile = [&](float* __restrict__ X, const f32x16& acc, float scale) {
#pragma unroll
    for (int e = 0; e < 16; ++e) X[acc_row(e, lane) * kTS + li] = scale * acc[e];
  };
  // ---- A = L L^T
#pragma unroll 1
  for (int j = 0; j < NT; ++j) {
    if (w == 0) {
      float ls = 0.f, pmin = s_log[1], pmax = s_log[2];
      const bool ok = chol_diag_tile(P(j, j), ls, pmin, pmax);
      if (lane == 0) {
        if (!ok) *s_flag = 0;
        s_log[0] += ls;
        s_log[1] = pmin;
        s_log[2] = pmax;
      }
    }
    __syncthreads();
    if (*s_flag == 0) return false;  // (uniform)
    for (int i = j + 1 + w; i < NT; i += kWaves) {  // panel: L_ij = A_ij T_jj^T
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_tile(P(i, j), kTS, 1, P(j, j), 1, kTS, 32, acc);  // B[k][c] = T[c][k]
      store_tile(P(i, j), acc, 1.f);                         // (this wave was the only reader of A_ij)
    }
    __syncthreads();
    {  // trailing update: A_ik -= L_ij L_kj^T for j < k <= i
      const int nrem = NT - 1 - j, ntile = nrem * (nrem + 1) / 2;
      for (int t = w; t < ntile; t += kWaves) {
        int a = 0, rem = t;
        while (rem > a) {
          rem -= a + 1;
          ++a;
        }
        const int i = j + 1 + a, k = j + 1 + rem;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mfma_tile(P(i, j), kTS, 1, P(k, j), 1, kTS, 32, acc);  // B[q][c] = L_kj[c][q]
        float* __restrict__ dst = P(i, k);
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[acc_row(e, lane) * kTS + li] -= acc[e];
      }
    }
    __syncthreads();
  }
  logdet = s_log[0];
  pivot_ratio = s_log[2] / s_log[1];
  // ---- W = L^-1, off-diagonal tiles by distance d = i - j:  W_ij = -T_ii sum_{k=j}^{i-1} L_ik W_kj
  constexpr int kPerD = NT > 1 ? (NT - 1 + kWaves - 1) / kWaves : 1;
#pragma unroll 1
  for (int d = 1; d < NT; ++d) {
    f32x16 acc[kPerD];
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      const int j = w + kWaves * n, i = j + d;
      if (i < NT) {
        for (int k = j; k < i; ++k) mfma_tile(P(i, k), kTS, 1, W(k, j), kTS, 1, 32, acc[n]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + kWaves * n, i = j + d;
      if (i < NT) store_tile(Q(i, j), acc[n], 1.f);  // park the sum in W_ij's place (nobody reads W_ij before it is final)
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + kWaves * n, i = j + d;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      if (i < NT) mfma_tile(P(i, i), kTS, 1, Q(i, j), kTS, 1, 32, acc[n]);  // T_ii times the parked sum
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + kWaves * n, i = j + d;
      if (i < NT) store_tile(Q(i, j), acc[n], -1.f);
    }
    __syncthreads();
  }
  // ---- X_IJ = sum_{k >= J} W_kI^T W_kJ on the upper tiles, in registers; then over the dead L tiles
  {
    constexpr int kCount = NT * (NT + 1) / 2, kPer = (kCount + kWaves - 1) / kWaves;
    f32x16 acc[kPer];
#pragma unroll
    for (int n = 0; n < kPer; ++n) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      const int t = w + kWaves * n;
      if (t < kCount) {
        int I, J;
        Tiles<NT, true>::ij(t, I, J);
        for (int k = J; k < NT; ++k) mfma_tile(W(k, I), 1, kTS, W(k, J), kTS, 1, 32, acc[n]);  // A[r][q] = W_kI[q][r]
      }
    }
    __syncthreads();  // every wave is done reading W (the diagonal slots of sP among it)
#pragma unroll
    for (int n = 0; n < kPer; ++n) {
      const int t = w + kWaves * n;
      if (t < kCount) {
        int I, J;
        Tiles<NT, true>::ij(t, I, J);
        store_tile(P(J, I), acc[n], 1.f);
      }
    }
  }
  __syncthreads();
  return true;
}

// entry (i, j) of the symmetric result of chol_inverse_packed (the upper triangle is authoritative: exactly symmetric)
__device__ __forceinline__ float chol_packed_at(const float* __restrict__ sP, int i, int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;  // X[lo][hi], tile (lo >> 5, hi >> 5) in slot (hi >> 5, lo >> 5)
  return sP[chol_slot(hi >> 5, lo >> 5) * kTF + (lo & 31) * kTS + (hi & 31)];
}

// ---------------------------------------------------------------------------------------------------------------- L D L^T
// The same blocked scheme WITHOUT the positivity requirement, for the sizes beyond the eigensolver (wide_ns.h), where no eigen path can
// take over a matrix Cholesky refuses: A = L D L^T with unit lower triangular L and diagonal D of either sign (no pivoting), so that
// log|det A| = sum log|d_i|, sign(det A) = parity of the negative d_i (torch.logdet's NaN for det < 0 comes from that), and
// A^-1 = W^T D^-1 W with W = L^-1 -- also for an indefinite A, e.g. the Theta_L the reference itself produces at D = 512 with parameters
// trained at D = 25 (10 of 512 eigenvalues negative, tests/golden/cell_d512_b1_L15_trained.npz).  For a positive definite matrix this
// is Cholesky's arithmetic up to where the square roots sit.  A pivot that is zero or NaN returns false.
//
// One 32 x 32 diagonal block by one wave: L_jj (unit lower) -> sL, T = L_jj^-1 -> sW, the pivots -> dv[d0 .. d0 + 31].
template <int LD>
__device__ __forceinline__ bool ldl_diag_block(float* __restrict__ sL, float* __restrict__ sW, float* __restrict__ dv, int d0, float& log2sum,
                                               int& negatives) {
  const int lane = threadIdx.x & 63, r = lane & 31;
  float b[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) b[c] = sL[(d0 + r) * LD + d0 + c];
  bool ok = true;
  float mine = 1.f;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float piv = bcast_lane(b[c], c);  // d_c
    ok = ok && (piv != 0.f) && (piv == piv);
    log2sum += __builtin_amdgcn_logf(fabsf(piv));
    negatives += (piv < 0.f) ? 1 : 0;
    if (r == c) mine = piv;
    const float l = (r == c) ? 1.f : ((r > c) ? b[c] / piv : 0.f);  // column c of L
    // a[r][c2] -= l[r][c] d_c l[c2][c], and d_c l[c2][c] is what lane c2 still holds in b[c]
#pragma unroll
    for (int c2 = c + 1; c2 < 32; ++c2) b[c2] = fmaf(-l, bcast_lane(b[c], c2), b[c2]);
    b[c] = l;
  }
  float t[32];  // T = L^-1 (unit lower), lane = column q
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    float acc = (i == r) ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < i; ++k) acc = fmaf(-bcast_lane(b[k], i), t[k], acc);
    t[i] = acc;
  }
  if (lane < 32) {
#pragma unroll
    for (int c = 0; c < 32; ++c) sL[(d0 + r) * LD + d0 + c] = b[c];
#pragma unroll
    for (int i = 0; i < 32; ++i) sW[(d0 + i) * LD + d0 + r] = t[i];
    dv[d0 + r] = mine;
  }
  return ok;
}

// sL: the symmetric matrix (DP x DP, row stride DP + 1, identity on the padding) -> overwritten; sW: scratch of the same size; sX: the
// result A^-1 (both triangles), same layout; dv: DP floats.  logabsdet and `negatives` (number of negative pivots = negative
// eigenvalues) are uniform.  s_flag: one int, s_acc: two floats of LDS.  NW: waves of the workgroup (all of them call).
template <int NT, int NW = kWaves>
__device__ __forceinline__ bool ldl_inverse(float* __restrict__ sL, float* __restrict__ sW, float* __restrict__ sX, float* __restrict__ dv,
                                            float& logabsdet, int& negatives, int* __restrict__ s_flag, float* __restrict__ s_acc, int nt = NT) {
  // nt <= NT: the leading nt x nt tiles hold the matrix (the layout stays that of NT tiles); the padding is never touched
  constexpr int DP = NT * 32, LD = DP + 1;
  const int dp = nt * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int li = lane & 31;
  if (tid == 0) {
    *s_flag = 1;
    s_acc[0] = 0.f;
    s_acc[1] = 0.f;
  }
  for (int idx = tid; idx < dp * dp; idx += (64 * NW)) sW[(idx / dp) * LD + idx % dp] = 0.f;
  __syncthreads();
  auto store_tile = [&](float* __restrict__ X, int I, int J, const f32x16& acc, float scale) {
#pragma unroll
    for (int e = 0; e < 16; ++e) X[(I * 32 + acc_row(e, lane)) * LD + J * 32 + li] = scale * acc[e];
  };
#pragma unroll 1
  for (int j = 0; j < nt; ++j) {
    if (w == 0) {
      float l2 = 0.f;
      int neg = 0;
      const bool ok = ldl_diag_block<LD>(sL, sW, dv, 32 * j, l2, neg);
      if (lane == 0) {
        if (!ok) *s_flag = 0;
        s_acc[0] += l2;
        s_acc[1] += (float)neg;
      }
    }
    __syncthreads();
    if (*s_flag == 0) return false;  // (uniform)
    // panel, i > j:  M_ij = A_ij T_jj^T (= L_ij D_j) -> sW(i, j);  L_ij = M_ij D_j^-1 -> sL(i, j)
    for (int i = j + 1 + w; i < nt; i += NW) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_tile(sL + (32 * i) * LD + 32 * j, LD, 1, sW + (32 * j) * LD + 32 * j, 1, LD, 32, acc);
      const float invd = 1.0f / dv[32 * j + li];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int at = (i * 32 + acc_row(e, lane)) * LD + j * 32 + li;
        sW[at] = acc[e];
        sL[at] = acc[e] * invd;
      }
    }
    __syncthreads();
    {  // trailing update: A_ik -= M_ij L_kj^T for j < k <= i
      const int nrem = nt - 1 - j, ntile = nrem * (nrem + 1) / 2;
      for (int t = w; t < ntile; t += NW) {
        int a = 0, rem = t;
        while (rem > a) {
          rem -= a + 1;
          ++a;
        }
        const int i = j + 1 + a, k = j + 1 + rem;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mfma_tile(sW + (32 * i) * LD + 32 * j, LD, 1, sL + (32 * k) * LD + 32 * j, 1, LD, 32, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) sL[(i * 32 + acc_row(e, lane)) * LD + k * 32 + li] -= acc[e];
      }
    }
    __syncthreads();
  }
  logabsdet = 0.69314718056f * s_acc[0];
  negatives = (int)s_acc[1];
  // ---- W = L^-1 in sW's lower tiles (the parked M_ij are dead; every tile is written before it is read)
  constexpr int kPerD = NT > 1 ? (NT - 1 + NW - 1) / NW : 1;
#pragma unroll 1
  for (int d = 1; d < nt; ++d) {
    f32x16 acc[kPerD];
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      const int j = w + NW * n, i = j + d;
      if (i < nt) {
        for (int k = j; k < i; ++k) mfma_tile(sL + (32 * i) * LD + 32 * k, LD, 1, sW + (32 * k) * LD + 32 * j, LD, 1, 32, acc[n]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + NW * n, i = j + d;
      if (i < nt) store_tile(sW, i, j, acc[n], 1.f);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + NW * n, i = j + d;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
      if (i < nt) mfma_tile(sW + (32 * i) * LD + 32 * i, LD, 1, sW + (32 * i) * LD + 32 * j, LD, 1, 32, acc[n]);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < kPerD; ++n) {
      const int j = w + NW * n, i = j + d;
      if (i < nt) store_tile(sW, i, j, acc[n], -1.f);
    }
    __syncthreads();
  }
  // ---- V = D^-1 W (rows scaled) -> sL, lower tiles incl. the diagonal ones (L is dead)
  for (int idx = tid; idx < dp * dp; idx += (64 * NW)) {
    const int i = idx / dp, k = idx - i * dp;
    if ((k >> 5) <= (i >> 5)) sL[i * LD + k] = sW[i * LD + k] / dv[i];
  }
  __syncthreads();
  // ---- X = W^T V on the upper tiles, mirrored -> sX
  {
    const int ntile = nt * (nt + 1) / 2;
    for (int t = w; t < ntile; t += NW) {
      int I = 0, J = t;  // tile t -> (I <= J) over the nt x nt upper triangle, row by row
      while (J >= nt - I) {
        J -= nt - I;
        ++I;
      }
      J += I;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      for (int k = J; k < nt; ++k) mfma_tile(sW + (32 * k) * LD + 32 * I, 1, LD, sL + (32 * k) * LD + 32 * J, LD, 1, 32, acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = I * 32 + acc_row(e, lane), jj = J * 32 + li;
        if (i <= jj) {
          sX[i * LD + jj] = acc[e];
          sX[jj * LD + i] = acc[e];
        }
      }
    }
  }
  __syncthreads();
  return true;
}

}  // namespace uglad
