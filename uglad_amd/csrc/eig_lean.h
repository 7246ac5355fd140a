// LDS-lean second stage of the eigensolver for matrices that fit the LDS (DP <= 128): ONE DP x (DP+1) matrix in LDS (the
// eigenvectors Q) plus ~8 KB of vectors, so that two workgroups share a CU and hide each other's latency-bound phases
// (secular equation, sorting, the short dependent chains between barriers).  Against eig_dc.h:
//   * the merge matrix W is never materialised: every wave generates the entries of its B operand
//     W'[k][i] = zhat_k / (d_k - lam_i) from four vectors while it issues the MFMAs that consume them; the column norms come
//     out of the same loop (two LDS adds per column: bit-reproducible);
//   * the reflectors stay in global memory (the slab of the output the caller lends, L2-resident) and feed the MFMA A operand
//     directly; the panel Y = V_b Q lives in the accumulator registers of the wave that needs it (the k order of an MFMA chain
//     is free, so an accumulator tile is a valid B operand as it stands) -- no Y panel, no reflector copy in LDS;
//   * Gram matrix and triangular factor of a reflector block are formed by one wave in registers (lane broadcasts instead of
//     LDS reads) and handed to the other waves through the caller's workspace.
#pragma once
#include "eig_dc.h"

namespace uglad {

// Packed upper triangle of a 32 x 32 Gram matrix: row j keeps columns gtri_start(j) .. 31 (the start rounded down to a multiple
// of 4 so that a row is read in 16-byte pieces), rows back to back.
constexpr int kGtriFloats = 544;
__device__ __forceinline__ int gtri_start(int j) { return 4 * ((j + 1) >> 2); }
__device__ __forceinline__ int gtri_off(int j) {  // = sum over j' < j of (32 - gtri_start(j'))
  const int a = j >> 2, r = j & 3;
  return 32 * j - 4 * (2 * a * (a - 1) + a * (r + 1));
}

template <int DP>
struct LeanScratch {
#ifdef UGLAD_STAMPS
  unsigned long long stamp[96];
#endif
  float d[DP], e[DP];                                   // tridiagonal; d ends up holding the eigenvalues (ascending)
  union {
    struct {                                            // divide & conquer
      float ds[DP], zs[DP], zh[DP], mu[DP], dk[DP], lam[DP], inv[DP];
      float dso[DP], invo[DP], nrm2[DP];                // poles / zhat in ORIGINAL column order; squared column norms
      int perm[DP], act[DP];
    };
    alignas(16) float gtri[(DP / 32) * kGtriFloats];    // back-transformation: packed upper rows of the Gram matrices
  };
  float rho[DP / 2 + 1];
  int skip[DP / 2 + 1], fix[DP / 2 + 1], bmax[DP / 2 + 1];
};

__device__ __forceinline__ float lane_xor32(float v) { return __shfl_xor(v, 32); }

// ------------------------------------------------------------------------------------------------ divide & conquer
// As dc_tridiagonal (eig_dc.h) up to the roots and the Gu-Eisenstat vector; the eigenvector update Q <- Q W' diag(1/||.||) then
// generates W' on the fly.
template <int NT>
__device__ __forceinline__ void dc_tridiagonal_lean(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws) {
  constexpr int DP = NT * 32, LD = DP + 1;
  constexpr float kEps = 5.96e-8f;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int idx = tid; idx < DP * DP; idx += kThreads) {
    const int i = idx / DP, j = idx - i * DP;
    Q[i * LD + j] = (i == j) ? 1.f : 0.f;
  }
  __syncthreads();
  if (2 * tid < n) {  // 2 x 2 leaves in closed form (see dc_tridiagonal)
    const int i0 = 2 * tid, i1 = i0 + 1;
    float a = ws.d[i0];
    if (i0 > 0) a -= fabsf(ws.e[i0 - 1]);
    if (i1 < n) {
      float b = ws.d[i1];
      if (i1 < n - 1) b -= fabsf(ws.e[i1]);
      const float c = ws.e[i0];
      float cs = 1.f, sn = 0.f, la = a, lb = b;
      if (c != 0.f) {
        const float th = (b - a) / (2.f * c);
        const float t = ((th >= 0.f) ? 1.f : -1.f) / (fabsf(th) + sqrtf(fmaf(th, th, 1.f)));
        cs = 1.0f / sqrtf(fmaf(t, t, 1.f));
        sn = t * cs;
        la = a - t * c;
        lb = b + t * c;
      }
      const bool sw = la > lb;
      ws.lam[i0] = sw ? lb : la;
      ws.lam[i1] = sw ? la : lb;
      Q[i0 * LD + i0] = sw ? sn : cs;
      Q[i1 * LD + i0] = sw ? cs : -sn;
      Q[i0 * LD + i1] = sw ? cs : sn;
      Q[i1 * LD + i1] = sw ? -sn : cs;
    } else {
      ws.lam[i0] = a;
    }
  }
  __syncthreads();
  if (tid < n) ws.d[tid] = ws.lam[tid];
  if (tid < DP / 2 + 1) {
    ws.bmax[tid] = 0;
    ws.fix[tid] = 0;
  }
  if (tid < DP) {  // the padding: never merged, column stays where it is
    ws.perm[tid] = tid;
    ws.act[tid] = 0;
    ws.dso[tid] = 3.0e38f;  // (a pole at infinity with weight zero: contributes an exact zero to every column)
    ws.invo[tid] = 0.f;
  }
  __syncthreads();

  int lvl = 1;
  for (int h = 2; h < n; h *= 2, ++lvl) {
    const int bs = 2 * h;
    UGLAD_STAMP(ws, 2 + 5 * lvl);
    {  // ---- L1: z, merged order, max |d| per merge  (thread g = original column)
      const int g = tid;
      if (g < n) {
        const int lo = (g / bs) * bs, mid = lo + h;
        const int hi = (lo + bs < n) ? lo + bs : n;
        const float dg = ws.d[g];
        int rank = g - lo;
        float z = 0.f;
        if (mid < n) {
          const float ec = ws.e[mid - 1];
          if (g < mid) {
#pragma unroll 4
            for (int j = mid; j < hi; ++j) rank += (ws.d[j] < dg) ? 1 : 0;
            z = Q[(mid - 1) * LD + g];
          } else {
            rank = g - mid;
#pragma unroll 4
            for (int j = lo; j < mid; ++j) rank += (ws.d[j] <= dg) ? 1 : 0;
            z = (ec >= 0.f) ? Q[mid * LD + g] : -Q[mid * LD + g];
          }
          z *= 0.70710678f;
          if (fabsf(z) < kZFloor) z = (z < 0.f) ? -kZFloor : kZFloor;
          atomicMax(&ws.bmax[g / bs], __float_as_int(fabsf(dg)));
        }
        ws.ds[lo + rank] = dg;
        ws.zs[lo + rank] = z;
        ws.perm[lo + rank] = g;
      }
    }
    __syncthreads();
    {  // ---- L2: coupling test, rz = rho z^2, detection of poles that need separating (p = sorted position)
      const int p = tid;
      if (p < n) {
        const int blk = p / bs, lo = blk * bs, mid = lo + h;
        int skip = 1;
        float rho = 0.f;
        if (mid < n) {
          rho = 2.f * fabsf(ws.e[mid - 1]);
          const float scale = fmaxf(__int_as_float(ws.bmax[blk]), rho);
          if (rho > 8.f * kEps * scale) {
            skip = 0;
            const float z = ws.zs[p];
            ws.zh[p] = rho * z * z;
            if (p > lo) {
              const float cur = ws.ds[p], prev = ws.ds[p - 1];
              const float gap = 4.f * kEps * fmaxf(fabsf(cur), fabsf(prev)) + 1e-10f * scale;
              if (cur < prev + gap) ws.fix[blk] = 1;
            }
          }
        }
        if (p == lo) {
          ws.rho[blk] = rho;
          ws.skip[blk] = skip;
        }
      }
    }
    __syncthreads();
    if (tid * bs < n && ws.fix[tid]) {  // rare: push equal poles a few ulps apart
      const int blo = tid * bs, bhi = (blo + bs < n) ? blo + bs : n;
      const float scale = fmaxf(__int_as_float(ws.bmax[tid]), ws.rho[tid]);
      float prev = ws.ds[blo];
      for (int j = blo + 1; j < bhi; ++j) {
        float cur = ws.ds[j];
        const float gap = 4.f * kEps * fmaxf(fabsf(cur), fabsf(prev)) + 1e-10f * scale;
        if (cur < prev + gap) cur = prev + gap;
        ws.ds[j] = cur;
        prev = cur;
      }
      ws.fix[tid] = 0;
    }
    __syncthreads();
    UGLAD_STAMP(ws, 3 + 5 * lvl);
    // ---- L3: secular roots, LPR lanes per root
    constexpr int LPR = (kThreads / DP >= 4) ? 4 : 2;
    const int p = tid / LPR, sub = tid % LPR;
    int lo = 0, hi = 0;
    bool act = false;
    if (p < n) {
      const int blk = p / bs;
      lo = blk * bs;
      hi = (lo + bs < n) ? lo + bs : n;
      act = (lo + h < n) && (ws.skip[blk] == 0);
    }
    int ta = 0, tbw = 0;
    if (bs * LPR >= 64) {
      const int pw = __builtin_amdgcn_readfirstlane(wv) * (64 / LPR);
      const int imin = pw - (pw / bs) * bs, imax = imin + 64 / LPR - 1;
      ta = (imin + 1) / LPR;
      tbw = (imax + LPR) / LPR;
    }
    if (p < DP) {
      int K = p - lo;
      float mu = 0.f;
      int evals = 0;
      if (act) {
        if (bs <= 2 * LPR)
          evals = secular_root<LPR, 2>(ws.ds + lo, ws.zh + lo, ws.rho[p / bs], hi - lo, p - lo, sub, 0, 0, K, mu);
        else if (bs <= 8 * LPR)
          evals = secular_root<LPR, 8>(ws.ds + lo, ws.zh + lo, ws.rho[p / bs], hi - lo, p - lo, sub, 0, 0, K, mu);
        else
          evals = secular_root<LPR>(ws.ds + lo, ws.zh + lo, ws.rho[p / bs], hi - lo, p - lo, sub, ta, tbw, K, mu);
      }
#ifdef UGLAD_STAMPS
      if (sub == 0 && lvl < 14) {
        atomicMax(reinterpret_cast<int*>(&ws.stamp[80 + lvl]), evals);
        atomicAdd(reinterpret_cast<int*>(&ws.stamp[80 + lvl]) + 1, evals);
      }
#else
      (void)evals;
#endif
      if (sub == 0 && p < n) {
        const float dK = ws.ds[lo + K];
        ws.dk[p] = dK;
        ws.mu[p] = mu;
        ws.lam[p] = dK + mu;
      }
    }
    __syncthreads();
    UGLAD_STAMP(ws, 4 + 5 * lvl);
    {  // ---- L4: Gu-Eisenstat zhat (pole j = p), stored with its pole in the ORIGINAL column order for the GEMM's B operand
      float prod = 1.f;
      if (act) {
        const float dj = ws.ds[p];
#pragma unroll 4
        for (int i = lo + sub; i < hi; i += LPR) {
          const float num = (ws.dk[i] - dj) + ws.mu[i];
          const float den = (i == p) ? 1.f : ws.ds[i] - dj;
          prod *= num * fast_rcp(den);
        }
      }
      prod = group_prod<LPR>(prod);
      if (sub == 0 && p < n) {
        ws.act[p] = act ? 1 : 0;
        ws.nrm2[p] = 0.f;
        if (act) {
          const float zhat = sqrtf(fmaxf(prod, 0.f));
          const int g = ws.perm[p];
          ws.invo[g] = (ws.zs[p] < 0.f) ? -zhat : zhat;
          ws.dso[g] = ws.ds[p];
        }
      }
    }
    __syncthreads();
    UGLAD_STAMP(ws, 5 + 5 * lvl);
    {  // ---- L7: Q <- (Q W') diag(1/||W'_i||) on the diagonal blocks of size tb, W' generated per lane
      const int tb = (bs > 32) ? bs : 32;
      const int TB = tb / 32;
      const int TBe = TB < NT ? TB : NT;
      const int ntile = NT * TBe;
      constexpr int kTPW = (NT * NT + kWaves - 1) / kWaves;
      const int li = lane & 31, kh = lane >> 5;
      f32x16 acc[kTPW];
#pragma unroll
      for (int s = 0; s < kTPW; ++s) {
        const int t = wv + kWaves * s;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        if (t < ntile) {
          const int I = t / TBe, J = (I / TB) * TB + (t - I * TBe);
          if (J < NT) {
            const int hh = (h > 32) ? h : 32;
            const int kb = (I * 32 / hh) * hh;
            int kend = kb + hh;
            if (kend > DP) kend = DP;
            const int col = J * 32 + li;
            const float dki = ws.dk[col], mui = ws.mu[col];
            const bool acti = ws.act[col] != 0;
            const int permi = ws.perm[col];
            const int blo = (col / bs) * bs;  // (only needed for merges smaller than a tile)
            const float* a = Q + (I * 32 + li) * LD + kb + kh;
            float s2 = 0.f;
            float av[8], dv[8], iv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              av[u] = a[2 * u];
              dv[u] = ws.dso[kb + 2 * u + kh];
              iv[u] = ws.invo[kb + 2 * u + kh];
            }
            for (int k0 = kb; k0 < kend; k0 += 16) {
              const int kn = (k0 + 16 < kend) ? k0 + 16 : k0;  // (the last chunk re-reads itself: no branch around the loads)
              float an[8], dn[8], in_[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                an[u] = a[(kn - kb) + 2 * u];
                dn[u] = ws.dso[kn + 2 * u + kh];
                in_[u] = ws.invo[kn + 2 * u + kh];
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int k = k0 + 2 * u + kh;
                float val = acti ? iv[u] * fast_rcp((dv[u] - dki) - mui) : ((k == permi) ? 1.f : 0.f);
                if (bs < 32 && (unsigned)(k - blo) >= (unsigned)bs) val = 0.f;
                s2 = fmaf(val, val, s2);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], val, acc[s], 0, 0, 0);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                av[u] = an[u];
                dv[u] = dn[u];
                iv[u] = in_[u];
              }
            }
            s2 += lane_xor32(s2);
            if (I * 32 == kb && kh == 0 && acti) atomicAdd(&ws.nrm2[col], s2);  // one tile per k range contributes
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < kTPW; ++s) {
        const int t = wv + kWaves * s;
        if (t < ntile) {
          const int I = t / TBe, J = (I / TB) * TB + (t - I * TBe);
          if (J < NT) {
            const int col = J * 32 + li;
            const float sc = (ws.act[col] != 0) ? 1.0f / sqrtf(ws.nrm2[col]) : 1.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) Q[(I * 32 + acc_row(e, lane)) * LD + col] = acc[s][e] * sc;
          }
        }
      }
      if (tid < n) ws.d[tid] = ws.lam[tid];
      if (tid < DP / 2 + 1) ws.bmax[tid] = 0;
    }
    __syncthreads();
    UGLAD_STAMP(ws, 6 + 5 * lvl);
  }
}

// ------------------------------------------------------------------------------------------------ back-transformation
// Four consecutive entries c .. c+3 of reflector row k (global memory; zero outside the matrix / beyond the last reflector).
__device__ __forceinline__ f4 load_reflector4(const float* __restrict__ R, int ldr, int k, int c, int nr, int n, bool vec) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (k < nr) {
    const float* p = R + (size_t)k * ldr + c;
    if (vec) {
      if (c < n) v = *reinterpret_cast<const f4*>(p);  // (n and ldr are multiples of 4 here: all four or none)
    } else {
      if (c < n) v.x = p[0];
      if (c + 1 < n) v.y = p[1];
      if (c + 2 < n) v.z = p[2];
      if (c + 3 < n) v.w = p[3];
    }
  }
  return v;
}

// Rows of an accumulator tile held by (register e, lane half h): the k index a lane half supplies in MFMA step e when the tile
// is fed back as a B operand (the A operand follows the same order).
__device__ __forceinline__ int acc_k(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// Q <- H_0 ... H_{n-3} Q, blocks of 32 reflectors, compact WY.  R: reflector rows in global memory (stride ldr), tau: their
// scalars (global), Tws: NT x 1024 floats of global scratch for the triangular factors.
template <int NT>
__device__ __forceinline__ void back_transform_lean(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws,
                                                    const float* __restrict__ R, int ldr, const float* __restrict__ tau,
                                                    float* __restrict__ Tws) {
  constexpr int DP = NT * 32, LD = DP + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, kh = lane >> 5;
  const int nr = n - 2;
  if (nr <= 0) return;
  const int nblk = (nr + 31) / 32;
  const bool vec = ((ldr & 3) == 0) && ((n & 3) == 0) && ((reinterpret_cast<size_t>(R) & 15) == 0);
  UGLAD_STAMP(ws, 42);
  // ---- Gram matrices G_b = V_b V_b^T, one wave per block: A[r][c] and B[c][r'] are the same numbers for r = r' = lane & 31, so
  // one fragment feeds both operands (k order: lane half h of chunk q supplies columns kb + 8 q + 4 h + {0, 1, 2, 3}).  The
  // upper rows go to LDS, packed (over the divide & conquer's vectors, which are dead by now).
  for (int b = wv; b < nblk; b += kWaves) {
    const int kb = 32 * b;
    f32x16 g;
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = 0.f;
    for (int c0 = kb; c0 < DP; c0 += 8) {
      const f4 x = load_reflector4(R, ldr, kb + li, c0 + 4 * kh, nr, n, vec);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, x.x, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, x.y, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, x.z, g, 0, 0, 0);
      g = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, x.w, g, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int j = acc_row(e, lane);
      if (li >= gtri_start(j)) ws.gtri[b * kGtriFloats + gtri_off(j) + (li - gtri_start(j))] = g[e];
    }
  }
  if (tid < DP) ws.e[tid] = (tid < nr) ? tau[tid] : 0.f;  // (e is dead after the divide & conquer: now the reflector scalars)
  __syncthreads();
  UGLAD_STAMP(ws, 44);
  // ---- T_b = (triu(G_b, 1) + diag(1 / tau))^-1 by back substitution, one column per thread, all blocks at once -> workspace
  if (tid < 32 * nblk) {
    const int b = tid >> 5, c = tid & 31;
    const float* G = ws.gtri + b * kGtriFloats;
    float y[32];
#pragma unroll
    for (int j = 31; j >= 0; --j) {
      float a0 = (j == c) ? 1.f : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four chains instead of one of length 31
#pragma unroll
      for (int q = (j + 1) / 4; q < 8; ++q) {
        const f4 g4 = *reinterpret_cast<const f4*>(&G[gtri_off(j) + 4 * q - gtri_start(j)]);
        if (4 * q + 0 > j) a0 = fmaf(-g4.x, y[4 * q + 0], a0);
        if (4 * q + 1 > j) a1 = fmaf(-g4.y, y[4 * q + 1], a1);
        if (4 * q + 2 > j) a2 = fmaf(-g4.z, y[4 * q + 2], a2);
        if (4 * q + 3 > j) a3 = fmaf(-g4.w, y[4 * q + 3], a3);
      }
      y[j] = ws.e[32 * b + j] * ((a0 + a1) + (a2 + a3));
      __builtin_amdgcn_sched_barrier(0);  // (keeps the loads of later rows from piling up in registers: 128 are all there is)
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) Tws[b * 1024 + j * 32 + c] = y[j];
  }
  __syncthreads();
  UGLAD_STAMP(ws, 45);
  // ---- the blocks, last to first.  Wave (J, sub): column tile J of Q; the row tiles of the update are dealt over the waves
  // that share J.
  constexpr int WPJ = kWaves / NT;
  const int J = wv % NT, sub = wv / NT;
  const bool active = wv < NT * WPJ;
  const int colj = J * 32 + li;
  for (int b = nblk - 1; b >= 0; --b) {
    const int kb = 32 * b;
    f32x16 yv;
    if (active) {
      // Y = V_b Q (rows < kb of Q do not contribute); A from global, chunk q: columns kb + 8 q + 4 h + s
#pragma unroll
      for (int e = 0; e < 16; ++e) yv[e] = 0.f;
      f4 x = load_reflector4(R, ldr, kb + li, kb + 4 * kh, nr, n, vec);
      for (int c0 = kb; c0 < DP; c0 += 8) {
        const int cn = (c0 + 8 < DP) ? c0 + 8 : c0;
        const f4 xn = load_reflector4(R, ldr, kb + li, cn + 4 * kh, nr, n, vec);
        const float* qb = Q + (c0 + 4 * kh) * LD + colj;
        const float q0 = qb[0], q1 = qb[LD], q2 = qb[2 * LD], q3 = qb[3 * LD];
        yv = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, q0, yv, 0, 0, 0);
        yv = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, q1, yv, 0, 0, 0);
        yv = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, q2, yv, 0, 0, 0);
        yv = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, q3, yv, 0, 0, 0);
        x = xn;
      }
      // Y <- T_b Y: B = the accumulator tile itself (k order acc_k), A[r'][k] = T_b[r'][k] from the workspace
      f32x16 y2;
#pragma unroll
      for (int e = 0; e < 16; ++e) y2[e] = 0.f;
      const float* Tb = Tws + b * 1024 + li * 32;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f4 t4 = *reinterpret_cast<const f4*>(Tb + 8 * m + 4 * kh);  // k = acc_k(4 m + s, h) = 8 m + 4 h + s
        y2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t4.x, yv[4 * m + 0], y2, 0, 0, 0);
        y2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t4.y, yv[4 * m + 1], y2, 0, 0, 0);
        y2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t4.z, yv[4 * m + 2], y2, 0, 0, 0);
        y2 = __builtin_amdgcn_mfma_f32_32x32x2f32(t4.w, yv[4 * m + 3], y2, 0, 0, 0);
      }
      yv = y2;
    }
    __syncthreads();  // every wave has read its column of Q
    UGLAD_STAMP(ws, 46 + 4 * b);
    if (active) {
      // Q(I, J) -= V_b(:, I)^T Y for the row tiles I >= b of this wave; A[i][r] = V_b[r][32 I + i], r = acc_k(e, h)
      for (int I = b + sub; I < NT; I += WPJ) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const int c = I * 32 + li;
        float av[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = kb + acc_k(e, kh);
          av[e] = (k < nr && c < n) ? R[(size_t)k * ldr + c] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], yv[e], acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) Q[(I * 32 + acc_row(e, lane)) * LD + colj] -= acc[e];
      }
    }
    __syncthreads();
    UGLAD_STAMP(ws, 48 + 4 * b);
  }
}

// Tridiagonal form (d, e, tau at `tri`), reflectors (rows of R) from tridiag_kernel -> eigenvalues in ws.d (ascending),
// eigenvectors in Q (stride DP + 1; identity on the padding).
template <int NT>
__device__ __forceinline__ void symeig_lean(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws,
                                            const float* __restrict__ tri, const float* __restrict__ R, int ldr,
                                            float* __restrict__ Tws) {
  constexpr int DP = NT * 32;
  for (int i = threadIdx.x; i < DP; i += kThreads) {
    ws.d[i] = (i < n) ? tri[i] : 0.f;
    ws.e[i] = (i < n) ? tri[DP + i] : 0.f;
  }
  __syncthreads();
  UGLAD_STAMP(ws, 1);
  dc_tridiagonal_lean<NT>(Q, n, ws);
  UGLAD_STAMP(ws, 40);
  back_transform_lean<NT>(Q, n, ws, R, ldr, tri + 2 * DP, Tws);
  UGLAD_STAMP(ws, 41);
}

}  // namespace uglad
