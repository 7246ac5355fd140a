// Batched symmetric eigensolver for one LDS-resident matrix per workgroup (gfx950, 512 threads = 8 waves):
//
//   1. Householder tridiagonalisation  A = H T H^T in its own kernel (tridiag.h): the matrix lives in REGISTERS, so that kernel
//      needs little LDS and four workgroups share a CU, hiding each other's serial reflector chains;
//   2. divide & conquer on T (Cuppen tearing down to 2x2 leaves solved in closed form, log2 n - 1 merge levels): per merge a
//      secular equation per eigenvalue (four lanes each, LAPACK-style starting point, "middle way" rational iteration with
//      bracketing, origin shifted to the nearest pole so all differences are relatively accurate), Gu-Eisenstat re-derivation
//      of z for orthogonality, and the eigenvector update Q <- Q W as block-diagonal GEMMs on the f32 MFMA.  Instead of
//      LAPACK's deflation (data-dependent control flow) equal poles are separated by a few ulps and vanishing z components
//      are floored at 1e-6: a backward error of O(eps ||T||) that keeps every lane on the same code path.  A merge whose
//      coupling is below 8 eps ||.|| is skipped (sorted only);
//   3. back-transformation Q <- H Q in blocks of 32 reflectors (compact WY): all reflectors come back into LDS at once, the
//      Gram matrices and triangular factors of all blocks are formed side by side, then V Q, T (V Q) and the rank-32 update
//      run on the MFMA block by block.
//
// Reflectors are parked in a caller-provided global scratch row by row while the LDS is needed for step 2 (the cell kernel
// lends the slab of its own output, which is not written before step 3 has finished).
#pragma once
#include "glad_device.h"
#include "tridiag.h"

namespace uglad {

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Components of the coupling vector z below this are floored instead of deflated (see the header of this file).
#ifndef UGLAD_ZFLOOR
#define UGLAD_ZFLOOR 1e-6f
#endif
constexpr float kZFloor = UGLAD_ZFLOOR;

// Diagnostic build only (-DUGLAD_STAMPS, scripts/stamp_symeig.py): shader-clock stamps at phase boundaries.
// -DUGLAD_PHASE_EXIT (scripts/phase_exit_probe.py): every wave ENDS at the boundary g_exit_at names, so that hardware counters of
// launches cut at successive boundaries give per-phase differences (the outputs of such a launch are garbage).
#if defined(UGLAD_PHASE_EXIT)
__device__ int g_exit_at;
#define UGLAD_STAMP(ws, i)                                  \
  do {                                                      \
    if (g_exit_at == (i)) __builtin_amdgcn_endpgm();        \
  } while (0)
#elif defined(UGLAD_STAMPS)
#define UGLAD_STAMP(ws, i)                                                        \
  do {                                                                            \
    if (threadIdx.x == 0 && (i) < 64) (ws).stamp[i] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define UGLAD_STAMP(ws, i) \
  do {                     \
  } while (0)
#endif

// Scratch in LDS besides the two big matrices.
template <int DP>
struct EigScratch {
#ifdef UGLAD_STAMPS
  unsigned long long stamp[96];  // [0, 64): shader clock at phase boundaries; [80 + level]: (sum << 32 | max) of secular evaluations
#endif
  float d[DP], e[DP], tau[DP];      // tridiagonal + reflector scalars; d ends up holding the eigenvalues (ascending)
  float ds[DP], zs[DP], zh[DP], mu[DP], inv[DP], lam[DP], dk[DP], nrm[DP];
  int perm[DP];
  float rho[DP / 2 + 1];
  int skip[DP / 2 + 1], fix[DP / 2 + 1], bmax[DP / 2 + 1];
  alignas(16) float Y[32 * (DP + 1)];  // back-transformation panel (first: the Gram matrices of all reflector blocks)
  float t0[32 * 33];       // triangular factor of reflector block 0
};

// ------------------------------------------------------------------------------------------------ 2. divide & conquer
// Secular equation 1 + sum_j rz[j] / (ds[j] - x) = 0 (rz = rho z^2, strictly increasing poles ds[0..nb)).  Two adjacent lanes
// (sub = 0/1) share root i: each sums every other pole and the pair combines with one xor-shuffle, so both lanes carry
// bitwise identical iterates and leave the loop together.  Returns the origin pole K and mu with x = ds[K] + mu.
// LPR adjacent lanes share one root / pole / column of the merge step.
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  v += lane_xor1(v);
  if (LPR == 4) v += dpp_move<0x4e>(v);  // quad_perm:[2,3,0,1]
  return v;
}
template <int LPR>
__device__ __forceinline__ float group_prod(float v) {
  v *= lane_xor1(v);
  if (LPR == 4) v *= dpp_move<0x4e>(v);
  return v;
}

// Poles t < ta (in units of LPR poles) lie left of the root for EVERY lane of the wave, poles t >= tb right of it; only the
// few in between need the per-lane test.  (ta = tb = 0 when a wave's roots belong to several merges: per-lane test everywhere.)
// NP > 0: small merge (nb <= LPR * NP): the lane's poles live in registers for the whole solve -- no LDS traffic and no loops
// in the iteration, whose cost at the small merge levels is all overhead.
template <int LPR, int NP = 0>
__device__ __forceinline__ int secular_root(const float* __restrict__ ds, const float* __restrict__ rz, float rho, int nb,
                                             int i, int sub, int ta, int tb, int& Kout, float& mu_out) {
  constexpr float kEps = 5.96e-8f;
  constexpr int NR = NP > 0 ? NP : 1;
  float pd[NR], pr[NR], pl[NR];  // pole, weight, 1 if the pole lies left of the root (j <= i) else 0; absent poles: weight 0
  if (NP > 0) {
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int j = sub + LPR * t;
      const bool ok = j < nb;
      pd[t] = ok ? ds[j] : 3.0e38f;
      pr[t] = ok ? rz[j] : 0.f;
      pl[t] = (ok && j <= i) ? 1.f : 0.f;
    }
  }
  if (ta < tb) nb = __builtin_amdgcn_readfirstlane(nb);  // one merge per wave: the pole count is wave-uniform
  const int tfull = nb / LPR;  // poles j = sub + LPR t with t < tfull exist for every sub
  tb = (tb > 0 && tb < tfull) ? tb : tfull;
  ta = (ta < tb) ? ta : tb;
  // Starting point (as in LAPACK's slaed4): evaluate at a test point -- the midpoint of the two neighbouring poles, or half
  // the upper bound rho for the last root -- then keep the two nearest poles exact and freeze the rest there:
  // rest (d1-x)(d2-x) + p (d2-x) + q (d1-x) = 0.
  const bool last = i == nb - 1;
  const int ia = last ? nb - 2 : i;  // the two nearest poles are ia, ia + 1
  const float hi_last = rho * 1.00001f + 1e-30f;
  const float dorg = ds[i];
  float test = 0.5f * (ds[last ? i : i + 1] - dorg);
  if (last) {
    // test point for the last root: the root of the two-pole equation 1 + p/(d1 - x) + q/(0 - x) = 0 (all other poles
    // ignored) instead of LAPACK's rho / 2 -- it lies close to the root, so freezing the far poles there costs one
    // iteration less on the root that otherwise keeps its whole wave waiting
    const float d1l = ds[nb - 2] - dorg, pl = rz[nb - 2], ql = rz[nb - 1];
    const float bl = d1l + pl + ql, cl = ql * d1l;
    const float x0 = 0.5f * (bl + __builtin_amdgcn_sqrtf(fmaxf(bl * bl - 4.f * cl, 0.f)));
    test = (x0 > 0.f && x0 < hi_last) ? x0 : 0.5f * hi_last;
  }
  float wsum = 0.f;
  if (NP > 0) {
#pragma unroll
    for (int t = 0; t < NR; ++t) wsum = fmaf(pr[t], fast_rcp((pd[t] - dorg) - test), wsum);
  } else {
#pragma unroll 4
    for (int j = sub; j < nb; j += LPR) wsum = fmaf(rz[j], fast_rcp((ds[j] - dorg) - test), wsum);
  }
  const float wt = 1.f + group_sum<LPR>(wsum);
  const int K = (last || wt > 0.f) ? i : i + 1;  // origin: the pole nearest to the root
  const float dK = ds[K];
  const float d1 = ds[ia] - dK, d2 = ds[ia + 1] - dK;
  const float p = rz[ia], q = rz[ia + 1];
  const float xt = (dorg - dK) + test;  // the test point relative to the origin
  const float rest = wt - p * fast_rcp(d1 - xt) - q * fast_rcp(d2 - xt);
  float lo, hi;
  if (last) {
    lo = (wt < 0.f) ? test : 0.f;
    hi = (wt < 0.f) ? hi_last : test;
  } else {
    lo = (K == i) ? 0.f : -test;
    hi = (K == i) ? test : 0.f;
  }
  const float bq = rest * (d1 + d2) + p + q;
  const float cq = rest * d1 * d2 + p * d2 + q * d1;
  const float sq = __builtin_amdgcn_sqrtf(fmaxf(bq * bq - 4.f * rest * cq, 0.f));
  float mu;
  if (!last)  // an interior root: the root of small magnitude of the model's quadratic, whichever of the two poles is the origin (eig_lean.h)
    mu = (bq > 0.f) ? 2.f * cq * fast_rcp(bq + sq) : (bq - sq) * fast_rcp(2.f * rest);
  else  // the last root lies above both poles: the other one
    mu = (bq < 0.f) ? 2.f * cq * fast_rcp(bq - sq) : (bq + sq) * fast_rcp(2.f * rest);
  if (!(mu > lo && mu < hi)) mu = 0.5f * (lo + hi);
  const int jl = i, jr = i + 1;
  const float dl1 = ds[jl] - dK, dl2 = (jr < nb) ? ds[jr] - dK : 0.f;
  // The step only has to be good enough to converge (the test on w decides when to stop), so its divisions and the square
  // root are the one-ulp hardware approximations.
  int it = 0;
  for (; it < 48; ++it) {
    float psi = 0.f, dpsi = 0.f, phi = 0.f, dphi = 0.f;
    if (NP > 0) {
#pragma unroll
      for (int t = 0; t < NR; ++t) {
        const float r = fast_rcp((pd[t] - dK) - mu);
        const float term = pr[t] * r;
        const float tr = term * r;
        const float tl = term * pl[t], trl = tr * pl[t];
        psi += tl;
        dpsi += trl;
        phi += term - tl;
        dphi += tr - trl;
      }
    } else {
#pragma unroll 4
      for (int t = 0; t < ta; ++t) {
        const int j = sub + LPR * t;
        const float r = fast_rcp((ds[j] - dK) - mu);
        const float term = rz[j] * r;
        psi += term;
        dpsi = fmaf(term, r, dpsi);
      }
      for (int t = ta; t < tb; ++t) {
        const int j = sub + LPR * t;
        const float r = fast_rcp((ds[j] - dK) - mu);
        const float term = rz[j] * r;
        const float tr = term * r;
        const bool left = j <= jl;
        psi += left ? term : 0.f;
        dpsi += left ? tr : 0.f;
        phi += left ? 0.f : term;
        dphi += left ? 0.f : tr;
      }
#pragma unroll 4
      for (int t = tb; t < tfull; ++t) {
        const int j = sub + LPR * t;
        const float r = fast_rcp((ds[j] - dK) - mu);
        const float term = rz[j] * r;
        phi += term;
        dphi = fmaf(term, r, dphi);
      }
      {
        const int j = sub + LPR * tfull;  // the ragged tail
        if (j < nb) {
          const float r = fast_rcp((ds[j] - dK) - mu);
          const float term = rz[j] * r;
          const float tr = term * r;
          const bool left = j <= jl;
          psi += left ? term : 0.f;
          dpsi += left ? tr : 0.f;
          phi += left ? 0.f : term;
          dphi += left ? 0.f : tr;
        }
      }
    }
    psi = group_sum<LPR>(psi);
    dpsi = group_sum<LPR>(dpsi);
    phi = group_sum<LPR>(phi);
    dphi = group_sum<LPR>(dphi);
    const float D1 = dl1 - mu, D2 = dl2 - mu;
    const float w = 1.f + psi + phi;
    if (fabsf(w) <= 8.f * kEps * (1.f + fabsf(psi) + fabsf(phi))) break;
    if (w < 0.f) lo = mu; else hi = mu;
    // step of the rational model, written without branches: eta = num / den with the numerically safe root of
    // a eta^2 - b eta + g = 0 between two poles, or the one-pole formula for the last root
    const float dsum = dpsi + dphi;
    const float a = w - D1 * dpsi - D2 * dphi;
    const float b = (D1 + D2) * w - D1 * D2 * dsum;
    const float g = D1 * D2 * w;
    const float sq = __builtin_amdgcn_sqrtf(fabsf(fmaf(b, b, -4.f * a * g)));
    const bool bneg = b <= 0.f, a0 = a == 0.f;
    float num = bneg ? (a0 ? g : b - sq) : 2.f * g;
    float den = bneg ? (a0 ? b : 2.f * a) : b + sq;
    float add = 0.f;
    if (jr >= nb) {  // (per lane: only the last root of a merge)
      const float c = w - dpsi * D1;
      num = (c != 0.f) ? dpsi * D1 * D1 : 0.f;
      den = (c != 0.f) ? c : 1.f;
      add = (c != 0.f) ? D1 : 0.f;
    }
    float eta = fmaf(num, fast_rcp(den), add);
    const float newton = -w * fast_rcp(dsum);
    if (!(fabsf(eta) < 3.0e38f) || w * eta >= 0.f) eta = newton;
    float nw = mu + eta;
    if (!(nw > lo && nw < hi)) nw = 0.5f * (lo + hi);
    if (nw == mu) break;
    mu = nw;
  }
  Kout = K;
  mu_out = mu;
  return it + 1;  // evaluations of the secular function
}

// The reflectors (rows of R in global memory) on their way into Vt for the back-transformation: the loads are issued before
// the last merge of the D&C and land in registers while its GEMM runs (DP <= 128; beyond that the registers do not suffice
// and the copy is done in place).
template <int DP>
struct ReflectorFetch {
  static constexpr bool kEarly = DP <= 128;
  static constexpr int kPer = kEarly ? (DP * DP + kThreads - 1) / kThreads : 1;
  float v[kPer];
  const float* R;
  int ldr, n;
  __device__ __forceinline__ void issue() {
    if (!kEarly) return;
    const int nr = n - 2;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
      const int idx = threadIdx.x + kThreads * i, k = idx / DP, c = idx - k * DP;
      v[i] = (k < nr && c < n) ? R[(size_t)k * ldr + c] : 0.f;
    }
  }
  __device__ __forceinline__ void land(float* __restrict__ Vt) const {
    constexpr int LD = DP + 1;
    if (kEarly) {
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int idx = threadIdx.x + kThreads * i, k = idx / DP, c = idx - k * DP;
        if (idx < DP * DP) Vt[k * LD + c] = v[i];
      }
    } else {
      const int nr = n - 2;
      for (int idx = threadIdx.x; idx < DP * DP; idx += kThreads) {
        const int k = idx / DP, c = idx - k * DP;
        Vt[k * LD + c] = (k < nr && c < n) ? R[(size_t)k * ldr + c] : 0.f;
      }
    }
  }
};

// Eigen-decomposition of the tridiagonal (ws.d, ws.e) of order n.  Q (DP x DP, stride LD = DP+1) receives the eigenvectors,
// W (same shape) is workspace.  ws.d returns the eigenvalues in ascending order.
// Thread layout from the secular step on: position p = tid >> 1 of the merged (sorted) order, sub = tid & 1.
template <int NT>
__device__ __forceinline__ void dc_tridiagonal(float* __restrict__ W, float* __restrict__ Q, int n, EigScratch<NT * 32>& ws,
                                               ReflectorFetch<NT * 32>& fetch) {
  constexpr int DP = NT * 32, LD = DP + 1;
  constexpr float kEps = 5.96e-8f;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int idx = tid; idx < DP * DP; idx += kThreads) {
    const int i = idx / DP, j = idx - i * DP;
    Q[i * LD + j] = (i == j) ? 1.f : 0.f;
  }
  __syncthreads();
  // Leaves are 2 x 2 (the last one 1 x 1 when n is odd), solved in closed form; the boundaries BETWEEN leaves are torn:
  // the rows on either side of boundary (2i+1 | 2i+2) give up |e_{2i+1}|.
  if (2 * tid < n) {
    const int i0 = 2 * tid, i1 = i0 + 1;
    float a = ws.d[i0];
    if (i0 > 0) a -= fabsf(ws.e[i0 - 1]);
    if (i1 < n) {
      float b = ws.d[i1];
      if (i1 < n - 1) b -= fabsf(ws.e[i1]);
      const float c = ws.e[i0];
      float cs = 1.f, sn = 0.f, la = a, lb = b;
      if (c != 0.f) {  // Jacobi rotation of [[a, c], [c, b]]
        const float th = (b - a) / (2.f * c);
        const float t = ((th >= 0.f) ? 1.f : -1.f) / (fabsf(th) + sqrtf(fmaf(th, th, 1.f)));
        cs = 1.0f / sqrtf(fmaf(t, t, 1.f));
        sn = t * cs;
        la = a - t * c;
        lb = b + t * c;
      }
      // eigenvectors (cs, -sn) for la and (sn, cs) for lb; ascending order
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
  __syncthreads();

  int lvl = 1;
  for (int h = 2; h < n; h *= 2, ++lvl) {
    const int bs = 2 * h;
    UGLAD_STAMP(ws, 2 + 5 * lvl);
    // ---- L1: z, merged order, max |d| per merge  (thread g = original column)
    {
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
          if (fabsf(z) < kZFloor) z = (z < 0.f) ? -kZFloor : kZFloor;  // floor instead of deflating
          atomicMax(&ws.bmax[g / bs], __float_as_int(fabsf(dg)));
        }
        ws.ds[lo + rank] = dg;
        ws.zs[lo + rank] = z;
        ws.perm[lo + rank] = g;
      }
    }
    __syncthreads();
    // ---- L2: coupling test, rz = rho z^2, detection of poles that need separating (p = sorted position)
    {
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
    // rare: one lane per merge walks its poles and pushes equal ones a few ulps apart
    if (tid * bs < n && ws.fix[tid]) {
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
    // lanes per root.  (Measured: 2 lanes per root -- half the waves, twice the poles per lane -- is 1.8x slower at the upper
    // levels: a round is bound by the serial work of one lane, not by the issue slots of the SIMD.)
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
    int ta = 0, tbw = 0;  // ta == tbw == 0: no wave-uniform split, every pole takes the per-lane test
    if (bs * LPR >= 64) {  // all roots of this wave belong to one merge: wave-uniform split of its poles into left / right
      const int pw = __builtin_amdgcn_readfirstlane(wv) * (64 / LPR);
      const int imin = pw - (pw / bs) * bs, imax = imin + 64 / LPR - 1;
      ta = (imin + 1) / LPR;
      tbw = (imax + LPR) / LPR;
    }
    if (p < DP) {  // (whole lane groups take the same branch)
      int K = p - lo;
      float mu = 0.f;
      int evals = 0;
      if (act) {
        if (bs <= 2 * LPR)  // (wave-uniform) small merges: poles in registers
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
    // ---- L4: Gu-Eisenstat z (pole j = p):  zhat_j^2 = (lam_j - d_j) prod_{i != j} (lam_i - d_j)/(d_i - d_j); the common factor
    //          1/rho drops out when the eigenvectors are normalised
    {
      float prod = 1.f;
      if (act) {
        const float dj = ws.ds[p];
#pragma unroll 4
        for (int i = lo + sub; i < hi; i += LPR) {
          const float num = (ws.dk[i] - dj) + ws.mu[i];  // lam_i - d_j
          const float den = (i == p) ? 1.f : ws.ds[i] - dj;
          prod *= num * fast_rcp(den);
        }
      }
      prod = group_prod<LPR>(prod);
      if (act && sub == 0) {
        const float zhat = sqrtf(fmaxf(prod, 0.f));
        ws.inv[p] = (ws.zs[p] < 0.f) ? -zhat : zhat;
      }
    }
    // diagonal blocks of W that the GEMM will read: zero them (only entries of merged poles are rewritten below)
    const int tb = (bs > 32) ? bs : 32;
    for (int idx = tid; idx < DP * tb; idx += kThreads) {
      const int col = idx / tb, rr = idx - col * tb;
      const int row0 = (col / tb) * tb;
      if (row0 + rr < DP) W[(row0 + rr) * LD + col] = 0.f;
    }
    __syncthreads();
    // ---- L5: column i = p of W':  W'[perm[j]][i] = zhat_j / (d_j - lam_i), left un-normalised; 1/||.|| goes into ws.nrm
    {
      float s = 0.f;
      if (act) {
        const float dK = ws.dk[p], mu = ws.mu[p];
#pragma unroll 4
        for (int j = lo + sub; j < hi; j += LPR) {
          const float t = ws.inv[j] * fast_rcp((ws.ds[j] - dK) - mu);
          s = fmaf(t, t, s);
          W[ws.perm[j] * LD + p] = t;
        }
      }
      s = group_sum<LPR>(s);
      if (sub == 0 && p < n) {
        ws.nrm[p] = act ? 1.0f / sqrtf(s) : 1.f;
        if (!act) W[ws.perm[p] * LD + p] = 1.f;  // nothing merged here: the column only moves to its sorted position
      }
    }
    for (int c = n + tid; c < DP; c += kThreads) {
      W[c * LD + c] = 1.f;
      ws.nrm[c] = 1.f;
    }
    __syncthreads();
    UGLAD_STAMP(ws, 5 + 5 * lvl);
    // ---- L7: Q <- (Q W') diag(nrm) on the diagonal blocks of size tb
    if (bs >= n) fetch.issue();
    {
      const int TB = tb / 32;
      const int TBe = TB < NT ? TB : NT;  // tiles per row that exist (the last merge may be wider than the padded matrix)
      const int ntile = NT * TBe;         // tiles (I, J) with I/TB == J/TB
      constexpr int kTPW = (NT * NT + kWaves - 1) / kWaves;
      f32x16 acc[kTPW];
#pragma unroll
      for (int s = 0; s < kTPW; ++s) {
        const int t = wv + kWaves * s;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        if (t < ntile) {
          const int I = t / TBe, J = (I / TB) * TB + (t - I * TBe);
          // rows of tile I are nonzero only inside the diagonal block (size h, the merge's input) that contains them
          const int hh = (h > 32) ? h : 32;
          const int kb = (I * 32 / hh) * hh;
          int kend = kb + hh;
          if (kend > DP) kend = DP;
          if (J < NT) mfma_tile(Q + (I * 32) * LD + kb, LD, 1, W + kb * LD + J * 32, LD, 1, kend - kb, acc[s]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < kTPW; ++s) {
        const int t = wv + kWaves * s;
        if (t < ntile) {
          const int I = t / TBe, J = (I / TB) * TB + (t - I * TBe);
          if (J < NT) {
            const float sc = ws.nrm[J * 32 + (lane & 31)];
#pragma unroll
            for (int e = 0; e < 16; ++e) Q[(I * 32 + acc_row(e, lane)) * LD + J * 32 + (lane & 31)] = acc[s][e] * sc;
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

// ------------------------------------------------------------------------------------------------ 3. back-transformation
// Q <- H_0 H_1 ... H_{n-3} Q in blocks of 32 reflectors, compact WY:  H_{k0} .. H_{k0+31} = I - V^T T V (V = 32 reflector rows).
// All reflectors are read back from R (row k = v_k) into Vt (DP x LD, the buffer the D&C no longer needs) ONCE; the Gram
// matrices and triangular factors of all blocks are then formed side by side (they do not depend on Q), so that the
// serial part per block is three MFMA products: Y = V Q, Y <- T Y, Q -= V^T Y.
// T_b is parked in columns 0..31 of block b's rows of Vt -- zeros of the reflectors that no product reads (b >= 1) -- and
// in ws.t0 for b = 0.  ws.Y first holds the Gram matrices (32 x 32 each, unpadded, float4-aligned rows).
// `hook()` runs once the reflector registers are free again: the caller's chance to start global loads of its own that should
// land behind the back-transformation (e.g. the operands of the cell's epilogue).
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
template <int NT, class Hook>
__device__ __forceinline__ void back_transform(float* __restrict__ Vt, float* __restrict__ Q, int n, EigScratch<NT * 32>& ws,
                                               const ReflectorFetch<NT * 32>& fetch, Hook&& hook) {
  constexpr int DP = NT * 32, LD = DP + 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nr = n - 2;
  if (nr <= 0) {
    hook();
    return;
  }
  const int nblk = (nr + 31) / 32;
  float* Y = ws.Y;
  UGLAD_STAMP(ws, 42);
  fetch.land(Vt);
  hook();
  __syncthreads();
  UGLAD_STAMP(ws, 43);
  // Gram matrices G_b = V_b V_b^T, one wave per block (rows <= 32 b of every reflector of block b are zero: K starts there)
  for (int b = wv; b < nblk; b += kWaves) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* Vb = Vt + 32 * b * LD + 32 * b;
    mfma_tile(Vb, LD, 1, Vb, 1, LD, DP - 32 * b, acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) Y[b * 1024 + acc_row(e, lane) * 32 + (lane & 31)] = acc[e];
  }
  __syncthreads();
  UGLAD_STAMP(ws, 44);
  // T_b = (triu(G_b, 1) + diag(1 / tau))^-1 by back substitution, one column per thread, all blocks at once
  if (tid < 32 * nblk) {
    const int b = tid >> 5, c = tid & 31;
    const float* G = Y + b * 1024;
    float y[32];
#pragma unroll
    for (int j = 31; j >= 0; --j) {
      float a0 = (j == c) ? 1.f : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four chains instead of one of length 31
#pragma unroll
      for (int q = (j + 1) / 4; q < 8; ++q) {
        const f4 g4 = *reinterpret_cast<const f4*>(&G[j * 32 + 4 * q]);
        if (4 * q + 0 > j) a0 = fmaf(-g4.x, y[4 * q + 0], a0);
        if (4 * q + 1 > j) a1 = fmaf(-g4.y, y[4 * q + 1], a1);
        if (4 * q + 2 > j) a2 = fmaf(-g4.z, y[4 * q + 2], a2);
        if (4 * q + 3 > j) a3 = fmaf(-g4.w, y[4 * q + 3], a3);
      }
      y[j] = ws.tau[32 * b + j] * ((a0 + a1) + (a2 + a3));
    }
    float* T = (b == 0) ? ws.t0 : Vt + 32 * b * LD;
    const int ts = (b == 0) ? 33 : LD;
#pragma unroll
    for (int j = 0; j < 32; ++j) T[j * ts + c] = y[j];
  }
  __syncthreads();
  UGLAD_STAMP(ws, 45);
  for (int b = nblk - 1; b >= 0; --b) {
    const int kb = 32 * b;
    const float* Vb = Vt + kb * LD;
    const float* T = (b == 0) ? ws.t0 : Vt + kb * LD;
    const int ts = (b == 0) ? 33 : LD;
    for (int J = wv; J < NT; J += kWaves) {  // Y = V_b Q  (rows < kb of Q do not contribute)
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_tile(Vb + kb, LD, 1, Q + kb * LD + J * 32, LD, 1, DP - kb, acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[acc_row(e, lane) * LD + J * 32 + (lane & 31)] = acc[e];
    }
    __syncthreads();
    UGLAD_STAMP(ws, 46 + 4 * b);
    for (int J = wv; J < NT; J += kWaves) {  // Y <- T_b Y, every wave inside its own column tile
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      mfma_tile(T, ts, 1, Y + J * 32, LD, 1, 32, acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) Y[acc_row(e, lane) * LD + J * 32 + (lane & 31)] = acc[e];
    }
    __syncthreads();
    UGLAD_STAMP(ws, 47 + 4 * b);
    {  // Q -= V_b^T Y on the row tiles >= b
      const int ntile = (NT - b) * NT;
#pragma unroll
      for (int s = 0; s < (NT * NT + kWaves - 1) / kWaves; ++s) {
        const int t = wv + kWaves * s;
        if (t < ntile) {
          const int I = b + t / NT, J = t % NT;
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          mfma_tile(Vb + I * 32, 1, LD, Y + J * 32, LD, 1, 32, acc);
#pragma unroll
          for (int e = 0; e < 16; ++e) Q[(I * 32 + acc_row(e, lane)) * LD + J * 32 + (lane & 31)] -= acc[e];
        }
      }
    }
    __syncthreads();
    UGLAD_STAMP(ws, 48 + 4 * b);
  }
}

// LDS floats the first big buffer needs: a DP x (DP+1) matrix.
template <int DP>
constexpr int eig_buf0_floats() {
  return DP * (DP + 1);
}

// The two big per-matrix buffers of a kernel: LDS while they fit (DP <= 128); beyond that, slabs of the caller's workspace
// (behind the M x 3 x DP tridiagonal region), which stay L2-resident -- the same code then runs on global pointers, slower
// but with no size limit from the 160 KB of LDS.  NA floats for the first buffer, NB for the second.
template <int DP>
constexpr int big_floats() {
  return 2 * ((eig_buf0_floats<DP>() + 3) & ~3);
}
// Workspace floats per matrix ahead of the big buffers: d, e, tau (3 DP) and the DP / 32 triangular factors of the
// back-transformation (32 x 32 each) that eig_lean.h hands from one wave to the others.
template <int DP>
constexpr int kWsPerMatrix = 3 * DP + (DP / 32) * 1024;
#define UGLAD_BIG_BUFFERS(A, NA, B, NB, GWS)                                                                     \
  constexpr bool kGM = DP > 128;                                                                                 \
  __shared__ __attribute__((aligned(16))) float A##_lds[kGM ? 4 : (NA)];                                         \
  __shared__ __attribute__((aligned(16))) float B##_lds[kGM ? 4 : (NB)];                                         \
  float* A = kGM ? (GWS) + (size_t)gridDim.x * kWsPerMatrix<DP> + (size_t)blockIdx.x * big_floats<DP>() : A##_lds; \
  float* B = kGM ? A + big_floats<DP>() / 2 : B##_lds;

// ------------------------------------------------------------------------------------------------ driver
// Tridiagonal form (d, e, tau: 3 x DP floats at `tri`) and reflectors (rows of R) come from tridiag_kernel.  Out: ws.d[0..n)
// eigenvalues (ascending), buf1 (stride DP+1) eigenvectors in columns 0..n-1 (identity on the padding); buf0 is scratch.
template <int NT, class Hook = NoHook>
__device__ __forceinline__ void symeig_from_tridiagonal(float* __restrict__ buf0, float* __restrict__ buf1, int n,
                                                        EigScratch<NT * 32>& ws, const float* __restrict__ tri,
                                                        const float* __restrict__ R, int ldr, Hook&& hook = Hook()) {
  constexpr int DP = NT * 32;
  for (int i = threadIdx.x; i < DP; i += kThreads) {
    ws.d[i] = (i < n) ? tri[i] : 0.f;
    ws.e[i] = (i < n) ? tri[DP + i] : 0.f;
    ws.tau[i] = (i < n) ? tri[2 * DP + i] : 0.f;
  }
  __syncthreads();
  UGLAD_STAMP(ws, 1);
  ReflectorFetch<DP> fetch;
  fetch.R = R;
  fetch.ldr = ldr;
  fetch.n = n;
  dc_tridiagonal<NT>(buf0, buf1, n, ws, fetch);
  UGLAD_STAMP(ws, 40);
  back_transform<NT>(buf0, buf1, n, ws, fetch, hook);
  UGLAD_STAMP(ws, 41);
}

}  // namespace uglad
