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
#include <type_traits>

#include "eig_dc.h"

namespace uglad {

// Reflectors are applied in blocks of kRB = 16 (compact WY).  Packed upper triangle of a 16 x 16 Gram matrix: row j keeps
// columns gtri_start(j) .. 15 (the start rounded down to a multiple of 4 so that a row is read in 16-byte pieces), rows back
// to back.
constexpr int kRB = 16;
constexpr int kGtriFloats = 144;
__device__ __forceinline__ int gtri_start(int j) { return 4 * ((j + 1) >> 2); }
__device__ __forceinline__ int gtri_off(int j) {  // = sum over j' < j of (16 - gtri_start(j'))
  const int a = j >> 2, r = j & 3;
  return 16 * j - 4 * (2 * a * (a - 1) + a * (r + 1));
}
// Staging area of one reflector block in LDS: 16 rows of DP + 4 floats (16-byte aligned rows; ds_read_b128 of 16 consecutive
// rows at one column offset hits 16 different 16-byte bank groups) followed by its triangular factor, 16 rows of 20 floats.
template <int DP>
constexpr int kStageFloats = kRB * (DP + 4) + kRB * 20;

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
    alignas(16) float gtri[(DP / kRB) * kGtriFloats];   // back-transformation: packed upper rows of the Gram matrices ...
    alignas(16) float stage[kStageFloats<DP>];          // ... then the reflector block being applied and its T factor
  };
  float rho[DP / 2 + 1];
  int skip[DP / 2 + 1], fix[DP / 2 + 1], bmax[DP / 2 + 1];
};

// ------------------------------------------------------------------------------------------------ secular equation
// secular_root (eig_dc.h) with the lane's poles held in REGISTERS for the whole solve, relative to the origin pole, and a
// leaner evaluation: the terms rz_j / (d_j - x) left of the root are the negative ones, so the sums the rational step needs
// come out of one pass without a per-pole left/right test:  w = 1 + sum t_j,  sum |t_j| (the error bound),
// dpsi = sum min(t_j, 0) / (d_j - x),  dphi = sum t_j / (d_j - x) - dpsi.  LPR lanes share a root (LPR = 1, 2, 4), NP poles
// per lane: merges of up to LPR * NP poles.
#ifndef UGLAD_SECULAR_MAXIT
#define UGLAD_SECULAR_MAXIT 48
#endif
constexpr int kSecularMaxIt = UGLAD_SECULAR_MAXIT;

template <int LPR>
__device__ __forceinline__ float group_sum_n(float v) {
  if (LPR >= 2) v += lane_xor1(v);
  if (LPR >= 4) v += dpp_move<0x4e>(v);
  if (LPR == 8) v += dpp_move<0x141>(v);  // row_half_mirror: lane l <- lane 7 - l of its group of eight (the other quad's sum)
  return v;
}

#ifdef UGLAD_STAMPS
__device__ unsigned long long g_sec[16][8];  // diagnostic build: stamps inside the secular solver (workgroup 0, thread 0), row = NP slot
#define SEC_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sec[(NP < 16 ? NP : 15)][k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEC_STAMP(k) do {} while (0)
#endif
template <int LPR, int NP>
__device__ __forceinline__ int secular_root_reg(const float* __restrict__ ds, const float* __restrict__ rz, float rho, int nb,
                                                int i, int sub, int& Kout, float& mu_out) {
  constexpr float kEps = 5.96e-8f;
  SEC_STAMP(0);
  float pd[NP], pr[NP];
  {
    // pole j = sub + LPR t, present while t < cnt.  The loads are unconditional (lo + LPR * NP may pass the end of ds / rz but stays
    // inside the caller's scratch vectors; what lies beyond nb is selected away) and the test compares the unrolled t, a literal, with ONE register: written as `j < nb` the
    // compiler keeps the NP values of j live across the caller's level loop -- 32 registers at the last merge, all of them spilled
    const int cnt = (nb - sub + LPR - 1) / LPR;
    const float* dsl = ds + sub;
    const float* rzl = rz + sub;
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const float dv = dsl[LPR * t], rv = rzl[LPR * t];
      pd[t] = (t < cnt) ? dv : 3.0e38f;  // absent poles: weight zero, infinitely far away
      pr[t] = (t < cnt) ? rv : 0.f;
    }
  }
  SEC_STAMP(1);
  // starting point as in secular_root: evaluate at a test point, keep the two nearest poles exact, freeze the rest
  const bool last = i == nb - 1;
  const int ia = last ? nb - 2 : i;
  const float hi_last = rho * 1.00001f + 1e-30f;
  const float dorg = ds[i];
  float test = 0.5f * (ds[last ? i : i + 1] - dorg);
  if (last) {
    const float d1l = ds[nb - 2] - dorg, pl = rz[nb - 2], ql = rz[nb - 1];
    const float bl = d1l + pl + ql, cl = ql * d1l;
    // (4 cl as an exponent step: written 4.f * cl the compiler packs {bl, 4} * {bl, cl} into one v_pk_mul_f32 and keeps the constant pair in
    // a register of its own across the caller's level loop -- a spill)
    const float x0 = 0.5f * (bl + __builtin_amdgcn_sqrtf(fmaxf(bl * bl - __builtin_ldexpf(cl, 2), 0.f)));
    test = (x0 > 0.f && x0 < hi_last) ? x0 : 0.5f * hi_last;
  }
  float wsum = 0.f;
#pragma unroll
  for (int t = 0; t < NP; ++t) wsum = fmaf(pr[t], fast_rcp((pd[t] - dorg) - test), wsum);
  const float wt = 1.f + group_sum_n<LPR>(wsum);
  const int K = (last || wt > 0.f) ? i : i + 1;  // origin: the pole nearest to the root
  SEC_STAMP(2);
  const float dK = ds[K];
#pragma unroll
  for (int t = 0; t < NP; ++t) pd[t] -= dK;
  const float d1 = ds[ia] - dK, d2 = ds[ia + 1] - dK;
  const float p = rz[ia], q = rz[ia + 1];
  const float xt = (dorg - dK) + test;
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
  const float sq0 = __builtin_amdgcn_sqrtf(fmaxf(bq * bq - 4.f * rest * cq, 0.f));
  // the root of the model  rest + p / (d1 - x) + q / (d2 - x) = 0  that lies in (lo, hi): rest x^2 - bq x + cq = 0.  For an interior root
  // it is the root of SMALL magnitude whichever pole is the origin (K = i: cq = p d2 > 0, K = i + 1: cq = q d1 < 0), 2 cq / (bq + sq0) for
  // bq > 0.  (Up to round 3 the K = i + 1 case took the other root, which is never in the bracket: 40 % of all roots started from the
  // midpoint of their bracket instead -- mean evaluations 3.1 -> 2.9 on the study of profiles/r04_secular_study.txt.)  The last root of a
  // merge (K = i = nb - 1, ia = nb - 2) lies to the RIGHT of both poles: there it is the other one.
  float mu;
  if (!last)
    mu = (bq > 0.f) ? 2.f * cq * fast_rcp(bq + sq0) : (bq - sq0) * fast_rcp(2.f * rest);
  else
    mu = (bq < 0.f) ? 2.f * cq * fast_rcp(bq - sq0) : (bq + sq0) * fast_rcp(2.f * rest);
  if (!(mu > lo && mu < hi)) mu = 0.5f * (lo + hi);
  const int jr = i + 1;
  const float dl1 = ds[i] - dK, dl2 = (jr < nb) ? ds[jr] - dK : 0.f;
  // The iteration (round 4): GRAGG'S scheme -- the model  c + s / (D1 - eta) + S / (D2 - eta)  of the secular function around the two poles that
  // bracket the root matches its value AND ITS FIRST TWO DERIVATIVES at the current point (cubic convergence; rounds 1-3 ran the "middle way",
  // which matches value and first derivative with the left and right sums taken apart: quadratic) -- and a step smaller than 2^-12 |mu| is
  // ACCEPTED WITHOUT THE EVALUATION THAT WOULD ONLY CONFIRM IT (what is left after a step of relative size h is h^3, at worst h^2 = 6e-8).
  // Same instruction count per evaluation (the third-order sum replaces the left-hand sum), but a wave runs as many evaluations as its slowest
  // root, and that is what moves: evaluations per root 2.9 -> 2.3 on average, roots needing five or more 6.6 % -> < 0.5 %, the relative accuracy of mu
  // unchanged (median 8e-8, 99.9 % below 7e-6 under either rule: profiles/r04_secular_study.txt).  For the last root of a merge every pole lies to its
  // left, so the left-hand sum the one-pole model of that case needs IS the first-derivative sum.
  constexpr float kAcceptStep = 2.44140625e-4f;  // 2^-12
  SEC_STAMP(3);
  int it = 0;
  for (; it < kSecularMaxIt; ++it) {
    float ws_ = 0.f, as_ = 0.f, da_ = 0.f, d3_ = 0.f;
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const float r = fast_rcp(pd[t] - mu);
      const float term = pr[t] * r;
      const float tr = term * r;
      ws_ += term;
      as_ += fabsf(term);
      da_ += tr;
      d3_ = fmaf(tr, r, d3_);
    }
    ws_ = group_sum_n<LPR>(ws_);
    as_ = group_sum_n<LPR>(as_);
    const float dsum = group_sum_n<LPR>(da_);
    const float d3 = group_sum_n<LPR>(d3_);
    const float D1 = dl1 - mu, D2 = dl2 - mu;
    const float w = 1.f + ws_;
    if (fabsf(w) <= 8.f * kEps * (1.f + as_)) break;
    if (w < 0.f) lo = mu; else hi = mu;
    // s = D1^3 u, S = D2^3 v with the cubes taken one factor at a time (D1^3 alone underflows for a root that hugs its pole)
    const float inv_den = fast_rcp(D2 - D1);
    const float u = (D2 * d3 - dsum) * inv_den, v = (dsum - D1 * d3) * inv_den;
    const float s1 = D1 * (D1 * u), S1 = D2 * (D2 * v);  // s / D1, S / D2
    const float a = w - s1 - S1;
    const float b = fmaf(a, D1 + D2, fmaf(D1, s1, D2 * S1));
    const float g = D1 * D2 * w;
    const float sq = __builtin_amdgcn_sqrtf(fabsf(fmaf(b, b, -4.f * a * g)));
    const bool bneg = b <= 0.f, a0 = a == 0.f;
    float num = bneg ? (a0 ? g : b - sq) : 2.f * g;
    float den = bneg ? (a0 ? b : 2.f * a) : b + sq;
    float add = 0.f;
    if (jr >= nb) {  // (per lane: only the last root of a merge)
      const float c = w - dsum * D1;
      num = (c != 0.f) ? dsum * D1 * D1 : 0.f;
      den = (c != 0.f) ? c : 1.f;
      add = (c != 0.f) ? D1 : 0.f;
    }
    float eta = fmaf(num, fast_rcp(den), add);
    const float newton = -w * fast_rcp(dsum);
    if (!(fabsf(eta) < 3.0e38f) || w * eta >= 0.f) eta = newton;
    float nw = mu + eta;
    // (a step that leaves the bracket is replaced by its midpoint -- and a midpoint is never accepted unseen: when the starting point's test
    // evaluation already sat on the root to rounding (a merge whose other poles carry no weight), the bracket is a few 1e-5 wide, the model's step
    // lands ON its end, and the midpoint, "small" by the rule below, was taken for the root: lambda 2.3e-5 off on a 3 x 3 matrix,
    // tests/test_gpu_parity.py::test_symeig_every_size_up_to_64)
    const bool outside = !(nw > lo && nw < hi);
    if (outside) nw = 0.5f * (lo + hi);
    if (nw == mu) break;
    const bool small = !outside && fabsf(nw - mu) <= kAcceptStep * fabsf(nw);
    mu = nw;
    if (small) break;
  }
  SEC_STAMP(4);
#ifdef UGLAD_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) g_sec[(NP < 16 ? NP : 15)][5] = it + 1;
#endif
  Kout = K;
  mu_out = mu;
  return it + 1;
}

// ------------------------------------------------------------------------------------------------ divide & conquer, first levels
// The merges inside a segment of SEG columns are the business of ONE wave: same steps as the workgroup-wide levels below, but
// ordered by the wave's own instruction stream instead of barriers -- the waves drift apart and fill each other's latencies,
// and seven workgroup barriers per level disappear.  SEG = 16: h = 2, 4, 8 on all eight waves (segment [16 w, 16 w + 16)),
// four lanes per root; the eigenvector update of the 16 x 16 diagonal block is one product on v_mfma_f32_16x16x4_f32.
// SEG = 32: h = 16 on waves 0..3, two lanes per root, the 32 x 32 block on v_mfma_f32_32x32x2_f32.
template <int NT, int SEG>
__device__ __forceinline__ void dc_local(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws, int h_first, int lvl, int sg) {
  constexpr int DP = NT * 32, LD = DP + 1;
  constexpr float kEps = 5.96e-8f;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = 64 / SEG;  // lanes per root
  const int seg = SEG * sg;          // (sg = wave index; beyond D = 128 a wave takes several segments in turn)
  if (seg >= n || seg >= DP) return;  // (wave-uniform)
  const int l16 = lane & 15, g4 = lane >> 4;
  // per-merge scalars (rho, skip, fix, bmax): slot 4 w + (merge index inside the segment) -- the waves are at different levels
  // at the same time, so the workgroup-wide numbering p / bs of the later levels would collide here
  auto slot = [&](int p, int bs) { return (SEG / 4) * sg + (p - seg) / bs; };
  for (int h = h_first; h < SEG && h < n; h *= 2, ++lvl) {
    const int bs = 2 * h;
    if (sg == 0) UGLAD_STAMP(ws, 2 + 5 * lvl);
    {  // z, merged order, max |d| per merge (lane = original column of the segment)
      const int g = seg + lane;
      if (lane < SEG && g < n) {
        const int lo = (g / bs) * bs, mid = lo + h;
        const int hi = (lo + bs < n) ? lo + bs : n;
        const float dg = ws.d[g];
        int rank = g - lo;
        float z = 0.f;
        if (mid < n) {
          const float ec = ws.e[mid - 1];
          if (g < mid) {
            for (int j = mid; j < hi; ++j) rank += (ws.d[j] < dg) ? 1 : 0;
            z = Q[(mid - 1) * LD + g];
          } else {
            rank = g - mid;
            for (int j = lo; j < mid; ++j) rank += (ws.d[j] <= dg) ? 1 : 0;
            z = (ec >= 0.f) ? Q[mid * LD + g] : -Q[mid * LD + g];
          }
          z *= 0.70710678f;
          if (fabsf(z) < kZFloor) z = (z < 0.f) ? -kZFloor : kZFloor;
          atomicMax(&ws.bmax[slot(g, bs)], __float_as_int(fabsf(dg)));
        }
        ws.ds[lo + rank] = dg;
        ws.zs[lo + rank] = z;
        ws.perm[lo + rank] = g;
      }
    }
    UGLAD_WAVE_SYNC();
    {  // coupling test, rz = rho z^2, poles that need separating (lane = sorted position)
      const int p = seg + lane;
      if (lane < SEG && p < n) {
        const int lo = (p / bs) * bs, mid = lo + h, blk = slot(p, bs);
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
    UGLAD_WAVE_SYNC();
    {  // rare: one lane per merge walks its poles and pushes equal ones a few ulps apart
      const int blo = seg + lane * bs, blk = (SEG / 4) * sg + lane;
      if (lane < SEG / bs && blo < n && ws.fix[blk]) {
        const int bhi = (blo + bs < n) ? blo + bs : n;
        const float scale = fmaxf(__int_as_float(ws.bmax[blk]), ws.rho[blk]);
        float prev = ws.ds[blo];
        for (int j = blo + 1; j < bhi; ++j) {
          float cur = ws.ds[j];
          const float gap = 4.f * kEps * fmaxf(fabsf(cur), fabsf(prev)) + 1e-10f * scale;
          if (cur < prev + gap) cur = prev + gap;
          ws.ds[j] = cur;
          prev = cur;
        }
        ws.fix[blk] = 0;
      }
    }
    UGLAD_WAVE_SYNC();
    if (sg == 0) UGLAD_STAMP(ws, 3 + 5 * lvl);
    // secular roots: root p = seg + lane / LPR, LPR lanes each
    const int p = seg + lane / LPR, sub = lane % LPR;
    int lo = 0, hi = 0;
    bool act = false;
    if (p < n) {
      lo = (p / bs) * bs;
      hi = (lo + bs < n) ? lo + bs : n;
      act = (lo + h < n) && (ws.skip[slot(p, bs)] == 0);
    }
    {
      int K = p - lo;
      float mu = 0.f;
      int evals = 0;
      if (act) {
        const float* dsb = ws.ds + lo;
        const float* zhb = ws.zh + lo;
        const float rho = ws.rho[slot(p, bs)];
        if (SEG == 32) evals = secular_root_reg<LPR, 32 / LPR>(dsb, zhb, rho, hi - lo, p - lo, sub, K, mu);
        else if (bs == 4) evals = secular_root_reg<LPR, (4 + LPR - 1) / LPR>(dsb, zhb, rho, hi - lo, p - lo, sub, K, mu);
        else if (bs == 8) evals = secular_root_reg<LPR, 8 / LPR>(dsb, zhb, rho, hi - lo, p - lo, sub, K, mu);
        else evals = secular_root_reg<LPR, 16 / LPR>(dsb, zhb, rho, hi - lo, p - lo, sub, K, mu);
      }
#ifdef UGLAD_STAMPS
      if (sub == 0 && p < n) {
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
    UGLAD_WAVE_SYNC();
    if (sg == 0) UGLAD_STAMP(ws, 4 + 5 * lvl);
    {  // Gu-Eisenstat zhat (pole j = p), stored with its pole in the original column order
      float prod = 1.f;
      if (act) {
        const float dj = ws.ds[p];
        for (int i = lo + sub; i < hi; i += LPR) {
          const float num = (ws.dk[i] - dj) + ws.mu[i];
          const float den = (i == p) ? 1.f : ws.ds[i] - dj;
          prod *= num * fast_rcp(den);
        }
      }
      prod = group_prod<LPR>(prod);
      if (sub == 0 && p < n) {
        ws.act[p] = act ? 1 : 0;
        if (act) {
          const float zhat = sqrtf(fmaxf(prod, 0.f));
          const int g = ws.perm[p];
          ws.invo[g] = (ws.zs[p] < 0.f) ? -zhat : zhat;
          ws.dso[g] = ws.ds[p];
        }
      }
    }
    UGLAD_WAVE_SYNC();
    if (sg == 0) UGLAD_STAMP(ws, 5 + 5 * lvl);
    if (SEG == 16) {  // Q(block) <- Q(block) W' diag(1/||.||): C[i][j] = sum_k Q[seg+i][seg+k] W'[seg+k][seg+j], lane group g4: k = 4 g4 + s
      const int col = seg + l16;
      const float dki = ws.dk[col], mui = ws.mu[col];
      const bool acti = (col < n) && ws.act[col] != 0;
      const int permi = (col < n) ? ws.perm[col] : col;
      const int blo = (col / bs) * bs;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      float s2 = 0.f;
      float av[4], bv[4];
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) {
        const int k = seg + 4 * g4 + ss;
        av[ss] = Q[(seg + l16) * LD + k];
        float val = acti ? ws.invo[k] * fast_rcp((ws.dso[k] - dki) - mui) : ((k == permi) ? 1.f : 0.f);
        if ((unsigned)(k - blo) >= (unsigned)bs) val = 0.f;
        bv[ss] = val;
        s2 = fmaf(val, val, s2);
      }
#pragma unroll
      for (int ss = 0; ss < 4; ++ss) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ss], bv[ss], acc, 0, 0, 0);
      s2 += __shfl_xor(s2, 16);
      s2 = sum_halves(s2);
      const float sc = acti ? 1.0f / sqrtf(s2) : 1.f;
      UGLAD_WAVE_SYNC();  // (every lane has read its A operands: the block may be overwritten)
#pragma unroll
      for (int r = 0; r < 4; ++r) Q[(seg + 4 * g4 + r) * LD + col] = acc[r] * sc;
    } else {  // the 32 x 32 block on the 32 x 32 x 2 MFMA: lane half kh takes k = 2 u + kh
      const int li = lane & 31, kh = lane >> 5;
      const int col = seg + li;
      const float dki = ws.dk[col], mui = ws.mu[col];
      const bool acti = (col < n) && ws.act[col] != 0;
      const int permi = (col < n) ? ws.perm[col] : col;
      const int blo = (col / bs) * bs;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      float s2 = 0.f;
      float av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = seg + 2 * u + kh;
        av[u] = Q[(seg + li) * LD + k];
        float val = acti ? ws.invo[k] * fast_rcp((ws.dso[k] - dki) - mui) : ((k == permi) ? 1.f : 0.f);
        if ((unsigned)(k - blo) >= (unsigned)bs) val = 0.f;
        bv[u] = val;
        s2 = fmaf(val, val, s2);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      s2 = sum_halves(s2);
      const float sc = acti ? 1.0f / sqrtf(s2) : 1.f;
      UGLAD_WAVE_SYNC();
#pragma unroll
      for (int e = 0; e < 16; ++e) Q[(seg + acc_row(e, lane)) * LD + col] = acc[e] * sc;
    }
    {
      if (lane < SEG && seg + lane < n) ws.d[seg + lane] = ws.lam[seg + lane];
      if (lane < SEG / 4) ws.bmax[(SEG / 4) * sg + lane] = 0;
    }
    UGLAD_WAVE_SYNC();
    if (sg == 0) UGLAD_STAMP(ws, 6 + 5 * lvl);
  }
}

// ------------------------------------------------------------------------------------------------ divide & conquer
// As dc_tridiagonal (eig_dc.h) up to the roots and the Gu-Eisenstat vector; the eigenvector update Q <- Q W' diag(1/||.||) then
// generates W' on the fly.
// With `last` != nullptr (few large matrices, wide_bwd.h) the LAST merge is only prepared -- merged order, z, coupling test, poles
// pushed apart -- and handed over in global memory for launches with many workgroups per matrix to carry out:
//   last[0 .. DP) ds (poles, sorted), [DP .. 2DP) zs, [2DP .. 3DP) rho z^2, [3DP .. 4DP) perm (int), [4DP] rho, [4DP + 1] skip (int).
constexpr int kLastFloats(int DP) { return 7 * DP; }  // (+ dk, mu, column scale appended by the secular launch)
template <int NT>
__device__ __forceinline__ void dc_tridiagonal_lean(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws,
                                                    float* __restrict__ last = nullptr) {
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
  for (int sg = wv; 16 * sg < DP; sg += kWaves) dc_local<NT, 16>(Q, n, ws, 2, 1, sg);   // merges up to 16 columns, wave-local
  __syncthreads();
  for (int sg = wv; 32 * sg < DP; sg += kWaves) dc_local<NT, 32>(Q, n, ws, 16, 4, sg);  // merges to 32 columns (two lanes per root)
  __syncthreads();
  if (tid < DP / 2 + 1) {  // (the per-merge scalars are numbered workgroup-wide from here on)
    ws.bmax[tid] = 0;
    ws.fix[tid] = 0;
  }
  __syncthreads();

  int lvl = 5;
  for (int h = 32; h < n; h *= 2, ++lvl) {
    const int bs = 2 * h;
    // (shadow the outer ones: nothing derived from the thread index is hoisted out of the level loop -- the hoisted values, all of
    // them one instruction from tid, were what the kernel spilled: profiles/r04_kernel_meta.txt)
    const int tid = opaque_v(threadIdx.x), lane = tid & 63, wv = tid >> 6;
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
    if (last != nullptr && bs >= n) {  // (uniform) the last merge is carried out by other launches
      if (tid < n) {
        last[tid] = ws.ds[tid];
        last[DP + tid] = ws.zs[tid];
        last[2 * DP + tid] = ws.skip[0] ? 0.f : ws.zh[tid];
        reinterpret_cast<int*>(last)[3 * DP + tid] = ws.perm[tid];
      }
      if (tid == 0) {
        last[4 * DP] = ws.rho[0];
        reinterpret_cast<int*>(last)[4 * DP + 1] = ws.skip[0];
      }
      return;
    }
    // ---- L3: secular roots, four lanes per root, the lane's poles in registers.  (Measured: ONE lane per root at the small
    // merges -- a quarter of the issue slots -- is slower, 15 k instead of 9 k cycles per level: the solve is a dependent
    // chain, and what counts is its length per lane, not the number of lanes.)
    {
      auto roots = [&](auto lpr_c, auto np_c) {
        constexpr int L = decltype(lpr_c)::value, NPv = decltype(np_c)::value;
        const int pr_ = tid / L, sb = tid % L;
        if (pr_ < n) {
          const int blk = pr_ / bs, lo_ = blk * bs;
          const int hi_ = (lo_ + bs < n) ? lo_ + bs : n;
          const bool on = (lo_ + h < n) && (ws.skip[blk] == 0);
          int K = pr_ - lo_;
          float mu = 0.f;
          int evals = 0;
          if (on) evals = secular_root_reg<L, NPv>(ws.ds + lo_, ws.zh + lo_, ws.rho[blk], hi_ - lo_, pr_ - lo_, sb, K, mu);
#ifdef UGLAD_STAMPS
          if (sb == 0 && lvl < 14) {
            atomicMax(reinterpret_cast<int*>(&ws.stamp[80 + lvl]), evals);
            atomicAdd(reinterpret_cast<int*>(&ws.stamp[80 + lvl]) + 1, evals);
          }
#else
          (void)evals;
#endif
          if (sb == 0) {
            const float dK = ws.ds[lo_ + K];
            ws.dk[pr_] = dK;
            ws.mu[pr_] = mu;
            ws.lam[pr_] = dK + mu;
          }
        }
      };
      using std::integral_constant;
      constexpr int LR = (kThreads / DP >= 4) ? 4 : 2;
      if (bs <= 64) {
        roots(integral_constant<int, LR>(), integral_constant<int, 64 / LR>());
      } else if (bs == 128) {
        roots(integral_constant<int, LR>(), integral_constant<int, 128 / LR>());  // (64 poles per lane at LR = 2: 256-register kernels only)
      } else {
        // bs = 256 (D > 128): 256 poles are too many for the registers of two or four lanes -- EIGHT lanes per root, 32 poles each,
        // 64 roots per pass (the LDS-resident solver of eig_dc.h with two lanes per root took 180 k cycles here)
        for (int pass = 0; 64 * pass < n; ++pass) {
          const int pr_ = 64 * pass + tid / 8, sb = tid % 8;
          if (pr_ < n) {
            const int blk = pr_ / bs, lo_ = blk * bs;
            const int hi_ = (lo_ + bs < n) ? lo_ + bs : n;
            int K = pr_ - lo_;
            float mu = 0.f;
            if ((lo_ + h < n) && (ws.skip[blk] == 0))
              (void)secular_root_reg<8, 32>(ws.ds + lo_, ws.zh + lo_, ws.rho[blk], hi_ - lo_, pr_ - lo_, sb, K, mu);
            if (sb == 0) {
              const float dK = ws.ds[lo_ + K];
              ws.dk[pr_] = dK;
              ws.mu[pr_] = mu;
              ws.lam[pr_] = dK + mu;
            }
          }
        }
      }
    }
    __syncthreads();
    UGLAD_STAMP(ws, 4 + 5 * lvl);
    {  // ---- L4: Gu-Eisenstat zhat (pole j = p), stored with its pole in the ORIGINAL column order for the GEMM's B operand
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
      // NT = 4, last merge (16 tiles, two per wave): the two tiles of a wave are the row tiles 2 m, 2 m + 1 of ONE column tile --
      // same k range, same B operand, which is then generated once for both (the generation, not the MFMA, bounds this loop)
      const bool paired = (NT == 4) && (kTPW == 2) && (ntile == 16) && (h == 64);
      if (paired) {
        const int J = wv & 3, I0 = 2 * (wv >> 2);
        const int kb = (I0 / 2) * 64, kend = kb + 64;
        const int col = J * 32 + li;
        const float dki = ws.dk[col], mui = ws.mu[col];
        const bool acti = ws.act[col] != 0;
        const int permi = ws.perm[col];
        const float* a0 = Q + (I0 * 32 + li) * LD + kb + kh;
        const float* a1 = a0 + 32 * LD;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = acc[kTPW - 1][e] = 0.f;
        float s2 = 0.f;
        float av0[8], av1[8], dv[8], iv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          av0[u] = a0[2 * u];
          av1[u] = a1[2 * u];
          dv[u] = ws.dso[kb + 2 * u + kh];
          iv[u] = ws.invo[kb + 2 * u + kh];
        }
        for (int k0 = kb; k0 < kend; k0 += 16) {
          const int kn = (k0 + 16 < kend) ? k0 + 16 : k0;
          float an0[8], an1[8], dn[8], in_[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            an0[u] = a0[(kn - kb) + 2 * u];
            an1[u] = a1[(kn - kb) + 2 * u];
            dn[u] = ws.dso[kn + 2 * u + kh];
            in_[u] = ws.invo[kn + 2 * u + kh];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + 2 * u + kh;
            const float val = acti ? iv[u] * fast_rcp((dv[u] - dki) - mui) : ((k == permi) ? 1.f : 0.f);
            s2 = fmaf(val, val, s2);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u], val, acc[0], 0, 0, 0);
            acc[kTPW - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u], val, acc[kTPW - 1], 0, 0, 0);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            av0[u] = an0[u];
            av1[u] = an1[u];
            dv[u] = dn[u];
            iv[u] = in_[u];
          }
        }
        s2 = sum_halves(s2);
        if (kh == 0 && acti) atomicAdd(&ws.nrm2[col], s2);  // the two waves of a column tile hold its two k ranges
        __syncthreads();
        const float sc = acti ? 1.0f / sqrtf(ws.nrm2[col]) : 1.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          Q[(I0 * 32 + acc_row(e, lane)) * LD + col] = acc[0][e] * sc;
          Q[((I0 + 1) * 32 + acc_row(e, lane)) * LD + col] = acc[kTPW - 1][e] * sc;
        }
      } else {
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
            s2 = sum_halves(s2);
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
      }  // !paired
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

// Q <- H_0 ... H_{n-3} Q in blocks of 16 reflectors (compact WY: H_{kb} .. H_{kb+15} = I - V^T T V, V = 16 reflector rows).
// R: reflector rows in global memory (stride ldr), tau: their scalars (global), Tws: 16 x 16 floats per block of global scratch
// for the triangular factors.
//   1. Gram matrices G_b = V_b V_b^T, one wave per block (v_mfma_f32_16x16x4_f32; A and B fragments are the same registers),
//      upper rows packed into LDS over the divide & conquer's vectors;
//   2. T_b = (triu(G_b, 1) + diag(1 / tau))^-1 by back substitution, one column per thread, all blocks at once -> workspace;
//   3. the blocks, last to first.  The block V_b (16 x DP) and T_b are staged in LDS by all threads (the next block travels
//      from global memory into registers meanwhile: the reflectors were written by the previous launch and are several
//      hundred cycles away, and every wave needs all of V_b).  Wave w owns columns 16 w .. 16 w + 15 of Q for ALL rows, so
//      Y = V_b Q, Y <- T_b Y and Q -= V_b^T Y of its strip are local to the wave: nothing is computed twice and the panel Y
//      lives in accumulator registers (the k order of an MFMA chain is free: lane group g = lane >> 4 supplies the rows
//      4 g + s its accumulator registers hold; in Y = V_b Q it takes the columns c0 + 16 (g & 1) + 8 (g >> 1) + 4 p + s --
//      16-byte LDS reads, and the two groups of a half-wave read Q rows 16 apart: no bank conflicts).
template <int NT>
__device__ __forceinline__ void back_transform_lean(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws,
                                                    const float* __restrict__ R, int ldr, const float* __restrict__ tau,
                                                    float* __restrict__ Tws, float* __restrict__ strips = nullptr, int wg = 0) {
  // strips != nullptr (few large matrices, several workgroups per matrix): wave w of workgroup wg owns the ONE strip
  // kWaves wg + w and keeps it in LDS (strips + w * DP * 16, row stride 16) for the whole back-transformation -- loaded from Q
  // here, left there for the caller to store; every workgroup forms the T factors for itself (its own Tws).
  constexpr int DP = NT * 32, LD = DP + 1, SV = DP + 4, NQ = DP / 16;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int tid = opaque_v(threadIdx.x), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (opaque: see the level loop)
  const int l16 = lane & 15, g = lane >> 4;
  const int nr = n - 2;
  if (nr <= 0) return;
  const int nblk = (nr + kRB - 1) / kRB;  // <= DP / 16 <= kWaves
  const bool vec = ((ldr & 3) == 0) && ((n & 3) == 0) && ((reinterpret_cast<size_t>(R) & 15) == 0);
  UGLAD_STAMP(ws, 42);
  // ---- 1. Gram matrices
  for (int b = wv; b < nblk; b += kWaves) {
    const int kb = kRB * b;
    f4 x[NQ];  // all fragments of the block requested at once
#pragma unroll
    for (int q = 0; q < NQ; ++q) x[q] = load_reflector4(R, ldr, kb + l16, kb + 16 * q + 4 * g, nr, n, vec);
    f32x4 ga = {0.f, 0.f, 0.f, 0.f}, gb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (kb + 16 * q < DP) {
        ga = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].x, x[q].x, ga, 0, 0, 0);
        gb = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].y, x[q].y, gb, 0, 0, 0);
        ga = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].z, x[q].z, ga, 0, 0, 0);
        gb = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q].w, x[q].w, gb, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // G[4 g + r][l16]
      const int j = 4 * g + r;
      if (l16 >= gtri_start(j)) ws.gtri[b * kGtriFloats + gtri_off(j) + (l16 - gtri_start(j))] = ga[r] + gb[r];
    }
  }
  if (tid < DP) ws.e[tid] = (tid < nr) ? tau[tid] : 0.f;  // (e is dead after the divide & conquer: now the reflector scalars)
  __syncthreads();
  UGLAD_STAMP(ws, 44);
  // ---- 2. triangular factors
  if (tid < kRB * nblk) {
    const int b = tid >> 4, c = tid & 15;
    const float* G = ws.gtri + b * kGtriFloats;
    float y[kRB];
#pragma unroll
    for (int j = kRB - 1; j >= 0; --j) {
      float a0 = (j == c) ? 1.f : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int q = (j + 1) / 4; q < kRB / 4; ++q) {
        const f4 g4 = *reinterpret_cast<const f4*>(&G[gtri_off(j) + 4 * q - gtri_start(j)]);
        if (4 * q + 0 > j) a0 = fmaf(-g4.x, y[4 * q + 0], a0);
        if (4 * q + 1 > j) a1 = fmaf(-g4.y, y[4 * q + 1], a1);
        if (4 * q + 2 > j) a2 = fmaf(-g4.z, y[4 * q + 2], a2);
        if (4 * q + 3 > j) a3 = fmaf(-g4.w, y[4 * q + 3], a3);
      }
      y[j] = ws.e[kRB * b + j] * ((a0 + a1) + (a2 + a3));
    }
#pragma unroll
    for (int j = 0; j < kRB; ++j) Tws[b * 256 + j * 16 + c] = y[j];
  }
  __syncthreads();
  UGLAD_STAMP(ws, 45);
  // ---- 3. the blocks
  float* sV = ws.stage;
  float* sT = ws.stage + kRB * SV;
  // this thread's share of a staged block: one 16-byte piece of V_b (row sr, columns 4 sc ..) and, for the first 64 threads,
  // one of T_b
  constexpr int kPieces = kRB * (DP / 4);                       // 16-byte pieces of a block
  constexpr int kPPT = (kPieces + kThreads - 1) / kThreads;     // per thread: 1 up to D = 128, 2 beyond
  struct Piece {
    f4 v[kPPT], t;
  };
  auto fetch_block = [&](int b, Piece& pc) {
#pragma unroll
    for (int q = 0; q < kPPT; ++q) {
      const int idx = tid + kThreads * q, sr = idx / (DP / 4), sc = idx - sr * (DP / 4);
      pc.v[q] = (idx < kPieces) ? load_reflector4(R, ldr, kRB * b + sr, 4 * sc, nr, n, vec) : f4{0.f, 0.f, 0.f, 0.f};
    }
    pc.t = (tid < 64) ? *reinterpret_cast<const f4*>(Tws + b * 256 + 4 * tid) : f4{0.f, 0.f, 0.f, 0.f};
  };
  Piece pc, pc2;  // this block's pieces and the next one's (two blocks in flight)
  fetch_block(nblk - 1, pc);
  pc2 = pc;
  if (nblk > 1) fetch_block(nblk - 2, pc2);
  const int goff = 16 * (g & 1) + 8 * (g >> 1);
  const int my_strip = kWaves * wg + wv;
  float* const sQw = strips ? strips + (size_t)wv * DP * 16 : nullptr;
  if (strips && 16 * my_strip < DP) {
    for (int r0 = 0; r0 < DP; r0 += 16) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = Q[(r0 + 4 * u + g) * LD + 16 * my_strip + l16];
#pragma unroll
      for (int u = 0; u < 4; ++u) sQw[(r0 + 4 * u + g) * 16 + l16] = v[u];
    }
  }
  for (int b = nblk - 1; b >= 0; --b) {
    const int kb = kRB * b;
#pragma unroll
    for (int q = 0; q < kPPT; ++q) {
      const int idx = tid + kThreads * q, sr = idx / (DP / 4), sc = idx - sr * (DP / 4);
      if (idx < kPieces) *reinterpret_cast<f4*>(sV + sr * SV + 4 * sc) = pc.v[q];
    }
    if (tid < 64) *reinterpret_cast<f4*>(sT + (tid >> 2) * 20 + 4 * (tid & 3)) = pc.t;
    __syncthreads();
    pc = pc2;
    if (b > 1) fetch_block(b - 2, pc2);  // travels while this block and the next are applied
    for (int strip = strips ? my_strip : wv; 16 * strip < DP; strip += strips ? DP : kWaves) {  // (one strip per wave up to D = 128)
      const int colw = strip * 16 + l16;
      float* const Qc = strips ? sQw + l16 : Q + colw;  // this lane's column of the strip
      const int qs = strips ? 16 : LD;                   // ... and its row stride
      // Y = V_b Q over the columns from kb on (in chunks of 32 from the multiple of 32 below kb: the reflectors are zero there)
      f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = {0.f, 0.f, 0.f, 0.f};
      for (int c0 = kb & ~31; c0 < DP; c0 += 32) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int c = c0 + goff + 4 * pp;
          const f4 va = *reinterpret_cast<const f4*>(sV + l16 * SV + c);
          const float* qb = Qc + c * qs;
          const float q0 = qb[0], q1 = qb[qs], q2 = qb[2 * qs], q3 = qb[3 * qs];
          ya = __builtin_amdgcn_mfma_f32_16x16x4f32(va.x, q0, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_16x16x4f32(va.y, q1, yb, 0, 0, 0);
          ya = __builtin_amdgcn_mfma_f32_16x16x4f32(va.z, q2, ya, 0, 0, 0);
          yb = __builtin_amdgcn_mfma_f32_16x16x4f32(va.w, q3, yb, 0, 0, 0);
        }
      }
      // Y <- T_b Y: B = the accumulator itself (lane group g holds rows 4 g + s), A[r'][k] = T_b[r'][4 g + s]
      const f4 t4 = *reinterpret_cast<const f4*>(sT + l16 * 20 + 4 * g);
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.x, ya[0] + yb[0], z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.y, ya[1] + yb[1], z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.z, ya[2] + yb[2], z, 0, 0, 0);
      z = __builtin_amdgcn_mfma_f32_16x16x4f32(t4.w, ya[3] + yb[3], z, 0, 0, 0);
      // Q(rows i0 .. i0+15, strip) -= V_b(:, rows)^T Y for every 16-row tile from kb on; A[i][k] = V_b[4 g + s][i0 + i]
      const float* va0 = sV + (4 * g) * SV + l16;
      for (int i0 = kb; i0 < DP; i0 += 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[i0], z[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[SV + i0], z[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[2 * SV + i0], z[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(va0[3 * SV + i0], z[3], acc, 0, 0, 0);
        float* qo = Qc + (i0 + 4 * g) * qs;
#pragma unroll
        for (int r = 0; r < 4; ++r) qo[r * qs] -= acc[r];
      }
    }
    __syncthreads();  // the staging area is free again (and, after the last block, Q is complete)
    UGLAD_STAMP(ws, 46 + b);
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

// The same in two launches with the last merge in between (few large matrices): front = everything up to the prepared last
// merge (n > 128: there is one), back = the back-transformation of the merged eigenvectors.
template <int NT>
__device__ __forceinline__ void symeig_lean_front(float* __restrict__ Q, int n, LeanScratch<NT * 32>& ws,
                                                  const float* __restrict__ tri, float* __restrict__ last) {
  constexpr int DP = NT * 32;
  for (int i = threadIdx.x; i < DP; i += kThreads) {
    ws.d[i] = (i < n) ? tri[i] : 0.f;
    ws.e[i] = (i < n) ? tri[DP + i] : 0.f;
  }
  __syncthreads();
  dc_tridiagonal_lean<NT>(Q, n, ws, last);
}

}  // namespace uglad
