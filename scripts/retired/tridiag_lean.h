// EXPERIMENT, measured and rejected in round 3 (profiles/r03_tridiag_experiments.txt): 412 us per launch at M = 1024, D = 128 against the
// shipped tridiag_kernel's 310 us (ONE workgroup alone: 239 vs 194 us).  The premise below -- the step is bound by LDS round trips in
// series -- was wrong: with the round trips cut from twelve to four per step the chain still takes 2170 cycles and the sweep 1776
// (old: 2020 / 1540).  A wave issues one instruction every ~5 cycles whatever it is; the step is bound by the NUMBER of instructions
// on its critical wave (chain ~250 + sweep ~170), and this form has more of them (four rows per chain lane: every elementwise
// operation four times; copies at the switch entry).  Not compiled by anything; it was wired into LAUNCH_TRIDIAG (glad_kernels.hip)
// as tridiag_lean_kernel<NT> for NT <= 4 behind UGLAD_TRIDIAG_LEAN.
//
// Householder tridiagonalisation for D <= 128, latency-lean (round 3).  Same algorithm, same outputs and the same resources per workgroup
// as tridiag_kernel (tridiag.h: matrix in registers, 64 VGPRs, ~37 KB of LDS, four workgroups per CU) -- what changed is the length of
// the dependent chain of ONE step, which is what the kernel is bound by: measured on MI355X, ONE workgroup alone on the chip needs
// 235 us for D = 128 (2000 cycles per step in the reflector chain of wave 0, 1530 in the sweep), and 1024 of them, four per CU, 325 us
// (profiles/r03_tridiag_stamps.txt).  Co-residency is nearly free; the time is the latency of a step.  That latency was LDS round
// trips in series: the chain summed its 16 partial products per row in four load-wait-add rounds, the sweep fetched the three column
// scalars of every column slot right before using them, eight round trips per sweep.  Here
//   * chain: lane l of wave 0 owns the four consecutive rows of row group l / LPR; the per-row partial sums are laid out so that a lane
//     fetches its eight float4 partials with eight 16-byte loads issued together (conflict-free), the LPR lanes of a row group combine on
//     the DPP path, and every vector of the chain (v, w, the exported column, the next reflector) moves as one float4 per lane;
//   * sweep: the column scalars {v_k[c], w_k[c], v_{k+1}[c]} travel as ONE 16-byte record per column, fetched two slots ahead of their
//     use; a wave enters the unrolled slot sequence at its first live slot through one switch, so no slot carries a test of its own.
#pragma once
#include "../../uglad_amd/csrc/tridiag.h"

namespace uglad {

#ifdef UGLAD_SIMT_EMUL
#define UGLAD_OPAQUE_V(x) ((void)0)
#else
#define UGLAD_OPAQUE_V(x) asm volatile("" : "+v"(x))  // the compiler must assume x changed: nothing derived from it is loop-invariant
#endif

template <int NT>
__global__ __launch_bounds__(kThreads, 8) void tridiag_lean_kernel(const float* __restrict__ A0, const float* __restrict__ A1,
                                                                   const float* __restrict__ lam_ptr, float* __restrict__ Rbase,
                                                                   float* __restrict__ tri_base, int D, int gs) {
  static_assert(kThreads == 512 && NT >= 1 && NT <= 4, "thread -> matrix map of the lean tridiagonalisation");
  constexpr int DP = NT * 32;                 // stride of the outputs d, e, tau
  constexpr int RG = NT >= 3 ? 32 : NT * 8;   // row groups of four rows
  constexpr int DPe = 4 * RG;                 // rows (and columns) the thread grid covers: 32, 64, 128
  constexpr int NCG = kThreads / RG;          // column groups: thread (r4, cg) holds columns cg, cg + NCG, ...
  constexpr int NC = DPe / NCG > 0 ? DPe / NCG : 1;  // column slots per thread: 1, 2, 8
  constexpr int NCOL = NCG * NC;              // columns addressable by (cg, slot) (64 for NT = 1: the upper half holds zeros)
  constexpr int LPR = 64 / RG;                // chain: lanes of wave 0 per row group (they hold the same four rows)
  constexpr int NPL = NCG / LPR;              // chain: partial sums (float4) per lane = 8
#ifndef UGLAD_TL_NL
#define UGLAD_TL_NL 3
#endif
#ifndef UGLAD_TL_PF
#define UGLAD_TL_PF 1
#endif
  constexpr int NL = NT >= 3 ? UGLAD_TL_NL : 0;  // column slots kept in thread-private LDS instead of registers (64-VGPR budget)
  constexpr int PF = UGLAD_TL_PF;                // column records fetched this many slots ahead of their use (1 or 2)
  constexpr int PSTR = NCG + 1;               // float4 stride between the row groups of the partial sums (bank spread)
  static_assert(NPL == 8, "eight partial sums per chain lane");
  __shared__ f4 s_a[NL > 0 ? NL : 1][NL > 0 ? kThreads : 1];
  __shared__ f4 s_trip[NCOL];                 // per column c: {v_k[c], w_k[c], v_{k+1}[c], -}
  __shared__ __attribute__((aligned(16))) float s_v[2][DPe], s_w[DPe], s_col[2][DPe];
  __shared__ f4 s_part[RG * PSTR];            // [row group][column group]: partial products A v' of the last sweep
  __shared__ __attribute__((aligned(16))) float s_dotp[8];
  __shared__ float s_corner, s_tau;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wv_u = __builtin_amdgcn_readfirstlane(wv);
  const int r4 = tid % RG, cg = tid / RG;
  const int cgmax = (wv_u * 64 + 63) / RG;    // largest column group held by this wave
  const int n = D;
  const size_t base = (size_t)blockIdx.x * D * D;
  float* R = Rbase + base;
  float* tri = tri_base + (size_t)blockIdx.x * 3 * DP;
  const float inv_lam = lam_ptr ? 1.0f / lam_ptr[blockIdx.x / gs] : 1.0f;
  const bool vec4 = ((D & 3) == 0) && ((reinterpret_cast<size_t>(Rbase) & 15) == 0);

  // ---- load (row c of the symmetric A as column c: 16 bytes per lane, RG lanes per row)
  f4 a[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = cg + NCG * i;
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * r4 + q;
      float v = 0.f;
      if (c < n && r < n) {
        v = A0[base + (size_t)c * D + r];
        if (A1) v = fmaf(inv_lam, v, -A1[base + (size_t)c * D + r]);
      }
      t[q] = v;
    }
    a[i] = {t[0], t[1], t[2], t[3]};
    if (i < NL) s_a[i < NL ? i : 0][tid] = a[i];
  }
  const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < NCOL; i += kThreads) s_trip[i] = zero4;
  for (int i = tid; i < DPe; i += kThreads) {
    s_v[0][i] = 0.f;
    s_v[1][i] = 0.f;
    s_w[i] = 0.f;
    s_col[0][i] = 0.f;
    s_col[1][i] = 0.f;
  }
  for (int i = tid; i < DP; i += kThreads) {
    tri[i] = 0.f;
    tri[DP + i] = 0.f;
    tri[2 * DP + i] = 0.f;
  }
  for (int i = tid; i < RG * PSTR; i += kThreads) s_part[i] = zero4;
  if (tid < 8) s_dotp[tid] = 0.f;
  __syncthreads();
  if (cg == 0) *reinterpret_cast<f4*>(&s_col[0][4 * r4]) = a[0];  // column 0
  if (r4 == ((n - 1) >> 2)) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (cg + NCG * i == n - 1) s_corner = f4_elem(a[i], (n - 1) & 3);
  }
  __syncthreads();
  if (n == 1) {
    if (tid == 0) tri[0] = s_corner;
    return;
  }

  // value of row r of a chain vector (one float4 per lane, row group = lane / LPR) in every lane
  auto elem = [&](const f4& x, int r) -> float { return bcast_lane(f4_elem(x, r & 3), LPR * (r >> 2)); };

  int cv = 0, cur = 0;  // s_v[cv] = v_k; s_col[cur] = the column exported by the last sweep
  float tau_k = 0.f;
  for (int k = -1; k <= n - 3; ++k) {
    const int k1 = k + 1;
    {
    TSTAMP_BEGIN();
    if (wv_u == 0) {
      if (UGLAD_TRIDIAG_PRIO) __builtin_amdgcn_s_setprio(3);
      // (lane-derived addresses and predicates are re-derived every step from an opaque copy of the lane number: hoisted out of the
      // loop they would be spilled to scratch by the 64-register budget, and a scratch reload costs more than recomputing them)
      int ln = lane;
      UGLAD_OPAQUE_V(ln);
      const int g = ln / LPR, h = ln % LPR;
      // ---- finish step k: p = tau A v (partials of the last sweep), w = p - (tau/2)(p.v) v
      // (two batches of four 16-byte loads: eight at once plus the matrix do not fit the 64 registers of a wave)
      const f4* pp = &s_part[g * PSTR + h * NPL];
      float p[4];
      {
        const f4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
        p[0] = (p0.x + p1.x) + (p2.x + p3.x);
        p[1] = (p0.y + p1.y) + (p2.y + p3.y);
        p[2] = (p0.z + p1.z) + (p2.z + p3.z);
        p[3] = (p0.w + p1.w) + (p2.w + p3.w);
      }
      __builtin_amdgcn_sched_barrier(0);
      const f4 p4 = pp[4], p5 = pp[5], p6 = pp[6], p7 = pp[7];
      const f4 vv = *reinterpret_cast<const f4*>(&s_v[cv][4 * g]);
      const f4 col = *reinterpret_cast<const f4*>(&s_col[cur][4 * g]);
      const f4 d0 = *reinterpret_cast<const f4*>(&s_dotp[0]), d1 = *reinterpret_cast<const f4*>(&s_dotp[4]);
      p[0] += (p4.x + p5.x) + (p6.x + p7.x);
      p[1] += (p4.y + p5.y) + (p6.y + p7.y);
      p[2] += (p4.z + p5.z) + (p6.z + p7.z);
      p[3] += (p4.w + p5.w) + (p6.w + p7.w);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // the LPR lanes of a row group hold disjoint column groups: combine
        if (LPR >= 2) p[q] += dpp_move<0xb1>(p[q]);   // quad_perm:[1,0,3,2]
        if (LPR >= 4) p[q] += dpp_move<0x4e>(p[q]);   // quad_perm:[2,3,0,1]
        if (LPR == 8) p[q] += dpp_move<0x141>(p[q]);  // row_half_mirror
      }
      const float vAv = ((d0.x + d0.y) + (d0.z + d0.w)) + ((d1.x + d1.y) + (d1.z + d1.w));
      const float alpha = 0.5f * tau_k * tau_k * vAv;
      const float vq[4] = {vv.x, vv.y, vv.z, vv.w}, cq[4] = {col.x, col.y, col.z, col.w};
      float wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * g + q;
        const float pt = (r > k && r < n) ? p[q] * tau_k : 0.f;
        wq[q] = pt - alpha * vq[q];
      }
      const f4 w = {wq[0], wq[1], wq[2], wq[3]};
      // ---- look ahead: row k1 after update k = exported row (after update k-1) - v[k1] w - w[k1] v
      const float w_k1 = elem(w, k1), v_k1 = elem(vv, k1);
      float xq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xq[q] = cq[q] - v_k1 * wq[q] - w_k1 * vq[q];  // (rows >= n: all three terms are zero)
      const f4 x = {xq[0], xq[1], xq[2], xq[3]};
      const float dk1 = elem(x, k1);
      if (k1 <= n - 3) {
        const int c0 = k1 + 1;
        const float x0 = elem(x, c0);
        float sig = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * g + q;
          if (h == 0 && r > c0 && r < n) sig = fmaf(xq[q], xq[q], sig);
        }
        sig = wave_sum(sig);
        float beta = x0, tau1 = 0.f, sc = 0.f;
        if (sig > 0.f) {
          // hardware square root (1 ulp) and rcp + Newton divisions; tau and the scaling come from the same rounded beta, so H stays
          // orthogonal to rounding error.  (Tiny norms: library path.)
          const float nrm2 = fmaf(x0, x0, sig);
          if (nrm2 > 1e-30f) {
            beta = -copysignf(__builtin_amdgcn_sqrtf(nrm2), x0);
            const float dd = x0 - beta;  // |dd| >= |x0|: no cancellation
            sc = div_acc(1.0f, dd);
            tau1 = -dd * div_acc(1.0f, beta);
          } else {
            beta = -copysignf(sqrtf(nrm2), x0);
            tau1 = (beta - x0) / beta;
            sc = 1.0f / (x0 - beta);
          }
        }
        float nq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 4 * g + q;
          nq[q] = (r == c0) ? 1.f : ((r > c0 && r < n) ? xq[q] * sc : 0.f);
        }
        const f4 vn = {nq[0], nq[1], nq[2], nq[3]};
        if (h == 0) {
          *reinterpret_cast<f4*>(&s_v[cv ^ 1][4 * g]) = vn;
          *reinterpret_cast<f4*>(&s_w[4 * g]) = w;
          if (vec4) {
            if (4 * g < n) *reinterpret_cast<f4*>(&R[(size_t)k1 * D + 4 * g]) = vn;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (4 * g + q < n) R[(size_t)k1 * D + 4 * g + q] = nq[q];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)  // the column records of rows 4 g .. 4 g + 3, dealt to the lanes of the row group
          if (h == (q * LPR) / 4) s_trip[4 * g + q] = {vq[q], wq[q], nq[q], 0.f};
        if (ln == 0) {
          tri[k1] = dk1;
          tri[DP + k1] = beta;
          tri[2 * DP + k1] = tau1;
          s_tau = tau1;
        }
      } else {
        // k1 == n-2: the trailing 2x2 block
        const float e_last = elem(x, n - 1), w_n1 = elem(w, n - 1), v_n1 = elem(vv, n - 1);
        if (ln == 0) {
          tri[k1] = dk1;
          tri[DP + k1] = e_last;
          tri[n - 1] = s_corner - 2.f * v_n1 * w_n1;
        }
      }
      if (UGLAD_TRIDIAG_PRIO) __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    TSTAMP_ADD(0);
    }
    if (k1 > n - 3) break;
    tau_k = s_tau;
    TSTAMP_BEGIN();
    // ---- sweep: rank-2 update in registers + partial products with the next reflector
    {
      int tq = tid;
      UGLAD_OPAQUE_V(tq);
      const int r4 = tq % RG, cg = tq / RG;  // (shadow the hoisted copies: see the chain)
      const f4 v4 = *reinterpret_cast<const f4*>(&s_v[cv][4 * r4]);
      const f4 w4 = *reinterpret_cast<const f4*>(&s_w[4 * r4]);
      f4 acc = zero4;
      // slot i of this wave is live while some column cg' + NCG i of it (cg' <= cgmax) is still in the trailing matrix (> k1)
      const int ifirst = (k1 < cgmax) ? 0 : (k1 - cgmax) / NCG + 1;
      const int k2 = k1 + 1, i2 = k2 / NCG, cgo = k2 - i2 * NCG;             // owners of column k2: exported for the next chain
      const int ic = (n - 1) / NCG, cgc = (n - 1) - ic * NCG;                 // owner of the corner (read after the last sweep)
      const bool last = (k1 == n - 3);
      if (ifirst < NC) {
        f4 pre0 = s_trip[cg + NCG * ifirst];
        f4 pre1 = zero4;
        if (PF == 2) pre1 = s_trip[cg + NCG * (ifirst + 1 < NC ? ifirst + 1 : NC - 1)];
        f4 tpre = zero4;
        if (NL > 0 && ifirst < NL) tpre = s_a[ifirst < NL ? ifirst : 0][tq];
#define UGLAD_TRI_SLOT(I)                                                                                                     \
  case I:                                                                                                                     \
    if (I < NC) {                                                                                                             \
      const f4 tr = pre0;                                                                                                     \
      if (PF == 2) {                                                                                                          \
        pre0 = pre1;                                                                                                          \
        if (I + 2 < NC) pre1 = s_trip[cg + NCG * (I + 2 < NC ? I + 2 : 0)];                                                   \
      } else if (I + 1 < NC) {                                                                                                \
        pre0 = s_trip[cg + NCG * (I + 1 < NC ? I + 1 : 0)];                                                                   \
      }                                                                                                                       \
      f4 t;                                                                                                                   \
      if (I < NL) {                                                                                                           \
        t = tpre;                                                                                                             \
        if (I + 1 < NL) tpre = s_a[I + 1 < NL ? I + 1 : 0][tq];                                                              \
      } else {                                                                                                                \
        t = a[I < NC ? I : 0];                                                                                                \
      }                                                                                                                       \
      t.x = t.x - tr.x * w4.x - tr.y * v4.x;                                                                                  \
      t.y = t.y - tr.x * w4.y - tr.y * v4.y;                                                                                  \
      t.z = t.z - tr.x * w4.z - tr.y * v4.z;                                                                                  \
      t.w = t.w - tr.x * w4.w - tr.y * v4.w;                                                                                  \
      acc.x = fmaf(t.x, tr.z, acc.x);                                                                                         \
      acc.y = fmaf(t.y, tr.z, acc.y);                                                                                         \
      acc.z = fmaf(t.z, tr.z, acc.z);                                                                                         \
      acc.w = fmaf(t.w, tr.z, acc.w);                                                                                         \
      if (I < NL) s_a[I < NL ? I : 0][tq] = t;                                                                               \
      else a[I < NC ? I : 0] = t;                                                                                             \
      if (I == i2 && cg == cgo) *reinterpret_cast<f4*>(&s_col[cur ^ 1][4 * r4]) = t;                                          \
      if (last && I == ic && cg == cgc && r4 == ((n - 1) >> 2)) s_corner = f4_elem(t, (n - 1) & 3);                           \
    }
        switch (ifirst) {
          UGLAD_TRI_SLOT(0)
          UGLAD_TRI_SLOT(1)
          UGLAD_TRI_SLOT(2)
          UGLAD_TRI_SLOT(3)
          UGLAD_TRI_SLOT(4)
          UGLAD_TRI_SLOT(5)
          UGLAD_TRI_SLOT(6)
          UGLAD_TRI_SLOT(7)
          default:
            break;
        }
#undef UGLAD_TRI_SLOT
      }
      const f4 n4 = *reinterpret_cast<const f4*>(&s_v[cv ^ 1][4 * r4]);
      float vav = acc.x * n4.x + acc.y * n4.y + acc.z * n4.z + acc.w * n4.w;  // this thread's share of v'.(A v')
      s_part[r4 * PSTR + cg] = acc;
      vav = wave_sum(vav);
      if ((tq & 63) == 0) s_dotp[tq >> 6] = vav;
    }
    __syncthreads();
    TSTAMP_ADD(1);
    cv ^= 1;
    cur ^= 1;
  }
}

}  // namespace uglad
