// Householder tridiagonalisation A = H T H^T of one symmetric matrix per workgroup with the matrix held in REGISTERS.
//
// Thread (r4, cg) owns rows 4*r4 .. 4*r4+3 of columns c = cg, cg + NCG, cg + 2 NCG, ... (one float4 each): 32 floats per thread
// at D = 128 with 512 threads.  A step is (i) a serial chain on wave 0 -- finish w_k, take the reflector of step k+1 one step
// ahead from the column (= row, A is symmetric) the previous sweep exported, wave reductions on the DPP path -- and (ii) one
// sweep by all waves that applies A <- A - v w^T - w v^T in registers and accumulates the partial products A v_{k+1}; the 32
// threads that own column k+2 then export it.  The sweep needs no per-entry test: v and w vanish on the rows and columns
// that have left the trailing matrix, so updating them is a no-op, and whole column slots are skipped per wave.  The only
// LDS traffic is the three vectors and the per-column-group partial sums (~13 KB), so several workgroups fit a CU and the
// chains of one overlap the sweeps of the others (the chain is latency-, not throughput-bound).
//
// Outputs (global): tri[0..DP) = diagonal d, tri[DP..2DP) = off-diagonal e, tri[2DP..3DP) = tau; row k of R = reflector v_k
// (v_k[c] = 0 for c <= k, 1 at c = k+1).  H = H_0 H_1 ... H_{n-3},  H_k = I - tau_k v_k v_k^T.
#pragma once
#include "glad_device.h"

namespace uglad {

#ifdef UGLAD_STAMPS  // diagnostic build: cycles workgroup 0 spends in the reflector chain / in the sweep (scripts/stamp_cell.py)
__device__ unsigned long long g_tstamps[4];
__device__ unsigned long long g_twg[4096][3];  // per workgroup: start, end (s_memtime), hardware id (XCC / SE / CU)
#define TSTAMP_BEGIN() const unsigned long long t_begin_ = __builtin_amdgcn_s_memtime()
#define TSTAMP_ADD(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_tstamps[i] += __builtin_amdgcn_s_memtime() - t_begin_; } while (0)
#else
#define TSTAMP_BEGIN() do {} while (0)
#define TSTAMP_ADD(i) do {} while (0)
#endif

struct alignas(16) f4 {
  float x, y, z, w;
};

__device__ __forceinline__ float f4_elem(const f4& a, int q) { return q == 0 ? a.x : (q == 1 ? a.y : (q == 2 ? a.z : a.w)); }

// v[idx] for a wave-uniform idx without dynamic register indexing
template <int N>
__device__ __forceinline__ float pick(const float (&v)[N], int idx) {
  float r = v[0];
#pragma unroll
  for (int s = 1; s < N; ++s) r = (idx == s) ? v[s] : r;
  return r;
}

// A = A0 (plain symmetric matrix) when A1 == nullptr, else A = A0 / lam - A1 (the GLAD cell's b = S/lam - Z).
// TH threads per workgroup: kThreads everywhere except for FEW matrices beyond D = 128, which get 1024 (four waves per SIMD hide the
// LDS latency of the sweep that two cannot: 755 -> 620 us per launch at D = 256, one workgroup per CU either way).
template <int NT, int TH>
__global__ __launch_bounds__(TH, NT <= 4 ? 8 : (TH > 512 ? 1 : 2)) void tridiag_kernel(const float* __restrict__ A0, const float* __restrict__ A1,
                                                           const float* __restrict__ lam_ptr, float* __restrict__ Rbase,
                                                           float* __restrict__ tri_base, int D, int gs,
                                                           const int* __restrict__ only_flagged) {
  if (only_flagged && only_flagged[blockIdx.x] == 0) return;  // (Theta_0 / loss: the Cholesky kernel has done this matrix; nullptr: all)
  constexpr int DP = NT * 32, RG = DP / 4, NCG = TH / RG, NC = (DP + NCG - 1) / NCG;
  constexpr int NS = (DP > 128) ? (DP + 63) / 64 : 2;  // elements per lane of wave 0 in the chain
  // D = 128: four workgroups must share a CU (1024 matrices on 256 CUs = one round instead of two), i.e. <= 64 VGPRs.  The
  // first NL column slots of every thread -- the columns that leave the trailing matrix first, after at most NL * NCG
  // steps -- therefore live in thread-private LDS slots instead of registers.
  constexpr int NL = (NT == 4 && TH == 512) ? 3 : 0;
  __shared__ f4 s_a[NL > 0 ? NL : 1][NL > 0 ? TH : 1];
  // w, and the reflectors of this and the next step (the latter two swap roles every step: offsets ov / on), in ONE array so
  // that every access of a thread is its own base address plus a wave-uniform offset
  __shared__ __attribute__((aligned(16))) float s_vec[5 * DP];  // [3 DP, 5 DP): the two exported columns
  // Experiment (-DUGLAD_TRIDIAG_ROWWAVES=1, D = 128): a wave holds 16 consecutive ROWS (4 row groups x all 16 column groups)
  // instead of 2 column groups x all rows, so that the waves whose rows have left the trailing matrix skip the sweep
  // altogether.  Measured on MI355X (A/B in one session, M = 1024): the sweeps of a workgroup get shorter (301 k -> 193 k
  // cycles over the 126 steps) but the launch gets LONGER, 0.32 -> 0.38 ms: the work is no longer spread evenly -- the wave
  // with the last rows carries all eight column slots to the very end, where the default layout has every wave down to
  // one slot -- and the prologue reads half cache lines (64-byte pieces of a row per wave).  Off by default.
#ifndef UGLAD_TRIDIAG_ROWWAVES
#define UGLAD_TRIDIAG_ROWWAVES 0
#endif
  // Round 3, measured on MI355X (profiles/r03_tridiag_experiments.txt):
  //  UGLAD_TRIDIAG_PRIO: the chain wave raises its priority (s_setprio 3) for the serial reflector chain.  The chain is ~100 dependent
  //    instructions on ONE wave that shares its SIMD with seven sweeping waves of the co-resident workgroups; at equal priority every one
  //    of them waits its turn in the issue arbitration.
  //  UGLAD_TRIDIAG_PAIRSUM: the two column groups of a wave (lanes l and l ^ 32 hold the same rows) add their partial products in the
  //    sweep (v_permlane32_swap), so the chain gathers NCG / 2 instead of NCG partial sums per row.
#ifndef UGLAD_TRIDIAG_PRIO
#define UGLAD_TRIDIAG_PRIO 1
#endif
#ifndef UGLAD_TRIDIAG_PAIRSUM
#define UGLAD_TRIDIAG_PAIRSUM 0
#endif
// UGLAD_TRIDIAG_LANE (round 4, profiles/r04_tridiag_lane_experiment.txt): at D = 128 the kernel is held to 64 registers and spills `lane`
// -- one 8-byte scratch reload per step on every wave.  Taking the lane index from the hardware instead removes the spill and is SLOWER on a
// same-box A/B (variant 1: 310.8 us, variant 2: 303.9 us, the spilling variant 0: 301.8 us per launch, three alternating runs each): the
// reload is issued right after the sweep and is back before the next chain needs it, while the fresh value costs the chain -- the critical
// path -- instructions (tests the compiler had hoisted into scalar masks are evaluated in every step).  0 stays.
#ifndef UGLAD_TRIDIAG_LANE
#define UGLAD_TRIDIAG_LANE 0
#endif
  // UGLAD_TRIDIAG_DPPSUM (late round 4, D = 128): the 16 column groups of a row group sit in the 16 lanes of one DPP row (cg = lane & 15, row group
  // = wave + 8 (lane >> 4): rows dealt out cyclically, every wave keeps work to the end), so the sweep adds its partial products up across the
  // column groups in registers (four DPP steps per component) and publishes ONE total per row -- the chain's gather of 16 partial sums per row,
  // ~60 of its ~280 instructions, is gone; no spill at 64 VGPRs.  Measured on MI355X, same box, three alternating rounds
  // (profiles/r04_tridiag_dppsum_ab.txt): 0.340 ms per launch against 0.300, the step 26.9 ms against 25.8 -- SLOWER by 760 cycles per reflector,
  // far more than the 16 DPP adds at the sweep's tail: what the chain saves, every one of the eight waves pays before the barrier the chain
  // waits at (the same outcome as UGLAD_TRIDIAG_PAIRSUM in round 3).  Off.
#ifndef UGLAD_TRIDIAG_DPPSUM
#define UGLAD_TRIDIAG_DPPSUM 0
#endif
  constexpr bool kDppSum = UGLAD_TRIDIAG_DPPSUM && !UGLAD_TRIDIAG_ROWWAVES && (NT == 4 && TH == 512);
  constexpr bool kRowWaves = UGLAD_TRIDIAG_ROWWAVES && (NT == 4 && TH == 512);
  constexpr bool kPairSum = UGLAD_TRIDIAG_PAIRSUM && !kRowWaves && RG == 32 && (NCG % 2 == 0);  // (a wave = two column groups x 32 row groups)
  constexpr int NPG = kDppSum ? 1 : (kPairSum ? NCG / 2 : NCG);  // partial sums per row the chain gathers
  constexpr int PS = kRowWaves ? DP + 4 : DP;  // row stride of the partial sums
  __shared__ __attribute__((aligned(16))) float s_part[kRowWaves ? (TH / (DP / 4)) * (DP + 4) : TH * 4];
  __shared__ float s_dotp[(TH / 64)];
  __shared__ float s_corner, s_tau;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wv_u = __builtin_amdgcn_readfirstlane(wv);  // (provably wave-uniform: s_setprio ignores EXEC and needs a scalar branch around it)
#ifdef UGLAD_STAMPS
  if (tid == 0 && blockIdx.x < 4096) {
    g_twg[blockIdx.x][0] = __builtin_amdgcn_s_memtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_twg[blockIdx.x][2] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int r4 = kDppSum ? wv + 8 * (lane >> 4) : (kRowWaves ? 4 * wv + (lane & 3) : tid % RG);
  const int cg = kDppSum ? (lane & 15) : (kRowWaves ? lane >> 2 : tid / RG);
  const int cgw = (kRowWaves || kDppSum) ? NCG - 1 : (__builtin_amdgcn_readfirstlane(wv) * 64 + 63) / RG;
  const int cgmax = cgw < NCG - 1 ? cgw : NCG - 1;  // largest column group held by this wave
  const int row_last = kRowWaves ? 16 * __builtin_amdgcn_readfirstlane(wv) + 15 : DP;  // last row held by this wave
  const int n = D;
  const size_t base = (size_t)blockIdx.x * D * D;
  float* R = Rbase + base;
  float* tri = tri_base + (size_t)blockIdx.x * 3 * DP;
  const float inv_lam = lam_ptr ? 1.0f / lam_ptr[blockIdx.x / gs] : 1.0f;  // gs matrices share one lambda (one group)

  // ---- load: coalesced along the rows of A (32 lanes x 16 bytes per column)
  f4 a[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = cg + NCG * i;
    float t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * r4 + q;
      float v = 0.f;
      if (cg < NCG && c < n && r < n) {
        // (kDppSum: consecutive lanes hold consecutive COLUMNS -- read the mirror entry, 16 lanes = 64 contiguous bytes; the matrix is symmetric)
        const size_t at = kDppSum ? base + (size_t)r * D + c : base + (size_t)c * D + r;
        v = A0[at];
        if (A1) v = fmaf(inv_lam, v, -A1[at]);
      }
      t[q] = v;
    }
    a[i] = {t[0], t[1], t[2], t[3]};
    if (i < NL) s_a[i < NL ? i : 0][tid] = a[i];
  }
  for (int i = tid; i < DP; i += TH) {
    s_vec[i] = 0.f;
    s_vec[DP + i] = 0.f;
    s_vec[2 * DP + i] = 0.f;
    tri[i] = 0.f;
    tri[DP + i] = 0.f;
    tri[2 * DP + i] = 0.f;
  }
  for (int i = tid; i < (int)(sizeof(s_part) / sizeof(float)); i += TH) s_part[i] = 0.f;
  if (tid < (TH / 64)) s_dotp[tid] = 0.f;
  // export column 0 (entries A[0][r]) and the corner A[n-1][n-1]
  if (cg == 0) *reinterpret_cast<f4*>(&s_vec[3 * DP + 4 * r4]) = a[0];
  if (cg < NCG && r4 == ((n - 1) >> 2)) {
#pragma unroll
    for (int i = 0; i < NC; ++i)
      if (cg + NCG * i == n - 1) s_corner = f4_elem(a[i], (n - 1) & 3);
  }
  __syncthreads();
  if (n == 1) {
    if (tid == 0) tri[0] = s_corner;
    return;
  }

#ifdef UGLAD_STAMPS
  if (tid == 0 && blockIdx.x == 0) g_tstamps[2] += __builtin_amdgcn_s_memtime() - g_twg[0][0];  // prologue (load + set-up)
#endif
  int ov = DP, on = 2 * DP;  // offsets of v_k and v_{k+1} in s_vec (w sits at 0)
  float tau_k = 0.f;
  int cur = 0;
  for (int k = -1; k <= n - 3; ++k) {
    const int k1 = k + 1;
    {
    TSTAMP_BEGIN();
    if (wv_u == 0) {
      if (UGLAD_TRIDIAG_PRIO) __builtin_amdgcn_s_setprio(3);
      // (at D = 128 the kernel is held to 64 registers and `lane`, live across the sweep, was spilled: one scratch reload per step on every
      // wave in front of the step's second barrier.  UGLAD_TRIDIAG_LANE: 0 = that; 1 = the lane index taken from the hardware here, live only
      // inside the chain -- but then the tests `lane < n`, `lane == 0`, ..., which the compiler had hoisted into scalar masks, are evaluated
      // in every step; 2 = from the hardware for what depends on the step, the outer one for the loop-invariant tests)
#if UGLAD_TRIDIAG_LANE == 1
      const int lane = lane_now();
      const int lane_k = lane;
#elif UGLAD_TRIDIAG_LANE == 2
      const int lane_k = lane_now();
#else
      const int lane_k = lane;
#endif
      // ---- finish step k: p = tau A v, w = p - (tau/2)(p.v) v; v.(A v) was reduced per wave at the end of the last sweep
      float pv[NS], vv[NS], wl[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int rr = lane_k + 64 * s;
        float p = 0.f;
        if (rr > k && lane + 64 * s < n) {
#pragma unroll 8
          for (int g = 0; g < NPG; ++g) p += s_part[g * PS + rr];
          p *= tau_k;
        }
        pv[s] = p;
        vv[s] = (lane + 64 * s < DP) ? s_vec[ov + rr] : 0.f;
      }
      float vAv2[2] = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < (TH / 64); ++q) vAv2[q & 1] += s_dotp[q];
      const float vAv = vAv2[0] + vAv2[1];
      const float alpha = 0.5f * tau_k * tau_k * vAv;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int rr = lane_k + 64 * s;
        wl[s] = pv[s] - alpha * vv[s];
        if (lane + 64 * s < DP) s_vec[rr] = wl[s];
      }
      // ---- look ahead: row k1 after update k = exported row (after update k-1) - v[k1] w - w[k1] v
      const float w_k1 = bcast_lane(pick(wl, k1 >> 6), k1 & 63);
      const float v_k1 = bcast_lane(pick(vv, k1 >> 6), k1 & 63);
      float x[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = lane_k + 64 * s;
        x[s] = (lane + 64 * s < n) ? (s_vec[(3 + cur) * DP + c] - v_k1 * wl[s] - w_k1 * vv[s]) : 0.f;
      }
      const float dk1 = bcast_lane(pick(x, k1 >> 6), k1 & 63);
      if (k1 <= n - 3) {
        const int c0 = k1 + 1;
        const float x0 = bcast_lane(pick(x, c0 >> 6), c0 & 63);
        float sig = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int c = lane_k + 64 * s;
          if (c > c0 && lane + 64 * s < n) sig = fmaf(x[s], x[s], sig);
        }
        sig = wave_sum(sig);
        float beta = x0, tau1 = 0.f, sc = 0.f;
        if (sig > 0.f) {
          // on the serial path of every step: hardware square root (1 ulp) and rcp + Newton divisions; tau and the scaling
          // come from the same rounded beta, so H stays orthogonal to rounding error.  (Tiny norms: library path.)
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
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int c = lane_k + 64 * s;
          if (lane + 64 * s < DP) {
            float vc = 0.f;
            if (c == c0) vc = 1.f;
            else if (c > c0 && lane + 64 * s < n) vc = x[s] * sc;
            s_vec[on + c] = vc;
            if (lane + 64 * s < n) R[(size_t)k1 * D + c] = vc;
          }
        }
        if (lane == 0) {
          tri[k1] = dk1;
          tri[DP + k1] = beta;
          tri[2 * DP + k1] = tau1;
          s_tau = tau1;
        }
      } else {
        // k1 == n-2: the trailing 2x2 block
        const float e_last = bcast_lane(pick(x, (n - 1) >> 6), (n - 1) & 63);
        const float w_n1 = bcast_lane(pick(wl, (n - 1) >> 6), (n - 1) & 63);
        const float v_n1 = bcast_lane(pick(vv, (n - 1) >> 6), (n - 1) & 63);
        if (lane == 0) {
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
    float vav = 0.f;
    const int k2 = k1 + 1;
    if (cg < NCG) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* vr = s_vec + 4 * r4;
      const float* vcol = s_vec + cg;
      const f4 v4 = *reinterpret_cast<const f4*>(vr + ov);
      const f4 w4 = *reinterpret_cast<const f4*>(vr);
      // PB column slots per batch: their v / w / v' values are fetched from LDS together, so a batch pays one LDS round
      // trip instead of one per slot.  Only the instantiations beyond D = 128 have the registers for that (one workgroup per CU,
      // nobody else to hide the latency); at D <= 128 the kernel is held to 64 VGPRs and four co-resident workgroups do
      // the hiding.
      constexpr int PB = (NT > 4) ? 8 : 1;
#pragma unroll
      for (int i0 = 0; i0 < NC; i0 += PB) {
        // wave-uniform: some column of this batch is still in the trailing matrix -- and (D = 128 layout) some row of this wave:
        // rows <= k have left it, v, w and the next reflector vanish there
        if (cgmax + NCG * (i0 + PB - 1) > k1 && row_last > k) {
          float vcv[PB], wcv[PB], ncv[PB];
#pragma unroll
          for (int u = 0; u < PB; ++u) {
            const int cc = cg + NCG * (i0 + u);
            const int c = ((NCG * NC > DP && cc >= DP) || i0 + u >= NC) ? 0 : cc;  // (slots past the padded size hold zeros)
            const int co = c - cg;  // compile-time unless the slot is past the padded size
            vcv[u] = vcol[ov + co];
            wcv[u] = vcol[co];
            ncv[u] = vcol[on + co];
          }
#pragma unroll
          for (int u = 0; u < PB; ++u) {
            const int i = i0 + u;
            if (i < NC && cgmax + NCG * i > k1) {
              const float vc = vcv[u], wc = wcv[u], nc = ncv[u];
              f4 t = (i < NL) ? s_a[i < NL ? i : 0][tid] : a[i < NC ? i : 0];
              t.x = t.x - vc * w4.x - wc * v4.x;
              t.y = t.y - vc * w4.y - wc * v4.y;
              t.z = t.z - vc * w4.z - wc * v4.z;
              t.w = t.w - vc * w4.w - wc * v4.w;
              acc.x = fmaf(t.x, nc, acc.x);
              acc.y = fmaf(t.y, nc, acc.y);
              acc.z = fmaf(t.z, nc, acc.z);
              acc.w = fmaf(t.w, nc, acc.w);
              if (i < NL) s_a[i < NL ? i : 0][tid] = t;
              else a[i < NC ? i : 0] = t;
            }
          }
        }
      }
      const f4 n4 = *reinterpret_cast<const f4*>(vr + on);
      vav = acc.x * n4.x + acc.y * n4.y + acc.z * n4.z + acc.w * n4.w;  // this thread's share of v'.(A v')
      if (kDppSum) {
        acc.x += dpp_move<0xb1>(acc.x); acc.y += dpp_move<0xb1>(acc.y); acc.z += dpp_move<0xb1>(acc.z); acc.w += dpp_move<0xb1>(acc.w);      // quad_perm:[1,0,3,2]
        acc.x += dpp_move<0x4e>(acc.x); acc.y += dpp_move<0x4e>(acc.y); acc.z += dpp_move<0x4e>(acc.z); acc.w += dpp_move<0x4e>(acc.w);      // quad_perm:[2,3,0,1]
        acc.x += dpp_move<0x124>(acc.x); acc.y += dpp_move<0x124>(acc.y); acc.z += dpp_move<0x124>(acc.z); acc.w += dpp_move<0x124>(acc.w);  // row_ror:4
        acc.x += dpp_move<0x128>(acc.x); acc.y += dpp_move<0x128>(acc.y); acc.z += dpp_move<0x128>(acc.z); acc.w += dpp_move<0x128>(acc.w);  // row_ror:8
        if (cg == 0) *reinterpret_cast<f4*>(&s_part[4 * r4]) = acc;  // the row group's total over all column groups
      } else if (kPairSum) {
        acc.x = sum_halves(acc.x);
        acc.y = sum_halves(acc.y);
        acc.z = sum_halves(acc.z);
        acc.w = sum_halves(acc.w);
        if (lane < 32) *reinterpret_cast<f4*>(&s_part[(cg >> 1) * PS + 4 * r4]) = acc;
      } else {
        *reinterpret_cast<f4*>(&s_part[cg * PS + 4 * r4]) = acc;
      }
      // export column k2 (its owners: column group k2 % NCG, slot k2 / NCG) and, after the last sweep, the corner
      const int i2 = k2 / NCG;
      if (cg == k2 - i2 * NCG) {  // (i2 is wave-uniform: one scalar branch per slot, one store executes)
        f4* dst = reinterpret_cast<f4*>(s_vec + 4 * r4 + (3 + (cur ^ 1)) * DP);
#pragma unroll
        for (int i = 0; i < NC; ++i)
          if (i2 == i) *dst = (i < NL) ? s_a[i < NL ? i : 0][tid] : a[i];
      }
      if (k1 == n - 3) {
        const int ic = (n - 1) / NCG;
        if (cg == (n - 1) - ic * NCG && r4 == ((n - 1) >> 2)) {
#pragma unroll
          for (int i = 0; i < NC; ++i)
            if (ic == i) s_corner = f4_elem((i < NL) ? s_a[i < NL ? i : 0][tid] : a[i], (n - 1) & 3);
        }
      }
    }
    vav = wave_sum(vav);
#if UGLAD_TRIDIAG_LANE == 0
    if (lane == 0) s_dotp[wv] = vav;
#else
    if (lane_now() == 0) s_dotp[wv_u] = vav;
#endif
    __syncthreads();
    TSTAMP_ADD(1);
    const int t = ov;
    ov = on;
    on = t;
    cur ^= 1;
  }
#ifdef UGLAD_STAMPS
  if (tid == 0 && blockIdx.x < 4096) g_twg[blockIdx.x][1] = __builtin_amdgcn_s_memtime();
#endif
}

}  // namespace uglad
