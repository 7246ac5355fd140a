// Householder tridiagonalisation of a SMALL symmetric matrix (D <= 64) by ONE wave, the matrix in its registers, no barrier anywhere.
//
// tridiag.h spreads a matrix over 128 NT threads and pays, per reflector, two workgroup barriers and an LDS gather of partial sums: ~2,500
// cycles per step whatever the size, 25 us for a 25 x 25 matrix (BASELINE config 1), 62 us at D = 64 (config 2).  Here lane c holds COLUMN c
// (D <= 32: lanes c and c + 32 hold its upper and lower 16 rows), so the product A v is a sum over the lane's own registers, the rank-2
// update touches nothing but registers, and the only exchanges are two wave reductions (DPP) and two broadcasts of a vector through LDS
// (the entries of v and A v a lane needs are those of its ROWS: the same for every lane, read as 16-byte pieces), each broadcast in flight
// while a reduction runs.
//
// The registers SLIDE: at step p register r of a lane holds row p + r (+ 16 for the lower half), so the pivot row is always register 0 and
// every register index is a literal -- the update writes row p + r into register r - 1 (no extra instruction), v and w are written to LDS
// shifted by p.  Rows past the matrix are zeros and stay zeros.
//
// Measured on MI355X (profiles/r04_tridiag_wave_ab.txt, rocprofv3): D = 25, one matrix: 25 -> 15.5 us per launch (config 1: 1.11 -> 0.975 ms per
// training pass, forward-only 16.9 k -> 19.9 k unroll-steps/s); M = 4096, D = 32: 64 -> 47 us.  What is left is ISSUE: one wave64 retires a
// vector instruction every four cycles and a step is ~250 of them (two DPP reductions, the square root and two divisions, 16 + 32 FMAs, the
// masks), ~1,600 cycles -- overlapping the two chains of a step (below) changed nothing.  The NT = 2 instantiation (lane = a whole column of 64
// rows, 192 FMAs per step) is SLOWER than the workgroup kernel, 69 vs 62 us per launch at M = 128, D = 64, and is not built or dispatched.
//
// Same outputs and conventions as tridiag_kernel (tri = d | e | tau, row k of R = reflector v_k with v_k[k+1] = 1).
#pragma once
#include "glad_device.h"

namespace uglad {

// Ordering point between two phases of ONE wave on its own LDS data (see UGLAD_WAVE_SYNC): LDS only -- the reflector rows this kernel
// stores to global memory every step must not be waited for.  (The DS unit executes a wave's instructions in issue order, so a pure compiler
// barrier would do on the hardware; measured, it buys nothing -- 14.13 vs 14.27 us per launch at D = 25 -- and the wait for the wave's own
// write, ~60 cycles, is kept.)  The reads of a broadcast vector are issued right behind its write, ahead of the reduction that runs meanwhile.
#ifdef UGLAD_SIMT_EMUL
#define UGLAD_WAVE_SYNC_LDS() simt::wave_sync_point()
#else
#define UGLAD_WAVE_SYNC_LDS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// the value lane (l & 31) / lane (l & 31) + 32 holds, in both halves of the wave (v_permlane32_swap, gfx950)
__device__ __forceinline__ float low_half(float v) {
#ifdef UGLAD_SIMT_EMUL
  return __shfl(v, (int)(threadIdx.x & 31));
#else
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]);
#endif
}
__device__ __forceinline__ float high_half(float v) {
#ifdef UGLAD_SIMT_EMUL
  return __shfl(v, (int)(threadIdx.x & 31) + 32);
#else
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[1]);
#endif
}

template <int NT>
__global__ __launch_bounds__(64) void tridiag_wave_kernel(const float* __restrict__ A0, const float* __restrict__ A1,
                                                          const float* __restrict__ lam_ptr, float* __restrict__ Rbase,
                                                          float* __restrict__ tri_base, int D, int gs, const int* __restrict__ only_flagged) {
  static_assert(NT == 1 || NT == 2, "one wave holds at most 64 columns");
  if (only_flagged && only_flagged[blockIdx.x] == 0) return;
  constexpr int DP = NT * 32, HV = 64 / DP, RPL = DP / HV;  // halves of a column, rows per lane (16 at D <= 32, 64 beyond)
  constexpr int G = 16, NG = RPL / G;                       // rows are processed in groups of 16 registers; whole groups past the matrix are skipped
  __shared__ __attribute__((aligned(16))) float s_v[DP], s_w[DP];
  const int lane = threadIdx.x, c = lane % DP, hf = lane / DP;
  const int n = D;
  const size_t base = (size_t)blockIdx.x * D * D;
  float* R = Rbase + base;
  float* tri = tri_base + (size_t)blockIdx.x * 3 * DP;
  const float inv_lam = lam_ptr ? 1.0f / lam_ptr[blockIdx.x / gs] : 1.0f;

  // ---- load: register r = entry (row hf RPL + r, column c); a wave reads whole rows (the matrix is symmetric)
  float a[RPL];
#pragma unroll
  for (int r = 0; r < RPL; ++r) {
    const int row = hf * RPL + r;
    float v = 0.f;
    if (row < n && c < n) {
      v = A0[base + (size_t)row * D + c];
      if (A1) v = fmaf(inv_lam, v, -A1[base + (size_t)row * D + c]);
    }
    a[r] = v;
  }
  // entries of d | e | tau no step writes
  for (int i = lane; i < DP; i += 64) {
    if (i >= n) tri[i] = 0.f;
    if (i >= n - 1) tri[DP + i] = 0.f;
    if (i >= n - 2) tri[2 * DP + i] = 0.f;
  }
  if (n == 1) {
    if (lane == 0) tri[0] = a[0];
    return;
  }

  for (int p = 0; p + 2 < n; ++p) {
    // ---- row p (= column p) is register 0 of the upper halves, row p + 1 register 1.  The step is two chains that do not wait for each
    // other: the norm of the reflector (reduction, square root, two divisions) and the product of A with the UNSCALED column -- with
    // v = sc x~ + e_{p+1} (x~ = x beyond entry p + 1), A v = sc A x~ + A e_{p+1}, and A e_{p+1} is row p + 1, already in every lane.
    const float x = (HV == 2) ? low_half(a[0]) : a[0];
    const float row1 = (HV == 2) ? low_half(a[1]) : a[1];
    const int c0 = p + 1;
    const float xm = (c > c0 && c < n) ? x : 0.f;
    if (hf == 0) s_v[(c - p) & (DP - 1)] = xm;  // shifted by p: entry i = x~[p + i]; the lanes c < p fill the tail with their zeros
    UGLAD_WAVE_SYNC_LDS();
    // x~ at this lane's ROWS, requested now: the latency runs under the norm's reduction, square root and divisions
    const int rem = n - p;  // rows p .. n-1 are left: registers r < rem of the upper half
    float xr[RPL];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * G < rem) {
#pragma unroll
        for (int q = 0; q < G; q += 4) {
          const f4 t = *reinterpret_cast<const f4*>(&s_v[hf * RPL + g * G + q]);
          xr[g * G + q] = t.x;
          xr[g * G + q + 1] = t.y;
          xr[g * G + q + 2] = t.z;
          xr[g * G + q + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < G; ++q) xr[g * G + q] = 0.f;
      }
    }
    const float dp = bcast_lane(x, p), x0 = bcast_lane(x, c0);
    float sig = wave_sum(hf == 0 ? xm * xm : 0.f);
    float beta = x0, tau = 0.f, sc = 0.f;
    if (sig > 0.f) {  // (as tridiag.h: hardware square root, rcp + Newton divisions; tau and the scaling from the same rounded beta)
      const float nrm2 = fmaf(x0, x0, sig);
      if (nrm2 > 1e-30f) {
        beta = -copysignf(__builtin_amdgcn_sqrtf(nrm2), x0);
        const float dd = x0 - beta;
        sc = div_acc(1.0f, dd);
        tau = -dd * div_acc(1.0f, beta);
      } else {
        beta = -copysignf(sqrtf(nrm2), x0);
        tau = (beta - x0) / beta;
        sc = 1.0f / (x0 - beta);
      }
    }
    const float vc = (c == c0) ? 1.f : xm * sc;  // v[c]: this lane's column
    if (hf == 0 && c < n) R[(size_t)p * D + c] = vc;
    if (lane == 0) {
      tri[p] = dp;
      tri[DP + p] = beta;
      tri[2 * DP + p] = tau;
    }
    // ---- (A x~)[c] = sum over the rows of column c
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * G < rem) {
#pragma unroll
        for (int q = 0; q < G; ++q) acc[q & 3] = fmaf(a[g * G + q], xr[g * G + q], acc[q & 3]);
      }
    }
    float ax = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (HV == 2) ax = sum_halves(ax);
    const bool live = c > p && c < n;
    const float av = live ? fmaf(sc, ax, row1) : 0.f;  // (A v)[c]
    if (hf == 0) s_w[(c - p) & (DP - 1)] = av;         // on its way to the rows while v.(A v) is reduced
    UGLAD_WAVE_SYNC_LDS();
    float pr[RPL];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * G < rem) {
#pragma unroll
        for (int q = 0; q < G; q += 4) {
          const f4 t = *reinterpret_cast<const f4*>(&s_w[hf * RPL + g * G + q]);
          pr[g * G + q] = t.x;
          pr[g * G + q + 1] = t.y;
          pr[g * G + q + 2] = t.z;
          pr[g * G + q + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < G; ++q) pr[g * G + q] = 0.f;
      }
    }
    const float vAv = wave_sum(hf == 0 ? vc * av : 0.f);
    const float kk = 0.5f * tau * tau * vAv;
    // w = tau A v - kk v; entry (j, c) loses v[j] w[c] + w[j] v[c] = x~[j] (sc al) + (A v)[j] (tau v[c]) + [j = p + 1] al, al = w[c] - kk v[c]
    const float wc = live ? fmaf(tau, av, -kk * vc) : 0.f;
    const float al = fmaf(-kk, vc, wc), als = sc * al, be = tau * vc;
    // ---- A <- A - v w^T - w v^T, row p + r into register r - 1
    float carry = 0.f;  // the lower half's first row moves into the upper half's last register
    if (HV == 2) {
      const float u0 = a[0] - xr[0] * als - pr[0] * be;
      const float up = high_half(u0);
      carry = (hf == 0) ? up : 0.f;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * G < rem) {
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int r = g * G + q;
          if (r > 0) {
            float t = a[r] - xr[r] * als - pr[r] * be;
            if (r == 1) t -= (hf == 0) ? al : 0.f;  // (row p + 1, where v is 1)
            a[r - 1] = t;
          }
        }
        if (g == NG - 1) a[RPL - 1] = carry;
        else if ((g + 1) * G >= rem) a[(g + 1) * G - 1] = 0.f;  // (the next group is past the matrix: a zero row moves in)
      }
    }
    UGLAD_WAVE_SYNC_LDS();  // every lane has read both vectors: the next step may overwrite them
  }
  // ---- the trailing 2 x 2 block: rows n-2 and n-1 are registers 0 and 1
  {
    const float x = (HV == 2) ? low_half(a[0]) : a[0];
    const float y = (HV == 2) ? low_half(a[1]) : a[1];
    const float d2 = bcast_lane(x, n - 2), e2 = bcast_lane(x, n - 1), d1 = bcast_lane(y, n - 1);
    if (lane == 0) {
      tri[n - 2] = d2;
      tri[DP + n - 2] = e2;
      tri[n - 1] = d1;
    }
  }
}

}  // namespace uglad
