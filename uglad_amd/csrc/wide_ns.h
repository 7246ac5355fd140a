// The matrix-iteration path: the cell for matrices beyond the eigensolver's size (D > 256; uglad_set_matrix_iteration(1) forces it for
// any D -- tests, A/B).  The reference has no size limit (glad.py:103-151): it computes the square root of b^T b + 4/lam I with ten
// Newton-Schulz steps of dense products and differentiates it with ten steps of an iterative Lyapunov solver (torch_sqrtm.py:13-46).
// Up to D = 256 this library replaces both by an eigen-decomposition (the same function per eigenvalue); a
// one-workgroup tridiagonalisation does not scale further, whereas dense products are what the chip is built for.  So here the cell IS
// the reference's iteration, product by product, with the elementwise steps fused into the epilogues:
//
//   forward   b = S/lam - Z                                                  ns_b_kernel
//             A = b^T b + 4/lam I, ||A||_F^2 per tile                        product<TN, kNsAffine>
//             Y0 = A / ||A||_F ; T0 = (3 I - Y0) / 2 ; Z1 = T0               ns_norm_kernel, ns_start_kernel  (Z0 = I: two products saved)
//             Y1 = Y0 T0                                                     product<NN, kNsAffine>
//             t = 1..9:  T = (3 I - Z Y) / 2 ; Y <- Y T ; Z <- T Z           three products (no Z after the last)
//             the last Y T carries theta_half = (sqrt(||A||_F) Y - b) / 2, rhoNN + threshold, ||Z - theta_half||^2      kNsTheta
//   backward  phase A (wide_bwd.h): G_half, the direct part of dL/dZ, the 28 rhoNN gradients
//             A0 = sqrtm / ||sqrtm||_F ; Q0 = (G_half / 2) / ||sqrtm||_F      ns_frob_kernel, ns_bwd_start_kernel
//             ten times: P = 3 I - A A ; R = A^T Q - Q A ; Q <- (Q P - A^T R) / 2 ; A <- A P / 2          six products
//             dL/db = b (Q + Q^T) / 2 - G_half / 2 ; G_out -= dL/db ; dL/dlam                           ns_symm_kernel, kNsGout
//
// 28 products forward, 60 backward per step, every one a launch of (D / 64)^2 workgroups per matrix.
//
// ARITHMETIC: fp64 (v_mfma_f64_16x16x4_f64), fp32 at the boundary.  The iteration amplifies rounding errors with D: the reference's own
// fp32 gradients sit 1e-3 (D = 320) to 2.4e-2 (D = 512) from the fp64 value of the same function (tests/golden/grad_noise_floor.json), and
// a second fp32 evaluation with another summation order (the k-ordered fma chain of v_mfma_f32_32x32x2_f32 instead of MKL's blocking)
// lands just as far away on its own side -- measured: up to 4.9e-3 from the reference at D = 320, profiles/r03_ns_noise_probe_fp32.txt.
// In fp64 the kernels return that function's value itself, so their distance from the reference is the reference's noise and nothing
// more -- the property the spectral path has by construction.  The square root travels from the forward to the backward pass in fp32, in
// the slot that holds the eigenvectors on the spectral path (U), as the reference's backward starts from its fp32 square root.
//
// Theta_0 and the loss's logdet / inverse for D > 256: blocked L D L^T (chol.h) on slabs laid out for 1024, one workgroup per matrix
// on three slabs of the workspace, then two Newton steps on tile products (fp32).  D carries the signs, so torch.logdet's rules (finite
// for an even number of negative eigenvalues, NaN for an odd one) hold as on the spectral path; no pivoting.
//
// Workspace per matrix: the header of kWsPerMatrix floats the other paths use (partial sums, scalars) and a region of eight D x D fp64
// slabs plus one fp32 slab (G_half).
#pragma once
#include "chol.h"
#include "wide_bwd.h"

namespace uglad {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int kNsSlabs = 8;      // fp64 slabs per matrix
// The products are size-generic; what bounds D is the layout of the L D L^T factorisation behind Theta_0 and the loss: three slabs of F x (F + 1)
// floats per matrix, F = ns_fact_dim(D) -- 1024 up to D = 1024 (rounds 3's layout and timings: a row stride of 2049 floats for every D made the
// factorisation of a 512 x 512 matrix 8 x slower, 17 ms, rows 64 cache lines apart), 2048 beyond (round 4: D up to 2048).
constexpr int kNsMaxD = 2048;
constexpr int kNsFactSmall = 1024;
__host__ __device__ constexpr int ns_fact_dim(int D) { return D <= kNsFactSmall ? kNsFactSmall : kNsMaxD; }
// offsets inside a matrix's header (floats), behind the backward's partial sums (wide_partial_floats): per-tile fp32 sums (the norm of
// the forward cell, the loss's trace), then -- 8-byte aligned -- fp64: per-tile sums, four scalars, nt row-block maxima.  "Per tile" is
// sized for the 32 x 32 tiling ((2 nt)^2 entries); the products run on 64 x 64 or 32 x 32 output tiles (ns_gemm64_kernel).
__host__ __device__ constexpr int ns_tiles_max(int D) { return 4 * wide_tiles(D) * wide_tiles(D); }
__host__ __device__ constexpr int ns_off_tiles(int D) { return wide_partial_floats(D); }
__host__ __device__ constexpr int ns_off_dbl(int D) { return (ns_off_tiles(D) + ns_tiles_max(D) + 1) & ~1; }
__host__ __device__ constexpr int ns_dscal(int D) { return ns_tiles_max(D); }  // (in doubles from ns_off_dbl; the tile sums sit at 0)
__host__ __device__ constexpr int ns_dcond(int D) { return ns_dscal(D) + 4; }
enum { kNsNormA = 0, kNsSqrtNormA = 1 };
__device__ __forceinline__ double* ns_dbl(float* hdr, size_t hdr_stride, int m, int D) {
  return reinterpret_cast<double*>(hdr + (size_t)m * hdr_stride + ns_off_dbl(D));
}
__device__ __forceinline__ const double* ns_dbl(const float* hdr, size_t hdr_stride, int m, int D) {
  return reinterpret_cast<const double*>(hdr + (size_t)m * hdr_stride + ns_off_dbl(D));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------- tile product, fp64
// C(i, j) = sum_k A(i, k) B(k, j) on the T x T tile (blockIdx.y, blockIdx.x) of matrix blockIdx.z: four waves, each a 32 x 32 block
// as 2 x 2 accumulators of v_mfma_f64_16x16x4_f64 (lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; result register r of lane l
// is C[(l >> 4) + 4 r][l & 15]).  A(i, k) = ta ? Ag[k][i] : Ag[i][k], B(k, j) = Bg[k][j]; every matrix D x D, row stride D,
// `stride` doubles from one matrix of the batch to the next.  Operands are staged through LDS in k chunks ([k][x]), the next chunk
// prefetched into registers.
enum { kNsAffine = 0, kNsTheta = 1, kNsGout = 2 };
struct NsEpi {
  // kNsAffine: C = alpha acc + beta C + gamma delta_ij (NsProd; beta == 0: C is not read; gamma_div: gamma / gamma_div[m / gs]); with
  // hdr (single product): the sum of C^2 over the tile -> the header's fp64 tile sums
  const float* gamma_div;
  float* hdr;
  size_t hdr_stride;
  int gs;
  // kNsTheta (upper tiles, mirrored): theta_half = (sqrt(||A||_F) acc - b) / 2 -> fp32, then rhoNN + soft threshold exactly as the
  // spectral path's epilogue; the square root itself -> sqrt_out; per-tile ||Z - theta_half||^2 -> the header's fp32 tile sums
  // kNsGout: acc = b (Q + Q^T): bbar = acc / 2 - G_half / 2; G_out -= bbar; per-tile dL/dlam -> the header's fp64 tile sums
  const double* b;       // slab, same stride as the operands
  const float* S;
  const float* Zin;
  const float* params;
  const float* lam;
  float* Zout;           // kNsTheta: Z_out; kNsGout: G_out (holds the direct part of dL/dZ_in since phase A)
  float* half_out;       // may be null (inference)
  float* sqrt_out;       // may be null
  const float* Gh;       // kNsGout: G_half, row stride D, gh_stride floats per matrix
  size_t gh_stride;
};

// T = 64: every wave a 32 x 32 quarter of the tile, k chunks of 32.  T = 32 (few matrices: four times the workgroups, so that one
// 512 x 512 product covers the chip): every wave the WHOLE 32 x 32 tile over a quarter of each k chunk of 64, the four partial results
// added through LDS in a fixed order at the end; wave w then owns the 16 x 16 quadrant (w >> 1, w & 1) in the epilogue.
template <int T>
struct NsTile {
  static constexpr int kK = (T == 64) ? 32 : 64;   // k chunk (T = 32: 128 measured the same 19.4 us per 512^3 product as 64)
  static constexpr int kRuns = T * kK / (kWThreads * 4);  // runs of four source elements per thread, chunk and operand
  static constexpr int kLd = (T == 64) ? 80 : 48;  // LDS row stride in doubles: = 16 mod 32, the four k rows of a read start 32 banks apart
  static constexpr int kQ = (T == 64) ? 2 : 1;     // 16 x 16 blocks per wave and dimension in the epilogue
};

// Up to three independent products per launch (blockIdx.z = matrix * n + product): while one product's tiles do not fill the chip, the
// steps of the iteration that do not depend on each other share a launch -- {Y T, T Z} forward; {A A, A^T Q}, {Q A, Q P, A P} backward.
struct NsProd {
  const double* A;
  const double* B;
  double* C;
  double alpha, beta, gamma;  // kNsAffine (per product)
  int ta;                     // A is read transposed
};
struct NsBatch {
  NsProd p[3];
  int n;
};

// PRE (round 4; T = 32, D <= 4 k chunks = 256 -- config 5's one 256 x 256 matrix per GPU): ALL k chunks of both operands are requested before the first
// one is used (128 registers of operands in flight per thread) instead of one chunk ahead: the loop then never waits for memory -- per chunk it had
// taken 4.2 k cycles against 1 k of MFMA issue (profiles/r03_ns_tile_probe.txt); an instantiation of its own so that larger D keeps its occupancy.
template <int EPI, int T, bool PRE = false>
__global__ __launch_bounds__(kWThreads) void ns_gemm64_kernel(NsBatch batch, size_t stride, int D, NsEpi ep) {
  constexpr int kK = NsTile<T>::kK, kLd = NsTile<T>::kLd, kQ = NsTile<T>::kQ, kRuns = NsTile<T>::kRuns;
  __shared__ __attribute__((aligned(16))) double s_stage[2 * kK * kLd];  // 40 KB (T = 64) / 48 KB (T = 32): three workgroups per CU
  double* const sA = s_stage;
  double* const sB = s_stage + kK * kLd;
  double* const s_red = s_stage;  // T = 32, after the last chunk: [wave][a][c][r][lane], 4096 doubles
  static_assert(T == 64 || 2 * kK * kLd >= 4 * 16 * 64, "the reduction buffer reuses the staging area");
  __shared__ double s4d[4];
  if (EPI == kNsTheta && blockIdx.y > blockIdx.x) return;  // (symmetric: upper tiles only; uniform per workgroup)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m = blockIdx.z / batch.n, I = blockIdx.y, J = blockIdx.x;
  const NsProd pr = batch.p[blockIdx.z - m * batch.n];
  const double* __restrict__ A = pr.A + (size_t)m * stride;
  const double* __restrict__ B = pr.B + (size_t)m * stride;
  const bool TA = pr.ta != 0;
  constexpr bool TB = false;  // (no step of the iteration reads its second operand transposed)
  const int i0 = I * T, j0 = J * T;

  // staging: a chunk is T (x) x kK (k) doubles per operand: kRuns runs of four consecutive source elements per thread (run p)
  //   source contiguous in k (A not transposed / B transposed):  T = 64: x = tid / 8 + 32 p, k = 4 (tid % 8)          T = 32: x = tid / 8, k = 4 (tid % 8) + 32 p
  //   source contiguous in x (A transposed / B not transposed):  T = 64: k = tid / 16 + 16 p, x = 4 (tid % 16)       T = 32: k = tid / 8 + 32 p, x = 4 (tid % 8)
  double pa[4 * kRuns], pb[4 * kRuns];
  auto coords = [&](bool contig_k, int pp, int& x, int& k) {  // first element of run pp (x or k runs over four consecutive values)
    if (T == 64) {
      x = contig_k ? (tid >> 3) + 32 * pp : 4 * (tid & 15);
      k = contig_k ? 4 * (tid & 7) : (tid >> 4) + 16 * pp;
    } else {
      x = contig_k ? (tid >> 3) : 4 * (tid & 7);
      k = contig_k ? 4 * (tid & 7) + 32 * pp : (tid >> 3) + 32 * pp;
    }
  };
  auto fetch = [&](const double* __restrict__ src, bool contig_k, int x0, int k0, double (&p)[4 * kRuns]) {
    const bool vec = ((D & 1) == 0) && ((reinterpret_cast<size_t>(src) & 15) == 0);
#pragma unroll
    for (int pp = 0; pp < kRuns; ++pp) {
      int x, k;
      coords(contig_k, pp, x, k);
      const int row = contig_k ? x0 + x : k0 + k;
      const int col = contig_k ? k0 + k : x0 + x;
      if (vec) {  // (D even, col even: a pair is inside the matrix or outside as a whole)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f64x2 v = {0.0, 0.0};
          if (row < D && col + 2 * h < D) v = *reinterpret_cast<const f64x2*>(src + (size_t)row * D + col + 2 * h);
          p[4 * pp + 2 * h] = v.x;
          p[4 * pp + 2 * h + 1] = v.y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) p[4 * pp + c] = (row < D && col + c < D) ? src[(size_t)row * D + col + c] : 0.0;
      }
    }
  };
  // LDS layout of a staged chunk: [k][x] (row stride kLd) for an operand whose source runs along x -- the run of a thread lands in one row,
  // and the MFMA operand read (16 consecutive x, four k) is conflict-free.  An operand whose source runs along k (A, not transposed) is
  // kept [x][k] instead (row stride kLdk = kK + 2 doubles: rows 4 banks apart, the two k of a half wave 2 banks apart -- conflict-free
  // reads again) so that a thread's run of four k is 32 contiguous bytes: staged [k][x] those four stores went to four rows, eight lanes to
  // each bank (measured: as long as the chunk's MFMAs).
  constexpr int kLdk = kK + 2;
  static_assert(T * kLdk <= kK * kLd, "the [x][k] layout fits the operand's staging area");
  auto stash = [&](double* dst, bool contig_k, const double (&p)[4 * kRuns]) {
#pragma unroll
    for (int pp = 0; pp < kRuns; ++pp) {
      int x, k;
      coords(contig_k, pp, x, k);
      if (contig_k) {
        f64x2* q = reinterpret_cast<f64x2*>(dst + x * kLdk + k);  // (k a multiple of 4, kLdk even: 16-byte aligned)
        q[0] = f64x2{p[4 * pp], p[4 * pp + 1]};
        q[1] = f64x2{p[4 * pp + 2], p[4 * pp + 3]};
      } else {
        f64x2* q = reinterpret_cast<f64x2*>(dst + k * kLd + x);  // (x a multiple of 4, kLd a multiple of 16: 16-byte aligned)
        q[0] = f64x2{p[4 * pp], p[4 * pp + 1]};
        q[1] = f64x2{p[4 * pp + 2], p[4 * pp + 3]};
      }
    }
  };
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = (f64x4){0.0, 0.0, 0.0, 0.0};
  const int l16 = lane & 15, kq = lane >> 4;
  const int ri = (T == 64) ? (w >> 1) * 32 : 0, rj = (T == 64) ? (w & 1) * 32 : 0;  // the 32 x 32 block this wave multiplies
  KSTAMP(10);
  constexpr int kSteps = (T == 64) ? kK / 4 : kK / 16;  // k steps of four per wave and chunk
  auto multiply_chunk = [&]() {
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks) {
      const int k = 4 * ((T == 64) ? ks : kSteps * w + ks) + kq;
      const double a0 = TA ? sA[k * kLd + ri + l16] : sA[(ri + l16) * kLdk + k];
      const double a1 = TA ? sA[k * kLd + ri + 16 + l16] : sA[(ri + 16 + l16) * kLdk + k];
      const double b0 = sB[k * kLd + rj + l16], b1 = sB[k * kLd + rj + 16 + l16];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  if constexpr (PRE) {
    static_assert(T == 32, "all chunks in flight: the 32 x 32 tiling of few matrices only");
    constexpr int kChunks = 4;  // D <= kChunks * kK (the launcher's condition)
    double qa[kChunks][4 * kRuns], qb[kChunks][4 * kRuns];
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (c * kK < D) {  // (uniform)
        fetch(A, !TA, i0, c * kK, qa[c]);
        fetch(B, TB, j0, c * kK, qb[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < kChunks; ++c) {
      if (c * kK < D) {
        __syncthreads();  // (the previous chunk has been consumed)
        if (c == 0) KSTAMP(11);
        stash(sA, !TA, qa[c]);
        stash(sB, TB, qb[c]);
        __syncthreads();
        if (c == 0) KSTAMP(12);
        multiply_chunk();
      }
    }
  } else {
    fetch(A, !TA, i0, 0, pa);
    fetch(B, TB, j0, 0, pb);
    for (int k0 = 0; k0 < D; k0 += kK) {
      __syncthreads();  // (the previous chunk has been consumed)
      if (k0 == 0) KSTAMP(11);
      stash(sA, !TA, pa);
      stash(sB, TB, pb);
      __syncthreads();
      if (k0 == 0) KSTAMP(12);
      if (k0 + kK < D) {
        fetch(A, !TA, i0, k0 + kK, pa);
        fetch(B, TB, j0, k0 + kK, pb);
      }
      multiply_chunk();
    }
  }
  KSTAMP(13);
  int wi = ri, wj = rj;
  if (T == 32) {  // add the four waves' partial results (wave order: deterministic); wave w keeps quadrant (w >> 1, w & 1) in acc[0][0]
    __syncthreads();  // (every wave is done reading the last chunk: the staging area becomes the reduction buffer)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[((w * 4 + a * 2 + c) * 4 + r) * 64 + lane] = acc[a][c][r];
    __syncthreads();
    const int qa = w >> 1, qc = w & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) v += s_red[((ww * 4 + qa * 2 + qc) * 4 + r) * 64 + lane];
      acc[0][0][r] = v;
    }
    wi = 16 * qa;
    wj = 16 * qc;
  }

  // ---- epilogue: acc[a][c][r] of lane l = C[i0 + wi + 16 a + (l >> 4) + 4 r][j0 + wj + 16 c + (l & 15)]
  KSTAMP(14);
  double part = 0.0;
  if (EPI == kNsAffine) {
    double* C = pr.C + (size_t)m * stride;
    const double gam = ep.gamma_div ? pr.gamma / (double)ep.gamma_div[m / ep.gs] : pr.gamma;
#pragma unroll
    for (int a = 0; a < kQ; ++a)
#pragma unroll
      for (int c = 0; c < kQ; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wi + 16 * a + kq + 4 * r, j = j0 + wj + 16 * c + l16;
          if (i < D && j < D) {
            double v = pr.alpha * acc[a][c][r];
            if (pr.beta != 0.0) v += pr.beta * C[(size_t)i * D + j];
            if (i == j) v += gam;
            C[(size_t)i * D + j] = v;
            part += v * v;
          }
        }
  } else if (EPI == kNsTheta) {
    const size_t base = (size_t)m * D * D;
    const float* Sm = ep.S + base;
    const float* Zm = ep.Zin + base;
    float* Zo = ep.Zout + base;
    float* Hm = ep.half_out ? ep.half_out + base : nullptr;
    float* Qm = ep.sqrt_out ? ep.sqrt_out + base : nullptr;
    const double* bm = ep.b + (size_t)m * stride;
    const float* prm = ep.params + (size_t)(m / ep.gs) * kNParam;
    const double hs = ns_dbl(ep.hdr, ep.hdr_stride, m, D)[ns_dscal(D) + kNsSqrtNormA];
    float nrm = 0.f;
#pragma unroll
    for (int a = 0; a < kQ; ++a)
#pragma unroll
      for (int c = 0; c < kQ; ++c)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {  // two entries per pass on the packed pipe
          const int j = j0 + wj + 16 * c + l16;
          int iv[2];
          float xv[2], sv[2], zv[2];
          bool in[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            iv[u] = i0 + wi + 16 * a + kq + 4 * (r + u);
            in[u] = iv[u] < D && j < D && iv[u] <= j;
            sv[u] = in[u] ? Sm[(size_t)iv[u] * D + j] : 0.f;
            zv[u] = in[u] ? Zm[(size_t)iv[u] * D + j] : 0.f;
            xv[u] = 0.f;
            if (in[u]) {
              const double sq = hs * acc[a][c][r + u];
              xv[u] = (float)(0.5 * (sq - bm[(size_t)iv[u] * D + j]));  // glad.py:142
              if (Qm) {
                Qm[(size_t)iv[u] * D + j] = (float)sq;
                Qm[(size_t)j * D + iv[u]] = (float)sq;
              }
            }
          }
          RhoAct2 act;
          rho_forward2(prm, (v2f){xv[0], xv[1]}, (v2f){sv[0], sv[1]}, (v2f){zv[0], zv[1]}, act);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (in[u]) {
              const int i = iv[u];
              const float zn = soft_threshold(xv[u], u ? act.rho.y : act.rho.x);
              const float d = zn - xv[u];
              nrm = fmaf((i == j) ? 1.f : 2.f, d * d, nrm);
              Zo[(size_t)i * D + j] = zn;
              Zo[(size_t)j * D + i] = zn;
              if (Hm) {
                Hm[(size_t)i * D + j] = xv[u];
                Hm[(size_t)j * D + i] = xv[u];
              }
            }
          }
        }
    part = (double)nrm;
  } else {  // kNsGout
    const size_t base = (size_t)m * D * D;
    const float* Sm = ep.S + base;
    float* Go = ep.Zout + base;
    const float* Gh = ep.Gh + (size_t)m * ep.gh_stride;
    const double lam = (double)ep.lam[m / ep.gs], inv_lam2 = 1.0 / (lam * lam);
#pragma unroll
    for (int a = 0; a < kQ; ++a)
#pragma unroll
      for (int c = 0; c < kQ; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + wi + 16 * a + kq + 4 * r, j = j0 + wj + 16 * c + l16;
          if (i < D && j < D) {
            const double bbar = 0.5 * acc[a][c][r] - 0.5 * (double)Gh[(size_t)i * D + j];
            Go[(size_t)i * D + j] = (float)((double)Go[(size_t)i * D + j] - bbar);
            part -= (double)Sm[(size_t)i * D + j] * inv_lam2 * bbar;
            if (i == j) part -= inv_lam2 * B[(size_t)i * D + i];  // c = 4/lam: -4/lam^2 tr(Abar), tr(Abar) = tr(Q + Q^T) / 4
          }
        }
  }
  KSTAMP(15);
  if (EPI != kNsAffine || ep.hdr != nullptr) {  // (uniform per launch)
    part = wave_sum_f64(part);
    __syncthreads();
    if (lane == 0) s4d[w] = part;
    __syncthreads();
    if (tid == 0) {
      const double v = (s4d[0] + s4d[1]) + (s4d[2] + s4d[3]);
      const int t = I * gridDim.x + J;
      if (EPI == kNsAffine) {
        ns_dbl(ep.hdr, ep.hdr_stride, m, D)[t] = v;
      } else if (EPI == kNsTheta) {
        ep.hdr[(size_t)m * ep.hdr_stride + ns_off_tiles(D) + t] = (float)v;
      } else {
        ns_dbl(ep.hdr, ep.hdr_stride, m, D)[t] = v;  // (ns_glam_kernel hands the sum to wide_reduce_kernel)
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- elementwise steps
// b = S/lam - Z in fp64 from the fp32 inputs, all entries
__global__ void ns_b_kernel(const float* __restrict__ S, const float* __restrict__ Zin, const float* __restrict__ lam_ptr, double* __restrict__ Bout,
                            size_t stride, int D, int gs) {
  const int m = blockIdx.y;
  const double lam = (double)lam_ptr[m / gs];
  const size_t base = (size_t)m * D * D, dd = (size_t)D * D;
  double* Bm = Bout + (size_t)m * stride;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dd; idx += (size_t)gridDim.x * blockDim.x)
    Bm[idx] = (double)S[base + idx] / lam - (double)Zin[base + idx];
}

// Gershgorin: cond(A) <= max_i sum_j |A_ij| / (4/lam) for A = b^T b + 4/lam I (its smallest eigenvalue is at least 4/lam) -- the
// regime diagnostic of this path: an upper bound, where the spectral path reports the condition number itself.  One workgroup per 64
// rows, one wave per row; the maximum over the block's rows -> header (ns_norm_kernel finishes).
__global__ __launch_bounds__(256) void ns_cond_kernel(const double* __restrict__ A, size_t stride, float* __restrict__ hdr, size_t hdr_stride, int D) {
  __shared__ double s4[4];
  const int m = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* Am = A + (size_t)m * stride;
  double best = 0.0;
  const int r1 = (blockIdx.x * kWT + kWT < D) ? blockIdx.x * kWT + kWT : D;
  for (int i = blockIdx.x * kWT + w; i < r1; i += 4) {
    double v = 0.0;
    for (int j = lane; j < D; j += 64) v += fabs(Am[(size_t)i * D + j]);
    best = fmax(best, wave_sum_f64(v));
  }
  if (lane == 0) s4[w] = best;
  __syncthreads();
  if (threadIdx.x == 0) ns_dbl(hdr, hdr_stride, m, D)[ns_dcond(D) + blockIdx.x] = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
}

// per-tile sums -> ||A||_F and its square root (one wave per matrix: lane-strided partial sums, then the shuffle tree -- a fixed order);
// with cond_max: the row-sum maxima of ns_cond_kernel -> running maximum of the Gershgorin bound of cond(A)
__global__ __launch_bounds__(64) void ns_norm_kernel(float* __restrict__ hdr, size_t hdr_stride, const float* __restrict__ lam_ptr,
                                                     float* __restrict__ cond_max, int D, int gs, int ntiles) {
  const int m = blockIdx.x, lane = threadIdx.x, nt = wide_tiles(D);
  double* h = ns_dbl(hdr, hdr_stride, m, D);
  double v = 0.0;
  for (int t = lane; t < ntiles; t += 64) v += h[t];
  v = wave_sum_f64(v);
  if (lane != 0) return;
  const double n = sqrt(v);
  h[ns_dscal(D) + kNsNormA] = n;
  h[ns_dscal(D) + kNsSqrtNormA] = sqrt(n);
  if (cond_max) {
    double r = 0.0;
    for (int t = 0; t < nt; ++t) r = fmax(r, h[ns_dcond(D) + t]);
    cond_max[m] = fmaxf(cond_max[m], (float)(r * (double)lam_ptr[m / gs] * 0.25));
  }
}

// Y0 = A / ||A||_F (in place); T0 = (3 I - Y0) / 2 -> T and Z (torch_sqrtm.py:17-25 with Z0 = I)
__global__ void ns_start_kernel(double* __restrict__ Y, double* __restrict__ T, double* __restrict__ Z, size_t stride, const float* __restrict__ hdr,
                                size_t hdr_stride, int D) {
  const int m = blockIdx.y;
  const double n = ns_dbl(hdr, hdr_stride, m, D)[ns_dscal(D) + kNsNormA];
  const size_t dd = (size_t)D * D, off = (size_t)m * stride;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dd; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / D), j = (int)(idx - (size_t)i * D);
    const double y = Y[off + idx] / n;
    const double t = 0.5 * (((i == j) ? 3.0 : 0.0) - y);
    Y[off + idx] = y;
    T[off + idx] = t;
    Z[off + idx] = t;
  }
}

// ||X||_F^2 of an fp32 matrix in kNsFrobBlocks slices, one workgroup each -> the header's fp64 tile sums (ns_bwd_start_kernel adds them)
constexpr int kNsFrobBlocks = 32;
__global__ __launch_bounds__(256) void ns_frob_kernel(const float* __restrict__ X, size_t x_stride, float* __restrict__ hdr, size_t hdr_stride, int D) {
  __shared__ double s4[4];
  const int m = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* Xm = X + (size_t)m * x_stride;
  const size_t dd = (size_t)D * D, per = (dd + kNsFrobBlocks - 1) / kNsFrobBlocks;
  const size_t lo = blockIdx.x * per, hi = (lo + per < dd) ? lo + per : dd;
  double v = 0.0;
  for (size_t idx = lo + threadIdx.x; idx < hi; idx += 256) v += (double)Xm[idx] * (double)Xm[idx];
  v = wave_sum_f64(v);
  if (lane == 0) s4[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) ns_dbl(hdr, hdr_stride, m, D)[blockIdx.x] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// A0 = sqrtm / ||sqrtm||_F ; Q0 = (G_half / 2) / ||sqrtm||_F   (torch_sqrtm.py:37-41; grad_output = G_half / 2 by glad.py:142)
__global__ void ns_bwd_start_kernel(const float* __restrict__ sqrtm, const float* __restrict__ Gh, size_t gh_stride, double* __restrict__ A0,
                                    double* __restrict__ Q0, size_t stride, const float* __restrict__ hdr, size_t hdr_stride, int D) {
  const int m = blockIdx.y;
  const double* h = ns_dbl(hdr, hdr_stride, m, D);
  double n2 = 0.0;
  for (int t = 0; t < kNsFrobBlocks; ++t) n2 += h[t];  // (ns_frob_kernel's slices, fixed order)
  const double n = sqrt(n2);
  const size_t dd = (size_t)D * D, off = (size_t)m * stride, base = (size_t)m * dd, goff = (size_t)m * gh_stride;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dd; idx += (size_t)gridDim.x * blockDim.x) {
    A0[off + idx] = (double)sqrtm[base + idx] / n;
    Q0[off + idx] = (0.5 * (double)Gh[goff + idx]) / n;
  }
}

// out = Q + Q^T
__global__ void ns_symm_kernel(const double* __restrict__ Q, double* __restrict__ out, size_t stride, int D) {
  const int m = blockIdx.y;
  const size_t dd = (size_t)D * D, off = (size_t)m * stride;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < dd; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / D), j = (int)(idx - (size_t)i * D);
    out[off + idx] = Q[off + idx] + Q[off + (size_t)j * D + i];
  }
}

// forward: per-tile ||Z - theta_half||^2 of the upper tiles (ntd per dimension) -> normF_partial[m]; one wave per matrix, fixed order
__global__ __launch_bounds__(64) void ns_norm_reduce_kernel(const float* __restrict__ hdr, size_t hdr_stride, int ntd, float* __restrict__ normF_partial,
                                                            int D) {
  const int m = blockIdx.x, lane = threadIdx.x;
  float v = 0.f;
  for (int t = lane; t < ntd * ntd; t += 64) {
    const int I = t / ntd, J = t - I * ntd;
    if (I <= J) v += hdr[(size_t)m * hdr_stride + ns_off_tiles(D) + t];
  }
  v = wave_sum(v);
  if (lane == 0) normF_partial[m] = v;
}

// backward: the last product's per-tile dL/dlam sums -> the slot wide_reduce_kernel adds up (its 2 nt^2 per-tile entries: the sum, then
// zeros); one wave per matrix
__global__ __launch_bounds__(64) void ns_glam_kernel(float* __restrict__ hdr, size_t hdr_stride, int ntiles, int partial_off, int D) {
  const int m = blockIdx.x, lane = threadIdx.x, nt = wide_tiles(D);
  const double* h = ns_dbl(hdr, hdr_stride, m, D);
  double v = 0.0;
  for (int t = lane; t < ntiles; t += 64) v += h[t];
  v = wave_sum_f64(v);
  float* p = hdr + (size_t)m * hdr_stride + partial_off;
  for (int t = lane; t < 2 * nt * nt; t += 64) p[t] = (t == 0) ? (float)v : 0.f;
}

// sum of nt^2 per-tile fp32 partials -> out[m] * scale (gt_partial = -<G0^T, Theta0^2>)
__global__ void ns_tile_sum_kernel(const float* __restrict__ hdr, size_t hdr_stride, float* __restrict__ out, float scale, int M, int D) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x, nt = wide_tiles(D);
  if (m >= M) return;
  float v = 0.f;
  for (int t = 0; t < nt * nt; ++t) v += hdr[(size_t)m * hdr_stride + ns_off_tiles(D) + t];
  out[m] = scale * v;
}

// ---- inverse and log-determinant beyond the eigensolver's size: L D L^T (chol.h) of the matrix padded to the next multiple of 32
// (identity on the padding) on three workspace slabs of row stride FD + 1 (FD = ns_fact_dim(D)), one workgroup per matrix.  X0 = (src + shift I)^-1 is left in the third slab,
// the log-determinant in logdet_out[m] with torch.logdet's rules (NaN for a negative determinant); a zero / NaN pivot leaves NaN in both.
// The caller polishes X0 with Newton steps on tile products.
constexpr int kNsLdlWaves = 16;  // operands live in L2 here, not LDS: a tile product is a round trip of latency, so 16 waves share the tiles
template <int FD>
__global__ __launch_bounds__(64 * kNsLdlWaves) void ns_ldl_kernel(const float* __restrict__ src, const float* __restrict__ shift, int shift_stride,
                                                          float* __restrict__ slabs, size_t slab_stride, float* __restrict__ logdet_out, int D,
                                                          int gs) {
  constexpr int DP = FD, LD = DP + 1;
  __shared__ int s_flag;
  __shared__ float s_acc[2];
  __shared__ float s_dv[DP];
  const int m = blockIdx.x, tid = threadIdx.x;
  float* sL = slabs + (size_t)m * slab_stride;
  float* sW = sL + (size_t)DP * LD;
  float* sX = sW + (size_t)DP * LD;
  const float* Am = src + (size_t)m * D * D;
  const float sh = shift ? shift[(size_t)(m / gs) * shift_stride] : 0.f;
  const int nt = (D + 31) / 32, dp = nt * 32;  // only the leading nt x nt tiles are factorised (identity on their padding)
  for (int idx = tid; idx < dp * dp; idx += 64 * kNsLdlWaves) {
    const int i = idx / dp, k = idx - i * dp;
    sL[i * LD + k] = (i < D && k < D) ? Am[(size_t)i * D + k] + ((i == k) ? sh : 0.f) : ((i == k) ? 1.f : 0.f);
  }
  __syncthreads();
  float logdet;
  int neg;
  const bool ok = ldl_inverse<FD / 32, kNsLdlWaves>(sL, sW, sX, s_dv, logdet, neg, &s_flag, s_acc, nt);
  const float nan = __builtin_nanf("");
  if (!ok) {
    for (int idx = tid; idx < dp * dp; idx += 64 * kNsLdlWaves) sX[(idx / dp) * LD + idx % dp] = nan;
    logdet = nan;
  } else if (neg & 1) {
    logdet = nan;  // torch.logdet of a matrix with negative determinant
  }
  if (tid == 0 && logdet_out) logdet_out[m] = logdet;
}

// ---- the same factorisation as a SEQUENCE OF LAUNCHES (round 4): ldl_inverse's phases, one launch each, the tiles of a phase dealt out over the waves
// of as many workgroups as the phase has work for instead of over the 16 waves of one workgroup (D = 1024: 10.5 ms on one CU; the arithmetic is
// 1 GFLOP).  A launch boundary is the barrier between phases; per 32-column block: the diagonal block (one wave), the panel, the trailing update;
// then W = L^-1 diagonal by diagonal (two launches each), V = D^-1 W and X = W^T V.  Same tile products in the same order per tile as ldl_inverse:
// the same bits.  acc4: four floats of scratch per matrix -- sum of log2 |pivot|, number of negative pivots, ok flag (1 = fine) -- zeroed by kLdlInit.
enum { kLdlInit = 0, kLdlDiag, kLdlPanel, kLdlTrail, kLdlWSum, kLdlWMul, kLdlScale, kLdlX, kLdlFinish };
constexpr int kLdlWavesPerWg = 4;
template <int FD>
__global__ __launch_bounds__(64 * kLdlWavesPerWg) void ns_ldl_phase_kernel(int phase, int jd, const float* __restrict__ src, const float* __restrict__ shift,
                                                                           int shift_stride, float* __restrict__ slabs, size_t slab_stride,
                                                                           float* __restrict__ logdet_out, int D, int gs) {
  constexpr int DP = FD, LD = DP + 1;
  const int m = blockIdx.y, tid = threadIdx.x, lane = tid & 63, li = lane & 31;
  const int gw = blockIdx.x * kLdlWavesPerWg + (tid >> 6), nw = gridDim.x * kLdlWavesPerWg;  // this wave among the launch's waves (per matrix)
  float* sL = slabs + (size_t)m * slab_stride;
  float* sW = sL + (size_t)DP * LD;
  float* sX = sW + (size_t)DP * LD;
  float* dv = sX + (size_t)DP * LD;  // the pivots and the four accumulators: the head of the region behind the three slabs (the Newton steps' residual,
  float* acc4 = dv + DP;             // written only after the factorisation: >= D^2 > DP + 4 floats)
  const int nt = (D + 31) / 32, dp = nt * 32;
  auto store_tile = [&](float* __restrict__ X, int I, int J, const f32x16& acc, float scale) {
#pragma unroll
    for (int e = 0; e < 16; ++e) X[(I * 32 + acc_row(e, lane)) * LD + J * 32 + li] = scale * acc[e];
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  switch (phase) {
    case kLdlInit: {
      const float sh = shift ? shift[(size_t)(m / gs) * shift_stride] : 0.f;
      const float* Am = src + (size_t)m * D * D;
      for (int idx = blockIdx.x * blockDim.x + tid; idx < dp * dp; idx += gridDim.x * blockDim.x) {
        const int i = idx / dp, k = idx - i * dp;
        sL[i * LD + k] = (i < D && k < D) ? Am[(size_t)i * D + k] + ((i == k) ? sh : 0.f) : ((i == k) ? 1.f : 0.f);
        sW[i * LD + k] = 0.f;
      }
      if (blockIdx.x == 0 && tid < 4) acc4[tid] = (tid == 2) ? 1.f : 0.f;
      break;
    }
    case kLdlDiag: {  // one wave
      if (gw != 0) break;
      float l2 = 0.f;
      int neg = 0;
      const bool ok = ldl_diag_block<LD>(sL, sW, dv, 32 * jd, l2, neg);
      if (lane == 0) {
        acc4[0] += l2;
        acc4[1] += (float)neg;
        if (!ok) acc4[2] = 0.f;
      }
      break;
    }
    case kLdlPanel: {  // i > j:  M_ij = A_ij T_jj^T -> sW(i, j);  L_ij = M_ij D_j^-1 -> sL(i, j)
      const int j = jd;
      for (int i = j + 1 + gw; i < nt; i += nw) {
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
      break;
    }
    case kLdlTrail: {  // A_ik -= M_ij L_kj^T for j < k <= i
      const int j = jd, nrem = nt - 1 - j, ntile = nrem * (nrem + 1) / 2;
      for (int t = gw; t < ntile; t += nw) {
        int a = 0, rem = t;
        while (rem > a) {
          rem -= a + 1;
          ++a;
        }
        const int i = j + 1 + a, k = j + 1 + rem;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mfma_tile(sW + (32 * i) * LD + 32 * j, LD, 1, sL + (32 * k) * LD + 32 * j, 1, LD, 32, acc);
#pragma unroll
        for (int e = 0; e < 16; ++e) sL[(i * 32 + acc_row(e, lane)) * LD + k * 32 + li] -= acc[e];
      }
      break;
    }
    case kLdlWSum: {  // diagonal d of W = L^-1: sum_k L_ik W_kj -> sW(i, j) (reads diagonals < d only)
      const int d = jd;
      for (int j = gw; j + d < nt; j += nw) {
        const int i = j + d;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int k = j; k < i; ++k) mfma_tile(sL + (32 * i) * LD + 32 * k, LD, 1, sW + (32 * k) * LD + 32 * j, LD, 1, 32, acc);
        store_tile(sW, i, j, acc, 1.f);
      }
      break;
    }
    case kLdlWMul: {  // W_ij = -T_ii (that sum)
      const int d = jd;
      for (int j = gw; j + d < nt; j += nw) {
        const int i = j + d;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        mfma_tile(sW + (32 * i) * LD + 32 * i, LD, 1, sW + (32 * i) * LD + 32 * j, LD, 1, 32, acc);
        UGLAD_WAVE_SYNC();  // (the tile is read by this wave alone and overwritten by it: every lane has its operands)
        store_tile(sW, i, j, acc, -1.f);
      }
      break;
    }
    case kLdlScale: {  // V = D^-1 W (rows scaled) -> sL, lower tiles incl. the diagonal ones (L is dead)
      for (int idx = blockIdx.x * blockDim.x + tid; idx < dp * dp; idx += gridDim.x * blockDim.x) {
        const int i = idx / dp, k = idx - i * dp;
        if ((k >> 5) <= (i >> 5)) sL[i * LD + k] = sW[i * LD + k] / dv[i];
      }
      break;
    }
    case kLdlX: {  // X = W^T V on the upper tiles, mirrored -> sX
      const int ntile = nt * (nt + 1) / 2;
      for (int t = gw; t < ntile; t += nw) {
        int I = 0, J = t;
        while (J >= nt - I) {
          J -= nt - I;
          ++I;
        }
        J += I;
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
      break;
    }
    default: {  // kLdlFinish: the log-determinant with torch.logdet's rules; NaN everywhere after a zero / NaN pivot
      const bool ok = acc4[2] != 0.f;
      const float nan = __builtin_nanf("");
      if (!ok)
        for (int idx = blockIdx.x * blockDim.x + tid; idx < dp * dp; idx += gridDim.x * blockDim.x) sX[(idx / dp) * LD + idx % dp] = nan;
      if (blockIdx.x == 0 && tid == 0 && logdet_out)
        logdet_out[m] = (!ok || (((int)acc4[1]) & 1)) ? nan : 0.69314718056f * acc4[0];
      break;
    }
  }
}

// loss_partial[m] = -logdet[m] + trace term (wide_loss_trace_kernel's per-row-block sums at `off`)
__global__ void ns_loss_finish_kernel(const float* __restrict__ hdr, size_t hdr_stride, int off, const float* logdet, float* loss_partial, int M,
                                      int D) {  // (logdet may be loss_partial itself)
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* h = hdr + (size_t)m * hdr_stride;
  float tr = 0.f;
  for (int t = 0; t < wide_tiles(D); ++t) tr += h[off + t];
  loss_partial[m] = -logdet[m] + tr;
}

}  // namespace uglad
