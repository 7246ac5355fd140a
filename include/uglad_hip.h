/*
 * uglad_hip.h -- C ABI of libuglad_hip.so: the unrolled-GLAD hot path of Harshs27/uGLAD as hand-written
 * gfx950 (MI355X) HIP kernels.
 *
 * The reference has no FFI: its seam is Python-function level.  Each entry point below
 * names the reference code it replaces (paths relative to the reference repo).  INTEGRATION.md shows the
 * ctypes stub a uGLAD maintainer would add to call these from uglad/glad/glad.py and uglad/main.py.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to a caller-owned, contiguous, row-major fp32 buffer; sizes are explicit;
 *  - `stream` is the hipStream_t the work is enqueued on (pass torch's current stream); nothing synchronises;
 *  - no allocation or free of device memory; the library never keeps a device pointer after returning; entry points that
 *    run the eigensolver take a caller-owned `workspace` of uglad_workspace_floats(M, D) floats;
 *  - process-wide state, all of it host-side: (1) the grouped whole-pass calls set a thread-local group count for their own
 *    duration (restored on return), which the per-step entry points read as 1 otherwise; (2) the kernel-shape switch of
 *    uglad_set_wide_mode().  Nothing else: no cache, no device allocation, no synchronisation -- a caller may capture any
 *    sequence of these calls into a hipGraph of its own;
 *  - return value: 0 ok, <0 argument error (UGLAD_E_*), >0 a hipError_t from the launch;
 *  - scalars that live on the device (lambda_k, the upstream loss gradient) are passed BY POINTER so that the
 *    L-step loop never needs a device->host copy (the reference does one per step: glad.py:147);
 *  - batch index m = 0..M-1 selects matrix m of a (M, D, D) tensor.  D <= uglad_max_dim().
 *
 * Parameter vector `params` (42 floats, the order of GladParams.state_dict(), glad_params.py:13-59):
 *   [0]      theta_init_offset
 *   [1..9]   rho_l1.0.weight (3x3, row = output unit)   [10..12] rho_l1.0.bias
 *   [13..21] rho_l1.2.weight (3x3)                      [22..24] rho_l1.2.bias
 *   [25..27] rho_l1.4.weight (1x3)                      [28]     rho_l1.4.bias
 *   [29..34] lambda_f.0.weight (3x2)                    [35..37] lambda_f.0.bias
 *   [38..40] lambda_f.2.weight (1x3)                    [41]     lambda_f.2.bias
 */
#ifndef UGLAD_HIP_H
#define UGLAD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uglad_stream_t; /* hipStream_t */

#define UGLAD_NPARAM 42
#define UGLAD_NRHO 28 /* params[1..28] */

#define UGLAD_E_NULL (-1)  /* a required pointer is NULL */
#define UGLAD_E_DIM (-2)   /* D < 1 or D > uglad_max_dim(), or M < 1 */
#define UGLAD_E_MODE (-3)  /* unknown sqrt mode / init mode */
#define UGLAD_E_RCCL (-4)  /* RCCL is not loadable in this process, or one of its calls failed */

/* How the matrix square root in Theta_{k+1/2} = 1/2(-b + (b^T b + 4/lam I)^{1/2}) is evaluated on the spectrum of b:
 * EXACT: r_i = sqrt(beta_i^2 + 4/lam).
 * NS10 : the value the reference's 10-step Newton-Schulz iteration returns for that eigenvalue, and in the backward
 *        the reference's 10-step approximate Lyapunov operator (torch_sqrtm.py:13-46) -- bit-for-bit the same
 *        spectral function as the reference, at O(D) / O(D^2) cost.  This is the drop-in default. */
#define UGLAD_SQRT_EXACT 0
#define UGLAD_SQRT_NS10 1

int uglad_version(void);
int uglad_max_dim(void);

/* Largest cond_2(b^T b + 4/lam I) (the cond_max diagnostic of uglad_cell_fwd) up to which parity with the reference is validated by
 * reference-made goldens (tests/golden/regime_*.npz, tests/golden/regime_sweep.json): below it the reference's fp32 Newton-Schulz matrix
 * iteration and this library's spectral evaluation of the same iteration agree within the 1e-4 tolerance; beyond it the reference's own
 * arithmetic is no longer a function of the spectrum alone.  fit() / predict() warn when a pass exceeds it. */
float uglad_validated_cond(void);
/* 1 when a cell call of this shape reports the Gershgorin UPPER BOUND max_i sum_j |A_ij| / (4/lam) >= cond(A) in cond_max (the
 * matrix-iteration path, which has no spectrum to read: D > uglad_max_eig_dim(), and few matrices of 128 < D <= 256), 0 when it reports
 * the condition number itself (the spectral path); `training` = the call saves state for a backward pass.  The bound runs 1.3 ... 1.4 x
 * the true value on sample covariances and can reach ~sqrt(D) x: a caller that warns should hold it to a correspondingly larger threshold. */
int uglad_cond_is_upper_bound(int M, int D, int training, int sqrt_mode);

/* Few, large matrices (D > 128): one workgroup per matrix leaves the chip idle (BASELINE config 5 puts ONE 256 x 256 matrix on
 * each GPU), so the backward cell and the forward cell's part after the eigen-decomposition run as several launches with many
 * workgroups per matrix instead (csrc/wide_bwd.h).  mode -1 (default): chosen per call from (M, D); 0: never; 1: whenever D > 128.
 * Process-wide host-side state; UGLAD_WIDE_BWD=0/1 in the environment presets it.  Same results up to the summation order of the
 * products.  Returns 0, or UGLAD_E_MODE. */
int uglad_set_wide_mode(int mode);

/* Beyond the eigensolver's size (uglad_max_eig_dim() = 256 < D <= uglad_max_dim() = 2048 since round 4; 1024 in round 3) the cell is the reference's own matrix
 * iteration -- ten Newton-Schulz steps forward, ten steps of its Lyapunov iteration backward (torch_sqrtm.py:13-46) -- as dense tile
 * products with many workgroups per matrix (csrc/wide_ns.h): 28 + 60 products per step.  Only UGLAD_SQRT_NS10 exists there
 * (UGLAD_E_MODE otherwise); U's slot of the saved tensors carries the square root, beta's stays unused; cond_max receives the
 * Gershgorin UPPER BOUND of the condition number; Theta_0 and the loss's logdet / inverse use an L D L^T factorisation without
 * pivoting (torch.logdet's rules from the signs of D: finite for an even number of negative eigenvalues, NaN for an odd one; a zero
 * pivot gives NaN where a singular matrix gives -inf in torch).  The entry points of the path (init_theta, cell_fwd / cell_bwd, loss_*, glad_forward* / glad_backward*) take every
 * D <= uglad_max_dim(); uglad_symeig, uglad_cell_fwd_stage2, uglad_tridiagonalize, uglad_covariance, uglad_conditional_mean and
 * uglad_support_metrics stay at uglad_max_eig_dim().  A call on this path takes at most 21845 matrices (UGLAD_E_DIM beyond: matrix x
 * product, up to three products, share one grid dimension).
 * The same path is taken automatically (mode -1, the default) for FEW matrices of 128 < D <= 256 under UGLAD_SQRT_NS10, where one
 * workgroup's Householder chain is most of the spectral cell: training calls (half_out / U_out given) up to 8 matrices and
 * M * ceil(D/64)^2 <= 96, forward-only calls while M * ceil(D/64)^2 <= 256 (one 256 x 256 matrix: 9.4 vs 16.9 ms per 15-step training
 * pass, 4.2 vs 15.5 forward only); uglad_cell_bwd follows the training rule, so it matches the forward call that saved its state.
 * uglad_set_matrix_iteration(0) keeps the spectral path wherever it exists, (1) takes this path for EVERY D (tests, A/B measurements);
 * UGLAD_MATRIX_ITERATION=0/1 in the environment presets it.  Process-wide host-side state; size the workspace after setting it. */
int uglad_max_eig_dim(void);
int uglad_set_matrix_iteration(int mode);

/* Floats of caller-owned device workspace for a batch of M matrices of order D (DP = D rounded up to 32): the tridiagonal
 * form d, e, tau per matrix (3 DP floats), handed from the tridiagonalisation launch to the divide & conquer launch, plus --
 * for 128 < D <= 256, where two D x D fp32 buffers no longer fit the 160 KB of LDS -- two L2-resident DP x (DP+1) slabs per
 * matrix on which the same kernels then work through global pointers; for D > 256 (or every D under uglad_set_matrix_iteration(1))
 * the header and, per matrix, eight D x D fp64 slabs + one fp32 slab, or the three padded F x (F + 1) fp32 slabs + D x D of the
 * factorisation (F = 1024 up to D = 1024, 2048 beyond), whichever is larger (csrc/wide_ns.h).  Every entry point that takes `workspace` accepts a buffer of this size
 * (uglad_cell_bwd and uglad_init_theta_bwd only read it for D > 128 and accept NULL otherwise).  Negative on bad arguments. */
int uglad_workspace_floats(int M, int D);

/* Theta_0.  Replaces glad.py:103-119.  init_diag 0: (S + t I)^-1 -- the reference calls torch.inverse, an LU -- by blocked Cholesky
 * in LDS (D <= 128; a matrix that is not positive definite goes through the eigen path: V diag(1/(s_i + t)) V^T + a Newton step), by
 * the path's eigensolver (128 < D <= 256), by a blocked L D L^T + two Newton steps (D > 256); the result is exactly symmetric and
 * agrees with the LU's to fp32 round-off.  init_diag 1: diag(1/(S_ii + t)).  t = params[0]. */
int uglad_init_theta(const float* S, const float* params, int init_diag, float* theta0, float* workspace, int M, int D,
                     uglad_stream_t stream);

/* d loss / d theta_init_offset, one partial per matrix: gt_partial[m] = -<G0_m, Theta0_m^2> (init_diag 0)
 * or -sum_i G0_ii Theta0_ii^2 (init_diag 1).  Autograd counterpart of glad.py:107-117. */
int uglad_init_theta_bwd(const float* theta0, const float* G0, int init_diag, float* gt_partial, float* workspace, int M,
                         int D, uglad_stream_t stream);

/* lambda_0 = LambdaNN([lambda_init, 0]).  Replaces glad.py:135 (note the reference passes lambda_init in the normF slot).
 * Writes lam_out[0] and the two inputs to lam_in[0..1] (kept for uglad_lambda_bwd). */
int uglad_lambda_init(const float* params, float lambda_init, float* lam_out, float* lam_in, uglad_stream_t stream);

/* One GLAD cell for every matrix of the batch.  Replaces glad.py:139-144 + torch_sqrtm.py:13-29 + glad_params.py:61-81:
 *   b = S/lam - Z_in (symmetric; the upper triangle is read), b = U diag(beta) U^T (batched symmetric eigensolver in LDS),
 *   theta_half = phi(b) assembled as -alpha b + U diag(phi(beta) + alpha beta) U^T (an identity for every alpha; alpha in [0,1] chosen per
 *   matrix to minimise the part that goes through the fp32 eigenvectors, the product on the f32 MFMA, the spectral function in fp64),
 *   Z_out = soft-threshold(theta_half, rhoNN(theta_half, S, Z_in)),  normF_partial[m] = ||Z_out_m - theta_half_m||_F^2.
 * lam points at lambda_k on the device.  half_out / U_out (M,D,D) and beta_out (M,D) may be NULL (inference);
 * when given they are what uglad_cell_bwd needs.  Z_out must not alias Z_in.
 * cond_max (M floats, or NULL): regime diagnostic, cond_max[m] = max(cond_max[m], cond_2(b_m^T b_m + 4/lam I)) -- a RUNNING maximum, so
 * the caller zeroes it before the first step of a pass.  The reference's 10 Newton-Schulz steps (torch_sqrtm.py:13-29) are an accurate
 * square root only while this number is small (hard part 1; validated bound: uglad_validated_cond()). */
int uglad_cell_fwd(const float* S, const float* Z_in, const float* lam, const float* params, float* Z_out,
                   float* half_out, float* U_out, float* beta_out, float* normF_partial, float* cond_max, float* workspace, int M,
                   int D, int sqrt_mode, uglad_stream_t stream);

/* Second launch of uglad_cell_fwd alone (divide & conquer, back-transformation, U phi U^T, rhoNN epilogue), for callers that
 * already ran uglad_tridiagonalize(S, Z_in, lam, Z_out, workspace) on the same stream -- profiling and tests; same arguments
 * as uglad_cell_fwd. */
int uglad_cell_fwd_stage2(const float* S, const float* Z_in, const float* lam, const float* params, float* Z_out,
                          float* half_out, float* U_out, float* beta_out, float* normF_partial, float* cond_max, float* workspace,
                          int M, int D, int sqrt_mode, uglad_stream_t stream);

/* out[0] = sum_i partials[i], summed in index order (deterministic).  Local leg of the per-step normF collective
 * (get_frobenius_norm, glad.py:60-71) and of the loss / gradient reductions. */
int uglad_sum_partials(const float* partials, int n, float* out, uglad_stream_t stream);

/* lambda_{k+1} = LambdaNN([normF_sum * inv_M, lambda_k]).  Replaces glad.py:146-150 + glad_params.py:83-95.
 * normF_sum: device scalar (already summed over the local batch, and all-reduced over ranks when sharded);
 * inv_M = 1 / global batch size (get_frobenius_norm's batch mean).  Writes lam_next[0] and lam_in_next[0..1]. */
int uglad_lambda_step(const float* normF_sum, float inv_M, const float* lam_prev, const float* params, float* lam_next,
                      float* lam_in_next, uglad_stream_t stream);

/* Backward of one cell.  Replaces autograd through glad.py:139-144, torch_sqrtm.py:32-46 and glad_params.py:61-81.
 * G_next = dL/dZ_out.  Writes G_out = dL/dZ_in, ADDS the 28 rhoNN gradients of matrix m to grad_rho_partial[m*28..]
 * (zero it once per backward pass) and WRITES glam_partial[m] = this matrix's contribution to dL/dlambda_k.
 * G_out must not alias G_next.  G_out is symmetric up to rounding (exactly symmetric when one workgroup handles the matrix);
 * only the symmetric part of G_next is used.  For D > 128 the whole workspace is scratch (nothing of a forward call survives). */
int uglad_cell_bwd(const float* G_next, const float* S, const float* Z_in, const float* half, const float* U,
                   const float* beta, const float* lam, const float* params, float* G_out, float* grad_rho_partial,
                   float* glam_partial, float* workspace, int M, int D, int sqrt_mode, uglad_stream_t stream);

/* glasso loss, one partial per matrix.  Replaces main.py:306-311,325-332:
 *   loss_partial[m] = -logdet(Theta_m) + sum_ij S_ij Theta_ji [+ sum_ij log cosh(Theta_ij * ((1 - struct_ij) - delta_ij))].
 * logdet follows torch.logdet: NaN when det < 0, -inf when det = 0.  S holds s_batch matrices (M, or 1 = broadcast,
 * the missing-data call of main.py:620-622); struct (s_batch, D, D) may be NULL.  theta_inv_out (M,D,D) receives
 * Theta^-1 for uglad_loss_bwd.  The caller divides the summed partials by s_batch (main.py:306,315). */
int uglad_loss_fwd(const float* theta, const float* S, int s_batch, const float* struct_theta, float* loss_partial,
                   float* theta_inv_out, float* workspace, int M, int D, uglad_stream_t stream);

/* G = g_up[0] * scale * (-Theta^-T + S^T [+ tanh(Theta o mask) o mask]).  g_up: device scalar (upstream gradient of the
 * loss), scale = 1/s_batch. */
int uglad_loss_bwd(const float* theta, const float* theta_inv, const float* S, int s_batch, const float* struct_theta,
                   const float* g_up, float scale, float* G_out, int M, int D, uglad_stream_t stream);

/* Finish the 42 parameter gradients of one backward pass (sums over the LOCAL batch; the caller all-reduces them):
 *   grad[0]      = sum_m gt_partial[m]
 *   grad[1..28]  = sum_m grad_rho_partial[m][:]
 *   grad[29..41] = sum_k (sum_m glam_partial[k][m]) * dLambdaNN(lam_in[k])/dparams   (inputs are constants, glad_params.py:94)
 * glam_partial is (L, M); lam_in is (L+1, 2) as written by uglad_lambda_init/_step. */
int uglad_finish_grads(const float* gt_partial, const float* grad_rho_partial, const float* glam_partial,
                       const float* lam_in, const float* params, float* grad, int L, int M, uglad_stream_t stream);

/* The whole unrolled pass of glad.py:103-151 enqueued by ONE call: Theta_0, lambda_0, then L x {uglad_cell_fwd,
 * uglad_sum_partials, uglad_lambda_step} -- for the single-process case (a sharded batch needs the all-reduce between the last
 * two and drives the steps itself).  Z holds z_slabs slabs of (M,D,D): step k reads slab k % z_slabs and writes slab
 * (k+1) % z_slabs (z_slabs = L+1 keeps every Theta_k for the backward pass, 2 is enough for inference).  half/U (L,M,D,D) and
 * beta (L,M,D) may be NULL together.  lam (L+1), lam_in (L+1,2), nf_partial (M), nf_sum (1) as in the per-step calls.  cond_max (M floats
 * or NULL) is zeroed here and receives, per matrix, the maximum over the L steps of cond(b^T b + 4/lam I) (see uglad_cell_fwd). */
int uglad_glad_forward(const float* S, const float* params, float lambda_init, int init_diag, int L, float* Z, int z_slabs,
                       float* half, float* U, float* beta, float* lam, float* lam_in, float* nf_partial, float* nf_sum,
                       float* cond_max, float* workspace, int M, int D, int sqrt_mode, uglad_stream_t stream);

/* Its reverse: L x uglad_cell_bwd (ping-ponging gbuf0/gbuf1, each (M,D,D)), uglad_init_theta_bwd, uglad_finish_grads.
 * G_L = dL/dTheta_L; grad receives the 42 gradients.  grad_rho_partial (M,28) is zeroed here; glam_partial (L,M); gt_partial (M). */
int uglad_glad_backward(const float* G_L, const float* S, const float* params, int init_diag, int L, const float* Z,
                        const float* half, const float* U, const float* beta, const float* lam, const float* lam_in,
                        float* gbuf0, float* gbuf1, float* grad_rho_partial, float* glam_partial, float* gt_partial,
                        float* grad, float* workspace, int M, int D, int sqrt_mode, uglad_stream_t stream);

/* Grouped passes (the folds of CV mode as ONE batch): the M matrices are `groups` independent problems of
 * M / groups consecutive matrices, each with its own parameters and its own lambda sequence.  Same arguments as
 * uglad_glad_forward / uglad_glad_backward with: params (groups, 42); lam (L+1, groups); lam_in (L+1, groups, 2);
 * nf_sum (groups); grad (groups, 42).  groups == 1 is the plain call.  M % groups must be 0. */
int uglad_glad_forward_grouped(const float* S, const float* params, float lambda_init, int init_diag, int L, float* Z,
                               int z_slabs, float* half, float* U, float* beta, float* lam, float* lam_in, float* nf_partial,
                               float* nf_sum, float* cond_max, float* workspace, int M, int D, int groups, int sqrt_mode,
                               uglad_stream_t stream);
int uglad_glad_backward_grouped(const float* G_L, const float* S, const float* params, int init_diag, int L, const float* Z,
                                const float* half, const float* U, const float* beta, const float* lam, const float* lam_in,
                                float* gbuf0, float* gbuf1, float* grad_rho_partial, float* glam_partial, float* gt_partial,
                                float* grad, float* workspace, int M, int D, int groups, int sqrt_mode, uglad_stream_t stream);

/* The unrolled pass of ONE RANK of a batch-sharded run, enqueued by one call (collective site i: the batch mean
 * behind get_frobenius_norm, glad.py:60-71,147, is the only coupling between the matrices of a batch).  As uglad_glad_forward on the M
 * local matrices, except that per step the local sum of nf_partial goes through `exchange` -- SUM over the ranks, in place, enqueued
 * on `stream`, no host synchronisation -- before lambda_{k+1} = LambdaNN([sum / m_global, lambda_k]) is formed, on every rank from the
 * same bits.  m_global = matrices over all ranks (the shards may differ in size).  The backward pass needs no exchange (the reference
 * detaches LambdaNN's inputs, glad_params.py:94): uglad_glad_backward on the local matrices, then one SUM of the 42 gradients per epoch.
 * exchange(buf, n, ctx, stream): 0 = ok; anything else is returned as is.  uglad_rccl_allreduce_sum below is one such function. */
typedef int (*uglad_allreduce_fn)(float* device_buf, int n, void* ctx, uglad_stream_t stream);
int uglad_glad_forward_sharded(const float* S, const float* params, float lambda_init, int init_diag, int L, float* Z, int z_slabs,
                               float* half, float* U, float* beta, float* lam, float* lam_in, float* nf_partial, float* nf_sum,
                               float* cond_max, float* workspace, int M, int D, int m_global, int sqrt_mode,
                               uglad_allreduce_fn exchange, void* exchange_ctx, uglad_stream_t stream);

/* RCCL over xGMI as that exchange: ncclAllReduce(SUM, fp32) issued from the library on the compute stream, one per unroll step, so
 * the 4-byte message costs a link latency and no Python.  The library resolves RCCL at run time (dlopen of the copy PyTorch-ROCm has
 * loaded, else the system's librccl.so); it has no link-time dependency on it.  One process per GPU:
 *   rank 0: uglad_rccl_unique_id(&id); ship the 128 bytes to the other ranks (e.g. torch.distributed.broadcast);
 *   every rank, after hipSetDevice: uglad_rccl_comm_init(&id, nranks, rank, &comm);  ... uglad_rccl_comm_destroy(comm).
 * uglad_rccl_allreduce_sum has the signature of uglad_allreduce_fn (ctx = the communicator).  UGLAD_E_RCCL when RCCL cannot be
 * loaded or a call fails. */
typedef struct { char internal[128]; } uglad_rccl_id; /* = ncclUniqueId */
int uglad_rccl_unique_id(uglad_rccl_id* id_out);
int uglad_rccl_comm_init(const uglad_rccl_id* id, int nranks, int rank, void** comm_out);
int uglad_rccl_comm_destroy(void* comm);
int uglad_rccl_comm_count(void* comm, int* nranks_out); /* ncclCommCount: the ranks the communicator spans */
int uglad_rccl_allreduce_sum(float* device_buf, int n, void* comm, uglad_stream_t stream);

/* Consensus over K precision matrices (main.py:700-716, type="min"), split so that a sharded batch can all-reduce
 * in between: partial -> absmin (D,D) = min_k |Theta_k|, signsum (D,D) = sum_k sign(Theta_k);
 * combine -> out = (signsum >= 0 ? +1 : -1) * absmin. */
int uglad_consensus_partial(const float* theta_K, int K, int D, float* absmin, float* signsum, uglad_stream_t stream);
int uglad_consensus_combine(const float* absmin, const float* signsum, int D, float* out, uglad_stream_t stream);

/* Batched symmetric eigendecomposition A_m = U_m diag(beta_m) U_m^T (A symmetric: both triangles are read; beta ascending):
 * Householder tridiagonalisation + divide & conquer + blocked back-transformation, the solver inside uglad_cell_fwd,
 * exported for unit tests.  U must not alias A (its slab doubles as reflector scratch).  No rescaling of the input: measured
 * on structured matrices (profiles/r04_symeig_sweep.txt) the errors are those of unit scale (<= 2.4e-6) for ||A|| from 1e-9 to
 * 1e18 and grow below (1e-4 at 1e-11: intermediate cubes of pole distances leave the fp32 range); the cell's b = S/lam - Z is O(1). */
int uglad_symeig(const float* A, float* U, float* beta, float* workspace, int M, int D, uglad_stream_t stream);

/* Covariance front-end of fit() (replaces prepare_data.py:328-356 get_covariance and, with normalize = 1,
 * the min-max normalisation of prepare_data.py:597-613 / main.py:85): X (K,N,D) row-major tables -> S_out (K,D,D) =
 * sum_n (x_n - mean)(x_n - mean)^T / N of the (normalised) columns.  With eig_scratch != NULL (K*D*D + K*D floats) and a
 * workspace of uglad_workspace_floats(K, D) the reference's repair follows: where the smallest eigenvalue is <= 1e-6,
 * S += (eval_offset - min eig) I.  A constant column under normalize = 1 gives NaN, as in the reference. */
int uglad_covariance(const float* X, int K, int N, int D, int normalize, float eval_offset, float* S_out, float* eig_scratch,
                     float* workspace, uglad_stream_t stream);

/* First half of the eigensolver on its own (unit tests, profiling): Householder tridiagonalisation of A = A0 (A1 == NULL) or
 * A = A0/lam[0] - A1 (the cell's b = S/lam - Z).  Row k of R_m (M, D, D) receives reflector v_k; workspace receives d, e, tau
 * (3 x 32*ceil(D/32) floats per matrix).  uglad_cell_fwd / uglad_symeig / uglad_init_theta / uglad_loss_fwd launch this
 * kernel themselves before their divide & conquer kernel. */
int uglad_tridiagonalize(const float* A0, const float* A1, const float* lam, float* R, float* workspace, int M, int D,
                         uglad_stream_t stream);

/* After the path: conditional Gaussian given observed coordinates, K independent problems.  Replaces
 * conditional_gaussian_with_probabilities + compute_map_estimate (main.py:1176-1260; scipy.linalg.solve / np.linalg.inv /
 * multivariate_normal.pdf on the host) with the path's own eigensolver on the masked precision matrix.
 *   precision (K,D,D) symmetric (upper triangle read); mean (K,D); observed (K,D): non-zero where the coordinate is observed;
 *   values (K,D): read at the observed coordinates.
 *   full_mean (K,D): values at the observed coordinates, mean_u - L_uu^-1 L_uo (x_o - mean_o) elsewhere; clip01 != 0 clamps
 *                    it to [0,1] (compute_map_estimate, main.py:1260);
 *   cond_cov (K,D,D): L_uu^-1 on the (unobserved, unobserved) block, identity on the observed coordinates (required: its slab
 *                    doubles as the solver's reflector scratch);
 *   log_pdf (K) or NULL: log N(map; map, L_uu^-1) = -n_u/2 log(2 pi) + 1/2 logdet L_uu (NaN unless L_uu is positive definite);
 *   scratch: K*D*D floats; workspace: uglad_workspace_floats(K, D). */
int uglad_conditional_mean(const float* precision, const float* mean, const float* observed, const float* values,
                           float* full_mean, float* cond_cov, float* log_pdf, float* scratch, float* workspace, int K, int D,
                           int clip01, uglad_stream_t stream);

/* After the path.  Partial correlations rho_ij = -p_ij / sqrt(p_ii p_jj) from the upper triangle, mirrored,
 * ones on the diagonal (get_partial_correlations, main.py:796-821), K matrices at once. */
int uglad_partial_correlations(const float* precision, float* rho, int K, int D, uglad_stream_t stream);

/* Support-recovery metrics of report_metrics_all (utils/metrics.py:25-108) for K (true, predicted) pairs: out (K, 11) DOUBLES on
 * the device = FDR, TPR, FPR, SHD, nnzTrue, nnzPred, precision, recall, Fbeta, aupr, auc, unrounded (the reference rounds to 3
 * decimals).  Integer counting throughout; AUC / AUPR as sklearn defines them (ties included).  2 <= D <= uglad_max_eig_dim(). */
int uglad_support_metrics(const float* true_theta, const float* pred_theta, double* out, int K, int D, int beta,
                          uglad_stream_t stream);

/* The same decomposition by two-sided cyclic Jacobi (round-robin ordering, Rutishauser rotations): slower, independent of
 * the divide & conquer solver; beta comes back unsorted.  Cross-check only. */
int uglad_symeig_jacobi(const float* A, float* U, float* beta, int M, int D, uglad_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UGLAD_HIP_H */
