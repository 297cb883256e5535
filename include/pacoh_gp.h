/*
 * pacoh_gp.h -- C ABI of the MI355X-native PACOH task-GP hot path (libpacoh_gp.so).
 *
 * The reference (jonasrothfuss/meta_learning_pacoh) has no FFI layer: its boundary for this path
 * is the Python class API of meta_learn/GPR_meta_{mll,svgd,vi}.py, and all arithmetic below that
 * API is delegated to torch/gpytorch calls.  Each entry point here replaces one such group of
 * calls; the comment above it cites the reference lines it stands in for (paths relative to the
 * reference repository root).  The Python host package (meta_learning_pacoh_amd) binds these
 * symbols with ctypes and passes `tensor.data_ptr()` of PyTorch-ROCm tensors.
 *
 * Conventions
 *   - plain C, no exceptions, no allocation, no ownership transfer: every buffer is caller-owned
 *     device memory (gfx950 HBM); scratch is passed in, sized by the *_workspace_bytes() query;
 *   - every launch goes to the caller's `stream` (a hipStream_t passed as void*), no sync inside,
 *     safe to capture into a hipGraph; re-entrant; no global state;
 *   - return value: 0 = launched, <0 = PACOH_E* argument/limit error (nothing launched);
 *     numerical status (Cholesky jitter / failure) is reported per GP in the optional `info[]`;
 *   - layout: row-major, batch-major.  A "GP problem" b in [0,B) is the pair
 *       (task t, hyper-parameter set p) with b = t*P + p   (P = particles / posterior samples).
 *     Arrays shared between problems are addressed with an integer divisor:
 *       z[(b / z_div), n, f]   (z_div = 1: per problem, e.g. NN features; z_div = P: per task, SE kernel on x)
 *       y[(b / y_div), n]      (y_div = P: targets are per task)
 *     hyper-parameters are per set p = b % P:  lengthscale[P,f], outputscale[P] (NULL = 1), noise[P];
 *   - dtype: PACOH_F32 or PACOH_F64 for ALL floating buffers of a call;
 *   - ragged tasks: optional n_valid[(b / y_div)] (int32) gives the number of real points of the
 *     task (<= n); rows beyond it are ignored (treated as an identity block) and receive zero grads.
 */
#ifndef PACOH_GP_H
#define PACOH_GP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PACOH_F32 0
#define PACOH_F64 1

#define PACOH_OK 0
#define PACOH_EINVAL (-1)      /* NULL / negative / inconsistent argument                       */
#define PACOH_ELIMIT (-2)      /* shape outside what the kernels support (see each function)    */
#define PACOH_EDTYPE (-3)
#define PACOH_ELAUNCH (-4)     /* hipGetLastError() after launch != hipSuccess                  */

#define PACOH_MEAN_ZERO 0      /* gpytorch ZeroMean                                              */
#define PACOH_MEAN_VECTOR 1    /* mean[B,n]   (NN mean, models.py:513-514)                       */
#define PACOH_MEAN_CONST 2     /* mean[P]     (ConstantMeanLight, models.py:406-416)             */

#define PACOH_MAX_FEATURES 16  /* f (kernel input dim) <= 16                                     */
/* Kernel family of the GP entry points (round 3).  Default: ARD-RBF.  The family rides in the bits above the feature count of
 * the `f` argument, f_arg = f | (PACOH_KERNEL_x << PACOH_KERNEL_SHIFT), of pacoh_gram_rbf_ard, pacoh_gp_lml_fwd / _fwdbwd /
 * _predict, pacoh_gp_lml_dense / pacoh_gp_predict_dense, and of pacoh_hyper_fwd / pacoh_hyper_bwd / pacoh_step_begin, where it
 * says that the family has ONE raw scale parameter shared by all f input dimensions.
 *   PACOH_KERNEL_COSINE  k(x, x') = os * cos(pi * |x - x'| / period)   -- gpytorch.kernels.CosineKernel, which the reference's own
 *                        suite hands to the single-task learner (tests/test_GPR.py:95-101; GPR_mll.py:66-78 takes any Kernel
 *                        object): `lengthscale[p, 0..f)` all hold the period length; d_lengthscale[b, c] comes back per
 *                        dimension and its sum over c is the gradient of the period.  General (LDS-resident / HBM-resident)
 *                        kernels only: not a hot path. */
#define PACOH_KERNEL_RBF 0
#define PACOH_KERNEL_COSINE 1
#define PACOH_KERNEL_SHIFT 8
#define PACOH_MLP_MAX_HIDDEN_LAYERS 63   /* per-particle MLP: any layer_sizes up to this depth ...          */
#define PACOH_MLP_MAX_WIDTH 65536        /* ... and this width (the reference has no limit: models.py:328-349) */
#define PACOH_SVGD_MAX_PARTICLES 1024    /* SVGD entry points, RBF and IMQ kernel (the reference has no limit; its sweeps use 10 / 50) */

/* info[b] values written by the GP kernels (LAPACK-style): 0 = clean Cholesky; 1..3 = succeeded
 * after adding diagonal jitter base*10^(k-1), base = 1e-6 (f32) / 1e-8 (f64) -- the retry ladder of
 * gpytorch.utils.cholesky.psd_safe_cholesky that the reference relies on; -1 = not positive
 * definite even with jitter (outputs for that problem are NaN). */

int pacoh_abi_version(void);

/* The PACOH_* environment switches (test and A/B switches that force an implementation) are read ONCE, when
 * the library is loaded; no entry point calls getenv.  A host that changes the environment afterwards and wants the change seen
 * calls this (not thread-safe against concurrent launches). */
void pacoh_reload_env(void);

/* ---- A4: Gram build (materialised) -------------------------------------------------------------
 * K[b,i,j] = os_p * exp(-0.5 * sum_k ((z1[b,i,k] - z2[b,j,k]) / l_pk)^2)  (+ noise_p if i==j and
 * add_noise_diag != 0 and z1 == z2 semantics, i.e. square Gram).
 * Replaces SEKernelLight.forward (meta_learn/models.py:428-446) and
 * ScaleKernel(RBFKernel(ard)) (GPR_meta_mll.py:218,223); also the K_xs / K_ss blocks of the exact
 * posterior (GPR_meta_mll.py:174-181).  Any n, m >= 1; f <= 16.  HBM-write bound. */
int pacoh_gram_rbf_ard(const void* z1, int z1_div, const void* z2, int z2_div,
                       const void* lengthscale, const void* outputscale, const void* noise,
                       int add_noise_diag, void* K,
                       int B, int P, int n, int m, int f, int dtype, void* stream);

/* ---- A5+A6: fused per-datapoint log marginal likelihood, small n -------------------------------
 * lml[b] = log N(y; mean, os*K + noise*I) / n_b      (the reference's MLL is divided by n)
 * Replaces gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)(model(x), y) at
 * GPR_meta_mll.py:111-113 and random_gp.py:83-85 (Gram build, +noise, Cholesky with jitter retry,
 * triangular solves, log-det) without ever writing K to HBM.
 * Limits: n <= pacoh_gp_small_max_n(dtype, want_grad).  Optional outputs (NULL to skip):
 * alpha_out[B,n] = (os*K+noise*I)^-1 (y-mean), L_out[B,n,n] lower Cholesky factor, info[B]. */
int pacoh_gp_small_max_n(int dtype, int want_grad);

int pacoh_gp_lml_fwd(const void* z, int z_div, const void* mean, int mean_mode,
                     const void* y, int y_div, const void* lengthscale, const void* outputscale,
                     const void* noise, const int32_t* n_valid,
                     void* lml, void* alpha_out, void* L_out, int32_t* info,
                     int B, int P, int n, int f, int dtype, void* stream);

/* Forward + backward in one launch: also writes the gradient of (g_lml[b] * lml[b]) with respect to
 * the problem's own inputs (replaces loss.backward() through the gpytorch graph,
 * GPR_meta_mll.py:115, svgd.py:16, GPR_meta_vi.py:108):
 *   d_z[B,n,f] (NULL if z is data), d_mean ([B,n] for MEAN_VECTOR, [B] for MEAN_CONST, NULL ok),
 *   d_lengthscale[B,f], d_outputscale[B] (NULL ok), d_noise[B].
 * Gradients are per problem (un-reduced); g_lml NULL means 1.  Closed forms: SURVEY.md 7(4). */
int pacoh_gp_lml_fwdbwd(const void* z, int z_div, const void* mean, int mean_mode,
                        const void* y, int y_div, const void* lengthscale, const void* outputscale,
                        const void* noise, const int32_t* n_valid, const void* g_lml,
                        void* lml, void* d_z, void* d_mean, void* d_lengthscale,
                        void* d_outputscale, void* d_noise, int32_t* info,
                        int B, int P, int n, int f, int dtype, void* stream);

/* ---- A11: exact posterior predictive -----------------------------------------------------------
 * mu[b,s]  = mt[b,s] + K*x (Kxx + noise I)^-1 (y - mean)
 * var[b,s] = os - |L^-1 k*s|^2 + noise                 (diagonal of the predictive covariance)
 * cov[B,m,m] (optional) = K** - K*x (Kxx+noise I)^-1 Kx* + noise I
 * Replaces eval-mode ExactGP.__call__ + likelihood(...) (GPR_meta_mll.py:174-181,
 * GPR_meta_svgd.py:203-212, GPR_meta_vi.py:229-252).  n as for pacoh_gp_lml_fwd; any m >= 1.
 * mean_tst follows mean_mode (vector: [B,m]).  workspace: pacoh_gp_predict_workspace_bytes(). */
size_t pacoh_gp_predict_workspace_bytes(int B, int n, int m, int dtype, int want_cov);

int pacoh_gp_predict(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode,
                     const void* y, int y_div, const void* z_tst, int zt_div, const void* mean_tst,
                     const void* lengthscale, const void* outputscale, const void* noise,
                     const int32_t* n_valid, void* mu, void* var, void* cov, int32_t* info,
                     void* workspace, int B, int P, int n, int m, int f, int dtype, void* stream);

/* ---- dense path (large n): Cholesky-based Gaussian log-density of materialised covariances -----
 * logp[b] = log N(resid[b]; 0, A[b]) * scale,  A[B,n,n] symmetric (lower triangle read), destroyed
 * (overwritten by its Cholesky factor).  With A from pacoh_gram_rbf_ard(..., add_noise_diag=1) and
 * scale = 1/n this is the LML of the large-context configuration (n = 512, fp64); with A = cov from
 * pacoh_gp_predict it is the joint test log-likelihood of RegressionModelMetaLearned.eval
 * (meta_learn/abstract.py:134-163).  Any n; one workgroup per matrix, matrix stays in L2/MALL. */
int pacoh_mvn_logprob_dense(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info,
                            double scale, int B, int n, int dtype, void* stream);

/* ---- large-n GP path: the same LML (+ gradients) and posterior predictive with every n x n matrix in HBM ----------
 * Same arguments, outputs, jitter ladder and ragged-task rules as pacoh_gp_lml_fwdbwd / pacoh_gp_predict (and the same
 * reference lines), for context sets beyond pacoh_gp_small_max_n(): Gram build -> MFMA-panel Cholesky -> in-place
 * triangular inverse -> K^-1 = Z^T Z (batched MFMA GEMM) -> gradient contractions.  d_lengthscale == NULL selects
 * forward only (lml, info).  workspace: the *_workspace_bytes() queries (O(B n^2)).  pacoh_gp_lml_dense takes the size of what it was
 * given: a workspace smaller than the query's answer for B problems makes it run the batch in slabs of whole tasks (problem b = t P + p,
 * z_div / y_div in {1, P}; at least the query's answer for P problems, else PACOH_EINVAL) -- a host bounds the O(B n^2) scratch that way.
 * Context sizes whose rows are not a multiple of 16 bytes (odd n; n mod 4 != 0 in fp32) between 97 and 1024 are run as a ragged batch
 * of the next aligned size inside the call (the padded rows are identity rows; the workspace queries include the padded copies), so
 * that they reach the left-looking / two-level kernels as well (round 6: this lived in the Python binding before).
 * Sizes: n <= 512 runs the left-looking Cholesky / inverse (one workgroup per matrix, rows of a multiple of 16 bytes); 512 < n <= 1024
 * (same alignment) a two-level factorisation + inverse -- diagonal sub-blocks <= 512 on those kernels, the off-diagonal block on an
 * LDS-tiled batched GEMM -- in both dtypes; other sizes the right-looking kernels of rounds 1-3, whose 32-column panel must fit in LDS
 * (n <= ~1000 fp32, ~520 fp64), else PACOH_ELIMIT for a call that needs the inverse (gradients, predictive). */
size_t pacoh_gp_lml_dense_workspace_bytes(int B, int n, int f, int dtype, int want_grad);
int pacoh_gp_lml_dense(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div,
                       const void* lengthscale, const void* outputscale, const void* noise,
                       const int32_t* n_valid, const void* g_lml, void* lml, void* d_z, void* d_mean,
                       void* d_lengthscale, void* d_outputscale, void* d_noise, int32_t* info,
                       void* workspace, size_t workspace_bytes, int B, int P, int n, int f, int dtype, void* stream);
size_t pacoh_gp_predict_dense_workspace_bytes(int B, int n, int m, int dtype);
int pacoh_gp_predict_dense(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode, const void* y,
                           int y_div, const void* z_tst, int zt_div, const void* mean_tst,
                           const void* lengthscale, const void* outputscale, const void* noise,
                           const int32_t* n_valid, void* mu, void* var, void* cov, int32_t* info,
                           void* workspace, int B, int P, int n, int m, int f, int dtype, void* stream);

/* ---- A2: per-particle ("vectorised") MLP ------------------------------------------------------
 * out[b] = MLP_{theta_p}(x[b / x_div]),  tanh hidden layers, linear output.  theta points at the
 * network's block inside the particle matrix, consecutive particles are theta_stride elements apart;
 * block layout per layer: bias[out] then weight[out,in] row-major -- the reference's flattened layout
 * (LinearVectorized.parameter_shapes, models.py:319-323).  P = 1 gives the shared-weight network of
 * PACOH-MAP (NeuralNetwork.forward, models.py:211-217).
 * Replaces NeuralNetworkVectorized.forward / LinearVectorized.forward (models.py:295-317,343-349).
 * hidden: HOST array of n_hidden layer widths -- ANY layer_sizes, as the reference (models.py:328-349; the launchers run
 * 4 x 32 and 4 x 128: experiments/meta_GPR_SVGD_base_exp.py:29-30, meta_GPR_mll_base_exp.py:29-30), n_hidden >= 0.
 * Shapes outside the register-/LDS-resident kernels (fp32: <= 4 hidden layers of width <= 32 with d_in <= 4, d_out <= 2, or
 * <= 2 layers with d_in <= 16, d_out <= 8; both dtypes: <= 3 layers of width <= 64) run layer by layer on the matrix cores
 * through `workspace` (pacoh_mlp_fwd_workspace_bytes; 0 bytes = not needed, NULL allowed). */
size_t pacoh_mlp_fwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out,
                                     int dtype);
int pacoh_mlp_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                  int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out, void* workspace,
                  int B, int n, int dtype, void* stream);

/* Backward: d_theta[P, D_net] (row stride d_theta_stride) = sum over the problems b of particle p of
 * d out[b]/d theta_p ^T g_out[b]; activations are recomputed from x (nothing is saved by the forward).
 * Deterministic two-stage reduction through `workspace` (pacoh_mlp_bwd_workspace_bytes). If
 * `accumulate` != 0 the result is added to d_theta, else it overwrites the block. */
size_t pacoh_mlp_bwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden,
                                     int n_hidden, int d_out, int dtype);

int pacoh_mlp_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                  int d_in, const int32_t* hidden, int n_hidden, int d_out, const void* g_out,
                  void* d_theta, long d_theta_stride, int accumulate, void* workspace,
                  int B, int n, int dtype, void* stream);

/* The mean network AND the kernel-feature network of one step (VectorizedGP.forward evaluates both on the same inputs,
 * random_gp.py:54-68; LearnedGPRegressionModel.forward, models.py:505-519) in one call: two networks of the SAME hidden
 * shape whose blocks start at element offsets off_a / off_b of the theta rows (theta, d_theta point at the ROW start here).
 * Same semantics as two pacoh_mlp_fwd / pacoh_mlp_bwd calls; on the fused fp32 path it is ONE launch.
 * Activation stash (what autograd's saved tensors are to the reference's backward, models.py:313-315): `stash` (optional, NULL =
 * none; pacoh_mlp2_stash_bytes() bytes, 0 = this shape keeps none) receives the activations of every hidden layer but the first from the forward
 * (round 6: on the layer-by-layer path -- fp64, wide or deep networks -- each network's repacked weights and all its hidden activations, so that
 * the backward neither repacks nor recomputes: 18 launches less per step with two 4 x 128 networks);
 * handed to the pacoh_mlp2_bwd call of the SAME x / theta / shapes it replaces their recomputation.  Same results either way. */
size_t pacoh_mlp2_fwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                                      int d_out_b, int dtype);
size_t pacoh_mlp2_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                              int d_out_b, int dtype);
int pacoh_mlp2_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                   const int32_t* hidden, int n_hidden, long off_a, int d_out_a, void* out_a, long off_b, int d_out_b,
                   void* out_b, void* workspace, void* stash, int B, int n, int dtype, void* stream);
size_t pacoh_mlp2_bwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                                      int d_out_b, int dtype);
int pacoh_mlp2_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                   const int32_t* hidden, int n_hidden, long off_a, int d_out_a, const void* g_a, long off_b,
                   int d_out_b, const void* g_b, void* d_theta, long d_theta_stride, int accumulate,
                   void* workspace, const void* stash, int B, int n, int dtype, void* stream);
/* The optimizer step of a PACOH-MAP iteration folded into the gradient epilogue (round 3): with ONE parameter row (P = 1) and no
 * exchange between gradient and update (world size 1), pacoh_hyper_bwd / pacoh_mlp_bwd_hyper / pacoh_mlp2_bwd_hyper apply AdamW
 * (torch.optim.AdamW's op order, as pacoh_adam_step_dev) to every gradient entry in the launch that finishes it -- the slab
 * reduction's threads hold the networks' entries, the hyper-parameter reduction's blocks the rest -- restricted to the column
 * ranges [seg_lo[k], seg_hi[k]) (learning_mode), advance *step_counter and add the loss (lik[0]) to *loss_cum: the AdamW launch
 * of GPR_meta_mll.py:115-117 disappears.  Where the fused slab reduction does not apply the same calls issue pacoh_adam_step_dev
 * per range behind the gradient, so that the caller never has to know.  scalars: the PACOH_SC_ADAM block (4 values, device). */
/* next (optional; fused fp32 networks only -- pacoh_mlp_fused_path(); PACOH_ELIMIT otherwise): the pipelined feed of "The pipelined
 * SVGD step" below for PACOH-MAP.  The backward launch of the call advances *counter, the slab-reduction launch behind it reads the
 * step's scalars from sc2[*counter & 1] (`scalars` and `step_counter` above are then unused), publishes softplus of the raw
 * hyper-parameters it has just updated to ls[1,f] / os[1] / noise[1] (+ noise_floor), copies row *counter + 1 of sc_all into the other
 * row of sc2 and gathers that row's tb tasks into out_x / out_y / out_n_valid: an iteration is FOUR launches (forward, GP, backward,
 * reduction); pacoh_step_begin runs once per chunk of iterations (row 0, then *counter := -1). */
typedef struct pacoh_step_next {
    int64_t* counter; void* sc2; int n_sc;
    const int64_t* idx_all; int tb; const void* sc_all;
    const void* x; const void* y; const int32_t* n_valid; void* out_x; void* out_y; int32_t* out_n_valid; int n, d;
    double noise_floor; void* ls; void* os; void* noise;
} pacoh_step_next;
typedef struct pacoh_adam_inline {
    void* param; void* exp_avg; void* exp_avg_sq;
    const void* scalars;
    double beta1, beta2;
    int n_seg; int seg_lo[4]; int seg_hi[4];
    int64_t* step_counter; void* loss_cum;
    const pacoh_step_next* next;
} pacoh_adam_inline;
/* 1 if these network shapes run on the fused fp32 kernels, else 0 (host-side query) */
int pacoh_mlp_fused_path(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype);

/* pacoh_mlp2_bwd followed by pacoh_hyper_bwd (below) on the same d_theta rows -- the whole gradient epilogue of a step
 * (loss.backward() reaching the networks AND the raw GP hyper-parameters: GPR_meta_mll.py:115, svgd.py:16) in one call.  On the
 * fused fp32 path the hyper-parameter reduction runs in extra workgroups of the backward's slab-reduction launch: one launch and one
 * launch boundary less per step.  Arguments: those of pacoh_mlp2_bwd (accumulate must be 0), then those of pacoh_hyper_bwd with
 * grad = d_theta. */
int pacoh_mlp2_bwd_hyper(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                         const int32_t* hidden, int n_hidden, long off_a, int d_out_a, const void* g_a, long off_b,
                         int d_out_b, const void* g_b, void* d_theta, long d_theta_stride, int accumulate,
                         void* workspace, const void* stash, int B, int n,
                         int T, int off_ls, int f, int off_os, int off_noise, int off_const, const void* d_lengthscale,
                         const void* d_outputscale, const void* d_noise, const void* d_const, const void* lml, void* lik,
                         double lik_scale, const int32_t* info, int32_t* fail_flag, void* svgd_workspace, int svgd_P, int svgd_D,
                         const pacoh_adam_inline* opt, int dtype, void* stream);

/* The same for configurations with ONE network (NN mean with an SE kernel: BASELINE config #2; or NN kernel features with a
 * constant mean): pacoh_mlp_bwd (accumulate = 0) followed by pacoh_hyper_bwd, the reduction riding in the slab reduction's launch
 * on the fused fp32 path.  theta / d_theta point at the network's block inside the rows, theta_rows / grad_rows at the rows.
 * stash (optional): the activation stash as for the two-network calls -- pacoh_mlp_stash_bytes() bytes, filled by
 * pacoh_mlp_fwd_stash (= pacoh_mlp_fwd with that one argument more) on the same inputs. */
size_t pacoh_mlp_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype);
int pacoh_mlp_fwd_stash(const void* x, int x_div, const void* theta, long theta_stride, int P,
                        int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out, void* workspace, void* stash,
                        int B, int n, int dtype, void* stream);
int pacoh_mlp_bwd_hyper(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                        const int32_t* hidden, int n_hidden, int d_out, const void* g_out, void* d_theta, long d_theta_stride,
                        void* workspace, const void* stash, int B, int n,
                        const void* theta_rows, void* grad_rows, int T, int off_ls, int f, int off_os, int off_noise,
                        int off_const, const void* d_lengthscale, const void* d_outputscale, const void* d_noise, const void* d_const,
                        const void* lml, void* lik, double lik_scale, const int32_t* info, int32_t* fail_flag,
                        void* svgd_workspace, int svgd_P, int svgd_D, const pacoh_adam_inline* opt, int dtype, void* stream);

/* ---- A3 + A7: parameter transforms, hyper-prior ------------------------------------------------
 * softplus with optional floor, forward:  out = log(1+exp(raw)) + floor            (random_gp.py:69-74;
 * MAP: gpytorch Positive / GreaterThan(1e-3) constraints, GPR_meta_mll.py:54-55)
 * backward: d_raw = g * sigmoid(raw).  Elementwise over `count` values. */
int pacoh_softplus_fwd(const void* raw, void* out, double floor, long count, int dtype, void* stream);
int pacoh_softplus_bwd(const void* raw, const void* g, void* d_raw, int accumulate, long count,
                       int dtype, void* stream);

/* All hyper-parameter transforms of one step in one launch: for every particle p read the raw values inside
 * theta[P, theta_stride] at element offsets off_ls (f values), off_os (-1: no outputscale), off_noise and write
 * ls[P,f] = softplus, os[P] = softplus, noise[P] = softplus + noise_floor.  Same reference lines as softplus. */
int pacoh_hyper_fwd(const void* theta, long theta_stride, int P, int off_ls, int f, int off_os, int off_noise,
                    double noise_floor, void* ls, void* os, void* noise, int dtype, void* stream);

/* ... and their backward: grad[p, off] = sigmoid(raw) * sum_t d_x[t, p, .] for lengthscale / outputscale / noise
 * (d_ls[T,P,f], d_os[T,P] or NULL, d_noise[T,P]: the per-problem outputs of pacoh_gp_lml_fwdbwd) and the plain sum
 * for a constant mean (d_const[T,P] at off_const, or NULL / -1).  Optionally (lml, lik both non-NULL) the same pass over the
 * tasks also writes lik[p] = lik_scale * sum_t lml[t,p], the likelihood term of RandomGPMeta.log_prob (random_gp.py:204-222).
 * Deterministic (fixed summation order).
 * info[T*P] / fail_flag (both or neither): the per-problem status of pacoh_gp_lml_fwdbwd rides along; *fail_flag |= 1 if any
 * problem's Cholesky failed even with jitter -- where gpytorch's psd_safe_cholesky raises NotPSDError; the host reads the flag
 * at its next synchronisation point and raises there.
 * svgd_workspace (optional; pacoh_svgd_update_dev_workspace_bytes(svgd_P, svgd_D), svgd_P <= 64): one more workgroup of this launch
 * computes the SVGD step's median-heuristic bandwidth (svgd.py:45-51) from the particles' distance matrix at the head of that
 * workspace into the workspace's bandwidth slot, where pacoh_svgd_update_next(bandwidth_ready = 1) picks it up: the register sort
 * (7 us of the update's dependent chain, behind the all-reduce at N > 1) runs beside the reduction instead. */
int pacoh_hyper_bwd(const void* theta, long theta_stride, int P, int T, int off_ls, int f, int off_os, int off_noise,
                    int off_const, const void* d_ls, const void* d_os, const void* d_noise, const void* d_const,
                    void* grad, long grad_stride, const void* lml, void* lik, double lik_scale,
                    const int32_t* info, int32_t* fail_flag, void* svgd_workspace, int svgd_P, int svgd_D,
                    const pacoh_adam_inline* opt, int dtype, void* stream);

/* logp[p] = sum_d log N(theta[p,d]; prior_mean[d], prior_std[d]);  grad[p,d] (optional, += scaled):
 * grad += grad_scale * d logp / d theta.  Replaces CatDist.log_prob over the Normal blocks
 * (random_gp.py:128-157,179-180; models.py:159-181) and its autograd backward. */
int pacoh_prior_logprob_grad(const void* theta, const void* prior_mean, const void* prior_std,
                             void* logp, void* grad, double grad_scale, int P, int D, int dtype,
                             void* stream);
/* The same for a captured step (round 4, the IMQ-SVGD update): score[p,d] := *score_scale * score[p,d] + prior_factor * d logp / d theta
 * with the likelihood pre-factor (random_gp.py:209-212, 221-222) read from device memory (a step-scalar row's
 * PACOH_SC_SCORE_SCALE entry): pacoh_scale_dev + pacoh_prior_logprob_grad in one launch. */
int pacoh_prior_score_dev(const void* theta, const void* prior_mean, const void* prior_std, void* score,
                          double prior_factor, const void* score_scale, int P, int D, int dtype, void* stream);

/* ---- A9: SVGD update direction -----------------------------------------------------------------
 * phi[i,:] = ( sum_j k_ij score[j,:] + 2 gamma sum_j k_ij (X[i,:] - X[j,:]) ) / P,
 * k_ij = exp(-gamma |X_i - X_j|^2), gamma = 1/(1e-8 + 2 bw^2); bandwidth <= 0 selects the median
 * heuristic bw = sqrt(median(|X_i-X_j|^2 over the full PxP matrix incl. the zero diagonal) /
 * (2 ln(P+1))), numpy-median semantics.  Replaces SVGD.phi + RBF_Kernel (meta_learn/svgd.py:12-59).
 * P <= PACOH_SVGD_MAX_PARTICLES (up to 64 particles the median is a register sort inside the consuming kernel, beyond that one
 * extra launch finds it by bisection).  workspace: pacoh_svgd_workspace_bytes().  neg != 0 writes -phi (the "gradient" handed to
 * the optimizer, svgd.py:27).  bw_out (optional, 1 value): the bandwidth used. */
size_t pacoh_svgd_workspace_bytes(int P, int D, int dtype);
int pacoh_svgd_phi(const void* X, const void* score, double bandwidth, int neg, void* phi,
                   void* bw_out, void* workspace, int P, int D, int dtype, void* stream);

/* One whole SVGD step on the particles in two launches (distances; bandwidth + kernel row + update):
 * X_out[i,:] = optimizer_step(X[i,:], grad = -phi[i,:]) with phi as above computed from score[j,:] + prior_factor *
 * d log N(X[j,:]; prior_mean, prior_std) / dX (prior_mean/std NULL: score is used as is).  use_adam != 0: torch.optim.Adam
 * (lr, beta1, beta2, eps, 1-based step; state exp_avg / exp_avg_sq updated in place); use_adam == 0: X_out = X + lr * phi
 * (SGD on grad = -phi).  X_out must not alias X.  Replaces RandomGPMeta's prior backward + SVGD.step
 * (random_gp.py:128-157, svgd.py:12-28, GPR_meta_svgd.py:220-223).  workspace as for pacoh_svgd_phi. */
int pacoh_svgd_update(const void* X, const void* score, const void* prior_mean, const void* prior_std,
                      double prior_factor, double bandwidth, int use_adam, double lr, double beta1, double beta2,
                      double eps, long step, void* exp_avg, void* exp_avg_sq, void* X_out, void* bw_out,
                      void* workspace, int P, int D, int dtype, void* stream);

/* ---- whole steps as hipGraphs: per-step operands in device memory ----------------------------------------------------------
 * A meta-training step is ~12 launches; issued one by one from Python the host needs ~0.35 ms per step, more than the GPU
 * needs on small configurations or small per-GPU shards.  The step is therefore captured once and replayed; everything that
 * changes from step to step -- the sampled task indices (GPR_meta_svgd.py:102, GPR_meta_mll.py:109), the harmonic pre-factor
 * of the batch (random_gp.py:209-212), the learning rate of the StepLR schedule and Adam's bias corrections -- is uploaded for
 * many steps at once and selected on the device:
 *   pacoh_step_select: row = *counter; idx_out[0..tb) = idx_all[row, :]; sc_out[0..n_sc) = sc_all[row, :]; aux_out[0..n_aux) =
 *     aux_all[row, :] (optional payload of `dtype` values: PACOH-VI's reparameterisation noise of the step); *counter = row + 1.
 *   step scalars (one row of `dtype` values, PACOH_SC_COUNT of them):
 *     [PACOH_SC_SCORE_SCALE] factor on the likelihood score (the pre-factor), [PACOH_SC_LR] learning rate,
 *     [PACOH_SC_ADAM .. +3] = {1 - lr*weight_decay, lr/(1-beta1^step), sqrt(1-beta2^step), eps}: the operand block of
 *     pacoh_adam_step_dev.
 *   pacoh_scale_dev:   buf[0..count) *= *scalar   (scalar in device memory, e.g. &sc[PACOH_SC_SCORE_SCALE]).
 *   pacoh_svgd_update_dev: pacoh_svgd_update with score_scale, lr and the Adam scalars read from `scalars` (a step-scalar row):
 *     phi is built from score_scale * score[j,:] + prior_factor * d log prior / dX; X is updated IN PLACE (the distance launch
 *     snapshots the particles into the workspace first: pacoh_svgd_update_dev_workspace_bytes). */
#define PACOH_SC_SCORE_SCALE 0
#define PACOH_SC_LR 1
#define PACOH_SC_ADAM 4
#define PACOH_SC_COUNT 8
int pacoh_step_select(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                      int64_t* counter, int64_t* idx_out, void* sc_out, void* aux_out, int dtype, void* stream);
int pacoh_scale_dev(void* buf, const void* scalar, long count, int dtype, void* stream);
/* pacoh_step_begin: the first launch of a captured step -- pacoh_step_select (without the index copy), pacoh_gather_tasks on
 * idx_all[row, :] (tb tasks; x[T,n,d], y[T,n], optional n_valid) and, when theta is given, pacoh_hyper_fwd, all in ONE launch;
 * advance != 0: a one-thread launch behind it advances *counter; advance == 0: the caller hands `counter` to the step's LAST
 * launch instead (pacoh_svgd_update_dev / pacoh_adam_step_dev, step_counter), which advances it -- nothing in between reads it.
 * svgd_X (optional, [svgd_P, svgd_D]): the step's particles; their squared-distance matrix and the snapshot the in-place update
 * reads are then produced by extra workgroups of this launch into svgd_workspace (pacoh_svgd_update_dev_workspace_bytes), and
 * pacoh_svgd_update_dev is called with dist_done = 1.  (`ticket`: reserved, one int32 of device memory.)
 * Every launch of a step costs ~4.5 us between dependent kernels, more than these kernels need: six launches become one. */
int pacoh_step_begin(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                     int64_t* counter, int32_t* ticket, void* sc_out, void* aux_out,
                     const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y, int32_t* out_n_valid, int n, int d,
                     const void* theta, long theta_stride, int P, int off_ls, int f, int off_os, int off_noise, double noise_floor,
                     void* ls, void* os, void* noise, int advance, const void* svgd_X, void* svgd_workspace, int svgd_P, int svgd_D,
                     int dtype, void* stream);
/* pacoh_step_begin for a PACOH-VI step with a diagonal posterior (round 3): the same launch also draws the step's S samples
 * theta[S,D] = loc + exp(scale) * eps from posterior[2,D] and the step's noise row (aux row of n_aux = S*D values), their log q[S]
 * (pacoh_vi_sample: Normal(loc, scale.exp()).to_event(1).rsample / .log_prob, random_gp.py:244-248, GPR_meta_vi.py:220-221) and
 * the samples' transformed hyper-parameters ls[S,f] / os[S] / noise[S] (pacoh_hyper_fwd, random_gp.py:69-74; ls = NULL: not
 * wanted): two launches less per step. */
int pacoh_step_begin_vi(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                        int64_t* counter, int32_t* ticket, void* sc_out, void* aux_out,
                        const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y, int32_t* out_n_valid, int n, int d,
                        const void* posterior, int S, int D, void* theta_out, void* log_q_out,
                        int off_ls, int f, int off_os, int off_noise, double noise_floor, void* ls, void* os, void* noise,
                        int advance, int dtype, void* stream);
size_t pacoh_svgd_update_dev_workspace_bytes(int P, int D, int dtype);
int pacoh_svgd_update_dev(void* X, const void* score, const void* prior_mean, const void* prior_std,
                          double prior_factor, double bandwidth, int use_adam, const void* scalars, double beta1,
                          double beta2, void* exp_avg, void* exp_avg_sq, void* bw_out, void* workspace, int P, int D,
                          int dist_done, int64_t* step_counter, int dtype, void* stream);

/* The pipelined SVGD step (round 3; csrc/step_tail.h): what pacoh_step_begin does for a step is spread over launches the step has
 * anyway, so that a captured step is FIVE launches (forward, GP, backward, slab reduction, update) instead of six.
 *   prologue of a chunk of steps (host): pacoh_step_begin on row 0 (sc_out = sc2 row 0, theta given, no svgd_X, advance = 0),
 *     then *counter := -1;
 *   pacoh_mlp2_fwd_svgd = pacoh_mlp2_fwd + pacoh_svgd_dist_advance: *counter += 1, the particles' squared distances and their
 *     snapshot into svgd_workspace -- in extra workgroups of the forward launch on the fused fp32 path, as a launch behind it
 *     elsewhere (pacoh_svgd_dist_advance alone: configurations without the two networks);
 *   pacoh_svgd_update_next = pacoh_svgd_update_dev (dist_done) with the step scalars read from sc2[*counter & 1][n_sc]; its own
 *     threads write softplus of the hyper-parameter entries they have just updated to ls[P,f] / os[P] / noise[P] (ls = NULL: not
 *     wanted), and extra workgroups of the launch copy row *counter + 1 of sc_all into sc2[(*counter + 1) & 1] and gather that
 *     row's tb tasks (idx_all[row, tb]; x[T,n,d], y[T,n], optional n_valid) into out_x / out_y / out_n_valid: idx_all and sc_all
 *     must hold one valid row beyond the last step of the chunk.  bandwidth_ready != 0 (bandwidth <= 0, P <= 64): the median
 *     bandwidth is read from the workspace's bandwidth slot (pacoh_hyper_bwd / pacoh_mlp2_bwd_hyper with svgd_workspace).
 * Replaces, together: the softplus transforms of the raw hyper-parameters (random_gp.py:69-74), the gather of the drawn task
 * batch (GPR_meta_svgd.py:102) and the pairwise distances of SVGD.phi / RBF_Kernel (svgd.py:12-59) as separate launches. */
int pacoh_svgd_dist_advance(const void* X, void* workspace, int P, int D, int64_t* counter, int dtype, void* stream);
int pacoh_mlp2_fwd_svgd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                        const int32_t* hidden, int n_hidden, long off_a, int d_out_a, void* out_a, long off_b, int d_out_b,
                        void* out_b, void* workspace, void* stash, int B, int n,
                        const void* svgd_X, void* svgd_workspace, int svgd_P, int svgd_D, int64_t* counter,
                        int dtype, void* stream);
int pacoh_svgd_update_next(void* X, const void* score, const void* prior_mean, const void* prior_std, double prior_factor,
                           double bandwidth, int use_adam, double beta1, double beta2, void* exp_avg, void* exp_avg_sq,
                           void* bw_out, void* workspace, int P, int D,
                           const int64_t* counter, void* sc2, int n_sc, const int64_t* idx_all, int tb, const void* sc_all,
                           const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y,
                           int32_t* out_n_valid, int n, int d,
                           int off_ls, int f, int off_os, int off_noise, double noise_floor, void* ls, void* os, void* noise,
                           int bandwidth_ready, int dtype, void* stream);

/* Same update direction with the IMQ particle kernel k_ij = (alpha + sum_d (X_jd - X_id)^2 / h_d)^beta
 * (alpha > 0, beta < 0).  bandwidth > 0: h_d = bandwidth for every d.  bandwidth <= 0: per-dimension median
 * heuristic h_d = lower-median_{a<b} (X_bd - X_ad)^2 / ln(P+1) (torch.median semantics), written to h_out[D]
 * (required in that case); phi then also carries the derivative through h_d that the reference's autograd
 * produces (the bandwidth is built from the differentiable squared differences).
 * Replaces SVGD.phi + IMQSteinKernel (meta_learn/svgd.py:12-23, 63-97; selected at GPR_meta_svgd.py:176-177).
 * P <= PACOH_SVGD_MAX_PARTICLES (round 3; the pair table of rounds 1-2 capped it at 64).  workspace: pacoh_svgd_imq_workspace_bytes(). */
size_t pacoh_svgd_imq_workspace_bytes(int P, int D, int dtype);
int pacoh_svgd_phi_imq(const void* X, const void* score, double alpha, double beta, double bandwidth,
                       int neg, void* phi, void* h_out, void* workspace, int P, int D, int dtype,
                       void* stream);

/* ---- A8/A9/A10: optimizer step -----------------------------------------------------------------
 * One fused Adam / AdamW (decoupled weight decay) step over `count` parameters, state m,v in place;
 * `step` is the 1-based step count (bias correction), matching torch.optim.Adam / AdamW defaults
 * (GPR_meta_mll.py:255, GPR_meta_svgd.py:220-223, GPR_meta_vi.py:258-261). */
int pacoh_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq,
                    double lr, double beta1, double beta2, double eps, double weight_decay,
                    long step, long count, int dtype, void* stream);

/* Same update with the step-dependent scalars in DEVICE memory, scalars = {1 - lr*weight_decay, lr/(1-beta1^step),
 * sqrt(1-beta2^step), eps} (4 values of `dtype`), so that the launch can be captured once in a hipGraph and replayed
 * every iteration while the host only refreshes the 4 scalars.  step_counter (optional): advanced by one (see pacoh_step_begin).
 * loss_cum / loss (optional, one value each): *loss_cum += *loss, the running sum behind the averaged loss the reference logs
 * (GPR_meta_mll.py:119-125) -- one launch less per PACOH-MAP iteration. */
int pacoh_adam_step_dev(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, const void* scalars,
                        double beta1, double beta2, long count, int64_t* step_counter, void* loss_cum, const void* loss,
                        int dtype, void* stream);

/* The update half of a captured PACOH-VI step (diagonal posterior, Adam) in one launch: pre-factor scalars[PACOH_SC_SCORE_SCALE] on
 * the likelihood score[S,D] and values lik[S] (both left untouched), hyper-prior score and log-density at theta[S,D], the ELBO value
 * -> loss_out[1], the reparameterisation gradient of pacoh_vi_grad and the Adam step of pacoh_adam_step_dev on posterior[2,D]
 * (GPR_meta_vi.py:216-224, 258-261); step_counter as for pacoh_adam_step_dev.  workspace: pacoh_vi_update_dev_workspace_bytes(),
 * zeroed by the caller before the first use (every launch leaves it ready for the next). */
size_t pacoh_vi_update_dev_workspace_bytes(int D, int dtype);
int pacoh_vi_update_dev(void* posterior, const void* eps, const void* theta, const void* score, const void* lik,
                        const void* log_q, const void* prior_mean, const void* prior_std, double prior_factor,
                        const void* scalars, double beta1, double beta2, void* exp_avg, void* exp_avg_sq, void* loss_out,
                        int64_t* step_counter, void* workspace, int S, int D, int dtype, void* stream);

/* y += alpha * x: the plain SGD update of the optimizer='SGD' option (torch.optim.SGD(lr), GPR_meta_mll.py:257). */
int pacoh_axpy(void* y, const void* x, double alpha, long count, int dtype, void* stream);

/* ---- A10: PACOH-VI, diagonal Gaussian hyper-posterior ------------------------------------------------
 * posterior[2,D] = {loc, scale (= log std)}; eps[S,D] standard normal draws (host RNG stream of the reference).
 * sample: theta[s,:] = loc + exp(scale)*eps[s,:]  and  log_q[s] = log N(theta_s; loc, exp(scale)^2)
 * (Normal(loc, scale.exp()).to_event(1).rsample / .log_prob, random_gp.py:244-248, GPR_meta_vi.py:220-221).
 * grad: d(-mean_s elbo_s)/d posterior from the per-sample score (GPR_meta_vi.py:221-224 + backward):
 *   grad[0,:] = -mean_s score[s,:];  grad[1,:] = -mean_s (score[s,:]*exp(scale)*eps[s,:] + prior_factor). */
int pacoh_vi_sample(const void* posterior, const void* eps, void* theta, void* log_q, int S, int D, int dtype, void* stream);
int pacoh_vi_grad(const void* posterior, const void* eps, const void* score, double prior_factor, void* grad,
                  int S, int D, int dtype, void* stream);

/* Full-covariance hyper-posterior (cov_type='full'): posterior[D+1,D] = {loc; tril_cov[D,D]} (entries above the
 * diagonal are ignored, as torch.tril does).  theta[s,:] = loc + tril(tril_cov) eps[s,:],
 * log_q[s] = log N(theta_s; loc, L L^T) (optional).  grad[D+1,D]: grad[0,:] = -mean_s score[s,:];
 * grad[1+i,j] = -mean_s score[s,i] eps[s,j] - [i==j] prior_factor / L[i,i] for j <= i, exactly 0 above the diagonal.
 * Replaces MultivariateNormal(loc, scale_tril=tril(tril_cov)).rsample / .log_prob and the autograd backward
 * (random_gp.py:249-251, GPR_meta_vi.py:220-224). */
int pacoh_vi_sample_full(const void* posterior, const void* eps, void* theta, void* log_q, int S, int D, int dtype,
                         void* stream);
int pacoh_vi_grad_full(const void* posterior, const void* eps, const void* score, double prior_factor, void* grad,
                       int S, int D, int dtype, void* stream);

/* ---- A12: the step's task batch -----------------------------------------------------------------------------------
 * out_x[b,:,:] = x[idx[b],:,:], out_y[b,:] = y[idx[b],:], out_n_valid[b] = n_valid[idx[b]] (both NULL for equal-sized tasks)
 * for the Tb task indices drawn by the host (with replacement: GPR_meta_mll.py:109, GPR_meta_svgd.py:102) -- one launch
 * instead of three framework gathers.  x[T,n,d], y[T,n], idx int64. */
int pacoh_gather_tasks(const void* x, const void* y, const int32_t* n_valid, const int64_t* idx, void* out_x,
                       void* out_y, int32_t* out_n_valid, int Tb, int n, int d, int dtype, void* stream);

/* ---- reductions used by the host between kernels ----------------------------------------------
 * out[p, :] (+)= scale * sum_t in[t, p, :]   (in is [T, P, W]); deterministic (fixed order). */
int pacoh_reduce_tasks(const void* in, void* out, double scale, int accumulate, int T, int P, int W,
                       int dtype, void* stream);

/* ---- 8f: marginal cdf, quantiles and calibration error of the posterior predictive ---------------------------------------
 * The predictive is a mixture of P Gaussians over m test points held in normalised space, mu[P,m], var[P,m] (P = particles /
 * posterior samples; P = 1: the single MAP Gaussian); y = y_mean + y_std * y_n is applied on the fly
 * (AffineTransformedDistribution, meta_learn/models.py:15-43).
 *   pacoh_mixture_cdf:  cdf[t,j] = mean_p Phi((value[t,j] - (y_mean + y_std mu[t,p,j])) / (y_std sqrt(var[t,p,j]))) for a batch of T
 *                       test tasks (mu, var [T,P,m] = the [T*P, m] output of pacoh_gp_predict; value, cdf [T,m]; T <= 65535)
 *                       EqualWeightedMixtureDist.cdf (models.py:124-131); used by _calib_error (abstract.py:260-272).
 *   pacoh_mixture_icdf: out[j] = the quantile[j]-quantile of that marginal.  closed_form = 0: the interval-halving search of
 *                       find_root_by_bounding (meta_learn/util.py:9-42) as called by EqualWeightedMixtureDist.icdf
 *                       (models.py:136-140: lo = -1e8, hi = 1e8, eps = 1e-6, max_iter = 10000) with the reference's stopping rule
 *                       (all elements halved together until the largest half-width <= eps; more than max_iter rounds -> NaN for
 *                       every element), in ONE launch; m <= PACOH_MAX_QUANTILES.  closed_form = 1 (P must be 1): the Gaussian
 *                       quantile y_mean + y_std (mu + sigma sqrt(2) erfinv(2q - 1)) of TransformedDistribution.icdf.
 *   pacoh_calib_error:  out[t] = sqrt(mean_k (#{j: cdf[t,j] <= c_k} / m - c_k)^2), c = linspace(0.05, 0.95, 20) (abstract.py:260-272);
 *                       one launch for the T test tasks of eval_datasets (abstract.py:165-181).
 * pacoh_mixture_icdf takes ONE task (mu, var [P,m]). */
#define PACOH_MAX_QUANTILES 2048
int pacoh_mixture_cdf(const void* mu, const void* var, const void* value, void* cdf, double y_mean, double y_std, int T, int P,
                      int m, int dtype, void* stream);
int pacoh_mixture_icdf(const void* mu, const void* var, const void* quantile, void* out, double y_mean, double y_std, double lo,
                       double hi, double eps, int max_iter, int closed_form, int P, int m, int dtype, void* stream);
int pacoh_calib_error(const void* cdf, void* out, int T, int m, int dtype, void* stream);

/* ---- K whole PACOH-MAP iterations per launch (round 5) ------------------------------------------
 * The reference's own regime -- a handful of small tasks per iteration (demo.py:14-26: 5 tasks x 5 points) -- is pure launch
 * latency as a sequence of launches.  pacoh_map_persist runs K iterations of the loop body of GPRegressionMetaLearned.meta_fit
 * (GPR_meta_mll.py:104-117: sample the task batch, per task LearnedGPRegressionModel.forward + ExactMarginalLogLikelihood
 * (models.py:505-519), loss = -sum_t mll_t, backward, AdamW step) in ONE launch of ONE workgroup that keeps parameters, Adam
 * moments and activations in LDS (csrc/map_persist.hip).  fp32, one parameter row (P = 1), world size 1.
 *   theta / exp_avg / exp_avg_sq [D]: read at the start, trained entries written back at the end;
 *   x [T, n, d], y [T, n], n_valid [T] | NULL: the resident task table; idx_rows [K, tb] (int64) the task draws and sc_rows [K, n_sc]
 *   the step scalars of the K iterations (rows of engine.StepFeed: PACOH_SC_ADAM block used);
 *   mean_mode PACOH_MEAN_ZERO / _VECTOR (NN at theta[off_mean ..), hidden widths mean_hidden[n_mean_hidden]) / _CONST (theta[off_mean]);
 *   kernel_nn != 0: features = NN at theta[off_kernel ..) with f outputs, else the raw inputs (f == d); ARD-RBF kernel with
 *   lengthscales softplus(theta[off_ls .. +f)), outputscale softplus(theta[off_os]) (off_os < 0: none), noise
 *   softplus(theta[off_noise]) + noise_floor; networks as models.py:319-323 (per layer bias before weight, tanh);
 *   seg_lo / seg_hi [n_seg <= 4]: trained column ranges (learning_mode); beta1 / beta2: Adam;
 *   *loss_last := loss of the last iteration, *loss_cum += sum of the K losses, *fail_flag |= 1 if a Cholesky failed even with the
 *   jitter ladder in any iteration (the parameters then hold NaN, as after the launch sequence).
 * Limits (PACOH_ELIMIT otherwise; pacoh_map_persist_supported answers without launching): n <= 32, d <= 4, f <= 4, tb <= 16,
 * tb n (d + 1) <= 1024, hidden widths <= 32, <= 4 hidden layers, K <= 1024, the LDS plan <= 160 KB.  Results equal the launch
 * sequence's to rounding (other summation order), not bit for bit. */
int pacoh_map_persist_supported(int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden, int kernel_nn,
                                const int32_t* kernel_hidden, int n_kernel_hidden, int f, int dtype);
int pacoh_map_persist(void* theta, void* exp_avg, void* exp_avg_sq, int D, const void* x, const void* y, const int32_t* n_valid,
                      int n, int d, const int64_t* idx_rows, int tb, const void* sc_rows, int n_sc, int K,
                      int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                      int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                      int off_ls, int off_os, int off_noise, double noise_floor,
                      const int32_t* seg_lo, const int32_t* seg_hi, int n_seg, double beta1, double beta2,
                      void* loss_last, void* loss_cum, int32_t* fail_flag, int dtype, void* stream);

/* The same loop body for task batches too LARGE for one workgroup (BASELINE config #2: 256 tasks x 32 points per iteration), as TWO
 * launches per iteration instead of four (round 5, csrc/map_task.hip): pacoh_map_task_step runs
 *   (1) forward of the networks, GP LML and its gradient, backward of the networks for every task -- one workgroup per task (several
 *       tasks per workgroup when they fit one 16-point tile), the task's activations staying in LDS between the stages --, writing one
 *       gradient slab per workgroup and network;
 *   (2) the slab reduction with the step's tail: sums over tasks into d_theta[1, D], hyper-parameter reduction with the softplus chain
 *       rule, *lik = lik_scale * sum_t lml_t, the failure flag, and -- opt / opt->next as for pacoh_mlp2_bwd_hyper -- the AdamW step on
 *       every entry, the next iteration's scalars, task gather and hyper-parameter transforms.
 * batch_x [tb, n, d] / batch_y [tb, n] / batch_n_valid [tb] | NULL: the iteration's gathered tasks (pacoh_step_begin / the previous
 * call's opt->next); ls [f] / os [1] | NULL / noise [1]: the transformed hyper-parameters of the ONE parameter row theta[1, D]; the
 * networks as for pacoh_map_persist (D = the length of the parameter row).  workspace: pacoh_map_task_workspace_bytes() bytes (0: shape outside the plan -> PACOH_ELIMIT
 * from the call -- or, any_size == 0, more workgroups than are resident at once: from the second round of workgroups on the four-launch
 * sequence is faster, 2048 tasks x 32 points 0.088 vs 0.053 ms per iteration, profiles/r06_task_fused_crossover.txt).  Limits: fp32, n <= 32, d <= 4, f <= 4, at least one network, hidden widths <= 32, <= 4 hidden layers.  The step
 * counter of opt->next is advanced by launch (1).
 * WIDE networks (round 6, csrc/map_wide.hip): hidden widths that are multiples of 16 up to 128 -- the reference's PACOH-MAP launcher
 * runs 2 tasks x 5 points per iteration through two 4 x 128 networks (experiments/meta_GPR_mll_base_exp.py:29-47) -- take the same three
 * entry points when the whole batch is at most 32 points of tasks of at most 16 (tb n <= 32, n <= 16): launch (1) is then one workgroup per NETWORK that streams the
 * weights from theta through the matrix cores layer by layer (no LDS image), the two meeting once in front of the GP; launch (2) is
 * unchanged.  pacoh_map_task_setup zeroes the workgroups' arrival counts in the workspace there (no parameter image exists).
 * pacoh_map_task_setup: the workspace also holds the networks' parameters in the padded layout launch (1) keeps them in (its prologue
 * copies them instead of decoding theta) -- written by this call from theta, kept current by launch (2)'s AdamW step.  Call it before the
 * first pacoh_map_task_step on a workspace and whenever theta was changed by anything else. */
int pacoh_map_task_setup(const void* theta, int D, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                         int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                         void* workspace, size_t workspace_bytes, int dtype, void* stream);
size_t pacoh_map_task_workspace_bytes(int D, int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden, int kernel_nn,
                                      const int32_t* kernel_hidden, int n_kernel_hidden, int f, int any_size, int dtype);
int pacoh_map_task_step(const void* theta, long theta_stride, const void* batch_x, const void* batch_y, const int32_t* batch_n_valid,
                        int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                        int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                        const void* ls, const void* os, const void* noise, int off_ls, int off_os, int off_noise,
                        void* d_theta, long d_theta_stride, void* lik, double lik_scale, int32_t* fail_flag,
                        void* workspace, size_t workspace_bytes, const pacoh_adam_inline* opt, int dtype, void* stream);

/* ---- the likelihood half of a PACOH-SVGD / PACOH-VI step for under-filled grids (round 6, csrc/map_task.hip) ----------------------
 * The reference's own SVGD / VI launchers run 2 tasks x 10 particles (posterior samples) of 20 points per step
 * (experiments/meta_GPR_SVGD_base_exp.py:28-49, meta_GPR_vi_base_exp.py:29-52): 20 GP problems, for which the general launch sequence
 * (networks forward -> GP -> networks backward -> slab reduction) is four kernel latencies.  pacoh_svgd_task_step runs the sum
 * random_gp.py:204-222 takes -- for every task t of the batch and every parameter row p: VectorizedGP.forward (random_gp.py:54-89), its
 * MLL / n and the gradient w.r.t. theta[p] -- as TWO launches:
 *   (1) pacoh_map_task_step's kernel with one workgroup per (task group, parameter row): the row's networks decoded into LDS, forward
 *       chains -> GP LML + gradient -> delta chains -> weight-gradient tiles, one gradient slab per workgroup and network; with
 *       svgd_X != NULL extra workgroups form the particles' distance matrix, its snapshot and advance *counter exactly as
 *       pacoh_mlp2_fwd_svgd does (svgd.py:45-51's operand);
 *   (2) the slab reduction with the step's tail: d_theta[p, :] = sum_t d mll[t, p] / d theta[p] (network blocks and, through the softplus
 *       chain rule, the hyper-parameter columns), lik[p] = lik_scale * sum_t mll[t, p], *fail_flag |= 1 if a Cholesky failed even with
 *       the jitter ladder, and -- want_bandwidth -- the median-heuristic bandwidth into the workspace's bandwidth slot.
 * theta [P, D] (row stride theta_stride): SVGD's particles or VI's posterior samples; batch_x [tb, n, d] / batch_y [tb, n] /
 * batch_n_valid [tb] | NULL: the step's gathered tasks; ls [P, f] / os [P] | NULL / noise [P]: the rows' transformed hyper-parameters;
 * the networks, kernel and limits as for pacoh_map_task_step (fp32, RBF, n <= 32, d <= 4, f <= 4, at least one network, hidden widths
 * <= 32, <= 4 hidden layers); svgd_workspace: pacoh_svgd_update_dev_workspace_bytes() bytes, as for pacoh_mlp2_fwd_svgd.
 * workspace: pacoh_svgd_task_workspace_bytes() bytes; 0 = the caller takes the general sequence: the shape is outside the plan, or --
 * any_size == 0 -- the tb x P workgroups would not all be resident at once (CUs x the occupancy of the kernel with its LDS plan), beyond
 * which the throughput kernels win (profiles/r06_task_fused_crossover.txt); any_size != 0: wherever the plan allows.  It also holds
 * the map from the networks' padded LDS layout to the columns of a parameter row, written once by pacoh_svgd_task_setup (a function
 * of the layout only -- no call is needed when the rows change).  Results equal the general sequence's to rounding. */
size_t pacoh_svgd_task_workspace_bytes(int D, int P, int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden,
                                       int kernel_nn, const int32_t* kernel_hidden, int n_kernel_hidden, int f, int any_size, int dtype);
int pacoh_svgd_task_setup(int D, int P, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                          int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                          void* workspace, size_t workspace_bytes, int dtype, void* stream);
int pacoh_svgd_task_step(const void* theta, long theta_stride, int P, const void* batch_x, const void* batch_y,
                         const int32_t* batch_n_valid, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden,
                         int n_mean_hidden, int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                         const void* ls, const void* os, const void* noise, int off_ls, int off_os, int off_noise,
                         void* d_theta, long d_theta_stride, void* lik, double lik_scale, int32_t* fail_flag,
                         void* workspace, size_t workspace_bytes, const void* svgd_X, void* svgd_workspace, int svgd_D, int64_t* counter,
                         int want_bandwidth, int dtype, void* stream);

/* ---- 8e: the step's one exchange ----------------------------------------------------------------------------------
 * buf[0..count) := sum over ranks of buf (in place), enqueued on the caller's stream: RCCL ncclAllReduce(ncclSum) over xGMI.
 * Sums the per-rank partial  sum_t mll[t,:]  and partial score [P,D] of the task-sharded objective
 * (random_gp.py:214-219; MAP: GPR_meta_mll.py:109-113) -- the reference is single-process and has no equivalent.
 * The communicator is an opaque handle owned by the caller: rank 0 calls pacoh_comm_unique_id(), ships the
 * PACOH_COMM_ID_BYTES bytes to the other ranks out of band (the host package uses its torch.distributed store), then every
 * rank calls pacoh_comm_init() with the HIP device it computes on current.  RCCL itself is bound at the first call
 * (dlopen of librccl.so.1, reusing the copy already mapped into the process if there is one), so the library loads on
 * hosts without RCCL; the comm functions then return PACOH_ENOCOMM.  >0 return = the ncclResult_t RCCL reported. */
#define PACOH_COMM_ID_BYTES 128
#define PACOH_ENOCOMM (-5)     /* librccl could not be loaded / symbol missing                   */
int pacoh_comm_unique_id(void* id_out);
int pacoh_comm_init(const void* id, int rank, int world, void** comm_out);
int pacoh_allreduce_sum(void* buf, long count, int dtype, void* comm, void* stream);
int pacoh_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* PACOH_GP_H */
