/*
 * cglb_hip.h — C ABI of the MI355X-native CGLB quadratic-term solver (libcglb_hip.so).
 *
 * The reference (awav/CGLB, pure Python) has no FFI; these entry points are what a binding for its
 * hot path would call.  Each one cites the reference seam it stands in for (paths relative to the
 * reference repo root).  See INTEGRATION.md for the ctypes stub that plugs them into cglb.backend.
 *
 * Conventions
 *   - All array arguments are plain pointers; matrices are row-major; "dev" pointers are device
 *     (HBM) addresses (e.g. torch.Tensor.data_ptr()), "host" pointers are ordinary host memory,
 *     "any" may be either (copied with hipMemcpyDefault).  The caller owns every array it passes;
 *     the library keeps no pointer past the call (set_data/set_hypers copy).
 *   - Element type of vectors/matrices follows the ctx dtype (CGLB_F64 -> double, CGLB_F32 -> float);
 *     hyper-parameters and returned scalars are always double.
 *   - Every function returns 0 (CGLB_OK) or an error code; cglb_last_error() gives the text.
 *   - One ctx = one GPU = one row shard [row_begin,row_end) of the N training rows.  With a single
 *     shard (row_begin=0,row_end=N) the fused calls (cglb_pcg_solve, cglb_objective_and_grad) do the
 *     whole job.  With several shards (one process per GPU) either the library runs the loops itself and
 *     issues the collectives on its stream (cglb_comm_init_* + cglb_dist_*: RCCL, or callbacks), or the
 *     host drives the cglb_shard_* / cglb_vec_* phases and performs the collectives between them
 *     (cglb_amd/distributed.py: the same scheme step by step, testable over gloo with CPU local ops).
 *   - Calls on one ctx are serialised by the caller (the reference is single-threaded Python with a
 *     blocking sync per CG iteration, conjugate_gradient.py:80-81).  Work is enqueued on the HIP stream
 *     given at creation; functions that return host scalars synchronise that stream, the others do not.
 */
#ifndef CGLB_HIP_H
#define CGLB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cglb_ctx cglb_ctx;

enum { CGLB_OK = 0, CGLB_ERR_BAD_ARG = 1, CGLB_ERR_NOT_PD = 2, CGLB_ERR_HIP = 3, CGLB_ERR_BLAS = 4, CGLB_ERR_STATE = 5, CGLB_ERR_COMM = 6 };
enum { CGLB_RBF = 0, CGLB_MATERN32 = 1 };      /* config.py:72-81 SquaredExponentialConfig / Matern32Config */
enum { CGLB_F64 = 0, CGLB_F32 = 1 };           /* pytorch/interface.py:94-104 set_default_float */

/* Number of entries of the packed gradient: [dl_1..dl_D, d variance, d noise, d mean, dZ (M*D row-major)] */
#define CGLB_GRAD_LEN(D, M) ((D) + 3 + (M) * (D))

int cglb_version(void);

/* Model + data container.  Stands in for CGLB(data, likelihood, kernel) built by
 * pytorch/interface.py:315-323 (create_model for CGLBConfig) and models.py:54-68 (v_vec buffer).
 * stream: a hipStream_t (0 = default stream).  device: HIP device ordinal. */
int cglb_ctx_create(cglb_ctx** out, int64_t n_total, int64_t row_begin, int64_t row_end, int d, int m,
                    int dtype, int kernel_kind, int device, void* stream);
int cglb_ctx_destroy(cglb_ctx* ctx);
const char* cglb_last_error(const cglb_ctx* ctx); /* never NULL; ctx may be NULL (creation errors) */

/* Training inputs (model.train_inputs / train_targets, pytorch/interface.py:318-322).
 * X: any [n_total, d] (all rows: the column side of K_ff is replicated), y: any [n_total]. */
int cglb_set_data(cglb_ctx* ctx, const void* X, const void* y);

/* Constrained hyper-parameters: kernel lengthscales/outputscale, likelihood noise, ConstantMean,
 * inducing points, Cholesky jitter (pytorch/interface.py:150-178 model_parameters;
 * backend.py:77-79 jitter).  lengthscales: host double[d]; Z: any [m, d] of ctx dtype. */
int cglb_set_hypers(cglb_ctx* ctx, const double* lengthscales, double variance, double noise, double mean,
                    const void* Z, double jitter);

/* ---- common terms: LowerBoundCG.logdet_and_quad_common_terms, models.py:176-213 ---------------- */
/* single shard: L = chol(K_uu + jitter I), A = L^-1 K_uf / sigma, B = A A^T + I, LB = chol(B), tr(AA^T). */
int cglb_setup(cglb_ctx* ctx);
/* sharded: _local computes L, the column shard A[:, rows] and the partial A_loc A_loc^T into the buffer
 * returned by cglb_aat_buffer (dev [m*m]); the host all-reduces that buffer; _finish does the rest. */
int cglb_shard_setup_local(cglb_ctx* ctx);
void* cglb_aat_buffer(cglb_ctx* ctx);
int cglb_shard_setup_finish(cglb_ctx* ctx);
/* logdet_estimator (models.py:215-244) value for the current common terms. */
int cglb_logdet(cglb_ctx* ctx, double* logdet);

/* ---- operator seam: `A @ p` with A = kernel(x).add_diag(sigma^2), models.py:251-252,
 *      conjugate_gradient.py:57,66,72 -------------------------------------------------------------- */
/* out[row_begin:row_end] = K_ff[rows, :] p + noise * p[rows].  p_full: dev [n_total]; out: dev [n_local]. */
int cglb_matvec(cglb_ctx* ctx, const void* p_full, void* out_local);
/* same, and pdot[0] = sum over local rows of p_i * out_i  ((p * Ap).sum(), conjugate_gradient.py:67). pdot: dev double[1]. */
int cglb_matvec_dot(cglb_ctx* ctx, const void* p_full, void* out_local, void* pdot);
/* right-hand side of the solve: out_local = y[rows] - mean  (err, models.py:253-254). */
int cglb_shard_rhs(cglb_ctx* ctx, void* out_local);
/* Rectangular kernel mat-vec  out[i] = sum_j k(xnew_i, x_j) v_j  (ksf @ v, models.py:320,334).
 * xnew: any [n_new, d]; v_full: dev [n_total]; out: dev [n_new]. Sums over ALL n_total columns. */
int cglb_cross_matvec(cglb_ctx* ctx, const void* xnew, int64_t n_new, const void* v_full, void* out);

/* ---- preconditioner seam: NystromPreconditioner.__call__, conjugate_gradient.py:95-113 ------------ */
/* single shard: z = (Q_ff + noise I)^-1 r, rz = r^T z.  r,z: dev [n]; rz: host. */
int cglb_precond_apply(cglb_ctx* ctx, const void* r, void* z, double* rz);
/* sharded: u_partial[m] = A_loc r_loc (host all-reduces u), then z_loc and the partial sum of rz. */
int cglb_shard_precond_u(cglb_ctx* ctx, const void* r_local, void* u_partial /* dev [m] */);
int cglb_shard_precond_z(cglb_ctx* ctx, const void* r_local, const void* u /* dev [m], reduced */,
                         void* z_local, void* rz_partial /* dev [1], or NULL when the caller forms r^T z itself */);

/* segmented form for the cyclic multi-GPU path: z_slot is this rank's slice of the all-gather buffer, dev [per + 1]: z for the local
 * rows in [0, n_local) and, in element `per`, this rank's partial of r^T z over its own rows (in the vector element type). */
int cglb_shard_precond_z_seg(cglb_ctx* ctx, const void* r_local, const void* u /* dev [m], reduced */, void* z_slot, int64_t per);

/* ---- vector primitives of the PCG loop (conjugate_gradient.py:67-75), for the sharded host loop ---- */
/* dot: out[0] = sum_i a_i b_i over local rows (dev scalar, deterministic order). */
int cglb_shard_dot(cglb_ctx* ctx, const void* a_local, const void* b_local, void* out /* dev [1] */);
/* v += gamma p ; r -= gamma Ap  with gamma = rz / pAp read from device scalars (:67-68,:72). */
int cglb_shard_update_v_r(cglb_ctx* ctx, void* v_local, void* r_local, const void* p_local, const void* Ap_local,
                          const void* rz /* dev [1] */, const void* pAp /* dev [1] */, int update_r);
/* r = b - Kv (restart / initial residual, :58,:72). */
int cglb_shard_residual(cglb_ctx* ctx, void* r_local, const void* b_local, const void* Kv_local);
/* p = z + p * (new_rz / rz)  or  p = z on restart (:75). */
int cglb_shard_update_p(cglb_ctx* ctx, void* p_local, const void* z_local, const void* new_rz /* dev [1] */,
                        const void* rz /* dev [1] */, int restart);

/* ---- solver seam: ConjugateGradient.__call__, conjugate_gradient.py:41-86 (single shard) ---------- */
/* Solves (K_ff + noise I) v = b warm-started at v_inout (overwritten with the solution; the Python shim
 * clones first so the caller's tensor is not mutated, :55).  Stops when 1/2 r^T P r <= max_error or
 * steps == max_cg_iter, predicate tested before every iteration (:65); exact-residual restart when
 * i % restart_cg_iter == restart_cg_iter-1 (:70-75).  steps / half_rz mirror ConjugateGradientStats (:25-28). */
int cglb_pcg_solve(cglb_ctx* ctx, const void* b /* dev [n] */, void* v_inout /* dev [n] */, double max_error,
                   int max_cg_iter, int restart_cg_iter, int* steps, double* half_rz);

/* ---- objective seam: LowerBoundCG.forward + quad_estimator, models.py:151-174, :246-286, and its
 *      gradient as torch.autograd.grad yields it with v detached (pytorch/optimizer.py:95-98) -------- */
/* Runs cglb_setup, (optionally) the PCG from v_inout with b = y - mean, the bound assembly and the
 * gradient.  out4 = {bound, lower, upper, logdet} (host); grad: host double[CGLB_GRAD_LEN(d,m)] or NULL
 * (value only).  v_inout: dev [n] — the persistent warm-start vector (models.py:59-72, :274). */
int cglb_objective_and_grad(cglb_ctx* ctx, void* v_inout, int run_cg, double max_error, int max_cg_iter,
                            int restart_cg_iter, double* out4, double* grad, int* steps, double* half_rz);

/* TF twin only (tensorflow/models.py:36-51,161-164: `joint_optimization` makes v a trainable parameter): gradient of the bound
 * wrt v at the v of the evaluation just made, d bound / d v = K_sigma (P r) - r with r = e - K_sigma v.  Must directly follow
 * cglb_objective_and_grad (single shard); one more K_ff mat-vec.  gv_out: dev [n]. */
int cglb_objective_grad_v(cglb_ctx* ctx, void* gv_out);

/* sharded pieces of the same evaluation (host all-reduces between phases):
 *   phase1: Kv = (K+sI) v (v_full gathered), r = e - Kv, u_partial = A_loc r              -> all-reduce u[m]
 *   phase2: w = P r (local), partial scalars sc[8] and aw_partial = A_loc w               -> all-reduce sc[8], aw[m]
 *   phase3: partial packed gradient (dev double[GRAD_LEN]); rank-replicated terms are added by the
 *           shard with row_begin == 0 only                                               -> all-reduce grad
 *   finish: bound/lower/upper from reduced sc (host). */
int cglb_shard_obj_phase1(cglb_ctx* ctx, const void* v_full, void* u_partial);
int cglb_shard_obj_phase2(cglb_ctx* ctx, const void* v_full, const void* u /* dev [m], reduced */, void* sc_partial /* dev double[8] */,
                          void* aw_partial /* dev [m] */);
int cglb_shard_obj_phase3(cglb_ctx* ctx, const void* v_full, const void* sc /* dev double[8], reduced */,
                          const void* aw /* dev [m], reduced */, void* grad_partial /* dev double[GRAD_LEN] */);
int cglb_shard_obj_finish(cglb_ctx* ctx, const void* sc /* dev double[8], reduced */, double* out4);

/* ---- cyclic-symmetric multi-GPU path ----------------------------------------------------------------------------
 * K_ff is symmetric, and the symmetric pair kernel (each kappa_ij used for out_i and out_j) halves the work of a mat-vec.
 * To keep that factor under sharding the GLOBAL upper triangle is dealt to the ranks by cyclic 256-row blocks:
 * cglb_matvec_cyclic produces this rank's full-length PARTIAL of (K_ff + noise I) p (the noise term is added by rank 0 only);
 * the host all-reduces it and every
 * rank holds the full vectors p, Ap, v, r (their updates are O(N) and done redundantly with the cglb_vec_* primitives).
 * Only the Nystrom panel stays column-sharded (rows [row_begin,row_end)): u = A r is all-reduced (M), z is all-gathered. */
int cglb_set_parallel(cglb_ctx* ctx, int world, int rank);
int cglb_matvec_cyclic(cglb_ctx* ctx, const void* p_full, void* out_full_partial /* dev [n_total] */);
int cglb_rhs_full(cglb_ctx* ctx, void* out_full /* dev [n_total] */); /* y - mean for all rows */
/* vector primitives with an explicit length (same kernels as the cglb_shard_* ones) */
int cglb_vec_dot(cglb_ctx* ctx, int64_t n, const void* a, const void* b, void* out /* dev double[1] */);
int cglb_vec_update_v_r(cglb_ctx* ctx, int64_t n, void* v, void* r, const void* p, const void* Ap, const void* rz, const void* pAp, int update_r);
int cglb_vec_residual(cglb_ctx* ctx, int64_t n, void* r, const void* b, const void* Kv);
int cglb_vec_update_p(cglb_ctx* ctx, int64_t n, void* p, const void* z, const void* new_rz, const void* rz, int restart);
/* p = z + p * (new_rz / rz) (or p = z) with z all-gathered in `world` slices of per + 1 elements (cglb_shard_precond_z_seg) and
 * new_rz = the sum, in rank order, of the slices' extra elements: the stop-test scalar is then identical on every rank by
 * construction.  new_rz: dev double[1], written; rz: dev double[1], the previous value (conjugate_gradient.py:75-76).
 * The kernel also prepares the operand of the NEXT cglb_matvec_cyclic(p): p must not be modified in between (the PCG loop does not). */
int cglb_vec_update_p_seg(cglb_ctx* ctx, int64_t n, int64_t per, int world, void* p, const void* zseg, void* new_rz, const void* rz, int restart);
int cglb_vec_axpy(cglb_ctx* ctx, int64_t n, double alpha, const void* x, void* y); /* y += alpha x */
/* objective phase 1 with (K_ff + noise I) v already computed for the local rows; phase 2 unchanged; the local slice of
 * w = P r for the all-gather of u = w + v/2; phase 3 with the cyclic share of the N^2 gradient form (u_full gathered). */
int cglb_shard_obj_phase1_kv(cglb_ctx* ctx, const void* Kv_local, void* u_partial);
int cglb_shard_obj_w(cglb_ctx* ctx, void* w_local_out);
int cglb_shard_obj_phase3_cyclic(cglb_ctx* ctx, const void* v_full, const void* u_full, const void* sc, const void* aw, void* grad_partial);

/* ---- prediction seam: PredictCG.forward, models.py:307-354 (needs common terms + a solved v) ------ */
/* f_mean, f_var: dev [n_new].  v_full: dev [n] solution at tolerance 1e-3 (models.py:291). single shard. */
int cglb_predict(cglb_ctx* ctx, const void* v_full, const void* xnew, int64_t n_new, void* f_mean, void* f_var);
/* sharded pieces of the same computation (host or library all-reduces between them):
 *   _u:    res = (y - mean) - Kv over the local rows (models.py:335), u_partial = A_loc res (:340)             -> all-reduce u[m]
 *   _rows: c = LB^-1 u / sigma (:343) and, for the new points handed to THIS rank (any subset of the rows of xnew - the driver deals
 *          contiguous slices to the ranks and all-gathers the results), cg_mean = k(xnew, X) v over ALL n_total columns (:334),
 *          tmp1, tmp2 (:344-345), f_mean, f_var (:347-351). */
int cglb_shard_predict_u(cglb_ctx* ctx, const void* Kv_local /* dev [n_local]: (K_ff + noise I) v on the local rows */, void* u_partial /* dev [m] */);
int cglb_shard_predict_rows(cglb_ctx* ctx, const void* v_full, const void* u /* dev [m], reduced */, const void* xnew, int64_t n_new, void* f_mean,
                            void* f_var);

/* ---- N ranks inside the library ------------------------------------------------------------------------------------
 * The cyclic-symmetric scheme above with its collectives issued by the library itself on the context stream, so that a PCG
 * iteration at world N is enqueued like the single-GPU one: no host work between the kernels of an iteration except the one
 * read-back of the stop statistic (conjugate_gradient.py:65,80-81), which the look-ahead hides.  One process per GPU; every
 * rank makes the same calls with the same arguments.  The context must own the rows of the contiguous partition
 * [rank * per, min((rank + 1) * per, n_total)), per = ceil(n_total / world), and use the stored-panel preconditioner.
 *   cglb_comm_get_unique_id: rank 0 obtains the RCCL id (CGLB_COMM_ID_BYTES bytes) and hands it to the other ranks through the
 *     launcher's store (torch.distributed's TCPStore in cglb_amd/dist_context.py); cglb_comm_init_rccl is collective.
 *   cglb_comm_init_callbacks: the same loops over collectives the HOST provides (buffers are device pointers of the context's
 *     device; in place; the callback must order itself after the work already enqueued on `stream` and leave the result visible
 *     to work enqueued on `stream` afterwards; return 0 on success).  Used to run the library's N-rank loops over gloo in the
 *     tests (several ranks sharing one GPU, which RCCL refuses) and available for any other fabric. */
#define CGLB_COMM_ID_BYTES 128
typedef int (*cglb_allreduce_fn)(void* user, void* buf, int64_t count, int dtype /* CGLB_F64 | CGLB_F32 */, void* stream);
/* in place: rank g's contribution sits at element g * count_per_rank of buf (world * count_per_rank elements) */
typedef int (*cglb_allgather_fn)(void* user, void* buf, int64_t count_per_rank, int dtype, void* stream);
int cglb_comm_get_unique_id(void* id_out /* host, CGLB_COMM_ID_BYTES */);
int cglb_comm_init_rccl(cglb_ctx* ctx, const void* unique_id /* host, CGLB_COMM_ID_BYTES */, int world, int rank);
int cglb_comm_init_callbacks(cglb_ctx* ctx, int world, int rank, cglb_allreduce_fn allreduce, cglb_allgather_fn allgather, void* user);
int cglb_comm_destroy(cglb_ctx* ctx);
/* common terms with B = I + sum_g A_g A_g^T (one all-reduce of m x m) */
int cglb_dist_setup(cglb_ctx* ctx);
/* out = (K_ff + noise I) x, full length on every rank (this rank's cyclic share + all-reduce). x_full, out_full: dev [n_total] */
int cglb_dist_matvec(cglb_ctx* ctx, const void* x_full, void* out_full);
/* z = (Q_ff + noise I)^-1 r on full replicated vectors; rz: host, may be NULL */
int cglb_dist_precond_apply(cglb_ctx* ctx, const void* r_full, void* z_full, double* rz);
/* cglb_pcg_solve / cglb_objective_and_grad / cglb_predict on N ranks: vectors are full length and replicated (b_full, v_full_inout,
 * f_mean, f_var identical on every rank afterwards); host outputs are identical on every rank.  The stop test is taken on
 * all-gathered per-rank partials of r^T P r, so every rank leaves the loop at the same step by construction. */
int cglb_dist_pcg_solve(cglb_ctx* ctx, const void* b_full, void* v_full_inout, double max_error, int max_cg_iter, int restart_cg_iter,
                        int* steps, double* half_rz);
int cglb_dist_objective_and_grad(cglb_ctx* ctx, void* v_full_inout, int run_cg, double max_error, int max_cg_iter, int restart_cg_iter,
                                 double* out4, double* grad, int* steps, double* half_rz);
int cglb_dist_predict(cglb_ctx* ctx, const void* v_full, const void* xnew, int64_t n_new, void* f_mean, void* f_var);

/* ---- inducing-point initialisation: InducingVariableConfig.init, config.py:55-65 ------------------------
 * The reference calls robustgp.ConditionalVariance(sample=False) (third-party): greedy maximisation of the conditional
 * variance under the INITIAL kernel (pivoted Cholesky of K_ff, lowest index on ties).  Needs set_data only; works on all n rows
 * whatever the row range of the context, so every rank of a sharded run obtains the same answer.  `lengthscales` host [d];
 * indices_out: host [min(m, n)] row indices in selection order; Z_out: dev [m, d] = X[indices] or NULL; trace_out: host,
 * sum of the remaining conditional variances, tr(K_ff - Q_ff) (+ n * jitter), or NULL.  Invalidates set_hypers (the scaled
 * operand buffers are used as scratch).  Blocking. */
int cglb_select_inducing(cglb_ctx* ctx, const double* lengthscales, double variance, double jitter, int64_t* indices_out, void* Z_out,
                         double* trace_out);

/* ---- introspection / measurement ------------------------------------------------------------------ */
/* Copy common-term matrices out (tests): which = 0:A [m,n_local] 1:L [m,m] lower 2:LB [m,m] lower; dst: any. */
int cglb_get_matrix(cglb_ctx* ctx, int which, void* dst);
/* Average duration (ms, HIP events on the ctx stream) of `reps` back-to-back launches of one kernel family:
 * which = 0: K_ff mat-vec (pair kernel + slab combine), 1: preconditioner apply, 2: gradient bilinear pass,
 * 3: the pair kernel of the mat-vec alone (the dominant kernel), 4: the same for this rank's cyclic share (cglb_set_parallel),
 * 5: K_uu + jitter I and its Cholesky factorisation (invalidates the common terms: call cglb_setup again afterwards). */
int cglb_time_kernel(cglb_ctx* ctx, int which, int reps, double* ms_avg);
/* In-situ measurement of the dominant kernel: after cglb_set_option(ctx, "k1_profile", 1) every launch of the symmetric pair
 * kernel (in mat-vecs, solves and evaluations alike) is bracketed by HIP events on the context stream; "k1_ms_total" and
 * "k1_launches" return the accumulated device time and launch count since then (the call synchronises with the pending launches).
 * "k1_pairs_per_launch": kernel pairs one launch of that kernel evaluates with the current geometry (~N(N+256)/2 on one GPU: the
 * symmetric form visits each unordered pair once) - the unit count of the roofline; "kpart_bytes": size of the partial-sum slabs;
 * after cglb_set_option(ctx, "eval_profile", 1): "eval_setup_ms" | "eval_pcg_ms" | "eval_final_ms" | "eval_grad_ms" = device time (HIP events on
 * the context stream) accumulated over the phases of every cglb_objective_and_grad since - common terms | PCG | final mat-vec, preconditioner
 * and bound scalars | gradient - and "eval_count" = the number of evaluations;
 * "comm_allreduce_calls" | "comm_allgather_calls": collectives issued by the library since cglb_comm_init_*;
 * "L_diag_ratio": max/min of diag(chol(K_uu + jitter I)) of the last cglb_setup (a cheap proxy of cond(L)). */
int cglb_get_stat(cglb_ctx* ctx, const char* name, double* value);
/* Tunables: name = "kff_variant" | "kff_jsplit" | "kff_rows" | "sym_chunk" | "precond_mode" | "chol_mode" | "pcg_lookahead" | "sym_order" | "aat_block" | "grad_gram" | "k1_profile" |
 * "grad_trsm" (gradient algebra against L = chol(K_uu): 0 products with the explicit inverse, 1 backward-stable triangular solves, 2 = default:
 *  the products followed by one step of iterative refinement against L - the accuracy of the solves for about two thirds of their time) |
 * "precision" (1, default: kernel values to <= 1e-13 relative - degree-3 table polynomial, one-step square root; 0: ~3e-16; 2: opt-in, degree-2
 *  polynomial, kernel values to ~1e-10 - inside the 1e-6 the bound needs, one instruction per pair cheaper) ...;
 * "pcg_lookahead" (0: the host waits for the stop statistic before enqueuing anything; 1, default: the next mat-vec is enqueued first while
 *  1/2 r^T P r of the previous iteration exceeds 32 x max_error; k >= 2: that factor - a mis-speculated mat-vec is wasted work, never a wrong result) |
 * "final_matvec" (cglb_objective_and_grad after a solve: 1 recomputes K v with a mat-vec like models.py:280; 0, the default, takes K v = e - r
 *  from the residual r the PCG recurrence carries - exact at the start of a solve and after every restart step; measured difference at the
 *  headline shape: bound <= 3e-15 relative, gradient <= 1e-10 of its largest entry, one N^2 pass saved per evaluation) |
 * "wide_grad_sym" (tiled K_ff gradient pass of wide inputs: 1, default - tiles on and right of the diagonal only; 0 - every tile) |
 * "wide_reg" (inputs of 33 ... 96 dimensions, fp64: 1, default - the symmetric K_ff mat-vec and the K_ff gradient pass stay register-resident,
 *  column operands handed out by v_fmac_f64 row_newbcast; 0 - both go through the Gram tiles like every other product of a wide input) |
 * "drop_weighted_operand" (any value: forget the pre-weighted copy cglb_vec_update_p_seg made of its p - for callers that modify p before the next mat-vec);
 * returns CGLB_ERR_BAD_ARG if unknown. */
int cglb_set_option(cglb_ctx* ctx, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* CGLB_HIP_H */
