// Internal declarations shared by the translation units of libcglb_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cglb_hip.h"

#define CGLB_MAX_D 1024        // widest input the context accepts (host-side arrays)
#define CGLB_MAX_D_NARROW 32   // widest input of the register-resident pair kernels; beyond it the "wide" path runs (kernels_wide.hip)
#define CGLB_MAX_D_MID 96     // ... except the symmetric K_ff mat-vec (fp64): up to here it still runs register-resident, the Gram chain in 16-wide slices
#define CGLB_WAVE 64

// Collectives of the N-rank path inside the library (cglb_comm_init_*, cglb_dist_*; include/cglb_hip.h): RCCL on the context stream, or
// host-provided callbacks (the same loop over another fabric, e.g. gloo in the tests).
struct cglb_comm_state {
    int kind = 0;  // 1 RCCL, 2 host callbacks
    int world = 1, rank = 0;
    int64_t per = 0;       // rows per rank of the contiguous panel partition, ceil(N / world)
    void* nccl = nullptr;  // ncclComm_t
    cglb_allreduce_fn ar = nullptr;
    cglb_allgather_fn ag = nullptr;
    void* user = nullptr;
    // replicated full-length work vectors (element type T) and gather buffers
    void *p = nullptr, *r = nullptr, *Ap = nullptr, *Kv = nullptr, *b = nullptr;  // [N]
    void* zseg = nullptr;   // [world * (per + 1)]: all-gather target of z with each rank's partial of r^T z
    void* ubuf = nullptr;   // [world * per]: all-gather target of u = w + v/2 (gradient phase)
    void *u = nullptr, *aw = nullptr;  // [M]
    double* sc = nullptr;   // [8]
    double* grad = nullptr; // [GRAD_LEN]
    void* gat = nullptr;    // prediction gather buffer
    size_t gat_cap = 0;
    long long n_allreduce = 0, n_allgather = 0;  // collectives issued since cglb_comm_init (cglb_get_stat)
};

struct cglb_ctx {
    cglb_comm_state* comm = nullptr;
    // geometry
    int64_t N = 0, r0 = 0, r1 = 0, nloc = 0, lda = 0;  // lda: leading dimension of At/Guf (nloc rounded up to 8)
    int D = 0, Dp = 0, M = 0, dtype = CGLB_F64, kind = CGLB_RBF, device = 0;
    int Dh = 0;        // 32 < D <= 96, fp64: padded width (48, 64, 80, 96) of the hot operand set Xh kept for the register-resident mat-vec
    int wide_grad_sym = 1;  // option "wide_grad_sym": 0 evaluates every tile of the tiled K_ff gradient pass (A/B)
    int wide_reg = 1;  // option "wide_reg": 0 sends that mat-vec through the Gram tiles of kernels_wide.hip as well (A/B, fallback)
    size_t esz = 8;
    hipStream_t stream = nullptr;
    rocblas_handle blas = nullptr;
    // state flags
    bool have_data = false, have_hypers = false, have_local = false, have_terms = false;
    bool obj_valid = false;  // w_r / w_z hold r = e - K v and w = P r of the last cglb_objective_and_grad
    // raw data (element type T)
    void *X = nullptr, *y = nullptr, *Z = nullptr;
    double xmean[CGLB_MAX_D] = {0};  // column means of X (centre for the Gram form)
    double xrange[CGLB_MAX_D] = {0}; // max |x - mean| per column (bounds the exponent range of the hot loops)
    double xradius2 = 0.0;           // max_i |x_i - mean|^2 (ball bound of the row norms)
    // hypers (host)
    double ls[CGLB_MAX_D] = {0};
    double var = 1, noise = 1, mean = 0, jitter = 1e-6;
    // scaled operands of the streaming kernels (T): xs = (x-c)/l*kscale padded to Dp, xa = per-row norm term
    void *Xs = nullptr, *xa = nullptr, *Zs = nullptr, *za = nullptr;
    void *Zh = nullptr, *zah = nullptr;  // hot-scaled inducing points (implicit preconditioner)
    void *Linv = nullptr, *LinvT = nullptr;  // L^-1 (column-major lower) and its transpose (implicit preconditioner)
    void* w_q = nullptr;                 // [M] scratch
    void* ppart = nullptr;               // partial slab of launch_pairs_rect
    size_t ppart_cap = 0;
    void* chol_blk = nullptr;            // dense copy of the current diagonal block + reciprocal diagonal (kernels_chol.hip)
    int chol_mode = 1;                   // 1: blocked LDS Cholesky (kernels_chol.hip), 0: rocSOLVER potrf
    int precond_mode = 0;                // 0: stored panel A (reference form), 1: implicit K_uf products
    void *Xhsq = nullptr;  // Xh squared element-wise: second-moment operand of the Gram-form gradient pass (kernels_grad.hip)
    const void* pwh_src = nullptr;       // vector whose weighted copy pwh currently holds (set by update_p, consumed once by the next symmetric mat-vec)
    // wide inputs (D > 32, kernels_wide.hip): element-wise squares of the scaled operands, tile / panel / moment scratch, device copies of
    // the per-dimension centre and scale
    void *Xsq = nullptr, *Zsq = nullptr;
    void *wtile = nullptr, *wpart = nullptr, *wS1 = nullptr, *wVX = nullptr, *wR = nullptr, *wC = nullptr, *wones = nullptr, *wpanel = nullptr;
    size_t wtile_cap = 0, wpart_cap = 0, wS1_cap = 0, wpanel_cap = 0, wvec_cap = 0;
    void* wlong = nullptr;   // slabs of a long-k GEMM cut into chunks (kernels_wide.hip: gemm_long_k)
    size_t wlong_cap = 0;
    double *wcenter = nullptr, *wscale = nullptr, *wsmall = nullptr;
    void* uwh = nullptr;                 // u o wh: second weighted column operand of the Gram-form gradient pass (allocated on first use)
    void *wh = nullptr, *pwh = nullptr;  // RBF column weights 2^(xah_j/T) and the weighted operand p_j * wh_j of the symmetric mat-vec (length N)
    void *Xh = nullptr, *xah = nullptr;  // hot operand set of the pair kernels: exponents in 1/T octave, T = 2^CGLB_TAB_BITS (devmath.h exp2_tab)
    double* exp_tab = nullptr;           // device table 2^(k/T) or 2^((k+1/2)/T) (CGLB_EXP_FLOOR), k < T = 2^CGLB_TAB_BITS, exponent pre-compensated (devmath.h)
    // common terms (column-major M x M unless noted)
    void* At = nullptr;      // A as [M][nloc] row-major == (nloc x M) column-major, ld = nloc
    void* Lc = nullptr;      // chol(Kuu + jitter I), lower, column-major
    void* LBc = nullptr;     // chol(B), lower, column-major
    void* LBinv = nullptr;   // LB^-1 lower, column-major, strict upper zeroed
    void* LBinvT = nullptr;  // its transpose (upper as col-major == rows of LB^-1 contiguous)
    void* AAt = nullptr;     // A A^T (full symmetric)
    void* Mtmp = nullptr;    // M x M scratch
    void* Mtmp2 = nullptr;   // M x M scratch
    void* Mtmp3 = nullptr;   // M x M scratch of the gradient algebra (allocated on first use)
    void* Mtmp4 = nullptr;   // M x M residual scratch of its refinement steps (allocated on first use)
    bool have_Linv = false;  // Linv holds L^-1 of the current K_uu factor
    bool Linv_unchecked = false;  // the trtri status of Linv (info_dev[2]) has not been read back yet
    int grad_trsm = 2;       // gradient algebra against L: 0 products with the explicit L^-1 (fastest), 1 rocBLAS trsm / trsv (backward stable),
                             // 2 (default) the products + one step of iterative refinement against L (accuracy of 1 at ~2/3 of its time)
    double L_diag_ratio = 1.0;  // max diag(L) / min diag(L) of the current K_uu factor (cglb_get_stat "L_diag_ratio")
    void* Guf = nullptr;     // adjoint of Kuf, same layout as At (allocated on first gradient)
    void *fragA = nullptr, *fragB = nullptr;  // MFMA-ordered augmented operands (kernels_kff_mfma.hip)
    size_t frag_cap = 0;
    bool frag_valid = false;  // fragA / fragB (operands of the experimental matrix-pipe variant) match the current hypers
    void* sym_items = nullptr;       // work list (row block, column chunk) of the symmetric mat-vec
    int64_t sym_n = -1, sym_chunk = 0, sym_chunk_opt = 0;
    double sym_pairs = 0.0;          // kernel pairs one launch of the symmetric pair kernel evaluates (current item list)
    // in-situ timing of the dominant kernel (cglb_set_option "k1_profile", cglb_get_stat): HIP event pairs around every launch of the
    // symmetric pair kernel on the context stream, resolved lazily
    bool k1_profile = false;
    std::vector<hipEvent_t> k1_events;
    size_t k1_events_used = 0;
    double k1_ms_total = 0.0;
    long long k1_launches = 0;
    // phase timing of cglb_objective_and_grad ("eval_profile"): 5 events per evaluation (start | common terms | PCG | final mat-vec +
    // preconditioner + scalars | gradient), resolved lazily by cglb_get_stat "eval_*_ms"
    bool eval_profile = false;
    std::vector<hipEvent_t> eval_events;
    size_t eval_events_used = 0;
    double eval_ms[4] = {0, 0, 0, 0};
    long long eval_count = 0;
    int grad_gram = 1;    // 1: Gram-form symmetric gradient pass (moments), 0: direct differences
    int aat_block = 512;  // block width of the lower-triangle-only split-K A A^T (0 or not dividing M: the full square)
    int sym_order = 1, sym_order_built = -1;  // item order of the symmetric kernel: 0 row-block major, 1 XCD-aware (kernels_kff_sym.hip)
    int sym_rbrows = 0, sym_nitems = 0, sym_world = 1, sym_rank = 0;
    int par_world = 1, par_rank = 0;  // cyclic distribution of the symmetric K_ff work over ranks (cglb_set_parallel)
    void* slabs = nullptr;   // split-K partial A A^T slabs [nslab][M][M]
    size_t slab_cap = 0;
    rocblas_int* info_dev = nullptr;
    double trace_AAt = 0, sum_log_diag_LB = 0;
    // work vectors (T): all length nloc unless noted
    void *w_r = nullptr, *w_z = nullptr, *w_p = nullptr, *w_Ap = nullptr, *w_Kv = nullptr, *w_e = nullptr;
    void* w_pfull = nullptr;       // [N] gathered p for single-shard solve
    void *w_u = nullptr, *w_t = nullptr, *w_t2 = nullptr;  // [M]
    void* kpart = nullptr;         // K_ff mat-vec partial row sums [jsplit][nloc]
    size_t kpart_cap = 0;
    void* tpart = nullptr;         // A^T t partial sums [msplit][nloc]
    double* dotpart = nullptr;     // block partials for dots (double always) [DOTPART_CAP]
    double* scal = nullptr;        // device scalars (double) [64]
    double* host_scal = nullptr;   // pinned host mirror for the asynchronous read of the stop-test scalar
    hipEvent_t scal_event = nullptr;
    int final_matvec = 0;          // 1: K v recomputed after the solve (models.py:280); 0 (default): K v = e - r from the residual the PCG recurrence carries
    int pcg_lookahead = 1;         // 1: enqueue the next mat-vec before waiting for the stop-test scalar (pcg_impl)
    double* gpart = nullptr;       // gradient partial buffers
    size_t gpart_cap = 0;
    double* gradbuf = nullptr;     // device packed gradient [GRAD_LEN]
    // tunables
    int precision = 1;  // CGLB_PREC_FAST (devmath.h): kernel values to <= 1e-13; 0 = CGLB_PREC_EXACT (~3e-16); 2 = CGLB_PREC_LOW (~1e-10, opt-in)
    int kff_variant = 2, kff_jsplit = 0, kff_rows = 4;  // 0 plain, 1 matrix-pipe Gram (fp64), 2 symmetric (default)
    bool exp_clamp = false;         // scaled operands so large that 2^x needs the range clamp (set by set_hypers)
    double m32_bias = 0.0;          // Matern-3/2: positivity bias of the squared distance in hot units^2 (devmath.h CGLB_M32_BIAS_*; set by set_hypers)
    bool kff_skip_combine = false;  // timing only: launch the pair kernel without the slab combine
    std::string err;
};

#define DOTPART_CAP 65536
// A mis-speculated mat-vec costs 2.9 ms at the headline shape, the stall it avoids 38 us: speculation must be right ~99 % of the time.
// Measured on 33 warm-started training evaluations (tools/lookahead_waste.py, profiles/r03_lookahead_waste.log): factor 4 wasted 16
// mat-vecs (median evaluation 32.1 ms), 16 one, 64 none (29.05 ms).
#ifndef CGLB_LOOKAHEAD_FACTOR
#define CGLB_LOOKAHEAD_FACTOR 32.0
#endif

#define HIP_CHECK(ctx, expr)                                                                         \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
            return CGLB_ERR_HIP;                                                                     \
        }                                                                                            \
    } while (0)

#define BLAS_CHECK(ctx, expr)                                                                        \
    do {                                                                                             \
        rocblas_status _s = (expr);                                                                  \
        if (_s != rocblas_status_success) {                                                          \
            (ctx)->err = std::string(#expr) + ": rocblas status " + std::to_string((int)_s);         \
            return CGLB_ERR_BLAS;                                                                    \
        }                                                                                            \
    } while (0)

#define CGLB_TRY(expr)                 \
    do {                               \
        int _rc = (expr);              \
        if (_rc != CGLB_OK) return _rc; \
    } while (0)

static inline int cglb_fail(cglb_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

static inline int pad_dim(int d) {
    const int sizes[] = {1, 2, 3, 4, 6, 8, 10, 12, 16, 20, 24, 28, 32};  // 10, 20 and 28: the reference's own data sets have D = 9, 17, 18, 26, 27
    for (int s : sizes)
        if (d <= s) return s;
    return d;  // wide inputs are not padded: their Gram products go through rocBLAS (kernels_wide.hip)
}
static inline bool is_wide(const cglb_ctx* c);

static inline bool is_wide(const cglb_ctx* c) { return c->Dp > CGLB_MAX_D_NARROW; }
// padded width of the hot operand set of a mid-width fp64 context (0: none)
static inline int mid_dim(int d, int dtype) {
    if (dtype != CGLB_F64 || d <= CGLB_MAX_D_NARROW || d > CGLB_MAX_D_MID) return 0;
    return d <= 48 ? 48 : d <= 64 ? 64 : d <= 80 ? 80 : 96;  // (a 128-wide row operand alone would fill the 256 VGPRs VALU instructions can address)
}
static inline bool mid_reg(const cglb_ctx* c) { return c->Dh > 0 && c->wide_reg != 0; }

// kernels_wide.hip: D > 32.  The Gram part of the pair value is a contraction with k = D and goes through rocBLAS in tiles; same
// contracts as the launchers below they stand in for.
int wide_prep_scaled(cglb_ctx* c, const void* Xraw, int64_t n, void* Xs_out, void* xa_out, void* Xsq_out);
int wide_after_hypers(cglb_ctx* c);
int wide_prep_hot(cglb_ctx* c);   // Xh (N x Dh, zero padded), xah in hot units for the register-resident mat-vec of a mid-width context
int wide_kuf(cglb_ctx* c);
int wide_kuu(cglb_ctx* c);
int wide_kus(cglb_ctx* c, const void* XsNew, const void* xaNew, int64_t n_new, int64_t ld, void* out);
int wide_matvec(cglb_ctx* c, const void* XsRow, const void* xaRow, int64_t row0_global, int64_t nrows, const void* p_full, void* out, bool diag_noise,
                double* pdot_slot, int tile_stride, int tile_offset);
int wide_grad_kff(cglb_ctx* c, const void* v_full, const void* u_rows, int64_t row0, int64_t nrows, int tile_stride, int tile_offset, double* out_dl);
int wide_grad_panel(cglb_ctx* c, const void* G, int64_t ldg, const void* cvec, const void* wvec, const void* XsCol, const void* xaCol, const void* XsqCol,
                    int64_t ncols, double zfactor, double* out);
void wide_free(cglb_ctx* c);

// ---- launchers implemented in the kernel translation units (all enqueue on ctx->stream) ----------
// kernels_prep.hip
int launch_prep_scaled(cglb_ctx* c, const void* Xraw, int64_t n, void* Xs_out, void* xa_out, bool hot = false);
double cglb_hot_scale(const cglb_ctx* c);  // xh = hot_scale * xs
int launch_select_inducing(cglb_ctx* c, double variance, double jitter, long long* chosen_dev, void* Z_out, double* trace_dev);  // kernels_select.hip
int launch_kuf(cglb_ctx* c);  // At <- Kuf[:, rows] (unscaled by sigma)
int launch_kuu(cglb_ctx* c);  // Lc <- Kuu + jitter I (full symmetric)
// kernels_kff.hip
int launch_kff_matvec(cglb_ctx* c, const void* p_full, void* out_local, double* pdot_slot);
int launch_cholesky_lower(cglb_ctx* c, void* A, int* info_slot);
int launch_frag_prep(cglb_ctx* c);
int launch_kff_sym(cglb_ctx* c, const void* p_full, void* out_local, double* pdot_slot);
int launch_hot_weights(cglb_ctx* c);
int launch_kff_sym_mid(cglb_ctx* c, const void* p_full, void* out, double* pdot_slot, bool cyclic);  // 32 < D <= 96, fp64 (kernels_kff_sym.hip)
int ensure_gpart(cglb_ctx* c, size_t need);   // partial-sum slab of the gradient passes (kernels_grad.hip)
int grad_fold_operands(cglb_ctx* c, const void* v_full, const void* u, int64_t off, int64_t n_u, bool* fold);  // column-side copies u o w, v o w (folded column norm)
int launch_grad_kff_mid(cglb_ctx* c, const void* v_full, const void* u_full, int world, int rank, double* out_dl);  // kernels_grad_mid.hip
int launch_hot_squares(cglb_ctx* c);  // Xhsq = Xh .* Xh after set_hypers
int k1_profile_collect(cglb_ctx* c);  // resolves the pending event pairs into k1_ms_total / k1_launches  // wh = 2^(xah/T) after set_hypers (RBF)
int launch_kff_sym_cyclic(cglb_ctx* c, const void* p_full, void* out_full_partial);  // this rank's share of the global upper triangle
int launch_grad_kff_cyclic(cglb_ctx* c, const void* v_full, const void* u_full, double* out_dl);
int launch_kff_plain_range(cglb_ctx* c, const void* p_full, int64_t col0, int64_t col1, void* part, int64_t* nslots);
int launch_kff_mfma_pairs(cglb_ctx* c, const double* p_full, int64_t* jsplit_out);
int launch_pairs_rect(cglb_ctx* c, const void* XsRow, const void* xaRow, int64_t nrows, const void* XsCol, const void* xaCol, const void* pcol,
                      int64_t col0, int64_t col1, void* out);
int launch_tri_rowdot(cglb_ctx* c, const void* Wrows, const void* x, int lower, void* out);
int launch_scale(cglb_ctx* c, void* x, double a, int64_t n);
int launch_precond_z_from(cglb_ctx* c, const void* r_local, const void* Ks_local, void* z_local, double* rz_slot);
int launch_cross_matvec(cglb_ctx* c, const void* Xs_new, const void* xa_new, int64_t n_new, const void* v_full, void* out);
// kernels_vec.hip
int launch_dot(cglb_ctx* c, const void* a, const void* b, int64_t n, double* out_slot);
int launch_update_v_r(cglb_ctx* c, void* v, void* r, const void* p, const void* Ap, const double* rz, const double* pAp, int update_r, int64_t n = -1);
int launch_residual(cglb_ctx* c, void* r, const void* b, const void* Kv, int64_t n = -1);
int launch_axpy(cglb_ctx* c, void* y, double alpha, const void* x, int64_t n);
int launch_update_p(cglb_ctx* c, void* p, const void* z, const double* new_rz, const double* rz, int restart, int64_t n = -1, bool fuse = false);
int launch_gemv_u(cglb_ctx* c, const void* r_local, void* u_out);               // u = A_loc r
int launch_tri_apply(cglb_ctx* c, const void* u, void* t_out);                  // t = LB^-T LB^-1 u
int launch_precond_z(cglb_ctx* c, const void* r_local, const void* t, void* z_local, double* rz_slot, void* rz_slot_T = nullptr);
int launch_update_p_seg(cglb_ctx* c, void* p, const void* zseg, int64_t n, int64_t per, int world, double* new_rz_out, const double* rz, int restart);
int launch_sub_scalar(cglb_ctx* c, void* out, const void* y_local, double mean, int64_t n);  // e = y - mean
int launch_tri_clean(cglb_ctx* c, void* Mc, int keep_lower);   // zero the other strict triangle
int launch_transpose(cglb_ctx* c, const void* src, void* dst); // M x M
int launch_add_identity_trace(cglb_ctx* c, void* Mc, double* trace_slot);        // trace then += I
int launch_sum_log_diag(cglb_ctx* c, const void* Mc, double* slot);
int launch_diag_minmax(cglb_ctx* c, const void* Mc, double* slot);  // slot[0..1] = min, max of the diagonal
int launch_symmetrize_lower(cglb_ctx* c, void* Mc);            // copy lower triangle to upper
int launch_obj_scalars(cglb_ctx* c, const void* v_local, const void* r, const void* Kv, const void* w, double* sc8);
// kernels_grad.hip
int launch_grad_kff(cglb_ctx* c, const void* v_full, const void* u_local, double* out_dl /* dev double[D], overwritten */);
int launch_grad_kuf(cglb_ctx* c, const void* cvec, const void* w_local, double* out /* packed gradient, accumulated */);
int launch_grad_kuu(cglb_ctx* c, const void* Guu, const void* cvec, const void* mhalf_c, double* out);
