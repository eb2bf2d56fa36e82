/*
 * vbmf_hip.h -- C ABI of the MI355X-native VB matrix-factorization sweep.
 *
 * The reference (vitskvara/VBMatrixFactorization.jl) has no FFI layer: its boundary for this path is
 * the Julia call surface of src/vbmf.jl.  Each entry point below names the reference interface it
 * replaces (file:line relative to the reference root).  A Julia host `ccall`s these (see
 * INTEGRATION.md and vbmatrixfactorization.jl_amd/julia/VBMatrixFactorizationHIP.jl); the tested
 * twin is the Python ctypes host in vbmatrixfactorization.jl_amd/.
 *
 * Conventions
 *   - extern "C", no exceptions cross the boundary.  Every function returns 0 on success or a
 *     negative vbmf_status; vbmf_last_error(ctx) gives the message (Julia `error(...)` analogue).
 *   - All matrices at the boundary are `double`, COLUMN-MAJOR, leading dimension >= rows (Julia
 *     Array{Float64,2} memory as is).  Indices are 0-based (the Julia wrapper subtracts 1 from labels).
 *   - Every pointer is borrowed for the duration of the call; the library copies in/out and never
 *     retains host pointers.  One ctx = one problem (one row-shard of it when nranks > 1) on one GPU;
 *     a ctx is not thread-safe, distinct ctxs are independent.
 *   - There is NO CPU fallback: vbmf_create fails (VBMF_ERR_NO_DEVICE) without a gfx950 device.
 */
#ifndef VBMF_HIP_H
#define VBMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vbmf_ctx vbmf_ctx;

typedef enum {
    VBMF_OK = 0,
    VBMF_ERR_INVALID = -1,     /* bad argument / shape / state */
    VBMF_ERR_NO_DEVICE = -2,   /* no usable HIP device (no CPU fallback exists) */
    VBMF_ERR_HIP = -3,         /* a HIP runtime call failed */
    VBMF_ERR_NUMERIC = -4,     /* non-positive pivot / non-finite value in the H x H algebra */
    VBMF_ERR_COMM = -5,        /* RCCL failure */
    VBMF_ERR_UNSUPPORTED = -6,
    VBMF_ERR_SYNC = -7         /* an in-launch hand-off gave up: the Y*A pass's register epilogue waited for the SigmaB
                                  table longer than its bounded spin allows (the control workgroup of the same launch
                                  did not run beside it); the state of that sweep is not valid */
} vbmf_status;

/* storage of Y on the device / arithmetic of the two streaming contractions */
#define VBMF_Y_F32 0           /* Y fp32, exact-f32 MFMA (v_mfma_f32_32x32x2_f32) */
#define VBMF_Y_BF16 1          /* Y bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulate */
/* factor operand (BHat in Y'B, AHat in Y*A) fed to the MFMA */
#define VBMF_FACTOR_AUTO 0     /* f32 with f32 Y; bf16x2 with bf16 Y */
#define VBMF_FACTOR_BF16 1     /* one bf16 rounding of the factor: ~3 significant digits in A/B, sigma2 only loosely (a speed
                                  option, not a parity mode); vbmf_create returns VBMF_ERR_UNSUPPORTED for H > VBMF_FACTOR_BF16_MAX_H */
#define VBMF_FACTOR_BF16_MAX_H 128
#define VBMF_FACTOR_BF16X2 2   /* hi+lo bf16 split (two MFMAs), ~16 mantissa bits */

#define VBMF_VARIANT_BASIC 0        /* src/vbmf.jl */
#define VBMF_VARIANT_SPARSE_DIAG 1  /* src/vbmf_sparse.jl, full_cov=false, diag_var=false */
#define VBMF_VARIANT_SPARSE_DIAGVAR 2 /* src/vbmf_sparse.jl, full_cov=false, diag_var=true: one noise precision per row */
#define VBMF_VARIANT_DUAL_DIAG 3    /* src/vbmf_dual.jl, full_cov=false, diag_var=false: two column groups A = [A0 A1] */
#define VBMF_VARIANT_TRIAL_DIAG 4   /* src/vbmf_trial.jl, full_cov=false, diag_var=false: three groups A = [A1 [A2; A3]] */
#define VBMF_VARIANT_DUAL_DIAGVAR 5  /* src/vbmf_dual.jl, full_cov=false, diag_var=true (rows' noise as in VBMF_VARIANT_SPARSE_DIAGVAR) */
#define VBMF_VARIANT_TRIAL_DIAGVAR 6 /* src/vbmf_trial.jl, full_cov=false, diag_var=true */

/* reference_compat bits (default: all set = behave like the reference) */
#define VBMF_COMPAT_SPECTRAL_DELTA 1u  /* d uses operator 2-norms (src/util.jl:27-29, Julia 0.5 norm) */
#define VBMF_COMPAT_SPARSE_REPEAT 2u   /* repeat(v, inner=M-1) layout of src/vbmf_sparse.jl:221 */
#define VBMF_COMPAT_DEFAULT 0xFFFFFFFFu

typedef struct {
    int32_t struct_size;       /* = sizeof(vbmf_opts); for ABI evolution */
    int32_t device;            /* HIP device ordinal */
    int32_t y_dtype;           /* VBMF_Y_* */
    int32_t factor_dtype;      /* VBMF_FACTOR_* */
    int32_t variant;           /* VBMF_VARIANT_* */
    uint32_t reference_compat; /* VBMF_COMPAT_* bitmask */
    int32_t nranks;            /* row-shards of Y (one process per GPU); 1 = single GPU */
    int32_t rank;
    int64_t L_global;          /* total rows over all ranks (0 => = L) */
    int64_t row_offset;        /* first global row owned by this ctx */
    int32_t pass1_splits;      /* split-K factor of the Y'B pass; 0 = auto */
    int32_t reserved;
} vbmf_opts;

/* update selectors for vbmf_step -- one bit per reference update function */
#define VBMF_STEP_A 1       /* updateA!      src/vbmf.jl:95-102 */
#define VBMF_STEP_B 2       /* updateB!      src/vbmf.jl:109-113 */
#define VBMF_STEP_CA 4      /* updateCA!     src/vbmf.jl:129-134 */
#define VBMF_STEP_CB 8      /* updateCB!     src/vbmf.jl:141-146 */
#define VBMF_STEP_SIGMA2 16 /* updateSigma2! src/vbmf.jl:153-157 */

void vbmf_default_opts(vbmf_opts* o);

/* One problem of L (local) x M with rank H.  Replaces the allocation half of vbmf_init
 * (src/vbmf.jl:48-73); the random draw stays on the host side. */
int vbmf_create(vbmf_ctx** out, int64_t L, int64_t M, int64_t H, const vbmf_opts* opts);
int vbmf_destroy(vbmf_ctx* ctx);
const char* vbmf_last_error(const vbmf_ctx* ctx); /* ctx may be NULL: error of a failed create */

/* Upload this rank's rows of Y (L x M, column-major, ld >= L): converts to the device dtype, builds
 * the two MFMA-fragment-tiled copies and caches ||Y||_F^2 (of the values as stored).  Replaces the
 * `Y::Array{Float64,2}` argument of vbmf!/updateA!/updateB!/updateSigma2! (src/vbmf.jl:95,109,153,175). */
int vbmf_set_Y(vbmf_ctx* ctx, const double* Y, int64_t ldY);
/* Device-side generator of the toy model (examples/toy_data.jl:7-18): Y = B* A*' + noise_std*N(0,1),
 * A* one-hot rows over Hstar columns, values rounded to the device dtype; counter-based, so the
 * matrix does not depend on nranks. */
int vbmf_set_Y_synthetic(vbmf_ctx* ctx, uint64_t seed, int64_t Hstar, double noise_std);
/* Read back Y exactly as stored on the device (rows [row0,row0+nrows) of this rank, column-major). */
int vbmf_get_Y(vbmf_ctx* ctx, double* Y, int64_t ldY, int64_t row0, int64_t nrows);
int vbmf_get_trYY(vbmf_ctx* ctx, double* trYY);

/* State = the numeric fields of `vbmf_parameters` (src/vbmf.jl:22-40).  AHat is M x H (ldA >= M),
 * BHat is L x H (this rank's rows, ldB >= L), SigmaA/SigmaB are H x H (ld = H), CA/CB are passed as
 * their diagonals (length H; the reference keeps them diagonal by construction, src/vbmf.jl:65-66,131,143).
 * labels0: 0-based rows of AHat whose last H1 columns are forced to zero (src/vbmf.jl:61,101). */
int vbmf_set_state(vbmf_ctx* ctx, const double* AHat, int64_t ldA, const double* BHat, int64_t ldB,
                   const double* SigmaA, const double* SigmaB, const double* CA_diag, const double* CB_diag,
                   double sigma2, const int64_t* labels0, int64_t nlabels, int64_t H1);
int vbmf_get_state(vbmf_ctx* ctx, double* AHat, int64_t ldA, double* BHat, int64_t ldB, double* SigmaA,
                   double* SigmaB, double* CA_diag, double* CB_diag, double* sigma2);

/* Apply the selected reference updates once, in the reference's order A, B, CA, CB, SIGMA2
 * (src/vbmf.jl:194-204).  For callers that drive the updates themselves (examples/mil_util.jl:183-185). */
int vbmf_step(vbmf_ctx* ctx, int which);

/* Fixed-basis inference, the reference's main caller of the update functions outside vbmf! -- vbls!
 * (examples/mil_util.jl:179-203, vbmf_parameters branch): niter x (updateA!, updateCA!, updateSigma2!) with BHat,
 * SigmaB, CB frozen.  B is fixed, so Y'B is formed once: ONE pass over Y per call instead of two per iteration. */
int vbmf_run_fixed_basis(vbmf_ctx* ctx, int64_t niter);

/* The vbmf! loop (src/vbmf.jl:187-214): while i <= niter && d > eps { A; B; [CA; CB]; [sigma2]; d }.
 * Runs entirely on the device; the stop test is evaluated device-side so the state freezes exactly
 * where the reference would stop.  iters_done = i-1 (src/vbmf.jl:221), d_last = last d.
 * trace (optional, niter x 4 doubles, row-major): per sweep d, sigma2, elbo, reserved.
 * eps: the reference's default 1e-6 (src/vbmf.jl:175) is at or below what d resolves on the device (BHat is stored in fp32 /
 * as bf16 hi + lo: d floors at ~1e-6 / ~1e-5 where the fp64 reference keeps falling).  A run that uses all niter sweeps with d
 * still above such an eps returns VBMF_OK and leaves a text starting with "note:" in vbmf_last_error (cleared by the next run);
 * the Python / Julia hosts turn it into a warning.
 * Row-sharded runs: a device error on one rank (VBMF_ERR_NUMERIC / VBMF_ERR_SYNC) stops every rank at the same sweep and every
 * rank returns that error class -- the ranks' error flags travel with the packed Gram all-reduce. */
int vbmf_run(vbmf_ctx* ctx, int64_t niter, double eps, int est_covs, int est_var, int64_t* iters_done,
             double* d_last, double* trace);

/* updateYHat! (src/vbmf.jl:120-122) on demand: YHat = BHat*AHat' (L x M column-major). */
int vbmf_get_YHat(vbmf_ctx* ctx, double* YHat, int64_t ld);
/* Build-defined ELBO of the basic model (the reference has none). */
int vbmf_elbo(vbmf_ctx* ctx, double* elbo);

/* ---- ARD-sparse variant: src/vbmf_sparse.jl with full_cov=false, diag_var=false (opts.variant =
 * VBMF_VARIANT_SPARSE_DIAG).  vec(A') is element-wise: index m*H + h (src/vbmf_sparse.jl:119,244-246).
 * Shapes: ATVecHat, diagSigmaATVec, CA, beta: M*H; BHat: L x H column-major; SigmaB: H x H; CB, delta: H.
 * sigmaHat is the noise PRECISION with Gamma posterior (eta, zeta) (src/vbmf_sparse.jl:35-39,321).
 * Derived constants (src/vbmf_sparse.jl:131,137,143): alpha = alpha0 + 1/2, gamma = gamma0 + L/2,
 * eta = eta0 + L*M/2 (L = L_global).  diag_var=true: VBMF_VARIANT_*_DIAGVAR + vbmf_sparse_set_noise_rows below;
 * full_cov=true: vbmf_sparse_set_full_cov below (per-column H x H blocks; the dense MH x MH fields are never formed). */
typedef struct { double alpha0, beta0, gamma0, delta0, eta0, zeta0; } vbmf_sparse_hyper;

#define VBMF_SSTEP_A 1      /* updateA! diagonal branch  src/vbmf_sparse.jl:204-247 */
#define VBMF_SSTEP_B 2      /* updateB!                  src/vbmf_sparse.jl:263-266 */
#define VBMF_SSTEP_CA 4     /* updateCA!                 src/vbmf_sparse.jl:284-288 */
#define VBMF_SSTEP_CB 8     /* updateCB!                 src/vbmf_sparse.jl:295-300 */
#define VBMF_SSTEP_SIGMA 16 /* updateSigma! homoscedastic src/vbmf_sparse.jl:317-321 */
#define VBMF_SSTEP_PRIORS 32 /* grouped models only, together with VBMF_SSTEP_CA: updateAlpha0g!, updateBeta0g!  src/vbmf_dual.jl:393-434, src/vbmf_trial.jl:442-507 */

int vbmf_sparse_set_state(vbmf_ctx* ctx, const double* ATVecHat, const double* diagSigmaATVec, const double* CA,
                          const double* beta, const double* BHat, int64_t ldB, const double* SigmaB,
                          const double* CB, const double* delta, double sigmaHat, double zeta,
                          const vbmf_sparse_hyper* hyper, const int64_t* labels0, int64_t nlabels, int64_t H1);
/* any output pointer may be NULL; SigmaA_diag: the diagonal of SigmaA (H), src/vbmf_sparse.jl:236-239 */
int vbmf_sparse_get_state(vbmf_ctx* ctx, double* ATVecHat, double* diagSigmaATVec, double* CA, double* beta,
                          double* SigmaA_diag, double* BHat, int64_t ldB, double* SigmaB, double* CB,
                          double* delta, double* sigmaHat, double* zeta);
/* Heteroscedastic rows (VBMF_VARIANT_SPARSE_DIAGVAR; src/vbmf_sparse.jl:207-212,229-230,256-261,308-315): the same
 * vbmf_sparse_* entry points then run the diag_var=true updates; the row-noise state travels separately:
 * sigmaVecHat, zetaVec (length L) and the common Gamma shape etaVec = eta0 + M/2 (:145-147).  set: after
 * vbmf_sparse_set_state (whose sigmaHat/zeta are ignored); get_state's sigmaHat is mean(sigmaVecHat). */
int vbmf_sparse_set_noise_rows(vbmf_ctx* ctx, const double* sigmaVecHat, const double* zetaVec, double etaVec);
int vbmf_sparse_get_noise_rows(vbmf_ctx* ctx, double* sigmaVecHat, double* zetaVec);
/* vbls! on the sparse model (examples/mil_util.jl:187-190): niter x (updateA!, updateCA!, updateSigma!), B frozen */
int vbmf_sparse_run_fixed_basis(vbmf_ctx* ctx, int64_t niter);
int vbmf_sparse_step(vbmf_ctx* ctx, int which);            /* reference order A, B, CA, CB, SIGMA (:369-376) */
/* vbmf_sparse! loop (src/vbmf_sparse.jl:344-410): returns d like the reference; trace: niter x 4 (d, sigmaHat, 0, 0) */
int vbmf_sparse_run(vbmf_ctx* ctx, int64_t niter, double eps, int est_cb, int64_t* iters_done, double* d_last,
                    double* trace);
/* lowerBound (src/vbmf_sparse.jl:435-471), verbatim quirks QS4; H(B) as L*logdet(SigmaB), clamped like
 * normalEntropy's det (src/util.jl:118-122) when clamp != 0 */
int vbmf_sparse_lower_bound(vbmf_ctx* ctx, int clamp, double* lb);
/* lowerBoundTrimmed (src/vbmf_sparse.jl:478-489; dual src/vbmf_dual.jl:606-617; trial src/vbmf_trial.jl:687-698; called by
 * examples/mil_util.jl:505): the bound with the entries of vec(A') whose |ATVecHat| <= trim removed from ATVecHat, beta, CA,
 * diagSigmaATVec and MH -- a mask in front of the M*H-long sums (AHat itself, and with it Y*AHat and AHat'AHat, stays
 * whole, as in the reference, which trims the vec fields only; the grouped models' per-group fields beta0/beta1/CA0/...
 * are not trimmed there either, so for them only MH, the CA-weighted second moment and H(vec(A')) change).
 * The comparison runs on the device's fp32 copy of ATVecHat. */
int vbmf_sparse_lower_bound_trimmed(vbmf_ctx* ctx, int clamp, double trim, double* lb);
/* full_cov = true of updateA! (src/vbmf_sparse.jl:178-202; dual :218-243; trial :252-277).  The reference's dense MH x MH
 * invSigmaATVec = sigmaHat*kron(I_M, B'B + L*SigmaB) + diag(CA) is block diagonal, so the device inverts the M H x H blocks
 * (one workgroup per column of Y) and never forms it: diagSigmaATVec = the blocks' diagonals, SigmaA = their sum (a full
 * H x H matrix).  Applies to VBMF_SSTEP_A, the run loops and run_fixed_basis from then on.  Up to H = 128 the H x H fp64 block of
 * a column lives in one workgroup's registers (two columns per round up to H = 64, one above); 128 < H <= 256 goes through a
 * blocked Schur inverse in a per-workgroup global workspace (~0.4 ms per column: for small M).  Either noise model: in the
 * *_DIAGVAR variants (:180-182, :192-193) the blocks are B' diag(sigmaVec) B + L mean(sigmaVec) SigmaB + diag(CA[m,:]) and the
 * mean carries no sigmaHat factor.  The dense SigmaATVec / invSigmaATVec fields are not materialised. */
int vbmf_sparse_set_full_cov(vbmf_ctx* ctx, int on);
/* SigmaA as a full H x H matrix, column-major (vbmf_sparse_set_state derives a diagonal one from diagSigmaATVec; set it
 * explicitly to continue a full_cov state) */
int vbmf_sparse_set_SigmaA(vbmf_ctx* ctx, const double* SigmaA);
int vbmf_sparse_get_SigmaA(vbmf_ctx* ctx, double* SigmaA);

/* ---- Two-group ARD variant (opts.variant = VBMF_VARIANT_DUAL_DIAG, or VBMF_VARIANT_DUAL_DIAGVAR for diag_var = true with the
 * rows' noise state of vbmf_sparse_set_noise_rows; src/vbmf_dual.jl, diagonal branch) -----------------
 * vbmf_dual_parameters (src/vbmf_dual.jl:59-112): A = [A0 A1], H = H0 + H1; the element-wise precisions of columns
 * h < H0 have the Gamma hyper-prior (alpha00, beta00), those of the other columns (alpha01, beta01); posterior shapes are
 * alpha0g + 1/2 (:324-325).  updateA!/updateB!/updateCB!/updateSigma! are the sparse model's bodies (no label mask), so the
 * state travels through vbmf_sparse_set_state / vbmf_sparse_get_state (CA and beta as the interleaved vectors of :146-165,
 * hyper.alpha0/beta0 = the initial value of both groups' priors), and vbmf_sparse_step, vbmf_sparse_run_fixed_basis
 * (vbls!, examples/mil_util.jl:190-193) and vbmf_sparse_lower_bound (lowerBound, src/vbmf_dual.jl:556-599) apply.
 * set_priors: after vbmf_sparse_set_state; alpha0 / alpha1 are the posterior shapes the fields of those names hold (what
 * the last updateCA! set; lowerBound reads them, :564-565).  get_priors: priors6 = {alpha00, beta00, alpha01, beta01, alpha0, alpha1}. */
int vbmf_dual_set_priors(vbmf_ctx* ctx, int64_t H0, double alpha00, double beta00, double alpha01, double beta01,
                         double alpha0, double alpha1);
int vbmf_dual_get_priors(vbmf_ctx* ctx, int64_t* H0, double* priors6);
/* vbmf_dual! loop (src/vbmf_dual.jl:455-530; convergence on BHat).  est_priors != 0 re-fits the four hyper-priors every
 * sweep (:491-495) on the device: alpha0g = the root of the reference's fAlpha0g on its bracket [1e-10, 1e10] (the
 * reference calls Roots.jl's fzero, a dependency it neither vendors nor pins: PARITY UNPINNED; unchanged when the bracket
 * holds no sign change, like the reference's `try ... end`), beta0g = M*Hg*alpha0g / sum(CAg).  trace as vbmf_sparse_run. */
int vbmf_dual_run(vbmf_ctx* ctx, int64_t niter, double eps, int est_cb, int est_priors, int64_t* iters_done,
                  double* d_last, double* trace);

/* ---- Three-group ARD variant (opts.variant = VBMF_VARIANT_TRIAL_DIAG / VBMF_VARIANT_TRIAL_DIAGVAR; src/vbmf_trial.jl,
 * diagonal branch) ---------------
 * vbmf_trial_parameters (src/vbmf_trial.jl:68-131): A = [A1 [A2; A3]] -- A1 the first H0 columns (all M rows), A2 / A3 the
 * other H1 columns of rows m < M0 / m >= M0; each block has its own Gamma hyper-prior (alpha0g, beta0g), g = 1, 2, 3
 * (:357-400).  Everything else is the two-group model's: state through vbmf_sparse_set_state / get_state, updates through
 * vbmf_sparse_step, vbls! through vbmf_sparse_run_fixed_basis (examples/mil_util.jl:194-197), lowerBound
 * (src/vbmf_trial.jl:630-680) through vbmf_sparse_lower_bound.
 * priors9 = {alpha01, beta01, alpha02, beta02, alpha03, beta03, alpha1, alpha2, alpha3} (the last three: the posterior shapes
 * the fields alpha1..alpha3 hold).  vbmf_trial_run: the vbmf_trial! loop (:528-604), est_priors as in vbmf_dual_run. */
int vbmf_trial_set_priors(vbmf_ctx* ctx, int64_t H0, int64_t M0, const double* priors9);
int vbmf_trial_get_priors(vbmf_ctx* ctx, int64_t* H0, int64_t* M0, double* priors9);
int vbmf_trial_run(vbmf_ctx* ctx, int64_t niter, double eps, int est_cb, int est_priors, int64_t* iters_done,
                   double* d_last, double* trace);


/* ---- preprocess (src/util.jl:73-86; examples/mil_util.jl:829) fused into the upload -------------------------
 * scaleY (:36-54: row mean / sqrt(row variance, n-1), variance <= 1e-15 -> 1, |y - mu| <= 1e-8 -> 0), drop the rows whose
 * scaled absolute sum is < 1e-5 (:75-78), multiply by lambda (:86).  Dropping rows changes L, which a context is
 * created with, hence two steps: open() uploads the caller's fp64 Y ONCE, keeps it resident on the device and
 * returns the kept-row count; the caller creates the context with that L and set_Y_preprocessed() tiles the kept
 * rows from the resident copy, transform applied on the fly (no second upload, no L x M temporaries). */
typedef struct vbmf_prep vbmf_prep;
int vbmf_preprocess_open(vbmf_prep** out, int device, const double* Y, int64_t L, int64_t M, int64_t ldY, int64_t* L_used);
/* optional read-back: kept rows (0-based, L_used of them), row means and denominators (L each); any may be NULL */
int vbmf_preprocess_rows(const vbmf_prep* plan, int64_t* used_rows0, double* mu, double* den);
int vbmf_set_Y_preprocessed(vbmf_ctx* ctx, const vbmf_prep* plan, double lambda);
int vbmf_preprocess_close(vbmf_prep* plan);

/* ---- multi-GPU: one process per GPU, Y row-sharded, RCCL all-reduce of Y'B and of the Grams ---- */
#define VBMF_UNIQUE_ID_BYTES 128
int vbmf_comm_unique_id(void* id128);                       /* rank 0 creates, host broadcasts */
int vbmf_comm_init(vbmf_ctx* ctx, const void* id128);       /* collective over all nranks ctxs */
/* Bring-up / test transport INSTEAD of RCCL: fn must sum `count` floats (is_double = 0) or doubles (1) of the DEVICE
 * buffer `buf` over all ranks in place, ordered after everything already enqueued on `hip_stream` and complete
 * (or stream-ordered) on return; returns 0 on success.  Lets N ranks share one GPU over a host-staged reduction
 * (tests/test_gpu_two_ranks.py), which RCCL refuses ("Duplicate GPU detected").  Not a performance path. */
typedef int (*vbmf_allreduce_fn)(void* user, void* buf, size_t count, int is_double, void* hip_stream);
int vbmf_comm_set_transport(vbmf_ctx* ctx, vbmf_allreduce_fn fn, void* user);

/* ---- measurement hooks (bench.py): HIP-event timing of the two streaming kernels ---- */
int vbmf_profile_enable(vbmf_ctx* ctx, int on);   /* 0: off; k > 0: time every k-th launch of each pass (the probe costs two
                                                     * event packets per timed launch, so a stride keeps it out of the throughput) */
/* out[0]=ms in Y'B pass, out[1]=launches, out[2]=ms in Y*A pass, out[3]=launches, out[4..7] reserved */
int vbmf_profile_read(vbmf_ctx* ctx, double* out8, int reset);
/* algorithmic bytes one launch of pass p (1|2) moves: Y once + factor in + result out */
int vbmf_pass_bytes(vbmf_ctx* ctx, int pass, double* bytes);
int vbmf_device_sync(vbmf_ctx* ctx);

/* ---- test hook: raw 32-bit words of an internal device buffer (layout tests, debugging) ---- */
#define VBMF_PEEK_P 0      /* pass-1 result slabs  [nsplit][Hp][Mp] fp32 */
#define VBMF_PEEK_Q 1      /* pass-2 result slabs  [nsplit][Hp][Lp] fp32 */
#define VBMF_PEEK_A32 2    /* AHat fp32 row-major  [Mp][Hp] */
#define VBMF_PEEK_B32 3    /* BHat fp32 row-major  [Lp][Hp] */
#define VBMF_PEEK_FA 4     /* AHat MFMA operand tiles */
#define VBMF_PEEK_FB 5     /* BHat MFMA operand tiles */
#define VBMF_PEEK_Y1 6     /* Y tiled for pass 1 */
#define VBMF_PEEK_Y2 7     /* Y tiled for pass 2 */
#define VBMF_PEEK_DIMS 8   /* int32 x 16: Hp, NH, mode, XT1, KS1, nsplit1, sps1, XT2, KS2, nsplit2, sps2, kstep, npart, narrow */
#define VBMF_PEEK_CHAIN 9  /* uint64 x 8 (16 words): last durations in 10 ns ticks of the in-launch control chain's parts
                              (ctrl_end, SigmaA, lambda_max(dB'dB) + loop test, SigmaB) and of the register epilogue's tail in
                              workgroup 0 of the Y*A pass (wait for + load of the SigmaB table, tiles, fold + store, reserved) */
int vbmf_debug_peek(vbmf_ctx* ctx, int what, uint32_t* out, int64_t nwords, int64_t word_offset);
/* tuning hook: average milliseconds of `iters` back-to-back launches of streaming pass p (1|2) alone */
int vbmf_debug_time_pass(vbmf_ctx* ctx, int pass, int iters, double* ms);
/* test hooks for the in-launch hand-off of the Y*A pass's register epilogue (VBMF_ERR_SYNC path) */
#define VBMF_DEBUG_EPI_SPIN_LIMIT 0   /* polls (~0.4 us each) the epilogue waits for the SigmaB table before giving up; default 2^22 */
#define VBMF_DEBUG_EPI_EXPECT_SKEW 1  /* != 0: the epilogue waits for a sequence number nobody publishes (forces the timeout) */
#define VBMF_DEBUG_SIGMA_B_PPM 2      /* test hook: the SigmaB / sigma2 table the B update multiplies by is scaled by (1 + value * 1e-6) --
                                         a deliberate, known-size regression that the parity asserts must catch (tests/test_gpu_soak.py) */
#define VBMF_DEBUG_EXACT_LAMBDA 3      /* lambda_max of the spectral norms at H <= 64: != 0 the Lanczos iteration every larger rank uses (exact inside
                                         eigenvalue clusters too), 0 (default) the repeated squaring: exact off clusters, up to ~2.5e-4 inside one, and
                                         faster -- 0.3-0.6 % of the sweep rate at 100k x 10k, 40 % on narrow problems and short row shards, where the
                                         control chain is the critical path.  Environment VBMF_EXACT_LAMBDA=1 at vbmf_create sets it for a whole job. */
int vbmf_debug_set(vbmf_ctx* ctx, int what, int64_t value);
/* test hook for the spectral norm behind `delta` (src/util.jl:27-29: norm(::Matrix) of Julia 0.5 = the largest singular value; for the
   H x H Gram the library keeps, its largest eigenvalue): lambda_max of the symmetric positive semi-definite H x H matrix G (column-major
   doubles, H = the context's rank) by the SAME device kernel the run loop uses for that rank (H <= 64: repeated squaring + Rayleigh
   quotient; H > 64: Lanczos), and the kernel's time in microseconds.  Overwrites the context's delta-Gram scratch (recomputed by every
   sweep), nothing else. */
int vbmf_debug_lambda_max(vbmf_ctx* ctx, const double* G, double* lambda_max, double* kernel_us);

#ifdef __cplusplus
}
#endif
#endif /* VBMF_HIP_H */
