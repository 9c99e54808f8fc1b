/* rcgp.h -- C ABI of librcgp.so: MI355X (gfx950) backend for romcomma's GP-regression + closed-form Sobol hot path.
 *
 * The reference (romcomma @ 2024_08_07) has no FFI of its own: the path is Python calling GPflow/TensorFlow.
 * Each entry point below replaces the GPflow/TF work behind one reference interface (paths relative to romcomma/):
 *
 *   rcgp_create / rcgp_set_y      GPR.__init__ pulling X (N,M), Y (N,L) from the Fold           gpr/models.py:306-308
 *                                 + MOGP.implementation (one gf.models.GPR per output)          gpr/models.py:340-342
 *   rcgp_set_hyper                RBF.implementation / Likelihood (variance, lengthscales, noise) gpr/kernels.py:172-177, gpr/models.py:341
 *   rcgp_lml, rcgp_lml_grad       gp.training_loss closure handed to gf.optimizers.Scipy          gpr/models.py:359-361
 *                                 + gp.log_marginal_likelihood()                                  gpr/models.py:365,370
 *   rcgp_factor                   MOGP.K_cho (Gram + noise + Cholesky), cached                    gpr/models.py:427-439
 *   rcgp_get_k_cho                MOGP.K_cho value for one output, (N,N) lower                    gpr/models.py:427-439
 *   rcgp_get_k_inv_y              MOGP.K_inv_Y for one output, (N,)                               gpr/models.py:441-444
 *   rcgp_predict                  MOGP.predict -> gp.predict_y / predict_f (mean, SD)             gpr/models.py:375-384
 *   rcgp_predict_gradient         MOGP.predict_gradient (tape.jacobian of K(X,x), triangular_solve) gpr/models.py:386-415
 *   rcgp_sobol_closed             ClosedSobol._calibrate/_V/marginalize, diagonal (l = j) term    gsa/calibrators.py:49-97
 *   rcgp_sobol_cross              the same einsum's off-diagonal (l != j) entries                 gsa/calibrators.py:79
 *   rcgp_sobol_error_terms        ClosedSobolWithError: mu_phi_mu, psi_factor, mu_psi_mu          gsa/calibrators.py:259-322
 *   rcgp_lml_grad_batch,          the same closure / K_cho for several (fold, output) units at once: the loops   gpr/models.py:340-342,360-361
 *   rcgp_factor_batch             over outputs and folds that the reference walks sequentially                   user/run.py:60-61,132-133
 *
 * Conventions: all matrices row-major float64; the caller owns every host buffer; the library owns device memory inside
 * the opaque handle until rcgp_destroy. A handle is bound to one device; it is not thread-safe. Distinct handles are
 * independent in what they compute, but all handles of a process on one device share one set of HIP streams (created with
 * the first handle, kept for the life of the process): their work is ordered behind each other on the device, so using two of
 * them from two threads gains nothing over using them in turn -- to have one GPU work on several handles AT ONCE use the batched
 * entries (rcgp_lml_grad_batch, rcgp_factor_batch). Return value: 0 = ok; k > 0 = LAPACK-style "leading minor k is not positive
 * definite" (TensorFlow raises InvalidArgumentError there); < 0 = bad argument (-1..-9) or HIP error (-100 - hipError_t).
 * rcgp_last_error gives the message. One process per GPU for multi-GPU runs.
 */
#ifndef RCGP_H
#define RCGP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rcgp_handle_s* rcgp_handle;

/* Library/ABI version (major*100 + minor). */
int rcgp_version(void);
/* Number of visible HIP devices, or a negative error. Does not create a context on any of them. */
int rcgp_device_count(void);

/* Upload the training fold: X is (N, M) row-major, y is (N). 1 <= M <= 256 (up to 64 dimensions the fused kernels stage a tile's inputs
 * once; beyond that chunk by chunk: slower per evaluation, same results). */
int rcgp_create(rcgp_handle* out, int device, int64_t N, int M, const double* X, const double* y);
int rcgp_destroy(rcgp_handle h);
const char* rcgp_last_error(rcgp_handle h);

/* Replace the output column (same X): one handle can serve the L independent outputs in turn. */
int rcgp_set_y(rcgp_handle h, const double* y);
/* Hyper-parameters in the constrained space: lengthscales ell[M] (pass the isotropic value M times), kernel variance,
 * likelihood (noise) variance. Invalidates any cached factor, unless every value is bit-identical to the current one (the
 * fit driver sets the optimum again after the last evaluation: nothing is recomputed then). */
int rcgp_set_hyper(rcgp_handle h, const double* ell, double variance, double noise);

/* Log marginal likelihood at the current hyper-parameters (Gram + Cholesky + forward solve). */
int rcgp_lml(rcgp_handle h, double* lml);
/* LML and its gradient w.r.t. (ell[0..M-1], variance, noise), constrained space; grad has M + 2 entries. */
int rcgp_lml_grad(rcgp_handle h, double* lml, double* grad);

/* Factor and cache L = chol(K + noise I), L^-1 and alpha = K^-1 y for predict / K_inv_Y / Sobol. */
int rcgp_factor(rcgp_handle h);
int rcgp_get_k_inv_y(rcgp_handle h, double* out /* N */);
int rcgp_get_k_cho(rcgp_handle h, double* out /* N*N, lower, zeros above */);
/* The Gram matrix K + noise I itself (lower triangle valid, upper mirrored), for tests. Invalidates the cached factor. */
int rcgp_get_gram(rcgp_handle h, double* out /* N*N */);

/* Posterior at n new points Xnew (n, M): mean[n] and standard deviation sd[n] (SD, not variance: gpr/models.py:384).
 * include_noise != 0 is predict_y, 0 is predict_f. */
int rcgp_predict(rcgp_handle h, int64_t n, const double* Xnew, int include_noise, double* mean, double* sd);

/* Gradient GP at n points (gpr/models.py:386-415): mean[(o*M + m)] = sum_N d k(X_N, x_o)/d x_om * alpha_N and
 * cov[(O*M + P) * (n*M) + (o*M + p)] = sum_N V[N][O,P] V[N][o,p] with V = L^-1 d k(X, x)/dx. The caller forms the reference's
 * var = -cov + diag term. Any n: the n * M derivative rows are processed 4096 at a time (memory for Np * n*M + (n*M)^2 doubles is the
 * only bound). */
int rcgp_predict_gradient(rcgp_handle h, int64_t n, const double* Xnew, double* mean, double* cov);

/* Closed-form Sobol conditional variances for this handle's output: V[s] for each slice [slices[2s], slices[2s+1]) of
 * the input dimensions (gsa/models.py:77-90). Requires rcgp_factor. */
int rcgp_sobol_closed(rcgp_handle h, int n_slices, const int32_t* slices, double* V);
/* Cross-output term V_lj: l = this handle's output, j given by its lengthscales ell_j[M], kernel variance var_j and
 * alpha_j[N] = K_j^-1 y_j (from rcgp_get_k_inv_y of the other output, possibly gathered from another GPU). */
int rcgp_sobol_cross(rcgp_handle h, const double* ell_j, double var_j, const double* alpha_j, int n_slices, const int32_t* slices,
                     double* V);

/* Ingredients of the standard errors T, W of the Sobol indices (ClosedSobolWithError, gsa/calibrators.py:146-402) for the
 * output pair (a, b): b = this handle's output (its Cholesky factor enters through psi_factor, :290-309); a = the same output
 * when ell_a == alpha_a == NULL, else the output with lengthscales ell_a[M], kernel variance var_a and alpha_a[N] = K_a^-1 y_a.
 * For every slice s -- ANY [m0, m1) as in ClosedSobolWithError.marginalize (:348-373); the first-order [m,m+1), closed [0,m) and
 * total-complement [m,M) slices that GSA asks for all come from one pass -- four numbers, WITHOUT the doubling of
 * a == b entries the reference applies (:281, :284, :322):
 *   phi_d = mu_phi_mu term of the DIAGONAL rank equations (:259-288), psi_d = |psi_factor_ab|^2 (:311-322),
 *   phi_m, psi_m = the same under the MIXED rank equation (used when is_T_partial is false).
 * The caller assembles W = (phi - psi) + transpose and T (:324-346). Requires rcgp_factor. Beyond M = 29 the canonical
 * slices take several passes (their column accumulators no longer fit in LDS together); beyond M = 64 the inputs pass through LDS in chunks. */
int rcgp_sobol_error_terms(rcgp_handle h, const double* ell_a, double var_a, const double* alpha_a, int n_slices, const int32_t* slices,
                           double* phi_d, double* psi_d, double* phi_m, double* psi_m);

/* ---- covariant (dependent multi-output) GP: the reference's romcomma.gpf path ----
 * One GP over L outputs sharing the N training inputs (gpf/models.py:33-139). System index a = l * N + n, output-major, as
 * the reference stacks Y (gpf/models.py:120). Kernel (gpf/kernels.py:93-104, 153-154) and likelihood (gpf/likelihoods.py:61-64):
 *   K[(l,n),(j,n')] = F[l][j] exp(-1/2 sum_m (x_nm / ell[l][m] - x_n'm / ell[j][m])^2) + Sigma[l][j] [n == n'].
 * rcgp_lml, rcgp_factor, rcgp_get_k_inv_y (L*N values = the reference's (L,1,N)), rcgp_get_k_cho / rcgp_get_gram ((L*N)^2),
 * rcgp_set_y (Y as N x L) and the stage / profiling entries work on such a handle; the single-output entries
 * (rcgp_set_hyper, rcgp_lml_grad, rcgp_predict, rcgp_predict_gradient, rcgp_sobol_closed/cross/error_terms) return -2. */
/* 1 <= M <= 256, 1 <= L <= 64. */
int rcgp_create_mo(rcgp_handle* out, int device, int64_t N, int M, int L, const double* X /* N x M */, const double* Y /* N x L */);
/* ell (L x M), F and Sigma (L x L, symmetric; diag(F) > 0). */
int rcgp_set_hyper_mo(rcgp_handle h, const double* ell, const double* F, const double* Sigma);
/* LML and its partial derivatives with every entry of ell, F and Sigma treated as independent (the caller applies the
 * Cholesky parametrisation of gpf/base.py:32-96 and the transforms): g_ell (L x M), g_F and g_Sigma (L x L, symmetric). */
int rcgp_lml_grad_mo(rcgp_handle h, double* lml, double* g_ell, double* g_F, double* g_Sigma);
/* MOGPR.predict_f / predict_y without full covariances (gpf/models.py:84-113, gpr/models.py:377-384): mean and SD as (n, L). */
int rcgp_predict_mo(rcgp_handle h, int64_t n, const double* Xnew, int include_noise, double* mean, double* sd);
/* Gradient GP of a covariant GP (gpr/models.py:386-415, covariant branch). Row r = (l * n + o) * M + m stands for d/dx_om of output l
 * at point o. mean[r] = sum_{(Lb,N)} dK[(Lb,N)][r] alpha[(Lb,N)]; cov[Lb][r][r'] = sum_N V[(Lb,N)][r] V[(Lb,N)][r'] with
 * V = L^-1 dK, ONE product per training output block Lb -- the reference's einsum 'LNlOM, LNlom -> OLolMm' (:398) keeps that index.
 * cov holds L * (L n M)^2 doubles; the caller applies the sign, picks l = l' and adds the diagonal term (:399-400, :406).
 * Any n (rows processed 4096 at a time). */
int rcgp_predict_gradient_mo(rcgp_handle h, int64_t n, const double* Xnew, double* mean, double* cov);
/* Closed-form Sobol with a non-diagonal F (gsa/calibrators.py:60-97 with is_F_diagonal false) works on "virtual outputs"
 * p = (l, J): phi_p = 1/(ell_l ell_J + 1), pre_p = F[l][J] sqrt(prod_m ell_lm ell_Jm phi_pm), alpha_p = K_inv_Y[J]. Weight vector
 * g_p[n] = pre_p exp(-1/2 sum_m phi_pm x_nm^2) alpha_p[n] - shift_p, the shift being the mean over (J, n) for that l (:90).
 * rcgp_sobol_weight_sum returns sum_n of the unshifted g_p; rcgp_sobol_pair returns, for every slice, the pair form
 * sum_{n,n'} g_a[n] g_b[n'] prod_{m in slice} h_m(n,n'); V_lj is its sum over the virtual outputs (l, .) x (j, .) (:79).
 * Both only use the handle's design matrix X (any handle on the same X will do). */
int rcgp_sobol_weight_sum(rcgp_handle h, const double* phi /* M */, double pre, const double* alpha /* N */, double* sum);
/* rcgp_sobol_error_terms for the output pair (out_a, out_b) of a covariant handle whose F is taken as diagonal (the only case
 * the reference computes errors for, gsa/calibrators.py:380-381): output l is (ell[l], F[l][l], its N entries of K_inv_Y), and
 * psi_factor solves with the Cholesky factor of the whole (L N) system, the vector sitting in block out_b (:304-308). */
int rcgp_sobol_error_terms_mo(rcgp_handle h, int out_a, int out_b, int n_slices, const int32_t* slices, double* phi_d, double* psi_d,
                              double* phi_m, double* psi_m);
int rcgp_sobol_pair(rcgp_handle h, const double* phi_a, double pre_a, const double* alpha_a, double shift_a, const double* phi_b,
                    double pre_b, const double* alpha_b, double shift_b, int n_slices, const int32_t* slices, double* V);

/* ---- several units at once on one GPU ----
 * The reference fits its L independent outputs and its K folds one after the other (the loops at gpr/models.py:340-342, 360-361 and
 * user/run.py:60-61, 132-133). A single factorisation of the sizes it is run at (benchmark_script.py:35-40: N <= 9840) cannot fill this
 * chip -- its chain of N/128 diagonal-block steps is latency-bound -- so these entries run ONE schedule over n handles (units): every
 * launch covers all units (the unit is blockIdx.z), the chains advance side by side, the update kernels of all units share the CUs.
 * Requirements: 1 <= n <= 16 distinct single-output handles of one device with equal M and equal padded size ceil(N / 128) (N itself may
 * differ: folds of a K-fold split), hyper-parameters set on each. A unit's numbers are bit-identical to what the single-handle call
 * returns for it. Units whose factor (or L^-1) is still valid are not recomputed.
 * status[u] = 0, or k > 0 when unit u's matrix is not positive definite at leading minor k (its lml / grad entries are NaN, the other
 * units are unaffected). Return value: 0 when the call ran (look at status), < 0 for a bad argument or a HIP error (rcgp_last_error(hs[0])). */
/* LML and gradient of every unit: lml[n], grad[n * (M + 2)] (unit-major, each as rcgp_lml_grad). */
int rcgp_lml_grad_batch(int n, rcgp_handle* hs, double* lml, double* grad, int* status);
/* rcgp_factor for every unit (factor, L^-1 and alpha cached on each). */
int rcgp_factor_batch(int n, rcgp_handle* hs, int* status);
/* Stage-level form for bench.py and the kernel tests: stage 0 = rcgp_stage_gram, 1 = rcgp_stage_potrf, 2 = rcgp_stage_trtri on all n units. */
int rcgp_stage_batch(int stage, int n, rcgp_handle* hs);

/* ---- stage-level entry points used by bench.py and the kernel tests ---- */
int rcgp_stage_gram(rcgp_handle h);      /* Z = X/ell; A = K + noise I (lower tiles) */
int rcgp_stage_potrf(rcgp_handle h);     /* blocked Cholesky in place + w = L^-1 y; requires rcgp_stage_gram */
int rcgp_stage_trtri(rcgp_handle h);     /* L^-1 and alpha; requires rcgp_stage_potrf */
int rcgp_sync(rcgp_handle h);

/* Process-wide work counters since the library was loaded, in units (a batched call counts each of its units): which = 0 Cholesky
 * factorisations, 1 inversions (L^-1 + alpha), 2 gradient evaluations, 3 batched rcgp_lml_grad_batch calls. For bench.py's count of the
 * factorisations one drop-in run.gpr + run.gsa pass costs (the reference factors at least eight times besides its fit: gpr/models.py:365,
 * 370, 439; gsa/calibrators.py:126-127 for each of three kinds). -1 for an unknown `which`. */
int64_t rcgp_stat(int which);

/* ---- profiling: HIP events around every kernel launch on the handle's stream ---- */
enum { RCGP_K_GRAM = 0, RCGP_K_GEMM = 1, RCGP_K_DIAG = 2, RCGP_K_SOBOL = 3, RCGP_K_MISC = 4, RCGP_K_GRAD = 5, RCGP_K_COUNT = 6 };
int rcgp_set_profiling(rcgp_handle h, int on);
int rcgp_profile_reset(rcgp_handle h);
/* For kernel class cls: number of launches, summed launch duration (ms) and summed ALGORITHMIC work since the last
 * reset: bytes for GRAM, flops for GEMM, DIAG and GRAD (k_grad alone: N^3/3 per launch), exp evaluations for SOBOL.
 * Kernels of the look-ahead Cholesky overlap on several streams, so the summed GEMM durations exceed the wall time. */
int rcgp_profile_get(rcgp_handle h, int cls, int64_t* launches, double* total_ms, double* work);

#ifdef __cplusplus
}
#endif
#endif
