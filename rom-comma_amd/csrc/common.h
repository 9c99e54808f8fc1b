// Shared declarations for librcgp (gfx950 only): handle layout, error macros, launch prototypes.
// The public C ABI is include/rcgp.h; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string>
#include <tuple>
#include <vector>

#define RC_TILE 128            // edge of a workgroup tile and of a diagonal Cholesky block
#define RC_BK 16               // k-depth of one LDS stage of the MFMA GEMM
#define RC_MAX_M 64            // largest input dimensionality of the fused FAST kernels (one staging of the Z / X panels per tile), of a covariant
                               // GP and of the Sobol standard errors
#define RC_MAX_M_WIDE 256      // largest input dimensionality of a single-output handle: beyond RC_MAX_M the Gram, gradient and Sobol kernels stage
                               // their panels chunk by chunk (k_gram in chunks of 64 dimensions, k_grad<.., WIDE> of 32, k_sobol_pairs of 64)
#define RC_MAX_L 64            // most outputs of one covariant GP (a bound on the arguments only: nothing is sized by it)
#define RC_SCAL_ELEMS 512      // h->scal: [0,2) LML sums, [RC_SCAL_INFO] the Cholesky status word, [8, 8+M+2) gradient sums
#define RC_SCAL_INFO 4
#ifndef RC_TRTRI_HALF_TILES
#define RC_TRTRI_HALF_TILES 1024
#endif
#ifndef RC_TAIL_BLOCKS
#define RC_TAIL_BLOCKS 64
#endif
#ifndef RC_FAR_GROUP
#define RC_FAR_GROUP 2
#endif
#ifndef RC_FAR_SPLIT_MIN
#define RC_FAR_SPLIT_MIN 40
#endif
#ifndef RC_LEAN_BLOCKS
#define RC_LEAN_BLOCKS 40
#endif
#define RC_NB_OUTER 1024       // outer panel width of the blocked Cholesky (K of the trailing update)
#define RC_MAX_BATCH 16        // most units (handles of equal padded size on one device) that one batched evaluation takes (rcgp_lml_grad_batch)

typedef double v4d __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// Batched units. Every kernel on the evaluation path (Gram, the whole factorisation, L^-1, alpha, K^-1 / gradient, the reductions) takes
// its per-unit pointers as a small table in the kernel arguments and picks its unit from blockIdx.z: ONE schedule -- the same launches on the
// same five streams in the same order -- serves 1..RC_MAX_BATCH independent units of equal padded size, so that the latency-bound chain
// steps of a factorisation (one workgroup per unit and step) cost what they cost for one unit and the column / update kernels of all
// units fill the chip together. A unit's arithmetic does not depend on who shares the launch: its results are bit-identical to a
// launch of its own. (Reference: the loops over outputs and folds that the reference runs one after the other, gpr/models.py:340-342,
// 360-361, user/run.py:60-61.)
// ---------------------------------------------------------------------------------------------------------------------
template <class T> struct RcBP { T* p[RC_MAX_BATCH]; };      // one pointer per unit
struct RcBN { int64_t v[RC_MAX_BATCH]; };                    // one count per unit (valid rows N: units share Np and M, not necessarily N)

// Profiled kernel classes (rcgp_profile_get). Order is part of the ABI (include/rcgp.h).
enum RcKernelClass {
  RC_K_GRAM = 0,       // Gram build
  RC_K_GEMM = 1,       // fp64 MFMA GEMM family (trailing update, trsm, trtri, predict), except k_grad
  RC_K_DIAG = 2,       // 128x128 diagonal-block potrf + inverse
  RC_K_SOBOL = 3,      // Sobol pair kernel
  RC_K_MISC = 4,       // reductions, gemv, prep
  RC_K_GRAD = 5,       // k_grad: fused K^-1 + LML-gradient reduction (the largest single kernel of an evaluation)
  RC_K_COUNT = 6
};

struct RcProfEvent { hipEvent_t start, stop; int cls; };

struct rcgp_handle_s {
  int device = 0;
  bool streams_acquired = false;     // this handle holds a reference on its device's shared stream set (api.hip)
  hipStream_t stream = nullptr;      // main stream: every public call is ordered on it (in rc_potrf: the window pieces)
  hipStream_t stream2 = nullptr;     // high priority: the chain of diagonal kernels + the chain's tile inside rc_potrf
  hipStream_t stream3 = nullptr;     // bulk trailing update of the look-ahead Cholesky
  hipStream_t stream5 = nullptr;     // high priority: column work of the panel chain (panel solve, near update)
  hipStream_t stream6 = nullptr;     // high priority: the far part of that column work (block columns the next chain step does not read)
  bool prep_attr_set = false, subst_attr_set = false, diag_attr_set = false;   // dynamic-LDS attributes set on this device
  hipStream_t launch = nullptr;      // the stream kernels are currently launched on
  hipEvent_t launch_stop = nullptr;  // if set: the next RC_LAUNCH attaches this event to its dispatch (no separate marker packet)
  int prof_pending = -1;             // index of the profiling bracket whose events the next RC_LAUNCH carries
  std::vector<hipEvent_t> la_events; // look-ahead dependency events (no timing), handed out in order by next_event (potrf.hip)
  size_t la_cursor = 0;
  // the run-time knobs of the factorisation's schedule (environment, read ONCE per process: api.hip): RCGP_LOOKAHEAD, RCGP_FINE, RCGP_NB,
  // RCGP_DEPTH, RCGP_EXT, RCGP_TAIL, RCGP_LEAN. far_split_min and far_group are compile-time only (-DRC_FAR_SPLIT_MIN, -DRC_FAR_GROUP).
  bool lookahead = true;             // RCGP_LOOKAHEAD: 0 = strictly sequential potrf on the main stream
  bool fine_chain = true;            // RCGP_FINE: 0 = one stream per panel chain (D, T, G in order), one-panel look-ahead
  int64_t nb_outer = RC_NB_OUTER;    // RCGP_NB: outer panel width
  int64_t batch_nb_outer = 512;      // ... and what a BATCHED factorisation uses (potrf.hip): the environment's values if given, else 512 / 16 --
  int batch_tail_blocks = 16;        // several units fill the chip, so the K = NB updates pay earlier (measured: DESIGN.md section 5)
  int chain_depth = 2;               // RCGP_DEPTH >= 1: column panels updated by their own kernels ahead of the bulk trailing update
  int tail_blocks = RC_TAIL_BLOCKS;   // RCGP_TAIL: the last this-many block columns of the factorisation form one fine-grained panel (potrf.hip)
  int lean_blocks = RC_LEAN_BLOCKS;   // RCGP_LEAN: chain steps with at most this many blocks below them carry ONE completion signal (potrf.hip)
  int far_split_min = RC_FAR_SPLIT_MIN;   // columns taller than this many blocks: a far update as two launches (potrf.hip)
  int far_group = RC_FAR_GROUP;      // far updates of the panel chain in groups of this many steps (K = 128 x group), potrf.hip
  int chain_ext = 4;                 // RCGP_EXT >= 1: 128-blocks past its own panel that a chain step keeps up to date
  int64_t trtri_half_tiles = RC_TRTRI_HALF_TILES;   // L^-1 launches of at most this many 128^2 tiles run on 64 x 128 half tiles (gemm.hip)
  int64_t N = 0, Np = 0;       // training rows per output; rows of the whole system, L * Nb
  int64_t Nb = 0;              // N padded to a multiple of RC_TILE: rows of one output block (Nb == Np when L == 1)
  int L = 1;                   // outputs modelled jointly (covariant GP, rcgp_create_mo): system row a = l * Nb + n
  int M = 0;
  // hyper-parameters (constrained space)
  std::vector<double> ell;     // L x M
  std::vector<double> Fm, Sm;  // L x L kernel variance and likelihood variance; var == Fm[0], noise == Sm[0] when L == 1
  double var = 1.0, noise = 0.0;
  bool hyper_set = false;
  // state flags
  bool gram_fresh = false;     // A holds the Gram matrix of the current hyper-parameters, not yet factored (rcgp_stage_potrf needs it)
  bool factored = false;       // A holds L (lower), w = L^-1 y, logdiag valid
  bool inverted = false;       // Linv holds L^-1, alpha valid
  // device buffers
  double *X = nullptr;         // Np x M (padded rows are 0; with L > 1 the N x M inputs repeated once per output block)
  double *Z = nullptr;         // Np x M, X / ell
  double *sq = nullptr;        // Np, -0.5 |z_i|^2
  double *y = nullptr;         // Np (padded 0): pristine targets
  double *w = nullptr;         // Np: running rhs during potrf, then L^-1 y
  double *alpha = nullptr;     // Np: K^-1 y
  double *A = nullptr;         // Np x Np: Gram, then L in the lower triangle
  double *Linv = nullptr;      // Np x Np: L^-1 (allocated on first use)
  double *S = nullptr;         // Np x Np scratch (trtri temporaries, predict) (allocated on first use)
  double *invdiag = nullptr;   // (Np/128) x 128 x 128: the eight 16x16 diagonal-block inverses of every diagonal block of L (in its diagonal 16-blocks)
  double *logdiag = nullptr;   // Np: log L_ii
  double *partial = nullptr;   // scratch for two-stage reductions
  size_t partial_elems = 0;
  double *scal = nullptr;      // small device scalars/vectors for results
  int *info = nullptr;         // device: 0 ok, k>0 = leading minor k not positive definite (lives inside scal: one copy fetches both)
  double *ell_d = nullptr;     // device copy of ell (L x M), followed in the same allocation by
  double *FS_d = nullptr;      // ... the device copies of Fm then Sm (2 x L x L)
  double *pin = nullptr;       // pinned host staging: [0, pin_result) hyper-parameters going up, [pin_result, ...) results coming down
  size_t pin_elems = 0, pin_result = 0;   // (pin_result: first element of the result block)
  hipEvent_t ev_hyper = nullptr;   // the last hyper-parameter upload has left the staging buffer
  bool hyper_in_flight = false;
  bool hyper_dirty = false;        // the host copy of the hyper-parameters is newer than the device's: uploaded by the next call that needs them
                                   // (one copy per handle, or ONE copy + one scatter kernel for all units of a batched call: api.hip)
  // predict scratch
  double *Xs = nullptr, *Zs = nullptr, *sqs = nullptr, *KsT = nullptr, *pmean = nullptr, *pvar = nullptr;
  int64_t pred_cap = 0;        // rows of KsT / pmean / pvar
  int64_t pts_cap = 0;         // points Xs / Zs / sqs hold (>= pred_cap; predict_gradient grows it)
  double *gV = nullptr, *gC = nullptr;   // predict_gradient scratch
  int64_t g_rows = 0;
  int g_blocks = 0;                       // V^T V products gC has room for
  // sobol scratch
  double *sob = nullptr;       // prep arrays
  size_t sob_elems = 0;
  // profiling
  bool profiling = false;
  std::vector<RcProfEvent> prof_events;
  std::vector<hipEvent_t> event_pool;     // recycled events (no create/destroy on the launch path after warm-up)
  double prof_ms[RC_K_COUNT] = {0, 0, 0, 0, 0, 0};
  long prof_count[RC_K_COUNT] = {0, 0, 0, 0, 0, 0};
  double prof_work[RC_K_COUNT] = {0, 0, 0, 0, 0, 0};   // algorithmic flops (GEMM) or bytes (Gram) or pair-terms (Sobol)
  // batch: the units a launch covers while this handle leads a batched call (api.hip: RcBatchScope). bh[0] == this.
  int nb = 1;
  rcgp_handle_s* bh[RC_MAX_BATCH] = {};
  double* bres_d = nullptr;    // 2 x RC_MAX_BATCH x RC_SCAL_ELEMS: the result blocks of a batched evaluation led by this handle, then the staging
  double* bres_pin = nullptr;  // of the units' hyper-parameters on their way up (allocated on first use); ... and their pinned host copy
  std::string err;
};

// Per-unit pointer tables for a launch led by h: `ptr` points into one of the LEADER's device buffers; unit u gets the same offset in its own
// buffer of that name. (The schedule code computes every address from the leader's buffers and knows nothing of batches.)
int rc_bp_bytes(rcgp_handle_s* h, const void* ptr, void** out);
template <class T> inline int rc_bp(rcgp_handle_s* h, T* ptr, RcBP<T>& out) {
  if (h->nb <= 1) { out.p[0] = ptr; for (int u = 1; u < RC_MAX_BATCH; ++u) out.p[u] = nullptr; return 0; }
  return rc_bp_bytes(h, (const void*)ptr, (void**)out.p);
}
inline RcBN rc_bn(rcgp_handle_s* h) {
  RcBN n;
  for (int u = 0; u < RC_MAX_BATCH; ++u) n.v[u] = (u < h->nb && h->nb > 1) ? h->bh[u]->N : h->N;
  return n;
}
#define RC_BP(T, name, ptr)                          \
  RcBP<T> name;                                      \
  {                                                  \
    const int rcbp_ = rc_bp<T>(h, (ptr), name);      \
    if (rcbp_) return rcbp_;                         \
  }

#define RC_HIP(call)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return -100 - (int)e_;                                                               \
    }                                                                                      \
  } while (0)

// Kernel launch on h->launch. Events ride on the dispatch itself (hipExtLaunchKernelGGL) instead of marker packets around it:
// a pending profiling bracket (RcProfScope with ride = true) contributes its start / stop events, otherwise a pending
// h->launch_stop (the panel chain's dependency events) is used as the stop event. On the panel chain every packet between two
// dependent kernels of a queue costs several microseconds.
#define RC_LAUNCH(kernel, grid, block, lds, ...)                                                              \
  do {                                                                                                       \
    if (h->prof_pending >= 0) {                                                                              \
      RcProfEvent& pe_ = h->prof_events[h->prof_pending];                                                    \
      h->prof_pending = -1;                                                                                  \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), h->launch, pe_.start, pe_.stop, 0u, __VA_ARGS__); \
    } else if (h->launch_stop) {                                                                             \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), h->launch, nullptr, h->launch_stop, 0u, __VA_ARGS__); \
      h->launch_stop = nullptr;                                                                              \
    } else {                                                                                                 \
      hipLaunchKernelGGL(kernel, grid, block, lds, h->launch, __VA_ARGS__);                                  \
    }                                                                                                        \
  } while (0)

// Profiling bracket of one kernel launch when profiling is on. ride = true: the launcher issues exactly one RC_LAUNCH inside the
// scope, which carries the two events; otherwise they are recorded around the launch.
struct RcProfScope {
  rcgp_handle_s* h; int idx; hipStream_t st; bool ride;
  RcProfScope(rcgp_handle_s* h_, int cls, double work, bool ride_ = false) : h(h_), idx(-1), st(h_->launch), ride(ride_) {
    if (!h->profiling) return;
    RcProfEvent ev; ev.cls = cls;
    auto take = [&](hipEvent_t* e) {
      if (!h->event_pool.empty()) { *e = h->event_pool.back(); h->event_pool.pop_back(); return true; }
      return hipEventCreate(e) == hipSuccess;
    };
    if (!take(&ev.start) || !take(&ev.stop)) return;
    h->prof_events.push_back(ev);
    idx = (int)h->prof_events.size() - 1;
    h->prof_work[cls] += work;
    if (ride) h->prof_pending = idx;
    else (void)hipEventRecord(ev.start, h->launch);
  }
  ~RcProfScope() {
    if (idx < 0) return;
    if (!ride) (void)hipEventRecord(h->prof_events[idx].stop, st);
    else if (h->prof_pending == idx) {                          // no launch happened inside the scope: an empty bracket
      h->prof_pending = -1;
      (void)hipEventRecord(h->prof_events[idx].start, st);
      (void)hipEventRecord(h->prof_events[idx].stop, st);
    }
  }
};

// ---- gram.hip
int rc_launch_scale(rcgp_handle_s* h);                       // Z = X/ell, sq = -0.5|z|^2
int rc_launch_scale_rows(rcgp_handle_s* h, const double* X, double* Z, double* sq, int64_t rows, int out = 0);   // new points, lengthscales of output `out`
int rc_launch_gram(rcgp_handle_s* h);                        // A lower tiles = var*exp(.) (+noise on diag; identity on padding)
int rc_launch_cross_gram(rcgp_handle_s* h, int64_t n, int64_t np, int out = 0);   // KsT (np x Np) from Zs, Z: n points of output `out`

// ---- gemm.hip (all matrices row-major, dims multiples of 128)
// C[i][j] -= sum_k P[i][k] P[j][k]   lower tiles of an n x n matrix, K = kk
int rc_launch_syrk_lower(rcgp_handle_s* h, double* C, int64_t ldc, const double* P, int64_t ldp, int64_t n, int64_t kk);
// C (m x n) -= Arows (m x kk) * Brows (n x kk)^T ; tiles strictly above the diagonal of C (given the global row/col offsets) skipped
int rc_launch_gemm_nt_sub(rcgp_handle_s* h, double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                          int64_t m, int64_t n, int64_t kk, int64_t row0, int64_t col0);
// panel solve by blocked substitution: P (m x 128) <- P * L_jj^-T (L_jj at Ljj with row stride ldp, its 16x16 diagonal-block inverses in the
// diagonal blocks of invL), rhs (m) -= P_new * wj (128)
int rc_launch_trsm_subst(rcgp_handle_s* h, double* P, int64_t ldp, const double* Ljj, const double* invL, int64_t m, double* rhs, const double* wj);
// critical step of the fine-grained chain: the tile T (128 x 128, right below L_jj) solved the same way on eight CUs, rhs (128) -= T_new * wj,
// then D (the next diagonal block) -= T_new * T_new^T
int rc_launch_chain_tile(rcgp_handle_s* h, double* T, double* D, int64_t ld, const double* Ljj, const double* invL, double* rhs, const double* wj);
// L^-1 by recursive doubling, level s: T = B * Ainv (lower-tri Ainv) for C-part row tiles [ti0, ti0+nti) of pairs
// [pair0, pair0+npairs); X21 = -Cinv * T for whole pairs
int rc_launch_trtri_T(rcgp_handle_s* h, int64_t s, int pair0, int npairs, int ti0, int nti);
int rc_launch_trtri_X(rcgp_handle_s* h, int64_t s, int pair0, int npairs);
// K^-1 tiles fused with the LML-gradient reduction; partial sums -> h->partial ; returns number of partial rows via *nrows
int rc_launch_grad(rcgp_handle_s* h, int* nrows);
int rc_launch_grad_mo(rcgp_handle_s* h, int* nrows);     // covariant GP: 2M + 2 sums per lower tile
// predict: colsum((Linv * Ks)^2) for np test points -> h->pvar (np)
int rc_launch_predict_var(rcgp_handle_s* h, int64_t np);

// predict_gradient: V (Np x rows) = Linv * D^T stored chunk by chunk (the chunk's rows of D sit in KsT), then C (rows x rows) = V^T V
int rc_launch_linv_rows(rcgp_handle_s* h, int64_t rows_chunk, double* V, int64_t ldv, int64_t col0);
int rc_launch_vtv(rcgp_handle_s* h, int64_t rows_padded, const double* V, double* C, bool per_block = false);

// ---- potrf.hip
int rc_potrf(rcgp_handle_s* h);                              // blocked Cholesky of A in place, w = L^-1 y, logdiag
int rc_launch_diag(rcgp_handle_s* h, int64_t j);             // diagonal block at row/col offset j: factor, log L_ii, 16x16 diagonal-block inverses, w_j
int rc_launch_inv128_batched(rcgp_handle_s* h);              // every 128x128 inverse of the factor's diagonal blocks into the diagonal of Linv, one launch

// ---- solve.hip
int rc_trtri(rcgp_handle_s* h);                              // Linv = L^-1 (allocates Linv / S on first use)
int rc_alpha(rcgp_handle_s* h);                              // alpha = Linv^T w
int rc_lml_value(rcgp_handle_s* h, double* lml);             // from logdiag and w
int rc_sobol_weight_sum(rcgp_handle_s* h, const double* phi, double pre, const double* alpha_host, double* sum);
int rc_sobol_pair(rcgp_handle_s* h, const double* phi_a, double pre_a, const double* alpha_a, double shift_a, const double* phi_b,
                  double pre_b, const double* alpha_b, double shift_b, int n_slices, const int32_t* slices, double* V_host);
int rc_grad_queue(rcgp_handle_s* h, int nrows);              // queue the gradient's final reduction (before rc_lml_value)
int rc_batch_lml_reduce(rcgp_handle_s* h);                   // batched evaluation: every unit's result block -> the leader's h->bres_d
int rc_grad_finish(rcgp_handle_s* h, double* grad);          // host: gradient from the pinned result block (after rc_lml_value)
int rc_grad_queue_mo(rcgp_handle_s* h);
int rc_grad_finish_mo(rcgp_handle_s* h, double* g_ell, double* g_F, double* g_S);

// ---- sobol.hip
int rc_sobol(rcgp_handle_s* h, const double* ell_j, double var_j, const double* alpha_j_host, int n_slices, const int32_t* slices,
             double* V_host);

int rc_sobol_error_terms(rcgp_handle_s* h, const double* ell_a, double var_a, const double* alpha_a_host, int n_slices,
                         const int32_t* slices, double* phi_d, double* psi_d, double* phi_m, double* psi_m, int out_a = -1, int out_b = -1);

// ---- util
extern long long g_rc_stat[4];                               // process-wide counts (rcgp_stat): factorisations, inversions, gradient launches (units), batched calls
int rc_ensure_pred(rcgp_handle_s* h);                        // predict / psi scratch (KsT, pvar, ...)
int rc_ensure_partial(rcgp_handle_s* h, size_t elems);
int rc_prof_collect(rcgp_handle_s* h);
