// C ABI of librcgp.so (include/rcgp.h). Host-side orchestration only; kernels live in the sibling .hip files.
#include "../../include/rcgp.h"
#include "common.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>

#define RC_API extern "C" __attribute__((visibility("default")))
#define RC_CHECK_H(h)            \
  if (!(h)) return -1;           \
  if (hipSetDevice((h)->device) != hipSuccess) { (h)->err = "hipSetDevice failed"; return -3; }

static const int64_t PRED_CAP = 4096;

RC_API int rcgp_version(void) { return 100; }

long long g_rc_stat[4] = {0, 0, 0, 0};
RC_API int64_t rcgp_stat(int which) { return (which >= 0 && which < 4) ? (int64_t)g_rc_stat[which] : -1; }

RC_API int rcgp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE set of five streams per device, created by the first rcgp_create on that device and shared by every handle of the process on
// it until process exit -- whatever the environment says later. Every stream is a hardware queue that the runtime keeps anyway, and the
// multi-stream Cholesky slows down by a quarter once a few more queues than its own exist (three handles with streams of their own:
// 34 -> 43 ms at C2; idle queues created BETWEEN the library's: 33 -> 75 ms; DESIGN.md Appendix A.2, "stream placement"). Handles are used
// one call at a time by their owner, so sharing the streams only orders the work of different handles of one device behind each
// other. Creation order is fixed (main, chain, column work, far updates, bulk): with it the two streams that carry long K = NB
// kernels share a dispatch pipe and the chain's three have one each, the best of the 20 orders measured in round 2.
// Teardown joins ALL streams of the set before anything is released (round 1 did not, and stalled).
// ---------------------------------------------------------------------------------------------------------------------
struct RcDeviceStreams {
  hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr, stream5 = nullptr, stream6 = nullptr;
  int refs = 0;
};
static std::mutex g_streams_mutex;
static std::map<int, RcDeviceStreams> g_streams;      // by device

// At process exit the sets are destroyed by this handler, registered with the first set: it runs before the HIP runtime (and a
// profiler's intercept layer) are torn down, which handlers registered at load time outlive -- streams left to the runtime's own
// teardown crashed rocprofv3 at exit.
static void destroy_stream_sets() {
  std::lock_guard<std::mutex> lock(g_streams_mutex);
  for (auto& kv : g_streams) {
    RcDeviceStreams& ds = kv.second;
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    hipStream_t* all[] = {&ds.stream6, &ds.stream5, &ds.stream3, &ds.stream2, &ds.stream};
    for (auto sp : all)                                   // every queue idle before the first one goes
      if (*sp) (void)hipStreamSynchronize(*sp);
    for (auto sp : all)
      if (*sp) { (void)hipStreamDestroy(*sp); *sp = nullptr; }
  }
}

// Join every stream a handle can have work on. The multi-stream Cholesky ends with the main stream waiting for its side streams,
// so after a call that returned normally the main stream alone would do; a call that failed half-way (a HIP error between two
// launches) leaves the side streams running on their own, and the handle's buffers and events must outlive that work.
static void join_streams(rcgp_handle_s* h) {
  hipStream_t all[] = {h->stream2, h->stream5, h->stream6, h->stream3, h->stream};
  for (auto s : all)
    if (s) (void)hipStreamSynchronize(s);
}

static void release_streams(rcgp_handle_s* h) {
  if (!h->streams_acquired) return;
  std::lock_guard<std::mutex> lock(g_streams_mutex);
  --g_streams[h->device].refs;                         // bookkeeping only: the set stays
  h->stream = h->stream2 = h->stream3 = h->stream5 = h->stream6 = nullptr;
  h->streams_acquired = false;
}

static void free_all(rcgp_handle_s* h) {
  double** bufs[] = {&h->X, &h->Z, &h->sq, &h->y, &h->w, &h->alpha, &h->A, &h->Linv, &h->S, &h->invdiag, &h->logdiag, &h->partial,
                     &h->scal, &h->ell_d, &h->Xs, &h->Zs, &h->sqs, &h->KsT, &h->pmean, &h->pvar, &h->sob, &h->gV, &h->gC};
  for (auto b : bufs)
    if (*b) { hipFree(*b); *b = nullptr; }
  h->info = nullptr;                                       // (inside scal)
  h->FS_d = nullptr;                                       // (inside ell_d's allocation)
  if (h->bres_d) { hipFree(h->bres_d); h->bres_d = nullptr; }
  if (h->bres_pin) { (void)hipHostFree(h->bres_pin); h->bres_pin = nullptr; }
  if (h->pin) { (void)hipHostFree(h->pin); h->pin = nullptr; }
  if (h->ev_hyper) { (void)hipEventDestroy(h->ev_hyper); h->ev_hyper = nullptr; }
  for (auto& ev : h->prof_events) { (void)hipEventDestroy(ev.start); (void)hipEventDestroy(ev.stop); }
  h->prof_events.clear();
  for (auto& e : h->event_pool) (void)hipEventDestroy(e);
  h->event_pool.clear();
  for (auto& e : h->la_events) (void)hipEventDestroy(e);
  h->la_events.clear();
  release_streams(h);
}

static std::string g_create_error;

RC_API const char* rcgp_last_error(rcgp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int upload_padded(rcgp_handle_s* h, double* dst, const double* src, int64_t rows, int64_t rows_padded, int cols) {
  RC_HIP(hipMemsetAsync(dst, 0, (size_t)rows_padded * cols * sizeof(double), h->stream));
  RC_HIP(hipMemcpyAsync(dst, src, (size_t)rows * cols * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// Targets: y[N] for one output; Y (N x L, row-major, as the reference's Fold.Y) for a covariant GP, stored output-major
// (gpf/models.py:120: reshape(transpose(Y), [-1, 1])), every block padded with zeros to Nb rows.
static int upload_targets(rcgp_handle_s* h, const double* Y) {
  if (h->L == 1) return upload_padded(h, h->y, Y, h->N, h->Np, 1);
  std::vector<double> col((size_t)h->Np, 0.0);
  for (int l = 0; l < h->L; ++l)
    for (int64_t n = 0; n < h->N; ++n) col[(size_t)l * h->Nb + n] = Y[n * h->L + l];
  RC_HIP(hipMemcpyAsync(h->y, col.data(), col.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

static int create_streams(rcgp_handle_s* h, RcDeviceStreams& ds) {
  int lo = 0, hi = 0;
  RC_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));            // hi is the numerically lowest = highest priority
  RC_HIP(hipStreamCreateWithFlags(&ds.stream, hipStreamNonBlocking));
  RC_HIP(hipStreamCreateWithPriority(&ds.stream2, hipStreamNonBlocking, hi));
  RC_HIP(hipStreamCreateWithPriority(&ds.stream5, hipStreamNonBlocking, hi));
  RC_HIP(hipStreamCreateWithPriority(&ds.stream6, hipStreamNonBlocking, hi));
  RC_HIP(hipStreamCreateWithFlags(&ds.stream3, hipStreamNonBlocking));
  return 0;
}

// The schedule's run-time knobs: environment variables, read ONCE per process (at the first rcgp_create) -- a value changed later is
// ignored with a message, so that two handles of one process can never run different schedules on the shared streams.
struct RcKnobs {
  bool read = false;
  bool lookahead = true, fine = true;
  int64_t nb = RC_NB_OUTER;
  int depth = 2, ext = 4;
  int tail = RC_TAIL_BLOCKS, lean = RC_LEAN_BLOCKS, farg = RC_FAR_GROUP;
  bool nb_given = false, tail_given = false;
  int64_t half_tiles = RC_TRTRI_HALF_TILES;
  std::string seen;                                    // the values as first read, to recognise a later change
};
static RcKnobs g_knobs;

static std::string knob_string() {
  std::string out;
  for (const char* name : {"RCGP_LOOKAHEAD", "RCGP_FINE", "RCGP_NB", "RCGP_DEPTH", "RCGP_EXT", "RCGP_TAIL", "RCGP_LEAN", "RCGP_FARG", "RCGP_HALF_TILES"}) {
    const char* e = getenv(name);
    out += std::string(name) + "=" + (e ? e : "") + ";";
  }
  return out;
}

static void read_knobs_once() {                          // (under g_streams_mutex)
  const std::string now = knob_string();
  if (g_knobs.read) {
    if (now != g_knobs.seen) {
      static bool warned = false;
      if (!warned) fprintf(stderr, "[rcgp] RCGP_* knobs changed after the first rcgp_create of this process: ignored (%s in force)\n", g_knobs.seen.c_str());
      warned = true;
    }
    return;
  }
  g_knobs.read = true;
  g_knobs.seen = now;
  if (const char* e = getenv("RCGP_LOOKAHEAD")) g_knobs.lookahead = (e[0] != '0');   // 0 = strictly sequential potrf
  if (const char* e = getenv("RCGP_FINE")) g_knobs.fine = (e[0] != '0');             // 0 = one stream per panel chain (D, T, G in order)
  if (const char* e = getenv("RCGP_EXT")) {
    const int x = atoi(e);
    if (x >= 1 && x <= 16) g_knobs.ext = x;
  }
  if (const char* e = getenv("RCGP_DEPTH")) {
    const int x = atoi(e);
    if (x >= 1 && x <= 64) g_knobs.depth = x;
  }
  if (const char* e = getenv("RCGP_NB")) {
    const int64_t nb = atoll(e);
    if (nb >= 128 && nb <= 4096 && nb % 128 == 0) { g_knobs.nb = nb; g_knobs.nb_given = true; }
  }
  if (const char* e = getenv("RCGP_TAIL")) {           // block columns of the fine-grained tail panel; 0 = no tail
    const int x = atoi(e);
    if (x >= 0 && x <= 4096) { g_knobs.tail = x; g_knobs.tail_given = true; }
  }
  if (const char* e = getenv("RCGP_FARG")) {           // far updates of the panel chain in groups of this many steps
    const int x = atoi(e);
    if (x >= 1 && x <= 8) g_knobs.farg = x;
  }
  if (const char* e = getenv("RCGP_HALF_TILES")) {     // L^-1 launches of at most this many 128^2 tiles run on half tiles
    const int64_t x = atoll(e);
    if (x >= 0) g_knobs.half_tiles = x;
  }
  if (const char* e = getenv("RCGP_LEAN")) {           // chain steps with at most this many blocks below them are lean; 0 = none
    const int x = atoi(e);
    if (x >= 0 && x <= 4096) g_knobs.lean = x;
  }
}

static int create_impl(rcgp_handle_s* h, const double* X, const double* y) {
  const int64_t Np = h->Np;
  const int M = h->M;
  RC_HIP(hipSetDevice(h->device));
  {
    std::lock_guard<std::mutex> lock(g_streams_mutex);
    read_knobs_once();
    RcDeviceStreams& ds = g_streams[h->device];
    if (!ds.stream) {
      static bool registered = false;
      if (!registered) { atexit(destroy_stream_sets); registered = true; }
      int rcs = create_streams(h, ds);
      if (rcs) {                                        // leave no half-built set behind
        hipStream_t* all[] = {&ds.stream6, &ds.stream5, &ds.stream3, &ds.stream2, &ds.stream};
        for (auto sp : all)
          if (*sp) { (void)hipStreamDestroy(*sp); *sp = nullptr; }
        return rcs;
      }
    }
    ++ds.refs;
    h->streams_acquired = true;
    h->stream = ds.stream; h->stream2 = ds.stream2; h->stream3 = ds.stream3; h->stream5 = ds.stream5; h->stream6 = ds.stream6;
    h->lookahead = g_knobs.lookahead;
    h->fine_chain = g_knobs.fine;
    h->nb_outer = g_knobs.nb;
    h->chain_depth = g_knobs.depth;
    h->chain_ext = g_knobs.ext;
    h->tail_blocks = g_knobs.tail;
    if (g_knobs.nb_given) h->batch_nb_outer = g_knobs.nb;          // an explicit knob holds for batches too
    if (g_knobs.tail_given) h->batch_tail_blocks = g_knobs.tail;
    h->lean_blocks = g_knobs.lean;
    h->far_group = g_knobs.farg;
    h->trtri_half_tiles = g_knobs.half_tiles;
  }
  h->launch = h->stream;
  RC_HIP(hipMalloc(&h->X, (size_t)Np * M * sizeof(double)));
  RC_HIP(hipMalloc(&h->Z, (size_t)Np * M * sizeof(double)));
  RC_HIP(hipMalloc(&h->sq, (size_t)Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->y, (size_t)Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->w, (size_t)Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->alpha, (size_t)Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->A, (size_t)Np * Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->invdiag, (size_t)Np * 128 * sizeof(double)));
  RC_HIP(hipMalloc(&h->logdiag, (size_t)Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->scal, RC_SCAL_ELEMS * sizeof(double)));
  RC_HIP(hipMemsetAsync(h->scal, 0, RC_SCAL_ELEMS * sizeof(double), h->stream));
  h->info = reinterpret_cast<int*>(h->scal + RC_SCAL_INFO);
  const size_t n_hyper = (size_t)h->L * M + (size_t)2 * h->L * h->L;
  RC_HIP(hipMalloc(&h->ell_d, n_hyper * sizeof(double)));                  // ell, then F and Sigma: one upload per hyper-parameter change
  h->FS_d = h->ell_d + (size_t)h->L * M;
  h->pin_result = (n_hyper + 7) / 8 * 8;
  h->pin_elems = h->pin_result + RC_SCAL_ELEMS + (size_t)(h->L * (h->L + 1) / 2) * (2 * M + 2);
  RC_HIP(hipHostMalloc(&h->pin, h->pin_elems * sizeof(double), hipHostMallocDefault));
  RC_HIP(hipEventCreateWithFlags(&h->ev_hyper, hipEventDisableTiming));
  int rc;
  for (int l = 0; l < h->L; ++l)                                   // the same inputs under every output block
    if ((rc = upload_padded(h, h->X + (size_t)l * h->Nb * M, X, h->N, h->Nb, M))) return rc;
  return upload_targets(h, y);
}

RC_API int rcgp_create_mo(rcgp_handle* out, int device, int64_t N, int M, int L, const double* X, const double* Y) {
  if (!out) return -1;
  *out = nullptr;
  if (N < 1 || M < 1 || M > RC_MAX_M_WIDE || L < 1 || L > RC_MAX_L || !X || !Y) {
    g_create_error = "rcgp_create: bad argument (need N>=1, 1<=M<=256, 1<=L<=64, X, y)";
    return -2;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "rcgp_create: no HIP device available"; return -4; }
  if (device < 0 || device >= ndev) { g_create_error = "rcgp_create: device index out of range"; return -2; }
  rcgp_handle_s* h = new rcgp_handle_s();
  h->device = device;
  h->N = N;
  h->L = L;
  h->Nb = ((N + RC_TILE - 1) / RC_TILE) * RC_TILE;
  h->Np = h->Nb * L;
  h->M = M;
  h->ell.assign((size_t)L * M, 1.0);
  h->Fm.assign((size_t)L * L, 0.0);
  h->Sm.assign((size_t)L * L, 0.0);
  int rc = create_impl(h, X, Y);
  if (rc) {
    g_create_error = h->err;
    free_all(h);
    delete h;
    return rc;
  }
  *out = h;
  return 0;
}

RC_API int rcgp_create(rcgp_handle* out, int device, int64_t N, int M, const double* X, const double* y) {
  return rcgp_create_mo(out, device, N, M, 1, X, y);
}

RC_API int rcgp_destroy(rcgp_handle h) {
  if (!h) return -1;
  hipSetDevice(h->device);
  join_streams(h);                                         // all six, not just the main one: nothing of this handle is in flight below
  free_all(h);
  delete h;
  return 0;
}

RC_API int rcgp_set_y(rcgp_handle h, const double* y) {
  RC_CHECK_H(h);
  if (!y) { h->err = "rcgp_set_y: null y"; return -2; }
  int rc = upload_targets(h, y);
  h->factored = h->inverted = false;
  return rc;
}

// One asynchronous copy from pinned memory, no stream synchronisation: the evaluation that follows is ordered behind it on the main
// stream. The staging block is rewritten only after the previous upload has left it (an event that has long completed by then).
static int upload_hyper(rcgp_handle_s* h) {                     // (the values are noted; they travel with the next call that needs them)
  h->hyper_set = true;
  h->hyper_dirty = true;
  h->factored = h->inverted = h->gram_fresh = false;
  return 0;
}

static int flush_hyper(rcgp_handle_s* h) {
  if (!h->hyper_dirty) return 0;
  const size_t LL = (size_t)h->L * h->L, LM = h->ell.size();
  if (h->hyper_in_flight) { RC_HIP(hipEventSynchronize(h->ev_hyper)); h->hyper_in_flight = false; }
  memcpy(h->pin, h->ell.data(), LM * sizeof(double));
  memcpy(h->pin + LM, h->Fm.data(), LL * sizeof(double));
  memcpy(h->pin + LM + LL, h->Sm.data(), LL * sizeof(double));
  RC_HIP(hipMemcpyAsync(h->ell_d, h->pin, (LM + 2 * LL) * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipEventRecord(h->ev_hyper, h->stream));
  h->hyper_in_flight = true;
  h->hyper_dirty = false;
  return 0;
}

RC_API int rcgp_set_hyper_mo(rcgp_handle h, const double* ell, const double* F, const double* Sigma) {
  RC_CHECK_H(h);
  if (!ell || !F || !Sigma) { h->err = "rcgp_set_hyper_mo: null argument"; return -2; }
  const int L = h->L, M = h->M;
  for (int i = 0; i < L * M; ++i)
    if (!(ell[i] > 0.0) || !isfinite(ell[i])) { h->err = "rcgp_set_hyper_mo: lengthscales must be positive and finite"; return -2; }
  for (int l = 0; l < L; ++l) {
    if (!(F[l * L + l] > 0.0) || !(Sigma[l * L + l] >= 0.0)) { h->err = "rcgp_set_hyper_mo: need diag(F) > 0 and diag(Sigma) >= 0"; return -2; }
    for (int j = 0; j < L; ++j)
      if (!isfinite(F[l * L + j]) || !isfinite(Sigma[l * L + j]) || F[l * L + j] != F[j * L + l] || Sigma[l * L + j] != Sigma[j * L + l]) {
        h->err = "rcgp_set_hyper_mo: F and Sigma must be finite and symmetric";
        return -2;
      }
  }
  const size_t LL = (size_t)L * L;
  if (h->hyper_set && memcmp(h->ell.data(), ell, h->ell.size() * sizeof(double)) == 0 && memcmp(h->Fm.data(), F, LL * sizeof(double)) == 0 &&
      memcmp(h->Sm.data(), Sigma, LL * sizeof(double)) == 0)
    return 0;                                    // unchanged: keep the factor (see rcgp_set_hyper)
  h->ell.assign(ell, ell + (size_t)L * M);
  h->Fm.assign(F, F + LL);
  h->Sm.assign(Sigma, Sigma + LL);
  h->var = F[0];
  h->noise = Sigma[0];
  return upload_hyper(h);
}

#define RC_SINGLE_OUTPUT(h, name) \
  if ((h)->L != 1) { (h)->err = name ": single-output entry point called on a covariant (L > 1) handle"; return -2; }

RC_API int rcgp_set_hyper(rcgp_handle h, const double* ell, double variance, double noise) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_set_hyper");
  if (!ell) { h->err = "rcgp_set_hyper: null ell"; return -2; }
  for (int m = 0; m < h->M; ++m)
    if (!(ell[m] > 0.0) || !isfinite(ell[m])) { h->err = "rcgp_set_hyper: lengthscales must be positive and finite"; return -2; }
  if (!(variance > 0.0) || !isfinite(variance) || !(noise >= 0.0) || !isfinite(noise)) {
    h->err = "rcgp_set_hyper: variance must be > 0 and noise >= 0";
    return -2;
  }
  if (h->hyper_set && h->var == variance && h->noise == noise && memcmp(h->ell.data(), ell, (size_t)h->M * sizeof(double)) == 0)
    return 0;                                    // unchanged: the Gram matrix, its factor and L^-1 stay valid (the fit driver sets
                                                 // the optimum again after the last evaluation, which is normally at that point)
  h->ell.assign(ell, ell + h->M);
  h->var = variance;
  h->noise = noise;
  h->Fm[0] = variance;
  h->Sm[0] = noise;
  return upload_hyper(h);
}

static int need_hyper(rcgp_handle_s* h) {
  if (!h->hyper_set) { h->err = "hyper-parameters not set: call rcgp_set_hyper first"; return -5; }
  return 0;
}

static int do_gram(rcgp_handle_s* h) {
  int rc;
  if ((rc = flush_hyper(h))) return rc;
  if ((rc = rc_launch_scale(h))) return rc;
  if ((rc = rc_launch_gram(h))) return rc;
  h->factored = h->inverted = false;
  h->gram_fresh = true;
  return 0;
}

static int ensure_factor(rcgp_handle_s* h, bool want_inverse) {
  int rc;
  if ((rc = need_hyper(h))) return rc;
  if (!h->factored) {
    if ((rc = do_gram(h))) return rc;
    if ((rc = rc_potrf(h))) return rc;
  }
  if (want_inverse && !h->inverted) {
    if ((rc = rc_trtri(h))) return rc;
    if ((rc = rc_alpha(h))) return rc;
    h->inverted = true;
  }
  return 0;
}

RC_API int rcgp_stage_gram(rcgp_handle h) {
  RC_CHECK_H(h);
  int rc;
  if ((rc = need_hyper(h))) return rc;
  return do_gram(h);
}

RC_API int rcgp_stage_potrf(rcgp_handle h) {
  RC_CHECK_H(h);
  int rc;
  if ((rc = need_hyper(h))) return rc;
  if (!h->gram_fresh) {                       // A holds a factor (or nothing): factoring it again would silently produce garbage
    h->err = "rcgp_stage_potrf: no fresh Gram matrix (call rcgp_stage_gram first; a factorisation consumes it)";
    return -5;
  }
  return rc_potrf(h);
}

RC_API int rcgp_stage_trtri(rcgp_handle h) {
  RC_CHECK_H(h);
  if (!h->factored) { h->err = "rcgp_stage_trtri: no Cholesky factor"; return -5; }
  int rc;
  if ((rc = rc_trtri(h))) return rc;
  if ((rc = rc_alpha(h))) return rc;
  h->inverted = true;
  return 0;
}

RC_API int rcgp_sync(rcgp_handle h) {
  RC_CHECK_H(h);
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

RC_API int rcgp_lml(rcgp_handle h, double* lml) {
  RC_CHECK_H(h);
  if (!lml) return -2;
  int rc;
  if ((rc = ensure_factor(h, false))) return rc;
  return rc_lml_value(h, lml);
}

RC_API int rcgp_lml_grad_mo(rcgp_handle h, double* lml, double* g_ell, double* g_F, double* g_Sigma) {
  RC_CHECK_H(h);
  if (!lml || !g_ell || !g_F || !g_Sigma) return -2;
  int rc;
  if ((rc = ensure_factor(h, true))) return rc;
  int nrows = 0;
  if ((rc = rc_launch_grad_mo(h, &nrows))) return rc;
  if ((rc = rc_grad_queue_mo(h))) return rc;
  if ((rc = rc_lml_value(h, lml))) return rc;            // the one host synchronisation of an evaluation
  return rc_grad_finish_mo(h, g_ell, g_F, g_Sigma);
}

RC_API int rcgp_lml_grad(rcgp_handle h, double* lml, double* grad) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_lml_grad");
  if (!lml || !grad) return -2;
  int rc;
  if ((rc = ensure_factor(h, true))) return rc;
  int nrows = 0;
  if ((rc = rc_launch_grad(h, &nrows))) return rc;          // queued behind the factorisation
  if ((rc = rc_grad_queue(h, nrows))) return rc;
  if ((rc = rc_lml_value(h, lml))) return rc;               // the one host synchronisation of an evaluation
  return rc_grad_finish(h, grad);
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched units: ONE schedule for up to RC_MAX_BATCH handles of one device with equal padded size and M (common.h, "Batched units").
// The reference fits its L outputs and its K folds one after the other (gpr/models.py:340-342, 360-361; user/run.py:60-61); a single
// factorisation of the sizes it is run at (N <= 10^4, benchmark_script.py:35-40) is latency-bound on this chip -- a chain of N/128 steps
// with most CUs idle -- so the units' chains run side by side in the same launches here.
// ---------------------------------------------------------------------------------------------------------------------
struct RcBatchScope {                                       // h leads `units` for the launches inside the scope
  rcgp_handle_s* h;
  RcBatchScope(rcgp_handle_s* lead, const std::vector<rcgp_handle_s*>& units) : h(lead) {
    h->nb = (int)units.size();
    for (int u = 0; u < RC_MAX_BATCH; ++u) h->bh[u] = (u < h->nb) ? units[u] : nullptr;
  }
  ~RcBatchScope() {
    h->nb = 1;
    for (int u = 0; u < RC_MAX_BATCH; ++u) h->bh[u] = nullptr;
  }
};

static int batch_check(int n, rcgp_handle* hs, const char* who) {
  if (n < 1 || !hs || !hs[0]) return -1;
  rcgp_handle_s* h = hs[0];
  if (n > RC_MAX_BATCH) { h->err = std::string(who) + ": at most " + std::to_string(RC_MAX_BATCH) + " units per call"; return -2; }
  for (int u = 0; u < n; ++u) {
    if (!hs[u]) { h->err = std::string(who) + ": null handle"; return -2; }
    if (hs[u]->L != 1) { h->err = std::string(who) + ": single-output handles only"; return -2; }
    if (hs[u]->device != h->device || hs[u]->Np != h->Np || hs[u]->M != h->M) {
      h->err = std::string(who) + ": the units of a batch must share the device, the padded size (ceil(N / 128)) and M";
      return -2;
    }
    for (int v = 0; v < u; ++v)
      if (hs[v] == hs[u]) { h->err = std::string(who) + ": the same handle twice"; return -2; }
    if (!hs[u]->hyper_set) { h->err = hs[u]->err = "hyper-parameters not set: call rcgp_set_hyper first"; return -5; }
  }
  if (hipSetDevice(h->device) != hipSuccess) { h->err = "hipSetDevice failed"; return -3; }
  return 0;
}

static int ensure_batch_buffers(rcgp_handle_s* h) {
  if (!h->bres_d) {
    RC_HIP(hipMalloc(&h->bres_d, (size_t)2 * RC_MAX_BATCH * RC_SCAL_ELEMS * sizeof(double)));
    RC_HIP(hipHostMalloc(&h->bres_pin, (size_t)2 * RC_MAX_BATCH * RC_SCAL_ELEMS * sizeof(double), hipHostMallocDefault));
  }
  return 0;
}

__global__ void k_scatter_hyper(RcBP<double> dst, const double* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst.p[blockIdx.z][i] = src[(size_t)blockIdx.z * RC_SCAL_ELEMS + i];
}

// The new hyper-parameters of every unit of a batched call in ONE copy from pinned memory and one scatter kernel (a copy per unit
// otherwise: with 16 small units they were a third of the operations of an evaluation). Staging: the second half of the leader's
// batch buffers; it is repacked only after the previous upload has left it.
static int flush_hyper_batch(rcgp_handle_s* h, const std::vector<rcgp_handle_s*>& units) {
  std::vector<rcgp_handle_s*> dirty;
  for (auto hu : units)
    if (hu->hyper_dirty) dirty.push_back(hu);
  if (dirty.size() <= 1) {
    for (auto hu : dirty) {
      int rc = flush_hyper(hu);
      if (rc) { h->err = hu->err; return rc; }
    }
    return 0;
  }
  int rc;
  if ((rc = ensure_batch_buffers(h))) return rc;
  const int M = h->M, n = M + 2;                                   // single-output units: ell[M], then the kernel and the likelihood variance
  double* stage_pin = h->bres_pin + (size_t)RC_MAX_BATCH * RC_SCAL_ELEMS;
  double* stage_d = h->bres_d + (size_t)RC_MAX_BATCH * RC_SCAL_ELEMS;
  if (h->hyper_in_flight) { RC_HIP(hipEventSynchronize(h->ev_hyper)); h->hyper_in_flight = false; }
  RcBP<double> dst;
  for (int k = 0; k < RC_MAX_BATCH; ++k) dst.p[k] = nullptr;
  for (size_t k = 0; k < dirty.size(); ++k) {
    double* row = stage_pin + k * RC_SCAL_ELEMS;
    memcpy(row, dirty[k]->ell.data(), (size_t)M * sizeof(double));
    row[M] = dirty[k]->Fm[0];
    row[M + 1] = dirty[k]->Sm[0];
    dst.p[k] = dirty[k]->ell_d;                                    // (ell, then F and Sigma: one allocation per handle)
  }
  RC_HIP(hipMemcpyAsync(stage_d, stage_pin, dirty.size() * RC_SCAL_ELEMS * sizeof(double), hipMemcpyHostToDevice, h->stream));
  RC_HIP(hipEventRecord(h->ev_hyper, h->stream));
  h->hyper_in_flight = true;
  hipLaunchKernelGGL(k_scatter_hyper, dim3((unsigned)((n + 255) / 256), 1, (unsigned)dirty.size()), dim3(256), 0, h->stream, dst, (const double*)stage_d, n);
  RC_HIP(hipGetLastError());
  for (auto hu : dirty) hu->hyper_dirty = false;
  return 0;
}

// Gram + factorisation for the units that have no factor, L^-1 + alpha for those that lack them: each stage one batched schedule.
static int batch_ensure(int n, rcgp_handle* hs, bool want_inverse) {
  int rc;
  std::vector<rcgp_handle_s*> todo;
  for (int u = 0; u < n; ++u)
    if (!hs[u]->factored) todo.push_back(hs[u]);
  if (!todo.empty()) {
    rcgp_handle_s* h = todo[0];
    if ((rc = flush_hyper_batch(h, todo))) { hs[0]->err = h->err; return rc; }
    RcBatchScope scope(h, todo);
    if ((rc = rc_launch_scale(h)) || (rc = rc_launch_gram(h))) { hs[0]->err = h->err; return rc; }
    for (auto hu : todo) { hu->factored = hu->inverted = false; hu->gram_fresh = true; }
    if ((rc = rc_potrf(h))) { hs[0]->err = h->err; return rc; }
  }
  if (!want_inverse) return 0;
  todo.clear();
  for (int u = 0; u < n; ++u)
    if (!hs[u]->inverted) todo.push_back(hs[u]);
  if (!todo.empty()) {
    rcgp_handle_s* h = todo[0];
    RcBatchScope scope(h, todo);
    if ((rc = rc_trtri(h)) || (rc = rc_alpha(h))) { hs[0]->err = h->err; return rc; }
    for (auto hu : todo) hu->inverted = true;
  }
  return 0;
}

// The numbers of a batched evaluation: every unit's LML sums, gradient sums and status word gathered into the leader's result table,
// ONE copy into pinned memory, ONE synchronisation. status[u] = 0 or the leading minor that failed (that unit's factor is dropped).
static int batch_values(int n, rcgp_handle* hs, double* lml, double* grad, int* status) {
  rcgp_handle_s* h = hs[0];
  int rc;
  if ((rc = ensure_batch_buffers(h))) return rc;
  std::vector<rcgp_handle_s*> units(hs, hs + n);
  {
    RcBatchScope scope(h, units);
    if ((rc = rc_batch_lml_reduce(h))) return rc;
  }
  RC_HIP(hipMemcpyAsync(h->bres_pin, h->bres_d, (size_t)n * RC_SCAL_ELEMS * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  if (h->profiling) rc_prof_collect(h);
  const int M = h->M;
  for (int u = 0; u < n; ++u) {
    const double* host = h->bres_pin + (size_t)u * RC_SCAL_ELEMS;
    int info = 0;
    memcpy(&info, host + RC_SCAL_INFO, sizeof(int));
    status[u] = info;
    if (info != 0) {
      hs[u]->err = "matrix is not positive definite: leading minor " + std::to_string(info);
      hs[u]->factored = hs[u]->inverted = false;
      if (lml) lml[u] = NAN;
      if (grad) for (int m = 0; m < M + 2; ++m) grad[(size_t)u * (M + 2) + m] = NAN;
      continue;
    }
    if (lml) lml[u] = -0.5 * host[0] - host[1] - 0.5 * (double)hs[u]->N * 1.8378770664093454836;   // log(2 pi)
    if (grad) {
      double* g = grad + (size_t)u * (M + 2);
      for (int m = 0; m < M; ++m) g[m] = 0.5 * host[8 + m] / hs[u]->ell[m];     // dK/dell_m = K (z_im - z_jm)^2 / ell_m
      g[M] = 0.5 * host[8 + M] / hs[u]->var;
      g[M + 1] = 0.5 * host[8 + M + 1];
    }
  }
  return 0;
}

RC_API int rcgp_lml_grad_batch(int n, rcgp_handle* hs, double* lml, double* grad, int* status) {
  int rc;
  if ((rc = batch_check(n, hs, "rcgp_lml_grad_batch"))) return rc;
  if (!lml || !grad || !status) { hs[0]->err = "rcgp_lml_grad_batch: null argument"; return -2; }
  if ((rc = batch_ensure(n, hs, true))) return rc;
  g_rc_stat[3] += 1;
  rcgp_handle_s* h = hs[0];
  std::vector<rcgp_handle_s*> units(hs, hs + n);
  {
    RcBatchScope scope(h, units);
    int nrows = 0;
    if ((rc = rc_launch_grad(h, &nrows)) || (rc = rc_grad_queue(h, nrows))) return rc;
  }
  return batch_values(n, hs, lml, grad, status);
}

RC_API int rcgp_factor_batch(int n, rcgp_handle* hs, int* status) {
  int rc;
  if ((rc = batch_check(n, hs, "rcgp_factor_batch"))) return rc;
  if (!status) { hs[0]->err = "rcgp_factor_batch: null argument"; return -2; }
  if ((rc = batch_ensure(n, hs, true))) return rc;
  return batch_values(n, hs, nullptr, nullptr, status);
}

RC_API int rcgp_stage_batch(int stage, int n, rcgp_handle* hs) {
  int rc;
  if ((rc = batch_check(n, hs, "rcgp_stage_batch"))) return rc;
  rcgp_handle_s* h = hs[0];
  std::vector<rcgp_handle_s*> units(hs, hs + n);
  RcBatchScope scope(h, units);
  switch (stage) {
    case 0:
      if ((rc = flush_hyper_batch(h, units))) return rc;
      if ((rc = rc_launch_scale(h)) || (rc = rc_launch_gram(h))) return rc;
      for (auto hu : units) { hu->factored = hu->inverted = false; hu->gram_fresh = true; }
      return 0;
    case 1:
      for (auto hu : units)
        if (!hu->gram_fresh) { h->err = "rcgp_stage_batch: no fresh Gram matrix on every unit (stage 0 first; a factorisation consumes it)"; return -5; }
      return rc_potrf(h);
    case 2:
      for (auto hu : units)
        if (!hu->factored) { h->err = "rcgp_stage_batch: a unit has no Cholesky factor"; return -5; }
      if ((rc = rc_trtri(h)) || (rc = rc_alpha(h))) return rc;
      for (auto hu : units) hu->inverted = true;
      return 0;
    default:
      h->err = "rcgp_stage_batch: stage must be 0 (Gram), 1 (Cholesky) or 2 (L^-1 + alpha)";
      return -2;
  }
}

RC_API int rcgp_factor(rcgp_handle h) {
  RC_CHECK_H(h);
  int rc;
  if ((rc = ensure_factor(h, true))) return rc;
  double lml;
  return rc_lml_value(h, &lml);          // synchronises and surfaces a non-PD failure
}

RC_API int rcgp_get_k_inv_y(rcgp_handle h, double* out) {
  RC_CHECK_H(h);
  if (!out) return -2;
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  RC_HIP(hipMemcpy2DAsync(out, (size_t)h->N * sizeof(double), h->alpha, (size_t)h->Nb * sizeof(double), (size_t)h->N * sizeof(double),
                          (size_t)h->L, hipMemcpyDeviceToHost, h->stream));          // L blocks of N: (L, 1, N) of gpr/models.py:444
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// The (L N) x (L N) system matrix without the padding rows of each output block
static int copy_square(rcgp_handle_s* h, double* out) {
  const size_t NT = (size_t)h->N * h->L;
  for (int bi = 0; bi < h->L; ++bi)
    for (int bj = 0; bj < h->L; ++bj)
      RC_HIP(hipMemcpy2DAsync(out + ((size_t)bi * h->N) * NT + (size_t)bj * h->N, NT * sizeof(double),
                              h->A + ((size_t)bi * h->Nb) * h->Np + (size_t)bj * h->Nb, (size_t)h->Np * sizeof(double),
                              (size_t)h->N * sizeof(double), (size_t)h->N, hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

RC_API int rcgp_get_k_cho(rcgp_handle h, double* out) {
  RC_CHECK_H(h);
  if (!out) return -2;
  int rc;
  if ((rc = ensure_factor(h, false))) return rc;
  double lml;
  if ((rc = rc_lml_value(h, &lml))) return rc;
  if ((rc = copy_square(h, out))) return rc;
  const int64_t N = h->N * h->L;
  for (int64_t i = 0; i < N; ++i) memset(out + i * N + i + 1, 0, (size_t)(N - i - 1) * sizeof(double));
  return 0;
}

RC_API int rcgp_get_gram(rcgp_handle h, double* out) {
  RC_CHECK_H(h);
  if (!out) return -2;
  int rc;
  if ((rc = need_hyper(h))) return rc;
  if ((rc = do_gram(h))) return rc;
  if ((rc = copy_square(h, out))) return rc;
  const int64_t N = h->N * h->L;
  for (int64_t i = 0; i < N; ++i)
    for (int64_t j = i + 1; j < N; ++j) out[i * N + j] = out[j * N + i];
  return 0;
}

// mean[j] = sum_k KsT[j][k] alpha[k] : one block per test point, fixed-order tree
__global__ void __launch_bounds__(256) k_rowdot(const double* __restrict__ KsT, int64_t ld, const double* __restrict__ alpha, int64_t n,
                                                double* __restrict__ out) {
  __shared__ double sm[256];
  const double* row = KsT + (int64_t)blockIdx.x * ld;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += 256) s = fma(row[k], alpha[k], s);
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

int rc_ensure_pred(rcgp_handle_s* h) {
  if (h->KsT) return 0;
  const int M = h->M;
  const int64_t Np = h->Np;
  RC_HIP(hipMalloc(&h->Xs, (size_t)PRED_CAP * M * sizeof(double)));
  RC_HIP(hipMalloc(&h->Zs, (size_t)PRED_CAP * M * sizeof(double)));
  RC_HIP(hipMalloc(&h->sqs, (size_t)PRED_CAP * sizeof(double)));
  RC_HIP(hipMalloc(&h->KsT, (size_t)PRED_CAP * Np * sizeof(double)));
  RC_HIP(hipMalloc(&h->pmean, (size_t)PRED_CAP * sizeof(double)));
  RC_HIP(hipMalloc(&h->pvar, (size_t)PRED_CAP * sizeof(double)));
  h->pred_cap = PRED_CAP;
  h->pts_cap = PRED_CAP;
  return 0;
}

// Room for `pts` new points in Xs / Zs / sqs (the prediction path itself never needs more than PRED_CAP at a time).
static int ensure_points(rcgp_handle_s* h, int64_t pts) {
  int rc;
  if ((rc = rc_ensure_pred(h))) return rc;
  if (pts <= h->pts_cap) return 0;
  RC_HIP(hipStreamSynchronize(h->stream));
  double** bufs[] = {&h->Xs, &h->Zs, &h->sqs};
  for (auto b : bufs) { RC_HIP(hipFree(*b)); *b = nullptr; }
  h->pts_cap = 0;
  RC_HIP(hipMalloc(&h->Xs, (size_t)pts * h->M * sizeof(double)));
  RC_HIP(hipMalloc(&h->Zs, (size_t)pts * h->M * sizeof(double)));
  RC_HIP(hipMalloc(&h->sqs, (size_t)pts * sizeof(double)));
  h->pts_cap = pts;
  return 0;
}

// Scratch of the gradient GP: V (Np x rows) and `blocks` products V^T V (rows x rows).
static int ensure_gradient_scratch(rcgp_handle_s* h, int64_t rows_padded, int blocks) {
  if (h->g_rows >= rows_padded && h->g_blocks >= blocks) return 0;
  RC_HIP(hipStreamSynchronize(h->stream));
  if (h->gV) { RC_HIP(hipFree(h->gV)); h->gV = nullptr; }
  if (h->gC) { RC_HIP(hipFree(h->gC)); h->gC = nullptr; }
  h->g_rows = 0;
  h->g_blocks = 0;
  RC_HIP(hipMalloc(&h->gV, (size_t)h->Np * rows_padded * sizeof(double)));
  RC_HIP(hipMalloc(&h->gC, (size_t)blocks * rows_padded * rows_padded * sizeof(double)));
  h->g_rows = rows_padded;
  h->g_blocks = blocks;
  return 0;
}

// Posterior of output `out` at n points; results strided by the number of outputs (mean[o * L + out]).
static int predict_output(rcgp_handle_s* h, int64_t n, const double* Xnew, int include_noise, int out, double* mean, double* sd) {
  int rc;
  const int M = h->M, L = h->L;
  const double prior = h->Fm[(size_t)out * L + out], noise = h->Sm[(size_t)out * L + out];
  const int64_t Np = h->Np;
  if ((rc = rc_ensure_pred(h))) return rc;
  std::vector<double> hv(PRED_CAP), hm(PRED_CAP);
  for (int64_t o0 = 0; o0 < n; o0 += PRED_CAP) {
    const int64_t nc = (n - o0 < PRED_CAP) ? n - o0 : PRED_CAP;
    const int64_t ncp = ((nc + 127) / 128) * 128;
    if ((rc = upload_padded(h, h->Xs, Xnew + o0 * M, nc, ncp, M))) return rc;
    if ((rc = rc_launch_scale_rows(h, h->Xs, h->Zs, h->sqs, ncp, out))) return rc;
    if ((rc = rc_launch_cross_gram(h, nc, ncp, out))) return rc;
    {
      RcProfScope ps(h, RC_K_MISC, 0.0);
      hipLaunchKernelGGL(k_rowdot, dim3((unsigned)nc), dim3(256), 0, h->stream, h->KsT, Np, h->alpha, Np, h->pmean);
      RC_HIP(hipGetLastError());
    }
    if ((rc = rc_launch_predict_var(h, ncp))) return rc;
    RC_HIP(hipMemcpyAsync(hm.data(), h->pmean, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC_HIP(hipMemcpyAsync(hv.data(), h->pvar, (size_t)nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    RC_HIP(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < nc; ++i) {
      double v = prior - hv[i];                        // k** - |L^-1 k*|^2  (GPflow base_conditional)
      if (include_noise) v += noise;                   // predict_y adds the likelihood variance (its diagonal: gpf/likelihoods.py:83)
      mean[(o0 + i) * L + out] = hm[i];
      sd[(o0 + i) * L + out] = sqrt(v);                // SD, not variance (gpr/models.py:384)
    }
  }
  return 0;
}

RC_API int rcgp_predict_mo(rcgp_handle h, int64_t n, const double* Xnew, int include_noise, double* mean, double* sd) {
  RC_CHECK_H(h);
  if (n < 0 || (n > 0 && (!Xnew || !mean || !sd))) { h->err = "rcgp_predict: bad argument"; return -2; }
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  for (int out = 0; out < h->L; ++out)
    if ((rc = predict_output(h, n, Xnew, include_noise, out, mean, sd))) return rc;
  return 0;
}

RC_API int rcgp_predict(rcgp_handle h, int64_t n, const double* Xnew, int include_noise, double* mean, double* sd) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_predict");
  return rcgp_predict_mo(h, n, Xnew, include_noise, mean, sd);
}

// D[(o*M + m)][n] = d k(X_n, x_o) / d x_om = -(x_om - X_nm) / ell_m^2 * k(X_n, x_o)   (zero on the padding)
__global__ void k_dkernel_rows(const double* __restrict__ Zs, const double* __restrict__ sqs, const double* __restrict__ Z,
                               const double* __restrict__ sq, const double* __restrict__ ell, int64_t n_pts, int64_t N, int64_t Np, int M,
                               double var, double* __restrict__ D, int64_t row0) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = row0 + blockIdx.y;                      // row of the whole derivative matrix; D holds the chunk starting at row0
  if (n >= Np) return;
  const int64_t o = r / M;
  const int m = (int)(r - o * M);
  double v = 0.0;
  if (o < n_pts && n < N) {
    double dot = 0.0;
    for (int mm = 0; mm < M; ++mm) dot = fma(Zs[o * M + mm], Z[n * M + mm], dot);
    const double k = var * exp(sqs[o] + sq[n] + dot);
    v = -(Zs[o * M + m] - Z[n * M + m]) / ell[m] * k;              // (x - X)/ell^2 = (z - Z)/ell
  }
  D[(r - row0) * Np + n] = v;
}

RC_API int rcgp_predict_gradient(rcgp_handle h, int64_t n, const double* Xnew, double* mean, double* cov) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_predict_gradient");
  const int M = h->M;
  if (n < 1 || !Xnew || !mean || !cov) { h->err = "rcgp_predict_gradient: bad argument"; return -2; }
  const int64_t rows = n * M, rows_padded = ((rows + 127) / 128) * 128;
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  const int64_t np = ((n + 127) / 128) * 128;
  if ((rc = ensure_points(h, np))) return rc;
  if ((rc = ensure_gradient_scratch(h, rows_padded, 1))) return rc;
  const int64_t Np = h->Np;
  if ((rc = upload_padded(h, h->Xs, Xnew, n, np, M))) return rc;
  if ((rc = rc_launch_scale_rows(h, h->Xs, h->Zs, h->sqs, np))) return rc;
  // the (n M) derivative rows pass through KsT PRED_CAP at a time (the reference forms them all at once by tape.jacobian,
  // gpr/models.py:407-410): mean chunk by chunk, V = L^-1 D^T column block by column block, then one V^T V over all of them
  for (int64_t r0 = 0; r0 < rows_padded; r0 += PRED_CAP) {
    const int64_t rc_rows = (rows_padded - r0 < PRED_CAP) ? rows_padded - r0 : PRED_CAP;
    const int64_t live = (rows - r0 < rc_rows) ? rows - r0 : rc_rows;
    {
      RcProfScope ps(h, RC_K_MISC, 0.0);
      hipLaunchKernelGGL(k_dkernel_rows, dim3((unsigned)((Np + 255) / 256), (unsigned)rc_rows), dim3(256), 0, h->stream, h->Zs, h->sqs, h->Z, h->sq,
                         h->ell_d, n, h->N, Np, M, h->var, h->KsT, r0);
      RC_HIP(hipGetLastError());
      if (live > 0) {
        hipLaunchKernelGGL(k_rowdot, dim3((unsigned)live), dim3(256), 0, h->stream, h->KsT, Np, h->alpha, Np, h->pmean);
        RC_HIP(hipGetLastError());
      }
    }
    if ((rc = rc_launch_linv_rows(h, rc_rows, h->gV, rows_padded, r0))) return rc;
    if (live > 0) {
      RC_HIP(hipMemcpyAsync(mean + r0, h->pmean, (size_t)live * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      RC_HIP(hipStreamSynchronize(h->stream));              // pmean and KsT are rewritten by the next chunk
    }
  }
  if ((rc = rc_launch_vtv(h, rows_padded, h->gV, h->gC))) return rc;
  RC_HIP(hipMemcpy2DAsync(cov, (size_t)rows * sizeof(double), h->gC, (size_t)rows_padded * sizeof(double), (size_t)rows * sizeof(double),
                          (size_t)rows, hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

// Covariant GP: row r = (l * n + o) * M + m holds d K[(Lb, N), (l, o)] / d x_om = -(u_m - U_m) / ell_lm * F[Lb][l] exp(-1/2 |U - u|^2) with
// U = X_N / ell_Lb and u = x_o / ell_l (Zs holds the L * n scaled points output-major); zero on the padding.
__global__ void k_dkernel_rows_mo(const double* __restrict__ Zs, const double* __restrict__ sqs, const double* __restrict__ Z,
                                  const double* __restrict__ sq, const double* __restrict__ ell, const double* __restrict__ F, int L,
                                  int64_t n_pts, int64_t N, int64_t Nb, int64_t Np, int M, double* __restrict__ D, int64_t row0) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = row0 + blockIdx.y;
  if (a >= Np) return;
  const int64_t p = r / M;                                 // scaled point (l, o)
  const int m = (int)(r - p * M);
  const int64_t l = p / n_pts;
  const int64_t Lb = a / Nb;
  double v = 0.0;
  if (l < L && a - Lb * Nb < N) {
    double dot = 0.0;
    for (int mm = 0; mm < M; ++mm) dot = fma(Zs[p * M + mm], Z[a * M + mm], dot);
    const double k = F[Lb * L + l] * exp(sqs[p] + sq[a] + dot);
    v = -(Zs[p * M + m] - Z[a * M + m]) / ell[l * M + m] * k;
  }
  D[(r - row0) * Np + a] = v;
}

RC_API int rcgp_predict_gradient_mo(rcgp_handle h, int64_t n, const double* Xnew, double* mean, double* cov) {
  RC_CHECK_H(h);
  const int M = h->M, L = h->L;
  if (n < 1 || !Xnew || !mean || !cov) { h->err = "rcgp_predict_gradient_mo: bad argument"; return -2; }
  const int64_t rows = (int64_t)L * n * M, rows_padded = ((rows + 127) / 128) * 128;
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  if ((rc = ensure_points(h, (int64_t)L * n))) return rc;
  if ((rc = ensure_gradient_scratch(h, rows_padded, L))) return rc;
  const int64_t Np = h->Np;
  if ((rc = upload_padded(h, h->Xs, Xnew, n, n, M))) return rc;
  // the L scaled copies of the points, output-major: Zs block l = x / ell_l
  for (int l = 0; l < L; ++l)
    if ((rc = rc_launch_scale_rows(h, h->Xs, h->Zs + (size_t)l * n * M, h->sqs + (size_t)l * n, n, l))) return rc;
  for (int64_t r0 = 0; r0 < rows_padded; r0 += PRED_CAP) {
    const int64_t rc_rows = (rows_padded - r0 < PRED_CAP) ? rows_padded - r0 : PRED_CAP;
    const int64_t live = (rows - r0 < rc_rows) ? rows - r0 : rc_rows;
    {
      RcProfScope ps(h, RC_K_MISC, 0.0);
      hipLaunchKernelGGL(k_dkernel_rows_mo, dim3((unsigned)((Np + 255) / 256), (unsigned)rc_rows), dim3(256), 0, h->stream, h->Zs, h->sqs, h->Z,
                         h->sq, h->ell_d, h->FS_d, L, n, h->N, h->Nb, Np, M, h->KsT, r0);
      RC_HIP(hipGetLastError());
      if (live > 0) {
        hipLaunchKernelGGL(k_rowdot, dim3((unsigned)live), dim3(256), 0, h->stream, h->KsT, Np, h->alpha, Np, h->pmean);
        RC_HIP(hipGetLastError());
      }
    }
    if ((rc = rc_launch_linv_rows(h, rc_rows, h->gV, rows_padded, r0))) return rc;
    if (live > 0) {
      RC_HIP(hipMemcpyAsync(mean + r0, h->pmean, (size_t)live * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      RC_HIP(hipStreamSynchronize(h->stream));
    }
  }
  if ((rc = rc_launch_vtv(h, rows_padded, h->gV, h->gC, true))) return rc;
  for (int b = 0; b < L; ++b)
    RC_HIP(hipMemcpy2DAsync(cov + (size_t)b * rows * rows, (size_t)rows * sizeof(double), h->gC + (size_t)b * rows_padded * rows_padded,
                            (size_t)rows_padded * sizeof(double), (size_t)rows * sizeof(double), (size_t)rows, hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

RC_API int rcgp_sobol_closed(rcgp_handle h, int n_slices, const int32_t* slices, double* V) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_sobol_closed");
  if (n_slices < 0 || (n_slices > 0 && (!slices || !V))) { h->err = "rcgp_sobol_closed: bad argument"; return -2; }
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  return rc_sobol(h, nullptr, 0.0, nullptr, n_slices, slices, V);
}

RC_API int rcgp_sobol_cross(rcgp_handle h, const double* ell_j, double var_j, const double* alpha_j, int n_slices, const int32_t* slices,
                            double* V) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_sobol_cross");
  if (!ell_j || !alpha_j || n_slices < 0 || (n_slices > 0 && (!slices || !V))) { h->err = "rcgp_sobol_cross: bad argument"; return -2; }
  for (int m = 0; m < h->M; ++m)
    if (!(ell_j[m] > 0.0)) { h->err = "rcgp_sobol_cross: lengthscales must be positive"; return -2; }
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  return rc_sobol(h, ell_j, var_j, alpha_j, n_slices, slices, V);
}

RC_API int rcgp_sobol_error_terms(rcgp_handle h, const double* ell_a, double var_a, const double* alpha_a, int n_slices,
                                  const int32_t* slices, double* phi_d, double* psi_d, double* phi_m, double* psi_m) {
  RC_CHECK_H(h);
  RC_SINGLE_OUTPUT(h, "rcgp_sobol_error_terms");
  if (n_slices < 0 || (n_slices > 0 && (!slices || !phi_d || !psi_d || !phi_m || !psi_m)) || ((ell_a == nullptr) != (alpha_a == nullptr))) {
    h->err = "rcgp_sobol_error_terms: bad argument";
    return -2;
  }
  if (ell_a)
    for (int m = 0; m < h->M; ++m)
      if (!(ell_a[m] > 0.0)) { h->err = "rcgp_sobol_error_terms: lengthscales must be positive"; return -2; }
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  return rc_sobol_error_terms(h, ell_a, var_a, alpha_a, n_slices, slices, phi_d, psi_d, phi_m, psi_m);
}

RC_API int rcgp_sobol_error_terms_mo(rcgp_handle h, int out_a, int out_b, int n_slices, const int32_t* slices, double* phi_d, double* psi_d,
                                     double* phi_m, double* psi_m) {
  RC_CHECK_H(h);
  if (out_a < 0 || out_a >= h->L || out_b < 0 || out_b >= h->L || n_slices < 0 ||
      (n_slices > 0 && (!slices || !phi_d || !psi_d || !phi_m || !psi_m))) {
    h->err = "rcgp_sobol_error_terms_mo: bad argument";
    return -2;
  }
  int rc;
  if ((rc = rcgp_factor(h))) return rc;
  return rc_sobol_error_terms(h, nullptr, 0.0, nullptr, n_slices, slices, phi_d, psi_d, phi_m, psi_m, out_a, out_b);
}

static int check_phi(rcgp_handle_s* h, const double* phi, const char* who) {
  for (int m = 0; m < h->M; ++m)
    if (!(phi[m] > 0.0) || !(phi[m] < 1.0)) { h->err = std::string(who) + ": phi must lie in (0, 1)"; return -2; }
  return 0;
}

RC_API int rcgp_sobol_weight_sum(rcgp_handle h, const double* phi, double pre, const double* alpha, double* sum) {
  RC_CHECK_H(h);
  if (!phi || !alpha || !sum) { h->err = "rcgp_sobol_weight_sum: bad argument"; return -2; }
  int rc;
  if ((rc = check_phi(h, phi, "rcgp_sobol_weight_sum"))) return rc;
  return rc_sobol_weight_sum(h, phi, pre, alpha, sum);
}

RC_API int rcgp_sobol_pair(rcgp_handle h, const double* phi_a, double pre_a, const double* alpha_a, double shift_a, const double* phi_b,
                           double pre_b, const double* alpha_b, double shift_b, int n_slices, const int32_t* slices, double* V) {
  RC_CHECK_H(h);
  if (!phi_a || !alpha_a || !phi_b || !alpha_b || n_slices < 0 || (n_slices > 0 && (!slices || !V))) {
    h->err = "rcgp_sobol_pair: bad argument";
    return -2;
  }
  int rc;
  if ((rc = check_phi(h, phi_a, "rcgp_sobol_pair")) || (rc = check_phi(h, phi_b, "rcgp_sobol_pair"))) return rc;
  return rc_sobol_pair(h, phi_a, pre_a, alpha_a, shift_a, phi_b, pre_b, alpha_b, shift_b, n_slices, slices, V);
}

// ---------------------------------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------------------------------
int rc_prof_collect(rcgp_handle_s* h) {
  RC_HIP(hipStreamSynchronize(h->stream));
  for (auto& ev : h->prof_events) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.start, ev.stop) == hipSuccess) {
      h->prof_ms[ev.cls] += (double)ms;
      h->prof_count[ev.cls] += 1;
    }
    h->event_pool.push_back(ev.start);
    h->event_pool.push_back(ev.stop);
  }
  h->prof_events.clear();
  return 0;
}

RC_API int rcgp_set_profiling(rcgp_handle h, int on) {
  RC_CHECK_H(h);
  int rc = rc_prof_collect(h);
  h->profiling = (on != 0);
  return rc;
}

RC_API int rcgp_profile_reset(rcgp_handle h) {
  RC_CHECK_H(h);
  int rc = rc_prof_collect(h);
  for (int c = 0; c < RC_K_COUNT; ++c) { h->prof_ms[c] = 0.0; h->prof_count[c] = 0; h->prof_work[c] = 0.0; }
  return rc;
}

RC_API int rcgp_profile_get(rcgp_handle h, int cls, int64_t* launches, double* total_ms, double* work) {
  RC_CHECK_H(h);
  if (cls < 0 || cls >= RC_K_COUNT) return -2;
  int rc = rc_prof_collect(h);
  if (launches) *launches = (int64_t)h->prof_count[cls];
  if (total_ms) *total_ms = h->prof_ms[cls];
  if (work) *work = h->prof_work[cls];
  return rc;
}
