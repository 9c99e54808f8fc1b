// Everything between the Cholesky factor and the numbers the host wants: L^-1 (recursive doubling on the MFMA GEMM),
// alpha = K^-1 y, the LML value, and the final reduction of the fused gradient partials.
// Reference call sites: GPflow multivariate_normal / cholesky_solve at gpr/models.py:360-370, 441-444.
#include "common.h"
#include <string.h>

static int ensure_partial_one(rcgp_handle_s* h, rcgp_handle_s* hu, size_t elems) {
  if (hu->partial_elems >= elems) return 0;
  if (hu->partial) { RC_HIP(hipStreamSynchronize(h->stream)); RC_HIP(hipFree(hu->partial)); hu->partial = nullptr; hu->partial_elems = 0; }
  RC_HIP(hipMalloc(&hu->partial, elems * sizeof(double)));
  hu->partial_elems = elems;
  return 0;
}

// Room for `elems` doubles in the reduction scratch -- of every unit when h leads a batched call.
int rc_ensure_partial(rcgp_handle_s* h, size_t elems) {
  int rc;
  for (int u = 0; u < h->nb; ++u)
    if ((rc = ensure_partial_one(h, (h->nb > 1) ? h->bh[u] : h, elems))) return rc;
  return 0;
}

// Per-unit pointer table of a launch led by h (common.h): the leader's buffer that contains `ptr`, the same offset in every unit's.
int rc_bp_bytes(rcgp_handle_s* h, const void* ptr, void** out) {
  struct Buf { double* rcgp_handle_s::*member; size_t bytes; };
  const size_t np = (size_t)h->Np, d = sizeof(double);
  const Buf bufs[] = {{&rcgp_handle_s::A, np * np * d},      {&rcgp_handle_s::Linv, np * np * d},  {&rcgp_handle_s::S, np * np * d},
                      {&rcgp_handle_s::invdiag, np * 128 * d}, {&rcgp_handle_s::w, np * d},         {&rcgp_handle_s::logdiag, np * d},
                      {&rcgp_handle_s::alpha, np * d},       {&rcgp_handle_s::sq, np * d},         {&rcgp_handle_s::y, np * d},
                      {&rcgp_handle_s::Z, np * h->M * d},    {&rcgp_handle_s::X, np * h->M * d},   {&rcgp_handle_s::scal, RC_SCAL_ELEMS * d},
                      {&rcgp_handle_s::partial, h->partial_elems * d},
                      {&rcgp_handle_s::ell_d, ((size_t)h->L * h->M + (size_t)2 * h->L * h->L) * d}};
  const char* p = static_cast<const char*>(ptr);
  for (const Buf& b : bufs) {
    const char* base = reinterpret_cast<const char*>(h->*(b.member));
    if (!base || p < base || p >= base + b.bytes) continue;
    const size_t off = (size_t)(p - base);
    for (int u = 0; u < RC_MAX_BATCH; ++u) {
      out[u] = nullptr;
      if (u >= h->nb) continue;
      char* ub = reinterpret_cast<char*>(h->bh[u]->*(b.member));
      if (!ub) { h->err = "batched launch: a unit lacks a buffer the leader has"; return -8; }
      out[u] = ub + off;
    }
    return 0;
  }
  h->err = "batched launch: pointer outside the leader's buffers";
  return -8;
}

// L^-1 by recursive doubling. Level 0: the 128x128 inverses of the diagonal blocks, all in one launch (k_inv128_batched, potrf.hip).
// Level k has block size s = 128 << k; pair p of that level inverts rows [2ps, 2ps+2s) from the inverses of its halves -- A part
// [2ps, 2ps+s), C part [2ps+s, min(2ps+2s, Np)): T = B A^-1 (k_trtri_T), X21 = -C^-1 T (k_trtri_X) -- every pair of a level in one
// launch of each (the last pair may be short: a launch of its own). One stream, in order.
int rc_trtri(rcgp_handle_s* h) {
  const int64_t Np = h->Np;
  int rc;
  g_rc_stat[1] += h->nb;
  for (int u = 0; u < h->nb; ++u) {                               // (every unit of a batched call; h alone otherwise)
    rcgp_handle_s* hu = (h->nb > 1) ? h->bh[u] : h;
    if (!hu->Linv) {
      RC_HIP(hipMalloc(&hu->Linv, (size_t)Np * Np * sizeof(double)));
      // the strictly upper 128-blocks are never written afterwards: zero once (readers may then run a k-range from a block boundary)
      RC_HIP(hipMemsetAsync(hu->Linv, 0, (size_t)Np * Np * sizeof(double), h->stream));
    }
    if (!hu->S) RC_HIP(hipMalloc(&hu->S, (size_t)Np * Np * sizeof(double)));
  }
  if ((rc = rc_launch_inv128_batched(h))) return rc;
  for (int64_t s = 128; s < Np; s *= 2) {
    const int total_pairs = (int)((Np - s + 2 * s - 1) / (2 * s));            // pairs with a non-empty C part: 2ps + s < Np
    int p = 0;
    while (p < total_pairs) {
      const int64_t rowC = 2 * s * (int64_t)p + s, rowEnd = (rowC + s < Np) ? rowC + s : Np;
      int q = p;
      while (q < total_pairs) {                                               // consecutive pairs of the same shape
        const int64_t rc2 = 2 * s * (int64_t)q + s, re2 = (rc2 + s < Np) ? rc2 + s : Np;
        if (re2 - rc2 != rowEnd - rowC) break;
        ++q;
      }
      if ((rc = rc_launch_trtri_T(h, s, p, q - p, 0, (int)((rowEnd - rowC) / 128)))) return rc;
      if ((rc = rc_launch_trtri_X(h, s, p, q - p))) return rc;
      p = q;
    }
  }
  return 0;
}

// partial[kc][j] = sum_{k in chunk kc, k >= tile start of j} Linv[k][j] * w[k]   (rows above the diagonal tile are never read)
__global__ void __launch_bounds__(256) k_gemvT_partial(RcBP<const double> Linvb, int64_t ld, RcBP<const double> wb, int64_t Np, int rows_per_chunk,
                                                       RcBP<double> partialb) {
  const double* __restrict__ Linv = Linvb.p[blockIdx.z];
  const double* __restrict__ w = wb.p[blockIdx.z];
  double* __restrict__ partial = partialb.p[blockIdx.z];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k0c = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t k1 = (k0c + rows_per_chunk < Np) ? k0c + rows_per_chunk : Np;
  double s = 0.0;
  if (j < Np) {
    int64_t k0 = (j >> 7) << 7;
    if (k0 < k0c) k0 = k0c;
    // eight loads in flight per lane and four independent accumulators (fixed order: bit-reproducible); one dependent fma per load
    // left the kernel at 2.9 TB/s
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t k = k0;
    for (; k + 8 <= k1; k += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = Linv[(k + q) * ld + j];
      s0 = fma(v[0], w[k], s0);
      s1 = fma(v[1], w[k + 1], s1);
      s2 = fma(v[2], w[k + 2], s2);
      s3 = fma(v[3], w[k + 3], s3);
      s0 = fma(v[4], w[k + 4], s0);
      s1 = fma(v[5], w[k + 5], s1);
      s2 = fma(v[6], w[k + 6], s2);
      s3 = fma(v[7], w[k + 7], s3);
    }
    for (; k < k1; ++k) s0 = fma(Linv[k * ld + j], w[k], s0);
    s = (s0 + s1) + (s2 + s3);
    partial[(int64_t)blockIdx.y * Np + j] = s;
  }
}

__global__ void k_colreduce2(RcBP<const double> partialb, int64_t rows, int64_t n, RcBP<double> outb) {
  const double* __restrict__ partial = partialb.p[blockIdx.z];
  double* __restrict__ out = outb.p[blockIdx.z];
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int64_t r = 0; r < rows; ++r) s += partial[r * n + j];
  out[j] = s;
}

int rc_alpha(rcgp_handle_s* h) {
  const int64_t Np = h->Np;
  const int rows_per_chunk = 512;
  const int64_t chunks = (Np + rows_per_chunk - 1) / rows_per_chunk;
  int rc = rc_ensure_partial(h, (size_t)chunks * Np);
  if (rc) return rc;
  RC_BP(const double, Lb, h->Linv)
  RC_BP(const double, wb, h->w)
  RC_BP(double, pb, h->partial)
  RC_BP(const double, pcb, (const double*)h->partial)
  RC_BP(double, ab, h->alpha)
  RcProfScope ps(h, RC_K_MISC, 0.0);
  hipLaunchKernelGGL(k_gemvT_partial, dim3((unsigned)((Np + 255) / 256), (unsigned)chunks, (unsigned)h->nb), dim3(256), 0, h->stream, Lb, Np, wb, Np,
                     rows_per_chunk, pb);
  RC_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_colreduce2, dim3((unsigned)((Np + 255) / 256), 1, (unsigned)h->nb), dim3(256), 0, h->stream, pcb, chunks, Np, ab);
  RC_HIP(hipGetLastError());
  return 0;
}

// Deterministic single-block reductions: out[0] = sum w^2, out[1] = sum logdiag.
// With `gather` (a batched evaluation): the unit's whole result block -- these two sums, its gradient sums (already in scal[8 ...]) and its
// Cholesky status word -- is copied into row blockIdx.z of the leader's result table, so that ONE copy brings every unit's numbers down.
__global__ void __launch_bounds__(1024) k_lml_reduce(RcBP<const double> wb, RcBP<const double> logdiagb, int64_t n, RcBP<double> outb, double* gather,
                                                     int ncopy) {
  __shared__ double sa[1024], sb[1024];
  const double* __restrict__ w = wb.p[blockIdx.z];
  const double* __restrict__ logdiag = logdiagb.p[blockIdx.z];
  double* out = outb.p[blockIdx.z];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    a = fma(w[i], w[i], a);
    b += logdiag[i];
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sa[threadIdx.x] += sa[threadIdx.x + o];
      sb[threadIdx.x] += sb[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sa[0]; out[1] = sb[0]; }
  if (gather) {
    double* g = gather + (size_t)blockIdx.z * RC_SCAL_ELEMS;
    if (threadIdx.x == 0) { g[0] = sa[0]; g[1] = sb[0]; }
    else if (threadIdx.x >= 2 && (int)threadIdx.x < ncopy) g[threadIdx.x] = out[threadIdx.x];       // status word [4], gradient sums [8, 8 + M + 2)
  }
}

// The numbers of one evaluation leave the device in ONE copy into pinned host memory behind ONE synchronisation: the two LML sums,
// the Cholesky status word and -- when the caller has queued them (rc_grad_queue / rc_grad_queue_mo) -- the reduced gradient sums.
int rc_lml_value(rcgp_handle_s* h, double* lml) {
  {
    RcProfScope ps(h, RC_K_MISC, 0.0);
    RcBP<const double> wb = {{h->w}}, lb = {{h->logdiag}};
    RcBP<double> ob = {{h->scal}};
    hipLaunchKernelGGL(k_lml_reduce, dim3(1), dim3(1024), 0, h->stream, wb, lb, h->Np, ob, (double*)nullptr, 0);
    RC_HIP(hipGetLastError());
  }
  double* host = h->pin + h->pin_result;
  const size_t n_result = 8 + (h->L == 1 ? (size_t)h->M + 2 : 0);       // LML sums, status word, (single output) the gradient sums
  RC_HIP(hipMemcpyAsync(host, h->scal, n_result * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  RC_HIP(hipStreamSynchronize(h->stream));
  if (h->profiling) rc_prof_collect(h);      // the stream is idle here: harvesting the events costs no extra sync
  int info = 0;
  memcpy(&info, host + RC_SCAL_INFO, sizeof(int));
  if (info != 0) {
    h->err = "matrix is not positive definite: leading minor " + std::to_string(info);
    h->factored = false;
    return info;
  }
  *lml = -0.5 * host[0] - host[1] - 0.5 * (double)(h->N * h->L) * 1.8378770664093454836;   // log(2 pi)
  return 0;
}

// out[c] = sum_r partial[r][c], one block per column, fixed-order tree.
__global__ void __launch_bounds__(256) k_rowreduce(RcBP<const double> partialb, int64_t rows, int cols, RcBP<double> outb) {
  __shared__ double sm[256];
  const double* __restrict__ partial = partialb.p[blockIdx.z];
  double* __restrict__ out = outb.p[blockIdx.z];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += partial[r * cols + c];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = sm[0];
}

// Queue the final reduction of the gradient partials (-> scal[8 ...]); rc_lml_value then brings them down with the LML sums.
int rc_grad_queue(rcgp_handle_s* h, int nrows) {
  RC_BP(const double, pb, (const double*)h->partial)
  RC_BP(double, ob, h->scal + 8)
  RcProfScope ps(h, RC_K_MISC, 0.0);
  hipLaunchKernelGGL(k_rowreduce, dim3((unsigned)(h->M + 2), 1, (unsigned)h->nb), dim3(256), 0, h->stream, pb, (int64_t)nrows, h->M + 2, ob);
  RC_HIP(hipGetLastError());
  return 0;
}

// Batched evaluation led by h: the two LML sums of every unit, and every unit's result block gathered into h->bres_d (k_lml_reduce).
int rc_batch_lml_reduce(rcgp_handle_s* h) {
  RC_BP(const double, wb, h->w)
  RC_BP(const double, lb, h->logdiag)
  RC_BP(double, ob, h->scal)
  RcProfScope ps(h, RC_K_MISC, 0.0);
  hipLaunchKernelGGL(k_lml_reduce, dim3(1, 1, (unsigned)h->nb), dim3(1024), 0, h->stream, wb, lb, h->Np, ob, h->bres_d, 8 + h->M + 2);
  RC_HIP(hipGetLastError());
  return 0;
}

// After rc_lml_value: the gradient from the sums in the pinned result block.
int rc_grad_finish(rcgp_handle_s* h, double* grad) {
  const int M = h->M;
  const double* host = h->pin + h->pin_result + 8;
  for (int m = 0; m < M; ++m) grad[m] = 0.5 * host[m] / h->ell[m];     // dK/dell_m = K (z_im - z_jm)^2 / ell_m
  grad[M] = 0.5 * host[M] / h->var;
  grad[M + 1] = 0.5 * host[M + 1];
  return 0;
}

// Covariant GP: out[p][c] = sum over the lower tiles of block pair p = (bi, bj), bj <= bi, of partial[tile][c]
// (tile (ti, tj) is row ti (ti + 1) / 2 + tj of partial); one workgroup per (c, p), fixed-order tree.
__global__ void __launch_bounds__(256) k_reduce_pairs(const double* __restrict__ partial, int tb, int cols, double* __restrict__ out) {
  __shared__ double sm[256];
  const int c = blockIdx.x, p = blockIdx.y;
  int bi = 0, rem = p;
  while (rem > bi) { rem -= bi + 1; ++bi; }
  const int bj = rem;
  double s = 0.0;
  for (int e = threadIdx.x; e < tb * tb; e += 256) {
    const int64_t ti = (int64_t)bi * tb + e / tb, tj = (int64_t)bj * tb + e % tb;
    if (tj <= ti) s += partial[(ti * (ti + 1) / 2 + tj) * cols + c];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(int64_t)p * cols + c] = sm[0];
}

// Partial derivatives of the LML with every entry of ell (L x M), F (L x L) and Sigma (L x L) treated as independent
// (k_grad_mo explains the per-tile sums and their weights). rc_grad_queue_mo queues the reduction over the tiles of every block pair
// and its copy into the pinned result block; rc_grad_finish_mo runs on the host after rc_lml_value has synchronised.
int rc_grad_queue_mo(rcgp_handle_s* h) {
  const int M = h->M, L = h->L, cols = 2 * M + 2, npairs = L * (L + 1) / 2;
  const int tb = (int)(h->Nb / 128);
  const int64_t T = h->Np / 128;
  double* out_d = h->partial + (size_t)(T * (T + 1) / 2) * cols;          // behind the per-tile rows
  if (h->partial_elems < (size_t)(T * (T + 1) / 2) * cols + (size_t)npairs * cols) { h->err = "rc_grad_queue_mo: scratch too small"; return -7; }
  if (h->pin_elems < h->pin_result + RC_SCAL_ELEMS + (size_t)npairs * cols) { h->err = "rc_grad_queue_mo: staging buffer too small"; return -7; }
  {
    RcProfScope ps(h, RC_K_MISC, 0.0);
    hipLaunchKernelGGL(k_reduce_pairs, dim3((unsigned)cols, (unsigned)npairs), dim3(256), 0, h->stream, h->partial, tb, cols, out_d);
    RC_HIP(hipGetLastError());
  }
  RC_HIP(hipMemcpyAsync(h->pin + h->pin_result + RC_SCAL_ELEMS, out_d, (size_t)npairs * cols * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return 0;
}

int rc_grad_finish_mo(rcgp_handle_s* h, double* g_ell, double* g_F, double* g_S) {
  const int M = h->M, L = h->L, cols = 2 * M + 2;
  const double* host = h->pin + h->pin_result + RC_SCAL_ELEMS;
  std::vector<double> R((size_t)L * M, 0.0);
  int p = 0;
  for (int bi = 0; bi < L; ++bi)
    for (int bj = 0; bj <= bi; ++bj, ++p) {
      const double* o = host + (size_t)p * cols;
      for (int m = 0; m < M; ++m) {
        R[(size_t)bi * M + m] += o[m];
        if (bi != bj) R[(size_t)bj * M + m] -= o[M + m];
      }
      const double f = (bi == bj) ? 0.5 : 0.25;                           // 1/2 tr(W dK); an off-diagonal pair sum holds both mirrored blocks
      g_F[bi * L + bj] = g_F[bj * L + bi] = f * o[2 * M];
      g_S[bi * L + bj] = g_S[bj * L + bi] = f * o[2 * M + 1];
    }
  for (int l = 0; l < L; ++l)
    for (int m = 0; m < M; ++m) g_ell[l * M + m] = 0.5 * R[(size_t)l * M + m] / h->ell[(size_t)l * M + m];
  return 0;
}
