// Single-head MultiHeadAttention block of GModule / TGCN (reference models/transformer.py:28-78, version "v2", num_heads = 1) as ONE
// entry point per direction.
//
//   k = key Wk^T + bk, v = value Wv^T + bv, q = query Wq^T + bq;  P = softmax(q k^T * scale);  A = P (.) mask_att (dropout);
//   ctx = A v;  L = ctx Wf^T + bf;  z = query + L (.) mask_out (dropout);  out = LayerNorm(z);  returns (out, A)
//
// Nothing here is a new GPU algorithm: the entry points issue the SAME kernels the composed form issues (ge_gemm, ge_softmax_*,
// ge_layernorm_*, ge_colsum*), in the same order, plus three element-wise kernels for the two dropout masks and the residual.  What
// changes is the host: at 4 + 4 frames per step GModule's ~400 eager launches are the critical path of the training step (the pyramid
// backward waits for them), each costing ~25 us of Python / autograd / ctypes time; the four attention blocks of a step were 11
// forward and ~22 backward launches each.  One call per direction leaves ~3 us per launch (tools/host_profile.py).
// The dropout masks are produced by the caller with torch's generator (same draws, same order as F.dropout in the composed form).
#include "ge_common.h"

extern "C" {
int ge_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam, long long sak, long long sbk,
            long long sbn, long long scm, long long scn, int batch, long long bsA, long long bsB, long long bsC, float alpha, int bias_mode,
            int relu, int accumulate, void* stream);
int ge_softmax_fwd(const float* x, float* y, int R, int D, float scale, void* stream);
int ge_softmax_bwd(const float* dy, const float* p, float* dx, int R, int D, float scale, void* stream);
int ge_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd, int R, int D, float eps,
                     void* stream);
int ge_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* invstd, float* dx,
                     float* dgamma_part, float* dbeta_part, float* dgamma, float* dbeta, int R, int D, void* stream);
int ge_gemm_rowsum_ok(int M, int N, int K, int batch);
int ge_gemm_rowsum(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                   long long scm, long long scn, int batch, long long bsA, long long bsB, long long bsC, float alpha, int accumulate,
                   float* asum, int asum_accumulate, void* stream);
int ge_colsum(const float* in, float* out, int R, int C, void* stream);
int ge_colsum_accumulate(const float* in, float* out, int R, int C, void* stream);
}

// out = a * (m * ms) (m nullable: copy)
__global__ __launch_bounds__(256) void att_mul_kernel(const float* a, const float* m, float ms, float* out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = m ? a[i] * (m[i] * ms) : a[i];
}
// out = r + a * m
__global__ __launch_bounds__(256) void att_res_kernel(const float* __restrict__ r, const float* __restrict__ a, const float* __restrict__ m,
                                                      float ms, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = r[i] + (m ? a[i] * (m[i] * ms) : a[i]);
}
// out = (a + b) * m   (b, m nullable)
__global__ __launch_bounds__(256) void att_addmul_kernel(const float* a, const float* b, const float* m, float ms, float* out,
                                                         long long n) {      // (a == out allowed)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = b ? a[i] + b[i] : a[i];
    out[i] = m ? v * (m[i] * ms) : v;
  }
}

#define ATT_TRY(call)       \
  do {                      \
    const int rc__ = (call); \
    if (rc__ != GE_OK) return rc__; \
  } while (0)

// y[R][N] = x[R][K] W[N][K]^T + b
static int att_linear(const float* x, const float* W, const float* b, float* y, int R, int N, int K, void* st) {
  return ge_gemm(x, W, b, y, R, N, K, K, 1, 1, K, N, 1, 1, 0, 0, 0, 1.f, b ? 2 : 0, 0, 0, st);
}
// dx[R][K] (+)= dy[R][N] W[N][K];  dW[N][K] (+)= dy^T x;  db[N] (+)= column sums of dy
static int att_linear_bwd(const float* x, const float* W, const float* dy, float* dx, int dx_acc, float* dW, int dW_acc, float* db, int db_acc,
                          int R, int N, int K, void* st) {
  if (dx) ATT_TRY(ge_gemm(dy, W, nullptr, dx, R, K, N, N, 1, K, 1, K, 1, 1, 0, 0, 0, 1.f, 0, 0, dx_acc, st));
  if (dW && db && ge_gemm_rowsum_ok(N, K, R, 1)) {      // dW = dy^T x and db = column sums of dy in ONE launch (the A operand IS dy^T)
    ATT_TRY(ge_gemm_rowsum(dy, x, dW, N, K, R, 1, N, K, 1, K, 1, 1, 0, 0, 0, 1.f, dW_acc, db, db_acc, st));
    return GE_OK;
  }
  if (dW) ATT_TRY(ge_gemm(dy, x, nullptr, dW, N, K, R, 1, N, K, 1, K, 1, 1, 0, 0, 0, 1.f, 0, 0, dW_acc, st));
  if (db) ATT_TRY(db_acc ? ge_colsum_accumulate(dy, db, R, N, st) : ge_colsum(dy, db, R, N, st));
  return GE_OK;
}

extern "C" {

// floats of scratch the backward needs (beyond what the forward saved): dz / dquery [Nq][D] is an OUTPUT; this is dL, dC, dq [Nq][D] each,
// dk, dv [Nk][D] each, dA / dP and dS [Nq][Nk] each, the LayerNorm partials [2][ceil(Nq / 32)][D]
long long ge_mha1_bwd_workspace(int Nk, int Nq, int D) {
  return 3ll * Nq * D + 2ll * Nk * D + 2ll * Nq * Nk + 2ll * ((Nq + 31) / 32) * D;
}

// Forward.  Saved for the backward (caller-owned): k, v [Nk][D]; q, ctx, z [Nq][D]; P, A [Nq][Nk]; mean, invstd [Nq].
// mask_att [Nq][Nk] / mask_out [Nq][D]: 0 / 1 keep masks (null: no dropout); att_scale / out_scale = 1 / keep probability of each site.
int ge_mha1_fwd(const float* key, const float* value, const float* query, const float* Wk, const float* bk, const float* Wv,
                const float* bv, const float* Wq, const float* bq, const float* Wf, const float* bf, const float* gamma, const float* beta,
                const float* mask_att, const float* mask_out, float* k, float* v, float* q, float* P, float* A, float* ctx, float* z,
                float* mean, float* invstd, float* out, int Nk, int Nq, int D, float scale, float att_scale, float out_scale, float eps, void* stream) {
  GE_REQUIRE(key && value && query && Wk && Wv && Wq && Wf && k && v && q && P && A && ctx && z && mean && invstd && out,
             "mha1_fwd: null pointer");
  GE_REQUIRE(Nk > 0 && Nq > 0 && D > 0, "mha1_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  ATT_TRY(att_linear(key, Wk, bk, k, Nk, D, D, stream));
  ATT_TRY(att_linear(value, Wv, bv, v, Nk, D, D, stream));
  ATT_TRY(att_linear(query, Wq, bq, q, Nq, D, D, stream));
  // S = q k^T into P's buffer, softmax in place is not offered by ge_softmax_fwd: S lives in A's buffer for a moment
  ATT_TRY(ge_gemm(q, k, nullptr, A, Nq, Nk, D, D, 1, 1, D, Nk, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  ATT_TRY(ge_softmax_fwd(A, P, Nq, Nk, scale, stream));
  const long long na = (long long)Nq * Nk, nd = (long long)Nq * D;
  hipLaunchKernelGGL(att_mul_kernel, dim3(ge_stream_grid(na, 256)), dim3(256), 0, st, P, mask_att, att_scale, A, na);
  GE_CHECK_LAUNCH("mha1_att_mask");
  ATT_TRY(ge_gemm(A, v, nullptr, ctx, Nq, D, Nk, Nk, 1, D, 1, D, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  // L = ctx Wf^T + bf into `out` for a moment, z = query + L (.) mask_out, out = LayerNorm(z)
  ATT_TRY(att_linear(ctx, Wf, bf, out, Nq, D, D, stream));
  hipLaunchKernelGGL(att_res_kernel, dim3(ge_stream_grid(nd, 256)), dim3(256), 0, st, query, out, mask_out, out_scale, z, nd);
  GE_CHECK_LAUNCH("mha1_residual");
  ATT_TRY(ge_layernorm_fwd(z, gamma, beta, out, mean, invstd, Nq, D, eps, stream));
  return GE_OK;
}

// Backward.  d_out [Nq][D] required, d_att [Nq][Nk] nullable.  Outputs: dkey, dvalue [Nk][D], dquery [Nq][D] (written, never
// accumulated); parameter gradients: d* pointer + *_acc flag (1: add to what is there -- the flat gradient buffers); dgamma / dbeta
// are written.  ws: ge_mha1_bwd_workspace floats.
int ge_mha1_bwd(const float* key, const float* value, const float* query, const float* Wk, const float* Wv, const float* Wq,
                const float* Wf, const float* gamma, const float* mask_att, const float* mask_out, const float* k, const float* v,
                const float* q, const float* P, const float* A, const float* ctx, const float* z, const float* mean, const float* invstd,
                const float* d_out, const float* d_att, float* dkey, float* dvalue, float* dquery, float* dWk, float* dbk, float* dWv,
                float* dbv, float* dWq, float* dbq, float* dWf, float* dbf, int w_acc, int b_acc, float* dgamma, float* dbeta, float* ws,
                int Nk, int Nq, int D, float scale, float att_scale, float out_scale, void* stream) {
  GE_REQUIRE(key && value && query && Wk && Wv && Wq && Wf && k && v && q && P && A && ctx && z && mean && invstd && d_out && dkey &&
                 dvalue && dquery && ws,
             "mha1_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long nd = (long long)Nq * D, nkd = (long long)Nk * D, na = (long long)Nq * Nk;
  float* dL = ws;
  float* dC = dL + nd;
  float* dq = dC + nd;
  float* dk = dq + nd;
  float* dv = dk + nkd;
  float* dA = dv + nkd;
  float* dS = dA + na;
  float* part = dS + na;
  const int nblk = (Nq + 31) / 32;
  // LayerNorm: dz -> dquery (the residual's share of the query gradient; the q projection's share is accumulated below)
  ATT_TRY(ge_layernorm_bwd(d_out, z, gamma, mean, invstd, dquery, gamma ? part : nullptr, gamma ? part + (size_t)nblk * D : nullptr,
                           gamma ? dgamma : nullptr, gamma ? dbeta : nullptr, Nq, D, stream));
  hipLaunchKernelGGL(att_mul_kernel, dim3(ge_stream_grid(nd, 256)), dim3(256), 0, st, dquery, mask_out, out_scale, dL, nd);
  GE_CHECK_LAUNCH("mha1_bwd_out_mask");
  ATT_TRY(att_linear_bwd(ctx, Wf, dL, dC, 0, dWf, w_acc, dbf, b_acc, Nq, D, D, stream));
  // ctx = A v:  dA = dC v^T,  dv = A^T dC
  ATT_TRY(ge_gemm(dC, v, nullptr, dA, Nq, Nk, D, D, 1, 1, D, Nk, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  ATT_TRY(ge_gemm(A, dC, nullptr, dv, Nk, D, Nq, 1, Nk, D, 1, D, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  // A = P (.) mask (+ the gradient arriving at the returned attention), softmax backward (scale inside)
  hipLaunchKernelGGL(att_addmul_kernel, dim3(ge_stream_grid(na, 256)), dim3(256), 0, st, dA, d_att, mask_att, att_scale, dA, na);
  GE_CHECK_LAUNCH("mha1_bwd_att_mask");
  ATT_TRY(ge_softmax_bwd(dA, P, dS, Nq, Nk, scale, stream));
  // S = q k^T:  dq = dS k,  dk = dS^T q
  ATT_TRY(ge_gemm(dS, k, nullptr, dq, Nq, D, Nk, Nk, 1, D, 1, D, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  ATT_TRY(ge_gemm(dS, q, nullptr, dk, Nk, D, Nq, 1, Nk, D, 1, D, 1, 1, 0, 0, 0, 1.f, 0, 0, 0, stream));
  ATT_TRY(att_linear_bwd(query, Wq, dq, dquery, 1, dWq, w_acc, dbq, b_acc, Nq, D, D, stream));
  ATT_TRY(att_linear_bwd(key, Wk, dk, dkey, 0, dWk, w_acc, dbk, b_acc, Nk, D, D, stream));
  ATT_TRY(att_linear_bwd(value, Wv, dv, dvalue, 0, dWv, w_acc, dbv, b_acc, Nk, D, D, stream));
  return GE_OK;
}

}  // extern "C"
