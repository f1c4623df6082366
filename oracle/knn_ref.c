/* Oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the reference's dense k-NN graph
 *   models/vig.py:262-274 xy_pairwise_distance, :312-329 xy_dense_knn_matrix, :357-381 DenseDilatedKnnGraph
 * with the arithmetic ORDER pinned (PyTorch leaves it unspecified) so that the HIP kernel can be compared
 * bit for bit:  every reduction over channels is an ascending fmaf chain, dist = (sqx + (-2*inner)) + sqy,
 * top-k = K smallest distances, ties -> lowest index, sorted ascending.
 * Layout: x [B][C][N], y [B][C][M] (channel-major, as (B,C,N,1) tensors are stored). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void prepare(const float* x, float* xn, float* sq, int C, int P, int normalize) {
  for (int p = 0; p < P; ++p) {
    float denom = 1.0f;
    if (normalize) {
      float s = 0.0f;
      for (int c = 0; c < C; ++c) s = fmaf(x[(size_t)c * P + p], x[(size_t)c * P + p], s);
      denom = fmaxf(sqrtf(s), 1e-12f); /* F.normalize(p=2, dim=1): x / max(||x||, eps)  (vig.py:372-378) */
    }
    float q = 0.0f;
    for (int c = 0; c < C; ++c) {
      float v = x[(size_t)c * P + p];
      if (normalize) v = v / denom;
      xn[(size_t)c * P + p] = v;
      q = fmaf(v, v, q);
    }
    sq[p] = q;
  }
}

/* edge [2][B][N][Kout] int64; dist_out (optional) [B][N][M]; relpos (optional) [N][M]. Returns 0. */
int knn_ref(const float* x, const float* y, const float* relpos, int64_t* edge, float* dist_out, int B, int C, int N,
            int M, int K, int dilation, int normalize) {
  const int Kout = (K + dilation - 1) / dilation;
  float* xn = (float*)malloc(sizeof(float) * (size_t)C * N);
  float* yn = (float*)malloc(sizeof(float) * (size_t)C * M);
  float* sqx = (float*)malloc(sizeof(float) * N);
  float* sqy = (float*)malloc(sizeof(float) * M);
  float* d = (float*)malloc(sizeof(float) * M);
  unsigned char* used = (unsigned char*)malloc(M);
  if (!xn || !yn || !sqx || !sqy || !d || !used) return -1;
  for (int b = 0; b < B; ++b) {
    const float* xb = x + (size_t)b * C * N;
    prepare(xb, xn, sqx, C, N, normalize);
    if (y) {
      prepare(y + (size_t)b * C * M, yn, sqy, C, M, normalize);
    } else {
      for (size_t i = 0; i < (size_t)C * M; ++i) yn[i] = xn[i];
      for (int m = 0; m < M; ++m) sqy[m] = sqx[m];
    }
    for (int n = 0; n < N; ++n) {
      for (int m = 0; m < M; ++m) {
        float inner = 0.0f;
        for (int c = 0; c < C; ++c) inner = fmaf(xn[(size_t)c * N + n], yn[(size_t)c * M + m], inner);
        float v = (sqx[n] + (-2.0f * inner)) + sqy[m]; /* vig.py:271-274 association order */
        if (relpos) v += relpos[(size_t)n * M + m];     /* vig.py:326 */
        d[m] = v;
        used[m] = 0;
        if (dist_out) dist_out[((size_t)b * N + n) * M + m] = v;
      }
      for (int t = 0; t < K; ++t) { /* selection: smallest distance, lowest index on ties */
        int best = -1;
        for (int m = 0; m < M; ++m)
          if (!used[m] && (best < 0 || d[m] < d[best])) best = m;
        used[best] = 1;
        if (t % dilation == 0) { /* DenseDilated: edge_index[..., ::dilation]  (vig.py:351-353) */
          const size_t o = ((size_t)b * N + n) * Kout + t / dilation;
          edge[o] = best;
          edge[(size_t)B * N * Kout + o] = n; /* centre ids (vig.py:328) */
        }
      }
    }
  }
  free(xn); free(yn); free(sqx); free(sqy); free(d); free(used);
  return 0;
}
