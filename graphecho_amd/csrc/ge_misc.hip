// Remaining hot-path kernels: fused Affinity MLP (models/affinity_layer.py:52-73), row softmax for the
// single-head attention (models/transformer.py:5-23), segmentation-loss reductions (utils/losses.py,
// nn.BCEWithLogitsLoss) and the fused flat-buffer optimizers (Adam / SGD-momentum, train_camus_echo.py:425-435).
#include "ge_common.h"

// ---------------------------------------------------------------------------------------------
// Affinity: M[i][j] = b2 + sum_h w2[h] * relu(P[i][h] + Q[j][h] + b1[h])
// where P = project_sr(X) @ W1[:, :d]^T and Q = project_tg(Y) @ W1[:, d:]^T (built with ge_gemm), i.e. the
// reference's (N1,N2,2d) broadcast-concat + Linear(2d,2d) is never materialised.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affinity_fwd_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, float* __restrict__ M, int N1,
                                                           int N2, int H) {
  __shared__ float Ps[16][65], Qs[16][65], ws[64];
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int ti = threadIdx.x / 16, tj = threadIdx.x % 16;
  float acc = 0.f;
  for (int h0 = 0; h0 < H; h0 += 64) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int r = e / 64, h = e % 64;
      const bool hok = h0 + h < H;
      // fold b1 into the P tile so the inner loop is add + relu + fma
      Ps[r][h] = (i0 + r < N1 && hok) ? P[(size_t)(i0 + r) * H + h0 + h] + b1[h0 + h] : 0.f;
      Qs[r][h] = (j0 + r < N2 && hok) ? Q[(size_t)(j0 + r) * H + h0 + h] : 0.f;
    }
    if (threadIdx.x < 64) ws[threadIdx.x] = (h0 + threadIdx.x < H) ? w2[h0 + threadIdx.x] : 0.f;
    __syncthreads();
#pragma unroll 8
    for (int h = 0; h < 64; ++h) acc = fmaf(ws[h], fmaxf(Ps[ti][h] + Qs[tj][h], 0.f), acc);
    __syncthreads();
  }
  if (i0 + ti < N1 && j0 + tj < N2) M[(size_t)(i0 + ti) * N2 + j0 + tj] = acc + b2[0];
}

// dA[a][h] = w2[h] * sum_b dM(a,b) * [A[a][h] + Bm[b][h] + b1[h] > 0]      (A = P, Bm = Q, or swapped)
// dw2_part[a-tile][h] = sum_{a in tile} sum_b dM(a,b) * relu(pre)         (only when dw2_part != null)
// dM(a,b) = dM[a*sa + b*sb].  4 rows of A per workgroup, threads walk h.
__global__ __launch_bounds__(256) void affinity_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ dM, long long sa, long long sb,
                                                           float* __restrict__ dA, float* __restrict__ dw2_part,
                                                           int NA, int NB, int H) {
  // workgroup = 2 rows of A x 256 values of h (grid.y walks h): 4x the workgroups of the 4-row / all-h form, which put
  // 59 workgroups on 256 CUs for GModule's ~240 nodes (the loop is VALU-bound: 3 * NA * NB * H compare / add steps)
  const int a0 = blockIdx.x * 2;
  const int h = blockIdx.y * 256 + threadIdx.x;
  if (h >= H) return;
  float pa[2], acc[2] = {0.f, 0.f}, accw = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) pa[r] = (a0 + r < NA) ? A[(size_t)(a0 + r) * H + h] + b1[h] : -INFINITY;
  // eight rows of Bm in flight per step; b ascending: the sums are the same, term for term
  for (int b0 = 0; b0 < NB; b0 += 8) {
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = Bm[(size_t)min(b0 + u, NB - 1) * H + h];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u;
      if (b >= NB) break;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (a0 + r < NA) {
          const float pre = pa[r] + q[u];
          const float g = dM[(size_t)(a0 + r) * sa + (size_t)b * sb];
          if (pre > 0.f) {
            acc[r] += g;
            accw += g * pre;
          }
        }
      }
    }
  }
  const float wh = w2[h];
#pragma unroll
  for (int r = 0; r < 2; ++r)
    if (a0 + r < NA) dA[(size_t)(a0 + r) * H + h] = wh * acc[r];
  if (dw2_part) dw2_part[(size_t)blockIdx.x * H + h] = accw;
}

// ---------------------------------------------------------------------------------------------
// Row softmax (last dim), one wave per row; optional input scale.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int R,
                                                          int D, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* xp = x + (size_t)row * D;
  float* yp = y + (size_t)row * D;
  float mx = -INFINITY;
  for (int i = lane; i < D; i += 64) mx = fmaxf(mx, xp[i] * scale);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float e = expf(xp[i] * scale - mx);
    yp[i] = e;
    s += e;
  }
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int i = lane; i < D; i += 64) yp[i] *= inv;
}
// dx = scale * p * (dy - sum(dy * p))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ p,
                                                          float* __restrict__ dx, int R, int D, float scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* gp = dy + (size_t)row * D;
  const float* pp = p + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += gp[i] * pp[i];
  s = wave_sum(s);
  for (int i = lane; i < D; i += 64) dx[(size_t)row * D + i] = scale * pp[i] * (gp[i] - s);
}

// ---------------------------------------------------------------------------------------------
// BCE-with-logits (mean).  target: tensor t[i], or the constant tconst when t == null.
// partial[blk] = sum over the block's slice of max(x,0) - x*t + log1p(exp(-|x|)).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                      float tconst, float* __restrict__ partial, long long n) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i], tt = t ? t[i] : tconst;
    s += fmaxf(v, 0.f) - v * tt + log1pf(expf(-fabsf(v)));
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                      float tconst, const float* __restrict__ g,
                                                      float* __restrict__ dx, long long n, float inv_n) {
  const float gs = g[0] * inv_n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i], tt = t ? t[i] : tconst;
    dx[i] = (1.f / (1.f + expf(-v)) - tt) * gs;
  }
}
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int n, float scale) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// PReLU with one learned slope (act_layer 'prelu', models/vig.py:441-442): y = x > 0 ? x : a*x.
// Backward: dx likewise; da = sum over x <= 0 of dy*x (per-block partials, then sum_partials_kernel).
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                        float* __restrict__ y, long long n) {
  const float slope = a[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    y[i] = v > 0.f ? v : v * slope;
  }
}
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ a, float* __restrict__ dx,
                                                        float* __restrict__ partial, long long n) {
  __shared__ float red[16];
  const float slope = a[0];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = dy[i], v = x[i];
    dx[i] = v > 0.f ? g : g * slope;
    if (!(v > 0.f)) acc += g * v;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// ---------------------------------------------------------------------------------------------
// DiceLoss (utils/losses.py:81-95): softmax over C, then per (b, c): sum p*t, sum p^2, sum t^2 over HW.
// prob [B][C][HW] is written for the backward pass; sums [B][C][3].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dice_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                       float* __restrict__ prob, float* __restrict__ partial, int C,
                                                       int HW, int NBLK) {
  __shared__ float red[16];
  const int b = blockIdx.y, blk = blockIdx.x;
  const float* xb = x + (size_t)b * C * HW;
  const float* tb = t + (size_t)b * C * HW;
  float* pb = prob + (size_t)b * C * HW;
  for (int c = 0; c < C; ++c) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = blk * 256 + threadIdx.x; i < HW; i += NBLK * 256) {
      float mx = -INFINITY;
      for (int k = 0; k < C; ++k) mx = fmaxf(mx, xb[(size_t)k * HW + i]);
      float s = 0.f;
      for (int k = 0; k < C; ++k) s += expf(xb[(size_t)k * HW + i] - mx);
      const float p = expf(xb[(size_t)c * HW + i] - mx) / s;
      const float tt = tb[(size_t)c * HW + i];
      pb[(size_t)c * HW + i] = p;
      a0 += p * tt;
      a1 += p * p;
      a2 += tt * tt;
    }
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    a2 = block_sum(a2, red);
    if (threadIdx.x == 0) {
      float* o = partial + (((size_t)b * C + c) * NBLK + blk) * 3;
      o[0] = a0;
      o[1] = a1;
      o[2] = a2;
    }
  }
}
// Same for C <= CMAX: the softmax of a pixel is evaluated once (the loop above re-evaluates it for every class) and
// the 3*C sums live in registers.
template <int CMAX>
__global__ __launch_bounds__(256) void dice_fwd_small_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                             float* __restrict__ prob, float* __restrict__ partial,
                                                             int C, int HW, int NBLK) {
  __shared__ float red[16];
  const int b = blockIdx.y, blk = blockIdx.x;
  const float* xb = x + (size_t)b * C * HW;
  const float* tb = t + (size_t)b * C * HW;
  float* pb = prob + (size_t)b * C * HW;
  float a0[CMAX], a1[CMAX], a2[CMAX];
#pragma unroll
  for (int k = 0; k < CMAX; ++k) a0[k] = a1[k] = a2[k] = 0.f;
  for (int i = blk * 256 + threadIdx.x; i < HW; i += NBLK * 256) {
    float v[CMAX];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < CMAX; ++k)
      if (k < C) {
        v[k] = xb[(size_t)k * HW + i];
        mx = fmaxf(mx, v[k]);
      }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k)
      if (k < C) {
        v[k] = expf(v[k] - mx);
        s += v[k];
      }
#pragma unroll
    for (int k = 0; k < CMAX; ++k)
      if (k < C) {
        const float p = v[k] / s;
        const float tt = tb[(size_t)k * HW + i];
        pb[(size_t)k * HW + i] = p;
        a0[k] += p * tt;
        a1[k] += p * p;
        a2[k] += tt * tt;
      }
  }
#pragma unroll
  for (int k = 0; k < CMAX; ++k)
    if (k < C) {
      const float r0 = block_sum(a0[k], red), r1 = block_sum(a1[k], red), r2 = block_sum(a2[k], red);
      if (threadIdx.x == 0) {
        float* o = partial + (((size_t)b * C + k) * NBLK + blk) * 3;
        o[0] = r0;
        o[1] = r1;
        o[2] = r2;
      }
    }
}
// dp[b][c][i] = ca[b][c]*t + cb[b][c]*2p ; dx_c = p_c (dp_c - sum_k dp_k p_k)
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ prob, const float* __restrict__ t,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       float* __restrict__ dx, int B, int C, int HW) {
  const long long total = (long long)B * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int b = (int)(e / HW), i = (int)(e - (long long)b * HW);
    const float* pb = prob + (size_t)b * C * HW + i;
    const float* tb = t + (size_t)b * C * HW + i;
    float dot = 0.f;
    for (int k = 0; k < C; ++k) {
      const float p = pb[(size_t)k * HW];
      dot += (ca[b * C + k] * tb[(size_t)k * HW] + cb[b * C + k] * 2.f * p) * p;
    }
    for (int k = 0; k < C; ++k) {
      const float p = pb[(size_t)k * HW];
      const float dp = ca[b * C + k] * tb[(size_t)k * HW] + cb[b * C + k] * 2.f * p;
      dx[(size_t)b * C * HW + (size_t)k * HW + i] = p * (dp - dot);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Flat-buffer optimizers.  torch.optim.Adam semantics (L2 weight decay added to the gradient, bias correction).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, float gscale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
// torch.optim.SGD: g += wd*p; buf = momentum*buf + g (buf = g on the first step); p -= lr*buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, long long n, float lr, float momentum,
                                                  float wd, int first, float gscale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    if (momentum != 0.f) {
      const float bi = first ? gi : momentum * buf[i] + gi;
      buf[i] = bi;
      gi = bi;
    }
    p[i] = pi - lr * gi;
  }
}

// The same updates with the "received a gradient" decision taken ON THE DEVICE (data-parallel training: the map is the
// MAX over ranks of per-parameter flags, all-reduced as one small device tensor -- the host never reads it).  seg_end:
// ascending exclusive end offsets of the parameters inside the flat buffer; used[s] > 0: step parameter s.  i0: offset
// of p[0] in the flat buffer (sharded steps pass sub-ranges).  A workgroup owns a contiguous piece, so a thread's
// successive elements stay inside a segment or move to the next one.
__device__ __forceinline__ int seg_of(const int* __restrict__ s_end, int nseg, long long gi, int hint) {
  if (hint < nseg && gi < s_end[hint] && (hint == 0 || gi >= s_end[hint - 1])) return hint;
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (gi < s_end[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void adam_masked_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, long long n,
                                                          long long i0, const int* __restrict__ seg_end,
                                                          const float* __restrict__ used, int nseg, float lr, float beta1,
                                                          float beta2, float eps, float wd, float bc1, float bc2_sqrt,
                                                          float gscale) {
  const long long per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long long beg = blockIdx.x * per, end = min(beg + per, n);
  int sg = 0;
  for (long long i = beg + threadIdx.x; i < end; i += 256) {
    sg = seg_of(seg_end, nseg, i0 + i, sg);
    if (!(used[sg] > 0.f)) continue;
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
// started[s] > 0: parameter s has a momentum buffer already (else buf = g: torch.optim.SGD's first step of a parameter)
__global__ __launch_bounds__(256) void sgd_masked_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ buf, long long n, long long i0,
                                                         const int* __restrict__ seg_end, const float* __restrict__ used,
                                                         const float* __restrict__ started, int nseg, float lr,
                                                         float momentum, float wd, float gscale) {
  const long long per = ((n + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
  const long long beg = blockIdx.x * per, end = min(beg + per, n);
  int sg = 0;
  for (long long i = beg + threadIdx.x; i < end; i += 256) {
    sg = seg_of(seg_end, nseg, i0 + i, sg);
    if (!(used[sg] > 0.f)) continue;
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    if (momentum != 0.f) {
      const float bi = started[sg] > 0.f ? momentum * buf[i] + gi : gi;
      buf[i] = bi;
      gi = bi;
    }
    p[i] = pi - lr * gi;
  }
}
__global__ void flags_max_kernel(float* __restrict__ a, const float* __restrict__ b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = fmaxf(a[i], b[i]);
}

__global__ void strided_sum3_kernel(const float* __restrict__ partial, float* __restrict__ sums, int n, int nb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 3) return;
  const int e = i / 3, q = i - e * 3;
  float s = 0.f;
  for (int k = 0; k < nb; ++k) s += partial[((size_t)e * nb + k) * 3 + q];
  sums[i] = s;
}
extern "C" int ge_strided_sum3(const float* partial, float* sums, int n, int nb, void* stream) {
  hipLaunchKernelGGL(strided_sum3_kernel, dim3(ge_cdiv(n * 3, 64)), dim3(64), 0, (hipStream_t)stream, partial, sums, n,
                     nb);
  GE_CHECK_LAUNCH("strided_sum3");
  return GE_OK;
}

// mean(x^2) over a whole tensor (activation-energy loss of the config-2 harness): partial sums of squares per
// workgroup, one finishing workgroup; backward dx = x * (2/n) * g with g read from device memory.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                            long long n) {
  __shared__ float red[16];
  float s0 = 0.f, s1 = 0.f;
  const long long n4 = n >> 2;
  const float4* x4 = (const float4*)x;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 a = x4[i], b = x4[i + stride];
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    s1 += (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
  }
  if (i < n4) {
    const float4 a = x4[i];
    s0 += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
  }
  if (blockIdx.x == 0)
    for (long long t = (n4 << 2) + threadIdx.x; t < n; t += 256) s0 += x[t] * x[t];
  const float s = block_sum(s0 + s1, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float inv_n,
                                                          float* __restrict__ out) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = s * inv_n;
}
__global__ __launch_bounds__(256) void scale_by_device_scalar_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ g, float alpha,
                                                                     float* __restrict__ dx, long long n) {
  const float k = alpha * g[0];
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = ((const float4*)x)[i];
    v.x *= k;
    v.y *= k;
    v.z *= k;
    v.w *= k;
    ((float4*)dx)[i] = v;
  }
  if (blockIdx.x == 0)
    for (long long t = (n4 << 2) + threadIdx.x; t < n; t += 256) dx[t] = x[t] * k;
}

extern "C" {

int ge_affinity_fwd(const float* P, const float* Q, const float* b1, const float* w2, const float* b2, float* M,
                    int N1, int N2, int H, void* stream) {
  GE_REQUIRE(P && Q && b1 && w2 && b2 && M && N1 > 0 && N2 > 0 && H > 0, "affinity_fwd: bad arguments");
  hipLaunchKernelGGL(affinity_fwd_kernel, dim3(ge_cdiv(N2, 16), ge_cdiv(N1, 16)), dim3(256), 0, (hipStream_t)stream, P,
                     Q, b1, w2, b2, M, N1, N2, H);
  GE_CHECK_LAUNCH("affinity_fwd");
  return GE_OK;
}

// dP [N1][H], dQ [N2][H]; dw2_part: [ceil(N1/4)][H] workspace (column-sum it for dw2).
int ge_affinity_bwd(const float* P, const float* Q, const float* b1, const float* w2, const float* dM, float* dP,
                    float* dQ, float* dw2_part, int N1, int N2, int H, void* stream) {
  GE_REQUIRE(P && Q && b1 && w2 && dM && dP && dQ && dw2_part, "affinity_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(affinity_bwd_kernel, dim3(ge_cdiv(N1, 2), ge_cdiv(H, 256)), dim3(256), 0, st, P, Q, b1, w2, dM,
                     (long long)N2, 1ll, dP, dw2_part, N1, N2, H);
  hipLaunchKernelGGL(affinity_bwd_kernel, dim3(ge_cdiv(N2, 2), ge_cdiv(H, 256)), dim3(256), 0, st, Q, P, b1, w2, dM, 1ll,
                     (long long)N2, dQ, (float*)nullptr, N2, N1, H);
  GE_CHECK_LAUNCH("affinity_bwd");
  return GE_OK;
}

int ge_softmax_fwd(const float* x, float* y, int R, int D, float scale, void* stream) {
  GE_REQUIRE(x && y && R > 0 && D > 0, "softmax_fwd: bad arguments");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(ge_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, y, R, D, scale);
  GE_CHECK_LAUNCH("softmax_fwd");
  return GE_OK;
}
int ge_softmax_bwd(const float* dy, const float* p, float* dx, int R, int D, float scale, void* stream) {
  GE_REQUIRE(dy && p && dx && R > 0 && D > 0, "softmax_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(ge_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, dy, p, dx, R, D,
                     scale);
  GE_CHECK_LAUNCH("softmax_bwd");
  return GE_OK;
}

// loss[0] = mean BCE-with-logits; partial: >= 1024 floats workspace.
int ge_bce_logits_fwd(const float* x, const float* t, float tconst, float* partial, float* loss, long long n,
                      void* stream) {
  GE_REQUIRE(x && partial && loss && n > 0, "bce_logits_fwd: bad arguments");
  int nb = ge_stream_grid(n, 1024);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, t, tconst, partial, n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, loss, nb,
                     1.f / (float)n);
  GE_CHECK_LAUNCH("bce_logits_fwd");
  return GE_OK;
}
int ge_bce_logits_bwd(const float* x, const float* t, float tconst, const float* g, float* dx, long long n,
                      void* stream) {
  GE_REQUIRE(x && g && dx && n > 0, "bce_logits_bwd: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, t, tconst, g,
                     dx, n, 1.f / (float)n);
  GE_CHECK_LAUNCH("bce_logits_bwd");
  return GE_OK;
}

int ge_prelu_num_partials(long long n) { return ge_stream_grid(n, 256); }
int ge_prelu_fwd(const float* x, const float* slope, float* y, long long n, void* stream) {
  GE_REQUIRE(x && slope && y && n > 0, "prelu_fwd: bad arguments");
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, slope, y, n);
  GE_CHECK_LAUNCH("prelu_fwd");
  return GE_OK;
}
// partial: ge_prelu_num_partials(n) floats of workspace; dslope: 1 float (overwritten).
int ge_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* partial, float* dslope,
                 long long n, void* stream) {
  GE_REQUIRE(dy && x && slope && dx && partial && dslope && n > 0, "prelu_bwd: bad arguments");
  const int nb = ge_stream_grid(n, 256);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, slope, dx, partial, n);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, dslope, nb, 1.f);
  GE_CHECK_LAUNCH("prelu_bwd");
  return GE_OK;
}

// prob [B][C][HW]; partial [B][C][nblk][3] with nblk = ge_dice_num_partials(HW); sums [B][C][3].
int ge_dice_num_partials(int HW) {
  int nb = ge_cdiv(HW, 1024);
  return nb > 64 ? 64 : nb;
}
int ge_dice_fwd(const float* x, const float* t, float* prob, float* partial, float* sums, int B, int C, int HW,
                void* stream) {
  GE_REQUIRE(x && t && prob && partial && sums && B > 0 && C > 0 && HW > 0, "dice_fwd: bad arguments");
  const int nb = ge_dice_num_partials(HW);
  if (C <= 8)
    hipLaunchKernelGGL(dice_fwd_small_kernel<8>, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, t, prob, partial, C,
                       HW, nb);
  else
    hipLaunchKernelGGL(dice_fwd_kernel, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, x, t, prob, partial, C, HW, nb);
  GE_CHECK_LAUNCH("dice_fwd");
  // sums[(b,c)][q] = sum_blk partial[(b,c)][blk][q]: view as R=nb rows with row stride 3 -> small colsum per (b,c)
  return ge_strided_sum3(partial, sums, B * C, nb, stream);
}
int ge_dice_bwd(const float* prob, const float* t, const float* ca, const float* cb, float* dx, int B, int C, int HW,
                void* stream) {
  GE_REQUIRE(prob && t && ca && cb && dx, "dice_bwd: null pointer");
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(ge_stream_grid((long long)B * HW, 256)), dim3(256), 0, (hipStream_t)stream,
                     prob, t, ca, cb, dx, B, C, HW);
  GE_CHECK_LAUNCH("dice_bwd");
  return GE_OK;
}

int ge_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream) {
  GE_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
  GE_CHECK_LAUNCH("adam_step");
  return GE_OK;
}
int ge_adam_step_masked(float* p, const float* g, float* m, float* v, long long n, long long i0, const int* seg_end,
                        const float* used, int nseg, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, void* stream) {
  GE_REQUIRE(p && g && m && v && seg_end && used && n > 0 && nseg > 0 && step >= 1, "adam_step_masked: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_masked_kernel, dim3(ge_stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                     i0, seg_end, used, nseg, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
  GE_CHECK_LAUNCH("adam_step_masked");
  return GE_OK;
}
int ge_sgd_step_masked(float* p, const float* g, float* buf, long long n, long long i0, const int* seg_end,
                       const float* used, const float* started, int nseg, float lr, float momentum, float weight_decay,
                       float grad_scale, void* stream) {
  GE_REQUIRE(p && g && seg_end && used && n > 0 && nseg > 0 && (momentum == 0.f || (buf && started)),
             "sgd_step_masked: bad arguments");
  hipLaunchKernelGGL(sgd_masked_kernel, dim3(ge_stream_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, i0,
                     seg_end, used, started, nseg, lr, momentum, weight_decay, grad_scale);
  GE_CHECK_LAUNCH("sgd_step_masked");
  return GE_OK;
}
// a[i] = max(a[i], b[i]): "has a momentum buffer" |= "stepped this time", every optimizer's flags in one launch
int ge_flags_max(float* a, const float* b, int n, void* stream) {
  GE_REQUIRE(a && b && n > 0, "flags_max: bad arguments");
  hipLaunchKernelGGL(flags_max_kernel, dim3(ge_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, n);
  GE_CHECK_LAUNCH("flags_max");
  return GE_OK;
}
int ge_sgd_step(float* p, const float* g, float* buf, long long n, float lr, float momentum, float weight_decay,
                int first_step, float grad_scale, void* stream) {
  GE_REQUIRE(p && g && n > 0 && (momentum == 0.f || buf), "sgd_step: bad arguments");
  hipLaunchKernelGGL(sgd_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr,
                     momentum, weight_decay, first_step, grad_scale);
  GE_CHECK_LAUNCH("sgd_step");
  return GE_OK;
}

// out[0] = mean(x^2); partial: ge_mean_square_blocks(n) floats of workspace
int ge_mean_square_blocks(long long n) {
  long long nb = (n / 4 + 2047) / 2048;
  if (nb > 2048) nb = 2048;
  return nb < 1 ? 1 : (int)nb;
}
int ge_mean_square_fwd(const float* x, float* partial, float* out, long long n, void* stream) {
  GE_REQUIRE(x && partial && out && n > 0 && ((uintptr_t)x & 15) == 0, "mean_square_fwd: bad arguments");
  const int nb = ge_mean_square_blocks(n);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, partial, n);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nb, 1.0f / (float)n, out);
  GE_CHECK_LAUNCH("mean_square_fwd");
  return GE_OK;
}
// dx = x * (2/n) * g[0]   (g: device scalar, the gradient of the mean)
int ge_mean_square_bwd(const float* x, const float* g, float* dx, long long n, void* stream) {
  GE_REQUIRE(x && g && dx && n > 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dx & 15) == 0,
             "mean_square_bwd: bad arguments");
  hipLaunchKernelGGL(scale_by_device_scalar_kernel, dim3(ge_stream_grid(n / 4 + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, g, 2.0f / (float)n, dx, n);
  GE_CHECK_LAUNCH("mean_square_bwd");
  return GE_OK;
}

}  // extern "C"

