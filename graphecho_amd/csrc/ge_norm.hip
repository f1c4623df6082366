// Normalisation kernels (HBM-bound): BatchNorm2d (train statistics, Chan-merged moments, SyncBN-ready),
// GroupNorm, LayerNorm, InstanceNorm-over-matrix; forward + backward.  fp32, NCHW contiguous.
// Semantics follow the PyTorch defaults the reference relies on.
#include "ge_common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// BatchNorm2d
// ---------------------------------------------------------------------------------------------
// Work decomposition of the per-channel reductions: slice `blk` of channel c is either a 4096-element segment
// of one (b, c) plane (large planes) or a run of whole planes (small planes), so every slice is contiguous
// runs of HW-major memory and is read with 16-byte loads and 32-bit index arithmetic only.
struct BnSlice {
  int planes_per_blk;   // >= 1 when planes are small (HW < 4096), else 0
  int segs_per_plane;   // >= 1 when planes are large
  int NB;
};
static inline BnSlice bn_slice(int B, int HW) {
  BnSlice s;
  if (HW >= 4096) {
    s.planes_per_blk = 0;
    s.segs_per_plane = (HW + 4095) / 4096;
    s.NB = B * s.segs_per_plane;
  } else {
    s.planes_per_blk = 4096 / HW;
    if (s.planes_per_blk < 1) s.planes_per_blk = 1;
    s.segs_per_plane = 0;
    s.NB = (B + s.planes_per_blk - 1) / s.planes_per_blk;
  }
  return s;
}
// Calls f(offset_of_run, length_of_run) for every contiguous run of slice blk of channel c.
template <class F>
__device__ __forceinline__ void bn_for_runs(int blk, int c, int B, int C, int HW, int planes_per_blk,
                                            int segs_per_plane, F f) {
  if (segs_per_plane > 0) {
    const int b = blk / segs_per_plane, sgm = blk - b * segs_per_plane;
    const int beg = sgm * 4096, len = min(4096, HW - beg);
    f(((size_t)b * C + c) * HW + beg, len);
  } else {
    const int b0 = blk * planes_per_blk, b1 = min(b0 + planes_per_blk, B);
    for (int b = b0; b < b1; ++b) f(((size_t)b * C + c) * HW, HW);
  }
}

// Small planes (HW < 4096, a multiple of 4): the slice's planes_per_blk planes as ONE index space of float4 groups, so
// that all 256 threads have work even when a plane is 8x8 (run by run, an 8x8 plane keeps 16 threads busy).
// Calls f(float4 index) for every group of slice blk of channel c.
template <class F>
__device__ __forceinline__ void bn_for_groups4(int blk, int c, int B, int C, int HW, int planes_per_blk, F f) {
  const int b0 = blk * planes_per_blk, nb = min(planes_per_blk, B - b0);
  const int hw4 = HW >> 2, total = nb * hw4;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int pl = e / hw4, i = e - pl * hw4;
    f(((((size_t)(b0 + pl) * C + c) * HW) >> 2) + i);
  }
}

// partial[c][blk] = (n, mean, M2) over slice blk of channel c.
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                               int B, int C, int HW, int NB, int ppb, int spp) {
  __shared__ float red[3 * 4];
  const int c = blockIdx.y, blk = blockIdx.x;
  float s = 0.f, q = 0.f, cnt = 0.f, shift = 0.f;
  bool have_shift = false;
  if (ppb > 1 && (HW & 3) == 0) {
    shift = x[((size_t)(blk * ppb) * C + c) * HW];
    have_shift = true;
    bn_for_groups4(blk, c, B, C, HW, ppb, [&](size_t i4) {
      const float4 v = ((const float4*)x)[i4];
      const float a0 = v.x - shift, a1 = v.y - shift, a2 = v.z - shift, a3 = v.w - shift;
      s += (a0 + a1) + (a2 + a3);
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      cnt += 4.f;
    });
  } else
  bn_for_runs(blk, c, B, C, HW, ppb, spp, [&](size_t off, int len) {
    const float* xp = x + off;
    if (!have_shift) {   // shift by the slice's first element so the sum of squares does not cancel
      shift = xp[0];
      have_shift = true;
    }
    if ((len & 3) == 0 && ((off & 3) == 0)) {
      const float4* x4 = (const float4*)xp;
      for (int i = threadIdx.x; i < (len >> 2); i += 256) {
        const float4 v = x4[i];
        const float a0 = v.x - shift, a1 = v.y - shift, a2 = v.z - shift, a3 = v.w - shift;
        s += (a0 + a1) + (a2 + a3);
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        cnt += 4.f;
      }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) {
        const float v = xp[i] - shift;
        s += v;
        q += v * v;
        cnt += 1.f;
      }
    }
  });
  float mean = cnt > 0.f ? s / cnt : 0.f;
  float m2 = cnt > 0.f ? fmaxf(q - s * mean, 0.f) : 0.f;
  mean += shift;
  wave_moments(cnt, mean, m2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[w * 3 + 0] = cnt;
    red[w * 3 + 1] = mean;
    red[w * 3 + 2] = m2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float n = red[0], mu = red[1], mm = red[2];
    for (int i = 1; i < 4; ++i) moments_merge(n, mu, mm, red[i * 3], red[i * 3 + 1], red[i * 3 + 2]);
    float* o = partial + ((size_t)c * NB + blk) * 3;
    o[0] = n;
    o[1] = mu;
    o[2] = mm;
  }
}

// Merge NB partial moments per channel; partial element (c, i) lives at partial[c*sc + i*sb + {0,1,2}].
// Writes stats[c] = (n, mean, M2), mean/invstd, and (optionally) the running statistics update.
__global__ void bn_finalize_kernel(const float* __restrict__ partial, long long sc, long long sb, int NB, int C,
                                   float eps, float momentum, float* __restrict__ stats, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int i = 0; i < NB; ++i) {
    const float* p = partial + (size_t)c * sc + (size_t)i * sb;
    moments_merge(n, mu, m2, p[0], p[1], p[2]);
  }
  if (stats) {
    stats[c * 3 + 0] = n;
    stats[c * 3 + 1] = mu;
    stats[c * 3 + 2] = m2;
  }
  const float var = n > 0.f ? m2 / n : 0.f;
  if (mean_out) mean_out[c] = mu;
  if (invstd_out) invstd_out[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// Same merge with one wave per channel (lanes stride over the partials, then a Chan butterfly): used when the
// partials come from the conv epilogue (hundreds to thousands per channel).
__global__ __launch_bounds__(256) void bn_finalize_wave_kernel(const float* __restrict__ partial, long long sc,
                                                               long long sb, int NB, int C, float eps, float momentum,
                                                               float* __restrict__ stats, float* __restrict__ mean_out,
                                                               float* __restrict__ invstd_out,
                                                               float* __restrict__ running_mean,
                                                               float* __restrict__ running_var) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  const int lane = threadIdx.x & 63;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  // the lane's triples are merged strictly in order (the rounding of the statistics is part of what the VGG16 gradient
  // tests pin); only the LOADS are batched: eight triples in flight instead of one round trip per merge
  const float* pc = partial + (size_t)c * sc;
  for (int i0 = lane; i0 < NB; i0 += 64 * 8) {
    float tn[8], tm[8], tq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 64 * u;
      const float* p = pc + (size_t)(i < NB ? i : lane) * sb;
      tn[u] = p[0];
      tm[u] = p[1];
      tq[u] = p[2];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 64 * u < NB) moments_merge(n, mu, m2, tn[u], tm[u], tq[u]);
  }
  wave_moments(n, mu, m2);
  if (lane != 0) return;
  if (stats) {
    stats[c * 3 + 0] = n;
    stats[c * 3 + 1] = mu;
    stats[c * 3 + 2] = m2;
  }
  const float var = n > 0.f ? m2 / n : 0.f;
  if (mean_out) mean_out[c] = mu;
  if (invstd_out) invstd_out[c] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

// The affine form every BN kernel evaluates -- y = fma(x, sc, sh), sc = invstd*gamma, sh = fma(-mean, sc, beta) --
// spelled with explicit fmaf so that the backward kernels can recompute the ReLU mask (y > 0) from x bit-identically
// instead of reading the saved output.
__device__ __forceinline__ void bn_scale_shift(int c, const float* __restrict__ mean, const float* __restrict__ invstd,
                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                               float& sc, float& sh) {
  sc = invstd[c] * (gamma ? gamma[c] : 1.f);
  sh = fmaf(-mean[c], sc, beta ? beta[c] : 0.f);
}

// Channel of flat index i in a [B][C][HW] tensor: mul-hi division when the index fits 31 bits (fd_* built for HW, C),
// 64-bit division otherwise.
__device__ __forceinline__ int plane_channel(long long i, int HW, int C, const FastDiv& fd_hw, const FastDiv& fd_c) {
  if (i < (1ll << 31)) {
    const uint32_t q = fd_div((uint32_t)i, fd_hw);
    return (int)(q - fd_div(q, fd_c) * (uint32_t)C);
  }
  return (int)((i / HW) % C);
}

// Fused activation code of the BatchNorm kernels: 0 none, 1 ReLU, 2 GELU (erf).  The backward kernels recompute the
// activation's derivative from x with the forward's exact fma(x, sc, sh) (code 1 / 2 in `recompute`).
// y = (x - mean) * invstd * gamma + beta (+ residual)(relu / gelu)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, float* __restrict__ y,
                                                       long long n4, int C, int HW4, int relu, FastDiv fd_hw,
                                                       FastDiv fd_c) {
  const float4* x4 = (const float4*)x;
  const float4* r4 = (const float4*)residual;
  float4* y4 = (float4*)y;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = plane_channel(i, HW4, C, fd_hw, fd_c);
    float sc, sh;
    bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
    float4 v = x4[i];
    v.x = fmaf(v.x, sc, sh);
    v.y = fmaf(v.y, sc, sh);
    v.z = fmaf(v.z, sc, sh);
    v.w = fmaf(v.w, sc, sh);
    if (residual) {
      const float4 r = r4[i];
      v.x += r.x;
      v.y += r.y;
      v.z += r.z;
      v.w += r.w;
    }
    if (relu == 1) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    } else if (relu == 2) {
      v.x = gelu_f(v.x);
      v.y = gelu_f(v.y);
      v.z = gelu_f(v.z);
      v.w = gelu_f(v.w);
    }
    y4[i] = v;
  }
}
__global__ __launch_bounds__(256) void bn_apply_scalar_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ residual,
                                                              float* __restrict__ y, long long n, int C, int HW,
                                                              int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    float sc, sh;
    bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
    float v = fmaf(x[i], sc, sh);
    if (residual) v += residual[i];
    if (relu == 1) v = fmaxf(v, 0.f);
    else if (relu == 2) v = gelu_f(v);
    y[i] = v;
  }
}

// partial[c][blk] = (sum dy_m, sum dy_m * xhat); dy_m = dy masked by (out > 0) when out != null, or by the
// recomputed (fma(x, sc, sh) > 0) when recompute != 0 (BN + ReLU without residual: the output is never re-read).
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ out,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int recompute,
                                                             float* __restrict__ partial, int B, int C, int HW,
                                                             int NB, int ppb, int spp) {
  __shared__ float red[16];
  const int c = blockIdx.y, blk = blockIdx.x;
  const float mu = mean[c], is = invstd[c];
  float sc = 0.f, sh = 0.f;
  if (recompute) bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
  float s1 = 0.f, s2 = 0.f;
  if (ppb > 1 && (HW & 3) == 0) {
    bn_for_groups4(blk, c, B, C, HW, ppb, [&](size_t i4) {
      float4 g = ((const float4*)dy)[i4];
      const float4 xv = ((const float4*)x)[i4];
      if (recompute == 1) {
        g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
        g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
        g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
      } else if (recompute == 2) {
        g.x *= gelu_grad(fmaf(xv.x, sc, sh));
        g.y *= gelu_grad(fmaf(xv.y, sc, sh));
        g.z *= gelu_grad(fmaf(xv.z, sc, sh));
        g.w *= gelu_grad(fmaf(xv.w, sc, sh));
      } else if (out) {
        const float4 o = ((const float4*)out)[i4];
        g.x = o.x > 0.f ? g.x : 0.f;
        g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f;
        g.w = o.w > 0.f ? g.w : 0.f;
      }
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
    });
  } else
  bn_for_runs(blk, c, B, C, HW, ppb, spp, [&](size_t off, int len) {
    if ((len & 3) == 0 && ((off & 3) == 0)) {
      const float4* g4 = (const float4*)(dy + off);
      const float4* x4 = (const float4*)(x + off);
      const float4* o4 = out ? (const float4*)(out + off) : nullptr;
      for (int i = threadIdx.x; i < (len >> 2); i += 256) {
        float4 g = g4[i];
        const float4 xv = x4[i];
        if (recompute == 1) {
          g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
          g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
          g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
          g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
        } else if (recompute == 2) {
          g.x *= gelu_grad(fmaf(xv.x, sc, sh));
          g.y *= gelu_grad(fmaf(xv.y, sc, sh));
          g.z *= gelu_grad(fmaf(xv.z, sc, sh));
          g.w *= gelu_grad(fmaf(xv.w, sc, sh));
        } else if (o4) {
          const float4 o = o4[i];
          g.x = o.x > 0.f ? g.x : 0.f;
          g.y = o.y > 0.f ? g.y : 0.f;
          g.z = o.z > 0.f ? g.z : 0.f;
          g.w = o.w > 0.f ? g.w : 0.f;
        }
        s1 += (g.x + g.y) + (g.z + g.w);
        s2 += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
      }
    } else {
      for (int i = threadIdx.x; i < len; i += 256) {
        float g = dy[off + i];
        const float xv = x[off + i];
        if (recompute == 1) g = fmaf(xv, sc, sh) > 0.f ? g : 0.f;
        else if (recompute == 2) g *= gelu_grad(fmaf(xv, sc, sh));
        else if (out && !(out[off + i] > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * (xv - mu);
      }
    }
  });
  s2 *= is;
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    partial[((size_t)c * NB + blk) * 2 + 0] = s1;
    partial[((size_t)c * NB + blk) * 2 + 1] = s2;
  }
}

// sums[c] = (sum_dy, sum_dy_xhat) = sum over NB partials.
// Also emits the affine gradients: dgamma = sum dy*xhat, dbeta = sum dy (local sums, as SyncBN keeps them).
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int NB, int C, float* __restrict__ sums,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int i = 0; i < NB; ++i) {
    s1 += partial[((size_t)c * NB + i) * 2 + 0];
    s2 += partial[((size_t)c * NB + i) * 2 + 1];
  }
  sums[c * 2 + 0] = s1;
  sums[c * 2 + 1] = s2;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
}

// dx = gamma*invstd*(dy_m - s1/n - xhat*s2/n); dres = dy_m (optional).  n4 = elements / 4 when VEC.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ out,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int recompute,
                                                           const float* __restrict__ sums, float inv_count,
                                                           float* __restrict__ dx, float* __restrict__ dres,
                                                           long long n, int C, int HW, FastDiv fd_hw, FastDiv fd_c) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int c = plane_channel(i, HW, C, fd_hw, fd_c);
    const float is = invstd[c], mu = mean[c];
    float sc = 0.f, sh = 0.f;
    if (recompute) bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
    const float k = (gamma ? gamma[c] : 1.f) * is;
    const float a1 = sums[c * 2] * inv_count, a2 = sums[c * 2 + 1] * inv_count * is;
    if (VEC) {
      float4 g = ((const float4*)dy)[i];
      const float4 xv = ((const float4*)x)[i];
      if (recompute == 1) {
        g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
        g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
        g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
      } else if (recompute == 2) {
        g.x *= gelu_grad(fmaf(xv.x, sc, sh));
        g.y *= gelu_grad(fmaf(xv.y, sc, sh));
        g.z *= gelu_grad(fmaf(xv.z, sc, sh));
        g.w *= gelu_grad(fmaf(xv.w, sc, sh));
      } else if (out) {
        const float4 o = ((const float4*)out)[i];
        g.x = o.x > 0.f ? g.x : 0.f;
        g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f;
        g.w = o.w > 0.f ? g.w : 0.f;
      }
      float4 r;
      r.x = k * (g.x - a1 - (xv.x - mu) * a2);
      r.y = k * (g.y - a1 - (xv.y - mu) * a2);
      r.z = k * (g.z - a1 - (xv.z - mu) * a2);
      r.w = k * (g.w - a1 - (xv.w - mu) * a2);
      ((float4*)dx)[i] = r;
      if (dres) ((float4*)dres)[i] = g;
    } else {
      float g = dy[i];
      if (recompute == 1) g = fmaf(x[i], sc, sh) > 0.f ? g : 0.f;
      else if (recompute == 2) g *= gelu_grad(fmaf(x[i], sc, sh));
      else if (out && !(out[i] > 0.f)) g = 0.f;
      dx[i] = k * (g - a1 - (x[i] - mu) * a2);
      if (dres) dres[i] = g;
    }
  }
}

// The same apply pass CHANNEL-major and in REVERSE: workgroup (blk, c) of the grid walks slice NB-1-blk of channel C-1-c,
// i.e. the pass starts with what bn_bwd_partial_kernel (channels ascending) touched LAST -- those bytes of dy and x are
// still in the 256 MB Infinity Cache -- and its per-channel constants are wave-uniform (no index division per element).
__global__ __launch_bounds__(256) void bn_bwd_apply_cm_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              const float* __restrict__ out,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int recompute,
                                                              const float* __restrict__ sums, float inv_count,
                                                              float* __restrict__ dx, float* __restrict__ dres, int B, int C,
                                                              int HW, int NB, int ppb, int spp, int reverse,
                                                              const float* __restrict__ partial, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  const int c = reverse ? C - 1 - (int)blockIdx.y : (int)blockIdx.y, blk = reverse ? NB - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const float is = invstd[c], mu = mean[c];
  float sc = 0.f, sh = 0.f;
  if (recompute) bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
  const float k = (gamma ? gamma[c] : 1.f) * is;
  float s1, s2;
  if (partial) {
    // no finalize launch between the reduce and this pass: every workgroup adds up its channel's NB partial pairs itself
    // (wave-uniform addresses: scalar loads), in bn_bwd_finalize_kernel's order -- the same sums bit for bit -- and the
    // workgroup of slice 0 leaves the affine gradients
    s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < NB; ++i) {
      s1 += partial[((size_t)c * NB + i) * 2 + 0];
      s2 += partial[((size_t)c * NB + i) * 2 + 1];
    }
    if (blk == 0 && threadIdx.x == 0) {
      if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
      if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
    }
  } else {
    s1 = sums[c * 2];
    s2 = sums[c * 2 + 1];
  }
  const float a1 = s1 * inv_count, a2 = s2 * inv_count * is;
  bn_for_runs(blk, c, B, C, HW, ppb, spp, [&](size_t off, int len) {      // HW % 4 == 0: runs are float4-aligned
    const float4* g4 = (const float4*)(dy + off);
    const float4* x4 = (const float4*)(x + off);
    const float4* o4 = out ? (const float4*)(out + off) : nullptr;
    float4* d4 = (float4*)(dx + off);
    float4* r4 = dres ? (float4*)(dres + off) : nullptr;
    for (int i = threadIdx.x; i < (len >> 2); i += 256) {
      float4 g = g4[i];
      const float4 xv = x4[i];
      if (recompute == 1) {
        g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
        g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
        g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
      } else if (recompute == 2) {
        g.x *= gelu_grad(fmaf(xv.x, sc, sh));
        g.y *= gelu_grad(fmaf(xv.y, sc, sh));
        g.z *= gelu_grad(fmaf(xv.z, sc, sh));
        g.w *= gelu_grad(fmaf(xv.w, sc, sh));
      } else if (o4) {
        const float4 o = o4[i];
        g.x = o.x > 0.f ? g.x : 0.f;
        g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f;
        g.w = o.w > 0.f ? g.w : 0.f;
      }
      float4 r;
      r.x = k * (g.x - a1 - (xv.x - mu) * a2);
      r.y = k * (g.y - a1 - (xv.y - mu) * a2);
      r.z = k * (g.z - a1 - (xv.z - mu) * a2);
      r.w = k * (g.w - a1 - (xv.w - mu) * a2);
      d4[i] = r;
      if (r4) r4[i] = g;
    }
  });
}

// ---------------------------------------------------------------------------------------------
// Small layers (B*HW <= 16384 per channel: the 16x16 and 8x8 stages, everything at small per-GPU batches): ONE workgroup
// per channel does the whole BatchNorm of that channel in one launch -- the channel's 64 KB stay in L2 between the passes --
// where the general path needs a moments (or conv-epilogue) pass, a finalize launch and an apply launch (forward) or
// partial + finalize + apply (backward): at these sizes every launch is ~5-9 us of latency for ~1 us of work.
// ---------------------------------------------------------------------------------------------
// Forward.  partial != null: merge the channel's NB (count, mean, M2) triples exactly as bn_finalize_wave_kernel does;
// else take the moments from x (sum, then centred sum of squares).  Then y = fma(x, sc, sh) (+ residual)(relu).
__device__ __forceinline__ void bn_fwd_channel_body(const float* __restrict__ x, const float* __restrict__ partial,
                                                    long long sc_stride, long long sb_stride, int NB,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ residual, float* __restrict__ y,
                                                    float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    int B, int C, int HW4, float eps, float momentum, int relu,
                                                    float* red, float* s_stat) {
  const int c = blockIdx.x;
  const int total = B * HW4;                       // float4 groups of this channel
  const float4* x4 = (const float4*)x;
  auto idx4 = [&](int e) {
    const int b = e / HW4, i = e - b * HW4;
    return ((size_t)b * C + c) * HW4 + i;
  };
  float n, mu, m2;
  if (partial) {
    n = 0.f, mu = 0.f, m2 = 0.f;
    if (threadIdx.x < 64) {
      for (int i = threadIdx.x; i < NB; i += 64) {
        const float* p = partial + (size_t)c * sc_stride + (size_t)i * sb_stride;
        moments_merge(n, mu, m2, p[0], p[1], p[2]);
      }
      wave_moments(n, mu, m2);
    }
  } else {
    float sum = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
      const float4 v = x4[idx4(e)];
      sum += (v.x + v.y) + (v.z + v.w);
    }
    n = (float)total * 4.f;
    mu = block_sum(sum, red) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
      const float4 v = x4[idx4(e)];
      const float a = v.x - mu, b2 = v.y - mu, c2 = v.z - mu, d = v.w - mu;
      q += (a * a + b2 * b2) + (c2 * c2 + d * d);
    }
    m2 = block_sum(q, red);
  }
  if (threadIdx.x == 0) {
    const float var = n > 0.f ? m2 / n : 0.f;
    const float is = 1.0f / sqrtf(var + eps);
    s_stat[0] = mu;
    s_stat[1] = is;
    mean_out[c] = mu;
    invstd_out[c] = is;
    if (running_mean) {
      const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
  }
  __syncthreads();
  const float scv = s_stat[1] * (gamma ? gamma[c] : 1.f);
  const float shv = fmaf(-s_stat[0], scv, beta ? beta[c] : 0.f);      // == bn_scale_shift: the backward recomputes this
  const float4* r4 = (const float4*)residual;
  float4* y4 = (float4*)y;
  for (int e = threadIdx.x; e < total; e += 256) {
    const size_t i = idx4(e);
    float4 v = x4[i];
    v.x = fmaf(v.x, scv, shv);
    v.y = fmaf(v.y, scv, shv);
    v.z = fmaf(v.z, scv, shv);
    v.w = fmaf(v.w, scv, shv);
    if (residual) {
      const float4 r = r4[i];
      v.x += r.x;
      v.y += r.y;
      v.z += r.z;
      v.w += r.w;
    }
    if (relu == 1) {
      v.x = fmaxf(v.x, 0.f);
      v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f);
      v.w = fmaxf(v.w, 0.f);
    } else if (relu == 2) {
      v.x = gelu_f(v.x);
      v.y = gelu_f(v.y);
      v.z = gelu_f(v.z);
      v.w = gelu_f(v.w);
    }
    y4[i] = v;
  }
}

__global__ __launch_bounds__(256) void bn_fwd_channel_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ partial, long long sc_stride,
                                                             long long sb_stride, int NB,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ residual, float* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, int B, int C, int HW4,
                                                             float eps, float momentum, int relu) {
  __shared__ float red[16];
  __shared__ float s_stat[2];
  bn_fwd_channel_body(x, partial, sc_stride, sb_stride, NB, gamma, beta, residual, y, mean_out, invstd_out, running_mean,
                      running_var, B, C, HW4, eps, momentum, relu, red, s_stat);
}

// The independent passes concatenated in one batch (functional.bn_segments: source / target / clip frames), one launch:
// the workgroup of a channel walks the segments in order -- statistics per segment, running statistics updated segment
// by segment exactly as separate launches would (same thread, same order).
struct BnSegs {
  int S;
  int b0[16], bs[16], poff[16], nb[16];  // first frame, frames, offset (in triples) into the channel's partials, triples
                                         // (16: the time steps of a TGCN clip embedded as one pass)
};
__global__ __launch_bounds__(256) void bn_fwd_channel_segs_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ partial, long long sc_stride,
                                                                  long long sb_stride, BnSegs sg,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  const float* __restrict__ residual,
                                                                  float* __restrict__ y, float* __restrict__ mean_out,
                                                                  float* __restrict__ invstd_out,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, int C, int HW4, float eps,
                                                                  float momentum, int relu, long long seg_pstride) {
  __shared__ float red[16];
  __shared__ float s_stat[2];
  for (int s = 0; s < sg.S; ++s) {
    const size_t off = (size_t)sg.b0[s] * C * HW4 * 4;
    // (seg_pstride: SyncBN -- segment s reads the [world] gathered triples of its channel, seg_pstride floats behind segment s - 1's)
    bn_fwd_channel_body(x + off, partial ? partial + (size_t)sg.poff[s] * sb_stride + (size_t)s * seg_pstride : nullptr, sc_stride, sb_stride,
                        sg.nb[s], gamma, beta, residual ? residual + off : nullptr, y + off, mean_out + (size_t)s * C,
                        invstd_out + (size_t)s * C, running_mean, running_var, sg.bs[s], C, HW4, eps, momentum, relu, red,
                        s_stat);
    __syncthreads();      // red / s_stat are reused by the next segment
  }
}

// Backward: s1 = sum dy_m, s2 = sum dy_m * xhat over the channel, then dx = gamma*invstd*(dy_m - s1/n - xhat*s2/n) and
// (optionally) dres = dy_m; dgamma (+)= s2, dbeta (+)= s1.  Same masking rule as bn_bwd_partial_kernel / bn_bwd_apply_kernel.
template <bool REDUCE_ONLY>
__device__ __forceinline__ void bn_bwd_channel_body(const float* __restrict__ dy, const float* __restrict__ x,
                                                    const float* __restrict__ out, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int recompute,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                    float inv_count, float* __restrict__ dx, float* __restrict__ dres,
                                                    int B, int C, int HW4, float* red) {
  const int c = blockIdx.x;
  const int total = B * HW4;
  const float mu = mean[c], is = invstd[c];
  float sc = 0.f, sh = 0.f;
  if (recompute) bn_scale_shift(c, mean, invstd, gamma, beta, sc, sh);
  auto idx4 = [&](int e) {
    const int b = e / HW4, i = e - b * HW4;
    return ((size_t)b * C + c) * HW4 + i;
  };
  auto masked = [&](size_t i, float4& g, float4& xv) {
    g = ((const float4*)dy)[i];
    xv = ((const float4*)x)[i];
    if (recompute == 1) {
      g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
      g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
      g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
      g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
    } else if (recompute == 2) {
      g.x *= gelu_grad(fmaf(xv.x, sc, sh));
      g.y *= gelu_grad(fmaf(xv.y, sc, sh));
      g.z *= gelu_grad(fmaf(xv.z, sc, sh));
      g.w *= gelu_grad(fmaf(xv.w, sc, sh));
    } else if (out) {
      const float4 o = ((const float4*)out)[i];
      g.x = o.x > 0.f ? g.x : 0.f;
      g.y = o.y > 0.f ? g.y : 0.f;
      g.z = o.z > 0.f ? g.z : 0.f;
      g.w = o.w > 0.f ? g.w : 0.f;
    }
  };
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < total; e += 256) {
    float4 g, xv;
    masked(idx4(e), g, xv);
    s1 += (g.x + g.y) + (g.z + g.w);
    s2 += (g.x * (xv.x - mu) + g.y * (xv.y - mu)) + (g.z * (xv.z - mu) + g.w * (xv.w - mu));
  }
  s2 *= is;
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
    if (REDUCE_ONLY) {       // `dx` is the [C][2] sums buffer here
      dx[c * 2 + 0] = s1;
      dx[c * 2 + 1] = s2;
    }
  }
  if (REDUCE_ONLY) return;
  const float k = (gamma ? gamma[c] : 1.f) * is;
  const float a1 = s1 * inv_count, a2 = s2 * inv_count * is;
  for (int e = threadIdx.x; e < total; e += 256) {
    const size_t i = idx4(e);
    float4 g, xv;
    masked(i, g, xv);
    float4 r;
    r.x = k * (g.x - a1 - (xv.x - mu) * a2);
    r.y = k * (g.y - a1 - (xv.y - mu) * a2);
    r.z = k * (g.z - a1 - (xv.z - mu) * a2);
    r.w = k * (g.w - a1 - (xv.w - mu) * a2);
    ((float4*)dx)[i] = r;
    if (dres) ((float4*)dres)[i] = g;
  }
}

template <bool REDUCE_ONLY>
__global__ __launch_bounds__(256) void bn_bwd_channel_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ out,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int recompute,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate, float inv_count, float* __restrict__ dx,
                                                             float* __restrict__ dres, int B, int C, int HW4) {
  __shared__ float red[16];
  bn_bwd_channel_body<REDUCE_ONLY>(dy, x, out, mean, invstd, gamma, beta, recompute, dgamma, dbeta, accumulate, inv_count,
                                   dx, dres, B, C, HW4, red);
}

// all segments of a concatenated batch in one launch (see bn_fwd_channel_segs_kernel); dgamma / dbeta take the segments'
// sums in order, the first one overwriting unless `accumulate`
__global__ __launch_bounds__(256) void bn_bwd_channel_segs_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int recompute,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  int accumulate, BnSegs sg, float* __restrict__ dx,
                                                                  float* __restrict__ dres, int C, int HW4) {
  __shared__ float red[16];
  for (int s = 0; s < sg.S; ++s) {
    const size_t off = (size_t)sg.b0[s] * C * HW4 * 4;
    bn_bwd_channel_body<false>(dy + off, x + off, out ? out + off : nullptr, mean + (size_t)s * C, invstd + (size_t)s * C,
                               gamma, beta, recompute, dgamma, dbeta, (accumulate || s > 0) ? 1 : 0,
                               1.0f / ((float)sg.bs[s] * (float)HW4 * 4.f), dx + off, dres ? dres + off : nullptr, sg.bs[s],
                               C, HW4, red);
    __syncthreads();
  }
}

// ---- SyncBN over the same segments (round 6): the two halves of each direction, all segments per launch.  The statistics cross
// the ranks between the halves (functional._BatchNormFn), so a layer is local finalize -> all-gather -> merge + apply forward and
// reduce -> all-reduce -> apply backward: 3 + 3 launches for S segments where the per-segment calls took (2 S + 1) + (2 S + 1).
// Every segment's arithmetic is the per-segment kernel's (same merge order, same expressions): the results are the same bits.
struct BnSegScale {
  float inv_count[16];      // 1 / (frames * HW * world) of segment s, as the host computes it for ge_bn_bwd_apply
};
// stats[s][c] = (n, mean, M2) of segment s, channel c: the merge of its nb[s] conv-epilogue triples -- in bn_finalize_kernel's
// order when nb <= 16, bn_finalize_wave_kernel's otherwise (what ge_bn_finalize picks for a segment on its own).  One wave per
// (channel, segment).
__global__ __launch_bounds__(256) void bn_finalize_segs_kernel(const float* __restrict__ partial, long long sc, long long sb,
                                                               BnSegs sg, int C, float* __restrict__ stats) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), s = blockIdx.y;
  if (c >= C) return;
  const int lane = threadIdx.x & 63, NB = sg.nb[s];
  const float* pc = partial + (size_t)c * sc + (size_t)sg.poff[s] * sb;
  float n = 0.f, mu = 0.f, m2 = 0.f;
  if (NB > 16) {
    for (int i0 = lane; i0 < NB; i0 += 64 * 8) {
      float tn[8], tm[8], tq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        const float* p = pc + (size_t)(i < NB ? i : lane) * sb;
        tn[u] = p[0];
        tm[u] = p[1];
        tq[u] = p[2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 64 * u < NB) moments_merge(n, mu, m2, tn[u], tm[u], tq[u]);
    }
    wave_moments(n, mu, m2);
  } else if (lane == 0) {
    for (int i = 0; i < NB; ++i) {
      const float* p = pc + (size_t)i * sb;
      moments_merge(n, mu, m2, p[0], p[1], p[2]);
    }
  }
  if (lane != 0) return;
  float* o = stats + ((size_t)s * C + c) * 3;
  o[0] = n;
  o[1] = mu;
  o[2] = m2;
}

// The same buffer for layers whose convolution left no moments (split-K layers at small batches, the stem): (n, mean, M2) of every
// segment from x itself, one workgroup per channel -- sum, then the centred sum of squares, as bn_fwd_channel_body takes them.
__global__ __launch_bounds__(256) void bn_stats_channel_segs_kernel(const float* __restrict__ x, BnSegs sg, int C, int HW4,
                                                                    float* __restrict__ stats) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  for (int s = 0; s < sg.S; ++s) {
    const float4* x4 = (const float4*)x + (size_t)sg.b0[s] * C * HW4;
    const int total = sg.bs[s] * HW4;
    auto idx4 = [&](int e) {
      const int b = e / HW4, i = e - b * HW4;
      return ((size_t)b * C + c) * HW4 + i;
    };
    float sum = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
      const float4 v = x4[idx4(e)];
      sum += (v.x + v.y) + (v.z + v.w);
    }
    const float n = (float)total * 4.f;
    const float mu = block_sum(sum, red) / n;
    float q = 0.f;
    for (int e = threadIdx.x; e < total; e += 256) {
      const float4 v = x4[idx4(e)];
      const float a = v.x - mu, b2 = v.y - mu, c2 = v.z - mu, d = v.w - mu;
      q += (a * a + b2 * b2) + (c2 * c2 + d * d);
    }
    const float m2 = block_sum(q, red);
    if (threadIdx.x == 0) {
      float* o = stats + ((size_t)s * C + c) * 3;
      o[0] = n;
      o[1] = mu;
      o[2] = m2;
    }
    __syncthreads();
  }
}

// sums[s][c] = (sum dy_m, sum dy_m * xhat) of every segment (bn_bwd_channel_kernel<true> per segment), dgamma / dbeta (+)= in order
__global__ __launch_bounds__(256) void bn_bwd_reduce_channel_segs_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                         const float* __restrict__ out,
                                                                         const float* __restrict__ mean,
                                                                         const float* __restrict__ invstd,
                                                                         const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, int recompute,
                                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                         int accumulate, BnSegs sg, float* __restrict__ sums, int C,
                                                                         int HW4) {
  __shared__ float red[16];
  for (int s = 0; s < sg.S; ++s) {
    const size_t off = (size_t)sg.b0[s] * C * HW4 * 4;
    bn_bwd_channel_body<true>(dy + off, x + off, out ? out + off : nullptr, mean + (size_t)s * C, invstd + (size_t)s * C, gamma,
                              beta, recompute, dgamma, dbeta, (accumulate || s > 0) ? 1 : 0, 0.f, sums + (size_t)s * C * 2, nullptr,
                              sg.bs[s], C, HW4, red);
    __syncthreads();
  }
}

// dx (and dres) of every segment from the (all-reduced) sums[s][c]: bn_bwd_apply_cm_kernel's expressions, one workgroup per channel
__global__ __launch_bounds__(256) void bn_bwd_apply_channel_segs_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                        const float* __restrict__ out,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd,
                                                                        const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta, int recompute,
                                                                        const float* __restrict__ sums, BnSegs sg, BnSegScale sc_,
                                                                        float* __restrict__ dx, float* __restrict__ dres, int C,
                                                                        int HW4) {
  const int c = blockIdx.x;
  for (int s = 0; s < sg.S; ++s) {
    const size_t off4 = (size_t)sg.b0[s] * C * HW4;
    const float* mean_s = mean + (size_t)s * C;
    const float* invstd_s = invstd + (size_t)s * C;
    const float is = invstd_s[c], mu = mean_s[c];
    float sc = 0.f, sh = 0.f;
    if (recompute) bn_scale_shift(c, mean_s, invstd_s, gamma, beta, sc, sh);
    const float k = (gamma ? gamma[c] : 1.f) * is;
    const float s1 = sums[((size_t)s * C + c) * 2], s2 = sums[((size_t)s * C + c) * 2 + 1];
    const float a1 = s1 * sc_.inv_count[s], a2 = s2 * sc_.inv_count[s] * is;
    const float4* g4 = (const float4*)dy + off4;
    const float4* x4 = (const float4*)x + off4;
    const float4* o4 = out ? (const float4*)out + off4 : nullptr;
    float4* d4 = (float4*)dx + off4;
    float4* r4 = dres ? (float4*)dres + off4 : nullptr;
    const int total = sg.bs[s] * HW4;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int b = e / HW4, j = e - b * HW4;
      const size_t i = ((size_t)b * C + c) * HW4 + j;
      float4 g = g4[i];
      const float4 xv = x4[i];
      if (recompute == 1) {
        g.x = fmaf(xv.x, sc, sh) > 0.f ? g.x : 0.f;
        g.y = fmaf(xv.y, sc, sh) > 0.f ? g.y : 0.f;
        g.z = fmaf(xv.z, sc, sh) > 0.f ? g.z : 0.f;
        g.w = fmaf(xv.w, sc, sh) > 0.f ? g.w : 0.f;
      } else if (recompute == 2) {
        g.x *= gelu_grad(fmaf(xv.x, sc, sh));
        g.y *= gelu_grad(fmaf(xv.y, sc, sh));
        g.z *= gelu_grad(fmaf(xv.z, sc, sh));
        g.w *= gelu_grad(fmaf(xv.w, sc, sh));
      } else if (o4) {
        const float4 o = o4[i];
        g.x = o.x > 0.f ? g.x : 0.f;
        g.y = o.y > 0.f ? g.y : 0.f;
        g.z = o.z > 0.f ? g.z : 0.f;
        g.w = o.w > 0.f ? g.w : 0.f;
      }
      float4 r;
      r.x = k * (g.x - a1 - (xv.x - mu) * a2);
      r.y = k * (g.y - a1 - (xv.y - mu) * a2);
      r.z = k * (g.z - a1 - (xv.z - mu) * a2);
      r.w = k * (g.w - a1 - (xv.w - mu) * a2);
      d4[i] = r;
      if (r4) r4[i] = g;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm: one workgroup per (b, group); the group's Cg*HW floats are contiguous in NCHW.
// ---------------------------------------------------------------------------------------------
__global__ void gn_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean_out,
                              float* __restrict__ invstd_out, int C, int G, int HW, float eps, int relu) {
  __shared__ float red[16];
  const int bg = blockIdx.x;
  const int g = bg % G, b = bg / G;
  const int Cg = C / G;
  const long long L = (long long)Cg * HW;
  const size_t base = ((size_t)b * C + (size_t)g * Cg) * HW;
  const float* xp = x + base;
  float s = 0.f;
  for (long long i = threadIdx.x; i < L; i += blockDim.x) s += xp[i];
  const float mu = block_sum(s, red) / (float)L;
  float q = 0.f;
  for (long long i = threadIdx.x; i < L; i += blockDim.x) {
    const float d = xp[i] - mu;
    q += d * d;
  }
  const float var = block_sum(q, red) / (float)L;
  const float is = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[bg] = mu;
    invstd_out[bg] = is;
  }
  float* yp = y + base;
  for (long long i = threadIdx.x; i < L; i += blockDim.x) {
    const int c = g * Cg + (int)(i / HW);
    float v = (xp[i] - mu) * is;
    v = v * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
    if (relu) v = fmaxf(v, 0.f);
    yp[i] = v;
  }
}

// dgamma_part[b][c] = sum_hw dy_m*xhat, dbeta_part[b][c] = sum_hw dy_m; dx per group.
__global__ void gn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ out,
                              const float* __restrict__ gamma, const float* __restrict__ mean,
                              const float* __restrict__ invstd, float* __restrict__ dx,
                              float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int C, int G, int HW) {
  __shared__ float red[16];
  const int bg = blockIdx.x;
  const int g = bg % G, b = bg / G;
  const int Cg = C / G;
  const long long L = (long long)Cg * HW;
  const size_t base = ((size_t)b * C + (size_t)g * Cg) * HW;
  const float mu = mean[bg], is = invstd[bg];
  float s1 = 0.f, s2 = 0.f;  // sum dy*gamma, sum dy*gamma*xhat over the group
  for (int cc = 0; cc < Cg; ++cc) {
    const int c = g * Cg + cc;
    const size_t off = base + (size_t)cc * HW;
    float a = 0.f, bsum = 0.f;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
      float gdy = dy[off + i];
      if (out && !(out[off + i] > 0.f)) gdy = 0.f;
      a += gdy * (x[off + i] - mu) * is;
      bsum += gdy;
    }
    a = block_sum(a, red);
    bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) {
      dgamma_part[(size_t)b * C + c] = a;
      dbeta_part[(size_t)b * C + c] = bsum;
    }
    const float gm = gamma ? gamma[c] : 1.f;
    s1 += gm * bsum;
    s2 += gm * a;
  }
  const float invL = 1.f / (float)L;
  for (long long i = threadIdx.x; i < L; i += blockDim.x) {
    const int c = g * Cg + (int)(i / HW);
    float gdy = dy[base + i];
    if (out && !(out[base + i] > 0.f)) gdy = 0.f;
    const float xh = (x[base + i] - mu) * is;
    dx[base + i] = is * (gdy * (gamma ? gamma[c] : 1.f) - s1 * invL - xh * s2 * invL);
  }
}

// Register-resident GroupNorm for groups of up to NT*VPT*4 floats whose planes are multiples of 256 (so a wave's
// 64 float4 lie in one channel): the group is read from HBM ONCE (16-B loads), statistics and output come from
// registers.  Same two-pass mean / variance arithmetic as the streaming kernels above.
template <int NT, int VPT>
__global__ __launch_bounds__(NT) void gn_fwd_reg_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                        int C, int G, int HW, float eps, int relu) {
  __shared__ float red[16];
  const int bg = blockIdx.x;
  const int g = bg % G, b = bg / G;
  const int Cg = C / G;
  const int L4 = Cg * HW / 4;
  const size_t base = ((size_t)b * C + (size_t)g * Cg) * HW;
  const float4* x4 = (const float4*)(x + base);
  float4 v[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * NT;
    v[k] = i < L4 ? x4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  const float mu = block_sum(s, red) / (float)(L4 * 4);
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if (threadIdx.x + k * NT < L4) {
      const float a = v[k].x - mu, bq = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
      q += (a * a + bq * bq) + (c * c + d * d);
    }
  }
  const float var = block_sum(q, red) / (float)(L4 * 4);
  const float is = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[bg] = mu;
    invstd_out[bg] = is;
  }
  float4* y4 = (float4*)(y + base);
  const int hw4 = HW / 4;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < L4) {
      const int c = g * Cg + i / hw4;
      const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
      float4 o;
      o.x = (v[k].x - mu) * is * gm + bt;
      o.y = (v[k].y - mu) * is * gm + bt;
      o.z = (v[k].z - mu) * is * gm + bt;
      o.w = (v[k].w - mu) * is * gm + bt;
      if (relu) {
        o.x = fmaxf(o.x, 0.f);
        o.y = fmaxf(o.y, 0.f);
        o.z = fmaxf(o.z, 0.f);
        o.w = fmaxf(o.w, 0.f);
      }
      y4[i] = o;
    }
  }
}

// Backward, register-resident: dy, x (and the saved output for the ReLU mask) are read once; per-channel
// dgamma/dbeta partials are wave-reduced (a wave's float4s share a channel) into one LDS slot per (k, wave) -- the slots
// of a channel are consecutive -- and summed per channel in slot order: no atomics, bit-identical run to run.
template <int NT, int VPT>
__global__ __launch_bounds__(NT) void gn_bwd_reg_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ out,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ dx,
                                                        float* __restrict__ dgamma_part,
                                                        float* __restrict__ dbeta_part, int C, int G, int HW) {
  constexpr int NSLOT = VPT * (NT / 64);
  __shared__ float sgn[2 * NSLOT];   // [NSLOT] dgamma partials, [NSLOT] dbeta partials; slot = k * (NT / 64) + wave
  __shared__ float red[16];
  const int bg = blockIdx.x;
  const int g = bg % G, b = bg / G;
  const int Cg = C / G;
  const int L4 = Cg * HW / 4, hw4 = HW / 4;
  const size_t base = ((size_t)b * C + (size_t)g * Cg) * HW;
  const float mu = mean[bg], is = invstd[bg];
  const float4* g4 = (const float4*)(dy + base);
  const float4* x4 = (const float4*)(x + base);
  const float4* o4 = out ? (const float4*)(out + base) : nullptr;
  float4 gd[VPT], xh[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * NT;
    const bool ok = i < L4;
    gd[k] = ok ? g4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[k] = ok ? x4[i] : make_float4(mu, mu, mu, mu);
    if (o4 && ok) {
      const float4 o = o4[i];
      gd[k].x = o.x > 0.f ? gd[k].x : 0.f;
      gd[k].y = o.y > 0.f ? gd[k].y : 0.f;
      gd[k].z = o.z > 0.f ? gd[k].z : 0.f;
      gd[k].w = o.w > 0.f ? gd[k].w : 0.f;
    }
  }
  float s1 = 0.f, s2 = 0.f;   // sum dy*gamma, sum dy*gamma*xhat over the group (this thread's share)
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    xh[k].x = (xh[k].x - mu) * is;
    xh[k].y = (xh[k].y - mu) * is;
    xh[k].z = (xh[k].z - mu) * is;
    xh[k].w = (xh[k].w - mu) * is;
    const int i = threadIdx.x + k * NT;
    const int cc = i < L4 ? i / hw4 : 0;          // wave-uniform: hw4 is a multiple of 64
    float a = (gd[k].x * xh[k].x + gd[k].y * xh[k].y) + (gd[k].z * xh[k].z + gd[k].w * xh[k].w);
    float bs = (gd[k].x + gd[k].y) + (gd[k].z + gd[k].w);
    const float gm = gamma ? gamma[g * Cg + cc] : 1.f;
    s1 += gm * bs;
    s2 += gm * a;
    a = wave_sum(a);
    bs = wave_sum(bs);
    if ((threadIdx.x & 63) == 0) {          // slot's first float4 is i: its channel is i / hw4 (slots past L4 hold 0)
      const int slot = k * (NT / 64) + (threadIdx.x >> 6);
      sgn[slot] = a;
      sgn[NSLOT + slot] = bs;
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  __syncthreads();
  const int spc = hw4 / 64;                 // slots per channel (hw4 is a multiple of 64)
  for (int c = threadIdx.x; c < Cg; c += NT) {
    float da = 0.f, db = 0.f;
    for (int q = 0; q < spc; ++q) {
      da += sgn[c * spc + q];
      db += sgn[NSLOT + c * spc + q];
    }
    dgamma_part[(size_t)b * C + g * Cg + c] = da;
    dbeta_part[(size_t)b * C + g * Cg + c] = db;
  }
  const float invL = 1.f / (float)(L4 * 4);
  const float k1 = s1 * invL, k2 = s2 * invL;
  float4* d4 = (float4*)(dx + base);
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < L4) {
      const float gm = gamma ? gamma[g * Cg + i / hw4] : 1.f;
      float4 o;
      o.x = is * (gd[k].x * gm - k1 - xh[k].x * k2);
      o.y = is * (gd[k].y * gm - k1 - xh[k].y * k2);
      o.z = is * (gd[k].z * gm - k1 - xh[k].z * k2);
      o.w = is * (gd[k].w * gm - k1 - xh[k].w * k2);
      d4[i] = o;
    }
  }
}

// out[c] = sum_r in[r][c].  Workgroup = 64 columns x 16 row groups, four independent partial sums per thread.
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ in, float* __restrict__ out, int R,
                                                      int C, int accumulate = 0) {
  __shared__ float red[16][65];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int r = rg;
    for (; r + 48 < R; r += 64) {
      s0 += in[(size_t)r * C + c];
      s1 += in[(size_t)(r + 16) * C + c];
      s2 += in[(size_t)(r + 32) * C + c];
      s3 += in[(size_t)(r + 48) * C + c];
    }
    for (; r < R; r += 16) s0 += in[(size_t)r * C + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// two column sums of the same shape in one launch (a norm layer's d gamma and d beta partials): blockIdx.y picks the pair;
// the arithmetic per pair is colsum_kernel's
__global__ __launch_bounds__(1024) void colsum2_kernel(const float* __restrict__ in0, float* __restrict__ out0,
                                                       const float* __restrict__ in1, float* __restrict__ out1, int R, int C,
                                                       int accumulate) {
  __shared__ float red[16][65];
  const float* __restrict__ in = blockIdx.y ? in1 : in0;
  float* __restrict__ out = blockIdx.y ? out1 : out0;
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    int r = rg;
    for (; r + 48 < R; r += 64) {
      s0 += in[(size_t)r * C + c];
      s1 += in[(size_t)(r + 16) * C + c];
      s2 += in[(size_t)(r + 32) * C + c];
      s3 += in[(size_t)(r + 48) * C + c];
    }
    for (; r < R; r += 16) s0 += in[(size_t)r * C + c];
  }
  red[rg][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (D), one wave per row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                     int R, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xp = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xp[i];
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) {
    const float d = xp[i] - mu;
    q += d * d;
  }
  const float is = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) {
    mean_out[row] = mu;
    invstd_out[row] = is;
  }
  float* yp = y + (size_t)row * D;
  for (int i = lane; i < D; i += 64) {
    float v = (xp[i] - mu) * is;
    if (gamma) v = v * gamma[i] + beta[i];
    yp[i] = v;
  }
}

// dx per row; dgamma_part/dbeta_part[blk][D] are per-workgroup column partials (only when gamma != null).
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, float* __restrict__ dx,
                                                     float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                     int R, int D, int rows_per_block) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(r0 + rows_per_block, R);
  for (int row = r0 + w; row < r1; row += 4) {
    const float* xp = x + (size_t)row * D;
    const float* gp = dy + (size_t)row * D;
    const float mu = mean[row], is = invstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < D; i += 64) {
      const float g = gp[i] * (gamma ? gamma[i] : 1.f);
      s1 += g;
      s2 += g * (xp[i] - mu) * is;
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    float* dp = dx + (size_t)row * D;
    for (int i = lane; i < D; i += 64) {
      const float xh = (xp[i] - mu) * is;
      dp[i] = is * (gp[i] * (gamma ? gamma[i] : 1.f) - s1 - xh * s2);
    }
  }
  if (dgamma_part) {
    // column partials over this block's rows: thread t owns columns t, t+256, ...
    for (int i = threadIdx.x; i < D; i += 256) {
      float a = 0.f, b = 0.f;
      for (int row = r0; row < r1; ++row) {
        const float g = dy[(size_t)row * D + i];
        a += g * (x[(size_t)row * D + i] - mean[row]) * invstd[row];
        b += g;
      }
      dgamma_part[(size_t)blockIdx.x * D + i] = a;
      dbeta_part[(size_t)blockIdx.x * D + i] = b;
    }
  }
}


// Wide rows (D >= 4096, e.g. the whole affinity matrix as one row): one workgroup per row, no affine.
// Wide rows (D >= 4096, no affine: InstanceNorm over GModule's N1 x N2 affinity matrix, ONE row of ~5e4 elements): a
// 1024-thread workgroup per row with eight independent 16-byte loads in flight per thread and pass (the 256-thread form
// walked the row with one dependent 4-byte load at a time: 127 us forward / 98 us backward for 55 696 elements).
template <int VEC>
__global__ __launch_bounds__(1024) void ln_fwd_wide_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                           int D, float eps) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const float* xp = x + (size_t)row * D;
  float* yp = y + (size_t)row * D;
  const int nthr = blockDim.x;
  float s = 0.f;
  if (VEC == 4) {
    const float4* x4 = (const float4*)xp;
    const int n4 = D / 4;
    float s0 = 0.f, s1 = 0.f;
    int i = threadIdx.x;
    for (; i + nthr < n4; i += 2 * nthr) {
      const float4 a = x4[i], c = x4[i + nthr];
      s0 += (a.x + a.y) + (a.z + a.w);
      s1 += (c.x + c.y) + (c.z + c.w);
    }
    if (i < n4) {
      const float4 a = x4[i];
      s0 += (a.x + a.y) + (a.z + a.w);
    }
    s = s0 + s1;
  } else {
    for (int i = threadIdx.x; i < D; i += nthr) s += xp[i];
  }
  const float mu = block_sum(s, red) / (float)D;
  float q = 0.f;
  if (VEC == 4) {
    const float4* x4 = (const float4*)xp;
    const int n4 = D / 4;
    float q0 = 0.f, q1 = 0.f;
    int i = threadIdx.x;
    for (; i + nthr < n4; i += 2 * nthr) {
      const float4 a = x4[i], c = x4[i + nthr];
      q0 += ((a.x - mu) * (a.x - mu) + (a.y - mu) * (a.y - mu)) + ((a.z - mu) * (a.z - mu) + (a.w - mu) * (a.w - mu));
      q1 += ((c.x - mu) * (c.x - mu) + (c.y - mu) * (c.y - mu)) + ((c.z - mu) * (c.z - mu) + (c.w - mu) * (c.w - mu));
    }
    if (i < n4) {
      const float4 a = x4[i];
      q0 += ((a.x - mu) * (a.x - mu) + (a.y - mu) * (a.y - mu)) + ((a.z - mu) * (a.z - mu) + (a.w - mu) * (a.w - mu));
    }
    q = q0 + q1;
  } else {
    for (int i = threadIdx.x; i < D; i += nthr) {
      const float d = xp[i] - mu;
      q += d * d;
    }
  }
  const float is = 1.0f / sqrtf(block_sum(q, red) / (float)D + eps);
  if (threadIdx.x == 0) {
    mean_out[row] = mu;
    invstd_out[row] = is;
  }
  if (VEC == 4) {
    const float4* x4 = (const float4*)xp;
    float4* y4 = (float4*)yp;
    for (int i = threadIdx.x; i < D / 4; i += nthr) {
      const float4 a = x4[i];
      y4[i] = make_float4((a.x - mu) * is, (a.y - mu) * is, (a.z - mu) * is, (a.w - mu) * is);
    }
  } else {
    for (int i = threadIdx.x; i < D; i += nthr) yp[i] = (xp[i] - mu) * is;
  }
}
template <int VEC>
__global__ __launch_bounds__(1024) void ln_bwd_wide_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float* __restrict__ dx,
                                                           int D) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const float* xp = x + (size_t)row * D;
  const float* gp = dy + (size_t)row * D;
  float* dp = dx + (size_t)row * D;
  const float mu = mean[row], is = invstd[row];
  const int nthr = blockDim.x;
  float s1 = 0.f, s2 = 0.f;
  if (VEC == 4) {
    const float4 *x4 = (const float4*)xp, *g4 = (const float4*)gp;
    for (int i = threadIdx.x; i < D / 4; i += nthr) {
      const float4 a = x4[i], g = g4[i];
      s1 += (g.x + g.y) + (g.z + g.w);
      s2 += (g.x * (a.x - mu) + g.y * (a.y - mu)) + (g.z * (a.z - mu) + g.w * (a.w - mu));
    }
    s2 *= is;
  } else {
    for (int i = threadIdx.x; i < D; i += nthr) {
      s1 += gp[i];
      s2 += gp[i] * (xp[i] - mu) * is;
    }
  }
  s1 = block_sum(s1, red) / (float)D;
  s2 = block_sum(s2, red) / (float)D;
  if (VEC == 4) {
    const float4 *x4 = (const float4*)xp, *g4 = (const float4*)gp;
    float4* d4 = (float4*)dp;
    for (int i = threadIdx.x; i < D / 4; i += nthr) {
      const float4 a = x4[i], g = g4[i];
      d4[i] = make_float4(is * (g.x - s1 - (a.x - mu) * is * s2), is * (g.y - s1 - (a.y - mu) * is * s2),
                          is * (g.z - s1 - (a.z - mu) * is * s2), is * (g.w - s1 - (a.w - mu) * is * s2));
    }
  } else {
    for (int i = threadIdx.x; i < D; i += nthr) dp[i] = is * (gp[i] - s1 - (xp[i] - mu) * is * s2);
  }
}

// ---------------------------------------------------------------------------------------------
// Whole-tensor moments (InstanceNorm2d(1) over an N1 x N2 matrix): two small kernels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moments_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                              long long n) {
  __shared__ float red[12];
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  const long long beg = blockIdx.x * per, end = min(beg + per, n);
  float s = 0.f, q = 0.f, cnt = 0.f;
  const float shift = beg < end ? x[beg] : 0.f;
  for (long long i = beg + threadIdx.x; i < end; i += 256) {
    const float v = x[i] - shift;
    s += v;
    q += v * v;
    cnt += 1.f;
  }
  float mean = cnt > 0.f ? s / cnt : 0.f;
  float m2 = cnt > 0.f ? fmaxf(q - s * mean, 0.f) : 0.f;
  mean += shift;
  wave_moments(cnt, mean, m2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[w * 3] = cnt;
    red[w * 3 + 1] = mean;
    red[w * 3 + 2] = m2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float nn = red[0], mu = red[1], mm = red[2];
    for (int i = 1; i < 4; ++i) moments_merge(nn, mu, mm, red[i * 3], red[i * 3 + 1], red[i * 3 + 2]);
    partial[blockIdx.x * 3] = nn;
    partial[blockIdx.x * 3 + 1] = mu;
    partial[blockIdx.x * 3 + 2] = mm;
  }
}

extern "C" {

// Number of partial slices ge_bn_* kernels use for a (B, HW) extent; partial buffers are [C][nb][3] floats.
int ge_bn_num_partials(int B, int HW) { return bn_slice(B, HW).NB; }

int ge_bn_stats_partial(const float* x, float* partial, int B, int C, int HW, void* stream) {
  GE_REQUIRE(x && partial && B > 0 && C > 0 && HW > 0, "bn_stats_partial: bad arguments");
  const BnSlice sl = bn_slice(B, HW);
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(sl.NB, C), dim3(256), 0, (hipStream_t)stream, x, partial, B, C, HW,
                     sl.NB, sl.planes_per_blk, sl.segs_per_plane);
  GE_CHECK_LAUNCH("bn_stats_partial");
  return GE_OK;
}

int ge_bn_finalize(const float* partial, long long stride_c, long long stride_b, int NB, int C, float eps,
                   float momentum, float* stats, float* mean, float* invstd, float* running_mean, float* running_var,
                   void* stream) {
  GE_REQUIRE(partial && NB > 0 && C > 0, "bn_finalize: bad arguments");
  if (NB > 16)
    hipLaunchKernelGGL(bn_finalize_wave_kernel, dim3(ge_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial,
                       stride_c, stride_b, NB, C, eps, momentum, stats, mean, invstd, running_mean, running_var);
  else
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ge_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, partial, stride_c,
                       stride_b, NB, C, eps, momentum, stats, mean, invstd, running_mean, running_var);
  GE_CHECK_LAUNCH("bn_finalize");
  return GE_OK;
}

int ge_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                const float* residual, float* y, int B, int C, int HW, int relu, void* stream) {
  GE_REQUIRE(x && mean && invstd && y, "bn_apply: null pointer");
  const long long n = (long long)B * C * HW;
  if (HW % 4 == 0) {
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ge_stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, mean,
                       invstd, gamma, beta, residual, y, n / 4, C, HW / 4, relu, make_fastdiv(HW / 4), make_fastdiv(C));
  } else {
    hipLaunchKernelGGL(bn_apply_scalar_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       mean, invstd, gamma, beta, residual, y, n, C, HW, relu);
  }
  GE_CHECK_LAUNCH("bn_apply");
  return GE_OK;
}

// sums[C][2] = (sum dy_m, sum dy_m*xhat); partial: [C][nb][2] floats of workspace.
// ReLU mask: `out` (saved BN output) when given; else, when recompute_relu != 0, recomputed from x with gamma/beta.
int ge_bn_bwd_reduce(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int recompute_relu, float* partial, float* sums,
                     float* dgamma, float* dbeta, int accumulate, int B, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && partial && sums, "bn_bwd_reduce: null pointer");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_reduce: pass either the saved output or recompute_relu");
  const BnSlice sl = bn_slice(B, HW);
  const int NB = sl.NB;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(NB, C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, partial, B, C, HW, NB, sl.planes_per_blk, sl.segs_per_plane);
  GE_CHECK_LAUNCH("bn_bwd_partial");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ge_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, partial, NB, C,
                     sums, dgamma, dbeta, accumulate);
  GE_CHECK_LAUNCH("bn_bwd_finalize");
  return GE_OK;
}

int ge_bn_bwd_apply(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, int recompute_relu, const float* sums, float inv_count,
                    float* dx, float* dres, int B, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && sums && dx, "bn_bwd_apply: null pointer");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_apply: pass either the saved output or recompute_relu");
  const long long n = (long long)B * C * HW;
  static const int cm_on = getenv("GE_BN_APPLY_CM") ? atoi(getenv("GE_BN_APPLY_CM")) : 1;   // 0: linear pass, 2: channel-major but forward
  if (cm_on && HW % 4 == 0) {
    const BnSlice sl = bn_slice(B, HW);
    hipLaunchKernelGGL(bn_bwd_apply_cm_kernel, dim3(sl.NB, C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                       gamma, beta, recompute_relu, sums, inv_count, dx, dres, B, C, HW, sl.NB, sl.planes_per_blk,
                       sl.segs_per_plane, cm_on == 2 ? 0 : 1, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0);
  } else if (HW % 4 == 0)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ge_stream_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, x, out, mean, invstd, gamma, beta, recompute_relu, sums, inv_count, dx, dres, n / 4, C, HW / 4,
                       make_fastdiv(HW / 4), make_fastdiv(C));
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       dy, x, out, mean, invstd, gamma, beta, recompute_relu, sums, inv_count, dx, dres, n, C, HW,
                       make_fastdiv(HW), make_fastdiv(C));
  GE_CHECK_LAUNCH("bn_bwd_apply");
  return GE_OK;
}

// Train-mode BatchNorm backward of a big layer WITHOUT SyncBN in two launches instead of three: ge_bn_bwd_partials leaves the
// per-slice sums ([C][nb][2], nb = ge_bn_num_partials), ge_bn_bwd_apply_partials folds them per workgroup (finalize order, the
// same bits), writes dgamma / dbeta (+)= and dx (/ dres).  1 from ge_bn_bwd_two_launch_ok when the layer qualifies.
int ge_bn_bwd_two_launch_ok(int B, int HW) {
  static const int on = getenv("GE_BN_BWD2") ? atoi(getenv("GE_BN_BWD2")) : 1;
  static const int cm_on = getenv("GE_BN_APPLY_CM") ? atoi(getenv("GE_BN_APPLY_CM")) : 1;
  return on && cm_on && HW % 4 == 0 && bn_slice(B, HW).NB <= 256;
}
int ge_bn_bwd_partials(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                       const float* gamma, const float* beta, int recompute_relu, float* partial, int B, int C, int HW,
                       void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && partial, "bn_bwd_partials: null pointer");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_partials: pass either the saved output or recompute_relu");
  const BnSlice sl = bn_slice(B, HW);
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(sl.NB, C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd, gamma,
                     beta, recompute_relu, partial, B, C, HW, sl.NB, sl.planes_per_blk, sl.segs_per_plane);
  GE_CHECK_LAUNCH("bn_bwd_partials");
  return GE_OK;
}
int ge_bn_bwd_apply_partials(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int recompute_relu, const float* partial, float* dgamma,
                             float* dbeta, int accumulate, float inv_count, float* dx, float* dres, int B, int C, int HW,
                             void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && partial && dx && HW % 4 == 0, "bn_bwd_apply_partials: bad arguments");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_apply_partials: pass either the saved output or recompute_relu");
  static const int cm_on = getenv("GE_BN_APPLY_CM") ? atoi(getenv("GE_BN_APPLY_CM")) : 1;
  const BnSlice sl = bn_slice(B, HW);
  hipLaunchKernelGGL(bn_bwd_apply_cm_kernel, dim3(sl.NB, C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd, gamma,
                     beta, recompute_relu, (const float*)nullptr, inv_count, dx, dres, B, C, HW, sl.NB, sl.planes_per_blk,
                     sl.segs_per_plane, cm_on == 2 ? 0 : 1, partial, dgamma, dbeta, accumulate);
  GE_CHECK_LAUNCH("bn_bwd_apply_partials");
  return GE_OK;
}

// 1 if the one-workgroup-per-channel BatchNorm kernels take a layer with this batch and plane size.
int ge_bn_channel_ok(int B, int HW) {
  static const int on = getenv("GE_BN_CHANNEL") ? atoi(getenv("GE_BN_CHANNEL")) : 1;
  constexpr int lim = 16384;
  return on && HW % 4 == 0 && (long long)B * HW <= lim;
}

// Whole train-mode BatchNorm forward of a small layer in one launch (ge_bn_channel_ok): moments from `partial`
// ([C] x NB triples with the given strides, as ge_bn_finalize takes them) or, when partial is null, from x itself;
// writes mean / invstd (for the backward), updates the running statistics, applies (+ residual)(relu).
int ge_bn_fwd_channel(const float* x, const float* partial, long long stride_c, long long stride_b, int NB,
                      const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd,
                      float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum, int relu,
                      void* stream) {
  GE_REQUIRE(x && y && mean && invstd && B > 0 && C > 0 && HW > 0, "bn_fwd_channel: bad arguments");
  GE_REQUIRE(ge_bn_channel_ok(B, HW), "bn_fwd_channel: layer too large (B*HW = %lld) or HW %% 4 != 0", (long long)B * HW);
  GE_REQUIRE(!partial || NB > 0, "bn_fwd_channel: partials without a count");
  hipLaunchKernelGGL(bn_fwd_channel_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, partial, stride_c, stride_b,
                     NB, gamma, beta, residual, y, mean, invstd, running_mean, running_var, B, C, HW / 4, eps, momentum,
                     relu);
  GE_CHECK_LAUNCH("bn_fwd_channel");
  return GE_OK;
}

// Big layers, no SyncBN: finalize + apply in ONE launch.  Workgroup (slice, c): wave 0 merges the channel's NB conv-epilogue
// triples exactly as bn_finalize_wave_kernel does (lane-strided in order, eight loads in flight, then the Chan butterfly: the
// same statistics bit for bit), every workgroup of the channel does so redundantly (24 KB out of L2 for 2048 triples) and then
// applies its slice of frames with wave-uniform constants; slice 0 stores mean / invstd and updates the running statistics.
__global__ __launch_bounds__(256) void bn_fwd_merge_apply_kernel(const float* __restrict__ x, const float* __restrict__ partial,
                                                                 long long sc, long long sb, int NB,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ residual, float* __restrict__ y,
                                                                 float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                                 float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, int B, int C, int HW4,
                                                                 float eps, float momentum, int relu, int slices) {
  __shared__ float s_stat[2];
  const int c = blockIdx.y, sl = blockIdx.x;
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    float n = 0.f, mu = 0.f, m2 = 0.f;
    const float* pc = partial + (size_t)c * sc;
    for (int i0 = lane; i0 < NB; i0 += 64 * 8) {
      float tn[8], tm[8], tq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 64 * u;
        const float* p = pc + (size_t)(i < NB ? i : lane) * sb;
        tn[u] = p[0];
        tm[u] = p[1];
        tq[u] = p[2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 64 * u < NB) moments_merge(n, mu, m2, tn[u], tm[u], tq[u]);
    }
    wave_moments(n, mu, m2);
    if (lane == 0) {
      const float var = n > 0.f ? m2 / n : 0.f;
      const float is = 1.0f / sqrtf(var + eps);
      s_stat[0] = mu;
      s_stat[1] = is;
      if (sl == 0) {
        mean_out[c] = mu;
        invstd_out[c] = is;
        if (running_mean) {
          const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
          running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
          running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
      }
    }
  }
  __syncthreads();
  const float scv = s_stat[1] * (gamma ? gamma[c] : 1.f);
  const float shv = fmaf(-s_stat[0], scv, beta ? beta[c] : 0.f);      // == bn_scale_shift: the backward recomputes this
  const int b_lo = (int)((long long)B * sl / slices), b_hi = (int)((long long)B * (sl + 1) / slices);
  const float4* x4 = (const float4*)x;
  const float4* r4 = (const float4*)residual;
  float4* y4 = (float4*)y;
  for (int b = b_lo; b < b_hi; ++b) {
    const size_t base = ((size_t)b * C + c) * HW4;
    for (int e = threadIdx.x; e < HW4; e += 256) {
      float4 v = x4[base + e];
      v.x = fmaf(v.x, scv, shv);
      v.y = fmaf(v.y, scv, shv);
      v.z = fmaf(v.z, scv, shv);
      v.w = fmaf(v.w, scv, shv);
      if (residual) {
        const float4 r = r4[base + e];
        v.x += r.x;
        v.y += r.y;
        v.z += r.z;
        v.w += r.w;
      }
      if (relu == 1) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
      } else if (relu == 2) {
        v.x = gelu_f(v.x);
        v.y = gelu_f(v.y);
        v.z = gelu_f(v.z);
        v.w = gelu_f(v.w);
      }
      y4[base + e] = v;
    }
  }
}

// 1 when ge_bn_fwd_merge_apply takes a layer (the wave merge of ge_bn_finalize: NB > 16; 16-byte planes)
int ge_bn_fwd_merge_apply_ok(int NB, int HW) {
  static const int on = getenv("GE_BN_FWD1") ? atoi(getenv("GE_BN_FWD1")) : 1;
  return on && NB > 16 && HW % 4 == 0;
}
// ge_bn_finalize (from conv-epilogue partials) + ge_bn_apply of a big layer without SyncBN in one launch
int ge_bn_fwd_merge_apply(const float* x, const float* partial, long long stride_c, long long stride_b, int NB,
                          const float* gamma, const float* beta, const float* residual, float* y, float* mean, float* invstd,
                          float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum, int relu,
                          void* stream) {
  GE_REQUIRE(x && partial && y && mean && invstd && B > 0 && C > 0 && ge_bn_fwd_merge_apply_ok(NB, HW),
             "bn_fwd_merge_apply: bad arguments");
  int slices = (2048 + C - 1) / C;        // ~2048 workgroups, at most one per frame
  if (slices > B) slices = B;
  if (slices < 1) slices = 1;
  hipLaunchKernelGGL(bn_fwd_merge_apply_kernel, dim3(slices, C), dim3(256), 0, (hipStream_t)stream, x, partial, stride_c,
                     stride_b, NB, gamma, beta, residual, y, mean, invstd, running_mean, running_var, B, C, HW / 4, eps,
                     momentum, relu, slices);
  GE_CHECK_LAUNCH("bn_fwd_merge_apply");
  return GE_OK;
}

// SyncBN, big layer: every workgroup merges the [world] gathered triples of its channel itself (the wave order in which the
// small-layer kernels merge them) and applies its slice: one launch per segment where ge_bn_finalize + ge_bn_apply took two.
int ge_bn_fwd_merge_apply_sync(const float* x, const float* gathered, long long stride_c, long long stride_b, int world,
                               const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                               float* invstd, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                               float momentum, int relu, void* stream) {
  GE_REQUIRE(x && gathered && y && mean && invstd && B > 0 && C > 0 && world >= 1 && HW % 4 == 0,
             "bn_fwd_merge_apply_sync: bad arguments");
  int slices = (2048 + C - 1) / C;
  if (slices > B) slices = B;
  if (slices < 1) slices = 1;
  hipLaunchKernelGGL(bn_fwd_merge_apply_kernel, dim3(slices, C), dim3(256), 0, (hipStream_t)stream, x, gathered, stride_c,
                     stride_b, world, gamma, beta, residual, y, mean, invstd, running_mean, running_var, B, C, HW / 4, eps,
                     momentum, relu, slices);
  GE_CHECK_LAUNCH("bn_fwd_merge_apply_sync");
  return GE_OK;
}

static int bn_fill_segs(BnSegs& sg, const int* seg, int S, int HW, bool with_partial, bool small_only = true) {
  sg.S = S;
  for (int s = 0; s < S; ++s) {
    sg.b0[s] = seg[4 * s + 0];
    sg.bs[s] = seg[4 * s + 1];
    sg.poff[s] = with_partial ? seg[4 * s + 2] : 0;
    sg.nb[s] = with_partial ? seg[4 * s + 3] : 0;
    if (sg.bs[s] <= 0 || (small_only && !ge_bn_channel_ok(sg.bs[s], HW))) return 0;
  }
  return 1;
}

// ge_bn_fwd_channel for S <= 16 passes concatenated along the batch (functional.bn_segments) in ONE launch.  seg: host
// array of S x (first frame, frames, offset of the segment's triples inside a channel's partials, number of triples);
// mean / invstd: [S][C]; the running statistics are updated segment by segment, in order.
int ge_bn_fwd_channel_segs(const float* x, const float* partial, long long stride_c, long long stride_b, const int* seg,
                           int S, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                           float* invstd, float* running_mean, float* running_var, int C, int HW, float eps, float momentum,
                           int relu, void* stream) {
  GE_REQUIRE(x && y && mean && invstd && seg && S >= 1 && S <= 16 && C > 0 && HW > 0, "bn_fwd_channel_segs: bad arguments");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, partial != nullptr), "bn_fwd_channel_segs: a segment is too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_fwd_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, partial, stride_c,
                     stride_b, sg, gamma, beta, residual, y, mean, invstd, running_mean, running_var, C, HW / 4, eps,
                     momentum, relu, 0ll);
  GE_CHECK_LAUNCH("bn_fwd_channel_segs");
  return GE_OK;
}

// ---- SyncBN, all S segments of a small layer per launch (kernels above).  seg as for ge_bn_fwd_channel_segs.
// ge_bn_finalize_segs: stats [S][C][3] = each segment's local (n, mean, M2) from the conv-epilogue triples (what S calls of
// ge_bn_finalize(..., stats + s * C * 3, ...) leave) -- the buffer the ranks all-gather.
int ge_bn_finalize_segs(const float* partial, long long stride_c, long long stride_b, const int* seg, int S, int C, int HW,
                        float* stats, void* stream) {
  GE_REQUIRE(partial && stats && seg && S >= 1 && S <= 16 && C > 0, "bn_finalize_segs: bad arguments");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, true, false), "bn_finalize_segs: an empty segment");      // (layers of any size: only triples are read)
  for (int s = 0; s < S; ++s) GE_REQUIRE(sg.nb[s] > 0, "bn_finalize_segs: segment %d has no triples", s);
  hipLaunchKernelGGL(bn_finalize_segs_kernel, dim3(ge_cdiv(C, 4), S), dim3(256), 0, (hipStream_t)stream, partial, stride_c,
                     stride_b, sg, C, stats);
  GE_CHECK_LAUNCH("bn_finalize_segs");
  return GE_OK;
}
// ge_bn_stats_channel_segs: the same buffer from x itself (small layers whose convolution left no moments)
int ge_bn_stats_channel_segs(const float* x, const int* seg, int S, int C, int HW, float* stats, void* stream) {
  GE_REQUIRE(x && stats && seg && S >= 1 && S <= 16 && C > 0, "bn_stats_channel_segs: bad arguments");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, false), "bn_stats_channel_segs: a segment is too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_stats_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, sg, C, HW / 4, stats);
  GE_CHECK_LAUNCH("bn_stats_channel_segs");
  return GE_OK;
}
// ge_bn_fwd_channel_segs_sync: gathered = [world][S][C][3] (the all-gather of every rank's ge_bn_finalize_segs buffer); per
// segment: merge the world triples of the channel, mean / invstd [S][C], running statistics in segment order, apply.
int ge_bn_fwd_channel_segs_sync(const float* x, const float* gathered, int world, const int* seg, int S, const float* gamma,
                                const float* beta, const float* residual, float* y, float* mean, float* invstd,
                                float* running_mean, float* running_var, int C, int HW, float eps, float momentum, int relu,
                                void* stream) {
  GE_REQUIRE(x && gathered && y && mean && invstd && seg && world >= 1 && S >= 1 && S <= 16 && C > 0 && HW > 0,
             "bn_fwd_channel_segs_sync: bad arguments");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, false), "bn_fwd_channel_segs_sync: a segment is too large or HW %% 4 != 0");
  for (int s = 0; s < S; ++s) sg.nb[s] = world;
  hipLaunchKernelGGL(bn_fwd_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, gathered, 3ll,
                     (long long)S * C * 3, sg, gamma, beta, residual, y, mean, invstd, running_mean, running_var, C, HW / 4,
                     eps, momentum, relu, (long long)C * 3);
  GE_CHECK_LAUNCH("bn_fwd_channel_segs_sync");
  return GE_OK;
}
// ge_bn_bwd_reduce_channel_segs: sums [S][C][2] + local dgamma / dbeta (what S calls of ge_bn_bwd_reduce_channel leave);
// ge_bn_bwd_apply_channel_segs: dx / dres from the all-reduced sums; inv_count: HOST array of S floats, 1 / (frames * HW * world).
int ge_bn_bwd_reduce_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int recompute_relu, float* sums, float* dgamma,
                                  float* dbeta, int accumulate, const int* seg, int S, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && sums && seg && S >= 1 && S <= 16, "bn_bwd_reduce_channel_segs: bad arguments");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_reduce_channel_segs: pass either the saved output or recompute_relu");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, false), "bn_bwd_reduce_channel_segs: a segment is too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_bwd_reduce_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, dgamma, dbeta, accumulate, sg, sums, C, HW / 4);
  GE_CHECK_LAUNCH("bn_bwd_reduce_channel_segs");
  return GE_OK;
}
int ge_bn_bwd_apply_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                                 const float* gamma, const float* beta, int recompute_relu, const float* sums,
                                 const float* inv_count, const int* seg, int S, float* dx, float* dres, int C, int HW,
                                 void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && sums && inv_count && dx && seg && S >= 1 && S <= 16,
             "bn_bwd_apply_channel_segs: bad arguments");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_apply_channel_segs: pass either the saved output or recompute_relu");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, false), "bn_bwd_apply_channel_segs: a segment is too large or HW %% 4 != 0");
  BnSegScale sc;
  for (int s = 0; s < 16; ++s) sc.inv_count[s] = s < S ? inv_count[s] : 0.f;
  hipLaunchKernelGGL(bn_bwd_apply_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, sums, sg, sc, dx, dres, C, HW / 4);
  GE_CHECK_LAUNCH("bn_bwd_apply_channel_segs");
  return GE_OK;
}

// ge_bn_bwd_channel for the same segments in ONE launch; mean / invstd: [S][C]; seg as above (offsets unused)
int ge_bn_bwd_channel_segs(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, int recompute_relu, float* dgamma, float* dbeta,
                           int accumulate, const int* seg, int S, float* dx, float* dres, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && dx && seg && S >= 1 && S <= 16, "bn_bwd_channel_segs: bad arguments");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_channel_segs: pass either the saved output or recompute_relu");
  BnSegs sg;
  GE_REQUIRE(bn_fill_segs(sg, seg, S, HW, false), "bn_bwd_channel_segs: a segment is too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_bwd_channel_segs_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, dgamma, dbeta, accumulate, sg, dx, dres, C, HW / 4);
  GE_CHECK_LAUNCH("bn_bwd_channel_segs");
  return GE_OK;
}

// First half of the backward alone (SyncBN: the sums cross the ranks before dx can be formed): sums[c] = (sum dy_m,
// sum dy_m * xhat) of a small layer in ONE launch (one workgroup per channel) instead of partial + finalize;
// dgamma / dbeta (+)= the local sums, as ge_bn_bwd_reduce leaves them.
int ge_bn_bwd_reduce_channel(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int recompute_relu, float* sums, float* dgamma,
                             float* dbeta, int accumulate, int B, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && sums, "bn_bwd_reduce_channel: null pointer");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_reduce_channel: pass either the saved output or recompute_relu");
  GE_REQUIRE(ge_bn_channel_ok(B, HW), "bn_bwd_reduce_channel: layer too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_bwd_channel_kernel<true>, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, dgamma, dbeta, accumulate, 0.f, sums, nullptr, B, C, HW / 4);
  GE_CHECK_LAUNCH("bn_bwd_reduce_channel");
  return GE_OK;
}

// Whole BatchNorm backward of a small layer in one launch: dx (and dres), dgamma / dbeta (+)=.  Arguments as
// ge_bn_bwd_reduce + ge_bn_bwd_apply; not for SyncBN (its sums cross the ranks between the two halves).
int ge_bn_bwd_channel(const float* dy, const float* x, const float* out, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int recompute_relu, float* dgamma, float* dbeta,
                      int accumulate, float inv_count, float* dx, float* dres, int B, int C, int HW, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && dx, "bn_bwd_channel: null pointer");
  GE_REQUIRE(!(out && recompute_relu), "bn_bwd_channel: pass either the saved output or recompute_relu");
  GE_REQUIRE(ge_bn_channel_ok(B, HW), "bn_bwd_channel: layer too large or HW %% 4 != 0");
  hipLaunchKernelGGL(bn_bwd_channel_kernel<false>, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, x, out, mean, invstd,
                     gamma, beta, recompute_relu, dgamma, dbeta, accumulate, inv_count, dx, dres, B, C, HW / 4);
  GE_CHECK_LAUNCH("bn_bwd_channel");
  return GE_OK;
}

int ge_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                     int B, int C, int HW, int G, float eps, int relu, void* stream) {
  GE_REQUIRE(x && y && mean && invstd && G > 0 && C % G == 0, "groupnorm_fwd: bad arguments");
  const long long L = (long long)(C / G) * HW;
  if (HW % 256 == 0 && L <= 8192) {          // register-resident: 256 threads x 8 float4
    hipLaunchKernelGGL((gn_fwd_reg_kernel<256, 8>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                       mean, invstd, C, G, HW, eps, relu);
  } else if (HW % 256 == 0 && L <= 32768) {  // 1024 threads x 8 float4
    hipLaunchKernelGGL((gn_fwd_reg_kernel<1024, 8>), dim3(B * G), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta,
                       y, mean, invstd, C, G, HW, eps, relu);
  } else {
    const int threads = L <= 1024 ? 64 : 256;
    hipLaunchKernelGGL(gn_fwd_kernel, dim3(B * G), dim3(threads), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                       invstd, C, G, HW, eps, relu);
  }
  GE_CHECK_LAUNCH("groupnorm_fwd");
  return GE_OK;
}

// dgamma_part / dbeta_part: [B][C] workspaces; dgamma/dbeta: [C] (may be null when affine is off).
int ge_groupnorm_bwd(const float* dy, const float* x, const float* out, const float* gamma, const float* mean,
                     const float* invstd, float* dx, float* dgamma_part, float* dbeta_part, float* dgamma,
                     float* dbeta, int B, int C, int HW, int G, void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && dx && dgamma_part && dbeta_part, "groupnorm_bwd: null pointer");
  const long long L = (long long)(C / G) * HW;
  const size_t lds = 0;
  if (HW % 256 == 0 && L <= 8192) {
    hipLaunchKernelGGL((gn_bwd_reg_kernel<256, 8>), dim3(B * G), dim3(256), lds, (hipStream_t)stream, dy, x, out, gamma,
                       mean, invstd, dx, dgamma_part, dbeta_part, C, G, HW);
  } else if (HW % 256 == 0 && L <= 32768) {
    hipLaunchKernelGGL((gn_bwd_reg_kernel<1024, 8>), dim3(B * G), dim3(1024), lds, (hipStream_t)stream, dy, x, out,
                       gamma, mean, invstd, dx, dgamma_part, dbeta_part, C, G, HW);
  } else {
    const int threads = L <= 1024 ? 64 : 256;
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * G), dim3(threads), 0, (hipStream_t)stream, dy, x, out, gamma, mean,
                       invstd, dx, dgamma_part, dbeta_part, C, G, HW);
  }
  GE_CHECK_LAUNCH("groupnorm_bwd");
  if (dgamma) {
    hipLaunchKernelGGL(colsum2_kernel, dim3(ge_cdiv(C, 64), 2), dim3(1024), 0, (hipStream_t)stream, dgamma_part, dgamma,
                       dbeta_part, dbeta, B, C, 0);
    GE_CHECK_LAUNCH("groupnorm_bwd_colsum");
  }
  return GE_OK;
}

int ge_colsum(const float* in, float* out, int R, int C, void* stream) {
  GE_REQUIRE(in && out && R > 0 && C > 0, "colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3(ge_cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, in, out, R, C, 0);
  GE_CHECK_LAUNCH("colsum");
  return GE_OK;
}

// out[c] += sum_r in[r][c]: bias / affine gradients accumulated straight into a flat gradient buffer
int ge_colsum_accumulate(const float* in, float* out, int R, int C, void* stream) {
  GE_REQUIRE(in && out && R > 0 && C > 0, "colsum_accumulate: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3(ge_cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, in, out, R, C, 1);
  GE_CHECK_LAUNCH("colsum_accumulate");
  return GE_OK;
}

// ge_colsum_accumulate for two [R][C] sources at once (a norm layer's d gamma / d beta partials into the flat gradients)
int ge_colsum_accumulate2(const float* in0, float* out0, const float* in1, float* out1, int R, int C, void* stream) {
  GE_REQUIRE(in0 && out0 && in1 && out1 && R > 0 && C > 0, "colsum_accumulate2: bad arguments");
  hipLaunchKernelGGL(colsum2_kernel, dim3(ge_cdiv(C, 64), 2), dim3(1024), 0, (hipStream_t)stream, in0, out0, in1, out1, R, C,
                     1);
  GE_CHECK_LAUNCH("colsum_accumulate2");
  return GE_OK;
}

int ge_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                     int R, int D, float eps, void* stream) {
  GE_REQUIRE(x && y && mean && invstd && R > 0 && D > 0, "layernorm_fwd: bad arguments");
  GE_REQUIRE((gamma == nullptr) == (beta == nullptr), "layernorm_fwd: gamma/beta must both be set or both null");
  if (!gamma && D >= 4096) {
    const bool vec = D % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    if (vec)
      hipLaunchKernelGGL(ln_fwd_wide_kernel<4>, dim3(R), dim3(1024), 0, (hipStream_t)stream, x, y, mean, invstd, D, eps);
    else
      hipLaunchKernelGGL(ln_fwd_wide_kernel<1>, dim3(R), dim3(1024), 0, (hipStream_t)stream, x, y, mean, invstd, D, eps);
  } else
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ge_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean,
                       invstd, R, D, eps);
  GE_CHECK_LAUNCH("layernorm_fwd");
  return GE_OK;
}

// Row blocks of 32 rows; dgamma_part/dbeta_part: [ge_layernorm_bwd_blocks(R)][D] workspaces (null when no affine).
int ge_layernorm_bwd_blocks(int R) { return ge_cdiv(R, 32); }

int ge_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* invstd,
                     float* dx, float* dgamma_part, float* dbeta_part, float* dgamma, float* dbeta, int R, int D,
                     void* stream) {
  GE_REQUIRE(dy && x && mean && invstd && dx && R > 0 && D > 0, "layernorm_bwd: bad arguments");
  const int nblk = ge_cdiv(R, 32);
  if (!gamma && D >= 4096) {
    const bool vec = D % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    if (vec)
      hipLaunchKernelGGL(ln_bwd_wide_kernel<4>, dim3(R), dim3(1024), 0, (hipStream_t)stream, dy, x, mean, invstd, dx, D);
    else
      hipLaunchKernelGGL(ln_bwd_wide_kernel<1>, dim3(R), dim3(1024), 0, (hipStream_t)stream, dy, x, mean, invstd, dx, D);
    GE_CHECK_LAUNCH("layernorm_bwd_wide");
    return GE_OK;
  }
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, mean, invstd, dx,
                     dgamma_part, dbeta_part, R, D, 32);
  GE_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma_part && dgamma) {
    hipLaunchKernelGGL(colsum2_kernel, dim3(ge_cdiv(D, 64), 2), dim3(1024), 0, (hipStream_t)stream, dgamma_part, dgamma,
                       dbeta_part, dbeta, nblk, D, 0);
    GE_CHECK_LAUNCH("layernorm_bwd_colsum");
  }
  return GE_OK;
}

// Whole-tensor (n, mean, M2): partial is a [64][3] float workspace, stats receives the merged triple.
int ge_tensor_moments(const float* x, float* partial, float* stats, long long n, void* stream) {
  GE_REQUIRE(x && partial && stats && n > 0, "tensor_moments: bad arguments");
  long long nb = (n + 4095) / 4096;
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(moments_partial_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, partial, n);
  GE_CHECK_LAUNCH("moments_partial");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, 0ll, 3ll, (int)nb, 1,
                     0.f, 0.f, stats, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  GE_CHECK_LAUNCH("moments_finalize");
  return GE_OK;
}

}  // extern "C"
