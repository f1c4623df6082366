// HBM-bound spatial kernels on NCHW fp32: bilinear resize (align_corners=True) with fused lateral add
// (FPN._upsample_add / _upsample), max / average pooling, and the element-wise activations.
// Forward + backward; backward passes are gather-form (no atomics) so results are run-to-run identical.
#include "ge_common.h"
#include <algorithm>
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// Bilinear, align_corners=True:  src = dst * (in-1)/(out-1);  out = (1-l)*v0 + l*v1 per axis.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void bilin_src(int o, float scale, int in, int& i0, int& i1, float& l) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l = s - (float)i0;
}

// One thread per output element (VEC: per four consecutive output columns -- 16-B store / lateral-add load, shared
// row interpolation).  Index arithmetic is 32-bit mul-hi division; grid.y walks chunks of < 2^31 elements.
template <bool VEC>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                           float* __restrict__ y, long long planes, int Hi, int Wi,
                                                           int Ho, int Wo, float sh, float sw, FastDiv fd_w,
                                                           FastDiv fd_plane, uint32_t planes_per_chunk) {
  constexpr int V = VEC ? 4 : 1;
  const uint32_t per_plane = (uint32_t)Ho * (uint32_t)(Wo / V);
  const long long pl_base = (long long)blockIdx.y * planes_per_chunk;
  const uint32_t chunk_planes = (uint32_t)min((long long)planes_per_chunk, planes - pl_base);
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= chunk_planes * per_plane) return;
  uint32_t plc, e, oy, oxv;
  fd_divmod(g, fd_plane, plc, e);                             // plane within the chunk, element within the plane
  fd_divmod(e, fd_w, oy, oxv);                                // fd_w divides by Wo / V
  const int ox0 = (int)oxv * V;
  int y0, y1;
  float ly;
  bilin_src((int)oy, sh, Hi, y0, y1, ly);
  int x0[V], x1[V];
  float lx[V];
#pragma unroll
  for (int u = 0; u < V; ++u) bilin_src(ox0 + u, sw, Wi, x0[u], x1[u], lx[u]);
  const size_t o = (size_t)oy * Wo + ox0;
  {
    const long long pl = pl_base + plc;
    const float* r0 = x + (size_t)pl * Hi * Wi + (size_t)y0 * Wi;
    const float* r1 = x + (size_t)pl * Hi * Wi + (size_t)y1 * Wi;
    float v[V];
#pragma unroll
    for (int u = 0; u < V; ++u) {
      const float v00 = r0[x0[u]], v01 = r0[x1[u]], v10 = r1[x0[u]], v11 = r1[x1[u]];
      v[u] = (1.f - ly) * ((1.f - lx[u]) * v00 + lx[u] * v01) + ly * ((1.f - lx[u]) * v10 + lx[u] * v11);
    }
    const size_t po = (size_t)pl * Ho * Wo + o;
    if (VEC) {
      float4 r = make_float4(v[0], v[VEC ? 1 : 0], v[VEC ? 2 : 0], v[VEC ? 3 : 0]);
      if (add) {
        const float4 a = *(const float4*)(add + po);
        r.x += a.x;
        r.y += a.y;
        r.z += a.z;
        r.w += a.w;
      }
      *(float4*)(y + po) = r;
    } else {
      y[po] = add ? v[0] + add[po] : v[0];
    }
  }
}

// dx[iy][ix] = sum over output pixels whose 4-tap footprint touches (iy, ix), same weights as forward.
template <int MAXC>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                           long long planes, int Hi, int Wi, int Ho, int Wo, float sh,
                                                           float sw, float inv_sh, float inv_sw, FastDiv fd_w,
                                                           FastDiv fd_plane, uint32_t planes_per_chunk) {
  const uint32_t per_plane = (uint32_t)Hi * (uint32_t)Wi;
  const long long pl_base = (long long)blockIdx.y * planes_per_chunk;
  const uint32_t chunk_planes = (uint32_t)min((long long)planes_per_chunk, planes - pl_base);
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= chunk_planes * per_plane) return;
  uint32_t plc, e, uy, ux;
  fd_divmod(g, fd_plane, plc, e);
  fd_divmod(e, fd_w, uy, ux);
  const int iy = (int)uy, ix = (int)ux;
  {
    const long long pl = pl_base + plc;
    const long long i = pl * Hi * Wi + e;
    // candidate output rows/cols: those with floor(o*scale) in {i-1, i}; widen by one for rounding safety
    int oy_lo = sh > 0.f ? (int)floorf((float)(iy - 1) * inv_sh) - 1 : 0;
    int oy_hi = sh > 0.f ? (int)ceilf((float)(iy + 1) * inv_sh) + 1 : Ho - 1;
    int ox_lo = sw > 0.f ? (int)floorf((float)(ix - 1) * inv_sw) - 1 : 0;
    int ox_hi = sw > 0.f ? (int)ceilf((float)(ix + 1) * inv_sw) + 1 : Wo - 1;
    oy_lo = max(oy_lo, 0);
    ox_lo = max(ox_lo, 0);
    oy_hi = min(oy_hi, Ho - 1);
    ox_hi = min(ox_hi, Wo - 1);
    const float* gp = dy + (size_t)pl * Ho * Wo;
    float acc = 0.f;
    // MAXC = candidate columns kept in registers (2/scale + margins): 12 covers up-scaling to ~4.5x, 24 to ~10x
    if (ox_hi - ox_lo < MAXC) {
      // separable form: the column weights are computed once (registers), not once per candidate row
      float wxs[MAXC];
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const int ox = ox_lo + j;
        int x0, x1;
        float lx;
        bilin_src(min(ox, Wo - 1), sw, Wi, x0, x1, lx);
        float wx = 0.f;
        if (x0 == ix) wx += 1.f - lx;
        if (x1 == ix) wx += lx;
        wxs[j] = ox <= ox_hi ? wx : 0.f;
      }
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bilin_src(oy, sh, Hi, y0, y1, ly);
        float wy = 0.f;
        if (y0 == iy) wy += 1.f - ly;
        if (y1 == iy) wy += ly;
        if (wy == 0.f) continue;
        const float* gr = gp + (size_t)oy * Wo + ox_lo;
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
          if (wxs[j] != 0.f) acc += wy * wxs[j] * gr[j];
      }
    } else {
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float ly;
        bilin_src(oy, sh, Hi, y0, y1, ly);
        float wy = 0.f;
        if (y0 == iy) wy += 1.f - ly;
        if (y1 == iy) wy += ly;
        if (wy == 0.f) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          int x0, x1;
          float lx;
          bilin_src(ox, sw, Wi, x0, x1, lx);
          float wx = 0.f;
          if (x0 == ix) wx += 1.f - lx;
          if (x1 == ix) wx += lx;
          if (wx != 0.f) acc += wy * wx * gp[oy * Wo + ox];
        }
      }
    }
    dx[i] = acc;
  }
}

// One workgroup per plane with the dy plane staged in LDS, reduced SEPARABLY: first along W into T[ix][oy] (all 256
// threads: Wi*Ho partial sums, lanes along oy so the padded rows are bank-conflict free and the column weights are
// wave-uniform), then along H into dx[iy][ix].  dy is read from HBM once and every tap is an LDS read; per dx element
// the work is O(candidates) per pass instead of O(candidates^2) on the Hi*Wi threads the gather form keeps busy.
__device__ __forceinline__ float bilin_weight(int o, float scale, int in, int i) {
  int i0, i1;
  float l;
  bilin_src(o, scale, in, i0, i1, l);
  float w = 0.f;
  if (i0 == i) w += 1.f - l;
  if (i1 == i) w += l;
  return w;
}
__device__ __forceinline__ void bilin_candidates(int i, float scale, float inv_scale, int out, int& lo, int& hi) {
  lo = scale > 0.f ? (int)floorf((float)(i - 1) * inv_scale) - 1 : 0;
  hi = scale > 0.f ? (int)ceilf((float)(i + 1) * inv_scale) + 1 : out - 1;
  lo = max(lo, 0);
  hi = min(hi, out - 1);
}

__global__ __launch_bounds__(256) void upsample_bwd_sep_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                               int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                               float inv_sh, float inv_sw, FastDiv fd_wo,
                                                               FastDiv fd_ho, FastDiv fd_wi) {
  extern __shared__ __attribute__((aligned(16))) float sg[];   // dy plane [Ho][Wo + 1], then T [Wi][Ho + 1]
  const int P = Wo + 1, Q = Ho + 1;
  float* sT = sg + Ho * P;
  const size_t pl = blockIdx.x;
  const float* gp = dy + pl * (size_t)Ho * Wo;
  const uint32_t n_out = (uint32_t)(Ho * Wo);
  if ((Wo & 3) == 0) {
    for (uint32_t i = threadIdx.x * 4; i < n_out; i += 1024) {
      const float4 v = *(const float4*)(gp + i);
      uint32_t r, c;
      fd_divmod(i, fd_wo, r, c);
      float* d = sg + r * P + c;
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
    }
  } else {
    for (uint32_t i = threadIdx.x; i < n_out; i += 256) {
      uint32_t r, c;
      fd_divmod(i, fd_wo, r, c);
      sg[r * P + c] = gp[i];
    }
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < (uint32_t)(Wi * Ho); e += 256) {
    uint32_t ux, uy;
    fd_divmod(e, fd_ho, ux, uy);
    const int ix = (int)ux;
    int lo, hi;
    bilin_candidates(ix, sw, inv_sw, Wo, lo, hi);
    const float* row = sg + uy * P;
    float acc = 0.f;
    for (int ox = lo; ox <= hi; ++ox) acc += bilin_weight(ox, sw, Wi, ix) * row[ox];
    sT[ux * Q + uy] = acc;
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < (uint32_t)(Hi * Wi); e += 256) {
    uint32_t uy, ux;
    fd_divmod(e, fd_wi, uy, ux);
    const int iy = (int)uy;
    int lo, hi;
    bilin_candidates(iy, sh, inv_sh, Ho, lo, hi);
    const float* col = sT + ux * Q;
    float acc = 0.f;
    for (int oy = lo; oy <= hi; ++oy) acc += bilin_weight(oy, sh, Hi, iy) * col[oy];
    dx[pl * (size_t)Hi * Wi + e] = acc;
  }
}

// Streaming form for up-scaling (Ho >= Hi, Wo >= Wi: every bilinear backward on the FPN path).  dy is read from HBM
// exactly once, in full rows, and the only LDS is the half-reduced T[plane][iy][Wo + 1] (2 KB per plane at 8 -> 64
// instead of the 19 KB the staged-plane form needs): what bounds co-residency with the 34 KB-per-workgroup weight-
// gradient kernels that run beside the data-gradient chain on the side stream -- the staged form got ONE workgroup per
// CU next to them and took 690 us in the step against 46 us alone.
//   pass 1: a thread owns one dy column of one plane and walks oy; the source row y0(oy) is wave-uniform and
//           non-decreasing, so the two rows an output row feeds are two running registers, flushed to T when y0 moves;
//   pass 2: a thread owns one T row (plane, iy), walks ox the same way out of LDS (row pitch Wo + 1: conflict-free)
//           and overwrites the head of its own row with dx[iy][0..Wi) (x0 <= ox: never ahead of the read position);
//   pass 3: coalesced copy of the G * Hi * Wi results.
// Accumulation order is ascending oy, then ascending ox -- fixed, so results are run-to-run identical.
template <int UNROLL>
__global__ __launch_bounds__(256) void upsample_bwd_stream_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                  long long planes, int G, int Hi, int Wi, int Ho,
                                                                  int Wo, float sh, float sw, FastDiv fd_wo,
                                                                  FastDiv fd_hi, FastDiv fd_wi) {
  extern __shared__ __attribute__((aligned(16))) float sT[];   // [G][Hi][Wo + 1]
  const int P = Wo + 1;
  const long long pl0 = (long long)blockIdx.x * G;
  const int tid = threadIdx.x;
  uint32_t pg, ox;
  fd_divmod((uint32_t)tid, fd_wo, pg, ox);
  if ((int)pg < G && pl0 + pg < planes) {
    const float* col = dy + (size_t)(pl0 + pg) * Ho * Wo + ox;
    float* tcol = sT + (size_t)pg * Hi * P + ox;
    float a0 = 0.f, a1 = 0.f;
    int cur = 0;
    for (int oy0 = 0; oy0 < Ho; oy0 += UNROLL) {
      float v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = oy0 + u < Ho ? col[(size_t)(oy0 + u) * Wo] : 0.f;
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (oy0 + u >= Ho) break;
        int y0, y1;
        float ly;
        bilin_src(oy0 + u, sh, Hi, y0, y1, ly);
        while (cur < y0) {            // wave-uniform
          tcol[cur * P] = a0;
          a0 = a1;
          a1 = 0.f;
          ++cur;
        }
        a0 += (1.f - ly) * v[u];
        if (y1 == y0)
          a0 += ly * v[u];
        else
          a1 += ly * v[u];
      }
    }
    tcol[cur * P] = a0;
    if (cur + 1 < Hi) tcol[(cur + 1) * P] = a1;
    for (int r = cur + 2; r < Hi; ++r) tcol[r * P] = 0.f;
  }
  __syncthreads();
  if (tid < G * Hi) {
    uint32_t pr, iy;
    fd_divmod((uint32_t)tid, fd_hi, pr, iy);
    if (pl0 + pr < planes) {
      float* row = sT + (size_t)tid * P;     // (pr * Hi + iy) * P
      float a0 = 0.f, a1 = 0.f;
      int cur = 0;
      for (int x = 0; x < Wo; ++x) {
        int x0, x1;
        float lx;
        bilin_src(x, sw, Wi, x0, x1, lx);
        const float v = row[x];
        while (cur < x0) {            // wave-uniform; cur < x0 <= x: the slot was read in an earlier iteration
          row[cur] = a0;
          a0 = a1;
          a1 = 0.f;
          ++cur;
        }
        a0 += (1.f - lx) * v;
        if (x1 == x0)
          a0 += lx * v;
        else
          a1 += lx * v;
      }
      row[cur] = a0;
      if (cur + 1 < Wi) row[cur + 1] = a1;
      for (int r = cur + 2; r < Wi; ++r) row[r] = 0.f;
    }
  }
  __syncthreads();
  const uint32_t n_out = (uint32_t)min((long long)G, planes - pl0) * (uint32_t)(Hi * Wi);
  float* out = dx + (size_t)pl0 * Hi * Wi;
  for (uint32_t e = tid; e < n_out; e += blockDim.x) {
    uint32_t r, ix;
    fd_divmod(e, fd_wi, r, ix);       // r = plane_in_group * Hi + iy
    out[e] = sT[(size_t)r * P + ix];
  }
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(k, s, p): -inf padding, first arg-max in (kh, kw) scan order; arg index saved as uint8.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned char* __restrict__ arg, uint32_t total, int Hi,
                                                          int Wi, int Ho, int Wo, int k, int s, int p, FastDiv fd_wo,
                                                          FastDiv fd_ho) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    uint32_t t, uox, pl, uoy;
    fd_divmod(i, fd_wo, t, uox);
    fd_divmod(t, fd_ho, pl, uoy);
    const int ox = (int)uox, oy = (int)uoy;
    const float* xp = x + (size_t)pl * Hi * Wi;
    float best = -INFINITY;
    int bi = 255;
    if (k == 3) {
      // 3x3 window: the nine taps are loaded unconditionally from clamped positions (all in flight at once), then
      // scanned in (dy, dx) order with the out-of-image ones masked out
      float v[9];
      bool ok[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy * s - p + t / 3, ix = ox * s - p + t % 3;
        ok[t] = (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
        v[t] = xp[min(max(iy, 0), Hi - 1) * Wi + min(max(ix, 0), Wi - 1)];
      }
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (ok[t] && (bi == 255 || v[t] > best || v[t] != v[t])) {
          best = v[t];
          bi = t;
        }
      y[i] = best;
      arg[i] = (unsigned char)bi;
      continue;
    }
    for (int dy = 0; dy < k; ++dy) {
      const int iy = oy * s - p + dy;
      if ((unsigned)iy >= (unsigned)Hi) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = ox * s - p + dx;
        if ((unsigned)ix >= (unsigned)Wi) continue;
        const float v = xp[iy * Wi + ix];
        if (bi == 255 || v > best || v != v) {
          best = v;
          bi = dy * k + dx;
        }
      }
    }
    y[i] = best;
    arg[i] = (unsigned char)bi;
  }
}

// Gather form (no atomics): every input pixel looks at the <= ceil(k/s)^2 windows that contain it.  32-bit index
// arithmetic with mul-hi division: the 64-bit div/mod version spent more time on addresses than on memory.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const unsigned char* __restrict__ arg,
                                                          float* __restrict__ dx, uint32_t total, int Hi, int Wi,
                                                          int Ho, int Wo, int k, int s, int p, FastDiv fd_wi,
                                                          FastDiv fd_hi, FastDiv fd_s) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    uint32_t t, uix, pl, uiy;
    fd_divmod(i, fd_wi, t, uix);
    fd_divmod(t, fd_hi, pl, uiy);
    const int ix = (int)uix, iy = (int)uiy;
    float acc = 0.f;
    const uint32_t obase0 = pl * (uint32_t)(Ho * Wo);
    if (k == 3 && s == 2 && p == 1) {
      // ResNet stem pool: window o covers inputs 2o-1 .. 2o+1, so a pixel sits in window (i+1)/2 and, when i is odd, also
      // in the one before it.  All (up to four) candidates are loaded unconditionally -- eight independent loads in
      // flight instead of a dependent load per loop trip -- and selected afterwards.
      const int oy1 = (iy + 1) >> 1, ox1 = (ix + 1) >> 1;
      const int oys[2] = {oy1 - 1, oy1}, oxs[2] = {ox1 - 1, ox1};
      const bool vy[2] = {(iy & 1) != 0, oy1 < Ho}, vx[2] = {(ix & 1) != 0, ox1 < Wo};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int oy = min(max(oys[a], 0), Ho - 1), ox = min(max(oxs[b], 0), Wo - 1);
          const uint32_t o = obase0 + (uint32_t)(oy * Wo + ox);
          const int want = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));
          const int got = arg[o];
          const float g = dy[o];
          acc += (vy[a] && vx[b] && got == want) ? g : 0.f;
        }
      dx[i] = acc;
      continue;
    }
    // windows (oy, ox) with oy*s - p <= iy < oy*s - p + k
    const int oy_hi = min((int)fd_div((uint32_t)(iy + p), fd_s), Ho - 1);
    const int ox_hi = min((int)fd_div((uint32_t)(ix + p), fd_s), Wo - 1);
    const uint32_t obase = pl * (uint32_t)(Ho * Wo);
    for (int oy = oy_hi; oy >= 0 && oy * s - p + k > iy; --oy) {
      const int dyk = iy - (oy * s - p);
      for (int ox = ox_hi; ox >= 0 && ox * s - p + k > ix; --ox) {
        const int dxk = ix - (ox * s - p);
        const uint32_t o = obase + (uint32_t)(oy * Wo + ox);
        if (arg[o] == dyk * k + dxk) acc += dy[o];
      }
    }
    dx[i] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// avg_pool2d(x, r, r): no padding, floor output size.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint32_t total, int Hi, int Wi, int Ho, int Wo, int r,
                                                          FastDiv fd_wo, FastDiv fd_ho) {
  const float inv = 1.f / (float)(r * r);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    uint32_t t, ox, pl, oy;
    fd_divmod(i, fd_wo, t, ox);
    fd_divmod(t, fd_ho, pl, oy);
    const float* xp = x + (size_t)pl * Hi * Wi + (size_t)(oy * r) * Wi + ox * r;
    float s = 0.f;
    for (int dy = 0; dy < r; ++dy)
      for (int dx = 0; dx < r; ++dx) s += xp[dy * Wi + dx];
    y[i] = s * inv;
  }
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                          uint32_t total, int Hi, int Wi, int Ho, int Wo, int r,
                                                          FastDiv fd_wi, FastDiv fd_hi, FastDiv fd_r) {
  const float inv = 1.f / (float)(r * r);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    uint32_t t, ix, pl, iy;
    fd_divmod(i, fd_wi, t, ix);
    fd_divmod(t, fd_hi, pl, iy);
    const uint32_t oy = fd_div(iy, fd_r), ox = fd_div(ix, fd_r);
    dx[i] = (oy < (uint32_t)Ho && ox < (uint32_t)Wo) ? dy[(size_t)pl * Ho * Wo + (size_t)oy * Wo + ox] * inv : 0.f;
  }
}

// Global average over HW (AdaptiveAvgPool2d(1)): one wave per plane.
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         long long planes, int HW) {
  const long long pl = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < HW; i += 64) s += x[(size_t)pl * HW + i];
  s = wave_sum(s);
  if (lane == 0) y[pl] = s / (float)HW;
}

// ---------------------------------------------------------------------------------------------
// Element-wise activations.  mode: 0 relu, 1 gelu (erf form).
// ---------------------------------------------------------------------------------------------
// (gelu_f / gelu_grad live in ge_common.h: the BatchNorm kernels fuse the same function)

// mode 0 relu, 1 gelu (erf), 2 leaky relu (slope), 3 hardswish = x * relu6(x + 3) / 6.
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n,
                                                      int mode, float slope) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = x[i];
    float r;
    if (mode == 0) r = fmaxf(v, 0.f);
    else if (mode == 1) r = gelu_f(v);
    else if (mode == 2) r = v > 0.f ? v : v * slope;
    else r = v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
    y[i] = r;
  }
}
// relu: ref = output (mask out > 0); every other mode: ref = input.
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref,
                                                      float* __restrict__ dx, long long n, int mode, float slope) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float g = dy[i], v = ref[i];
    float r;
    if (mode == 0) r = v > 0.f ? g : 0.f;
    else if (mode == 1) r = g * gelu_grad(v);
    else if (mode == 2) r = v > 0.f ? g : g * slope;
    else r = v <= -3.f ? 0.f : (v < 3.f ? g * (v / 3.f + 0.5f) : g);   // torch's choice at the two kinks
    dx[i] = r;
  }
}

// Reductions over the last (neighbour) dimension of [rows][K] edge tensors: max with the first arg-max saved, sum.
__global__ __launch_bounds__(256) void lastdim_max_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              unsigned char* __restrict__ arg, long long rows, int K) {
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float* xp = x + r * K;
    float best = xp[0];
    int bk = 0;
    for (int k = 1; k < K; ++k) {
      const float v = xp[k];
      if (v > best || (v != v && best == best)) {
        best = v;
        bk = k;
      }
    }
    y[r] = best;
    arg[r] = (unsigned char)bk;
  }
}
__global__ __launch_bounds__(256) void lastdim_max_bwd_kernel(const float* __restrict__ dy,
                                                              const unsigned char* __restrict__ arg,
                                                              float* __restrict__ dx, long long total, FastDiv fd_k) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    uint32_t r, k;
    fd_divmod((uint32_t)i, fd_k, r, k);
    dx[i] = arg[r] == k ? dy[r] : 0.f;
  }
}
__global__ __launch_bounds__(256) void lastdim_sum_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          long long rows, int K) {
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < rows; r += (long long)gridDim.x * 256) {
    const float* xp = x + r * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += xp[k];
    y[r] = s;
  }
}
__global__ __launch_bounds__(256) void lastdim_bcast_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            long long total, FastDiv fd_k) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    dx[i] = dy[fd_div((uint32_t)i, fd_k)];
}

// out[c] = sum over (b, hw) of x[b][c][hw]  (conv bias gradient): one workgroup per (b, c) plane, then a
// short per-channel sum over the B partials.
__global__ void channel_sum_partial_kernel(const float* __restrict__ x,
                                                                  float* __restrict__ partial, int C, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const float* xp = x + ((size_t)b * C + c) * HW;
  float s = 0.f;
  if ((HW & 3) == 0) {
    const float4* x4 = (const float4*)xp;
    for (int i = threadIdx.x; i < (HW >> 2); i += blockDim.x) {
      const float4 v = x4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += xp[i];
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[(size_t)b * C + c] = s;
}
// Single-launch form: one workgroup per channel walks all B planes (used when there are enough channels to fill the
// chip, or the whole reduction is small).
__global__ __launch_bounds__(256) void channel_sum_direct_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                 int B, int C, int HW, int accumulate) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  float s0 = 0.f, s1 = 0.f;
  if ((HW & 3) == 0) {
    const int n4 = HW >> 2;
    for (int b = 0; b < B; ++b) {
      const float4* x4 = (const float4*)(x + ((size_t)b * C + c) * HW);
      int i = threadIdx.x;
      for (; i + 256 < n4; i += 512) {
        const float4 v = x4[i], w = x4[i + 256];
        s0 += (v.x + v.y) + (v.z + v.w);
        s1 += (w.x + w.y) + (w.z + w.w);
      }
      if (i < n4) {
        const float4 v = x4[i];
        s0 += (v.x + v.y) + (v.z + v.w);
      }
    }
  } else {
    for (int b = 0; b < B; ++b) {
      const float* xp = x + ((size_t)b * C + c) * HW;
      for (int i = threadIdx.x; i < HW; i += 256) s0 += xp[i];
    }
  }
  const float s = block_sum(s0 + s1, red);
  if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + s;
}
__global__ void channel_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int C,
                                         int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = accumulate ? out[c] : 0.f;
  for (int b = 0; b < B; ++b) s += partial[(size_t)b * C + c];
  out[c] = s;
}

extern "C" {

int ge_upsample_bilinear_fwd(const float* x, const float* add, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                             void* stream) {
  GE_REQUIRE(x && y && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "upsample_fwd: bad arguments");
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long long planes = (long long)B * C;
  const bool vec = (Wo & 3) == 0;
  const uint32_t per_plane = (uint32_t)Ho * (uint32_t)(Wo / (vec ? 4 : 1));
  GE_REQUIRE((long long)Ho * Wo < (1ll << 30), "upsample_fwd: plane too large");
  const uint32_t ppc = (uint32_t)std::max(1ll, std::min(planes, ((1ll << 31) - 256) / per_plane));
  const dim3 grid(ge_cdiv((long long)ppc * per_plane, 256), ge_cdiv(planes, ppc));
  const FastDiv fdw = make_fastdiv((uint32_t)(Wo / (vec ? 4 : 1))), fdp = make_fastdiv(per_plane);
  if (vec)
    hipLaunchKernelGGL(upsample_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, add, y, planes, Hi, Wi, Ho,
                       Wo, sh, sw, fdw, fdp, ppc);
  else
    hipLaunchKernelGGL(upsample_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, add, y, planes, Hi, Wi,
                       Ho, Wo, sh, sw, fdw, fdp, ppc);
  GE_CHECK_LAUNCH("upsample_fwd");
  return GE_OK;
}

int ge_upsample_bilinear_bwd(const float* dy, float* dx, int B, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
  GE_REQUIRE(dy && dx && B > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, "upsample_bwd: bad arguments");
  const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f;
  const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  const long long planes = (long long)B * C;
  const uint32_t per_plane = (uint32_t)Hi * (uint32_t)Wi;
  GE_REQUIRE((long long)Hi * Wi < (1ll << 30), "upsample_bwd: plane too large");
  const uint32_t ppc = (uint32_t)std::max(1ll, std::min(planes, ((1ll << 31) - 256) / per_plane));
  const dim3 grid(ge_cdiv((long long)ppc * per_plane, 256), ge_cdiv(planes, ppc));
  const float cand = sw > 0.f ? 2.f / sw + 4.f : (float)Wo;   // widest candidate range of the column loop
  static const int stream_on = getenv("GE_UPSAMPLE_BWD_STREAM") ? atoi(getenv("GE_UPSAMPLE_BWD_STREAM")) : 1;
  if (stream_on && Ho >= Hi && Wo >= Wi && Wo <= 256 && Hi <= 256 && (size_t)Hi * (Wo + 1) * 4 <= 48 * 1024) {
    // planes per workgroup: fill 256 threads with columns, but keep T within ~20 KB so that several workgroups fit
    // beside a weight-gradient workgroup; at least as many threads as T rows (pass 2)
    int G = std::max(1, 256 / Wo);
    while (G > 1 && (size_t)G * Hi * (Wo + 1) * 4 > 20 * 1024) --G;
    while (G > 1 && G * Hi > 256) --G;
    G = (int)std::min<long long>(G, planes);
    const int threads = std::max(64, ge_cdiv(std::max(G * Wo, G * Hi), 64) * 64);
    const size_t lds_t = (size_t)G * Hi * (Wo + 1) * sizeof(float);
    if (threads <= 256) {
      hipLaunchKernelGGL(upsample_bwd_stream_kernel<8>, dim3((unsigned)ge_cdiv(planes, G)), dim3(threads), lds_t,
                         (hipStream_t)stream, dy, dx, planes, G, Hi, Wi, Ho, Wo, sh, sw, make_fastdiv((uint32_t)Wo),
                         make_fastdiv((uint32_t)Hi), make_fastdiv((uint32_t)Wi));
      GE_CHECK_LAUNCH("upsample_bwd_stream");
      return GE_OK;
    }
  }
  const size_t lds = ((size_t)Ho * (Wo + 1) + (size_t)Wi * (Ho + 1)) * sizeof(float);
  if (lds <= 64 * 1024 && planes <= 0x7fffffffll) {
        GE_MAX_LDS(64 * 1024, (const void*)upsample_bwd_sep_kernel);
    hipLaunchKernelGGL(upsample_bwd_sep_kernel, dim3((unsigned)planes), dim3(256), lds, (hipStream_t)stream, dy, dx, Hi,
                       Wi, Ho, Wo, sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f,
                       make_fastdiv((uint32_t)Wo), make_fastdiv((uint32_t)Ho), make_fastdiv((uint32_t)Wi));
    GE_CHECK_LAUNCH("upsample_bwd_sep");
    return GE_OK;
  }
  if (cand <= 12.f)
    hipLaunchKernelGGL(upsample_bwd_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, planes, Hi, Wi, Ho, Wo,
                       sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f, make_fastdiv((uint32_t)Wi),
                       make_fastdiv(per_plane), ppc);
  else
    hipLaunchKernelGGL(upsample_bwd_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, dy, dx, planes, Hi, Wi, Ho, Wo,
                       sh, sw, sh > 0.f ? 1.f / sh : 0.f, sw > 0.f ? 1.f / sw : 0.f, make_fastdiv((uint32_t)Wi),
                       make_fastdiv(per_plane), ppc);
  GE_CHECK_LAUNCH("upsample_bwd");
  return GE_OK;
}

int ge_maxpool2d_fwd(const float* x, float* y, unsigned char* arg, int B, int C, int Hi, int Wi, int Ho, int Wo, int k,
                     int s, int p, void* stream) {
  GE_REQUIRE(x && y && arg && k > 0 && k * k < 255 && s > 0, "maxpool_fwd: bad arguments");
  const long long planes = (long long)B * C;
  GE_REQUIRE(planes * Hi * Wi < (1ll << 31), "maxpool_fwd: more than 2^31 elements");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ge_stream_grid(planes * Ho * Wo, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, arg, (uint32_t)(planes * Ho * Wo), Hi, Wi, Ho, Wo, k, s, p,
                     make_fastdiv((uint32_t)Wo), make_fastdiv((uint32_t)Ho));
  GE_CHECK_LAUNCH("maxpool_fwd");
  return GE_OK;
}

int ge_maxpool2d_bwd(const float* dy, const unsigned char* arg, float* dx, int B, int C, int Hi, int Wi, int Ho,
                     int Wo, int k, int s, int p, void* stream) {
  GE_REQUIRE(dy && dx && arg && k > 0 && s > 0, "maxpool_bwd: bad arguments");
  const long long planes = (long long)B * C;
  GE_REQUIRE(planes * Hi * Wi < (1ll << 31), "maxpool_bwd: more than 2^31 elements");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ge_stream_grid(planes * Hi * Wi, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, arg, dx, (uint32_t)(planes * Hi * Wi), Hi, Wi, Ho, Wo, k, s, p,
                     make_fastdiv((uint32_t)Wi), make_fastdiv((uint32_t)Hi), make_fastdiv((uint32_t)s));
  GE_CHECK_LAUNCH("maxpool_bwd");
  return GE_OK;
}

int ge_avgpool2d_fwd(const float* x, float* y, int B, int C, int Hi, int Wi, int r, void* stream) {
  GE_REQUIRE(x && y && r > 0 && Hi >= r && Wi >= r, "avgpool_fwd: bad arguments");
  const long long planes = (long long)B * C;
  const int Ho = Hi / r, Wo = Wi / r;
  GE_REQUIRE(planes * Hi * Wi < (1ll << 31), "avgpool_fwd: more than 2^31 elements");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ge_stream_grid(planes * Ho * Wo, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, (uint32_t)(planes * Ho * Wo), Hi, Wi, Ho, Wo, r,
                     make_fastdiv((uint32_t)Wo), make_fastdiv((uint32_t)Ho));
  GE_CHECK_LAUNCH("avgpool_fwd");
  return GE_OK;
}

int ge_avgpool2d_bwd(const float* dy, float* dx, int B, int C, int Hi, int Wi, int r, void* stream) {
  GE_REQUIRE(dy && dx && r > 0 && Hi >= r && Wi >= r, "avgpool_bwd: bad arguments");
  const long long planes = (long long)B * C;
  const int Ho = Hi / r, Wo = Wi / r;
  GE_REQUIRE(planes * Hi * Wi < (1ll << 31), "avgpool_bwd: more than 2^31 elements");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ge_stream_grid(planes * Hi * Wi, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, (uint32_t)(planes * Hi * Wi), Hi, Wi, Ho, Wo, r,
                     make_fastdiv((uint32_t)Wi), make_fastdiv((uint32_t)Hi), make_fastdiv((uint32_t)r));
  GE_CHECK_LAUNCH("avgpool_bwd");
  return GE_OK;
}

int ge_plane_mean(const float* x, float* y, long long planes, int HW, void* stream) {
  GE_REQUIRE(x && y && planes > 0 && HW > 0, "plane_mean: bad arguments");
  hipLaunchKernelGGL(plane_mean_kernel, dim3(ge_cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, x, y, planes, HW);
  GE_CHECK_LAUNCH("plane_mean");
  return GE_OK;
}

int ge_act_fwd(const float* x, float* y, long long n, int mode, float slope, void* stream) {
  GE_REQUIRE(x && y && n > 0 && mode >= 0 && mode <= 3, "act_fwd: bad arguments");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, mode,
                     slope);
  GE_CHECK_LAUNCH("act_fwd");
  return GE_OK;
}

int ge_act_bwd(const float* dy, const float* ref, float* dx, long long n, int mode, float slope, void* stream) {
  GE_REQUIRE(dy && ref && dx && n > 0 && mode >= 0 && mode <= 3, "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, ref, dx, n,
                     mode, slope);
  GE_CHECK_LAUNCH("act_bwd");
  return GE_OK;
}

int ge_lastdim_max_fwd(const float* x, float* y, unsigned char* arg, long long rows, int K, void* stream) {
  GE_REQUIRE(x && y && arg && rows > 0 && K > 0 && K <= 255, "lastdim_max_fwd: bad arguments");
  hipLaunchKernelGGL(lastdim_max_fwd_kernel, dim3(ge_stream_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     arg, rows, K);
  GE_CHECK_LAUNCH("lastdim_max_fwd");
  return GE_OK;
}
int ge_lastdim_max_bwd(const float* dy, const unsigned char* arg, float* dx, long long rows, int K, void* stream) {
  GE_REQUIRE(dy && arg && dx && rows > 0 && K > 0 && rows * K < (1ll << 31), "lastdim_max_bwd: bad arguments");
  hipLaunchKernelGGL(lastdim_max_bwd_kernel, dim3(ge_stream_grid(rows * K, 256)), dim3(256), 0, (hipStream_t)stream, dy,
                     arg, dx, rows * K, make_fastdiv((uint32_t)K));
  GE_CHECK_LAUNCH("lastdim_max_bwd");
  return GE_OK;
}
int ge_lastdim_sum_fwd(const float* x, float* y, long long rows, int K, void* stream) {
  GE_REQUIRE(x && y && rows > 0 && K > 0, "lastdim_sum_fwd: bad arguments");
  hipLaunchKernelGGL(lastdim_sum_kernel, dim3(ge_stream_grid(rows, 256)), dim3(256), 0, (hipStream_t)stream, x, y, rows,
                     K);
  GE_CHECK_LAUNCH("lastdim_sum_fwd");
  return GE_OK;
}
int ge_lastdim_sum_bwd(const float* dy, float* dx, long long rows, int K, void* stream) {
  GE_REQUIRE(dy && dx && rows > 0 && K > 0 && rows * K < (1ll << 31), "lastdim_sum_bwd: bad arguments");
  hipLaunchKernelGGL(lastdim_bcast_kernel, dim3(ge_stream_grid(rows * K, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx,
                     rows * K, make_fastdiv((uint32_t)K));
  GE_CHECK_LAUNCH("lastdim_sum_bwd");
  return GE_OK;
}

int ge_channel_sum(const float* x, float* out, float* partial, int B, int C, int HW, int accumulate, void* stream) {
  GE_REQUIRE(x && out && partial && B > 0 && C > 0 && HW > 0, "channel_sum: bad arguments");
  if (C >= 128 || (long long)B * HW <= 65536) {
    hipLaunchKernelGGL(channel_sum_direct_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, B, C, HW,
                       accumulate);
    GE_CHECK_LAUNCH("channel_sum_direct");
    return GE_OK;
  }
  hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(C, B), dim3(HW >= 1024 ? 256 : 64), 0, (hipStream_t)stream, x,
                     partial, C, HW);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3(ge_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, partial, out, B,
                     C, accumulate);
  GE_CHECK_LAUNCH("channel_sum");
  return GE_OK;
}

}  // extern "C"
