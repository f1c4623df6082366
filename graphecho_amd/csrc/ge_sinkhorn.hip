// Sinkhorn kernels, fp32, log domain.
//
// (1) SinkhornDistance (utils/sinkhorn_distance.py:27-86): squared-L2 cost, uniform marginals, <= max_iter
//     u/v updates, early exit on mean_b sum_i |u - u_prev| < thresh.  The reference syncs to the host every
//     iteration (err.item()); here every iteration is computed on the device, the per-iteration duals and
//     errors are kept, and the finalize kernel picks the iteration the reference would have stopped at.
//     Backward uses the closed-form reverse sweep (u^t depends only on v^{t-1} and C; v^t only on u^t and C).
// (2) sinkhorn_rpm (models/graph_matching.py:637-689, slack=True): row/column log-normalisation of the
//     zero-padded matrix, restated in dual form X = A - rho_i - gamma_j so the matrix is read-only:
//       rho^t_i   = LSE_{j<=N2}(Abar_ij - gamma^{t-1}_j)   (slack column contributes exp(0))
//       gamma^t_j = LSE_{i<=N1}(Abar_ij - rho^t_i)         (slack row contributes exp(0))
#include "ge_common.h"

// ---------------------------------------------------------------------------------------------
// (1) SinkhornDistance
// ---------------------------------------------------------------------------------------------
// C[b][i][j] = sum_d (x[b][i][d] - y[b][j][d])^2
__global__ __launch_bounds__(256) void sd_cost_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      float* __restrict__ Cm, int P1, int P2, int D) {
  __shared__ float xs[16][65], ys[16][65];
  const int b = blockIdx.z;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int ti = threadIdx.x / 16, tj = threadIdx.x % 16;
  const float* xb = x + (size_t)b * P1 * D;
  const float* yb = y + (size_t)b * P2 * D;
  float acc = 0.f;
  for (int d0 = 0; d0 < D; d0 += 64) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int r = e / 64, d = e % 64;
      xs[r][d] = (i0 + r < P1 && d0 + d < D) ? xb[(size_t)(i0 + r) * D + d0 + d] : 0.f;
      ys[r][d] = (j0 + r < P2 && d0 + d < D) ? yb[(size_t)(j0 + r) * D + d0 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int d = 0; d < 64; ++d) {
      const float t = xs[ti][d] - ys[tj][d];
      acc = fmaf(t, t, acc);
    }
    __syncthreads();
  }
  if (i0 + ti < P1 && j0 + tj < P2) Cm[((size_t)b * P1 + i0 + ti) * P2 + j0 + tj] = acc;
}

// One workgroup per batch element runs all T iterations.
// uh [B][T+1][P1], vh [B][T+1][P2] (slot 0 = zeros), err [B][T].
// The cost tile and the current potentials live in LDS (dynamic: [P1][P2+1] + P1 + P2 floats) when they fit -- the
// sweeps are then pure LDS traffic, the column sweep conflict-free through the odd row pitch; otherwise (in_lds = 0)
// the tile is read from global memory each sweep.
__global__ __launch_bounds__(256) void sd_iter_kernel(const float* __restrict__ Cm, float* __restrict__ uh,
                                                      float* __restrict__ vh, float* __restrict__ err, int P1, int P2,
                                                      int T, float eps, int in_lds) {
  extern __shared__ __attribute__((aligned(16))) float ssd[];
  __shared__ float red[16];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* Cb = Cm + (size_t)b * P1 * P2;
  float* ub = uh + (size_t)b * (T + 1) * P1;
  float* vb = vh + (size_t)b * (T + 1) * P2;
  float* su = ssd;            // current u [P1]
  float* sv = su + P1;        // current v [P2]
  const float* Cs = Cb;
  int ld = P2;
  if (in_lds) {
    float* sC = sv + P2;
    ld = P2 + 1;
    for (int e = threadIdx.x; e < P1 * P2; e += 256) {
      const int i = e / P2, j = e - i * P2;
      sC[i * ld + j] = Cb[e];
    }
    Cs = sC;
  }
  const float logmu = logf(1.f / (float)P1 + 1e-8f), lognu = logf(1.f / (float)P2 + 1e-8f);
  const float inv_eps = 1.f / eps;
  for (int i = threadIdx.x; i < P1; i += 256) {
    ub[i] = 0.f;
    su[i] = 0.f;
  }
  for (int j = threadIdx.x; j < P2; j += 256) {
    vb[j] = 0.f;
    sv[j] = 0.f;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    float* u1 = ub + (size_t)(t + 1) * P1;
    float* v1 = vb + (size_t)(t + 1) * P2;
    float e_acc = 0.f;
    for (int i = w; i < P1; i += 4) {  // u update: wave per row (only this wave touches su[i])
      const float ui = su[i];
      float mx = -INFINITY;
      for (int j = lane; j < P2; j += 64) mx = fmaxf(mx, (-Cs[(size_t)i * ld + j] + ui + sv[j]) * inv_eps);
      mx = wave_max(mx);
      float s = 0.f;
      for (int j = lane; j < P2; j += 64) s += expf((-Cs[(size_t)i * ld + j] + ui + sv[j]) * inv_eps - mx);
      s = wave_sum(s);
      const float un = eps * (logmu - (mx + logf(s))) + ui;
      if (lane == 0) {
        u1[i] = un;
        su[i] = un;
        e_acc += fabsf(un - ui);
      }
    }
    __syncthreads();
    for (int j = w; j < P2; j += 4) {  // v update with the new u: wave per column
      const float vj = sv[j];
      float mx = -INFINITY;
      for (int i = lane; i < P1; i += 64) mx = fmaxf(mx, (-Cs[(size_t)i * ld + j] + su[i] + vj) * inv_eps);
      mx = wave_max(mx);
      float s = 0.f;
      for (int i = lane; i < P1; i += 64) s += expf((-Cs[(size_t)i * ld + j] + su[i] + vj) * inv_eps - mx);
      s = wave_sum(s);
      if (lane == 0) {
        const float vn = eps * (lognu - (mx + logf(s))) + vj;
        v1[j] = vn;
        sv[j] = vn;
      }
    }
    const float e_tot = block_sum(e_acc, red);
    if (threadIdx.x == 0) err[(size_t)b * T + t] = e_tot;
    __syncthreads();
  }
}

// nits = first t (1-based) with mean_b err[b][t-1] < thresh, else T.  pi = exp((-C+u+v)/eps), cost_b = sum pi*C.
__global__ __launch_bounds__(256) void sd_finalize_kernel(const float* __restrict__ Cm, const float* __restrict__ uh,
                                                          const float* __restrict__ vh, const float* __restrict__ err,
                                                          float* __restrict__ pi, float* __restrict__ cost,
                                                          int* __restrict__ nits, int B, int P1, int P2, int T,
                                                          float eps, float thresh) {
  __shared__ float red[16];
  __shared__ int s_n;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    int n = T;
    for (int t = 0; t < T; ++t) {
      float m = 0.f;
      for (int bb = 0; bb < B; ++bb) m += err[(size_t)bb * T + t];
      m /= (float)B;
      if (m < thresh) {
        n = t + 1;
        break;
      }
    }
    s_n = n;
    if (b == 0) nits[0] = n;
  }
  __syncthreads();
  const int n = s_n;
  const float* u = uh + ((size_t)b * (T + 1) + n) * P1;
  const float* v = vh + ((size_t)b * (T + 1) + n) * P2;
  const float* Cb = Cm + (size_t)b * P1 * P2;
  float* pb = pi + (size_t)b * P1 * P2;
  const float inv_eps = 1.f / eps;
  float acc = 0.f;
  for (int e = threadIdx.x; e < P1 * P2; e += 256) {
    const int i = e / P2, j = e - i * P2;
    const float c = Cb[e];
    const float p = expf((-c + u[i] + v[j]) * inv_eps);
    pb[e] = p;
    acc += p * c;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) cost[b] = acc;
}

// ---------------------------------------------------------------------------------------------
// The whole forward in ONE launch (cost tile, all iterations, stopping rule, plan and cost): one 1024-thread workgroup
// per batch element.  The cost tile is computed straight into LDS (and written to HBM once, for the caller and the
// backward) in two layouts, [i][j] and [j][i], both with a row pitch == 16 mod 32 words: a sweep gives every
// 16-lane DPP row one row (u update) or one column (v update) of the tile -- four per wave, 64 per pass of the 16 waves,
// i.e. ONE pass for the 64 x 64 problems of TGCN -- with conflict-free LDS reads and 4-step DPP row reductions; no
// cross-wave traffic inside a sweep.  (The 3-launch form spent 69 us of its 85 us in the iteration kernel: four waves
// walking 16 rows each, one dependent max / exp / sum / log chain per row.)  The stopping rule needs every batch
// element's error: the workgroups meet at a counter (agent-scope release / relaxed poll / acquire, as
// cdna_hip_programming.md guideline 16 prescribes; all B <= 128 workgroups are resident) and the last one to leave
// resets it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1>(v, 0.f);
  v += dpp_f32<0x4E>(v, 0.f);
  v += dpp_f32<0x141>(v, 0.f);
  v += dpp_f32<0x140>(v, 0.f);
  return v;      // every lane of a 16-lane row holds the row's total
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v, v));
  v = fmaxf(v, dpp_f32<0x4E>(v, v));
  v = fmaxf(v, dpp_f32<0x141>(v, v));
  v = fmaxf(v, dpp_f32<0x140>(v, v));
  return v;
}
// smallest pitch >= n that is == 16 mod 32 words: the two DPP rows of a 32-lane LDS read group then hit disjoint banks
__host__ __device__ __forceinline__ int sd_pitch(int n) { return ((n + 15) / 32) * 32 + 16; }

static size_t sd_fused_lds(int P1, int P2) {      // tile in both orientations, duals, two operand stages, partial patches
  return ((size_t)P1 * sd_pitch(P2) + (size_t)P2 * sd_pitch(P1) + ((P1 + 3) & ~3) + ((P2 + 3) & ~3) + 2 * 64 * 68 +
          4 * 64 * 68) * sizeof(float);
}

__global__ __launch_bounds__(1024) void sd_fused_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ Cm, float* __restrict__ pi,
                                                        float* __restrict__ cost, int* __restrict__ nits,
                                                        float* __restrict__ uh, float* __restrict__ vh,
                                                        float* __restrict__ err, int* __restrict__ sync, int B, int P1,
                                                        int P2, int D, int T, float eps, float thresh) {
  extern __shared__ __attribute__((aligned(16))) float ssd[];
  __shared__ float red[16];
  __shared__ int s_n;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ldc = sd_pitch(P2), ldt = sd_pitch(P1);
  float* sC = ssd;                     // [P1][ldc]
  float* sT = sC + (size_t)P1 * ldc;   // [P2][ldt]  (transposed copy)
  float* su = sT + (size_t)P2 * ldt;   // [P1]
  float* sv = su + ((P1 + 3) & ~3);    // [P2]
  float* xs = sv + ((P2 + 3) & ~3);    // [64 d][68]  (16-byte aligned: su, sv start on 4-float boundaries below)
  float* ys = xs + 64 * 68;            // [64 d][68]
  float* part = ys + 64 * 68;          // [4 groups][64][68] partial cost patches
  const float* xb = x + (size_t)b * P1 * D;
  const float* yb = y + (size_t)b * P2 * D;
  float* Cb = Cm + (size_t)b * P1 * P2;

  // ---- cost tile, 64 x 64 blocks: thread = a 4 x 4 patch (rows ti0.., columns tj0..) for ONE QUARTER of the feature
  // dimension (group = tid / 256 takes d = 16 g .. 16 g + 15 of every 64-wide chunk); operands staged d-major so that the
  // four x values and the four y values of a step are one 16-byte LDS read each (2 reads per 16 FMAs; the first form read
  // one word per FMA and spent 36 of its 58 us here); the next chunk's global loads fly under the current chunk's FMAs;
  // the four partial patches meet in LDS and are added in group order. ----
  const int grp = tid >> 8, t8 = tid & 255;
  const int ti0 = (t8 >> 4) * 4, tj0 = (t8 & 15) * 4;
  for (int i0 = 0; i0 < P1; i0 += 64)
    for (int j0 = 0; j0 < P2; j0 += 64) {
      float acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[a][k] = 0.f;
      float rx[4], ry[4];      // this thread's share of a chunk: element e = tid + 1024 q -> row e / 64, d = e % 64
      auto gload = [&](int d0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = tid + 1024 * q, r = e >> 6, d = e & 63;
          rx[q] = (i0 + r < P1 && d0 + d < D) ? xb[(size_t)(i0 + r) * D + d0 + d] : 0.f;
          ry[q] = (j0 + r < P2 && d0 + d < D) ? yb[(size_t)(j0 + r) * D + d0 + d] : 0.f;
        }
      };
      gload(0);
      for (int d0 = 0; d0 < D; d0 += 64) {
        __syncthreads();                 // the previous chunk's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = tid + 1024 * q, r = e >> 6, d = e & 63;
          xs[d * 68 + r] = rx[q];
          ys[d * 68 + r] = ry[q];
        }
        __syncthreads();
        if (d0 + 64 < D) gload(d0 + 64);
#pragma unroll 4
        for (int dd = 0; dd < 16; ++dd) {
          const int d = grp * 16 + dd;
          const float4 xv = *(const float4*)(xs + d * 68 + ti0);
          const float4 yv = *(const float4*)(ys + d * 68 + tj0);
          const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ya[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float t = xa[a] - ya[k];
              acc[a][k] = fmaf(t, t, acc[a][k]);
            }
        }
      }
      // partial patches -> LDS [group][64][64+4], summed in group order by the thread that stores the entry
#pragma unroll
      for (int a = 0; a < 4; ++a)
        *(float4*)(part + ((size_t)grp * 64 + ti0 + a) * 68 + tj0) = make_float4(acc[a][0], acc[a][1], acc[a][2], acc[a][3]);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = tid + 1024 * q, r = e >> 6, c = e & 63;
        const float v = ((part[(size_t)r * 68 + c] + part[((size_t)64 + r) * 68 + c]) + part[((size_t)128 + r) * 68 + c]) +
                        part[((size_t)192 + r) * 68 + c];
        const int i = i0 + r, j = j0 + c;
        if (i < P1 && j < P2) {
          Cb[(size_t)i * P2 + j] = v;
          sC[(size_t)i * ldc + j] = v;
          sT[(size_t)j * ldt + i] = v;
        }
      }
      __syncthreads();
    }
  float* ub = uh + (size_t)b * (T + 1) * P1;
  float* vb = vh + (size_t)b * (T + 1) * P2;
  for (int i = tid; i < P1; i += 1024) {
    ub[i] = 0.f;
    su[i] = 0.f;
  }
  for (int j = tid; j < P2; j += 1024) {
    vb[j] = 0.f;
    sv[j] = 0.f;
  }
  __syncthreads();

  // ---- iterations: a 16-lane DPP row owns one row (column) of the tile ----
  const float logmu = logf(1.f / (float)P1 + 1e-8f), lognu = logf(1.f / (float)P2 + 1e-8f);
  const float inv_eps = 1.f / eps;
  const int l16 = lane & 15, slot = w * 4 + (lane >> 4);
  for (int t = 0; t < T; ++t) {
    float* u1 = ub + (size_t)(t + 1) * P1;
    float* v1 = vb + (size_t)(t + 1) * P2;
    float e_acc = 0.f;
    for (int i0 = 0; i0 < P1; i0 += 64) {
      const int i = i0 + slot;
      const bool ok = i < P1;
      const float ui = ok ? su[i] : 0.f;
      const float* row = sC + (size_t)(ok ? i : 0) * ldc;
      float mx = -INFINITY;
      for (int j = l16; j < P2; j += 16) mx = fmaxf(mx, (-row[j] + ui + sv[j]) * inv_eps);
      mx = row16_max(mx);
      float sm = 0.f;
      for (int j = l16; j < P2; j += 16) sm += expf((-row[j] + ui + sv[j]) * inv_eps - mx);
      sm = row16_sum(sm);
      const float un = eps * (logmu - (mx + logf(sm))) + ui;
      if (ok && l16 == 0) {
        u1[i] = un;
        su[i] = un;
        e_acc += fabsf(un - ui);
      }
    }
    __syncthreads();
    for (int j0 = 0; j0 < P2; j0 += 64) {
      const int j = j0 + slot;
      const bool ok = j < P2;
      const float vj = ok ? sv[j] : 0.f;
      const float* col = sT + (size_t)(ok ? j : 0) * ldt;
      float mx = -INFINITY;
      for (int i = l16; i < P1; i += 16) mx = fmaxf(mx, (-col[i] + su[i] + vj) * inv_eps);
      mx = row16_max(mx);
      float sm = 0.f;
      for (int i = l16; i < P1; i += 16) sm += expf((-col[i] + su[i] + vj) * inv_eps - mx);
      sm = row16_sum(sm);
      if (ok && l16 == 0) {
        const float vn = eps * (lognu - (mx + logf(sm))) + vj;
        v1[j] = vn;
        sv[j] = vn;
      }
    }
    const float e_tot = block_sum(e_acc, red);      // (contains the barriers that publish sv)
    if (tid == 0) err[(size_t)b * T + t] = e_tot;
    __syncthreads();
  }

  // ---- stopping rule: first t with mean_b err[b][t] < thresh (the reference's host-side err.item() test) ----
  if (tid == 0) {
    if (B > 1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // this workgroup's err row (and histories) are out
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < B) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    int n = T;
    for (int tt = 0; tt < T; ++tt) {
      float m = 0.f;
      for (int bb = 0; bb < B; ++bb) m += err[(size_t)bb * T + tt];
      m /= (float)B;
      if (m < thresh) {
        n = tt + 1;
        break;
      }
    }
    s_n = n;
    if (b == 0) nits[0] = n;
    if (B > 1) {      // the last workgroup to leave resets the meeting point for the next launch on this stream
      if (__hip_atomic_fetch_add(&sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == B - 1) {
        __hip_atomic_store(&sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  __syncthreads();
  const int n = s_n;
  // ---- plan and cost at iteration n: the duals of slot n (this workgroup wrote them; slot T is still in su / sv) ----
  const float* un = ub + (size_t)n * P1;
  const float* vn = vb + (size_t)n * P2;
  float* pb = pi + (size_t)b * P1 * P2;
  float acc = 0.f;
  for (int e = tid; e < P1 * P2; e += 1024) {
    const int i = e / P2, j = e - i * P2;
    const float c = sC[(size_t)i * ldc + j];
    const float pv = expf((-c + un[i] + vn[j]) * inv_eps);
    pb[e] = pv;
    acc += pv * c;
  }
  acc = block_sum(acc, red);
  if (tid == 0) cost[b] = acc;
}

// dC from (g_cost[b], g_pi (nullable), g_C (nullable)); gu/gv scratch in LDS.
__global__ __launch_bounds__(256) void sd_bwd_kernel(const float* __restrict__ Cm, const float* __restrict__ uh,
                                                     const float* __restrict__ vh, const int* __restrict__ nits,
                                                     const float* __restrict__ g_cost, const float* __restrict__ g_pi,
                                                     const float* __restrict__ g_C, float* __restrict__ dC, int P1,
                                                     int P2, int T, float eps) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* gu = sh;        // [P1]
  float* gv = sh + P1;   // [P2]
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = nits[0];
  const float* Cb = Cm + (size_t)b * P1 * P2;
  const float* ub = uh + (size_t)b * (T + 1) * P1;
  const float* vb = vh + (size_t)b * (T + 1) * P2;
  float* dCb = dC + (size_t)b * P1 * P2;
  const float* gpb = g_pi ? g_pi + (size_t)b * P1 * P2 : nullptr;
  const float* gcb = g_C ? g_C + (size_t)b * P1 * P2 : nullptr;
  const float gc = g_cost ? g_cost[b] : 0.f;
  const float inv_eps = 1.f / eps;
  const float inv_mu = 1.f / (1.f / (float)P1 + 1e-8f), inv_nu = 1.f / (1.f / (float)P2 + 1e-8f);
  const float* un = ub + (size_t)n * P1;
  const float* vn = vb + (size_t)n * P2;

  // direct terms through pi and cost; gu via row sums, gv via column sums
  for (int i = w; i < P1; i += 4) {
    float rs = 0.f;
    for (int j = lane; j < P2; j += 64) {
      const size_t e = (size_t)i * P2 + j;
      const float c = Cb[e];
      const float p = expf((-c + un[i] + vn[j]) * inv_eps);
      const float gp = gc * c + (gpb ? gpb[e] : 0.f);
      dCb[e] = gc * p - gp * p * inv_eps + (gcb ? gcb[e] : 0.f);
      rs += gp * p * inv_eps;
    }
    rs = wave_sum(rs);
    if (lane == 0) gu[i] = rs;
  }
  for (int j = w; j < P2; j += 4) {
    float cs = 0.f;
    for (int i = lane; i < P1; i += 64) {
      const size_t e = (size_t)i * P2 + j;
      const float c = Cb[e];
      const float p = expf((-c + un[i] + vn[j]) * inv_eps);
      const float gp = gc * c + (gpb ? gpb[e] : 0.f);
      cs += gp * p * inv_eps;
    }
    cs = wave_sum(cs);
    if (lane == 0) gv[j] = cs;
  }
  __syncthreads();

  for (int t = n; t >= 1; --t) {
    const float* ut = ub + (size_t)t * P1;
    const float* vt = vb + (size_t)t * P2;
    const float* vp = vb + (size_t)(t - 1) * P2;
    // v^t(u^t, C):  S_ij = exp((-C+u^t_i+v^t_j)/eps)/nu';  dC += S*gv_j;  gu_i -= sum_j S*gv_j
    for (int i = w; i < P1; i += 4) {
      float rs = 0.f;
      for (int j = lane; j < P2; j += 64) {
        const size_t e = (size_t)i * P2 + j;
        const float s = expf((-Cb[e] + ut[i] + vt[j]) * inv_eps) * inv_nu * gv[j];
        dCb[e] += s;
        rs += s;
      }
      rs = wave_sum(rs);
      if (lane == 0) gu[i] -= rs;
    }
    __threadfence_block();
    __syncthreads();
    // u^t(v^{t-1}, C):  R_ij = exp((-C+u^t_i+v^{t-1}_j)/eps)/mu';  dC += R*gu_i;  gv^{t-1}_j = -sum_i R*gu_i
    for (int j = w; j < P2; j += 4) {
      float cs = 0.f;
      for (int i = lane; i < P1; i += 64) {
        const size_t e = (size_t)i * P2 + j;
        const float r = expf((-Cb[e] + ut[i] + vp[j]) * inv_eps) * inv_mu * gu[i];
        dCb[e] += r;
        cs += r;
      }
      cs = wave_sum(cs);
      if (lane == 0) gv[j] = -cs;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = threadIdx.x; i < P1; i += 256) gu[i] = 0.f;
    __syncthreads();
  }
}

// dx[b][i][d] = sum_j dC_ij * 2 (x_id - y_jd);  dy[b][j][d] = -sum_i dC_ij * 2 (x_id - y_jd)
__global__ __launch_bounds__(256) void sd_cost_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ dC, float* __restrict__ dx,
                                                          float* __restrict__ dy, int P1, int P2, int D) {
  const int b = blockIdx.y;
  const int row = blockIdx.x;  // rows 0..P1-1 -> dx rows, P1..P1+P2-1 -> dy rows
  const float* xb = x + (size_t)b * P1 * D;
  const float* yb = y + (size_t)b * P2 * D;
  const float* dCb = dC + (size_t)b * P1 * P2;
  if (row < P1) {
    const int i = row;
    for (int d = threadIdx.x; d < D; d += 256) {
      const float xv = xb[(size_t)i * D + d];
      float acc = 0.f;
      for (int j = 0; j < P2; ++j) acc += dCb[(size_t)i * P2 + j] * 2.f * (xv - yb[(size_t)j * D + d]);
      dx[((size_t)b * P1 + i) * D + d] = acc;
    }
  } else {
    const int j = row - P1;
    for (int d = threadIdx.x; d < D; d += 256) {
      const float yv = yb[(size_t)j * D + d];
      float acc = 0.f;
      for (int i = 0; i < P1; ++i) acc -= dCb[(size_t)i * P2 + j] * 2.f * (xb[(size_t)i * D + d] - yv);
      dy[((size_t)b * P2 + j) * D + d] = acc;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (2) sinkhorn_rpm in dual form
// ---------------------------------------------------------------------------------------------
// rho_i = LSE_j(A_ij - gamma_j) over j < N2 plus the slack column (value 0).  Wave per row.
__global__ __launch_bounds__(256) void rpm_row_kernel(const float* __restrict__ A, const float* __restrict__ gamma,
                                                      float* __restrict__ rho, int N1, int N2) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N1) return;
  const int lane = threadIdx.x & 63;
  const float* a = A + (size_t)blockIdx.y * N1 * N2 + (size_t)i * N2;
  const float* g = gamma + (size_t)blockIdx.y * N2;
  float mx = 0.f;  // slack entry
  for (int j = lane; j < N2; j += 64) mx = fmaxf(mx, a[j] - g[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < N2; j += 64) s += expf(a[j] - g[j] - mx);
  s = wave_sum(s) + expf(0.f - mx);
  if (lane == 0) rho[(size_t)blockIdx.y * N1 + i] = mx + logf(s);
}

// gamma_j = LSE_i(A_ij - rho_i) over i < N1 plus the slack row (value 0).
// Workgroup = 16 columns x 16 row groups (a wave reads four 64-byte row segments); each thread keeps four independent
// online (max, sum) pairs so the exp chain is N1/64 long, not N1/4; merged through LDS.
constexpr int RPM_CW = 16, RPM_RG = 16;
__device__ __forceinline__ void lse_push(float& m, float& s, float v) {
  if (v > m) {
    s = s * expf(m - v) + 1.f;
    m = v;
  } else {
    s += expf(v - m);
  }
}
__global__ __launch_bounds__(256) void rpm_col_kernel(const float* __restrict__ A, const float* __restrict__ rho,
                                                      float* __restrict__ gamma, int N1, int N2) {
  __shared__ float sm[RPM_RG][RPM_CW + 1], ss[RPM_RG][RPM_CW + 1];
  const int cl = threadIdx.x & (RPM_CW - 1), rg = threadIdx.x / RPM_CW;
  const int j = blockIdx.x * RPM_CW + cl;
  const float* a = A + (size_t)blockIdx.y * N1 * N2;
  const float* r = rho + (size_t)blockIdx.y * N1;
  float m[4], sv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    m[u] = -INFINITY;
    sv[u] = 0.f;
  }
  if (j < N2) {
    for (int i = rg; i < N1; i += 4 * RPM_RG) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = i + RPM_RG * u;
        v[u] = ii < N1 ? a[(size_t)ii * N2 + j] - r[ii] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + RPM_RG * u < N1) lse_push(m[u], sv[u], v[u]);
    }
  }
  float M = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
  float S = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (sv[u] > 0.f) S += sv[u] * expf(m[u] - M);
  sm[rg][cl] = M;
  ss[rg][cl] = S;
  __syncthreads();
  if (rg == 0 && j < N2) {
    float Mt = 0.f;  // slack entry
    for (int q = 0; q < RPM_RG; ++q) Mt = fmaxf(Mt, sm[q][cl]);
    float St = expf(0.f - Mt);
    for (int q = 0; q < RPM_RG; ++q)
      if (ss[q][cl] > 0.f) St += ss[q][cl] * expf(sm[q][cl] - Mt);
    gamma[(size_t)blockIdx.y * N2 + j] = Mt + logf(St);
  }
}


// ---- sinkhorn_rpm forward in ONE launch of a few co-operating workgroups (round 5) ---------------------------------------------------
// The training step's problems are B = 1, N1 ~ 230-290, N2 ~ 235-410.  The 41-launch chain above is fine in isolation (0.23 ms) but on
// GModule's stream every small dependent launch costs 10-20 us while the convolutions of the main stream hold the CUs (~0.9 ms for
// the chain); a single-workgroup resident form is bound by ONE CU's VALU (0.5 ms; profiles/r05_sinkhorn_rpm_resident.txt).  Here
// RPM_G workgroups split the ROWS: a workgroup keeps its <= RW * 4 rows x N2 columns in registers (wave = rows w, w + 4, ..; lane =
// columns l, l + 64, ..), the row log-sum-exp is local, the column log-sum-exp is a per-workgroup (max, sum) partial per column,
// published with agent-scope stores (the XCDs' L2s are not coherent for plain accesses), one barrier on a monotonic counter per
// iteration, and every workgroup merges the RPM_G partials of its columns.  Partials are double-buffered by iteration parity.
// A workgroup that waits only sleeps, but every one of the RPM_G must get a slot while the others spin: the launch is co-operative
// (ge_common.h: ge_launch_coresident -- the runtime checks co-residency on the device the call runs on; round 6).
constexpr int RPM_G = 16;
__device__ __forceinline__ void rpm_grid_barrier(int* counter, int target) {
  // every publishing thread drains its own agent-scope stores first: the workgroup barrier does not wait for the other waves'
  // outstanding global stores (vmcnt), and thread 0's release below only orders wave 0's
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}
__device__ __forceinline__ void rpm_pub(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float rpm_get(const float* p) {
  return __hip_atomic_load(const_cast<float*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// part: [2][RPM_G][N2][2] floats; counter: one int, zero at launch
template <int C, int RW>
__global__ __launch_bounds__(256) void rpm_coop_fwd_kernel(const float* __restrict__ A, float* __restrict__ X, float* __restrict__ rho_hist,
                                                           float* __restrict__ gamma_hist, float* part, int* counter, int N1, int N2,
                                                           int n_iters) {
  __shared__ float sm[4][64 * C], ss[4][64 * C], sg[64 * C];
  const int g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = (N1 + RPM_G - 1) / RPM_G;      // rows per workgroup (<= 4 RW)
  const int row0 = g * rb, rows = max(0, min(rb, N1 - row0));
  float a[RW][C];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int il = wave + 4 * r;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      a[r][c] = (il < rows && j < N2) ? A[(size_t)(row0 + il) * N2 + j] : -INFINITY;
    }
  }
  float gam[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    gam[c] = 0.f;
    if (g == 0 && wave == 0 && lane + 64 * c < N2) gamma_hist[lane + 64 * c] = 0.f;      // gamma^0
  }
  float rho[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) rho[r] = 0.f;
  for (int t = 0; t < n_iters; ++t) {
    const int par = t & 1;
    // rho_i = LSE_j(A_ij - gamma_j) over j < N2 plus the slack column (value 0): local
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      if (wave + 4 * r < rows) {
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, a[r][c] - gam[c]);
        mx = wave_max(mx);
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) sv += expf(a[r][c] - gam[c] - mx);
        sv = wave_sum(sv) + expf(0.f - mx);
        rho[r] = mx + logf(sv);
        if (lane == 0) rho_hist[(size_t)t * N1 + row0 + wave + 4 * r] = rho[r];
      }
    }
    // this workgroup's (max, sum) per column over its rows: the wave's rows in-thread, the four waves through LDS
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float m = -INFINITY, sv = 0.f;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (wave + 4 * r < rows) {
          const float v = a[r][c] - rho[r];
          if (v > m) {
            sv = sv * expf(m - v) + 1.f;
            m = v;
          } else {
            sv += expf(v - m);
          }
        }
      }
      sm[wave][lane + 64 * c] = m;
      ss[wave][lane + 64 * c] = sv;
    }
    __syncthreads();
    float* mine = part + ((size_t)(par * RPM_G + g) * N2) * 2;
    for (int j = threadIdx.x; j < N2; j += 256) {
      float M = fmaxf(fmaxf(sm[0][j], sm[1][j]), fmaxf(sm[2][j], sm[3][j]));
      float S = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (ss[w][j] > 0.f) S += ss[w][j] * expf(sm[w][j] - M);
      rpm_pub(mine + 2 * j, M);
      rpm_pub(mine + 2 * j + 1, S);
    }
    rpm_grid_barrier(counter, (t + 1) * RPM_G);
    // gamma_j = LSE over all rows plus the slack row (value 0): merge the RPM_G partials, columns split over the threads
    const float* all = part + ((size_t)par * RPM_G * N2) * 2;
    for (int j = threadIdx.x; j < N2; j += 256) {
      float pm[RPM_G], ps[RPM_G];
#pragma unroll
      for (int q = 0; q < RPM_G; ++q) {
        pm[q] = rpm_get(all + ((size_t)q * N2 + j) * 2);
        ps[q] = rpm_get(all + ((size_t)q * N2 + j) * 2 + 1);
      }
      float Mt = 0.f;      // slack entry
#pragma unroll
      for (int q = 0; q < RPM_G; ++q)
        if (ps[q] > 0.f) Mt = fmaxf(Mt, pm[q]);
      float St = expf(0.f - Mt);
#pragma unroll
      for (int q = 0; q < RPM_G; ++q)
        if (ps[q] > 0.f) St += ps[q] * expf(pm[q] - Mt);
      const float gj = Mt + logf(St);
      sg[j] = gj;
      if (g == 0) gamma_hist[(size_t)(t + 1) * N2 + j] = gj;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) gam[c] = lane + 64 * c < N2 ? sg[lane + 64 * c] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int il = wave + 4 * r;
    if (il < rows) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = lane + 64 * c;
        if (j < N2) X[(size_t)(row0 + il) * N2 + j] = a[r][c] - rho[r] - gam[c];
      }
    }
  }
}
// floats of workspace of the co-operative form (0: the sizes go to the chain); the first 16 bytes are the barrier counter
static long long rpm_coop_workspace(int B, int N1, int N2) {
  if (B != 1 || N2 > 512 || N1 > RPM_G * 4 * 10 || N1 < RPM_G) return 0;
  return 4 + 2ll * RPM_G * N2 * 2;
}
template <int C>
static int rpm_coop_launch(const float* A, float* X, float* rho_hist, float* gamma_hist, float* ws, int N1, int N2, int n_iters,
                           hipStream_t st) {
  float* part = ws + 4;
  int* counter = reinterpret_cast<int*>(ws);
  const int rb = (N1 + RPM_G - 1) / RPM_G;
  void* args[] = {&A, &X, &rho_hist, &gamma_hist, &part, &counter, &N1, &N2, &n_iters};
  const void* fn = rb <= 20 ? (const void*)rpm_coop_fwd_kernel<C, 5> : (const void*)rpm_coop_fwd_kernel<C, 10>;
  return ge_launch_coresident(fn, dim3(RPM_G), dim3(256), args, 0, st, "sinkhorn_rpm_fwd_coop");
}


// backward of the same: the rows' gradients (gA rows, g_rho) are local to their workgroup, the column sums that make g_gamma are
// exchanged -- one barrier per iteration (+ one for the initial column sums of gX).  part: [2][RPM_G][N2] floats.
template <int C, int RW>
__global__ __launch_bounds__(256) void rpm_coop_bwd_kernel(const float* __restrict__ A, const float* __restrict__ gX,
                                                           const float* __restrict__ rho_hist, const float* __restrict__ gamma_hist,
                                                           float* __restrict__ gA, float* part, int* counter, int N1, int N2, int n_iters) {
  __shared__ float ss[4][64 * C], sg[64 * C];
  const int g = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int rb = (N1 + RPM_G - 1) / RPM_G;
  const int row0 = g * rb, rows = max(0, min(rb, N1 - row0));
  float a[RW][C], ga[RW][C], grho[RW], ggam[C];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int il = wave + 4 * r;
    float rs = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      const bool ok = il < rows && j < N2;
      a[r][c] = ok ? A[(size_t)(row0 + il) * N2 + j] : -INFINITY;
      ga[r][c] = ok ? gX[(size_t)(row0 + il) * N2 + j] : 0.f;
      rs += ga[r][c];
    }
    grho[r] = -wave_sum(rs);      // g_rho^T_i = -sum_j gX_ij
  }
  int bar = 0;
  // column sums of this workgroup's rows -> partial -> barrier -> every workgroup: ggam_j = -sum over the workgroups
  auto exchange = [&](int par) {
    __syncthreads();
    float* mine = part + (size_t)(par * RPM_G + g) * N2;
    for (int j = threadIdx.x; j < N2; j += 256) rpm_pub(mine + j, (ss[0][j] + ss[1][j]) + (ss[2][j] + ss[3][j]));
    rpm_grid_barrier(counter, ++bar * RPM_G);
    const float* all = part + (size_t)par * RPM_G * N2;
    for (int j = threadIdx.x; j < N2; j += 256) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < RPM_G; ++q) t += rpm_get(all + (size_t)q * N2 + j);
      sg[j] = -t;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < C; ++c) ggam[c] = lane + 64 * c < N2 ? sg[lane + 64 * c] : 0.f;
  };
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float sv = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r) sv += ga[r][c];
    ss[wave][lane + 64 * c] = sv;
  }
  exchange(0);      // g_gamma^T_j = -sum_i gX_ij
  for (int t = n_iters; t >= 1; --t) {
    const float* rho_t = rho_hist + (size_t)(t - 1) * N1 + row0;
    const float* gam_t = gamma_hist + (size_t)t * N2;
    const float* gam_p = gamma_hist + (size_t)(t - 1) * N2;
    float gt[C], gp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int j = lane + 64 * c;
      gt[c] = j < N2 ? gam_t[j] : 0.f;
      gp[c] = j < N2 ? gam_p[j] : 0.f;
    }
    float ri[RW];
    // gamma^t step: w = exp(A - rho^t_i - gamma^t_j) g_gamma_j;  gA += w;  g_rho_i = base_i - sum_j w   (base: g_rho at t = T, else 0)
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int il = wave + 4 * r;
      ri[r] = il < rows ? rho_t[il] : 0.f;
      float rs = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float wv = expf(a[r][c] - ri[r] - gt[c]) * ggam[c];      // (-inf entries: exp = 0)
        ga[r][c] += wv;
        rs += wv;
      }
      rs = wave_sum(rs);
      grho[r] = (t == n_iters ? grho[r] : 0.f) - rs;
    }
    // rho^t step: w = exp(A - gamma^{t-1}_j - rho^t_i) g_rho_i;  gA += w;  g_gamma_j = -sum_i w
#pragma unroll
    for (int c = 0; c < C; ++c) {
      float sv = 0.f;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const float wv = expf(a[r][c] - gp[c] - ri[r]) * grho[r];
        ga[r][c] += wv;
        sv += wv;
      }
      ss[wave][lane + 64 * c] = sv;
    }
    exchange((n_iters - t + 1) & 1);
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int il = wave + 4 * r;
    if (il < rows) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int j = lane + 64 * c;
        if (j < N2) gA[(size_t)(row0 + il) * N2 + j] = ga[r][c];
      }
    }
  }
}
template <int C>
static int rpm_coop_launch_bwd(const float* A, const float* gX, const float* rho_hist, const float* gamma_hist, float* gA, float* ws, int N1,
                               int N2, int n_iters, hipStream_t st) {
  float* part = ws + 4;
  int* counter = reinterpret_cast<int*>(ws);
  const int rb = (N1 + RPM_G - 1) / RPM_G;
  void* args[] = {&A, &gX, &rho_hist, &gamma_hist, &gA, &part, &counter, &N1, &N2, &n_iters};
  const void* fn = rb <= 20 ? (const void*)rpm_coop_bwd_kernel<C, 5> : (const void*)rpm_coop_bwd_kernel<C, 10>;
  return ge_launch_coresident(fn, dim3(RPM_G), dim3(256), args, 0, st, "sinkhorn_rpm_bwd_coop");
}

// X = A - rho_i - gamma_j
__global__ __launch_bounds__(256) void rpm_out_kernel(const float* __restrict__ A, const float* __restrict__ rho,
                                                      const float* __restrict__ gamma, float* __restrict__ X, int N1,
                                                      int N2) {
  const long long total = (long long)N1 * N2;
  const size_t bo = (size_t)blockIdx.y * total;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / N2), j = (int)(e - (long long)i * N2);
    X[bo + e] = A[bo + e] - rho[(size_t)blockIdx.y * N1 + i] - gamma[(size_t)blockIdx.y * N2 + j];
  }
}

// Backward helpers.  mode 0: init  gA = gX, g_rho_i = -sum_j gX_ij.
//                    mode 1: gamma^t step: w = exp(A - rho_i - gamma_j) * g_gamma_j; gA += w; g_rho_i = base_i - sum_j w
//                    (base = existing g_rho when keep_rho, else 0)
__global__ __launch_bounds__(256) void rpm_bwd_row_kernel(const float* __restrict__ A, const float* __restrict__ gX,
                                                          const float* __restrict__ rho,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ g_gamma, float* __restrict__ gA,
                                                          float* __restrict__ g_rho, int N1, int N2, int mode,
                                                          int keep_rho) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N1) return;
  const int lane = threadIdx.x & 63;
  const size_t bo = (size_t)blockIdx.y * N1 * N2 + (size_t)i * N2;
  float rs = 0.f;
  if (mode == 0) {
    for (int j = lane; j < N2; j += 64) {
      const float g = gX[bo + j];
      gA[bo + j] = g;
      rs += g;
    }
    rs = wave_sum(rs);
    if (lane == 0) g_rho[(size_t)blockIdx.y * N1 + i] = -rs;
  } else {
    const float ri = rho[(size_t)blockIdx.y * N1 + i];
    const float* gm = gamma + (size_t)blockIdx.y * N2;
    const float* gg = g_gamma + (size_t)blockIdx.y * N2;
    for (int j = lane; j < N2; j += 64) {
      const float wv = expf(A[bo + j] - ri - gm[j]) * gg[j];
      gA[bo + j] += wv;
      rs += wv;
    }
    rs = wave_sum(rs);
    if (lane == 0) {
      const size_t o = (size_t)blockIdx.y * N1 + i;
      g_rho[o] = (keep_rho ? g_rho[o] : 0.f) - rs;
    }
  }
}

// mode 0: g_gamma_j = -sum_i gX_ij.
// mode 1: rho^t step: w = exp(A - gamma_prev_j - rho_i) * g_rho_i; gA += w; g_gamma_j = -sum_i w.
__global__ __launch_bounds__(256) void rpm_bwd_col_kernel(const float* __restrict__ A, const float* __restrict__ gX,
                                                          const float* __restrict__ rho,
                                                          const float* __restrict__ gamma_prev,
                                                          const float* __restrict__ g_rho, float* __restrict__ gA,
                                                          float* __restrict__ g_gamma, int N1, int N2, int mode) {
  __shared__ float ss[RPM_RG][RPM_CW + 1];
  const int cl = threadIdx.x & (RPM_CW - 1), rg = threadIdx.x / RPM_CW;
  const int j = blockIdx.x * RPM_CW + cl;
  const size_t bo = (size_t)blockIdx.y * N1 * N2;
  float s = 0.f;
  if (j < N2) {
    if (mode == 0) {
      for (int i = rg; i < N1; i += RPM_RG) s += gX[bo + (size_t)i * N2 + j];
    } else {
      const float gp = gamma_prev[(size_t)blockIdx.y * N2 + j];
      const float* r = rho + (size_t)blockIdx.y * N1;
      const float* gr = g_rho + (size_t)blockIdx.y * N1;
      for (int i = rg; i < N1; i += RPM_RG) {
        const size_t e = bo + (size_t)i * N2 + j;
        const float wv = expf(A[e] - gp - r[i]) * gr[i];
        gA[e] += wv;
        s += wv;
      }
    }
  }
  ss[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && j < N2) {
    float t = 0.f;
    for (int q = 0; q < RPM_RG; ++q) t += ss[q][cl];
    g_gamma[(size_t)blockIdx.y * N2 + j] = -t;
  }
}


// ---------------------------------------------------------------------------------------------
// (3) one-to-one matching loss of GModule._forward_aff (models/graph_matching.py:577-590) on the log plan X of sinkhorn_rpm:
//       M = exp(X);  target_ij = (lab1_i == lab2_j)
//       tp_i = M[i, argmax_j(M_ij * target_ij)] (first maximum);  tp_loss = mean_i(-0.25 (1 - tp_i)^2 log tp_i) / N1
//       fp_loss = sum_ij(-0.75 M_ij^2 log(1 - M_ij) (1 - target_ij)) / sum(1 - target) / sum(M (1 - target)).detach()
//     loss = tp_loss + fp_loss.  The reference spells this as a dozen element-wise / reduce ops (and autograd as two dozen more), each a
//     launch of a few microseconds on GModule's host-bound stream; here: rows, finalize, and one backward pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mo2o_rows_kernel(const float* __restrict__ X, const float* __restrict__ lab1,
                                                        const float* __restrict__ lab2, float* __restrict__ M, int* __restrict__ idx,
                                                        float* __restrict__ rowpart, int N1, int N2) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N1) return;
  const int lane = threadIdx.x & 63;
  const long long li = (long long)lab1[i];
  float best = -1.f;      // M * target >= 0: the first candidate always beats this
  int bj = 0x7FFFFFFF;
  float fp = 0.f, cnt = 0.f, mfp = 0.f;
  for (int j = lane; j < N2; j += 64) {
    const float m = expf(X[(size_t)i * N2 + j]);
    M[(size_t)i * N2 + j] = m;
    const bool same = li == (long long)lab2[j];
    const float t = same ? m : 0.f;
    if (t > best) {      // ascending j per lane: the lane's FIRST maximum
      best = t;
      bj = j;
    }
    if (!same) {
      fp += -0.75f * m * m * logf(1.f - m);
      cnt += 1.f;
      mfp += m;
    }
  }
  // arg-max across the wave, lowest index among equal values: one 64-bit maximum of (value bits, ~index) -- values are >= 0 or -1
  unsigned long long key = ((unsigned long long)(best < 0.f ? 0u : (__float_as_uint(best) + 1u)) << 32) | (unsigned)(~(unsigned)bj);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)key, off), hi = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), off);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    key = o > key ? o : key;
  }
  fp = wave_sum(fp);
  cnt = wave_sum(cnt);
  mfp = wave_sum(mfp);
  if (lane == 0) {
    const int jw = (int)(~(unsigned)key);
    idx[i] = jw;
    rowpart[(size_t)i * 4 + 0] = expf(X[(size_t)i * N2 + jw]);      // tp_i (the same expf as M's)
    rowpart[(size_t)i * 4 + 1] = fp;
    rowpart[(size_t)i * 4 + 2] = cnt;
    rowpart[(size_t)i * 4 + 3] = mfp;
  }
}
// loss and the three scalars the backward needs: scal = (sum fp terms, sum fp_mask, sum M fp_mask)
__global__ __launch_bounds__(256) void mo2o_final_kernel(const float* __restrict__ rowpart, float* __restrict__ loss,
                                                         float* __restrict__ scal, int N1) {
  __shared__ float red[16];
  float tp = 0.f, fp = 0.f, cnt = 0.f, mfp = 0.f;
  for (int i = threadIdx.x; i < N1; i += 256) {
    const float t = rowpart[(size_t)i * 4];
    tp += -0.25f * (1.f - t) * (1.f - t) * logf(t);
    fp += rowpart[(size_t)i * 4 + 1];
    cnt += rowpart[(size_t)i * 4 + 2];
    mfp += rowpart[(size_t)i * 4 + 3];
  }
  tp = block_sum(tp, red);
  fp = block_sum(fp, red);
  cnt = block_sum(cnt, red);
  mfp = block_sum(mfp, red);
  if (threadIdx.x == 0) {
    loss[0] = tp / (float)N1 / (float)N1 + fp / cnt / mfp;
    scal[0] = fp;
    scal[1] = cnt;
    scal[2] = mfp;
  }
}
// gX = (d loss / d M * g_loss + gM) * M   (M = exp(X); gM: gradient arriving at the returned M, nullable)
__global__ __launch_bounds__(256) void mo2o_bwd_kernel(const float* __restrict__ M, const float* __restrict__ lab1,
                                                       const float* __restrict__ lab2, const int* __restrict__ idx,
                                                       const float* __restrict__ scal, const float* __restrict__ g_loss,
                                                       const float* __restrict__ gM, float* __restrict__ gX, int N1, int N2) {
  const long long total = (long long)N1 * N2;
  const float g = g_loss ? g_loss[0] : 0.f;
  const float cf = g / scal[1] / scal[2], ct = g / (float)N1 / (float)N1;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / N2), j = (int)(e - (long long)i * N2);
    const float m = M[e];
    float d = gM ? gM[e] : 0.f;
    if ((long long)lab1[i] != (long long)lab2[j]) d += cf * -0.75f * (2.f * m * logf(1.f - m) - m * m / (1.f - m));
    if (j == idx[i]) d += ct * -0.25f * (-2.f * (1.f - m) * logf(m) + (1.f - m) * (1.f - m) / m);
    gX[e] = d * m;
  }
}

__global__ void fill_kernel(float* __restrict__ p, long long n, float v) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = v;
}

extern "C" {

// x [B][P1][D], y [B][P2][D] -> C, pi [B][P1][P2], cost [B], nits [1];
// uh [B][T+1][P1], vh [B][T+1][P2], err [B][T] are kept for the backward pass.
int ge_sinkhorn_distance_fwd(const float* x, const float* y, float* Cm, float* pi, float* cost, int* nits, float* uh,
                             float* vh, float* err, int B, int P1, int P2, int D, float eps, int max_iter,
                             float thresh, void* stream) {
  GE_REQUIRE(x && y && Cm && pi && cost && nits && uh && vh && err, "sinkhorn_distance_fwd: null pointer");
  GE_REQUIRE(B > 0 && P1 > 0 && P2 > 0 && D > 0 && max_iter >= 1 && eps > 0.f, "sinkhorn_distance_fwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sd_cost_kernel, dim3(ge_cdiv(P2, 16), ge_cdiv(P1, 16), B), dim3(256), 0, st, x, y, Cm, P1, P2, D);
  GE_CHECK_LAUNCH("sd_cost");
  const size_t lds_full = ((size_t)P1 * (P2 + 1) + P1 + P2) * sizeof(float);
  const int in_lds = lds_full <= 60 * 1024;
  GE_REQUIRE((size_t)(P1 + P2) * sizeof(float) <= 60 * 1024, "sinkhorn_distance_fwd: P1+P2 too large");
  hipLaunchKernelGGL(sd_iter_kernel, dim3(B), dim3(256), in_lds ? lds_full : (size_t)(P1 + P2) * sizeof(float), st, Cm,
                     uh, vh, err, P1, P2, max_iter, eps, in_lds);
  GE_CHECK_LAUNCH("sd_iter");
  hipLaunchKernelGGL(sd_finalize_kernel, dim3(B), dim3(256), 0, st, Cm, uh, vh, err, pi, cost, nits, B, P1, P2,
                     max_iter, eps, thresh);
  GE_CHECK_LAUNCH("sd_finalize");
  return GE_OK;
}

// The same in ONE launch (sd_fused_kernel).  sync: two ints owned by the caller, ONE PAIR PER (device, stream) -- two
// launches in flight at once must not share a meeting point; they are zeroed on the stream in front of every launch
// (a launch that faulted may have left them anywhere).
// The workgroups meet at a spin barrier, so all B of them must be resident at once: the launch is co-operative
// (ge_launch_coresident, round 6: the runtime refuses a grid that cannot be co-resident and does not interleave two such
// launches) and B is additionally held to a quarter of occupancy x CU count of THIS kernel at THIS LDS size on the device
// the call runs on.  Returns GE_OK, or a negative code WITHOUT
// launching when the problem does not fit: the caller then takes ge_sinkhorn_distance_fwd.
static int sd_fused_max_batch(size_t lds) {
  if (lds > 163000) return 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  static int cus[64];                  // 0: not queried yet
  static bool attr_set[64];
  if (!attr_set[dev]) {                // per device: the attribute belongs to the function's image on that device
    if (hipFuncSetAttribute((const void*)sd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163000) != hipSuccess)
      return 0;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
    attr_set[dev] = true;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)sd_fused_kernel, 1024, lds) != hipSuccess)
    return 0;
  const long resident = (long)per_cu * cus[dev];
  return (int)(resident / 4 < 128 ? resident / 4 : 128);
}

int ge_sinkhorn_distance_fwd_fused(const float* x, const float* y, float* Cm, float* pi, float* cost, int* nits, float* uh,
                                   float* vh, float* err, int* sync, int B, int P1, int P2, int D, float eps, int max_iter,
                                   float thresh, void* stream) {
  GE_REQUIRE(x && y && Cm && pi && cost && nits && uh && vh && err && sync, "sinkhorn_distance_fwd_fused: null pointer");
  GE_REQUIRE(B > 0 && P1 > 0 && P2 > 0 && D > 0 && max_iter >= 1 && eps > 0.f, "sinkhorn_distance_fwd_fused: bad shape");
  const size_t lds = sd_fused_lds(P1, P2);
  GE_REQUIRE(B == 1 || B <= sd_fused_max_batch(lds),
             "sinkhorn_distance_fwd_fused: problem too large for the one-launch form on this device");
  GE_REQUIRE(lds <= 163000 && sd_fused_max_batch(lds) >= 1, "sinkhorn_distance_fwd_fused: tile beyond LDS");
  if (B > 1) {      // meeting point of the workgroups := 0 (all-zero bits; a kernel, never a memset node: ge_common.h)
    ge_init_async(reinterpret_cast<float*>(sync), nullptr, 2, (hipStream_t)stream);
    GE_CHECK_LAUNCH("sd_fused_init");
  }
  if (B > 1) {      // the B workgroups meet at the stopping rule: co-operative launch (ge_common.h)
    void* args[] = {&x, &y, &Cm, &pi, &cost, &nits, &uh, &vh, &err, &sync, &B, &P1, &P2, &D, &max_iter, &eps, &thresh};
    const int rc = ge_launch_coresident((const void*)sd_fused_kernel, dim3(B), dim3(1024), args, lds, (hipStream_t)stream,
                                        "sinkhorn_distance_fwd_fused");
    if (rc != GE_OK) return rc;
  } else {
    hipLaunchKernelGGL(sd_fused_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, x, y, Cm, pi, cost, nits, uh, vh, err,
                       sync, B, P1, P2, D, max_iter, eps, thresh);
  }
  GE_CHECK_LAUNCH("sd_fused");
  return GE_OK;
}
// 1 when ge_sinkhorn_distance_fwd_fused takes this problem on the current device
int ge_sinkhorn_distance_fused_ok(int B, int P1, int P2) {
  const int cap = sd_fused_max_batch(sd_fused_lds(P1, P2));
  return cap >= 1 && (B == 1 || B <= cap);
}

// g_cost [B] / g_pi [B][P1][P2] / g_C [B][P1][P2] may each be null.  dC is a [B][P1][P2] workspace.
int ge_sinkhorn_distance_bwd(const float* x, const float* y, const float* Cm, const float* uh, const float* vh,
                             const int* nits, const float* g_cost, const float* g_pi, const float* g_C, float* dC,
                             float* dx, float* dy, int B, int P1, int P2, int D, float eps, int max_iter,
                             void* stream) {
  GE_REQUIRE(x && y && Cm && uh && vh && nits && dC && dx && dy, "sinkhorn_distance_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(P1 + P2) * sizeof(float);
  GE_REQUIRE(lds <= 64 * 1024, "sinkhorn_distance_bwd: P1+P2 too large");
  hipLaunchKernelGGL(sd_bwd_kernel, dim3(B), dim3(256), lds, st, Cm, uh, vh, nits, g_cost, g_pi, g_C, dC, P1, P2,
                     max_iter, eps);
  GE_CHECK_LAUNCH("sd_bwd");
  hipLaunchKernelGGL(sd_cost_bwd_kernel, dim3(P1 + P2, B), dim3(256), 0, st, x, y, dC, dx, dy, P1, P2, D);
  GE_CHECK_LAUNCH("sd_cost_bwd");
  return GE_OK;
}

// A [B][N1][N2] -> X = log-plan [B][N1][N2]; rho_hist [T][B][N1], gamma_hist [T+1][B][N2] (slot 0 = zeros).
// floats of workspace ge_sinkhorn_rpm_fwd_coop needs for this problem; 0: not offered (B > 1, N2 > 512, N1 > 640 or < 16)
long long ge_sinkhorn_rpm_coop_workspace(int B, int N1, int N2) {
  static const int on = []() {
    const char* e = getenv("GE_RPM_COOP");
    return e ? atoi(e) : 1;
  }();
  return on ? rpm_coop_workspace(B, N1, N2) : 0;
}

// the same result as ge_sinkhorn_rpm_fwd in ONE launch of 16 co-operating workgroups (B = 1); workspace: ge_sinkhorn_rpm_coop_workspace
// floats, owned by the caller, one per (device, stream) in flight
int ge_sinkhorn_rpm_fwd_coop(const float* A, float* X, float* rho_hist, float* gamma_hist, float* workspace, int N1, int N2, int n_iters,
                             void* stream) {
  GE_REQUIRE(A && X && rho_hist && gamma_hist && workspace && n_iters >= 1, "sinkhorn_rpm_fwd_coop: bad arguments");
  GE_REQUIRE(rpm_coop_workspace(1, N1, N2) > 0, "sinkhorn_rpm_fwd_coop: size not offered (N1 %d, N2 %d)", N1, N2);
  hipStream_t st = (hipStream_t)stream;
  ge_init_async(workspace, nullptr, 4, st);      // the barrier counter := 0 (a kernel, never a memset node)
  const int cb = (N2 + 63) / 64;
  int rc = GE_OK;
  switch (cb) {
    case 1: case 2: rc = rpm_coop_launch<2>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
    case 3: case 4: rc = rpm_coop_launch<4>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
    case 5: rc = rpm_coop_launch<5>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
    case 6: rc = rpm_coop_launch<6>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
    case 7: rc = rpm_coop_launch<7>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
    default: rc = rpm_coop_launch<8>(A, X, rho_hist, gamma_hist, workspace, N1, N2, n_iters, st); break;
  }
  if (rc != GE_OK) return rc;
  GE_CHECK_LAUNCH("sinkhorn_rpm_coop");
  return GE_OK;
}

// backward of ge_sinkhorn_rpm_fwd(_coop) in one launch (same sizes, same workspace)
int ge_sinkhorn_rpm_bwd_coop(const float* A, const float* gX, const float* rho_hist, const float* gamma_hist, float* gA, float* workspace,
                             int N1, int N2, int n_iters, void* stream) {
  GE_REQUIRE(A && gX && rho_hist && gamma_hist && gA && workspace && n_iters >= 1, "sinkhorn_rpm_bwd_coop: bad arguments");
  GE_REQUIRE(rpm_coop_workspace(1, N1, N2) > 0, "sinkhorn_rpm_bwd_coop: size not offered (N1 %d, N2 %d)", N1, N2);
  hipStream_t st = (hipStream_t)stream;
  ge_init_async(workspace, nullptr, 4, st);
  int rc = GE_OK;
  switch ((N2 + 63) / 64) {
    case 1: case 2: rc = rpm_coop_launch_bwd<2>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
    case 3: case 4: rc = rpm_coop_launch_bwd<4>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
    case 5: rc = rpm_coop_launch_bwd<5>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
    case 6: rc = rpm_coop_launch_bwd<6>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
    case 7: rc = rpm_coop_launch_bwd<7>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
    default: rc = rpm_coop_launch_bwd<8>(A, gX, rho_hist, gamma_hist, gA, workspace, N1, N2, n_iters, st); break;
  }
  if (rc != GE_OK) return rc;
  GE_CHECK_LAUNCH("sinkhorn_rpm_bwd_coop");
  return GE_OK;
}

int ge_sinkhorn_rpm_fwd(const float* A, float* X, float* rho_hist, float* gamma_hist, int B, int N1, int N2, int n_iters,
                        void* stream) {
  GE_REQUIRE(A && X && rho_hist && gamma_hist && B > 0 && N1 > 0 && N2 > 0 && n_iters >= 1,
             "sinkhorn_rpm_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fill_kernel, dim3(ge_cdiv((long long)B * N2, 256)), dim3(256), 0, st, gamma_hist,
                     (long long)B * N2, 0.f);
  for (int t = 0; t < n_iters; ++t) {
    float* rho = rho_hist + (size_t)t * B * N1;
    const float* g0 = gamma_hist + (size_t)t * B * N2;
    float* g1 = gamma_hist + (size_t)(t + 1) * B * N2;
    hipLaunchKernelGGL(rpm_row_kernel, dim3(ge_cdiv(N1, 4), B), dim3(256), 0, st, A, g0, rho, N1, N2);
    hipLaunchKernelGGL(rpm_col_kernel, dim3(ge_cdiv(N2, RPM_CW), B), dim3(256), 0, st, A, rho, g1, N1, N2);
  }
  GE_CHECK_LAUNCH("sinkhorn_rpm_iter");
  const float* rT = rho_hist + (size_t)(n_iters - 1) * B * N1;
  const float* gT = gamma_hist + (size_t)n_iters * B * N2;
  hipLaunchKernelGGL(rpm_out_kernel, dim3(ge_stream_grid((long long)N1 * N2, 256), B), dim3(256), 0, st, A, rT, gT, X,
                     N1, N2);
  GE_CHECK_LAUNCH("sinkhorn_rpm_out");
  return GE_OK;
}

// gA [B][N1][N2] = d loss / d A given gX; g_rho [B][N1], g_gamma [B][N2] are workspaces.
int ge_sinkhorn_rpm_bwd(const float* A, const float* gX, const float* rho_hist, const float* gamma_hist, float* gA,
                        float* g_rho, float* g_gamma, int B, int N1, int N2, int n_iters, void* stream) {
  GE_REQUIRE(A && gX && rho_hist && gamma_hist && gA && g_rho && g_gamma, "sinkhorn_rpm_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 rgrid(ge_cdiv(N1, 4), B), cgrid(ge_cdiv(N2, RPM_CW), B);
  hipLaunchKernelGGL(rpm_bwd_row_kernel, rgrid, dim3(256), 0, st, A, gX, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, gA, g_rho, N1, N2, 0, 0);
  hipLaunchKernelGGL(rpm_bwd_col_kernel, cgrid, dim3(256), 0, st, A, gX, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, gA, g_gamma, N1, N2, 0);
  for (int t = n_iters; t >= 1; --t) {
    const float* rho = rho_hist + (size_t)(t - 1) * B * N1;       // rho^t
    const float* gam = gamma_hist + (size_t)t * B * N2;           // gamma^t
    const float* gprev = gamma_hist + (size_t)(t - 1) * B * N2;   // gamma^{t-1}
    hipLaunchKernelGGL(rpm_bwd_row_kernel, rgrid, dim3(256), 0, st, A, gX, rho, gam, g_gamma, gA, g_rho, N1, N2, 1,
                       t == n_iters ? 1 : 0);
    hipLaunchKernelGGL(rpm_bwd_col_kernel, cgrid, dim3(256), 0, st, A, gX, rho, gprev, g_rho, gA, g_gamma, N1, N2, 1);
  }
  GE_CHECK_LAUNCH("sinkhorn_rpm_bwd");
  return GE_OK;
}

// Matching loss of GModule._forward_aff ("o2o") from the log plan X [N1][N2] and the float class labels of both node sets:
// M [N1][N2] = exp(X), idx [N1] (int32), rowpart [N1][4], loss [1], scal [3].  (models/graph_matching.py:577-590)
int ge_match_o2o_fwd(const float* X, const float* lab1, const float* lab2, float* M, int* idx, float* rowpart, float* loss,
                     float* scal, int N1, int N2, void* stream) {
  GE_REQUIRE(X && lab1 && lab2 && M && idx && rowpart && loss && scal && N1 > 0 && N2 > 0, "match_o2o_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mo2o_rows_kernel, dim3(ge_cdiv(N1, 4)), dim3(256), 0, st, X, lab1, lab2, M, idx, rowpart, N1, N2);
  hipLaunchKernelGGL(mo2o_final_kernel, dim3(1), dim3(256), 0, st, rowpart, loss, scal, N1);
  GE_CHECK_LAUNCH("match_o2o_fwd");
  return GE_OK;
}
// gX [N1][N2] from g_loss [1] (nullable: 0) and gM [N1][N2] (nullable: 0)
int ge_match_o2o_bwd(const float* M, const float* lab1, const float* lab2, const int* idx, const float* scal, const float* g_loss,
                     const float* gM, float* gX, int N1, int N2, void* stream) {
  GE_REQUIRE(M && lab1 && lab2 && idx && scal && gX && N1 > 0 && N2 > 0, "match_o2o_bwd: bad arguments");
  hipLaunchKernelGGL(mo2o_bwd_kernel, dim3(ge_stream_grid((long long)N1 * N2, 256)), dim3(256), 0, (hipStream_t)stream, M, lab1,
                     lab2, idx, scal, g_loss, gM, gX, N1, N2);
  GE_CHECK_LAUNCH("match_o2o_bwd");
  return GE_OK;
}

}  // extern "C"
