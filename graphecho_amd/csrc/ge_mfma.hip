// fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM kernels for gfx950:
//   * conv_gemm  : NCHW conv2d forward and data-gradient, im2col-free (patch gather straight into LDS)
//   * conv_wgrad : NCHW conv2d weight-gradient, split-K over B*Ho*Wo with a deterministic slab reduce
//   * gemm       : strided (batched) GEMM for Linear / attention / affinity projections
// Replaces the ATen/cuDNN/cuBLAS calls behind nn.Conv2d / nn.Linear / torch.bmm at the reference call
// sites listed in include/graphecho_hip.h.
//
// Tiling: a workgroup is 4 wave64s; each wave owns TM x TN accumulators of 32x32 (16 VGPRs each) and
// issues one 32x32x2 MFMA per (tile, k-pair).  Operand tiles are staged global -> registers -> LDS,
// double buffered, so the next chunk's global loads are in flight under the current chunk's MFMAs.
// LDS layouts are chosen per operand so that both the staging writes and the fragment reads are
// bank-conflict free: "t-fast" [k][T] when lanes walk the M/N axis, "k-fast" [T][KC+1] when lanes walk K.
#include "ge_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WM_, int WN_, int TM_, int TN_, int KC_>
struct TileCfg {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, KC = KC_;
  static constexpr int MT = WM * TM * 32, NT = WN * TN * 32;
  static constexpr int NTHREADS = WM * WN * 64;
};

// Row of the 32x32 accumulator held in register r by a lane in half `hi` (cdna_hip_programming.md §3).
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// XCD-aware bijective block remap: consecutive logical ids land on the same XCD (same L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One K-chunk of MFMAs out of LDS.  (SKA,STA)/(SKB,STB) are the k / t strides of the two LDS tiles.
template <int TM, int TN, int KC, int SKA, int STA, int SKB, int STB>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ sA, const float* __restrict__ sB, int a_off,
                                          int b_off, int lane, f32x16 (&acc)[TM][TN]) {
  const int li = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < KC; kk += 2) {
    float a[TM], b[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = sA[(kk + hi) * SKA + (a_off + i * 32 + li) * STA];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = sB[(kk + hi) * SKB + (b_off + j * 32 + li) * STB];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
  }
}

template <int TM, int TN>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// =========================================================================================
// conv_gemm: dst[b, g*M+m, y, x] = bias[m] + sum_k wp[g][k][m] * patch(k, (b,y,x))
//   forward   : k=(ci,kh,kw), patch = src[b, g*Cs_g+ci, y*s-p+kh, x*s-p+kw]
//   transposed: k=(co,kh,kw), patch = src[b, g*Cs_g+co, (y+p-kh)/s, (x+p-kw)/s] when divisible
// =========================================================================================
struct ConvGemmParams {
  const float* wp;
  const float* src;
  const float* bias;
  float* dst;
  int B, Hs, Ws, Hd, Wd, Cs_total, Cd_total, Cs_g;
  int M, N, K;
  int stride, pad, kh, kw;
  int relu;
  int tiles_m, tiles_n;
  FastDiv div_hw, div_w;
};

template <class T, int KH, int KW, bool TRANSPOSED>
__global__ __launch_bounds__(T::NTHREADS) void conv_gemm_kernel(ConvGemmParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = T::KC, NTH = T::NTHREADS;
  constexpr int STEP_A = NTH / MT, EA = KC / STEP_A;
  constexpr int STEP_B = NTH / NT, EB = KC / STEP_B;
  static_assert(NTH % MT == 0 && NTH % NT == 0 && KC % STEP_A == 0 && KC % STEP_B == 0, "tile/thread mismatch");
  constexpr int STAGE = KC * (MT + NT);
  __shared__ float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tn = lid / p.tiles_m;
  const int m0 = tm * MT, n0 = tn * NT;
  const int kh_n = KH ? KH : p.kh, kw_n = KW ? KW : p.kw;
  const int khw = kh_n * kw_n;

  // A operand (packed weights [K][M], m fastest): lanes walk m.
  const int ta = tid % MT, ka0 = tid / MT;
  const int ma = m0 + ta;
  const float* wp = p.wp + (size_t)g * p.K * p.M + ma;
  const bool ma_ok = ma < p.M;

  // B operand (patch gather): lanes walk n = (b, y, x).
  const int tb = tid % NT, kb0 = tid / NT;
  const int nb = n0 + tb;
  const bool nb_ok = nb < p.N;
  uint32_t bb, rem, yy, xx;
  fd_divmod(nb_ok ? nb : 0, p.div_hw, bb, rem);
  fd_divmod(rem, p.div_w, yy, xx);
  const size_t plane = (size_t)p.Hs * p.Ws;
  const float* src = p.src + ((size_t)bb * p.Cs_total + (size_t)g * p.Cs_g) * plane;
  const int by = TRANSPOSED ? (int)yy + p.pad : (int)yy * p.stride - p.pad;
  const int bx = TRANSPOSED ? (int)xx + p.pad : (int)xx * p.stride - p.pad;

  float ra[EA], rb[EB];
  auto load = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const int k = k0 + ka0 + e * STEP_A;
      ra[e] = (ma_ok && k < p.K) ? wp[(size_t)k * p.M] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int k = k0 + kb0 + e * STEP_B;
      int c, dy, dx;
      if (KH == 1 && KW == 1) {
        c = k;
        dy = 0;
        dx = 0;
      } else {
        c = k / khw;
        const int t = k - c * khw;
        dy = t / kw_n;
        dx = t - dy * kw_n;
      }
      int iy, ix;
      bool ok = nb_ok && k < p.K;
      if (!TRANSPOSED) {
        iy = by + dy;
        ix = bx + dx;
      } else {
        const int ty = by - dy, tx = bx - dx;
        if (p.stride == 1) {
          iy = ty;
          ix = tx;
        } else if (p.stride == 2) {
          ok = ok && !((ty | tx) & 1);
          iy = ty >> 1;
          ix = tx >> 1;
        } else {
          iy = ty / p.stride;
          ix = tx / p.stride;
          ok = ok && ty >= 0 && tx >= 0 && iy * p.stride == ty && ix * p.stride == tx;
        }
      }
      ok = ok && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
      rb[e] = ok ? src[(size_t)c * plane + (size_t)iy * p.Ws + ix] : 0.f;
    }
  };
  auto stage = [&](float* s) {
    float* sA = s;
    float* sB = s + KC * MT;
#pragma unroll
    for (int e = 0; e < EA; ++e) sA[(ka0 + e * STEP_A) * MT + ta] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) sB[(kb0 + e * STEP_B) * NT + tb] = rb[e];
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;

  const int nchunks = (p.K + KC - 1) / KC;
  load(0);
  stage(smem);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const float* cur = smem + (c & 1) * STAGE;
    if (c + 1 < nchunks) load((c + 1) * KC);
    mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1>(cur, cur + KC * MT, a_off, b_off, lane, acc);
    if (c + 1 < nchunks) stage(smem + ((c + 1) & 1) * STAGE);
    __syncthreads();
  }

  // Epilogue: lanes walk n (contiguous x within an image row) -> coalesced 128 B segments.
  const int li = lane & 31, hi = lane >> 5;
  const size_t dplane = (size_t)p.Hd * p.Wd;
#pragma unroll
  for (int j = 0; j < T::TN; ++j) {
    const int n = n0 + b_off + j * 32 + li;
    if (n >= p.N) continue;
    uint32_t ob, orem;
    fd_divmod(n, p.div_hw, ob, orem);
    float* dst = p.dst + ((size_t)ob * p.Cd_total + (size_t)g * p.M) * dplane + orem;
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + a_off + i * 32 + acc_row(r, hi);
        if (m < p.M) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[g * p.M + m];
          if (p.relu) v = fmaxf(v, 0.f);
          dst[(size_t)m * dplane] = v;
        }
      }
    }
  }
}

// =========================================================================================
// conv_wgrad: slab[s][g*M+m][j] = sum_{n in split s} dY[b, g*M+m, oy, ox] * X[b, g*Ci_g+ci, oy*s-p+kh, ox*s-p+kw]
//   with n=(b,oy,ox), j=(ci,kh,kw).  Both operands are gathered with lanes walking n (k-fast LDS).
// =========================================================================================
struct WgradParams {
  const float* dy;
  const float* x;
  float* slab;  // [S][G*M][J]
  int B, Hi, Wi, Ho, Wo, Ci_total, Co_total, Ci_g;
  int M, J, Ktot;  // per group: M=Co_g, J=Ci_g*kh*kw, Ktot=B*Ho*Wo
  int stride, pad, kh, kw;
  int splits, klen;  // klen: K range per split (multiple of KC)
  int tiles_m, tiles_j;
  FastDiv div_hw, div_w;  // Ho*Wo, Wo
};

template <class T, int KH, int KW>
__global__ __launch_bounds__(T::NTHREADS) void conv_wgrad_kernel(WgradParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = T::KC, NTH = T::NTHREADS;
  constexpr int STEP = NTH / KC, EA = MT / STEP, EB = NT / STEP;
  static_assert(NTH % KC == 0 && MT % STEP == 0 && NT % STEP == 0, "tile/thread mismatch");
  constexpr int LDK = KC + 1;
  constexpr int STAGE = (MT + NT) * LDK;
  extern __shared__ __attribute__((aligned(16))) float dsmem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tj = lid / p.tiles_m;
  const int m0 = tm * MT, j0 = tj * NT;
  const int kh_n = KH ? KH : p.kh, kw_n = KW ? KW : p.kw;
  const int khw = kh_n * kw_n;

  const int kl = tid % KC, t0 = tid / KC;
  const int kbeg = sp * p.klen;
  const int kend = min(kbeg + p.klen, p.Ktot);
  const size_t oplane = (size_t)p.Ho * p.Wo, iplane = (size_t)p.Hi * p.Wi;

  float ra[EA], rb[EB];
  auto load = [&](int k0) {
    const int n = k0 + kl;
    const bool n_ok = n < kend;
    uint32_t bb, rem, oy, ox;
    fd_divmod(n_ok ? n : 0, p.div_hw, bb, rem);
    fd_divmod(rem, p.div_w, oy, ox);
    const float* dy = p.dy + ((size_t)bb * p.Co_total + (size_t)g * p.M) * oplane + rem;
    const float* x = p.x + ((size_t)bb * p.Ci_total + (size_t)g * p.Ci_g) * iplane;
    const int by = (int)oy * p.stride - p.pad, bx = (int)ox * p.stride - p.pad;
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      const int m = m0 + t0 + e * STEP;
      ra[e] = (n_ok && m < p.M) ? dy[(size_t)m * oplane] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int j = j0 + t0 + e * STEP;
      int c, dyy, dxx;
      if (KH == 1 && KW == 1) {
        c = j;
        dyy = 0;
        dxx = 0;
      } else {
        c = j / khw;
        const int t = j - c * khw;
        dyy = t / kw_n;
        dxx = t - dyy * kw_n;
      }
      const int iy = by + dyy, ix = bx + dxx;
      const bool ok = n_ok && j < p.J && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      rb[e] = ok ? x[(size_t)c * iplane + (size_t)iy * p.Wi + ix] : 0.f;
    }
  };
  auto stage = [&](float* s) {
    float* sA = s;
    float* sB = s + MT * LDK;
#pragma unroll
    for (int e = 0; e < EA; ++e) sA[(t0 + e * STEP) * LDK + kl] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) sB[(t0 + e * STEP) * LDK + kl] = rb[e];
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;

  const int nchunks = (kend - kbeg + KC - 1) / KC;
  if (nchunks > 0) {
    load(kbeg);
    stage(dsmem);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const float* cur = dsmem + (c & 1) * STAGE;
      if (c + 1 < nchunks) load(kbeg + (c + 1) * KC);
      mma_chunk<T::TM, T::TN, KC, 1, LDK, 1, LDK>(cur, cur + MT * LDK, a_off, b_off, lane, acc);
      if (c + 1 < nchunks) stage(dsmem + ((c + 1) & 1) * STAGE);
      __syncthreads();
    }
  }

  const int li = lane & 31, hi = lane >> 5;
  const int G = gridDim.z / p.splits;
  float* slab = p.slab + ((size_t)sp * G + g) * (size_t)p.M * p.J;
#pragma unroll
  for (int jn = 0; jn < T::TN; ++jn) {
    const int j = j0 + b_off + jn * 32 + li;
    if (j >= p.J) continue;
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + a_off + i * 32 + acc_row(r, hi);
        if (m < p.M) slab[(size_t)m * p.J + j] = acc[i][jn][r];
      }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int splits) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slab[(size_t)k * n + i];
    out[i] = s;
  }
}

// =========================================================================================
// Weight packing (OIHW -> K-major) so the A-operand tile is a coalesced [K][M] copy.
// =========================================================================================
// fwd : out[g][k=(ci,t)][m=co]  = w[(g*Co_g+co)][ci][t]
// dgrad: out[g][k=(co,t)][m=ci] = w[(g*Co_g+co)][ci][t]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int G, int Co_g, int Ci_g,
                                   int khw, int transposed) {
  const long long total = (long long)G * Co_g * Ci_g * khw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    long long rest = i;
    int g, co, ci, t;
    if (!transposed) {  // i = ((g*Ci_g + ci)*khw + t)*Co_g + co
      co = rest % Co_g;
      rest /= Co_g;
      t = rest % khw;
      rest /= khw;
      ci = rest % Ci_g;
      g = rest / Ci_g;
    } else {  // i = ((g*Co_g + co)*khw + t)*Ci_g + ci
      ci = rest % Ci_g;
      rest /= Ci_g;
      t = rest % khw;
      rest /= khw;
      co = rest % Co_g;
      g = rest / Co_g;
    }
    out[i] = w[(((long long)(g * Co_g + co)) * Ci_g + ci) * khw + t];
  }
}

// =========================================================================================
// Strided (batched) GEMM: C[m*scm + n*scn] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+bias)(+C)
// =========================================================================================
struct GemmParams {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  int M, N, K;
  long long sam, sak, sbk, sbn, scm, scn, bsA, bsB, bsC;
  float alpha;
  int bias_mode;  // 0 none, 1 per-m, 2 per-n
  int relu, accumulate;
  int tiles_m;
};

template <class T, bool A_LANE_K, bool B_LANE_K>
__global__ __launch_bounds__(T::NTHREADS) void gemm_kernel(GemmParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = T::KC, NTH = T::NTHREADS;
  constexpr int LDK = KC + 1;
  constexpr int A_SZ = A_LANE_K ? MT * LDK : KC * MT;
  constexpr int B_SZ = B_LANE_K ? NT * LDK : KC * NT;
  constexpr int STAGE = A_SZ + B_SZ;
  constexpr int EA = KC * MT / NTH, EB = KC * NT / NTH;
  __shared__ float smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tn = lid / p.tiles_m;
  const int m0 = tm * MT, n0 = tn * NT;
  const float* A = p.A + (size_t)blockIdx.z * p.bsA;
  const float* B = p.B + (size_t)blockIdx.z * p.bsB;
  float* C = p.C + (size_t)blockIdx.z * p.bsC;

  float ra[EA], rb[EB];
  auto load = [&](int k0) {
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      int kk, t;
      if (A_LANE_K) {
        kk = tid % KC;
        t = tid / KC + e * (NTH / KC);
      } else {
        t = tid % MT;
        kk = tid / MT + e * (NTH / MT);
      }
      const int k = k0 + kk, m = m0 + t;
      ra[e] = (k < p.K && m < p.M) ? A[(size_t)m * p.sam + (size_t)k * p.sak] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      int kk, t;
      if (B_LANE_K) {
        kk = tid % KC;
        t = tid / KC + e * (NTH / KC);
      } else {
        t = tid % NT;
        kk = tid / NT + e * (NTH / NT);
      }
      const int k = k0 + kk, n = n0 + t;
      rb[e] = (k < p.K && n < p.N) ? B[(size_t)k * p.sbk + (size_t)n * p.sbn] : 0.f;
    }
  };
  auto stage = [&](float* s) {
    float* sA = s;
    float* sB = s + A_SZ;
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      if (A_LANE_K)
        sA[(tid / KC + e * (NTH / KC)) * LDK + tid % KC] = ra[e];
      else
        sA[(tid / MT + e * (NTH / MT)) * MT + tid % MT] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      if (B_LANE_K)
        sB[(tid / KC + e * (NTH / KC)) * LDK + tid % KC] = rb[e];
      else
        sB[(tid / NT + e * (NTH / NT)) * NT + tid % NT] = rb[e];
    }
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;
  constexpr int SKA = A_LANE_K ? 1 : MT, STA = A_LANE_K ? LDK : 1;
  constexpr int SKB = B_LANE_K ? 1 : NT, STB = B_LANE_K ? LDK : 1;

  const int nchunks = (p.K + KC - 1) / KC;
  load(0);
  stage(smem);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const float* cur = smem + (c & 1) * STAGE;
    if (c + 1 < nchunks) load((c + 1) * KC);
    mma_chunk<T::TM, T::TN, KC, SKA, STA, SKB, STB>(cur, cur + A_SZ, a_off, b_off, lane, acc);
    if (c + 1 < nchunks) stage(smem + ((c + 1) & 1) * STAGE);
    __syncthreads();
  }

  const int li = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < T::TN; ++j) {
    const int n = n0 + b_off + j * 32 + li;
    if (n >= p.N) continue;
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + a_off + i * 32 + acc_row(r, hi);
        if (m < p.M) {
          float v = p.alpha * acc[i][j][r];
          if (p.bias_mode == 1) v += p.bias[m];
          if (p.bias_mode == 2) v += p.bias[n];
          float* c = C + (size_t)m * p.scm + (size_t)n * p.scn;
          if (p.accumulate) v += *c;
          if (p.relu) v = fmaxf(v, 0.f);
          *c = v;
        }
      }
  }
}

// =========================================================================================
// Host side
// =========================================================================================
typedef TileCfg<2, 2, 2, 2, 16> Tile128;      // 128 x 128, 4 waves x (64 x 64)
typedef TileCfg<2, 2, 1, 2, 16> Tile64x128;   // 64 x 128
typedef TileCfg<2, 2, 1, 1, 16> Tile64;       // 64 x 64
typedef TileCfg<2, 2, 2, 2, 32> WTile128;     // wgrad: K chunk 32 so a lane group covers a 128 B line
typedef TileCfg<2, 2, 1, 1, 32> WTile64;

template <class T, int KH, int KW, bool TR>
static int launch_conv_gemm(ConvGemmParams& p, int G, hipStream_t st) {
  p.tiles_m = ge_cdiv(p.M, T::MT);
  p.tiles_n = ge_cdiv(p.N, T::NT);
  dim3 grid(p.tiles_m * p.tiles_n, 1, G);
  hipLaunchKernelGGL((conv_gemm_kernel<T, KH, KW, TR>), grid, dim3(T::NTHREADS), 0, st, p);
  GE_CHECK_LAUNCH("conv_gemm");
  return GE_OK;
}

template <int KH, int KW, bool TR>
static int dispatch_conv_tile(ConvGemmParams& p, int G, hipStream_t st) {
  const long long t128 = (long long)ge_cdiv(p.M, 128) * ge_cdiv(p.N, 128) * G;
  const long long t64x128 = (long long)ge_cdiv(p.M, 64) * ge_cdiv(p.N, 128) * G;
  if (p.M > 64 && t128 >= 192) return launch_conv_gemm<Tile128, KH, KW, TR>(p, G, st);
  if (t64x128 >= 192) return launch_conv_gemm<Tile64x128, KH, KW, TR>(p, G, st);
  return launch_conv_gemm<Tile64, KH, KW, TR>(p, G, st);
}

template <bool TR>
static int dispatch_conv(ConvGemmParams& p, int G, hipStream_t st) {
  if (p.kh == 1 && p.kw == 1) return dispatch_conv_tile<1, 1, TR>(p, G, st);
  if (p.kh == 3 && p.kw == 3) return dispatch_conv_tile<3, 3, TR>(p, G, st);
  if (p.kh == 7 && p.kw == 7) return dispatch_conv_tile<7, 7, TR>(p, G, st);
  return dispatch_conv_tile<0, 0, TR>(p, G, st);
}

extern "C" {

// Pack OIHW weights into the K-major operand layout.  transposed=0: forward, 1: data-gradient.
int ge_conv2d_pack_weight(const float* w, float* out, int Cout, int Cin_g, int kh, int kw, int groups, int transposed,
                          void* stream) {
  GE_REQUIRE(w && out && Cout > 0 && Cin_g > 0 && groups > 0 && Cout % groups == 0, "pack_weight: bad arguments");
  const long long total = (long long)Cout * Cin_g * kh * kw;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, out,
                     groups, Cout / groups, Cin_g, kh * kw, transposed);
  GE_CHECK_LAUNCH("pack_weight");
  return GE_OK;
}

// y[B,Cout,Ho,Wo] = conv2d(x[B,Cin,Hi,Wi], w) (+bias)(+relu); wp = ge_conv2d_pack_weight(..., transposed=0).
int ge_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Hi, int Wi,
                  int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu, void* stream) {
  GE_REQUIRE(x && wp && y, "conv2d_fwd: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_fwd: bad shape");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_fwd: B*Ho*Wo overflows int32");
  ConvGemmParams p;
  p.wp = wp;
  p.src = x;
  p.bias = bias;
  p.dst = y;
  p.B = B;
  p.Hs = Hi;
  p.Ws = Wi;
  p.Hd = Ho;
  p.Wd = Wo;
  p.Cs_total = Cin;
  p.Cd_total = Cout;
  p.Cs_g = Cin / groups;
  p.M = Cout / groups;
  p.N = B * Ho * Wo;
  p.K = p.Cs_g * kh * kw;
  p.stride = stride;
  p.pad = pad;
  p.kh = kh;
  p.kw = kw;
  p.relu = relu;
  p.div_hw = make_fastdiv(Ho * Wo);
  p.div_w = make_fastdiv(Wo);
  return dispatch_conv<false>(p, groups, (hipStream_t)stream);
}

// dx[B,Cin,Hi,Wi] = conv2d data gradient of dy[B,Cout,Ho,Wo]; wp = ge_conv2d_pack_weight(..., transposed=1).
int ge_conv2d_dgrad(const float* dy, const float* wp, float* dx, int B, int Cin, int Hi, int Wi, int Cout, int Ho,
                    int Wo, int kh, int kw, int stride, int pad, int groups, void* stream) {
  GE_REQUIRE(dy && wp && dx, "conv2d_dgrad: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_dgrad: bad shape");
  GE_REQUIRE((long long)B * Hi * Wi < (1ll << 31), "conv2d_dgrad: B*Hi*Wi overflows int32");
  ConvGemmParams p;
  p.wp = wp;
  p.src = dy;
  p.bias = nullptr;
  p.dst = dx;
  p.B = B;
  p.Hs = Ho;
  p.Ws = Wo;
  p.Hd = Hi;
  p.Wd = Wi;
  p.Cs_total = Cout;
  p.Cd_total = Cin;
  p.Cs_g = Cout / groups;
  p.M = Cin / groups;
  p.N = B * Hi * Wi;
  p.K = p.Cs_g * kh * kw;
  p.stride = stride;
  p.pad = pad;
  p.kh = kh;
  p.kw = kw;
  p.relu = 0;
  p.div_hw = make_fastdiv(Hi * Wi);
  p.div_w = make_fastdiv(Wi);
  return dispatch_conv<true>(p, groups, (hipStream_t)stream);
}

}  // extern "C"

template <class T, int KH, int KW>
static int launch_wgrad(WgradParams& p, int G, float* dw, hipStream_t st) {
  p.tiles_m = ge_cdiv(p.M, T::MT);
  p.tiles_j = ge_cdiv(p.J, T::NT);
  const size_t lds = 2 * (size_t)(T::MT + T::NT) * (T::KC + 1) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, KH, KW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  dim3 grid(p.tiles_m * p.tiles_j, 1, G * p.splits);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW>), grid, dim3(T::NTHREADS), lds, st, p);
  GE_CHECK_LAUNCH("conv_wgrad");
  return GE_OK;
}

static void wgrad_plan(int M, int J, int G, int Ktot, int& big, int& splits, int& klen) {
  const long long t128 = (long long)ge_cdiv(M, 128) * ge_cdiv(J, 128) * G;
  big = (M > 64 && J > 64 && t128 >= 8) ? 1 : 0;
  const long long tiles = big ? t128 : (long long)ge_cdiv(M, 64) * ge_cdiv(J, 64) * G;
  const int kc = 32;
  const int chunks = ge_cdiv(Ktot, kc);
  long long want = (768 + tiles - 1) / tiles;
  int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;  // >= 8 chunks (256 positions) per split
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  const int per = ge_cdiv(chunks, want);
  klen = per * kc;
  splits = ge_cdiv(Ktot, klen);
}

extern "C" {

// Workspace (floats) needed by ge_conv2d_wgrad.
long long ge_conv2d_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  int big, splits, klen;
  wgrad_plan(Cout / groups, (Cin / groups) * kh * kw, groups, B * Ho * Wo, big, splits, klen);
  return (long long)splits * Cout * (Cin / groups) * kh * kw;
}

// dw[Cout, Cin/groups, kh, kw] = conv2d weight gradient.  workspace: ge_conv2d_wgrad_workspace() floats.
int ge_conv2d_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi,
                    int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream) {
  GE_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_wgrad: bad shape");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_wgrad: B*Ho*Wo overflows int32");
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.dy = dy;
  p.x = x;
  p.slab = workspace;
  p.B = B;
  p.Hi = Hi;
  p.Wi = Wi;
  p.Ho = Ho;
  p.Wo = Wo;
  p.Ci_total = Cin;
  p.Co_total = Cout;
  p.Ci_g = Cin / groups;
  p.M = Cout / groups;
  p.J = p.Ci_g * kh * kw;
  p.Ktot = B * Ho * Wo;
  p.stride = stride;
  p.pad = pad;
  p.kh = kh;
  p.kw = kw;
  p.div_hw = make_fastdiv(Ho * Wo);
  p.div_w = make_fastdiv(Wo);
  int big;
  wgrad_plan(p.M, p.J, groups, p.Ktot, big, p.splits, p.klen);
  int rc;
  if (kh == 1 && kw == 1)
    rc = big ? launch_wgrad<WTile128, 1, 1>(p, groups, dw, st) : launch_wgrad<WTile64, 1, 1>(p, groups, dw, st);
  else if (kh == 3 && kw == 3)
    rc = big ? launch_wgrad<WTile128, 3, 3>(p, groups, dw, st) : launch_wgrad<WTile64, 3, 3>(p, groups, dw, st);
  else if (kh == 7 && kw == 7)
    rc = big ? launch_wgrad<WTile128, 7, 7>(p, groups, dw, st) : launch_wgrad<WTile64, 7, 7>(p, groups, dw, st);
  else
    rc = big ? launch_wgrad<WTile128, 0, 0>(p, groups, dw, st) : launch_wgrad<WTile64, 0, 0>(p, groups, dw, st);
  if (rc) return rc;
  const long long n = (long long)Cout * p.J;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, st, workspace, dw, n, p.splits);
  GE_CHECK_LAUNCH("slab_reduce");
  return GE_OK;
}

// Strided batched GEMM (see GemmParams).  Either stride of each operand must be 1.
int ge_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam,
            long long sak, long long sbk, long long sbn, long long scm, long long scn, int batch, long long bsA,
            long long bsB, long long bsC, float alpha, int bias_mode, int relu, int accumulate, void* stream) {
  GE_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "gemm: bad arguments");
  GE_REQUIRE(bias_mode == 0 || bias, "gemm: bias_mode set without bias");
  GemmParams p;
  p.A = A;
  p.B = B;
  p.bias = bias;
  p.C = C;
  p.M = M;
  p.N = N;
  p.K = K;
  p.sam = sam;
  p.sak = sak;
  p.sbk = sbk;
  p.sbn = sbn;
  p.scm = scm;
  p.scn = scn;
  p.bsA = bsA;
  p.bsB = bsB;
  p.bsC = bsC;
  p.alpha = alpha;
  p.bias_mode = bias_mode;
  p.relu = relu;
  p.accumulate = accumulate;
  p.tiles_m = ge_cdiv(M, Tile64::MT);
  dim3 grid(p.tiles_m * ge_cdiv(N, Tile64::NT), 1, batch);
  hipStream_t st = (hipStream_t)stream;
  const bool a_k = (sak == 1 && sam != 1), b_k = (sbk == 1 && sbn != 1);
  if (a_k && b_k)
    hipLaunchKernelGGL((gemm_kernel<Tile64, true, true>), grid, dim3(Tile64::NTHREADS), 0, st, p);
  else if (a_k && !b_k)
    hipLaunchKernelGGL((gemm_kernel<Tile64, true, false>), grid, dim3(Tile64::NTHREADS), 0, st, p);
  else if (!a_k && b_k)
    hipLaunchKernelGGL((gemm_kernel<Tile64, false, true>), grid, dim3(Tile64::NTHREADS), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_kernel<Tile64, false, false>), grid, dim3(Tile64::NTHREADS), 0, st, p);
  GE_CHECK_LAUNCH("gemm");
  return GE_OK;
}

}  // extern "C"
