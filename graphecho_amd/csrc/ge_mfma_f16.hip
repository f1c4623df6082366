// fp16-input MFMA conv kernels for gfx950 (v_mfma_f32_32x32x16_f16: 16x the fp32 matrix rate) -- the "fp16 MFMA conv
// path + fp32 accumulate" of BASELINE.json config 5.  Opt-in (functional.CONV_PRECISION = "f16"); the headline fp32
// path is untouched.  Tensors stay fp32 in HBM (activations, master weights, gradients); operands are rounded to fp16
// (RTNE) on their way into LDS, products accumulate in fp32 inside the MFMA, the epilogue (bias / skip addend / ReLU /
// fused BatchNorm moments) is the fp32 one.
//
// Implicit GEMM, K ordered (tap, channel) with 32-channel chunks, so a chunk has ONE tap: per chunk a thread computes
// one source pixel + validity, loads its 8/16 channels of that pixel (lanes walk n: coalesced), converts and writes
// 16-byte rows into the k-fast LDS tile sB[n][32 (+8 pad)]; the weight operand is pre-packed to fp16 as
// Wp[g][tap][m][c] and copied with 16-byte loads into sA[m][32 (+8)].  A lane's MFMA fragment (8 consecutive k of one
// row) is one ds_read_b128; the 80-byte row pitch makes those reads bank-conflict free.
// Requires Cin/groups and Cout/groups to be multiples of 32 (other layers -- the 3-channel stem, the nc-channel
// classifier -- stay on the fp32 kernels).
//
// (Rounds 2 - 5 carried a second operand mode here, "bf16x3": fp32-accurate convolution as six bf16 MFMA products of exactly
// three-way split operands, parked behind `make BX3=1` since round 3 at +0.2 % on the headline.  Removed in round 6; the measurements
// are in docs/HISTORY.md section 7b.)
#include "ge_mfma_lp.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#include <type_traits>
constexpr int LP_KC = 32;              // channels per chunk = two K=16 MFMA steps
typedef TileCfg<2, 2, 2, 2, 32> LpT128x;   // 128 x 128, 4 waves x (64 x 64)
constexpr int LP_PITCH = LP_KC + 8;    // halves per LDS row (80 B)

__device__ __forceinline__ u32x4 buf_load128(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ unsigned pack_half2(float a, float b) {
  half2v h = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, h);
}
// One K = 16 step of a wave's TM x TN tiles.  fa[i][0] / fb[j][0]: fragment (8 consecutive k): one fp16 MFMA per tile.
template <int NP, int TM, int TN>
__device__ __forceinline__ void lp_mma_step(const u32x4 (&fa)[TM][NP], const u32x4 (&fb)[TN][NP], f32x16 (&acc)[TM][TN]) {
  static_assert(NP == 1, "one operand plane (fp16)");
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, fa[i][0]),
                                                         __builtin_bit_cast(half8, fb[j][0]), acc[i][j], 0, 0, 0);
}

// fwd  : out[g][t][m=co][c=ci] = (half) w[g*Co_g+co][ci][t]
// dgrad: out[g][t][m=ci][c=co] = (half) w[g*Co_g+co][ci][t]
template <int NP>
__global__ void pack_weight_lp_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int G, int Co_g,
                                      int Ci_g, int khw, int transposed) {
  const unsigned total = (unsigned)G * Co_g * Ci_g * khw;
  const unsigned Mx = transposed ? Ci_g : Co_g, Cx = transposed ? Co_g : Ci_g;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    unsigned rest = i;
    const unsigned c = rest % Cx;
    rest /= Cx;
    const unsigned m = rest % Mx;
    rest /= Mx;
    const unsigned t = rest % khw;
    const unsigned g = rest / khw;
    const unsigned co = transposed ? c : m, ci = transposed ? m : c;
    const float v = w[((size_t)(g * Co_g + co) * Ci_g + ci) * khw + t];
    static_assert(NP == 1, "one operand plane (fp16)");
    out[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
  }
}

// NP: operand planes (1: fp16).  STAGES: 2 = double-buffered LDS (one barrier per chunk), 1 = one
// LDS stage + register prefetch (two barriers per chunk, half the LDS: two workgroups per CU -- one's staging phase
// runs under the other's MFMA phase).
template <class T, bool TRANSPOSED, int NP, int STAGES>
__global__ __launch_bounds__(256, 1) void conv_gemm_lp_kernel(ConvGemmParams p) {
  constexpr int MT = T::MT, NT = T::NT;
  constexpr int VA = MT * 4 / 256;        // 16-byte weight vectors per thread, chunk and plane
  constexpr int KQ = 256 / NT;            // threads sharing one column n
  constexpr int CPT = LP_KC / KQ;         // channels per thread and chunk (8 or 16)
  constexpr int PLANE = (MT + NT) * LP_PITCH;   // 16-bit elements per operand plane of a stage
  constexpr int STAGE = NP * PLANE;
  static_assert(T::NTHREADS == 256 && VA >= 1 && CPT % 8 == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) unsigned short lpsmem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tn = lid / p.tiles_m;
  const int m0 = tm * MT, n0 = tn * NT;
  const int taps = p.kh * p.kw;
  const int cblocks = p.Cs_g / LP_KC;

  // A operand: vector v = tid + e*256 -> row v/4, 8-element part v%4
  const rsrc_t wrs = make_rsrc(p.wp, p.wp_bytes);
  uint32_t a_off[VA];      // byte offset of this thread's vector for (plane 0, tap 0, channel block 0)
  bool a_ok[VA];
  int a_lds[VA];
#pragma unroll
  for (int e = 0; e < VA; ++e) {
    const int v = tid + e * 256, row = v >> 2, part = v & 3;
    a_ok[e] = m0 + row < p.M;
    a_off[e] = (uint32_t)((((size_t)g * taps) * p.M + m0 + row) * p.Cs_g + part * 8) * 2u;
    a_lds[e] = row * LP_PITCH + part * 8;
  }
  const uint32_t a_tap_stride = (uint32_t)p.M * p.Cs_g * 2u;   // bytes between taps
  const uint32_t a_plane_stride = p.wp_bytes / NP;             // bytes between operand planes

  // B operand: column n = n0 + tid % NT, channels kq*CPT .. +CPT of the chunk
  const int tb = tid % NT, kq = tid / NT;
  const int nb = n0 + tb;
  const bool nb_ok = nb < p.N;
  uint32_t bb, rem, yy, xx;
  fd_divmod(nb_ok ? nb : 0, p.div_hw, bb, rem);
  fd_divmod(rem, p.div_w, yy, xx);
  const uint32_t plane = (uint32_t)p.Hs * p.Ws;
  const rsrc_t srs = make_rsrc(p.src, p.src_bytes);
  const uint32_t b_base = (bb * p.Cs_total + (uint32_t)g * p.Cs_g + (uint32_t)kq * CPT) * plane;
  const int by = TRANSPOSED ? (int)yy + p.pad : (int)yy * p.stride - p.pad;
  const int bx = TRANSPOSED ? (int)xx + p.pad : (int)xx * p.stride - p.pad;
  const int b_lds = MT * LP_PITCH + tb * LP_PITCH + kq * CPT;

  u32x4 ra[VA][NP];
  float rb[CPT];
  auto load = [&](int chunk) {
    const int t = chunk / cblocks, cb = chunk - t * cblocks;
#pragma unroll
    for (int e = 0; e < VA; ++e) {
      uint32_t off = a_off[e] + (uint32_t)t * a_tap_stride + (uint32_t)cb * (LP_KC * 2u);
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int q = 0; q < NP; ++q)
        ra[e][q] = buf_load128(wrs, a_ok[e] ? off + (uint32_t)q * a_plane_stride : GE_OOB);
    }
    const int dy = t / p.kw, dx = t - dy * p.kw;
    int iy, ix;
    bool ok = nb_ok;
    if (!TRANSPOSED) {
      iy = by + dy;
      ix = bx + dx;
    } else {
      const int ty = by - dy, tx = bx - dx;
      if (p.stride == 1) {
        iy = ty;
        ix = tx;
      } else {
        iy = ty / p.stride;
        ix = tx / p.stride;
        ok = ok && ty >= 0 && tx >= 0 && iy * p.stride == ty && ix * p.stride == tx;
      }
    }
    ok = ok && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
    // one saturating-add walk over the thread's channels: the all-ones sentinel (tap outside the image) stays put
    uint32_t off = (b_base + (uint32_t)cb * LP_KC * plane + (uint32_t)(ok ? iy * p.Ws + ix : 0)) * 4u;
    asm volatile("" : "+v"(off));
    off = ok ? off : GE_OOB;
    const uint32_t cstep = plane * 4u;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      rb[j] = buf_load(srs, off);
      off = __builtin_elementwise_add_sat(off, cstep);
    }
  };
  auto stage = [&](unsigned short* s) {
#pragma unroll
    for (int e = 0; e < VA; ++e)
#pragma unroll
      for (int q = 0; q < NP; ++q) *(u32x4*)(s + q * PLANE + a_lds[e]) = ra[e][q];
#pragma unroll
    for (int q = 0; q < CPT / 8; ++q) {
      u32x4 v;
      v.x = pack_half2(rb[q * 8 + 0], rb[q * 8 + 1]);
      v.y = pack_half2(rb[q * 8 + 2], rb[q * 8 + 3]);
      v.z = pack_half2(rb[q * 8 + 4], rb[q * 8 + 5]);
      v.w = pack_half2(rb[q * 8 + 6], rb[q * 8 + 7]);
      *(u32x4*)(s + b_lds + q * 8) = v;
    }
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_offr = wm * T::TM * 32, b_offr = wn * T::TN * 32;
  const int li = lane & 31, hi = lane >> 5;

  auto mma = [&](const unsigned short* s) {
    const unsigned short* pa = s + (a_offr + li) * LP_PITCH + hi * 8;
    const unsigned short* pb = s + MT * LP_PITCH + (b_offr + li) * LP_PITCH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[T::TM][NP], fb[T::TN][NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) fa[i][q] = *(const u32x4*)(pa + q * PLANE + i * 32 * LP_PITCH + ks * 16);
#pragma unroll
        for (int j = 0; j < T::TN; ++j) fb[j][q] = *(const u32x4*)(pb + q * PLANE + j * 32 * LP_PITCH + ks * 16);
      }
      lp_mma_step<NP, T::TM, T::TN>(fa, fb, acc);
    }
  };

  const int nchunks = taps * cblocks;
  load(0);
  stage(lpsmem);
  __syncthreads();
  if constexpr (STAGES == 2) {
    for (int c = 0; c + 1 < nchunks; ++c) {
      if (!(p.dbg & 8)) load(c + 1);
      mma(lpsmem + (c & 1) * STAGE);
      if (!(p.dbg & 16)) {
        stage(lpsmem + ((c + 1) & 1) * STAGE);
        __syncthreads();
      }
    }
    mma(lpsmem + ((nchunks - 1) & 1) * STAGE);
  } else {
    for (int c = 0; c + 1 < nchunks; ++c) {
      if (!(p.dbg & 8)) load(c + 1);          // in flight under this chunk's MFMAs
      if (!(p.dbg & 32)) mma(lpsmem);
      __syncthreads();
      if (!(p.dbg & 16)) stage(lpsmem);
      __syncthreads();
    }
    mma(lpsmem);
  }
  conv_epilogue<T>(p, acc, g, m0, n0, a_offr, b_offr, lane, tn, wn);
}

// =========================================================================================
// Weight gradient with fp16 MFMA inputs: slab[s][g*M+m][j] = sum_{n in split s} dY[b, g*M+m, oy, ox] * X[b, ci, iy, ix],
// n = (b, oy, ox), j = (ci, kh, kw).  Same decomposition as the fp32 conv_wgrad_kernel (lanes walk n: 128-B
// coalesced gathers of both operands, split-K slabs + deterministic reduce); the k-fast LDS tiles hold halves, written
// as half2 after a quad_perm exchange between neighbouring lanes (k, k+1), and feed 32x32x16 MFMAs.
// =========================================================================================
struct LpWgradParams {
  const float* dy;
  const float* x;
  float* slab;  // [S][G*M][J]
  int B, Hi, Wi, Ho, Wo, Ci_total, Co_total, Ci_g;
  int M, J, Ktot;
  int stride, pad, kh, kw;
  int splits, klen;
  int tiles_m, tiles_j;
  uint32_t dy_bytes, x_bytes;
  FastDiv div_hw, div_w;
};

__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
}

template <class T, int NP>
__global__ __launch_bounds__(256, 1) void conv_wgrad_lp_kernel(LpWgradParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = LP_KC;
  constexpr int STEP = 256 / KC, EA = MT / STEP, EB = NT / STEP;   // 8 rows per pass
  constexpr int PLANE = (MT + NT) * LP_PITCH;
  static_assert(EA % 2 == 0 && EB % 2 == 0, "rows are written in pairs");
  extern __shared__ __attribute__((aligned(16))) unsigned short lpsmem[];
  unsigned short* sA = lpsmem;
  unsigned short* sB = lpsmem + MT * LP_PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z / p.splits, sp = blockIdx.z % p.splits;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tj = lid / p.tiles_m;
  const int m0 = tm * MT, j0 = tj * NT;
  const int khw = p.kh * p.kw;

  const int kl = tid % KC, t0 = tid / KC;
  const int kbeg = sp * p.klen;
  const int kend = min(kbeg + p.klen, p.Ktot);
  const uint32_t oplane = (uint32_t)p.Ho * p.Wo, iplane = (uint32_t)p.Hi * p.Wi;
  const rsrc_t drs = make_rsrc(p.dy, p.dy_bytes);
  const rsrc_t xrs = make_rsrc(p.x, p.x_bytes);

  int w_coff[EB], w_tap[EB];
  uint32_t w_jok = 0, w_mok = 0;
#pragma unroll
  for (int e = 0; e < EB; ++e) {
    const int j = j0 + t0 + e * STEP;
    const int c = j / khw, t = j - c * khw;
    const int dyy = t / p.kw, dxx = t - dyy * p.kw;
    w_coff[e] = c * (int)iplane + dyy * p.Wi + dxx;
    w_tap[e] = dyy | (dxx << 8);
    w_jok |= (j < p.J ? 1u : 0u) << e;
  }
#pragma unroll
  for (int e = 0; e < EA; ++e) w_mok |= (m0 + t0 + e * STEP < p.M ? 1u : 0u) << e;

  float ra[EA], rb[EB];
  auto load = [&](int k0) {
    const int n = k0 + kl;
    const bool n_ok = n < kend;
    uint32_t bb, rem, oy, ox;
    fd_divmod(n_ok ? n : 0, p.div_hw, bb, rem);
    fd_divmod(rem, p.div_w, oy, ox);
    const uint32_t dy_base = (bb * p.Co_total + (uint32_t)g * p.M + m0 + t0) * oplane + rem;
    const int by = (int)oy * p.stride - p.pad, bx = (int)ox * p.stride - p.pad;
    const int x_base = (int)((bb * p.Ci_total + (uint32_t)g * p.Ci_g) * iplane) + by * p.Wi + bx;
#pragma unroll
    for (int e = 0; e < EA; ++e)
      ra[e] = buf_load(drs, guard_off(dy_base + (uint32_t)(e * STEP) * oplane, n_ok && ((w_mok >> e) & 1u)));
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int iy = by + (w_tap[e] & 255), ix = bx + (w_tap[e] >> 8);
      const bool ok = n_ok && ((w_jok >> e) & 1u) && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      rb[e] = buf_load(xrs, guard_off((uint32_t)(x_base + w_coff[e]), ok));
    }
  };
  // Lanes (k, k+1) exchange values (quad_perm [1,0,3,2]); the even lane writes the pair of the even row of a row
  // pair, the odd lane that of the odd row: every LDS write is a full dword (one per operand plane).
  const bool odd = kl & 1;
  const int kw2 = kl & ~1;
  auto put = [&](unsigned short* base, int row, float lo, float hi2) {
    unsigned short* d = base + row * LP_PITCH + kw2;
    *(unsigned*)d = pack_half2(lo, hi2);
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < EA / 2; ++q) {
      const float o0 = dpp_xor1(ra[2 * q]), o1 = dpp_xor1(ra[2 * q + 1]);
      put(sA, t0 + (2 * q + (odd ? 1 : 0)) * STEP, odd ? o1 : ra[2 * q], odd ? ra[2 * q + 1] : o0);
    }
#pragma unroll
    for (int q = 0; q < EB / 2; ++q) {
      const float o0 = dpp_xor1(rb[2 * q]), o1 = dpp_xor1(rb[2 * q + 1]);
      put(sB, t0 + (2 * q + (odd ? 1 : 0)) * STEP, odd ? o1 : rb[2 * q], odd ? rb[2 * q + 1] : o0);
    }
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;
  const int li = lane & 31, hi = lane >> 5;
  auto mma = [&]() {
    const unsigned short* pa = sA + (a_off + li) * LP_PITCH + hi * 8;
    const unsigned short* pb = sB + (b_off + li) * LP_PITCH + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 fa[T::TM][NP], fb[T::TN][NP];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) fa[i][q] = *(const u32x4*)(pa + q * PLANE + i * 32 * LP_PITCH + ks * 16);
#pragma unroll
        for (int j = 0; j < T::TN; ++j) fb[j][q] = *(const u32x4*)(pb + q * PLANE + j * 32 * LP_PITCH + ks * 16);
      }
      lp_mma_step<NP, T::TM, T::TN>(fa, fb, acc);
    }
  };

  const int nchunks = (kend - kbeg + KC - 1) / KC;
  if (nchunks > 0) {
    load(kbeg);
    stage();
    __syncthreads();
    for (int c = 0; c + 1 < nchunks; ++c) {
      load(kbeg + (c + 1) * KC);   // in flight under this chunk's MFMAs (single LDS stage + register prefetch)
      mma();
      __syncthreads();
      stage();
      __syncthreads();
    }
    mma();
  }

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

__global__ void lp_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int splits,
                                      int accumulate) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s0 = accumulate ? out[i] : 0.f, s1 = 0.f;
    int k = 0;
    for (; k + 2 <= splits; k += 2) {
      s0 += slab[(size_t)k * n + i];
      s1 += slab[(size_t)(k + 1) * n + i];
    }
    for (; k < splits; ++k) s0 += slab[(size_t)k * n + i];
    out[i] = s0 + s1;
  }
}

// Tile + split plan: parallelism comes from K splits (every split >= 8 chunks = 256 positions), aiming at ~3
// workgroups per CU with a balanced last round.
static void lp_wgrad_plan(int M, int J, int G, int Ktot, int& big, int& splits, int& klen) {
  const long long t128 = (long long)ge_cdiv(M, 128) * ge_cdiv(J, 128) * G;
  big = (M > 64 && J > 64 && t128 >= 8) ? 1 : 0;
  const long long tiles = big ? t128 : (long long)ge_cdiv(M, 64) * ge_cdiv(J, 64) * G;
  const int chunks = ge_cdiv(Ktot, LP_KC);
  const int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;
  int best_klen = chunks * LP_KC, best_splits = 1;
  double best_score = -1.0;
  for (int s = 1; s <= max_splits; ++s) {
    const int kl = ge_cdiv(chunks, s) * LP_KC;
    const int sa = ge_cdiv(Ktot, kl);
    const long long blocks = tiles * sa;
    if (blocks > 1024 && s > 1) break;
    const double per_cu = (double)blocks / 256.0;
    const double rounds = (double)((blocks + 255) / 256);
    double score = per_cu / rounds;
    if (per_cu < 1.0) score *= per_cu;
    score *= (per_cu >= 2.5 ? 1.0 : 0.85 + 0.06 * per_cu);
    if (score > best_score + 1e-9) {
      best_score = score;
      best_klen = kl;
      best_splits = sa;
    }
  }
  klen = best_klen;
  splits = best_splits;
}

typedef TileCfg<2, 2, 2, 2, LP_KC> LpT128;   // 128 x 128, 4 waves x (64 x 64)
typedef TileCfg<2, 2, 1, 1, LP_KC> LpT64;    // 64 x 64,   4 waves x (32 x 32)

static bool lp_big_tile(long long M, long long N, int G) {
  constexpr int min_big = 192;
  return M > 64 && (long long)ge_cdiv(M, 128) * ge_cdiv(N, 128) * G >= min_big;
}

template <class T, bool TR, int NP, int STAGES>
static void launch_lp_variant(const ConvGemmParams& p, const dim3& grid, size_t lds, hipStream_t st) {
    if (lds > 48 * 1024) GE_MAX_LDS((int)lds, (const void*)conv_gemm_lp_kernel<T, TR, NP, STAGES>);
  hipLaunchKernelGGL((conv_gemm_lp_kernel<T, TR, NP, STAGES>), grid, dim3(256), lds, st, p);
}

template <bool TR, int NP>
static int launch_lp(ConvGemmParams& p, int G, hipStream_t st) {
  const bool big = lp_big_tile(p.M, p.N, G);
  constexpr int dbg = 0;      // (phase-ablation bits: tuning builds edit this line)
  p.dbg = dbg;
  const int MT = big ? 128 : 64;
  p.tiles_m = ge_cdiv(p.M, MT);
  p.tiles_n = ge_cdiv(p.N, MT);
  const dim3 grid(p.tiles_m * p.tiles_n, 1, G);
  static_assert(NP == 1, "one operand plane (fp16)");
  // two LDS stages (40 / 20 KB), one barrier per chunk
  const int stages = 2;
  const size_t lds = (size_t)stages * NP * (MT + MT) * LP_PITCH * sizeof(unsigned short);
  if (big)
    launch_lp_variant<LpT128, TR, NP, 2>(p, grid, lds, st);
  else
    launch_lp_variant<LpT64, TR, NP, 2>(p, grid, lds, st);
  ge_note_kernel("conv_gemm_lp_kernel<TileCfg<2, 2, %d, %d, 32>, %s, %d, %d>", big ? 2 : 1, big ? 2 : 1,
                 TR ? "true" : "false", NP, stages);
  GE_CHECK_LAUNCH("conv_gemm_lp");
  return GE_OK;
}

static int lp_supported(int Cin, int Cout, int groups) {
  return groups > 0 && Cin % groups == 0 && Cout % groups == 0 && (Cin / groups) % LP_KC == 0 &&
         (Cout / groups) % LP_KC == 0;
}

template <int NP>
static int lp_pack_weight(const float* w, void* out, int Cout, int Cin_g, int kh, int kw, int groups, int transposed,
                          void* stream) {
  GE_REQUIRE(w && out && Cout > 0 && Cin_g > 0 && groups > 0 && Cout % groups == 0, "lp_pack_weight: bad arguments");
  const long long total = (long long)Cout * Cin_g * kh * kw;
  GE_REQUIRE(total * NP < (1ll << 31), "lp_pack_weight: weight too large");
  hipLaunchKernelGGL(pack_weight_lp_kernel<NP>, dim3(ge_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned short*)out, groups, Cout / groups, Cin_g, kh * kw, transposed);
  GE_CHECK_LAUNCH("lp_pack_weight");
  return GE_OK;
}

static int lp_fwd_stat_parts(int B, int Cout, int Ho, int Wo, int groups) {
  const long long N = (long long)B * Ho * Wo;
  return lp_big_tile(Cout / groups, N, groups) ? ge_cdiv(N, 128) * 2 : ge_cdiv(N, 64) * 2;
}

template <int NP>
static int lp_fwd(const float* x, const void* wp, const float* bias, float* y, float* stats, int B, int Cin, int Hi,
                  int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu,
                  void* stream) {
  GE_REQUIRE(x && wp && y, "conv2d_lp_fwd: null pointer");
  GE_REQUIRE(lp_supported(Cin, Cout, groups), "conv2d_lp_fwd: channel counts must be multiples of 32");
  GE_REQUIRE(!(stats && relu), "conv2d_lp_fwd: fused statistics need relu == 0");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_lp_fwd: B*Ho*Wo overflows int32");
  ConvGemmParams p = {};
  p.wp = (const float*)wp;
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
  p.os = 1;
  p.div_hw = make_fastdiv(Ho * Wo);
  p.div_w = make_fastdiv(Wo);
  const long long xb = 4ll * B * Cin * Hi * Wi, wb = 2ll * NP * Cout * p.Cs_g * kh * kw;
  GE_REQUIRE(xb < 0xFFFFFFF0ll && wb < 0xFFFFFFF0ll, "conv2d_lp_fwd: tensors of 4 GiB or more are not supported");
  p.src_bytes = (uint32_t)xb;
  p.wp_bytes = (uint32_t)wb;
  p.stats = stats;
  p.stats_parts = stats ? lp_fwd_stat_parts(B, Cout, Ho, Wo, groups) : 0;
  return launch_lp<false, NP>(p, groups, (hipStream_t)stream);
}

template <int NP>
static int lp_dgrad(const float* dy, const void* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi,
                    int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream) {
  GE_REQUIRE(dy && wp && dx, "conv2d_lp_dgrad: null pointer");
  GE_REQUIRE(lp_supported(Cin, Cout, groups), "conv2d_lp_dgrad: channel counts must be multiples of 32");
  GE_REQUIRE((long long)B * Hi * Wi < (1ll << 31), "conv2d_lp_dgrad: B*Hi*Wi overflows int32");
  ConvGemmParams p = {};
  p.wp = (const float*)wp;
  p.src = dy;
  p.dst = dx;
  p.addend = addend;
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
  p.os = 1;
  p.div_hw = make_fastdiv(Hi * Wi);
  p.div_w = make_fastdiv(Wi);
  const long long yb = 4ll * B * Cout * Ho * Wo, wb = 2ll * NP * Cout * (Cin / groups) * kh * kw;
  GE_REQUIRE(yb < 0xFFFFFFF0ll && wb < 0xFFFFFFF0ll, "conv2d_lp_dgrad: tensors of 4 GiB or more are not supported");
  p.src_bytes = (uint32_t)yb;
  p.wp_bytes = (uint32_t)wb;
  return launch_lp<true, NP>(p, groups, (hipStream_t)stream);
}

static long long lp_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  int big, splits, klen;
  lp_wgrad_plan(Cout / groups, (Cin / groups) * kh * kw, groups, B * Ho * Wo, big, splits, klen);
  return (long long)splits * Cout * (Cin / groups) * kh * kw;
}

template <int NP>
static int lp_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi,
                    int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate,
                    void* stream) {
  GE_REQUIRE(x && dy && dw && workspace, "conv2d_lp_wgrad: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0 &&
                 kh < 256 && kw < 256,
             "conv2d_lp_wgrad: bad shape");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_lp_wgrad: B*Ho*Wo overflows int32");
  hipStream_t st = (hipStream_t)stream;
  LpWgradParams p;
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
  const long long xb = 4ll * B * Cin * Hi * Wi, yb = 4ll * B * Cout * Ho * Wo;
  GE_REQUIRE(xb < 0xFFFFFFF0ll && yb < 0xFFFFFFF0ll, "conv2d_lp_wgrad: tensors of 4 GiB or more are not supported");
  p.x_bytes = (uint32_t)xb;
  p.dy_bytes = (uint32_t)yb;
  int big;
  lp_wgrad_plan(p.M, p.J, groups, p.Ktot, big, p.splits, p.klen);
  const int MT = big ? 128 : 64;
  p.tiles_m = ge_cdiv(p.M, MT);
  p.tiles_j = ge_cdiv(p.J, MT);
  const dim3 grid(p.tiles_m * p.tiles_j, 1, groups * p.splits);
  const size_t lds = (size_t)NP * 2 * MT * LP_PITCH * sizeof(unsigned short);
  if (big) {
        if (lds > 48 * 1024) GE_MAX_LDS((int)lds, (const void*)conv_wgrad_lp_kernel<LpT128, NP>);
    hipLaunchKernelGGL((conv_wgrad_lp_kernel<LpT128, NP>), grid, dim3(256), lds, st, p);
  } else {
    hipLaunchKernelGGL((conv_wgrad_lp_kernel<LpT64, NP>), grid, dim3(256), lds, st, p);
  }
  ge_note_kernel("conv_wgrad_lp_kernel<TileCfg<2, 2, %d, %d, 32>, %d>", big ? 2 : 1, big ? 2 : 1, NP);
  GE_CHECK_LAUNCH("conv_wgrad_lp");
  const long long n = (long long)Cout * p.J;
  ge_record_split_event(st);
  hipLaunchKernelGGL(lp_slab_reduce_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, st, workspace, dw, n, p.splits,
                     accumulate);
  GE_CHECK_LAUNCH("lp_slab_reduce");
  return GE_OK;
}

extern "C" {

// 1 when the 16-bit-operand kernels cover this layer (both channel counts per group multiples of 32)
int ge_conv2d_f16_supported(int Cin, int Cout, int groups) { return lp_supported(Cin, Cout, groups); }

// out: Cout*Cin_g*kh*kw halves (2 bytes each).  transposed=0: forward operand, 1: data-gradient operand.
int ge_conv2d_f16_pack_weight(const float* w, void* out, int Cout, int Cin_g, int kh, int kw, int groups, int transposed,
                              void* stream) {
  return lp_pack_weight<1>(w, out, Cout, Cin_g, kh, kw, groups, transposed, stream);
}

int ge_conv2d_f16_fwd_stat_parts(int B, int Cout, int Ho, int Wo, int groups) {
  return lp_fwd_stat_parts(B, Cout, Ho, Wo, groups);
}

// y = conv2d(x, w) (+bias)(+relu) with fp16 MFMA inputs / fp32 accumulation; wp from ge_conv2d_f16_pack_weight(.., 0).
// stats (nullable): [Cout][ge_conv2d_f16_fwd_stat_parts()][3] fused BatchNorm moments of y (requires relu == 0).
int ge_conv2d_f16_fwd(const float* x, const void* wp, const float* bias, float* y, float* stats, int B, int Cin, int Hi,
                      int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu,
                      void* stream) {
  return lp_fwd<1>(x, wp, bias, y, stats, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, relu, stream);
}

// dx = conv2d data-gradient (+addend); wp from ge_conv2d_*_pack_weight(.., 1)
int ge_conv2d_f16_dgrad(const float* dy, const void* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi,
                        int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream) {
  return lp_dgrad<1>(dy, wp, addend, dx, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, stream);
}

// Workspace (floats) of ge_conv2d_f16_wgrad
long long ge_conv2d_f16_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  return lp_wgrad_workspace(B, Cin, Cout, Ho, Wo, kh, kw, groups);
}

// dw[Cout, Cin/groups, kh, kw] (+)= weight gradient with 16-bit MFMA inputs, fp32 accumulation (any channel counts)
int ge_conv2d_f16_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi,
                        int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate,
                        void* stream) {
  return lp_wgrad<1>(x, dy, dw, workspace, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, accumulate, stream);
}

}  // extern "C"
