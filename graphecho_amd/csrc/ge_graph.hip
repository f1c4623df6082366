// Grapher path (models/vig.py:209-381, 88-105): dense k-NN graph build and max-relative edge aggregation.
//
// k-NN arithmetic is *defined* (pinned order, so a scalar C restatement on the test side reproduces it bit for bit):
//   nrm2[p]  = fmaf-chain over c ascending of x[c][p]^2                 (one thread per point)
//   xn[c][p] = x[c][p] / max(sqrt(nrm2[p]), 1e-12)                      (F.normalize, vig.py:372-378)
//   sq[p]    = fmaf-chain over c ascending of xn[c][p]^2
//   inner    = fmaf-chain over c ascending of xn[c][n]*yn[c][m] from 0  (v_mfma_f32_32x32x2_f32 is exactly this)
//   dist     = (sqx[n] + (-2*inner)) + sqy[m]  (+ relative_pos[n][m])   (vig.py:271-274, 298, 326)
//   top-k    = K smallest dist, ties -> lowest index, sorted ascending   (torch.topk(-dist), vig.py:299-327)
#include "ge_common.h"
#include <algorithm>
#include <limits.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Buffer loads with 32-bit byte offsets (as in ge_mfma.hip): the descriptor's range check returns 0 for the all-ones
// offset, so tile edges need no branches, and an address is one VGPR instead of a 64-bit pair.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define GE_OOB 0xFFFFFFFFu
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// 16 bytes per lane straight into LDS (as in ge_mfma.hip): lane l lands at lds_addr + 16 l; an out-of-range offset writes zeros
typedef unsigned int ge_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma16(ge_u32x4 rs, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_dma_wait() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
__device__ __forceinline__ ge_u32x4 make_rsrc_words(const void* p, uint32_t bytes) {
  const unsigned long long ad = (unsigned long long)p;
  ge_u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)ad);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(ad >> 32) & 0xFFFFu);
  rs.z = __builtin_amdgcn_readfirstlane(bytes);
  rs.w = 0x00020000u;
  return rs;
}

constexpr int KNN_PREP_U = 32;
// xn, sq from x [B][C][P]; normalize=0 keeps x and only computes sq.
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                       float* __restrict__ sq, int C, int P, int normalize) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= P) return;
  const float* xp = x + (size_t)b * C * P + p;
  float* op = xn + (size_t)b * C * P + p;
  // loads are issued KNN_PREP_U (32) at a time (round 4: 8 -- on TGCN's 64-node graphs the kernel is a chain of memory
  // round trips, 23 us for 64 of them); the fmaf chains stay strictly ascending in c (pinned arithmetic order)
  float denom = 1.f;
  if (normalize) {
    float s = 0.f;
    int c = 0;
    for (; c + KNN_PREP_U <= C; c += KNN_PREP_U) {
      float v[KNN_PREP_U];
#pragma unroll
      for (int u = 0; u < KNN_PREP_U; ++u) v[u] = xp[(size_t)(c + u) * P];
#pragma unroll
      for (int u = 0; u < KNN_PREP_U; ++u) s = fmaf(v[u], v[u], s);
    }
    for (; c < C; ++c) {
      const float v = xp[(size_t)c * P];
      s = fmaf(v, v, s);
    }
    denom = fmaxf(sqrtf(s), 1e-12f);
  }
  float q = 0.f;
  int c = 0;
  for (; c + KNN_PREP_U <= C; c += KNN_PREP_U) {
    float v[KNN_PREP_U];
#pragma unroll
    for (int u = 0; u < KNN_PREP_U; ++u) v[u] = xp[(size_t)(c + u) * P];
#pragma unroll
    for (int u = 0; u < KNN_PREP_U; ++u) {
      if (normalize) v[u] = v[u] / denom;
      op[(size_t)(c + u) * P] = v[u];
      q = fmaf(v[u], v[u], q);
    }
  }
  for (; c < C; ++c) {
    float v = xp[(size_t)c * P];
    if (normalize) v = v / denom;
    op[(size_t)c * P] = v;
    q = fmaf(v, v, q);
  }
  sq[(size_t)b * P + p] = q;
}

// ---- wave-level arg-min on 64-bit keys (order-preserving float bits << 32 | index) -----------------------------
// Lexicographic (distance, index) order becomes plain unsigned order, so "nearest, lowest index on ties" is one
// 64-bit min.  The reduction runs on DPP lane permutes (quad_perm, row_half_mirror, row_mirror) inside rows of 16
// and finishes with four readlanes: no LDS round trips (a ds_bpermute butterfly costs ~1 us per extraction).
typedef unsigned long long u64;
#define KNN_KEY_INF 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ u64 knn_key(float d, int idx) {
  unsigned u = __float_as_uint(d);
  u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;   // monotone map float -> uint
  return ((u64)u << 32) | (unsigned)idx;
}
template <int CTRL>
__device__ __forceinline__ u64 dpp_min64(u64 k) {
  const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
  const unsigned plo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
  const unsigned phi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
  const u64 p = ((u64)phi << 32) | plo;
  return p < k ? p : k;
}
// min over each 16-lane DPP row; every lane of the row receives its row's minimum
__device__ __forceinline__ u64 row16_min64(u64 k) {
  k = dpp_min64<0xB1>(k);
  k = dpp_min64<0x4E>(k);
  k = dpp_min64<0x141>(k);
  k = dpp_min64<0x140>(k);
  return k;
}
__device__ __forceinline__ u64 wave_min64(u64 k) {
  k = dpp_min64<0xB1>(k);    // quad_perm [1,0,3,2]
  k = dpp_min64<0x4E>(k);    // quad_perm [2,3,0,1]
  k = dpp_min64<0x141>(k);   // row_half_mirror: quad <-> neighbouring quad
  k = dpp_min64<0x140>(k);   // row_mirror: half-row <-> half-row; every lane of a 16-lane row now holds the row min
  u64 best = KNN_KEY_INF;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, r * 16);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), r * 16);
    const u64 v = ((u64)hi << 32) | lo;
    best = v < best ? v : best;
  }
  return best;   // wave-uniform
}

// Normalisation of 64 points (rows n0 .. n0 + 63 of batch item b) of x [B][C][N] by one workgroup of 256 threads: xn rows written to
// global, squared norms of the normalised rows to s_sq[64] (LDS); `lds`: >= KNN_PRO_CH * 64 floats of scratch, s_den: 64 more.
// Arithmetic = knn_prep_kernel's (the pinned order at the top of this file).  Ends with the block's LDS reads done (barrier).
constexpr int KNN_ROWS = 64, KNN_COLS = 128, KNN_KC = 16, KNN_DP = KNN_COLS + 1;
constexpr int KNN_PRO_CH = 128;      // channels per slab: [128][64] floats = 32 KB
__device__ __forceinline__ void knn_normalise_rows(const float* __restrict__ xraw, float* __restrict__ xn_out, int b, int C, int N,
                                                   int n0, int normalize, float* lds, float* s_sq, float* s_den) {
  const int tid = threadIdx.x;
  const uint32_t c_stepA = (uint32_t)N * 4u;
  {
    // ---- prologue: this workgroup's 64 query rows of the RAW tensor -> normalised rows in xn (global, read back by the operand
    // loader below: L2-hot) + their squared norms in LDS.  128 channels at a time through LDS: every thread loads 32 values
    // (coalesced over the rows), one thread per row runs the pinned fmaf chain over the slab.  C <= 256 (every Grapher): the raw
    // values stay in registers between the two chains -- x is read from memory exactly once.
    constexpr int PQ = KNN_PRO_CH / 4;
    float* t = lds;      // [KNN_PRO_CH][64]
    const rsrc_t rawrs = make_rsrc(xraw + (size_t)b * C * N, (uint32_t)C * (uint32_t)N * 4u);
    float* xw = xn_out + (size_t)b * C * N;
    const int pr = tid & 63, pk = tid >> 6;      // row; channels pk + 4 q of a slab
    const bool p_ok = n0 + pr < N;
    const uint32_t p_off0 = p_ok ? (uint32_t)(pk * N + n0 + pr) * 4u : GE_OOB;
    const uint32_t p_step = (uint32_t)(4 * N) * 4u;
    // channels past C read 0 through the descriptor's range check and add nothing to the chains
    auto gload = [&](int c0, float* v) {
      uint32_t o = __builtin_elementwise_add_sat(p_off0, (uint32_t)c0 * c_stepA);
#pragma unroll
      for (int q = 0; q < PQ; ++q) {
        v[q] = buf_load(rawrs, o);
        o = __builtin_elementwise_add_sat(o, p_step);
      }
    };
    auto gstore = [&](int c0, const float* v) {
      if (!p_ok) return;
#pragma unroll
      for (int q = 0; q < PQ; ++q) {
        const int c = c0 + pk + 4 * q;
        if (c < C) xw[(size_t)c * N + n0 + pr] = v[q];
      }
    };
    auto chain = [&](int c0, const float* v, float acc) {      // slab -> LDS, the row threads extend their chain
#pragma unroll
      for (int q = 0; q < PQ; ++q) t[(pk + 4 * q) * KNN_ROWS + pr] = v[q];
      __syncthreads();
      if (tid < KNN_ROWS) {
        const int kn = min(KNN_PRO_CH, C - c0);
        int k = 0;
        for (; k + 16 <= kn; k += 16) {
          float w[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) w[u] = t[(k + u) * KNN_ROWS + tid];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = fmaf(w[u], w[u], acc);
        }
        for (; k < kn; ++k) {
          const float w = t[k * KNN_ROWS + tid];
          acc = fmaf(w, w, acc);
        }
      }
      __syncthreads();
      return acc;
    };
    auto publish_den = [&](float acc) {
      if (tid < KNN_ROWS) s_den[tid] = fmaxf(sqrtf(acc), 1e-12f);
      __syncthreads();
    };
    if (C <= 2 * KNN_PRO_CH) {
      float v0[PQ], v1[PQ];
      gload(0, v0);
      gload(KNN_PRO_CH, v1);      // C <= 128: all zeros (out of the descriptor's range), no traffic
      if (normalize) {
        float acc = chain(0, v0, 0.f);
        if (C > KNN_PRO_CH) acc = chain(KNN_PRO_CH, v1, acc);
        publish_den(acc);
        const float dr = s_den[pr];
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
          v0[q] = v0[q] / dr;
          v1[q] = v1[q] / dr;
        }
      }
      gstore(0, v0);
      if (C > KNN_PRO_CH) gstore(KNN_PRO_CH, v1);
      float acc = chain(0, v0, 0.f);
      if (C > KNN_PRO_CH) acc = chain(KNN_PRO_CH, v1, acc);
      if (tid < KNN_ROWS) s_sq[tid] = acc;
    } else {
      float v[PQ];
      if (normalize) {
        float acc = 0.f;
        for (int c0 = 0; c0 < C; c0 += KNN_PRO_CH) {
          gload(c0, v);
          acc = chain(c0, v, acc);
        }
        publish_den(acc);
      }
      const float dr = normalize ? s_den[pr] : 1.f;
      float acc = 0.f;
      for (int c0 = 0; c0 < C; c0 += KNN_PRO_CH) {
        gload(c0, v);
        if (normalize) {
#pragma unroll
          for (int q = 0; q < PQ; ++q) v[q] = v[q] / dr;
        }
        gstore(c0, v);
        acc = chain(c0, v, acc);
      }
      if (tid < KNN_ROWS) s_sq[tid] = acc;
    }
  }
  __syncthreads();
}

// knn_prep_kernel's result from 64-point tiles (round 6): the candidate sets of the Graphers and TGCN's node sets are a few
// thousand points -- one thread per point is 32 workgroups on 256 CUs, each a chain of 16 dependent memory round trips (17 us for
// B32 x M256); here a workgroup owns 64 points, loads their channel columns with all 256 threads and only the chains are serial.
__global__ __launch_bounds__(256) void knn_prep_tile_kernel(const float* __restrict__ x, float* __restrict__ xn, float* __restrict__ sq,
                                                            int C, int P, int normalize) {
  __shared__ float buf[KNN_PRO_CH * KNN_ROWS + 2 * KNN_ROWS];
  const int b = blockIdx.y, n0 = blockIdx.x * KNN_ROWS;
  float* s_sq = buf + KNN_PRO_CH * KNN_ROWS;
  knn_normalise_rows(x, xn, b, C, P, n0, normalize, buf, s_sq, s_sq + KNN_ROWS);
  if (threadIdx.x < KNN_ROWS && n0 + threadIdx.x < P) sq[(size_t)b * P + n0 + threadIdx.x] = s_sq[threadIdx.x];
}

// One workgroup: 64 query rows x all M candidates, 128 candidates per pass.  out: int64 [2][B][N][Kout].
// Distance phase = a small GEMM pipeline: both operands go global -> registers -> double-buffered LDS chunks of 16
// channels (the next chunk's loads fly under this chunk's MFMAs); wave (wm, wn) owns rows 32*wm.. and columns
// 64*wn.. of the 64 x 128 tile (one A fragment feeds two 32x32x2 MFMAs).  The accumulation over c stays strictly
// ascending, so the distances are bit-for-bit those of the k-ordered fmaf chain the C oracle evaluates.
// Selection: G16 (K <= 16): a wave selects for FOUR query rows at once, one per 16-lane DPP row (each extraction
// round serves four rows, no cross-row step); otherwise one row per wave (K <= 64).
constexpr int KNN_STAGE = (KNN_ROWS + KNN_COLS) * KNN_KC;   // floats per LDS operand stage

#ifndef GE_KNN_WPS
#define GE_KNN_WPS 3
#endif
#ifndef GE_KNN_DBG
#define GE_KNN_DBG 0         // tuning builds only (wrong results): 1 = no selection, 2 = no distance GEMM, 4 = no operand loads
#endif
#ifndef GE_KNN_SORTED
#define GE_KNN_SORTED 1      // 0: the round-1..3 selection (minimum of eight + eight-slot knock-out per round)
#endif
// FUSE (round 6): the query side is normalised HERE -- `xraw` is the raw tensor, `xn` the buffer this workgroup writes its own 64
// normalised rows to (and reads back, L2-hot, as the A operand), sqx is unused.  Prologue: the rows' channel columns go through
// LDS 128 channels at a time (coalesced), one thread per row runs the two pinned fmaf chains (nrm2 over x, then sq over x / denom,
// both ascending in c; IEEE division per element = the bits knn_prep_kernel produces), sq stays in LDS.  The separate pass over x
// (read twice + written once: 0.069 of 0.332 ms at p2) is gone, x is read from memory once; candidates (M << N) keep
// knn_prep_kernel.  normalize = 0: the rows are copied, sq from the raw values.  (First form, measured and dropped: raw operand
// stages + a division of every A fragment on the way to the matrix pipe -- 4x redundant divisions, 0.314 vs 0.326 ms.)
template <bool G16, bool FUSE>
__global__ __launch_bounds__(256, G16 ? GE_KNN_WPS : 2) void knn_topk_kernel(const float* __restrict__ xn, const float* __restrict__ sqx,
                                                       const float* __restrict__ yn, const float* __restrict__ sqy,
                                                       const float* __restrict__ relpos, long long* __restrict__ out,
                                                       int B, int C, int N, int M, int K, int dil, int knn_dma_enabled, int normalize,
                                                       const float* __restrict__ xraw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // The distance tile [64][129] and the two operand stages ([16][64] query chunk + [16][128] candidate chunk each)
  // share the same LDS: the tile is written after the last chunk's MFMAs (barrier) and read before the next pass
  // stages operands again (barrier) -- 33 KB per workgroup instead of 58, twice the workgroups per CU.
  float* sD = lds;
  float* sOp = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int b = blockIdx.y, n0 = blockIdx.x * KNN_ROWS;
  // per-batch-item views through buffer descriptors (the host checks that each is < 4 GiB)
  const rsrc_t xrs = make_rsrc(xn + (size_t)b * C * N, (uint32_t)C * (uint32_t)N * 4u);
  const rsrc_t yrs = make_rsrc(yn + (size_t)b * C * M, (uint32_t)C * (uint32_t)M * 4u);
  const rsrc_t sxrs = make_rsrc(sqx + (size_t)b * N, (uint32_t)N * 4u);
  const rsrc_t syrs = make_rsrc(sqy + (size_t)b * M, (uint32_t)M * 4u);
  const rsrc_t rprs = make_rsrc(relpos, relpos ? (uint32_t)N * (uint32_t)M * 4u : 0u);

  // DMA loader (N % 4 == 0 and M % 4 == 0, GE_KNN_DMA != 0): per-lane byte offset of its 16 bytes inside a channel row
  const bool dma = knn_dma_enabled && (N & 3) == 0 && (M & 3) == 0;
  const ge_u32x4 xrw = make_rsrc_words(xn + (size_t)b * C * N, (uint32_t)C * (uint32_t)N * 4u);
  const ge_u32x4 yrw = make_rsrc_words(yn + (size_t)b * C * M, (uint32_t)C * (uint32_t)M * 4u);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)lds;
  const uint32_t a_dma = n0 + (lane & 15) * 4 < N ? (uint32_t)(n0 + (lane & 15) * 4) * 4u : GE_OOB;
  // loader roles: query chunk 16 x 64 -> 4 values per thread, candidate chunk 16 x 128 -> 8 values per thread
  const int ar = tid & 63, ak = tid >> 6;            // row, k = ak + 4e
  const int bc = tid & 127, bk = tid >> 7;           // column, k = bk + 2e
  const bool a_ok = n0 + ar < N;
  float ra[4], rb[8];
  // channels past C fall outside the descriptors (offset >= C * N * 4): they read 0 without a test of their own
  const uint32_t a_off0 = a_ok ? (uint32_t)(ak * N + n0 + ar) * 4u : GE_OOB;
  const uint32_t a_step = (uint32_t)(4 * N) * 4u, b_step = (uint32_t)(2 * M) * 4u, c_stepA = (uint32_t)N * 4u, c_stepB = (uint32_t)M * 4u;
  auto load = [&](int c0, int m0) {
    uint32_t oa = __builtin_elementwise_add_sat(a_off0, (uint32_t)c0 * c_stepA);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[e] = buf_load(xrs, oa);
      oa = __builtin_elementwise_add_sat(oa, a_step);
    }
    uint32_t ob = m0 + bc < M ? (uint32_t)((c0 + bk) * M + m0 + bc) * 4u : GE_OOB;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      rb[e] = buf_load(yrs, ob);
      ob = __builtin_elementwise_add_sat(ob, b_step);
    }
  };
  auto stage = [&](float* s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[(ak + 4 * e) * KNN_ROWS + ar] = ra[e];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[KNN_KC * KNN_ROWS + (bk + 2 * e) * KNN_COLS + bc] = rb[e];
  };

  float* s_sq = lds + KNN_ROWS * KNN_DP;       // [64] squared norms of the normalised query rows (FUSE)
  float* s_den = s_sq + KNN_ROWS;              // [64] denominators (prologue only)
  if (FUSE) {
    knn_normalise_rows(xraw, const_cast<float*>(xn), b, C, N, n0, normalize, lds, s_sq, s_den);
    // The normalised rows are read back by this workgroup only, through the L1 of the CU that wrote them (write-through, and
    // coherent for the waves of one CU): every wave's stores acknowledged (vmcnt 0), then the barrier.  An agent-scope fence here
    // (__threadfence) is a write-back of the XCD's whole L2 per workgroup on this part: measured 0.25 -> 0.54 ms at p2.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // running top-K lists as 64-bit keys.  G16: best[pass] = list of row wave*16 + pass*4 + (lane>>4), entry t in lane
  // t of that 16-lane group.  !G16: best[r] = list of row wave*16 + r, entry t in lane t.
  u64 best[G16 ? 4 : 16];
#pragma unroll
  for (int r = 0; r < (G16 ? 4 : 16); ++r) best[r] = KNN_KEY_INF;
  const int l16 = lane & 15, grp = lane >> 4;
  const int nchunks = (C + KNN_KC - 1) / KNN_KC;

  for (int m0 = 0; m0 < M; m0 += KNN_COLS) {
    // ---- distance tile
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    auto mma_chunk16 = [&](const float* cur) {
      const float* pa = cur + hi * KNN_ROWS + 32 * wm + li;
      const float* pb = cur + KNN_KC * KNN_ROWS + hi * KNN_COLS + 64 * wn + li;
      // all 24 fragment reads of the chunk are issued before the first MFMA (round 4): with read - wait - two MFMAs per
      // k-pair the matrix pipe idled through an LDS round trip every 128 cycles (distance phase 193 us against an MFMA
      // roof of 109 us at p2; tools: -DGE_KNN_DBG).  The MFMA order -- hence every bit of the distances -- is unchanged.
      float fa[KNN_KC / 2], fb0[KNN_KC / 2], fb1[KNN_KC / 2];
#pragma unroll
      for (int j = 0; j < KNN_KC / 2; ++j) {
        fa[j] = pa[2 * j * KNN_ROWS];
        fb0[j] = pb[2 * j * KNN_COLS];
        fb1[j] = pb[2 * j * KNN_COLS + 32];
      }

#pragma unroll
      for (int j = 0; j < KNN_KC / 2; ++j) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb1[j], acc1, 0, 0, 0);
      }
    };
    if (dma) {
      // Operand chunks global -> LDS by DMA (buffer_load_dwordx4 ... lds; N and M multiples of 4): the [k][row] stage layout IS
      // the lane-linear image of four (query) / two (candidate) channel rows per wave instruction, so a chunk is 3 DMA
      // instructions per wave instead of 12 loads + 12 ds_write_b32 per THREAD; chunk ch + 1 flies under chunk ch's MFMAs.
      const uint32_t cA = (uint32_t)N * 4u, cB = (uint32_t)M * 4u;
      const uint32_t b_col = m0 + (lane & 31) * 4 < M ? (uint32_t)(m0 + (lane & 31) * 4) * 4u : GE_OOB;
      auto issue = [&](int ch, int st) {
        const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane(wave);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)st * (KNN_STAGE * 4u)));
        const uint32_t c0 = (uint32_t)ch * KNN_KC;
        lds_dma16(xrw, base + wu * 1024u,
                  __builtin_elementwise_add_sat(a_dma, (c0 + 4u * wu + (lane >> 4)) * cA));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const uint32_t j = 2u * wu + e;
          lds_dma16(yrw, base + (uint32_t)(KNN_KC * KNN_ROWS * 4) + j * 1024u,
                    __builtin_elementwise_add_sat(b_col, (c0 + 2u * j + (lane >> 5)) * cB));
        }
      };
      const int nch = (GE_KNN_DBG & 2) ? 1 : nchunks;
      issue(0, 0);
      for (int ch = 0; ch < nch; ++ch) {
        lds_dma_wait<0>();            // this wave's part of chunk ch has landed
        __syncthreads();              // ... everybody's; and everybody is done with the stage of chunk ch - 1
        if (ch + 1 < nch) issue(ch + 1, (ch + 1) & 1);
        mma_chunk16(sOp + (ch & 1) * KNN_STAGE);
      }
      __syncthreads();
    } else {
    load(0, m0);
    stage(sOp);
    __syncthreads();
    for (int ch = 0; ch < ((GE_KNN_DBG & 2) ? 1 : nchunks); ++ch) {
      const float* cur = sOp + (ch & 1) * KNN_STAGE;
      if (ch + 1 < nchunks && !(GE_KNN_DBG & 4)) load((ch + 1) * KNN_KC, m0);
      mma_chunk16(cur);
      if (ch + 1 < nchunks) stage(sOp + ((ch + 1) & 1) * KNN_STAGE);
      __syncthreads();
    }
    }
    {
      const int mc0 = m0 + 64 * wn + li, mc1 = mc0 + 32;
      const float sy0 = buf_load(syrs, (uint32_t)mc0 * 4u), sy1 = buf_load(syrs, (uint32_t)mc1 * 4u);   // 0 past M
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const int n = n0 + row;
        const float sqr = FUSE ? s_sq[row] : buf_load(sxrs, (uint32_t)n * 4u);     // squared norm of the query row (0 past N: unused)
        float d0 = (sqr + (-2.f * acc0[r])) + sy0, d1 = (sqr + (-2.f * acc1[r])) + sy1;
        if (relpos) {
          d0 += buf_load(rprs, (uint32_t)(n * M + mc0) * 4u);
          d1 += buf_load(rprs, (uint32_t)(n * M + mc1) * 4u);
        }
        sD[row * KNN_DP + 64 * wn + li] = (n < N && mc0 < M) ? d0 : INFINITY;
        sD[row * KNN_DP + 64 * wn + 32 + li] = (n < N && mc1 < M) ? d1 : INFINITY;
      }
    }
    __syncthreads();

    // ---- selection: merge 128 new candidates into each row's sorted list (K arg-min extractions)
    if (GE_KNN_DBG & 1) {
      if (tid == 0 && sD[0] == 12345.f) best[0] = 0;      // keeps the tile alive
    } else if (G16) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int row = wave * 16 + pass * 4 + grp;
        u64 c[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int idx = m0 + l16 + 16 * q;
          c[q] = idx < M ? knn_key(sD[row * KNN_DP + l16 + 16 * q], idx) : KNN_KEY_INF;
        }
        // Round 4: the lane's eight candidates are SORTED first (Batcher's 19-comparator network), then the K extractions
        // are a 16-way merge: a lane offers min(head of its sorted run, its entry of the old list), and only the winner
        // advances its run (seven register moves under one condition) -- instead of re-deriving every lane's minimum of
        // eight and knocking the winner out of eight slots in every round (94 -> ~51 VALU instructions per round).
        // Same winners in the same order: the keys are unique.
#define GE_KNN_CE(i, j)                                   \
  {                                                       \
    const bool sw = c[j] < c[i];                          \
    const u64 lo = sw ? c[j] : c[i], hi = sw ? c[i] : c[j]; \
    c[i] = lo;                                            \
    c[j] = hi;                                            \
  }
        if (GE_KNN_SORTED) {
          GE_KNN_CE(0, 1) GE_KNN_CE(2, 3) GE_KNN_CE(4, 5) GE_KNN_CE(6, 7)
          GE_KNN_CE(0, 2) GE_KNN_CE(1, 3) GE_KNN_CE(4, 6) GE_KNN_CE(5, 7)
          GE_KNN_CE(1, 2) GE_KNN_CE(5, 6)
          GE_KNN_CE(0, 4) GE_KNN_CE(1, 5) GE_KNN_CE(2, 6) GE_KNN_CE(3, 7)
          GE_KNN_CE(2, 4) GE_KNN_CE(3, 5)
          GE_KNN_CE(1, 2) GE_KNN_CE(3, 4) GE_KNN_CE(5, 6)
        }
#undef GE_KNN_CE
        u64 old = best[pass], mine = KNN_KEY_INF;
        if (GE_KNN_SORTED) {
          for (int t = 0; t < K; ++t) {
            const u64 k = c[0] < old ? c[0] : old;
            const u64 win = row16_min64(k);
            const bool adv = c[0] == win;      // unique keys: at most one lane of the row, through c[0] or through old
            if (old == win) old = KNN_KEY_INF;
#pragma unroll
            for (int q = 0; q < 7; ++q) c[q] = adv ? c[q + 1] : c[q];
            c[7] = adv ? KNN_KEY_INF : c[7];
            if (l16 == t) mine = win;
          }
        } else {
        for (int t = 0; t < K; ++t) {
          u64 k = old;
#pragma unroll
          for (int q = 0; q < 8; ++q) k = c[q] < k ? c[q] : k;
          const u64 win = row16_min64(k);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (c[q] == win) c[q] = KNN_KEY_INF;   // keys are unique (distinct indices): exactly one slot matches
          if (old == win) old = KNN_KEY_INF;
          if (l16 == t) mine = win;
        }
        }
        best[pass] = mine;
        __builtin_amdgcn_sched_barrier(0);   // keep the passes sequential: their candidate registers must not overlap
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 16 + r;
        const int i0 = m0 + lane, i1 = m0 + 64 + lane;
        u64 k0 = i0 < M ? knn_key(sD[row * KNN_DP + lane], i0) : KNN_KEY_INF;
        u64 k1 = i1 < M ? knn_key(sD[row * KNN_DP + 64 + lane], i1) : KNN_KEY_INF;
        u64 k2 = best[r];
        u64 mine = KNN_KEY_INF;
        for (int t = 0; t < K; ++t) {
          u64 k = k0 < k1 ? k0 : k1;
          k = k2 < k ? k2 : k;
          const u64 win = wave_min64(k);
          if (k0 == win) k0 = KNN_KEY_INF;
          if (k1 == win) k1 = KNN_KEY_INF;
          if (k2 == win) k2 = KNN_KEY_INF;
          if (lane == t) mine = win;
        }
        best[r] = mine;
      }
    }
    __syncthreads();
  }

  const int Kout = (K + dil - 1) / dil;
  const size_t half = (size_t)B * N * Kout;
  if (G16) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int n = n0 + wave * 16 + pass * 4 + grp;
      if (n < N && l16 < K && l16 % dil == 0) {
        const size_t o = ((size_t)b * N + n) * Kout + l16 / dil;
        out[o] = (long long)(unsigned)best[pass];   // low word of the key = candidate index
        out[half + o] = (long long)n;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wave * 16 + r;
      if (n < N && lane < K && lane % dil == 0) {
        const size_t o = ((size_t)b * N + n) * Kout + lane / dil;
        out[o] = (long long)(unsigned)best[r];
        out[half + o] = (long long)n;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Max-relative aggregation (MRConv2d.forward, vig.py:96-104), output channel-interleaved [x_0, m_0, x_1, m_1, ...]
//   m_c[n] = max_k ( y[c][idx0[n][k]] - x[c][idx1[n][k]] ),  first k wins ties; argk saved for backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mr_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const long long* __restrict__ edge, float* __restrict__ out,
                                                     unsigned char* __restrict__ argk, int B, int C, int N, int M,
                                                     int K) {
  extern __shared__ __attribute__((aligned(16))) int sidx[];  // [64][K][2]
  const int b = blockIdx.y, n0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  const size_t half = (size_t)B * N * K;
  for (int e = tid; e < 64 * K; e += 256) {
    const int nl = e / K, k = e - nl * K;
    const int n = n0 + nl;
    int i0 = 0, i1 = 0;
    if (n < N) {
      const size_t o = ((size_t)b * N + n) * K + k;
      i0 = (int)edge[o];
      i1 = (int)edge[half + o];
    }
    sidx[e * 2] = i0;
    sidx[e * 2 + 1] = i1;
  }
  __syncthreads();
  const int nl = tid & 63, n = n0 + nl;
  if (n >= N) return;
  const float* xb = x + (size_t)b * C * N;
  const float* yb = y + (size_t)b * C * M;
  float* ob = out + (size_t)b * 2 * C * N;
  for (int c = tid >> 6; c < C; c += 4) {
    const float* xc = xb + (size_t)c * N;
    const float* yc = yb + (size_t)c * M;
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
      const float v = yc[sidx[(nl * K + k) * 2]] - xc[sidx[(nl * K + k) * 2 + 1]];
      if (k == 0 || v > best) {
        best = v;
        bk = k;
      }
    }
    ob[(size_t)(2 * c) * N + n] = xc[n];
    ob[(size_t)(2 * c + 1) * N + n] = best;
    argk[((size_t)b * C + c) * N + n] = (unsigned char)bk;
  }
}

// Tiled form for the centre-is-self graph (edge[1][b][n][k] == n, what ge_knn_topk writes) and M small enough for
// LDS.  One workgroup owns MR_CT channels x 256 nodes.  The candidate slab y[c0:c0+CT][0:M] (one contiguous run) is
// copied into LDS; a lane is a node, keeps its K neighbour ids in registers and walks the channels: the K gathers
// per (node, channel) are LDS reads instead of divergent global loads (the texture-address path retires ~4 divergent
// lanes per clock), and every global access -- x, both output channel planes, argk -- is coalesced along n.
constexpr int MR_CT = 32, MR_NT = 256;
// Channel tile of the backward kernel: 8 channels keep the LDS accumulator at 8*M floats (nine workgroups per CU
// instead of three) and give enough (b, c-tile) workgroups that the node range needs few splits -- the per-split
// accumulators are extra HBM traffic (67 MB written and read back at p2 with 32-channel tiles and 8 splits).
constexpr int MR_CTB = 8;

// KT > 0: compile-time neighbour count (ids in registers); KT == 0: runtime K (ids in LDS, [K][256]).
template <int KT>
__global__ __launch_bounds__(256) void mr_fwd_tile_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const long long* __restrict__ edge,
                                                          float* __restrict__ out, unsigned char* __restrict__ argk,
                                                          int B, int C, int N, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) float smr[];
  float* sY = smr;                       // [CT][M]
  int* sI = (int*)(sY + MR_CT * M);      // [K][256] (KT == 0 only)
  const int tid = threadIdx.x;
  const int n = blockIdx.x * MR_NT + tid, c0 = blockIdx.y * MR_CT, b = blockIdx.z;
  const int cn = min(MR_CT, C - c0);
  const float* ysrc = y + ((size_t)b * C + c0) * M;
  const int count = cn * M;
  if ((M & 3) == 0) {
    for (int i = tid * 4; i < count; i += 1024) *(float4*)(sY + i) = *(const float4*)(ysrc + i);
  } else {
    for (int i = tid; i < count; i += 256) sY[i] = ysrc[i];
  }
  const bool nok = n < N;
  const long long* ep = edge + ((size_t)b * N + (nok ? n : 0)) * K;
  int id[KT > 0 ? KT : 1];
  if (KT > 0) {
#pragma unroll
    for (int k = 0; k < KT; ++k) id[k] = (int)ep[k];
  } else {
    for (int k = 0; k < K; ++k) sI[k * 256 + tid] = (int)ep[k];
  }
  const float* xc = x + ((size_t)b * C + c0) * N + (nok ? n : 0);
  float xv[KT > 0 ? MR_CT : 1];
  if (KT > 0) {
#pragma unroll
    for (int c = 0; c < MR_CT; ++c) xv[c] = xc[(size_t)(c < cn ? c : 0) * N];
  }
  __syncthreads();
  if (!nok) return;
  float* oc = out + ((size_t)b * 2 * C + 2 * c0) * N + n;
  unsigned char* ac = argk + ((size_t)b * C + c0) * N + n;
  if (KT > 0) {
    // (x loads for all channels were issued before the barrier: one memory round trip per node, not one per channel)
#pragma unroll
    for (int c = 0; c < MR_CT; c += 4) {   // four channels in flight: 4*KT independent LDS gathers
      float yv[4][KT > 0 ? KT : 1];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < KT; ++k) yv[u][k] = sY[(c + u) * M + id[k]];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float best = yv[u][0] - xv[c + u];
        int bk = 0;
#pragma unroll
        for (int k = 1; k < KT; ++k) {
          const float v = yv[u][k] - xv[c + u];
          if (v > best) {   // first k wins ties
            best = v;
            bk = k;
          }
        }
        if (c + u < cn) {
          oc[(size_t)(2 * (c + u)) * N] = xv[c + u];
          oc[(size_t)(2 * (c + u) + 1) * N] = best;
          ac[(size_t)(c + u) * N] = (unsigned char)bk;
        }
      }
    }
  } else {
    for (int c = 0; c < cn; ++c) {
      const float xs = xc[(size_t)c * N];
      float best = sY[c * M + sI[tid]] - xs;
      int bk = 0;
      for (int k = 1; k < K; ++k) {
        const float v = sY[c * M + sI[k * 256 + tid]] - xs;
        if (v > best) {
          best = v;
          bk = k;
        }
      }
      oc[(size_t)(2 * c) * N] = xs;
      oc[(size_t)(2 * c + 1) * N] = best;
      ac[(size_t)c * N] = (unsigned char)bk;
    }
  }
}

// K = 9 (every Grapher of the reference), C a multiple of 4: the candidate slab sits in LDS TRANSPOSED, sYt[m][MR_CT + 4]
// (144-byte rows), so that a lane fetches FOUR channels of one neighbour with a single ds_read_b128 -- 72 wide gathers
// per node instead of 288 scalar ones (the scalar form spent as long in the LDS crossbar as in HBM: ~3 of the random
// 64 lanes land on every bank).  The fill reads y along m (coalesced) and writes 16-byte rows: eight consecutive m hit
// 32 different banks (36 m + c mod 32), conflict-free.
constexpr int MR_PITCH = MR_CT + 4;
__global__ __launch_bounds__(256) void mr_fwd_quad_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const long long* __restrict__ edge, float* __restrict__ out,
                                                          unsigned char* __restrict__ argk, int B, int C, int N, int M) {
  extern __shared__ __attribute__((aligned(16))) float smr[];   // [M][MR_PITCH]
  constexpr int K = 9;
  const int tid = threadIdx.x;
  const int n = blockIdx.x * MR_NT + tid, c0 = blockIdx.y * MR_CT, b = blockIdx.z;
  const int cn = min(MR_CT, C - c0);                             // multiple of 4
  const float* ysrc = y + ((size_t)b * C + c0) * M;
  const int items = M * (cn >> 2);                               // (candidate, channel quad) pairs, m fastest
  for (int i = tid; i < items; i += 256) {
    const int q = i / M, m = i - q * M;
    const float* yp = ysrc + (size_t)(4 * q) * M + m;
    float4 v;
    v.x = yp[0];
    v.y = yp[M];
    v.z = yp[2 * M];
    v.w = yp[3 * M];
    *(float4*)(smr + m * MR_PITCH + 4 * q) = v;
  }
  const bool nok = n < N;
  const long long* ep = edge + ((size_t)b * N + (nok ? n : 0)) * K;
  int id[K];
#pragma unroll
  for (int k = 0; k < K; ++k) id[k] = (int)ep[k] * MR_PITCH;
  const float* xc = x + ((size_t)b * C + c0) * N + (nok ? n : 0);
  float xv[MR_CT];
#pragma unroll
  for (int c = 0; c < MR_CT; ++c) xv[c] = xc[(size_t)(c < cn ? c : 0) * N];
  __syncthreads();
  if (!nok) return;
  float* oc = out + ((size_t)b * 2 * C + 2 * c0) * N + n;
  unsigned char* ac = argk + ((size_t)b * C + c0) * N + n;
#pragma unroll
  for (int q = 0; q < MR_CT / 4; ++q) {
    if (4 * q >= cn) break;
    float4 yv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) yv[k] = *(const float4*)(smr + id[k] + 4 * q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 4 * q + u;
      const float xs = xv[c];
      auto comp = [&](int k) { return u == 0 ? yv[k].x : (u == 1 ? yv[k].y : (u == 2 ? yv[k].z : yv[k].w)); };
      float best = comp(0) - xs;
      int bk = 0;
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const float v = comp(k) - xs;
        if (v > best) {   // first k wins ties
          best = v;
          bk = k;
        }
      }
      oc[(size_t)(2 * c) * N] = xs;
      oc[(size_t)(2 * c + 1) * N] = best;
      ac[(size_t)c * N] = (unsigned char)bk;
    }
  }
}

// Backward, tiled form (centre-is-self).  Workgroup (c-tile, node split s, b) walks its 256-node chunks; a lane is a
// node: it reads g = dout[2c+1][n] and the winning slot (both coalesced), looks the neighbour id up in the chunk's
// LDS id table and adds g into the LDS accumulator sD[c][m] (ds_add_f32), and writes the centre side
// dx[c][n] = dout[2c][n] - g directly.  The accumulator goes to part[s][b][c][m]; mr_bwd_sum_kernel folds the
// splits (no global atomics).
__global__ __launch_bounds__(256) void mr_bwd_tile_kernel(const float* __restrict__ dout,
                                                          const long long* __restrict__ edge,
                                                          const unsigned char* __restrict__ argk,
                                                          float* __restrict__ dx, float* __restrict__ part, int B,
                                                          int C, int N, int M, int K, int chunks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smr[];
  float* sD = smr;                       // [CT][M]
  int* sI = (int*)(sD + MR_CTB * M);      // [K][256]
  const int c0 = blockIdx.x * MR_CTB, s = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int cn = min(MR_CTB, C - c0);
  for (int i = tid; i < MR_CTB * M; i += 256) sD[i] = 0.f;
  const int nchunks = (N + MR_NT - 1) / MR_NT;
  const int ch0 = s * chunks_per_split, ch1 = min(nchunks, ch0 + chunks_per_split);
  for (int ch = ch0; ch < ch1; ++ch) {
    const int n = ch * MR_NT + tid;
    const bool nok = n < N;
    __syncthreads();   // sD zeroed / previous chunk done with the id table
    const long long* ep = edge + ((size_t)b * N + (nok ? n : 0)) * K;
    for (int k = 0; k < K; ++k) sI[k * 256 + tid] = (int)ep[k];   // own column only: no barrier needed to read it back
    if (!nok) continue;
    const float* gev = dout + ((size_t)b * 2 * C + 2 * c0) * N + n;
    const unsigned char* ac = argk + ((size_t)b * C + c0) * N + n;
    float* dxc = dx + ((size_t)b * C + c0) * N + n;
    // every channel's loads in flight at once (one memory round trip per chunk), then the LDS adds and the stores
    float ge[MR_CTB], go[MR_CTB];
    int kk[MR_CTB];
#pragma unroll
    for (int c = 0; c < MR_CTB; ++c) {
      const int cc = c < cn ? c : 0;
      ge[c] = gev[(size_t)(2 * cc) * N];
      go[c] = gev[(size_t)(2 * cc + 1) * N];
      kk[c] = ac[(size_t)cc * N];
    }
#pragma unroll
    for (int c = 0; c < MR_CTB; ++c) {
      if (c < cn) {
        const int i0 = sI[kk[c] * 256 + tid];
        dxc[(size_t)c * N] = ge[c] - go[c];
        if (go[c] != 0.f) atomicAdd(&sD[c * M + i0], go[c]);
      }
    }
  }
  __syncthreads();
  float* pb = part + (((size_t)s * B + b) * C + c0) * M;
  for (int i = tid; i < cn * M; i += 256) pb[i] = sD[i];
}

// dst[i] = (accumulate ? dst[i] : 0) + sum_s part[s][i]
__global__ __launch_bounds__(256) void mr_bwd_sum_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                         long long total, int S, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float a = accumulate ? dst[i] : 0.f;
    for (int s = 0; s < S; ++s) a += part[(size_t)s * total + i];
    dst[i] = a;
  }
}

// Backward of the max-relative aggregation: one workgroup per (b, c) accumulates the scatter in LDS
// (ds_add_f32), then writes each output once -- no global atomics.
//   dx[c][i] = dout[2c][i] - sum_{n: idx1[n][argk]=i} g_n (+ sum_{n: idx0[n][argk]=i} g_n when y is x)
//   dy[c][m] = sum_{n: idx0[n][argk]=m} g_n,   g_n = dout[2c+1][n]
__global__ __launch_bounds__(256) void mr_bwd_kernel(const float* __restrict__ dout, const long long* __restrict__ edge,
                                                     const unsigned char* __restrict__ argk, float* __restrict__ dx,
                                                     float* __restrict__ dy, int B, int C, int N, int M, int K,
                                                     int y_is_x) {
  extern __shared__ __attribute__((aligned(16))) float sacc[];  // [M] neighbour side, then [N] centre side
  float* sdy = sacc;
  float* sdx = sacc + M;
  const int c = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < M + N; i += 256) sacc[i] = 0.f;
  __syncthreads();
  const size_t half = (size_t)B * N * K;
  const float* god = dout + ((size_t)b * 2 * C + 2 * c + 1) * N;
  const unsigned char* ak = argk + ((size_t)b * C + c) * N;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float g = god[n];
    const size_t o = ((size_t)b * N + n) * K + ak[n];
    atomicAdd(&sdy[(int)edge[o]], g);
    atomicAdd(&sdx[(int)edge[half + o]], -g);
  }
  __syncthreads();
  const float* gev = dout + ((size_t)b * 2 * C + 2 * c) * N;
  float* dxp = dx + ((size_t)b * C + c) * N;
  for (int n = threadIdx.x; n < N; n += 256) dxp[n] = gev[n] + sdx[n] + (y_is_x ? sdy[n] : 0.f);
  if (!y_is_x) {
    float* dyp = dy + ((size_t)b * C + c) * M;
    for (int m = threadIdx.x; m < M; m += 256) dyp[m] = sdy[m];
  }
}

// Fallback for node sets too large for LDS: global atomics.
__global__ __launch_bounds__(256) void mr_bwd_init_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                          long long total, int N) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long bc = i / N;
    const int n = (int)(i - bc * N);
    dx[i] = dout[(size_t)(2 * bc) * N + n];
  }
}
__global__ __launch_bounds__(256) void mr_bwd_scatter_kernel(const float* __restrict__ dout,
                                                             const long long* __restrict__ edge,
                                                             const unsigned char* __restrict__ argk,
                                                             float* __restrict__ dx, float* __restrict__ dy, int B,
                                                             int C, int N, int M, int K) {
  const long long total = (long long)B * C * N;
  const size_t half = (size_t)B * N * K;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int n = (int)(i % N);
    const long long bc = i / N;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    const float g = dout[((size_t)b * 2 * C + 2 * c + 1) * N + n];
    if (g == 0.f) continue;
    const int k = argk[i];
    const size_t o = ((size_t)b * N + n) * K + k;
    const int i0 = (int)edge[o], i1 = (int)edge[half + o];
    atomicAdd(&dy[((size_t)b * C + c) * M + i0], g);
    atomicAdd(&dx[((size_t)b * C + c) * N + i1], -g);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward of the max-relative aggregation as a DETERMINISTIC GATHER over inverse neighbour lists (centre-is-self
// graphs, i.e. everything ge_knn_topk builds: vig.py:88-105 on the graphs of :262-329).
//
//   dy[b][c][m] = sum over the edges (n, k) with edge[b][n][k] == m AND argk[b][c][n] == k of dout[b][2c+1][n]
//
// The scatter form adds those terms with ds_add_f32 in whatever order the waves reach them (not bit-reproducible, and
// neighbouring nodes -- the lanes of a wave -- pick the same neighbour and serialise).  Here the edge list is inverted
// once per backward call: per (b, chunk of MRI_NCH nodes) the (n, k) pairs are grouped by candidate m in a FIXED order
// (wave, round, k, lane: a counting sort whose ranks come from wave ballots, integer LDS counters only), and the backward
// gives every candidate a lane that walks its list front to back.  The chunk's gradients and winning slots sit in LDS
// transposed ([node][8 channels]: one 8-byte and two 16-byte reads serve 8 channels of an entry), the list itself is
// streamed through LDS in whole tiles (coalesced).  No floating-point atomics anywhere: same bits every run.
// ---------------------------------------------------------------------------------------------
constexpr int MRI_NCH = 512;   // nodes per chunk (entry: node-in-chunk in bits 0..15, slot k in bits 16..23)
constexpr int MRI_CT = 8;      // channels per workgroup of the gather
// A chunk is NOT a contiguous run of nodes: granules of 64 consecutive nodes are dealt to the chunks round-robin
// (granule g -> chunk g mod nchunks, position g / nchunks inside it), so every chunk samples the whole node set.  On a
// feature map neighbouring nodes share neighbours: a contiguous chunk (8 image rows) puts all its edges on the ~50
// candidates under it and the other 200 lanes of the gather idle (measured: 268 us on smooth maps vs 185 us on random
// features); interleaved, every candidate gets its share of every chunk.  64 consecutive nodes = one coalesced wave load.
__device__ __forceinline__ int mri_node(int chunk, int nl, int nchunks) {      // global node of chunk-local index nl
  return (((nl >> 6) * nchunks + chunk) << 6) | (nl & 63);
}
__device__ __forceinline__ int mri_nodes_before(int chunk, int N, int nchunks) {   // nodes in chunks 0 .. chunk-1
  const int NG = (N + 63) >> 6;
  int granules = 0;
  for (int j = 0; j < chunk; ++j) granules += (NG - j + nchunks - 1) / nchunks;
  const int last = (NG - 1) % nchunks;                  // chunk holding the (possibly ragged) last granule
  return granules * 64 - (last < chunk ? NG * 64 - N : 0);
}

// The inversion of one chunk's edges by the 256 threads of a workgroup: `dst` receives the chunk's (node-in-chunk, k)
// entries grouped by candidate, `ofs[0..M]` the segment starts; hist: 4*M ints of LDS, part: 256 ints of LDS.  dst / ofs
// may point to global memory (mr_inv_build_kernel) or to LDS (mr_bwd_small_kernel).  Ends with a barrier.
// KT = 9 (every Grapher of the reference): a lane's 2 x 9 neighbour ids are loaded ONCE, all loads in flight together --
// with the ids fetched inside the counting and filling loops every one of the 36 loads was a memory round trip in front
// of an LDS atomic / a ballot chain (12 us of pure latency per workgroup); KT = 0: any K, ids re-read in the loops.
template <int KT>
__device__ __forceinline__ void mri_build_chunk(const long long* __restrict__ eb, unsigned* dst, int* ofs, int* hist,
                                                int* part, int chunk, int nchunks, int N, int M, int K) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  constexpr int PER_WAVE = MRI_NCH / 4, ROUNDS = PER_WAVE / 64;
  int ids[ROUNDS][KT > 0 ? KT : 1];
  bool valid[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int n = mri_node(chunk, w * PER_WAVE + r * 64 + lane, nchunks);
    valid[r] = n < N;
    if (KT > 0) {
      const long long* ep = eb + (size_t)(valid[r] ? n : 0) * KT;
#pragma unroll
      for (int k = 0; k < KT; ++k) ids[r][k] = (int)ep[k];
    }
  }
  for (int i = tid; i < 4 * M; i += 256) hist[i] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (valid[r]) {
      if (KT > 0) {
#pragma unroll
        for (int k = 0; k < KT; ++k) atomicAdd(&hist[w * M + ids[r][k]], 1);   // integer counters: order-independent
      } else {
        const long long* ep = eb + (size_t)mri_node(chunk, w * PER_WAVE + r * 64 + lane, nchunks) * K;
        for (int k = 0; k < K; ++k) atomicAdd(&hist[w * M + (int)ep[k]], 1);
      }
    }
  }
  __syncthreads();
  // exclusive scan over (m, wave): thread t owns candidates [t * per, (t + 1) * per)
  const int per = (M + 255) / 256;
  const int mb = tid * per, me = min(M, mb + per);
  int mine = 0;
  for (int m = mb; m < me; ++m) mine += hist[m] + hist[M + m] + hist[2 * M + m] + hist[3 * M + m];
  part[tid] = mine;
  __syncthreads();
  int base = 0;
  for (int t = 0; t < tid; ++t) base += part[t];
  for (int m = mb; m < me; ++m) {
    const int c0 = hist[m], c1 = hist[M + m], c2 = hist[2 * M + m], c3 = hist[3 * M + m];
    ofs[m] = base;
    hist[m] = base;                      // the counters become each wave's write cursor of the candidate
    hist[M + m] = base + c0;
    hist[2 * M + m] = base + c0 + c1;
    hist[3 * M + m] = base + c0 + c1 + c2;
    base += c0 + c1 + c2 + c3;
  }
  if (me == M && mb < M) ofs[M] = base;
  __syncthreads();
  volatile int* cur = hist + w * M;
  int mbits = 1;
  while ((1 << mbits) < M) ++mbits;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int nl = w * PER_WAVE + r * 64 + lane;
    const bool ok = valid[r];
    const long long* ep = eb + (size_t)(ok ? mri_node(chunk, nl, nchunks) : 0) * K;
    auto place = [&](int k, int m) {
      // lanes with the same candidate: AND over the key's bits of (lanes whose bit agrees with mine) -- one ballot per
      // bit instead of one LDS read-modify-write per DISTINCT candidate of the wave (50 passes on random graphs)
      unsigned long long same = __ballot(ok);
      for (int bit = 0; bit < mbits; ++bit) {
        const unsigned long long ones = __ballot((m >> bit) & 1);
        same &= ((m >> bit) & 1) ? ones : ~ones;
      }
      const int rank = __popcll(same & ((1ull << lane) - 1ull)), cnt = __popcll(same);
      const int at = ok ? cur[m] : 0;
      if (ok) dst[at + rank] = (unsigned)nl | ((unsigned)k << 16);
      __builtin_amdgcn_wave_barrier();            // every lane's read of the cursor precedes its update (LDS is in order)
      if (ok && rank == cnt - 1) cur[m] = at + cnt;      // one lane per distinct candidate
      __builtin_amdgcn_wave_barrier();
    };
    if (KT > 0) {
#pragma unroll
      for (int k = 0; k < KT; ++k) place(k, ok ? ids[r][k] : 0);
    } else {
      for (int k = 0; k < K; ++k) place(k, ok ? (int)ep[k] : 0);
    }
  }
  __syncthreads();
}

template <int KT>
__global__ __launch_bounds__(256) void mr_inv_build_kernel(const long long* __restrict__ edge, unsigned* __restrict__ inv,
                                                           int* __restrict__ off, int B, int N, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) int smi[];   // counters [4 waves][M], then 256 partial sums
  const int chunk = blockIdx.x, b = blockIdx.y, nchunks = gridDim.x;
  mri_build_chunk<KT>(edge + (size_t)b * N * K, inv + ((size_t)b * N + mri_nodes_before(chunk, N, nchunks)) * K,
                      off + ((size_t)b * nchunks + chunk) * (M + 1), smi, smi + 4 * M, chunk, nchunks, N, M, K);
}

constexpr int MRI_PITCH = 12;  // words per staged node row: 8 gradients + 4 pad (16-byte reads / writes of consecutive
                               // nodes then fall on disjoint bank quads)
// Stage one chunk's gradients (odd half of dout) and winning slots into LDS, transposed [node][8 channels]; `centre`
// (uniform): also write the centre side dx = dout_even - dout_odd of the chunk's nodes.  A thread owns (node, channel
// quad) items; every global access runs along the nodes.  All loads UNCONDITIONAL on clamped (always valid) addresses,
// selects afterwards: a predicated load is a branch with its own wait, and 48 of them in a row made the staging a chain
// of memory round trips.
__device__ __forceinline__ void mri_stage_chunk(const float* __restrict__ dob, const unsigned char* __restrict__ akb,
                                                float* __restrict__ dxb, float* sGo, unsigned* sArg, int chunk,
                                                int nchunks, int N, int cn, bool centre) {
  const int tid = threadIdx.x;
  constexpr int ITEMS = MRI_NCH * 2 / 256;
  float g[ITEMS][4], ge[ITEMS][4];
  unsigned a4[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = it * 256 + tid, q = idx / MRI_NCH, nl = idx - q * MRI_NCH;
    const int nc = min(mri_node(chunk, nl, nchunks), N - 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = min(4 * q + u, cn - 1);
      g[it][u] = dob[(size_t)(2 * cc + 1) * N + nc];
      a4[it] = (u == 0 ? 0u : a4[it]) | ((unsigned)akb[(size_t)cc * N + nc] << (8 * u));
    }
  }
  if (centre) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = it * 256 + tid, q = idx / MRI_NCH, nl = idx - q * MRI_NCH;
      const int nc = min(mri_node(chunk, nl, nchunks), N - 1);
#pragma unroll
      for (int u = 0; u < 4; ++u) ge[it][u] = dob[(size_t)(2 * min(4 * q + u, cn - 1)) * N + nc];
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = it * 256 + tid, q = idx / MRI_NCH, nl = idx - q * MRI_NCH;
    const int n = mri_node(chunk, nl, nchunks);
    const bool nok = n < N;
    unsigned av = a4[it];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = nok && 4 * q + u < cn;
      if (!ok) {
        g[it][u] = 0.f;                    // rows / channels beyond the tile: no gradient, a slot no entry has
        av |= 0xFFu << (8 * u);
      }
    }
    *(float4*)(sGo + nl * MRI_PITCH + 4 * q) = make_float4(g[it][0], g[it][1], g[it][2], g[it][3]);
    sArg[nl * 2 + q] = av;
    if (centre && nok) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * q + u < cn) dxb[(size_t)(4 * q + u) * N + n] = ge[it][u] - g[it][u];      // centre side
    }
  }
}

// A candidate's list segment [s0, s1) of sInv, front to back (two entries in flight; the order of the adds is fixed).
__device__ __forceinline__ void mri_walk(float (&acc)[MRI_CT], const unsigned* sInv, int s0, int s1, const unsigned* sArg,
                                         const float* sGo) {
  auto take = [&](unsigned e) {
    const unsigned nl = e & 0xFFFFu, k = e >> 16;
    const uint2 a = *(const uint2*)(sArg + nl * 2);
    const float4 g0 = *(const float4*)(sGo + nl * MRI_PITCH), g1 = *(const float4*)(sGo + nl * MRI_PITCH + 4);
    acc[0] += ((a.x & 0xFFu) == k) ? g0.x : 0.f;
    acc[1] += (((a.x >> 8) & 0xFFu) == k) ? g0.y : 0.f;
    acc[2] += (((a.x >> 16) & 0xFFu) == k) ? g0.z : 0.f;
    acc[3] += ((a.x >> 24) == k) ? g0.w : 0.f;
    acc[4] += ((a.y & 0xFFu) == k) ? g1.x : 0.f;
    acc[5] += (((a.y >> 8) & 0xFFu) == k) ? g1.y : 0.f;
    acc[6] += (((a.y >> 16) & 0xFFu) == k) ? g1.z : 0.f;
    acc[7] += ((a.y >> 24) == k) ? g1.w : 0.f;
  };
  int i = s0;
  for (; i + 1 < s1; i += 2) {
    const unsigned ea = sInv[i], eb = sInv[i + 1];
    take(ea);
    take(eb);
  }
  if (i < s1) take(sInv[i]);
}

template <bool SELF>
__global__ __launch_bounds__(256) void mr_bwd_gather_kernel(const float* __restrict__ dout,
                                                            const unsigned* __restrict__ inv,
                                                            const int* __restrict__ off,
                                                            const unsigned char* __restrict__ argk,
                                                            float* __restrict__ dx, float* __restrict__ dy, int B, int C,
                                                            int N, int M, int K, int nchunks, int mgroups, int cps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smg[];
  float* sGo = (float*)smg;                                        // [MRI_NCH][MRI_PITCH]
  unsigned* sArg = (unsigned*)(smg + MRI_NCH * MRI_PITCH * 4);     // [MRI_NCH][2] (8 winning slots, one byte each)
  unsigned* sInv = sArg + MRI_NCH * 2;                             // [<= MRI_NCH * K]
  const int mg = blockIdx.x % mgroups, split = blockIdx.x / mgroups;
  const int c0 = blockIdx.y * MRI_CT, b = blockIdx.z, tid = threadIdx.x;
  const int m = mg * 256 + tid;
  const bool mok = m < M;
  const int cn = min(MRI_CT, C - c0);
  float acc[MRI_CT];
#pragma unroll
  for (int c = 0; c < MRI_CT; ++c) acc[c] = 0.f;
  const float* dob = dout + ((size_t)b * 2 * C + 2 * c0) * N;
  const unsigned char* akb = argk + ((size_t)b * C + c0) * N;
  float* dxb = dx + ((size_t)b * C + c0) * N;
  const int ch0 = split * cps, ch1 = min(nchunks, ch0 + cps);
  for (int chunk = ch0; chunk < ch1; ++chunk) {
    __syncthreads();                         // the previous chunk's entries have been consumed
    mri_stage_chunk(dob, akb, dxb, sGo, sArg, chunk, nchunks, N, cn, !SELF && mg == 0);
    // ---- this chunk's list entries of the workgroup's candidates: one contiguous run ----
    const int* ofs = off + ((size_t)b * nchunks + chunk) * (M + 1);
    const int e0 = ofs[mg * 256], e1 = ofs[min(M, mg * 256 + 256)];
    const unsigned* ip = inv + ((size_t)b * N + mri_nodes_before(chunk, N, nchunks)) * K + e0;
    const int cnt = e1 - e0;
    for (int i = tid; i < cnt; i += 1024) {
      const unsigned v0 = ip[i], v1 = i + 256 < cnt ? ip[i + 256] : 0u, v2 = i + 512 < cnt ? ip[i + 512] : 0u,
                     v3 = i + 768 < cnt ? ip[i + 768] : 0u;
      sInv[i] = v0;
      if (i + 256 < cnt) sInv[i + 256] = v1;
      if (i + 512 < cnt) sInv[i + 512] = v2;
      if (i + 768 < cnt) sInv[i + 768] = v3;
    }
    int s0 = 0, s1 = 0;
    if (mok) {
      s0 = ofs[m] - e0;
      s1 = ofs[m + 1] - e0;
    }
    __syncthreads();
    mri_walk(acc, sInv, s0, s1, sArg, sGo);
  }
  if (!mok) return;
  const int S = gridDim.x / mgroups;
#pragma unroll
  for (int c = 0; c < MRI_CT; ++c) {
    if (c >= cn) break;
    if (SELF)      // y is x (one split): the node's own centre term and what its neighbours sent, in one store
      dxb[(size_t)c * N + m] = dob[(size_t)(2 * c) * N + m] - dob[(size_t)(2 * c + 1) * N + m] + acc[c];
    else           // S > 1: dy is the [S][B][C][M] partial buffer, folded in split order by mr_bwd_sum_kernel
      dy[(((size_t)(S > 1 ? split : 0) * B + b) * C + c0 + c) * M + m] = acc[c];
  }
}

// Graphs of at most MRI_NCH nodes (one chunk: the 16 x 16 and 8 x 8 pyramid levels, TGCN's 64-node graphs, the late
// stages of the pyramid ViG) in ONE launch: every (b, 8-channel) workgroup inverts the edge list itself, in LDS -- the
// same list for all channel tiles, rebuilt C/8 times, but 4 608 entries take a few microseconds and the separate build
// launch cost 12 us of latency (and its global round trip) for graphs whose whole backward is worth less than that.
template <bool SELF, int KT>
__global__ __launch_bounds__(256) void mr_bwd_small_kernel(const float* __restrict__ dout,
                                                           const long long* __restrict__ edge,
                                                           const unsigned char* __restrict__ argk,
                                                           float* __restrict__ dx, float* __restrict__ dy, int B, int C,
                                                           int N, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smg[];
  float* sGo = (float*)smg;                                        // [MRI_NCH][MRI_PITCH]
  unsigned* sArg = (unsigned*)(smg + MRI_NCH * MRI_PITCH * 4);     // [MRI_NCH][2]
  unsigned* sInv = sArg + MRI_NCH * 2;                             // [N * K]
  int* sOff = (int*)(sInv + (size_t)N * K);                        // [M + 1]
  int* hist = sOff + M + 1;                                        // [4 * M], then 256 partial sums
  const int c0 = blockIdx.x * MRI_CT, b = blockIdx.y, tid = threadIdx.x;
  const int cn = min(MRI_CT, C - c0);
  const float* dob = dout + ((size_t)b * 2 * C + 2 * c0) * N;
  const unsigned char* akb = argk + ((size_t)b * C + c0) * N;
  float* dxb = dx + ((size_t)b * C + c0) * N;
  mri_stage_chunk(dob, akb, dxb, sGo, sArg, 0, 1, N, cn, !SELF);
  mri_build_chunk<KT>(edge + (size_t)b * N * K, sInv, sOff, hist, hist + 4 * M, 0, 1, N, M, K);   // ends with a barrier
  for (int m = tid; m < M; m += 256) {
    float acc[MRI_CT];
#pragma unroll
    for (int c = 0; c < MRI_CT; ++c) acc[c] = 0.f;
    mri_walk(acc, sInv, sOff[m], sOff[m + 1], sArg, sGo);
#pragma unroll
    for (int c = 0; c < MRI_CT; ++c) {
      if (c >= cn) break;
      if (SELF)
        dxb[(size_t)c * N + m] = dob[(size_t)(2 * c) * N + m] - dob[(size_t)(2 * c + 1) * N + m] + acc[c];
      else
        dy[((size_t)b * C + c0 + c) * M + m] = acc[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// batched_index_select (vig.py:209-229): out[b][c][e] = src[b][c][idx[b][e]], e over the N*K edges.
// Backward: one workgroup per (b, c) scatters the edge gradients into an LDS copy of the row (ds_add_f32) and writes
// each dsrc element once -- no global atomics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_gather_fwd_kernel(const float* __restrict__ src,
                                                              const long long* __restrict__ idx,
                                                              float* __restrict__ out, int C, int M, int E) {
  const int b = blockIdx.z, c = blockIdx.y;
  const float* row = src + ((size_t)b * C + c) * M;
  const long long* ib = idx + (size_t)b * E;
  float* o = out + ((size_t)b * C + c) * E;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256) o[e] = row[ib[e]];
}
__global__ __launch_bounds__(256) void edge_gather_bwd_kernel(const float* __restrict__ dout,
                                                              const long long* __restrict__ idx,
                                                              float* __restrict__ dsrc, int C, int M, int E) {
  extern __shared__ __attribute__((aligned(16))) float srow[];
  const int b = blockIdx.y, c = blockIdx.x;
  for (int i = threadIdx.x; i < M; i += 256) srow[i] = 0.f;
  __syncthreads();
  const float* g = dout + ((size_t)b * C + c) * E;
  const long long* ib = idx + (size_t)b * E;
  for (int e = threadIdx.x; e < E; e += 256) atomicAdd(&srow[ib[e]], g[e]);
  __syncthreads();
  float* d = dsrc + ((size_t)b * C + c) * M;
  for (int i = threadIdx.x; i < M; i += 256) d[i] = srow[i];
}

extern "C" {

// xn [B][C][P], sq [B][P]
int ge_knn_prepare(const float* x, float* xn, float* sq, int B, int C, int P, int normalize, void* stream) {
  GE_REQUIRE(x && xn && sq && B > 0 && C > 0 && P > 0, "knn_prepare: bad arguments");
  // few points (candidate sets, TGCN's node sets): 64-point tiles keep more of the chip busy and turn the dependent memory round
  // trips into one (B32 x M256: 17 -> 11 us); many points: one thread per point streams at 4-5 TB/s.  Same bits either way.
  const bool tile = (long long)B * P <= 64 * 1024;
  if (tile)
    hipLaunchKernelGGL(knn_prep_tile_kernel, dim3(ge_cdiv(P, KNN_ROWS), B), dim3(256), 0, (hipStream_t)stream, x, xn, sq, C, P,
                       normalize);
  else
    hipLaunchKernelGGL(knn_prep_kernel, dim3(ge_cdiv(P, 256), B), dim3(256), 0, (hipStream_t)stream, x, xn, sq, C, P,
                       normalize);
  GE_CHECK_LAUNCH("knn_prepare");
  return GE_OK;
}

static int knn_launch(const float* xraw, const float* xn, const float* sqx, const float* yn, const float* sqy, const float* relpos,
                      long long* edge_index, int B, int C, int N, int M, int K, int dilation, bool fuse, int normalize,
                      hipStream_t st, const char* name) {
  GE_REQUIRE(K >= 1 && K <= 64 && K <= M && dilation >= 1, "%s: need 1 <= K <= min(64, M)", name);
  GE_REQUIRE(C >= 1, "%s: C must be positive", name);
  GE_REQUIRE(4ll * C * N < 0xFFFFFFF0ll && 4ll * C * M < 0xFFFFFFF0ll && 4ll * N * M < 0xFFFFFFF0ll,
             "%s: per-item operands of 4 GiB or more are not supported", name);
  // distance tile / operand stages / prologue slab share the space; the fused form keeps 2 x 64 floats behind the tile
  const size_t lds = (std::max((size_t)KNN_ROWS * KNN_DP, (size_t)2 * KNN_STAGE) + 2 * KNN_ROWS) * sizeof(float);
  static_assert(KNN_PRO_CH * KNN_ROWS <= KNN_ROWS * KNN_DP, "prologue slab inside the tile");
  static const int knn_dma = []() {
    const char* e = getenv("GE_KNN_DMA");
    return (e && e[0] == '0') ? 0 : 1;
  }();
  GE_MAX_LDS(160 * 1024, (const void*)knn_topk_kernel<true, false>, (const void*)knn_topk_kernel<false, false>,
             (const void*)knn_topk_kernel<true, true>, (const void*)knn_topk_kernel<false, true>);
  const dim3 grid(ge_cdiv(N, KNN_ROWS), B);
#define KNN_GO(G, F)                                                                                                     \
  hipLaunchKernelGGL((knn_topk_kernel<G, F>), grid, dim3(256), lds, st, xn, sqx, yn, sqy, relpos, edge_index, B, C, N, M, K, \
                     dilation, knn_dma, normalize, xraw)
  if (K <= 16) {
    if (fuse) KNN_GO(true, true); else KNN_GO(true, false);
  } else {
    if (fuse) KNN_GO(false, true); else KNN_GO(false, false);
  }
#undef KNN_GO
  return GE_OK;
}

// edge_index int64 [2][B][N][ceil(K/dilation)] from prepared operands; relpos: [N][M] or null.
int ge_knn_topk(const float* xn, const float* sqx, const float* yn, const float* sqy, const float* relpos,
                long long* edge_index, int B, int C, int N, int M, int K, int dilation, void* stream) {
  GE_REQUIRE(xn && sqx && yn && sqy && edge_index, "knn_topk: null pointer");
  const int rc = knn_launch(nullptr, xn, sqx, yn, sqy, relpos, edge_index, B, C, N, M, K, dilation, false, 1, (hipStream_t)stream,
                            "knn_topk");
  if (rc != GE_OK) return rc;
  GE_CHECK_LAUNCH("knn_topk");
  return GE_OK;
}
// The same graph from the RAW query tensor x [B][C][N] (round 6): the kernel normalises its own query rows (normalize != 0:
// x / max(||x||, 1e-12) per point, vig.py:372-378), writes them to xn (workspace of x's size, owned by the caller) and keeps their
// squared norms in LDS -- no ge_knn_prepare pass over x.  yn / sqy: the candidates, prepared by ge_knn_prepare with the same
// `normalize`.  Bit-identical to ge_knn_prepare(x) + ge_knn_topk.
int ge_knn_topk_fused(const float* x, float* xn, const float* yn, const float* sqy, const float* relpos, long long* edge_index,
                      int B, int C, int N, int M, int K, int dilation, int normalize, void* stream) {
  GE_REQUIRE(x && xn && yn && sqy && edge_index, "knn_topk_fused: null pointer");
  const int rc = knn_launch(x, xn, nullptr, yn, sqy, relpos, edge_index, B, C, N, M, K, dilation, true, normalize ? 1 : 0,
                            (hipStream_t)stream, "knn_topk_fused");
  if (rc != GE_OK) return rc;
  GE_CHECK_LAUNCH("knn_topk_fused");
  return GE_OK;
}

static size_t mr_tile_lds(int M, int K) { return ((size_t)MR_CT * M + (size_t)256 * K) * 4; }
// The tiled kernels need the centre-is-self property and the candidate slab in LDS.
static bool mr_tiled(int M, int K, int centre_is_self) { return centre_is_self && mr_tile_lds(M, K) <= 96 * 1024; }
// Node splits of the tiled backward: enough workgroups for ~4 per CU, at most one split per node chunk.
static int mr_bwd_splits(int B, int C, int N) {
  const int nchunks = ge_cdiv(N, MR_NT);
  const long long base = (long long)B * ge_cdiv(C, MR_CTB);
  int s = (int)((2048 + base - 1) / base);
  if (s > nchunks) s = nchunks;
  return s < 1 ? 1 : s;
}

// out [B][2C][N], argk uint8 [B][C][N]; edge int64 [2][B][N][K] (neighbour ids into y, centre ids into x).
// centre_is_self != 0 asserts edge[1][b][n][k] == n (the graphs ge_knn_topk builds) and enables the tiled kernel.
int ge_mrconv_gather_fwd(const float* x, const float* y, const long long* edge, float* out, unsigned char* argk, int B,
                         int C, int N, int M, int K, int centre_is_self, void* stream) {
  GE_REQUIRE(x && y && edge && out && argk && K >= 1 && K <= 255, "mrconv_gather_fwd: bad arguments");
  if (mr_tiled(M, K, centre_is_self)) {
    const size_t lds = mr_tile_lds(M, K);
        GE_MAX_LDS(96 * 1024, (const void*)mr_fwd_tile_kernel<9>, (const void*)mr_fwd_tile_kernel<0>);
    const dim3 grid(ge_cdiv(N, MR_NT), ge_cdiv(C, MR_CT), B);
    static const bool quad_on = !(getenv("GE_MR_QUAD") && atoi(getenv("GE_MR_QUAD")) == 0);
    const size_t lds_q = (size_t)M * MR_PITCH * sizeof(float);
    if (K == 9 && quad_on && (C & 3) == 0 && lds_q <= 96 * 1024) {
            GE_MAX_LDS(96 * 1024, (const void*)mr_fwd_quad_kernel);
      hipLaunchKernelGGL(mr_fwd_quad_kernel, grid, dim3(256), lds_q, (hipStream_t)stream, x, y, edge, out, argk, B, C, N,
                         M);
    } else if (K == 9)
      hipLaunchKernelGGL(mr_fwd_tile_kernel<9>, grid, dim3(256), lds, (hipStream_t)stream, x, y, edge, out, argk, B, C,
                         N, M, K);
    else
      hipLaunchKernelGGL(mr_fwd_tile_kernel<0>, grid, dim3(256), lds, (hipStream_t)stream, x, y, edge, out, argk, B, C,
                         N, M, K);
    GE_CHECK_LAUNCH("mrconv_gather_fwd_tile");
    return GE_OK;
  }
  const size_t lds = (size_t)64 * K * 2 * sizeof(int);
  hipLaunchKernelGGL(mr_fwd_kernel, dim3(ge_cdiv(N, 64), B), dim3(256), lds, (hipStream_t)stream, x, y, edge, out, argk,
                     B, C, N, M, K);
  GE_CHECK_LAUNCH("mrconv_gather_fwd");
  return GE_OK;
}

// floats of workspace ge_mrconv_gather_bwd needs (0 when the general kernels run)
long long ge_mrconv_gather_bwd_workspace(int B, int C, int N, int M, int K, int centre_is_self) {
  if (!mr_tiled(M, K, centre_is_self)) return 0;
  return (long long)mr_bwd_splits(B, C, N) * B * C * M;
}

// dx [B][C][N] and dy [B][C][M] are overwritten; pass dy == dx when y is x (self graph).
int ge_mrconv_gather_bwd(const float* dout, const long long* edge, const unsigned char* argk, float* dx, float* dy,
                         float* workspace, int B, int C, int N, int M, int K, int centre_is_self, void* stream) {
  GE_REQUIRE(dout && edge && argk && dx && dy, "mrconv_gather_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int y_is_x = dy == dx;
  GE_REQUIRE(!y_is_x || M == N, "mrconv_gather_bwd: dy == dx needs M == N");
  if (mr_tiled(M, K, centre_is_self)) {
    GE_REQUIRE(workspace, "mrconv_gather_bwd: workspace required (ge_mrconv_gather_bwd_workspace)");
    const int S = mr_bwd_splits(B, C, N);
    const int cps = ge_cdiv(ge_cdiv(N, MR_NT), S);
    const size_t lds = ((size_t)MR_CTB * M + (size_t)256 * K) * 4;
        GE_MAX_LDS(96 * 1024, (const void*)mr_bwd_tile_kernel);
    const bool direct = S == 1 && !y_is_x;   // a single split's accumulator IS dy
    hipLaunchKernelGGL(mr_bwd_tile_kernel, dim3(ge_cdiv(C, MR_CTB), S, B), dim3(256), lds, st, dout, edge, argk, dx,
                       direct ? dy : workspace, B, C, N, M, K, cps);
    GE_CHECK_LAUNCH("mrconv_bwd_tile");
    if (!direct) {
      const long long total = (long long)B * C * M;
      hipLaunchKernelGGL(mr_bwd_sum_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, st, workspace, dy, total, S,
                         y_is_x);
      GE_CHECK_LAUNCH("mrconv_bwd_sum");
    }
    return GE_OK;
  }
  const size_t lds = (size_t)(M + N) * sizeof(float);
  if (lds <= 64 * 1024) {
    hipLaunchKernelGGL(mr_bwd_kernel, dim3(C, B), dim3(256), lds, st, dout, edge, argk, dx, dy, B, C, N, M, K, y_is_x);
    GE_CHECK_LAUNCH("mrconv_bwd");
    return GE_OK;
  }
  const long long total = (long long)B * C * N;
  if (!y_is_x) ge_init_async(dy, nullptr, (long long)B * C * M, st);
  hipLaunchKernelGGL(mr_bwd_init_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, st, dout, dx, total, N);
  GE_CHECK_LAUNCH("mrconv_bwd_init");
  hipLaunchKernelGGL(mr_bwd_scatter_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, st, dout, edge, argk, dx,
                     dy, B, C, N, M, K);
  GE_CHECK_LAUNCH("mrconv_bwd_scatter");
  return GE_OK;
}

// ---- deterministic backward over inverse neighbour lists (centre-is-self graphs) ----
static size_t mr_det_lds(int K) { return (size_t)MRI_NCH * (MRI_PITCH * 4 + 8) + (size_t)MRI_NCH * K * 4; }
// the one-launch form (mr_bwd_small_kernel): one chunk, list + offsets + counters in LDS next to the staged chunk
static size_t mr_small_lds(int N, int M, int K) {
  return (size_t)MRI_NCH * (MRI_PITCH * 4 + 8) + ((size_t)N * K + (M + 1) + 4 * (size_t)M + 256) * 4;
}
static bool mr_small(int N, int M, int K) {
  // measured (tools/bench_graph_path.py, B 32, C 256): 64 nodes 19.3 us in one launch vs 23.6 us as build + gather,
  // 256 nodes 37.9 vs 32.0 (every workgroup repeating the inversion costs more than the second launch): up to 128 nodes.
  // GE_MR_SMALL = largest node count that takes the one-launch form (0: never)
  static const int upto = getenv("GE_MR_SMALL") ? atoi(getenv("GE_MR_SMALL")) : 128;
  return N <= upto && N <= MRI_NCH && mr_small_lds(N, M, K) <= 64 * 1024;
}
// 1 when ge_mrconv_gather_bwd_small takes this problem: no inverse lists to build first (ge_mr_inv_build)
int ge_mrconv_gather_bwd_small_ok(int N, int M, int K, int centre_is_self) {
  return centre_is_self && K >= 1 && K <= 255 && M >= 1 && N >= 1 && mr_small(N, M, K);
}
// The deterministic backward of graphs of at most ge_mr_inv_chunk() nodes in one launch (list inverted per workgroup, in
// LDS).  dx [B][C][N], dy [B][C][M] overwritten; dy == dx for the self graph.  Same bits as ge_mr_inv_build +
// ge_mrconv_gather_bwd_det (same list order, same walk).
int ge_mrconv_gather_bwd_small(const float* dout, const long long* edge, const unsigned char* argk, float* dx, float* dy,
                               int B, int C, int N, int M, int K, void* stream) {
  GE_REQUIRE(dout && edge && argk && dx && dy, "mrconv_gather_bwd_small: null pointer");
  GE_REQUIRE(ge_mrconv_gather_bwd_small_ok(N, M, K, 1), "mrconv_gather_bwd_small: problem not supported");
  const int self = dy == dx;
  GE_REQUIRE(!self || M == N, "mrconv_gather_bwd_small: dy == dx needs M == N");
  GE_REQUIRE(B <= 65535, "mrconv_gather_bwd_small: B must fit a grid dimension");
    GE_MAX_LDS(64 * 1024, (const void*)mr_bwd_small_kernel<true, 9>, (const void*)mr_bwd_small_kernel<false, 9>, (const void*)mr_bwd_small_kernel<true, 0>, (const void*)mr_bwd_small_kernel<false, 0>);
  const dim3 grid(ge_cdiv(C, MRI_CT), B);
  const size_t lds = mr_small_lds(N, M, K);
  hipStream_t st = (hipStream_t)stream;
#define GE_MR_SMALL_LAUNCH(SELF_, KT_) \
  hipLaunchKernelGGL((mr_bwd_small_kernel<SELF_, KT_>), grid, dim3(256), lds, st, dout, edge, argk, dx, dy, B, C, N, M, K)
  if (self && K == 9) GE_MR_SMALL_LAUNCH(true, 9);
  else if (self) GE_MR_SMALL_LAUNCH(true, 0);
  else if (K == 9) GE_MR_SMALL_LAUNCH(false, 9);
  else GE_MR_SMALL_LAUNCH(false, 0);
#undef GE_MR_SMALL_LAUNCH
  GE_CHECK_LAUNCH("mrconv_gather_bwd_small");
  return GE_OK;
}
// 1 when ge_mr_inv_build / ge_mrconv_gather_bwd_det take this problem (else: ge_mrconv_gather_bwd)
int ge_mrconv_gather_bwd_det_ok(int N, int M, int K, int centre_is_self) {
  return centre_is_self && K >= 1 && K <= 255 && M >= 1 && (size_t)(4 * M + 256) * 4 <= 96 * 1024 &&
         mr_det_lds(K) <= 96 * 1024 && N >= 1;
}
int ge_mr_inv_chunk(void) { return MRI_NCH; }
// chunk splits of the gather (each split owns a run of chunks; partial sums folded in split order): enough workgroups
// for ~8 per CU; the self graph (dy == dx) runs unsplit
static int mr_det_splits(int B, int C, int N, int M, int self) {
  const int nchunks = ge_cdiv(N, MRI_NCH);
  if (self || nchunks == 1) return 1;
  const long long base = (long long)B * ge_cdiv(C, MRI_CT) * ge_cdiv(M, 256);
  int s = (int)((2048 + base - 1) / base);
  if (s > nchunks) s = nchunks;
  return s < 1 ? 1 : s;
}
// floats of workspace ge_mrconv_gather_bwd_det needs (0: none)
long long ge_mrconv_gather_bwd_det_workspace(int B, int C, int N, int M, int K, int y_is_x) {
  const int S = mr_det_splits(B, C, N, M, y_is_x);
  return S > 1 ? (long long)S * B * C * M : 0;
}
// inv uint32 [B][N*K] (chunk j's entries of item b follow those of chunks 0..j-1), off int32 [B][ceil(N/chunk)][M+1]
int ge_mr_inv_build(const long long* edge, unsigned* inv, int* off, int B, int N, int M, int K, void* stream) {
  GE_REQUIRE(edge && inv && off && B > 0, "mr_inv_build: bad arguments");
  GE_REQUIRE(ge_mrconv_gather_bwd_det_ok(N, M, K, 1), "mr_inv_build: problem not supported (ge_mrconv_gather_bwd_det_ok)");
  GE_REQUIRE(B <= 65535, "mr_inv_build: B must fit a grid dimension");
  const size_t lds = (size_t)(4 * M + 256) * sizeof(int);
    GE_MAX_LDS(96 * 1024, (const void*)mr_inv_build_kernel<9>, (const void*)mr_inv_build_kernel<0>);
  if (K == 9)
    hipLaunchKernelGGL(mr_inv_build_kernel<9>, dim3(ge_cdiv(N, MRI_NCH), B), dim3(256), lds, (hipStream_t)stream, edge,
                       inv, off, B, N, M, K);
  else
    hipLaunchKernelGGL(mr_inv_build_kernel<0>, dim3(ge_cdiv(N, MRI_NCH), B), dim3(256), lds, (hipStream_t)stream, edge,
                       inv, off, B, N, M, K);
  GE_CHECK_LAUNCH("mr_inv_build");
  return GE_OK;
}
// dx [B][C][N] and dy [B][C][M] are overwritten; pass dy == dx when y is x (self graph, M == N).  Bit-reproducible.
int ge_mrconv_gather_bwd_det(const float* dout, const unsigned* inv, const int* off, const unsigned char* argk, float* dx,
                             float* dy, float* workspace, int B, int C, int N, int M, int K, void* stream) {
  GE_REQUIRE(dout && inv && off && argk && dx && dy, "mrconv_gather_bwd_det: null pointer");
  GE_REQUIRE(ge_mrconv_gather_bwd_det_ok(N, M, K, 1), "mrconv_gather_bwd_det: problem not supported");
  const int self = dy == dx;
  GE_REQUIRE(!self || M == N, "mrconv_gather_bwd_det: dy == dx needs M == N");
  GE_REQUIRE(B <= 65535 && ge_cdiv(C, MRI_CT) <= 65535, "mrconv_gather_bwd_det: B and C/8 must fit a grid dimension");
    GE_MAX_LDS(96 * 1024, (const void*)mr_bwd_gather_kernel<true>, (const void*)mr_bwd_gather_kernel<false>);
  hipStream_t st = (hipStream_t)stream;
  const int nchunks = ge_cdiv(N, MRI_NCH), mgroups = ge_cdiv(M, 256);
  const int S = mr_det_splits(B, C, N, M, self);
  GE_REQUIRE(S == 1 || workspace, "mrconv_gather_bwd_det: workspace required (ge_mrconv_gather_bwd_det_workspace)");
  const int cps = ge_cdiv(nchunks, S);
  const dim3 grid(mgroups * S, ge_cdiv(C, MRI_CT), B);
  if (self)
    hipLaunchKernelGGL(mr_bwd_gather_kernel<true>, grid, dim3(256), mr_det_lds(K), st, dout, inv, off, argk, dx, dy, B, C,
                       N, M, K, nchunks, mgroups, cps);
  else
    hipLaunchKernelGGL(mr_bwd_gather_kernel<false>, grid, dim3(256), mr_det_lds(K), st, dout, inv, off, argk, dx,
                       S > 1 ? workspace : dy, B, C, N, M, K, nchunks, mgroups, cps);
  GE_CHECK_LAUNCH("mrconv_gather_bwd_det");
  if (S > 1) {
    const long long total = (long long)B * C * M;
    hipLaunchKernelGGL(mr_bwd_sum_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, st, workspace, dy, total, S, 0);
    GE_CHECK_LAUNCH("mrconv_gather_bwd_det_sum");
  }
  return GE_OK;
}

// out [B][C][E] = src [B][C][M] gathered by idx [B][E] (int64 in [0, M)); E = N*K edges.
int ge_edge_gather_fwd(const float* src, const long long* idx, float* out, int B, int C, int M, int E, void* stream) {
  GE_REQUIRE(src && idx && out && B > 0 && C > 0 && M > 0 && E > 0, "edge_gather_fwd: bad arguments");
  GE_REQUIRE(B <= 65535 && C <= 65535, "edge_gather_fwd: B and C must fit a grid dimension");
  int gx = ge_cdiv(E, 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(edge_gather_fwd_kernel, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream, src, idx, out, C, M, E);
  GE_CHECK_LAUNCH("edge_gather_fwd");
  return GE_OK;
}
// dsrc [B][C][M] = scatter-add of dout [B][C][E] by idx (overwrites dsrc).
int ge_edge_gather_bwd(const float* dout, const long long* idx, float* dsrc, int B, int C, int M, int E, void* stream) {
  GE_REQUIRE(dout && idx && dsrc && B > 0 && C > 0 && M > 0 && E > 0, "edge_gather_bwd: bad arguments");
  GE_REQUIRE(B <= 65535 && (size_t)M * sizeof(float) <= 64 * 1024, "edge_gather_bwd: at most 16384 source nodes");
  hipLaunchKernelGGL(edge_gather_bwd_kernel, dim3(C, B), dim3(256), (size_t)M * sizeof(float), (hipStream_t)stream,
                     dout, idx, dsrc, C, M, E);
  GE_CHECK_LAUNCH("edge_gather_bwd");
  return GE_OK;
}

}  // extern "C"
