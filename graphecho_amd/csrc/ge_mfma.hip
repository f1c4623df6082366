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
#include <stdlib.h>
#include <algorithm>

typedef __attribute__((ext_vector_type(16))) float f32x16;

// 1: issue the next chunk's gathers inside the MFMA slots; 0: issue them as one cluster before the MFMAs.
// Measured on MI355X (C2 workload, bs32): clustered 602 frames/s vs interleaved 582 -- at 2-3 co-resident waves
// per SIMD the other waves already cover a wave's load phase, and VALU placed between MFMAs delays their issue.
#ifndef GE_CONV_WAVES_PER_SIMD
#define GE_CONV_WAVES_PER_SIMD 4   // register budget of the conv kernels: 512 / 4 = 128 VGPR+AGPR per lane
#endif
#ifndef GE_CONV_PF2
#define GE_CONV_PF2 1   // 1x1 vector-epilogue kernels: two chunks of operand loads in flight
#endif
#ifndef GE_INTERLEAVE_LOADS
#define GE_INTERLEAVE_LOADS 0
#endif

// Gathers go through buffer loads: the descriptor's hardware range check returns 0 for an out-of-range offset,
// so padding / tile-edge handling needs no branches and all loads of a chunk issue back to back
// (a predicated `ok ? p[i] : 0` makes hipcc emit an exec-mask branch + s_waitcnt vmcnt(0) per load).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define GE_OOB 0xFFFFFFFFu
typedef unsigned int ge_u32x4 __attribute__((ext_vector_type(4)));
// Buffer load of 16 bytes per lane STRAIGHT INTO LDS (buffer_load_dwordx4 ... lds, gfx950): lane l's vector lands at
// lds_addr + 16 l, no VGPR in between.  Issued as inline asm on purpose: hipcc does not know which LDS bytes such a
// load writes and would put s_waitcnt vmcnt(0) in front of every later ds_read; here the kernel counts its own loads
// (lds_dma_wait<N>: at most N of this wave's loads still in flight).  An out-of-range offset (GE_OOB) writes zeros.
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
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// Element offset -> byte offset, or the out-of-range sentinel when !ok.  The empty asm pins the offset
// computation as unconditional straight-line code; without it hipcc turns the select into a branch around
// the index arithmetic and duplicates the load into both arms, each followed by s_waitcnt vmcnt(0).
// the same with a wave-uniform byte offset in the instruction's SGPR field (bounds-checked together with the lane's offset, no
// 32-bit wrap: tools/microbench/soffset_check.hip) -- stepping through K chunks then costs no vector instruction (round 6: on this
// part every VALU instruction is time the SIMD's fp32 MFMAs do not run, profiles/r06_mfma_valu_microbench.txt)
__device__ __forceinline__ float buf_load_s(rsrc_t r, uint32_t byte_off, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, soff, 0));
}
__device__ __forceinline__ uint32_t guard_off(uint32_t elem_off, bool ok) {
  uint32_t off = elem_off * 4u;
  asm volatile("" : "+v"(off));
  return ok ? off : GE_OOB;
}

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
//  * fragments of k-pair kk+2 are read into a second register set before the MFMAs of k-pair kk issue, so the LDS
//    latency hides under the TM*TN 64-cycle MFMAs instead of being exposed before every group;
//  * `side(step)` is invoked after every MFMA group (step = 0 .. KC/2-1): the caller spreads the NEXT chunk's
//    global loads (address arithmetic + buffer_load) over these slots, where they issue for free while the matrix
//    pipe works, instead of clustering them in a VALU-only phase in front of the MFMAs.
// sched_barrier(0) pins this interleave (hipcc otherwise re-clusters loads and sinks the fragment reads).
struct NoSide {
  __device__ __forceinline__ void operator()(int) const {}
};
// SWAPOP: the MFMA gets (b, a) instead of (a, b): acc[i][j] still belongs to (A sub-tile i, B sub-tile j), but inside the
// 32x32 block a lane now holds ONE A-row (lane & 31) and sixteen B-columns (four runs of four consecutive ones) instead of
// one B-column and sixteen A-rows -- the products and their k order are the same, so the values are bit-identical.
template <int TM, int TN, int KC, int SKA, int STA, int SKB, int STB, class Side = NoSide, bool SWAPOP = false>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ sA, const float* __restrict__ sB, int a_off,
                                          int b_off, int lane, f32x16 (&acc)[TM][TN], Side side = Side()) {
  const int li = lane & 31, hi = lane >> 5;
  const float* pa = sA + hi * SKA + (a_off + li) * STA;
  const float* pb = sB + hi * SKB + (b_off + li) * STB;
  float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a0[i] = pa[i * 32 * STA];
#pragma unroll
  for (int j = 0; j < TN; ++j) b0[j] = pb[j * 32 * STB];
#pragma unroll
  for (int kk = 0; kk < KC; kk += 4) {
    if (kk + 2 < KC) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a1[i] = pa[(kk + 2) * SKA + i * 32 * STA];
#pragma unroll
      for (int j = 0; j < TN; ++j) b1[j] = pb[(kk + 2) * SKB + j * 32 * STB];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = SWAPOP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a0[i], acc[i][j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
    side(kk / 2);
    if (kk + 2 < KC) {
      if (kk + 4 < KC) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = pa[(kk + 4) * SKA + i * 32 * STA];
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = pb[(kk + 4) * SKB + j * 32 * STB];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = SWAPOP ? __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a1[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
      side(kk / 2 + 1);
    }
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
  uint32_t wp_bytes, src_bytes;
  FastDiv div_hw, div_w;   // of the output (sub-)grid the N axis enumerates
  // Output sub-grid: n enumerates (b, u, v); the result is written at (u*os + ooy, v*os + oox) of the Hd x Wd map.
  // Used by the strided data-gradient, which is decomposed by output parity so that no MFMA is spent on taps
  // that cannot contribute (os = stride).
  int os, ooy, oox;
  int ntaps, tap_shift, taps[4];   // SUBTAPS: the subset of (kh*KW+kw) taps that contribute to this parity class
  const float* addend;     // optional tensor added to the result (gradient of a skip connection), dst layout
  // optional fused BatchNorm statistics of the result: stats[(g*M+m)][tile_n*WN + wn] = (count, mean, M2) over the
  // columns of one wave's tile, so the BN layer that follows never re-reads the activation to get its moments
  float* stats;
  int stats_parts;
  // split-K (small-N layers whose tiles cannot fill the chip): blockIdx.y = split s handles K chunks
  // [s*split_chunks, (s+1)*split_chunks); split 0 writes dst (with bias / addend), split s >= 1 writes slab s-1 of
  // `ws` (each slab has dst's layout, slab_elems floats); slab_reduce_kernel then adds the slabs to dst.
  float* ws;
  long long slab_elems;
  int splits, split_chunks;
  int dbg;   // tuning only (GE_CONV_DEBUG): bit 0 = skip the epilogue, bit 1 = run a single K chunk
  int full;  // SWAP kernels: every tile is whole and lies inside one image (M % MT == 0, N % NT == 0, plane % NT == 0)
};

template <class T, int KH, int KW, bool SUBTAPS, bool THREE_STAGE = false>
constexpr int conv_waves_per_simd() {
  // the direct-to-LDS 128 x 128 tile holds 48 KB of LDS: three workgroups per CU at most, so 168 registers are free to use
  if (THREE_STAGE && T::MT + T::NT == 256) return 3;
  return (SUBTAPS || (KH * KW > 0 && T::KC % (KH * KW) == 0)) ? GE_CONV_WAVES_PER_SIMD : 3;
}

// EXACT (host guarantees K % KC == 0, TAPFIX layouts only): every staged element's byte offset advances by the same
// amount per chunk, so the loader is one saturating add (the all-ones "out of range" sentinel stays put) + one buffer
// load per element -- the per-chunk validity / address arithmetic (~6 VALU per element) disappears and its
// per-element tables stop occupying registers in the main loop.
// SWAP (host guarantees: output sub-grid = the whole map, plane size a multiple of 4): accumulators transposed inside
// the 32x32 blocks (mma_chunk SWAPOP), so a lane holds four CONSECUTIVE output positions of one channel per register quad
// and the epilogue moves 16-byte vectors (16 stores per 32x32 block-pair instead of 64 scalar ones).
template <class T, int KH, int KW, bool TRANSPOSED, bool SUBTAPS = false, bool EXACT = false, bool SWAP = false, bool LDSD = false>
__global__ __launch_bounds__(T::NTHREADS, (conv_waves_per_simd<T, KH, KW, SUBTAPS, LDSD>())) void conv_gemm_kernel(
    ConvGemmParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = T::KC, NTH = T::NTHREADS;
  constexpr int STEP_A = NTH / MT, EA = KC / STEP_A;
  constexpr int STEP_B = NTH / NT, EB = KC / STEP_B;
  static_assert(NTH % MT == 0 && NTH % NT == 0 && KC % STEP_A == 0 && KC % STEP_B == 0, "tile/thread mismatch");
  // When the chunk length is a multiple of the tap count, a thread's element e always maps to the same
  // (kh, kw) tap and only its channel advances by KC/KHW per chunk: the spatial part of the address and its
  // bounds check are computed once, and the per-chunk work per element is one add + one compare.
  constexpr int KHW_C = KH * KW;
  constexpr bool TAPFIX = SUBTAPS || (KHW_C > 0 && (KC % KHW_C == 0));   // SUBTAPS: ntaps in {1,2,4} divides KC = 16
  constexpr int STAGE = KC * (MT + NT);
  constexpr bool ADDPF = LDSD && TRANSPOSED;      // addend rows prefetched across the last chunk (see the LDSD loop)
  float4 addq[ADDPF ? T::TN * 4 : 1];
  constexpr bool PF2 = EXACT && SWAP && !LDSD && GE_CONV_PF2 && KH * KW == 1 && T::TM * T::TN < 4;   // (the 64 x 64 wave tile has no registers for a second set)
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  // (A persistent tile loop around this body was tried: it cost 20-70 % on the small-K shapes -- the loop-carried
  // state pushed the kernel over its 128-VGPR budget -- and gained nothing on the large ones.)
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tn = lid / p.tiles_m;
  const int m0 = tm * MT, n0 = tn * NT;
  const int nchunks = (p.dbg & 2) ? 1 : (p.K + KC - 1) / KC;
  int c0 = 0, c1 = nchunks;      // K chunks of this workgroup
  const float* o_bias = p.bias;      // epilogue operands: split s >= 1 writes its partial result to a slab, bare
  const float* o_addend = p.addend;
  float* o_dst = p.dst;
  if (p.splits > 1) {
    c0 = blockIdx.y * p.split_chunks;
    c1 = min(nchunks, c0 + p.split_chunks);
    if (blockIdx.y) {
      o_bias = nullptr;
      o_addend = nullptr;
      o_dst = p.ws + (size_t)(blockIdx.y - 1) * p.slab_elems;
    }
  }
  const int kh_n = KH ? KH : p.kh, kw_n = KW ? KW : p.kw;
  const int khw = SUBTAPS ? p.ntaps : kh_n * kw_n;
  const int khw_full = kh_n * kw_n;

  // A operand (packed weights [G][K][M], m fastest): lanes walk m.
  const int ta = tid % MT, ka0 = tid / MT;
  const int ma = m0 + ta;
  const rsrc_t wrs = make_rsrc(p.wp, p.wp_bytes);
  const bool ma_ok = ma < p.M;
  const uint32_t a_base = SUBTAPS ? (uint32_t)g * p.Cs_g * khw_full * p.M + ma : ((uint32_t)g * p.K + ka0) * p.M + ma;
  auto sub_tap = [&](int t) { return t == 0 ? p.taps[0] : (t == 1 ? p.taps[1] : (t == 2 ? p.taps[2] : p.taps[3])); };

  // B operand (patch gather): lanes walk n = (b, y, x).
  const int tb = tid % NT, kb0 = tid / NT;
  const int nb = n0 + tb;
  const bool nb_ok = nb < p.N;
  uint32_t bb, rem, yy, xx;
  fd_divmod(nb_ok ? nb : 0, p.div_hw, bb, rem);
  fd_divmod(rem, p.div_w, yy, xx);
  const uint32_t plane = (uint32_t)p.Hs * p.Ws;
  const rsrc_t srs = make_rsrc(p.src, p.src_bytes);
  const uint32_t b_base = (bb * p.Cs_total + (uint32_t)g * p.Cs_g) * plane;
  const int by = TRANSPOSED ? (int)yy * p.os + p.ooy + p.pad : (int)yy * p.stride - p.pad;
  const int bx = TRANSPOSED ? (int)xx * p.os + p.oox + p.pad : (int)xx * p.stride - p.pad;

  // tap (dy, dx) -> source pixel and validity
  auto tap_src = [&](int dy, int dx, int& iy, int& ix) -> bool {
    bool ok = true;
    if (!TRANSPOSED) {
      iy = by + dy;
      ix = bx + dx;
    } else {
      const int ty = by - dy, tx = bx - dx;
      if (p.stride == 1) {
        iy = ty;
        ix = tx;
      } else if (p.stride == 2) {
        ok = !((ty | tx) & 1);
        iy = ty >> 1;
        ix = tx >> 1;
      } else {
        iy = ty / p.stride;
        ix = tx / p.stride;
        ok = ty >= 0 && tx >= 0 && iy * p.stride == ty && ix * p.stride == tx;
      }
    }
    return ok && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
  };

  uint32_t sp_off[EB];   // TAPFIX: dc*plane + iy*Ws + ix of element e
  int sp_dc[EB];         // TAPFIX: channel offset inside the chunk
  uint32_t sp_ok = 0;    // TAPFIX: bit e = spatially valid
  if (TAPFIX) {
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      const int kl = kb0 + e * STEP_B;
      int dc, t;
      if (SUBTAPS) {
        dc = kl >> p.tap_shift;
        t = sub_tap(kl & (p.ntaps - 1));
      } else {
        dc = KHW_C == 1 ? kl : kl / KHW_C;
        t = kl - dc * KHW_C;
      }
      const int dy = KHW_C == 1 ? 0 : t / KW, dx = KHW_C == 1 ? 0 : t - dy * KW;
      int iy, ix;
      const bool ok = nb_ok && tap_src(dy, dx, iy, ix);
      sp_ok |= (ok ? 1u : 0u) << e;
      sp_dc[e] = dc;
      sp_off[e] = b_base + (uint32_t)dc * plane + (uint32_t)(ok ? iy * p.Ws + ix : 0);
    }
  }

  uint32_t xa_off[EXACT ? EA : 1], xb_off[EXACT ? EB : 1];   // EXACT: running byte offsets (GE_OOB = never valid)
  uint32_t xa_step = 0, xb_step = 0;
  if (EXACT) {
    static_assert(!EXACT || TAPFIX, "EXACT needs the fixed-tap layout");
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      uint32_t row;   // row of the packed operand this element reads in chunk 0, relative to a_base
      if (SUBTAPS) {  // channel * (all taps) + selected tap; ntaps in {1, 2, 4} divides KC, so the tap never changes
        const int k = ka0 + e * STEP_A;
        row = (uint32_t)(k >> p.tap_shift) * khw_full + sub_tap(k & (p.ntaps - 1));
      } else {
        row = (uint32_t)(e * STEP_A);   // a_base already holds ka0
      }
      xa_off[e] = ma_ok ? (a_base + row * p.M) * 4u : GE_OOB;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) xb_off[e] = ((sp_ok >> e) & 1u) ? sp_off[e] * 4u : GE_OOB;
    const uint32_t cpc = SUBTAPS ? (uint32_t)(KC >> p.tap_shift) : (uint32_t)(KC / (KHW_C > 0 ? KHW_C : 1));   // channels per chunk
    xa_step = (SUBTAPS ? cpc * khw_full : (uint32_t)KC) * p.M * 4u;
    xb_step = cpc * plane * 4u;
    if (c0) {   // split-K: start c0 chunks in (the "never valid" sentinel stays put)
      auto adv = [&](uint32_t off, uint32_t step) {
        const unsigned long long v = (unsigned long long)off + (unsigned long long)step * (unsigned)c0;
        return v >= (unsigned long long)GE_OOB ? GE_OOB : (uint32_t)v;
      };
#pragma unroll
      for (int e = 0; e < EA; ++e) xa_off[e] = adv(xa_off[e], xa_step);
#pragma unroll
      for (int e = 0; e < EB; ++e) xb_off[e] = adv(xb_off[e], xb_step);
    }
  }

  float ra[EA], rb[EB];
  // EXACT: the elements' offsets stay where chunk c0 put them; the chunk being fetched contributes a wave-uniform offset through the
  // buffer instruction's SGPR field (rounds 2 - 5: a saturating VALU add per element and chunk)
  uint32_t xs_a = 0, xs_b = 0;
  auto set_chunk = [&](int rel) {
    xs_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)rel * xa_step));
    xs_b = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)rel * xb_step));
  };
  auto load_a = [&](int k0, int e) {
    if (EXACT) {
      ra[e] = buf_load_s(wrs, xa_off[e], xs_a);
      return;
    }
    const int k = k0 + ka0 + e * STEP_A;
    if (SUBTAPS) {   // row of the full packed operand: channel * (all taps) + selected tap (ntaps is 1, 2 or 4)
      const int c = k >> p.tap_shift;
      const uint32_t row = (uint32_t)c * khw_full + sub_tap(k & (p.ntaps - 1));
      ra[e] = buf_load(wrs, guard_off(a_base + row * p.M, ma_ok && k < p.K));
    } else {
      ra[e] = buf_load(wrs, guard_off(a_base + (uint32_t)(k0 + e * STEP_A) * p.M, ma_ok && k < p.K));
    }
  };
  auto load_b = [&](int k0, int e) {
    if (EXACT) {
      rb[e] = buf_load_s(srs, xb_off[e], xs_b);
      return;
    }
    if (TAPFIX) {
      const int c0 = SUBTAPS ? (k0 >> p.tap_shift) : k0 / (KHW_C > 0 ? KHW_C : 1);   // k0 is a multiple of the tap count
      const bool ok = ((sp_ok >> e) & 1u) && (c0 + sp_dc[e] < p.Cs_g);
      rb[e] = buf_load(srs, guard_off(sp_off[e] + (uint32_t)c0 * plane, ok));
    } else {
      const int k = k0 + kb0 + e * STEP_B;
      const int c = k / khw;
      const int t = k - c * khw;
      const int dy = t / kw_n, dx = t - dy * kw_n;
      int iy, ix;
      const bool ok = nb_ok && k < p.K && tap_src(dy, dx, iy, ix);
      rb[e] = buf_load(srs, guard_off(b_base + (uint32_t)c * plane + (uint32_t)(iy * p.Ws + ix), ok));
    }
  };
  auto load = [&](int k0) {      // k0 = first K index of the chunk (a multiple of KC)
    if (EXACT) set_chunk(k0 / KC - c0);
#pragma unroll
    for (int e = 0; e < EA; ++e) load_a(k0, e);
#pragma unroll
    for (int e = 0; e < EB; ++e) load_b(k0, e);
  };
  // elements of the next chunk fetched in MFMA slot `step` (KC/2 slots per chunk)
  constexpr int NSLOT = KC / 2;
  constexpr int PA = (EA + NSLOT - 1) / NSLOT, PB = (EB + NSLOT - 1) / NSLOT;
  auto load_slot = [&](int k0, int step) {
    if (EXACT && step == 0) set_chunk(k0 / KC - c0);
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (step * PA + q < EA) load_a(k0, step * PA + q);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (step * PB + q < EB) load_b(k0, step * PB + q);
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
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;
  if (SWAP && o_bias) {
    // SWAP layout: every accumulator register of a lane belongs to ONE output channel, so the bias is the initial value
    // of the sum and the epilogue has nothing to add (a VALU instruction per result costs as much issue time as the
    // K = 64 layers spend on MFMAs)
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const int m = m0 + a_off + i * 32 + (lane & 31);
      const float bv = m < p.M ? o_bias[g * p.M + m] : 0.f;
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }
  } else {
    acc_zero<T::TM, T::TN>(acc);
  }

  if constexpr (LDSD) {
    // 1x1 / stride 1 / no padding, M and the plane multiples of 4: both operand slices of a chunk are runs of whole rows
    // (A: KC rows of MT floats, B: KC rows of NT floats, no pitch), so a wave fetches 256 consecutive LDS floats per
    // instruction with buffer_load_dwordx4 ... lds.  No staging registers and no ds_write: THREE stages of LDS, the loads
    // of chunk c + 2 are issued before the MFMAs of chunk c (two chunks of HBM latency cover), one barrier per chunk.
    constexpr int NST = 3;
    constexpr int INS_A = KC * MT / 1024, INS_B = KC * NT / 1024;       // wave instructions per chunk and operand
    static_assert(INS_A >= 1 && INS_B >= 1 && NTH == 256, "direct-to-LDS loader: tile too small");
    const ge_u32x4 wrw = make_rsrc_words(p.wp, p.wp_bytes), srw = make_rsrc_words(p.src, p.src_bytes);
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)smem;
    const uint32_t wave_lds = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)wave * 1024u);
    uint32_t la_off[INS_A], lb_off[INS_B];
    {
      const int ra_ = (4 * lane) / MT, ca_ = (4 * lane) % MT;           // row / column of this lane inside a 256-float run
      const int rb_ = (4 * lane) / NT, cb_ = (4 * lane) % NT;
      const bool a_ok = m0 + ca_ < p.M;
      const int n = n0 + cb_;
      const bool b_ok = n < p.N;
      uint32_t ob, orem;
      fd_divmod(b_ok ? n : 0, p.div_hw, ob, orem);
      const uint32_t plane_ = (uint32_t)p.Hs * p.Ws;
#pragma unroll
      for (int i = 0; i < INS_A; ++i) {
        const uint32_t k = (uint32_t)((i * 4 + wave) * (256 / MT) + ra_);
        la_off[i] = a_ok ? ((((uint32_t)g * p.K + k) * p.M) + m0 + ca_) * 4u : GE_OOB;
      }
#pragma unroll
      for (int i = 0; i < INS_B; ++i) {
        const uint32_t k = (uint32_t)((i * 4 + wave) * (256 / NT) + rb_);
        lb_off[i] = b_ok ? ((ob * p.Cs_total + (uint32_t)g * p.Cs_g + k) * plane_ + orem) * 4u : GE_OOB;
      }
    }
    const uint32_t la_step = (uint32_t)KC * p.M * 4u, lb_step = (uint32_t)KC * p.Hs * p.Ws * 4u;
    if (c0) {
      auto adv = [&](uint32_t off, uint32_t step) {
        const unsigned long long v = (unsigned long long)off + (unsigned long long)step * (unsigned)c0;
        return v >= (unsigned long long)GE_OOB ? GE_OOB : (uint32_t)v;
      };
#pragma unroll
      for (int i = 0; i < INS_A; ++i) la_off[i] = adv(la_off[i], la_step);
#pragma unroll
      for (int i = 0; i < INS_B; ++i) lb_off[i] = adv(lb_off[i], lb_step);
    }
    auto issue = [&](int st) {
      const uint32_t base = wave_lds + (uint32_t)st * (STAGE * 4u);
#pragma unroll
      for (int i = 0; i < INS_A; ++i) {
        lds_dma16(wrw, base + (uint32_t)i * 4096u, la_off[i]);
        la_off[i] = __builtin_elementwise_add_sat(la_off[i], la_step);
      }
#pragma unroll
      for (int i = 0; i < INS_B; ++i) {
        lds_dma16(srw, base + (uint32_t)(KC * MT * 4) + (uint32_t)i * 4096u, lb_off[i]);
        lb_off[i] = __builtin_elementwise_add_sat(lb_off[i], lb_step);
      }
    };
    const int n = c1 - c0;
    issue(0);
    if (n > 1) issue(1);
    int st = 0, st2 = 2;       // stage of chunk c, stage chunk c + 2 goes to
    for (int c = 0; c + 1 < n; ++c) {
      lds_dma_wait<INS_A + INS_B>();       // chunk c has landed (chunk c + 1 may still fly)
      __syncthreads();                     // ... for every wave; and everybody is done reading the stage of chunk c - 1
      if (c + 2 < n) issue(st2);
      const float* cur = smem + st * STAGE;
      mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1, NoSide, SWAP>(cur, cur + KC * MT, a_off, b_off, lane, acc);
      st = st == NST - 1 ? 0 : st + 1;
      st2 = st2 == NST - 1 ? 0 : st2 + 1;
    }
    lds_dma_wait<0>();
    __syncthreads();
    if constexpr (ADDPF) {
      // data gradient with a skip-connection addend, whole tiles: the addend rows of the first 32-channel block are
      // requested HERE, so that they arrive under the last chunk's MFMAs instead of one exposed round trip per quad
      if (o_addend && p.full) {
        uint32_t ob, orem;
        fd_divmod(n0, p.div_hw, ob, orem);
        const size_t row0 = ((size_t)ob * p.Cd_total + (size_t)g * p.M + m0 + a_off + (lane & 31)) * ((size_t)p.Hd * p.Wd) + orem +
                            b_off + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) addq[j * 4 + q] = *(const float4*)(o_addend + row0 + j * 32 + 8 * q);
      }
    }
    {
      const float* cur = smem + st * STAGE;
      mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1, NoSide, SWAP>(cur, cur + KC * MT, a_off, b_off, lane, acc);
    }
  } else if constexpr (PF2) {
    // Two chunks of operand loads in flight (1x1 layers stream both operands from HBM: with one chunk in flight the
    // loads have a single chunk of MFMAs -- ~0.9 us -- to come back, less than the loaded HBM latency; measured on a
    // bare MFMA + load loop, tools/microbench/mfma_with_traffic.hip: 96 TFLOP/s at depth 1, 135 at depth 2 for the
    // same 4 KB per wave and chunk).  Register set 1 holds chunk c + 1 at the top of the loop, set 2 receives c + 2.
    // The steady-state body has no branch around a load, so the staging writes wait with vmcnt(one set), not vmcnt(0).
    float ra2[EA], rb2[EB];
    auto load2 = [&](int rel) {      // rel: chunk relative to c0
      set_chunk(rel);
#pragma unroll
      for (int e = 0; e < EA; ++e) ra2[e] = buf_load_s(wrs, xa_off[e], xs_a);
#pragma unroll
      for (int e = 0; e < EB; ++e) rb2[e] = buf_load_s(srs, xb_off[e], xs_b);
    };
    auto stage2 = [&](float* s) {
      float* sA = s;
      float* sB = s + KC * MT;
#pragma unroll
      for (int e = 0; e < EA; ++e) sA[(ka0 + e * STEP_A) * MT + ta] = ra2[e];
#pragma unroll
      for (int e = 0; e < EB; ++e) sB[(kb0 + e * STEP_B) * NT + tb] = rb2[e];
    };
    auto mma = [&](int i) {
      const float* cur = smem + (i & 1) * STAGE;
      mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1, NoSide, SWAP>(cur, cur + KC * MT, a_off, b_off, lane, acc);
    };
    const int n = c1 - c0;
    load(c0 * KC);
    stage(smem);
    __syncthreads();
    if (n > 1) load((c0 + 1) * KC);
    int c = 0;
    for (; c + 3 < n; c += 2) {
      load2(c + 2);                    // chunk c + 2
      mma(c);
      stage(smem + ((c + 1) & 1) * STAGE);
      __syncthreads();
      load((c0 + c + 3) * KC);         // chunk c + 3
      mma(c + 1);
      stage2(smem + (c & 1) * STAGE);
      __syncthreads();
    }
    const int rem = n - c;             // 1, 2 or 3 chunks left; set 1 holds chunk c + 1 when rem >= 2
    if (rem == 3) {
      load2(c + 2);
      mma(c);
      stage(smem + ((c + 1) & 1) * STAGE);
      __syncthreads();
      mma(c + 1);
      stage2(smem + (c & 1) * STAGE);
      __syncthreads();
      mma(c + 2);
    } else if (rem == 2) {
      mma(c);
      stage(smem + ((c + 1) & 1) * STAGE);
      __syncthreads();
      mma(c + 1);
    } else {
      mma(c);
    }
  } else {
  load(c0 * KC);
  stage(smem);
  __syncthreads();
  // steady state: the next chunk's loads ride in the MFMA slots (no branch inside, so hipcc keeps them in flight
  // until the staging writes); the last chunk is peeled.
  for (int c = c0; c + 1 < c1; ++c) {
    const float* cur = smem + ((c - c0) & 1) * STAGE;
    const int knext = (c + 1) * KC;
#if GE_INTERLEAVE_LOADS
    mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1>(cur, cur + KC * MT, a_off, b_off, lane, acc,
                                              [&](int step) { load_slot(knext, step); });
#else
    if (!(p.dbg & 8)) load(knext);
    mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1, NoSide, SWAP>(cur, cur + KC * MT, a_off, b_off, lane, acc);
#endif
    if (!(p.dbg & 16)) {
      stage(smem + ((c + 1 - c0) & 1) * STAGE);
      __syncthreads();
    }
  }
  {
    const float* cur = smem + ((c1 - 1 - c0) & 1) * STAGE;
    mma_chunk<T::TM, T::TN, KC, MT, 1, NT, 1, NoSide, SWAP>(cur, cur + KC * MT, a_off, b_off, lane, acc);
  }
  }

  if ((p.dbg & 1) && acc[0][0][0] != 12345.678f) return;
  if constexpr (SWAP) {
    // lane = channel m (li) of sub-tile i; register quad q of sub-tile j = positions n .. n+3, n = n0 + b_off + 32 j + 8 q + 4 hi
    const int li = lane & 31, hi = lane >> 5;
    const size_t dplane = (size_t)p.Hd * p.Wd;
    if (p.full) {
      // whole tiles inside one image: one division per tile, the 4 * TN stores of a channel row at constant offsets
      uint32_t ob, orem;
      fd_divmod(n0, p.div_hw, ob, orem);
      const size_t row0 = ((size_t)ob * p.Cd_total + (size_t)g * p.M + m0 + a_off + li) * dplane + orem + b_off + 4 * hi;
      const bool plain = !p.stats && !o_addend && !p.relu;
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        float* drow = o_dst + row0 + (size_t)(i * 32) * dplane;
        if (plain) {      // the accumulator registers go out as they are
#pragma unroll
          for (int j = 0; j < T::TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(float4*)(drow + j * 32 + 8 * q) =
                  make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          continue;
        }
        const float* arow = o_addend ? o_addend + row0 + (size_t)(i * 32) * dplane : nullptr;
        float sv = 0.f, qv = 0.f;
        float4 addn[ADDPF ? T::TN * 4 : 1];      // the NEXT block's addend rows, requested before this block's stores
        if constexpr (ADDPF) {
          if (arow && i + 1 < T::TM) {
#pragma unroll
            for (int j = 0; j < T::TN; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) addn[j * 4 + q] = *(const float4*)(arow + (size_t)32 * dplane + j * 32 + 8 * q);
          }
        }
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            if (p.stats) {
              sv += (v.x + v.y) + (v.z + v.w);
              qv += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            if (arow) {
              float4 a4;
              if constexpr (ADDPF)
                a4 = addq[j * 4 + q];
              else
                a4 = *(const float4*)(arow + j * 32 + 8 * q);
              v.x += a4.x;
              v.y += a4.y;
              v.z += a4.z;
              v.w += a4.w;
            }
            if (p.relu) {
              v.x = fmaxf(v.x, 0.f);
              v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f);
              v.w = fmaxf(v.w, 0.f);
            }
            *(float4*)(drow + j * 32 + 8 * q) = v;
          }
        if constexpr (ADDPF) {
#pragma unroll
          for (int e = 0; e < T::TN * 4; ++e) addq[e] = addn[e];
        }
        if (p.stats) {
          sv += __shfl_xor(sv, 32, 64);
          qv += __shfl_xor(qv, 32, 64);
          if (hi == 0) {
            const float cnt = (float)(T::TN * 32);
            const float mean = sv * (1.f / cnt);
            const int m = m0 + a_off + i * 32 + li;
            float* o3 = p.stats + ((size_t)(g * p.M + m) * p.stats_parts + (size_t)tn * T::WN + wn) * 3;
            o3[0] = cnt;
            o3[1] = mean;
            o3[2] = fmaxf(qv - sv * mean, 0.f);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const int m = m0 + a_off + i * 32 + li;
      const bool m_ok = m < p.M;
      const float bias_v = 0.f;      // already in the accumulators
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int j = 0; j < T::TN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + b_off + j * 32 + 8 * q + 4 * hi;
          if (n >= p.N) continue;                       // N and the plane size are multiples of 4: all four or none
          uint32_t ob, orem;
          fd_divmod(n, p.div_hw, ob, orem);
          float4 v = make_float4(acc[i][j][4 * q] + bias_v, acc[i][j][4 * q + 1] + bias_v, acc[i][j][4 * q + 2] + bias_v,
                                 acc[i][j][4 * q + 3] + bias_v);
          if (p.stats) {
            sv += (v.x + v.y) + (v.z + v.w);
            qv += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          }
          if (m_ok) {
            const size_t o = ((size_t)ob * p.Cd_total + (size_t)g * p.M + m) * dplane + orem;
            if (o_addend) {
              const float4 a4 = *(const float4*)(o_addend + o);
              v.x += a4.x;
              v.y += a4.y;
              v.z += a4.z;
              v.w += a4.w;
            }
            if (p.relu) {
              v.x = fmaxf(v.x, 0.f);
              v.y = fmaxf(v.y, 0.f);
              v.z = fmaxf(v.z, 0.f);
              v.w = fmaxf(v.w, 0.f);
            }
            *(float4*)(o_dst + o) = v;
          }
        }
      }
      if (p.stats) {
        // moments of channel m over this wave's TN*32 columns: the two half-waves hold the two halves of them
        sv += __shfl_xor(sv, 32, 64);
        qv += __shfl_xor(qv, 32, 64);
        const int ncol0 = n0 + b_off;
        float cnt = 0.f;
#pragma unroll
        for (int j = 0; j < T::TN; ++j) cnt += (float)max(0, min(32, p.N - (ncol0 + j * 32)));
        if (hi == 0 && m_ok) {
          const float mean = cnt > 0.f ? sv / cnt : 0.f;
          float* o3 = p.stats + ((size_t)(g * p.M + m) * p.stats_parts + (size_t)tn * T::WN + wn) * 3;
          o3[0] = cnt;
          o3[1] = mean;
          o3[2] = fmaxf(qv - sv * mean, 0.f);
        }
      }
    }
    return;
  }
  // Epilogue: lanes walk n (contiguous x within an image row) -> coalesced 128 B segments.
  const int li = lane & 31, hi = lane >> 5;
  const size_t dplane = (size_t)p.Hd * p.Wd;
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + a_off + i * 32 + acc_row(r, hi);
      bias_r[r] = (o_bias && m < p.M) ? o_bias[g * p.M + m] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < T::TN; ++j) {
      const int n = n0 + b_off + j * 32 + li;
      if (n >= p.N) continue;
      uint32_t ob, orem, ou, ov;
      fd_divmod(n, p.div_hw, ob, orem);
      if (p.os != 1 || p.ooy | p.oox) {
        fd_divmod(orem, p.div_w, ou, ov);
        orem = (ou * p.os + p.ooy) * p.Wd + ov * p.os + p.oox;
      }
      const size_t dbase = ((size_t)ob * p.Cd_total + (size_t)g * p.M) * dplane + orem;
      float* dst = o_dst + dbase;
      if (o_addend) {   // gradient of a skip connection: issue the 16 loads before the dependent stores
        const float* add = o_addend + dbase;
        float addv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + a_off + i * 32 + acc_row(r, hi);
          addv[r] = add[(size_t)(m < p.M ? m : 0) * dplane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + a_off + i * 32 + acc_row(r, hi);
          if (m < p.M) dst[(size_t)m * dplane] = acc[i][j][r] + bias_r[r] + addv[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + a_off + i * 32 + acc_row(r, hi);
          if (m < p.M) {
            float v = acc[i][j][r] + bias_r[r];
            if (p.relu) v = fmaxf(v, 0.f);
            dst[(size_t)m * dplane] = v;
          }
        }
      }
    }
    if (p.stats) {
      // Per-row moments over this wave's TN*32 columns.  The 32 lanes of a half-wave hold the same 16 rows, so the
      // 32 per-lane partials (16 sums + 16 sums of squares) are reduced with a reduce-scatter butterfly: 31
      // shuffles instead of 160, after which lane li owns fully reduced value li.
      const int ncol0 = n0 + b_off;
      float cnt = 0.f;
#pragma unroll
      for (int j = 0; j < T::TN; ++j) cnt += (float)max(0, min(32, p.N - (ncol0 + j * 32)));
      float v[32];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int j = 0; j < T::TN; ++j) {
          const bool ok = ncol0 + j * 32 + li < p.N;
          const float t = ok ? acc[i][j][r] + bias_r[r] : 0.f;
          sv += t;
          qv += t * t;
        }
        v[r] = sv;
        v[16 + r] = qv;
      }
#pragma unroll
      for (int h = 16; h > 0; h >>= 1) {   // keep the upper half of the live values if the lane's bit h is set
        const bool up = (li & h) != 0;
#pragma unroll
        for (int k = 0; k < h; ++k) {
          const float send = up ? v[k] : v[k + h];
          const float keep = up ? v[k + h] : v[k];
          v[k] = keep + __shfl_xor(send, h, 64);
        }
      }
      // lane li < 16: sum of row r = li; lane li >= 16: sum of squares of row r = li - 16
      const float qsum = __shfl_down(v[0], 16, 64);
      if (li < 16) {
        const int m = m0 + a_off + i * 32 + acc_row(li, hi);
        if (m < p.M) {   // an all-padding wave tile still owns its slot: it writes an empty triple
          const float mean = cnt > 0.f ? v[0] / cnt : 0.f;
          float* o3 = p.stats + ((size_t)(g * p.M + m) * p.stats_parts + (size_t)tn * T::WN + wn) * 3;
          o3[0] = cnt;
          o3[1] = mean;
          o3[2] = fmaxf(qsum - v[0] * mean, 0.f);
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
  uint32_t dy_bytes, x_bytes;
  FastDiv div_hw, div_w;  // Ho*Wo, Wo
  int dbg;                // tuning only (GE_CONV_DEBUG): bit 3 = skip the global loads after the first chunk
  // Bias gradient in the same pass (conv_wgrad_kernel only): db[m] = sum_n dY[m][n] is the row sum of the operand tile
  // this kernel stages anyway; the workgroups of the first column tile add up the rows of every staged chunk (one
  // register, 32 conflict-free LDS reads per thread and chunk) and leave them behind the split's weight slab, where
  // the slab reduce folds both.  Replaces channel_sum_direct_kernel: a full read of dY on the critical stream per layer
  // (100 us for the 1024-channel FFN layers of the Graphers at 64 x 64).
  int bias;                   // 1: write row sums to slab + sp * slab_stride + bias_off + g*M + m
  long long slab_stride;      // floats between the slabs of consecutive splits (G*M*J, + Cout when bias)
  long long bias_off;         // G*M*J
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
  // grid.x = K-splits x output tiles, split-major: after the XCD remap the tiles of one split -- which all stream the same
  // K-range of dY and X -- sit on one XCD and share its L2 (spread over the eight XCDs each of them fetched that range
  // from HBM on its own: 1.4 GB per 3x3 launch against 0.27 GB of operands)
  const int g = blockIdx.z;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_m * p.tiles_j;
  const int sp = lid / ntile, tl = lid - sp * ntile;
  const int tm = tl % p.tiles_m, tj = tl / p.tiles_m;
  const int m0 = tm * MT, j0 = tj * NT;
  const int kh_n = KH ? KH : p.kh, kw_n = KW ? KW : p.kw;
  const int khw = kh_n * kw_n;

  const int kl = tid % KC, t0 = tid / KC;
  const int kbeg = sp * p.klen;
  const int kend = min(kbeg + p.klen, p.Ktot);
  const uint32_t oplane = (uint32_t)p.Ho * p.Wo, iplane = (uint32_t)p.Hi * p.Wi;
  const rsrc_t drs = make_rsrc(p.dy, p.dy_bytes);
  const rsrc_t xrs = make_rsrc(p.x, p.x_bytes);

  // The (ci, kh, kw) of a thread's B elements do not depend on the chunk: decode them once.
  int w_coff[EB];        // ci*iplane + kh*Wi + kw
  int w_tap[EB];         // kh | kw << 8
  uint32_t w_jok = 0;    // bit e: column j is inside J
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
    w_coff[e] = c * (int)iplane + dyy * p.Wi + dxx;
    w_tap[e] = dyy | (dxx << 8);
    w_jok |= (j < p.J ? 1u : 0u) << e;
  }
  uint32_t w_mok = 0;    // bit e: row m is inside M
#pragma unroll
  for (int e = 0; e < EA; ++e) w_mok |= (m0 + t0 + e * STEP < p.M ? 1u : 0u) << e;

  float ra[EA], rb[EB];
  // Lean loader for 1x1 and 3x3 filters: everything that does not depend on the chunk is folded into per-element
  // constants (row byte offset or the all-ones sentinel; channel/tap byte offset; the element's bit in a per-chunk
  // tap-validity mask), so a chunk costs one position decode + one 9-bit mask for the thread, then one saturating add
  // per dY element and and/compare/add/select per X element (the general path below re-derives the source pixel and
  // its bounds per element: ~9 VALU each, which measured 16-30 % of the kernel time).
  constexpr bool LEAN = (KH * KW == 1) || (KH == 3 && KW == 3);
  // Rows m >= M and columns j >= J need no guard of their own: they only feed accumulator rows / columns the
  // epilogue never stores (every output element depends on its own operand row and column alone).
  uint32_t lb_col[LEAN ? EB : 1], lb_bit[(LEAN && KH * KW > 1) ? EB : 1];
  if (LEAN) {
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      lb_col[e] = (uint32_t)w_coff[e] * 4u;
      if (KH * KW > 1) lb_bit[e] = 1u << ((w_tap[e] & 255) * KW + (w_tap[e] >> 8));
    }
  }
  const uint32_t la_rowstride = (uint32_t)STEP * oplane * 4u;
  const uint32_t la_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)la_rowstride);      // wave-uniform
  uint32_t la_base = GE_OOB, lb_base = 0, lb_mask = 0;
  // per-chunk position decode (shared by all elements of the chunk), then one element per call
  bool n_ok = false;
  uint32_t dy_base = 0;
  int x_base = 0, by = 0, bx = 0;
  auto chunk_pos = [&](int k0) {
    const int n = k0 + kl;
    n_ok = n < kend;
    uint32_t bb, rem, oy, ox;
    fd_divmod(n_ok ? n : 0, p.div_hw, bb, rem);
    fd_divmod(rem, p.div_w, oy, ox);
    dy_base = (bb * p.Co_total + (uint32_t)g * p.M + m0 + t0) * oplane + rem;
    by = (int)oy * p.stride - p.pad;
    bx = (int)ox * p.stride - p.pad;
    x_base = (int)((bb * p.Ci_total + (uint32_t)g * p.Ci_g) * iplane) + by * p.Wi + bx;
    if (LEAN) {
      la_base = n_ok ? dy_base * 4u : GE_OOB;
      lb_base = (uint32_t)x_base * 4u;          // may be "negative" (padding): only used when the tap is valid
      uint32_t rows = 0, cols = 0;
#pragma unroll
      for (int t = 0; t < (KH > 0 ? KH : 1); ++t) {
        rows |= ((unsigned)(by + t) < (unsigned)p.Hi ? 1u : 0u) << t;
        cols |= ((unsigned)(bx + t) < (unsigned)p.Wi ? 1u : 0u) << t;
      }
      uint32_t mask = 0;
#pragma unroll
      for (int t = 0; t < (KH > 0 ? KH : 1); ++t) mask |= ((rows >> t) & 1u) ? (cols << (t * KW)) : 0u;
      lb_mask = n_ok ? mask : 0u;
    }
  };
  auto load_a = [&](int e) {
    ra[e] = buf_load(drs, guard_off(dy_base + (uint32_t)(e * STEP) * oplane, n_ok && ((w_mok >> e) & 1u)));
  };
  auto load_b = [&](int e) {
    const int iy = by + (w_tap[e] & 255), ix = bx + (w_tap[e] >> 8);
    const bool ok = n_ok && ((w_jok >> e) & 1u) && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    rb[e] = buf_load(xrs, guard_off((uint32_t)(x_base + w_coff[e]), ok));
  };
  // 1x1 / stride 1 / pad 0 with Ho * Wo a multiple of the chunk (every 1x1 layer of the FPN): a chunk of KC positions lies inside
  // ONE image, so the position decode is wave-uniform -- scalar unit -- and every element's address is (a per-lane constant) + (the
  // chunk's offset + the element's row step, in the buffer instruction's SGPR field).  The per-lane decode below costs 7 v_mul_lo_u32
  // (quarter rate) + ~25 more VALU instructions per chunk and thread: 23 % of the 64 x 64 tile's MFMA time on a part whose fp32 MFMAs
  // and VALU share the datapath (round 6, profiles/r06_mfma_valu_microbench.txt).
  const bool uni = KH * KW == 1 && p.stride == 1 && p.pad == 0 && oplane % KC == 0 && iplane == oplane;
  const uint32_t ua_lane = (((uint32_t)g * p.M + m0 + t0) * oplane + kl) * 4u;
  const uint32_t ub_lane = (((uint32_t)g * p.Ci_g + j0 + t0) * iplane + kl) * 4u;
  const uint32_t ub_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)STEP * iplane * 4u));
  auto load = [&](int k0) {
    if (uni) {
      uint32_t bb, rem;
      fd_divmod((uint32_t)__builtin_amdgcn_readfirstlane(k0), p.div_hw, bb, rem);      // wave-uniform
      const bool live = k0 < kend;      // (klen and Ktot are multiples of KC: a chunk is inside the split or past it as a whole)
      const uint32_t ca = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((bb * p.Co_total * oplane + rem) * 4u)) : GE_OOB;
      const uint32_t cb = live ? (uint32_t)__builtin_amdgcn_readfirstlane((int)((bb * p.Ci_total * iplane + rem) * 4u)) : GE_OOB;
#pragma unroll
      for (int e = 0; e < EA; ++e) ra[e] = buf_load_s(drs, ua_lane, live ? ca + (uint32_t)e * la_rows : GE_OOB);
#pragma unroll
      for (int e = 0; e < EB; ++e) rb[e] = buf_load_s(xrs, ub_lane, live ? cb + (uint32_t)e * ub_rows : GE_OOB);
      return;
    }
    chunk_pos(k0);
    if (LEAN) {
      // la_base: all-ones when n is past the split (out of range whatever the SGPR offset); row e rides in the SGPR field
#pragma unroll
      for (int e = 0; e < EA; ++e) ra[e] = buf_load_s(drs, la_base, (uint32_t)e * la_rows);
      if (KH * KW == 1) {
        const uint32_t base = lb_mask ? lb_base : GE_OOB;
#pragma unroll
        for (int e = 0; e < EB; ++e) rb[e] = buf_load(xrs, __builtin_elementwise_add_sat(lb_col[e], base));
      } else {
#pragma unroll
        for (int e = 0; e < EB; ++e) {
          uint32_t o = lb_base + lb_col[e];
          asm volatile("" : "+v"(o));
          rb[e] = buf_load(xrs, (lb_mask & lb_bit[e]) ? o : GE_OOB);
        }
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < EA; ++e) load_a(e);
#pragma unroll
    for (int e = 0; e < EB; ++e) load_b(e);
  };
  constexpr int NSLOT = KC / 2;
  constexpr int PA = (EA + NSLOT - 1) / NSLOT, PB = (EB + NSLOT - 1) / NSLOT;
  auto load_slot = [&](int k0, int step) {
    if (step == 0) chunk_pos(k0);
#pragma unroll
    for (int q = 0; q < PA; ++q)
      if (step * PA + q < EA) load_a(step * PA + q);
#pragma unroll
    for (int q = 0; q < PB; ++q)
      if (step * PB + q < EB) load_b(step * PB + q);
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
  const bool do_bias = p.bias && tj == 0 && tid < MT;      // whole waves (MT is a multiple of 64)
  float bsum = 0.f;
  auto rowsum = [&]() {
    if (do_bias) {
      const float* r = dsmem + tid * LDK;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < KC; ++k) s += r[k];                // positions past the split were staged as zeros
      bsum += s;
    }
  };

  // Single LDS stage + register prefetch: the next chunk's global loads are in flight under this chunk's MFMAs;
  // halving the LDS footprint (34 KB) lets three workgroups share a CU, which hides more latency than a second
  // LDS stage does.
  const int nchunks = (kend - kbeg + KC - 1) / KC;
  if (nchunks > 0) {
    load(kbeg);
    stage(dsmem);
    __syncthreads();
    for (int c = 0; c + 1 < nchunks; ++c) {
      const int knext = kbeg + (c + 1) * KC;
#if GE_INTERLEAVE_LOADS
      rowsum();      // the fused bias gradient sums EVERY staged chunk, in this branch too
      mma_chunk<T::TM, T::TN, KC, 1, LDK, 1, LDK>(dsmem, dsmem + MT * LDK, a_off, b_off, lane, acc,
                                                  [&](int step) { load_slot(knext, step); });
#else
      if (!(p.dbg & 8)) load(knext);
      rowsum();
      mma_chunk<T::TM, T::TN, KC, 1, LDK, 1, LDK>(dsmem, dsmem + MT * LDK, a_off, b_off, lane, acc);
#endif
      __syncthreads();
      stage(dsmem);
      __syncthreads();
    }
    rowsum();
    mma_chunk<T::TM, T::TN, KC, 1, LDK, 1, LDK>(dsmem, dsmem + MT * LDK, a_off, b_off, lane, acc);
  }

  const int li = lane & 31, hi = lane >> 5;
  float* slab = p.slab + (size_t)sp * p.slab_stride + (size_t)g * p.M * p.J;
  if (do_bias && m0 + tid < p.M) p.slab[(size_t)sp * p.slab_stride + p.bias_off + (size_t)g * p.M + m0 + tid] = bsum;
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

// =========================================================================================
// conv_wgrad3x3: the 3x3 / stride 1 / pad 1 weight gradient (22 of config 2's 25 3x3 layers, the largest kernel of
// the step) with the X operand staged as a halo'd PATCH.  A chunk is KC = 32 consecutive output positions of one
// image -- one run of 32 pixels of a row (Wo % 32 == 0) or 32 / Wo whole rows (Wo = 16, 8) -- and the B tile
// B[j = (ci, kh, kw)][k] = X[ci][oy + kh - 1][ox + kw - 1] is a shifted view of the (rows + 2) x (WC + 2) patch of each
// of the tile's <= 16 channels: 1 632 staged elements per chunk instead of 4 096 (every X element was fetched once per
// tap it serves: nine times), 6.4 loads + and / select / add per thread instead of 16, and the LDS image shrinks from
// 16.9 KB to 6.7 KB.  Patch pitches: row LDC = 35 (== 3 mod 32), channel CS == 9 mod 32, so the word address of column
// j = 9 ci + 3 kh + kw is congruent to j: the 32 lanes of a fragment read (consecutive j) hit 32 different banks.
// =========================================================================================
#ifndef GE_WGRAD3_WPS
#define GE_WGRAD3_WPS 4
#endif
template <class T, int WC, bool DB>
__global__ __launch_bounds__(T::NTHREADS, GE_WGRAD3_WPS) void conv_wgrad3x3_kernel(WgradParams p) {
  constexpr int MT = T::MT, NT = T::NT, KC = T::KC, NTH = T::NTHREADS;
  static_assert(KC == 32 && KC % WC == 0, "chunk = 32 positions");
  constexpr int STEP = NTH / KC, EA = MT / STEP;
  constexpr int LDK = KC + 1;
  constexpr int ROWS = KC / WC + 2, PW = WC + 2, LDC = 35;
  constexpr int CS = ((ROWS * LDC - 9 + 31) / 32) * 32 + 9;
  constexpr int NCH = NT / 9 + 2;                       // channels NT consecutive columns can touch
  constexpr int PE = NCH * ROWS * PW, EB = (PE + NTH - 1) / NTH;
  constexpr int STAGE = MT * LDK + NCH * CS;           // floats per LDS stage: [MT][LDK] dY tile (k-fast), [NCH][CS] patches
  extern __shared__ __attribute__((aligned(16))) float dsmem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.z;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_m * p.tiles_j;
  const int sp = lid / ntile, tl = lid - sp * ntile;
  const int tm = tl % p.tiles_m, tj = tl / p.tiles_m;
  const int m0 = tm * MT, j0 = tj * NT;
  const int c0 = j0 / 9;                                // first channel of the tile
  const int kl = tid % KC, t0 = tid / KC;
  const int kbeg = sp * p.klen;
  const int kend = min(kbeg + p.klen, p.Ktot);
  const uint32_t oplane = (uint32_t)p.Ho * p.Wo;       // == input plane: stride 1, pad 1, 3x3
  const rsrc_t drs = make_rsrc(p.dy, p.dy_bytes);
  const rsrc_t xrs = make_rsrc(p.x, p.x_bytes);

  // patch elements of this thread: LDS word, byte offset relative to the chunk's base pixel, validity bit
  int pl_off[EB];
  uint32_t pg_off[EB], pbit[EB];
#pragma unroll
  for (int i = 0; i < EB; ++i) {
    const int e = tid + i * NTH;
    const int c = e / (ROWS * PW), rem = e - c * (ROWS * PW);
    const int r = rem / PW, col = rem - r * PW;
    pl_off[i] = c * CS + r * LDC + col;
    pg_off[i] = (uint32_t)((c * (int)oplane + (r - 1) * p.Wi + (col - 1)) * 4);
    const int cls = col == 0 ? 0 : (col == PW - 1 ? 2 : 1);
    pbit[i] = (e < PE && c0 + c < p.Ci_g) ? (1u << (r * 3 + cls)) : 0u;
    if (e >= PE) pl_off[i] = 0;                        // never written: the store below is guarded by e < PE
  }
  const uint32_t la_rowstride = (uint32_t)STEP * oplane * 4u;
  const uint32_t la_rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)la_rowstride);      // wave-uniform

  float ra[EA], rb[EB];
  auto load = [&](int k0) {
    // chunk position (wave-uniform): image bb, first pixel (oy0, ox0)
    uint32_t bb, rem, oy0, ox0;
    fd_divmod((uint32_t)k0, p.div_hw, bb, rem);
    fd_divmod(rem, p.div_w, oy0, ox0);
    const uint32_t off = ((bb * p.Co_total + (uint32_t)g * p.M + m0 + t0) * oplane + rem + kl) * 4u;
#pragma unroll
    for (int e = 0; e < EA; ++e) ra[e] = buf_load_s(drs, off, (uint32_t)e * la_rows);      // row e: SGPR offset field
    uint32_t V = 0;                                     // bit r*3 + cls: patch row r / column class cls inside the image
    const uint32_t cols = (ox0 > 0 ? 1u : 0u) | 2u | ((int)ox0 + WC < p.Wi ? 4u : 0u);
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
      V |= ((unsigned)((int)oy0 + r - 1) < (unsigned)p.Hi ? cols : 0u) << (r * 3);
    const uint32_t xbase = ((bb * p.Ci_total + (uint32_t)g * p.Ci_g + c0) * oplane + rem) * 4u;
#pragma unroll
    for (int i = 0; i < EB; ++i) {
      uint32_t o = xbase + pg_off[i];
      asm volatile("" : "+v"(o));
      rb[i] = buf_load(xrs, (V & pbit[i]) ? o : GE_OOB);
    }
  };
  auto stage = [&](float* s) {
    float* sA = s;
    float* sP = s + MT * LDK;
#pragma unroll
    for (int e = 0; e < EA; ++e) sA[(t0 + e * STEP) * LDK + kl] = ra[e];
#pragma unroll
    for (int i = 0; i < EB; ++i)
      if (tid + i * NTH < PE) sP[pl_off[i]] = rb[i];
  };

  f32x16 acc[T::TM][T::TN];
  acc_zero<T::TM, T::TN>(acc);
  const int wm = wave % T::WM, wn = wave / T::WM;
  const int a_off = wm * T::TM * 32, b_off = wn * T::TN * 32;
  const int li = lane & 31, hi = lane >> 5;
  // fragment bases (words inside a stage): A row (a_off + i*32 + li), k = kk + hi;  B column j -> patch word of
  // (ci, kh, kw), + hi
  const int pa_w = hi + (a_off + li) * LDK;
  int pb_w[T::TN];
#pragma unroll
  for (int jn = 0; jn < T::TN; ++jn) {
    int j = j0 + b_off + jn * 32 + li;
    j = j < p.J ? j : p.J - 1;                          // columns past J are never stored
    const int ci = j / 9, t = j - ci * 9, dy = t / 3, dx = t - dy * 3;
    pb_w[jn] = MT * LDK + (ci - c0) * CS + dy * LDC + dx + hi;
  }
  auto koff = [](int kk) { return (kk / WC) * LDC + (kk % WC); };
  auto mma = [&](const float* s) {
    const float* pa = s + pa_w;
    const float* pb[T::TN];
#pragma unroll
    for (int jn = 0; jn < T::TN; ++jn) pb[jn] = s + pb_w[jn];
    float a0[T::TM], b0[T::TN], a1[T::TM], b1[T::TN];
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a0[i] = pa[i * 32 * LDK];
#pragma unroll
    for (int j = 0; j < T::TN; ++j) b0[j] = pb[j][0];
#pragma unroll
    for (int kk = 0; kk < KC; kk += 4) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i) a1[i] = pa[(kk + 2) + i * 32 * LDK];
#pragma unroll
      for (int j = 0; j < T::TN; ++j) b1[j] = pb[j][koff(kk + 2)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
      if (kk + 4 < KC) {
#pragma unroll
        for (int i = 0; i < T::TM; ++i) a0[i] = pa[(kk + 4) + i * 32 * LDK];
#pragma unroll
        for (int j = 0; j < T::TN; ++j) b0[j] = pb[j][koff(kk + 4)];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    }
  };

  // Register prefetch of the next chunk under this chunk's MFMAs; DB: two LDS stages (one barrier per chunk, the
  // staging writes of the fast waves overlap the MFMAs of the slow ones), else one stage and two barriers.
  const int nchunks = (kend - kbeg) / KC;               // the host guarantees whole chunks inside one image
  if (nchunks > 0) {
    load(kbeg);
    stage(dsmem);
    __syncthreads();
    for (int c = 0; c + 1 < nchunks; ++c) {
      load(kbeg + (c + 1) * KC);
      if (DB) {
        mma(dsmem + (c & 1) * STAGE);
        stage(dsmem + ((c + 1) & 1) * STAGE);
        __syncthreads();
      } else {
        mma(dsmem);
        __syncthreads();
        stage(dsmem);
        __syncthreads();
      }
    }
    mma(dsmem + (DB ? ((nchunks - 1) & 1) * STAGE : 0));
  }

  float* slab = p.slab + (size_t)sp * p.slab_stride + (size_t)g * p.M * p.J;
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

// out2 / n1: elements i >= n1 of a slab (the bias row sums behind the weights) go to out2[i - n1]
__global__ void slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long long n, int splits,
                                   int accumulate, float* __restrict__ out2 = nullptr, long long n1 = 0) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float* dst = (out2 && i >= n1) ? out2 + (i - n1) : out + i;
    float s0 = accumulate ? *dst : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {   // four independent loads in flight per thread
      s0 += slab[(size_t)k * n + i];
      s1 += slab[(size_t)(k + 1) * n + i];
      s2 += slab[(size_t)(k + 2) * n + i];
      s3 += slab[(size_t)(k + 3) * n + i];
    }
    for (; k < splits; ++k) s0 += slab[(size_t)k * n + i];
    *dst = (s0 + s1) + (s2 + s3);
  }
}

struct SlabBatch {
  static constexpr int MAX = 16;
  const float* slab[MAX];
  float* out[MAX];
  long long n[MAX], stride[MAX];      // elements reduced per slab, floats between consecutive slabs (>= n)
  int splits[MAX], accumulate[MAX];
};
__global__ __launch_bounds__(256) void slab_reduce_batched_kernel(SlabBatch b) {
  const int e = blockIdx.y;
  const float* __restrict__ slab = b.slab[e];
  float* __restrict__ out = b.out[e];
  const long long n = b.n[e], sn = b.stride[e];
  const int splits = b.splits[e], accumulate = b.accumulate[e];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float s0 = accumulate ? out[i] : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // same association as slab_reduce_kernel
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      s0 += slab[(size_t)k * sn + i];
      s1 += slab[(size_t)(k + 1) * sn + i];
      s2 += slab[(size_t)(k + 2) * sn + i];
      s3 += slab[(size_t)(k + 3) * sn + i];
    }
    for (; k < splits; ++k) s0 += slab[(size_t)k * sn + i];
    out[i] = (s0 + s1) + (s2 + s3);
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

// Every conv weight of a model in one launch: table rows = (src_off, dst_off, total, G, Co_g, Ci_g, khw, transposed)
// as int64, offsets in floats relative to `w` (the model's flat parameter buffer) and `out`.  grid.y = table row.
// Both layouts are transposes, staged through LDS so that reads and writes are both coalesced (the direct form read
// the forward layout with a stride of Ci_g*kh*kw floats: ~16x the bytes on the HBM side):
//   forward : per group, W[Co_g][K] -> out[K][Co_g] with K = Ci_g*khw, in 64x64 tiles;
//   dgrad   : per output channel, [Ci_g][khw] -> [khw][Ci_g], one channel (<= 4160 floats) per iteration.
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                   const long long* __restrict__ table) {
  __shared__ float buf[64 * 65];
  const long long* row = table + (size_t)blockIdx.y * 8;
  const float* src = w + row[0];
  float* dst = out + row[1];
  const unsigned total = (unsigned)row[2];
  const unsigned G = (unsigned)row[3], Co_g = (unsigned)row[4], Ci_g = (unsigned)row[5], khw = (unsigned)row[6];
  const bool transposed = row[7] != 0;
  const unsigned tid = threadIdx.x;
  if (!transposed) {
    const unsigned K = Ci_g * khw, tiles_k = (K + 63) / 64, tiles_m = (Co_g + 63) / 64;
    const unsigned tx = tid & 63, ty = tid >> 6;
    for (unsigned tile = blockIdx.x; tile < G * tiles_k * tiles_m; tile += gridDim.x) {
      const unsigned g = tile / (tiles_k * tiles_m), r = tile - g * tiles_k * tiles_m;
      const unsigned m0 = (r / tiles_k) * 64, k0 = (r % tiles_k) * 64;
      const float* sg = src + (size_t)g * Co_g * K;
      float* dg = dst + (size_t)g * K * Co_g;
#pragma unroll 4
      for (unsigned rr = 0; rr < 16; ++rr) {
        const unsigned m = m0 + ty + 4 * rr, k = k0 + tx;
        buf[(ty + 4 * rr) * 65 + tx] = (m < Co_g && k < K) ? sg[(size_t)m * K + k] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (unsigned rr = 0; rr < 16; ++rr) {
        const unsigned k = k0 + ty + 4 * rr, m = m0 + tx;
        if (k < K && m < Co_g) dg[(size_t)k * Co_g + m] = buf[tx * 65 + ty + 4 * rr];
      }
      __syncthreads();
    }
    return;
  }
  const unsigned n = Ci_g * khw;
  if (n <= 64 * 65) {
    for (unsigned ch = blockIdx.x; ch < G * Co_g; ch += gridDim.x) {
      const size_t base = (size_t)ch * n;
      for (unsigned e = tid; e < n; e += 256) buf[e] = src[base + e];
      __syncthreads();
      for (unsigned e = tid; e < n; e += 256) {
        const unsigned t = e / Ci_g, ci = e - t * Ci_g;
        dst[base + e] = buf[ci * khw + t];
      }
      __syncthreads();
    }
    return;
  }
  for (unsigned i = blockIdx.x * blockDim.x + tid; i < total; i += gridDim.x * blockDim.x) {
    // i = ((g*Co_g + co)*khw + t)*Ci_g + ci
    unsigned rest = i;
    const unsigned ci = rest % Ci_g;
    rest /= Ci_g;
    const unsigned t = rest % khw;
    rest /= khw;
    dst[i] = src[((size_t)rest * Ci_g + ci) * khw + t];
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
  uint32_t a_bytes, b_bytes;
  // gemm_small_kernel only: asum[m] (+)= sum_k op(A)[m][k] -- for a Linear layer's weight gradient dW = dY^T X this is its
  // bias gradient, taken from the A chunks the kernel stages anyway (first column of tiles) instead of a colsum launch
  float* asum;
  int asum_accumulate;
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
  const rsrc_t ars = make_rsrc(p.A, p.a_bytes);
  const rsrc_t brs = make_rsrc(p.B, p.b_bytes);
  const uint32_t a_z = (uint32_t)(blockIdx.z * p.bsA), b_z = (uint32_t)(blockIdx.z * p.bsB);
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
      const bool ok = k < p.K && m < p.M;
      ra[e] = buf_load(ars, guard_off(a_z + (uint32_t)m * (uint32_t)p.sam + (uint32_t)k * (uint32_t)p.sak, ok));
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
      const bool ok = k < p.K && n < p.N;
      rb[e] = buf_load(brs, guard_off(b_z + (uint32_t)k * (uint32_t)p.sbk + (uint32_t)n * (uint32_t)p.sbn, ok));
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

// Small products (GModule / TGCN / attention: 240..1100 nodes x 256 x 256): a 64 x 64 tile grid is 16-70 workgroups and
// each wave walks all of K on ONE accumulator -- a chain of K/2 dependent 64-cycle MFMAs (8 192 cycles for K = 256) behind
// K/32 dependent L2 round trips.  Here a workgroup owns a 32 x 32 tile and its four waves split K four ways (k = 4 c + w
// chunks of 16): 4x the workgroups, a quarter of the chain, partial tiles added in wave order through LDS.
template <bool A_LANE_K, bool B_LANE_K>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmParams p) {
  constexpr int T = 32, KC = 16;
  __shared__ float sA[4][KC][T + 1], sB[4][KC][T + 1];      // [wave][k][t]: each wave stages and reads its own chunk
  __shared__ float sR[3][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x % p.tiles_m, tn = blockIdx.x / p.tiles_m;
  const int m0 = tm * T, n0 = tn * T;
  const rsrc_t ars = make_rsrc(p.A, p.a_bytes);
  const rsrc_t brs = make_rsrc(p.B, p.b_bytes);
  const uint32_t a_z = (uint32_t)(blockIdx.z * p.bsA), b_z = (uint32_t)(blockIdx.z * p.bsB);
  float* C = p.C + (size_t)blockIdx.z * p.bsC;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // a wave's 16 x 32 chunk of each operand: 8 elements per lane; lanes walk the unit-stride direction
  const int nchunks = (p.K + KC - 1) / KC;
  float ra[8], rb[8];
  auto load = [&](int c) {
    const int k0 = c * KC;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int kk, t;
      if (A_LANE_K) { kk = lane % KC; t = lane / KC + e * 4; } else { t = lane % T; kk = lane / T + e * 2; }
      const int k = k0 + kk, m = m0 + t;
      ra[e] = buf_load(ars, guard_off(a_z + (uint32_t)m * (uint32_t)p.sam + (uint32_t)k * (uint32_t)p.sak, k < p.K && m < p.M));
      if (B_LANE_K) { kk = lane % KC; t = lane / KC + e * 4; } else { t = lane % T; kk = lane / T + e * 2; }
      const int kb = k0 + kk, n = n0 + t;
      rb[e] = buf_load(brs, guard_off(b_z + (uint32_t)kb * (uint32_t)p.sbk + (uint32_t)n * (uint32_t)p.sbn, kb < p.K && n < p.N));
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int kk, t;
      if (A_LANE_K) { kk = lane % KC; t = lane / KC + e * 4; } else { t = lane % T; kk = lane / T + e * 2; }
      sA[wave][kk][t] = ra[e];
      if (B_LANE_K) { kk = lane % KC; t = lane / KC + e * 4; } else { t = lane % T; kk = lane / T + e * 2; }
      sB[wave][kk][t] = rb[e];
    }
  };
  const int li = lane & 31, hi = lane >> 5;
  const bool do_asum = p.asum && tn == 0 && hi == 0;
  float asum = 0.f;
  __shared__ float sS[3][32];
  int c = wave;
  if (c < nchunks) load(c);
  for (; c < nchunks; c += 4) {
    stage();                       // wave-private LDS: no workgroup barrier in the loop
    if (c + 4 < nchunks) load(c + 4);
    if (do_asum) {                 // rows past M / columns past K were staged as zeros
      float t = 0.f;
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) t += sA[wave][kk][li];
      asum += t;
    }
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[wave][kk + hi][li], sB[wave][kk + hi][li], acc, 0, 0, 0);
  }
  // partial tiles of waves 1..3 -> LDS; wave 0 adds them in wave order and writes
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave - 1][acc_row(r, hi)][li] = acc[r];
    if (do_asum) sS[wave - 1][li] = asum;
  }
  __syncthreads();
  if (wave == 0 && do_asum && m0 + li < p.M) {
    const float v = ((asum + sS[0][li]) + sS[1][li]) + sS[2][li];
    float* o = p.asum + (size_t)blockIdx.z * p.M + m0 + li;
    *o = (p.asum_accumulate ? *o : 0.f) + v;
  }
  if (wave == 0) {
    const int n = n0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = acc_row(r, hi), m = m0 + row;
      float v = ((acc[r] + sR[0][row][li]) + sR[1][row][li]) + sR[2][row][li];
      if (m < p.M && n < p.N) {
        v *= p.alpha;
        if (p.bias_mode == 1) v += p.bias[m];
        if (p.bias_mode == 2) v += p.bias[n];
        float* cp = C + (size_t)m * p.scm + (size_t)n * p.scn;
        if (p.accumulate) v += *cp;
        if (p.relu) v = fmaxf(v, 0.f);
        *cp = v;
      }
    }
  }
}

// =========================================================================================
// Host side
// =========================================================================================
// K-chunk per filter size: a multiple of kh*kw where that is cheap (3x3 -> 18 / 36, 1x1 -> 16) so the tap of
// every staged element is fixed (TAPFIX path); generic sizes use 16.
template <int KH>
struct ConvTiles {
  typedef TileCfg<2, 2, 2, 2, 16> T128;
  typedef TileCfg<2, 2, 1, 2, 16> T64x128;
  typedef TileCfg<2, 2, 1, 1, 16> T64;
};
template <>
struct ConvTiles<3> {
  typedef TileCfg<2, 2, 2, 2, 18> T128;      // 128 x 128, 4 waves x (64 x 64); 36.9 KB LDS
  typedef TileCfg<2, 2, 1, 2, 36> T64x128;   // STEP_A = 4 needs a multiple of 4 and 9
  typedef TileCfg<2, 2, 1, 1, 36> T64;
};
#ifndef GE_K1_KC
#define GE_K1_KC 16
#endif
template <>
struct ConvTiles<1> {   // 1x1: every chunk length keeps the tap fixed; GE_K1_KC is a tuning knob
  typedef TileCfg<2, 2, 2, 2, GE_K1_KC> T128;
  typedef TileCfg<2, 2, 1, 2, GE_K1_KC> T64x128;
  typedef TileCfg<2, 2, 1, 1, GE_K1_KC> T64;
};
typedef TileCfg<2, 2, 1, 1, 32> Tile64;       // strided GEMM: 32-deep chunks (GModule's 240..600 x 256 x 256 products are a chain of dependent L2 round trips: 8 instead of 16)
typedef TileCfg<2, 2, 2, 2, 32> WTile128;     // wgrad: K chunk 32 so a lane group covers a 128 B line
typedef TileCfg<2, 2, 1, 1, 32> WTile64;

template <class T, int KH, int KW, bool TR, bool SUB, bool EXACT, bool SWAP = false, bool LDSD = false>
static void launch_conv_gemm_variant(ConvGemmParams& p, const dim3& grid, size_t lds, hipStream_t st) {
    if (lds > 48 * 1024) GE_MAX_LDS((int)lds, (const void*)conv_gemm_kernel<T, KH, KW, TR, SUB, EXACT, SWAP, LDSD>);
  hipLaunchKernelGGL((conv_gemm_kernel<T, KH, KW, TR, SUB, EXACT, SWAP, LDSD>), grid, dim3(T::NTHREADS), lds, st, p);
}

template <class T, int KH, int KW, bool TR, bool SUB = false>
static int launch_conv_gemm(ConvGemmParams& p, int G, hipStream_t st) {
  p.tiles_m = ge_cdiv(p.M, T::MT);
  p.tiles_n = ge_cdiv(p.N, T::NT);
  constexpr int dbg = 0;      // (phase-ablation bits: tuning builds edit this line)
  p.dbg = dbg;
  if (p.splits > 1) {   // requested split count -> whole chunks of this tile's KC, no empty split
    const int nch = ge_cdiv(p.K, T::KC);
    p.split_chunks = ge_cdiv(nch, p.splits);
    p.splits = ge_cdiv(nch, p.split_chunks);
  }
  if (p.splits <= 1) p.splits = 1;
  dim3 grid(p.tiles_m * p.tiles_n, p.splits, G);
  const size_t lds = 2 * (size_t)T::KC * (T::MT + T::NT) * sizeof(float);
  constexpr bool TAPFIX_L = SUB || (KH * KW > 0 && (T::KC % (KH * KW) == 0));
  static const bool exact_on = !(getenv("GE_CONV_EXACT") && atoi(getenv("GE_CONV_EXACT")) == 0);
  const bool exact = TAPFIX_L && exact_on && p.K % T::KC == 0;
  // vector epilogue (SWAP): 1x1 layers on the exact loader whose result covers the whole map with 4-aligned planes
  static const bool swap_on = !(getenv("GE_CONV_SWAP") && atoi(getenv("GE_CONV_SWAP")) == 0);
  bool swap = false, ldsd = false;
  p.full = (p.M % T::MT == 0 && p.N % T::NT == 0 && (p.Hd * p.Wd) % T::NT == 0) ? 1 : 0;
  if constexpr (TAPFIX_L) {
    if constexpr (KH == 1 && KW == 1 && !SUB) {
      swap = exact && swap_on && p.os == 1 && p.ooy == 0 && p.oox == 0 && ((p.Hd * p.Wd) & 3) == 0;
      // operands straight into LDS (three stages): stride 1, no padding, same plane on both sides, M % 4 == 0
      static const bool ldsd_on = !(getenv("GE_CONV_LDSD") && atoi(getenv("GE_CONV_LDSD")) == 0);
      ldsd = swap && ldsd_on && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd && (p.M & 3) == 0;
      if (ldsd && T::MT + T::NT == 256) {
        // the 128 x 128 tile holds 48 KB of LDS with three stages: 3 workgroups per CU instead of 4.  Exactly one round
        // of 4 per CU (1024 tiles) became 2 rounds of 3 -- measured 47 -> 52 us on 128 -> 512 @32x32 x 32 -- so compare
        // rounds x residency and stay with the register loader when the three-stage plan loses more than a fifth
        const long long tiles = (long long)grid.x * grid.y * grid.z;
        const long long r4 = (tiles + 1023) / 1024 * 4, r3 = (tiles + 767) / 768 * 3;
        if (r3 * 5 > r4 * 6) ldsd = false;
      }
      if (ldsd)
        launch_conv_gemm_variant<T, KH, KW, TR, SUB, true, true, true>(p, grid, 3 * (size_t)T::KC * (T::MT + T::NT) * sizeof(float), st);
      else if (swap)
        launch_conv_gemm_variant<T, KH, KW, TR, SUB, true, true>(p, grid, lds, st);
    } else if constexpr (KH == 3 && KW == 3 && !SUB) {
      // the vector epilogue for the 3x3 DATA GRADIENT too (no bias, no moments: its values are bit-identical to the scalar
      // epilogue's).  The forward pass stays on the scalar one: with the bias as initial value and the moments taken per
      // 4-position quad the results differ in the last bit, and the VGG16 chain (no residual paths, train-mode BatchNorm)
      // amplifies that past the tolerance of test_fpn_forward_backward_vs_oracle[VGG16] (one BatchNorm-weight gradient
      // 2e-4 -> 9e-3 against fp64) -- GE_CONV_SWAP3=2 turns it on for the forward pass as well (+0.3 % on config 2)
      static const int swap3 = getenv("GE_CONV_SWAP3") ? atoi(getenv("GE_CONV_SWAP3")) : 1;
      swap = exact && swap_on && (swap3 == 2 || (swap3 == 1 && TR)) && p.os == 1 && p.ooy == 0 && p.oox == 0 &&
             ((p.Hd * p.Wd) & 3) == 0;
      if (swap) launch_conv_gemm_variant<T, KH, KW, TR, SUB, true, true>(p, grid, lds, st);
    }
    if (!swap) {
      if (exact)
        launch_conv_gemm_variant<T, KH, KW, TR, SUB, true>(p, grid, lds, st);
      else
        launch_conv_gemm_variant<T, KH, KW, TR, SUB, false>(p, grid, lds, st);
    }
  } else {
    launch_conv_gemm_variant<T, KH, KW, TR, SUB, false>(p, grid, lds, st);
  }
  if (swap)
    ge_note_kernel("conv_gemm_kernel<TileCfg<%d, %d, %d, %d, %d>, %d, %d, %s, %s, true, true, %s>", T::WM, T::WN, T::TM, T::TN,
                   T::KC, KH, KW, TR ? "true" : "false", SUB ? "true" : "false", ldsd ? "true" : "false");
  else
    ge_note_kernel("conv_gemm_kernel<TileCfg<%d, %d, %d, %d, %d>, %d, %d, %s, %s, %s, false, false>", T::WM, T::WN, T::TM, T::TN,
                   T::KC, KH, KW, TR ? "true" : "false", SUB ? "true" : "false", exact ? "true" : "false");
  GE_CHECK_LAUNCH("conv_gemm");
  if (p.splits > 1) {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ge_stream_grid(p.slab_elems, 256)), dim3(256), 0, st, p.ws, p.dst,
                       p.slab_elems, p.splits - 1, 1);
    GE_CHECK_LAUNCH("conv_split_reduce");
  }
  return GE_OK;
}

// Tile choice for an (M x N) x G implicit GEMM: 0 = 128x128, 1 = 64x128, 2 = 64x64.  A tile size is used only when it
// still gives >= 1.5 (128x128) / >= 4 (64x128) workgroups per CU: one wave per SIMD cannot hide its own LDS and global
// latency (tools/bench_tile_choice.py: with exactly 256 big tiles the 64x64 plan is 20 % faster on 512->128 @32x32,
// 512->2048 @8x8 and 128->128 3x3 @32x32).  GE_T128_MIN / GE_T64X128_MIN override the thresholds for tuning runs.
static int conv_tile_choice(long long M, long long N, int G) {
  constexpr int min128 = 384;
  constexpr int min64x128 = 1024;
  static const int force = getenv("GE_FORCE_TILE") ? atoi(getenv("GE_FORCE_TILE")) : -1;
  if (force >= 0) return force;
  const long long t128 = (long long)ge_cdiv(M, 128) * ge_cdiv(N, 128) * G;
  const long long t64x128 = (long long)ge_cdiv(M, 64) * ge_cdiv(N, 128) * G;
  if (M > 64 && t128 >= min128) return 0;
  if (t64x128 >= min64x128) return 1;
  return 2;
}

// Split-K plan for layers whose tile grid cannot fill the chip (N = B*Ho*Wo of a few thousand: the 16x16 and 8x8 stages
// at small per-GPU batches, the 8x8 stage at any batch).  The reduction over K = Cin*kh*kw is cut into `splits` ranges
// computed by different workgroups (grid.y); partial results go to slabs and one pass adds them up.  Returns the tile
// choice and sets `splits` (1 = no split).  GE_SPLITK=0 disables, GE_SPLITK_S / GE_SPLITK_TILE force a plan (tuning).
static int conv_split_plan(long long M, long long N, long long K, int G, int& splits) {
  static const int on = getenv("GE_SPLITK") ? atoi(getenv("GE_SPLITK")) : 1;
  static const int force_s = getenv("GE_SPLITK_S") ? atoi(getenv("GE_SPLITK_S")) : 0;
  static const int force_t = getenv("GE_SPLITK_TILE") ? atoi(getenv("GE_SPLITK_TILE")) : -1;
  constexpr int target = 512;
  constexpr int below = 256;
  int choice = conv_tile_choice(M, N, G);
  splits = 1;
  if (!on) return choice;
  const int mt[3] = {128, 64, 64}, nt[3] = {128, 128, 64};
  const long long blocks = (long long)ge_cdiv(M, mt[choice]) * ge_cdiv(N, nt[choice]) * G;
  if (force_s > 0) {
    splits = force_s;
    return force_t >= 0 ? force_t : choice;
  }
  // measured (tools/bench_splitk.py, profiles/r02_splitk_microbench.txt): below one workgroup per CU the split always
  // pays (512->512 3x3 @8x8 x 8 frames: 121 -> 37 us); at exactly one per CU only when K is long (>= 2048)
  if (blocks > below || K < 512 || (blocks == below && K < 2048)) return choice;
  // keep the small tile (its grid is the largest) and cut K until ~2 workgroups per CU exist; every split keeps at
  // least 256 of K so that the slab traffic stays small against the operand traffic
  if (force_t >= 0) choice = force_t;
  const long long b2 = (long long)ge_cdiv(M, mt[choice]) * ge_cdiv(N, nt[choice]) * G;
  long long s = (target + b2 - 1) / b2;
  s = std::min<long long>(s, K / 256);
  s = std::min<long long>(s, 16);
  splits = (int)std::max<long long>(s, 1);
  return choice;
}

template <int KH, int KW, bool TR, bool SUB = false>
static int dispatch_conv_tile(ConvGemmParams& p, int G, hipStream_t st) {
  typedef ConvTiles<SUB ? 0 : KH> CT;
  int choice = conv_tile_choice(p.M, p.N, G);
  if (p.ws) {
    int s = 1;
    choice = conv_split_plan(p.M, p.N, p.K, G, s);
    p.splits = s;
  }
  if constexpr (KH == 1 && KW == 1 && !SUB) {
    // 1x1 layers the direct-to-LDS loader takes: the 64 x 128 tile keeps 36 KB of LDS (four workgroups per CU) where the
    // 128 x 128 one needs 48 (three); measured per layer (tools/bench_conv1x1.py, GE_FORCE_TILE): the big tile only pays
    // when there are >= 2048 of them AND K is long enough to amortise its epilogue (512 -> 256 @64x64: 266 vs 280 us);
    // every layer with <= 1024 big tiles is 1-6 % faster on 64 x 128
    static const bool ldsd_tiles = !(getenv("GE_CONV_LDSD") && atoi(getenv("GE_CONV_LDSD")) == 0) &&
                                   !(getenv("GE_FORCE_TILE"));
    const long long t128 = (long long)ge_cdiv(p.M, 128) * ge_cdiv(p.N, 128) * G;
    if (ldsd_tiles && choice == 0 && p.splits <= 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd &&
        (p.M & 3) == 0 && p.K % 16 == 0 && ((p.Hd * p.Wd) & 3) == 0 && !(t128 >= 2048 && p.K >= 256))
      choice = 1;
  }
  if (choice == 0) return launch_conv_gemm<typename CT::T128, KH, KW, TR, SUB>(p, G, st);
  if (choice == 1) return launch_conv_gemm<typename CT::T64x128, KH, KW, TR, SUB>(p, G, st);
  return launch_conv_gemm<typename CT::T64, KH, KW, TR, SUB>(p, G, st);
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

// All conv weights of a model at once.  table: device int64 [n][8] rows (src_off, dst_off, total, G, Co_g, Ci_g,
// kh*kw, transposed); offsets in floats into `w` / `out`; every total < 2^32.
int ge_conv2d_pack_weights_batched(const float* w, float* out, const long long* table, int n, void* stream) {
  GE_REQUIRE(w && out && table && n > 0 && n <= 65535, "pack_weights_batched: bad arguments");
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(48, n), dim3(256), 0, (hipStream_t)stream, w, out, table);
  GE_CHECK_LAUNCH("pack_weights_batched");
  return GE_OK;
}

// y[B,Cout,Ho,Wo] = conv2d(x[B,Cin,Hi,Wi], w) (+bias)(+relu); wp = ge_conv2d_pack_weight(..., transposed=0).
// Number of (count, mean, M2) partials per channel that ge_conv2d_fwd writes into `stats` (mirrors the tile choice).
int ge_conv2d_fwd_stat_parts(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  const int M = Cout / groups;
  const long long N = (long long)B * Ho * Wo;
  (void)Cin;
  (void)kh;
  (void)kw;
  // T128 / T64x128: NT 128, 2 waves along N; T64: NT 64, 2 waves along N
  return conv_tile_choice(M, N, groups) <= 1 ? ge_cdiv(N, 128) * 2 : ge_cdiv(N, 64) * 2;
}

// stats (nullable): [Cout][ge_conv2d_fwd_stat_parts()][3] fused BatchNorm moments of y (requires relu == 0).
static int conv2d_fwd_impl(const float* x, const float* wp, const float* bias, float* y, float* stats, int B, int Cin,
                           int Hi, int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups,
                           int relu, float* workspace, void* stream) {
  GE_REQUIRE(!(stats && relu), "conv2d_fwd: fused statistics are those of the pre-activation output");
  GE_REQUIRE(!(workspace && (stats || relu)), "conv2d_fwd: the split-K path has no fused statistics / activation");
  GE_REQUIRE(x && wp && y, "conv2d_fwd: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_fwd: bad shape");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_fwd: B*Ho*Wo overflows int32");
  if (!stats && !relu && ge_conv3x3_c1_applies(Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, pad, groups) &&
      ge_conv3x3_c1_fwd_applies(B, Hi, Wi))
    return ge_conv3x3_c1_fwd(x, wp, bias, y, B, Cin, Hi, Wi, (hipStream_t)stream);   // packed == OIHW when Cout = 1
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
  p.os = 1;
  p.ooy = p.oox = 0;
  p.ntaps = 0;
  p.tap_shift = 0;
  p.addend = nullptr;
  p.stats = stats;
  p.stats_parts = stats ? ge_conv2d_fwd_stat_parts(B, Cin, Cout, Ho, Wo, kh, kw, groups) : 0;
  p.ws = workspace;
  p.slab_elems = (long long)B * Cout * Ho * Wo;
  p.splits = 1;
  p.split_chunks = 0;
  p.div_hw = make_fastdiv(Ho * Wo);
  p.div_w = make_fastdiv(Wo);
  const long long xb = 4ll * B * Cin * Hi * Wi, wb = 4ll * Cout * p.Cs_g * kh * kw;
  GE_REQUIRE(xb < 0xFFFFFFF0ll && wb < 0xFFFFFFF0ll, "conv2d_fwd: tensors of 4 GiB or more are not supported");
  p.src_bytes = (uint32_t)xb;
  p.wp_bytes = (uint32_t)wb;
  return dispatch_conv<false>(p, groups, (hipStream_t)stream);
}

int ge_conv2d_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int B, int Cin, int Hi,
                  int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int relu,
                  void* stream) {
  return conv2d_fwd_impl(x, wp, bias, y, stats, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, relu, nullptr,
                         stream);
}

// Floats of workspace the split-K plan wants for this layer (0: its tile grid fills the chip, use ge_conv2d_fwd).
long long ge_conv2d_fwd_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  int s = 1;
  (void)conv_split_plan(Cout / groups, (long long)B * Ho * Wo, (long long)(Cin / groups) * kh * kw, groups, s);
  return s > 1 ? (long long)(s - 1) * B * Cout * Ho * Wo : 0;
}

// ge_conv2d_fwd with the reduction over Cin*kh*kw split across workgroups (plus one pass that adds the partial results):
// for layers ge_conv2d_fwd_workspace() reports a workspace for.  No fused statistics / activation on this path.
int ge_conv2d_fwd_splitk(const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Hi, int Wi,
                         int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, float* workspace,
                         void* stream) {
  GE_REQUIRE(workspace, "conv2d_fwd_splitk: null workspace");
  return conv2d_fwd_impl(x, wp, bias, y, nullptr, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, 0, workspace,
                         stream);
}

// dx[B,Cin,Hi,Wi] = conv2d data gradient of dy[B,Cout,Ho,Wo] (+ addend, e.g. the gradient arriving through a skip
// connection); wp = ge_conv2d_pack_weight(..., transposed=1).
// stride 2 is decomposed by output parity: 1x1 -> one dense GEMM over the Ho x Wo grid scattered to the even
// positions of a zero-filled dx; 3x3 pad 1 -> four sub-convolutions with 1/2/2/4 taps (9 tap-GEMMs instead of 36).
static int conv2d_dgrad_impl(const float* dy, const float* wp, const float* addend, float* dx, int B, int Cin, int Hi,
                             int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups,
                             float* workspace, void* stream) {
  GE_REQUIRE(dy && wp && dx, "conv2d_dgrad: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_dgrad: bad shape");
  GE_REQUIRE((long long)B * Hi * Wi < (1ll << 31), "conv2d_dgrad: B*Hi*Wi overflows int32");
  hipStream_t st = (hipStream_t)stream;
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
  p.stride = stride;
  p.pad = pad;
  p.kh = kh;
  p.kw = kw;
  p.relu = 0;
  p.os = 1;
  p.ooy = p.oox = 0;
  p.ntaps = 0;
  p.tap_shift = 0;
  p.addend = addend;
  p.stats = nullptr;
  p.stats_parts = 0;
  p.ws = nullptr;
  p.slab_elems = (long long)B * Cin * Hi * Wi;
  p.splits = 1;
  p.split_chunks = 0;
  const long long yb = 4ll * B * Cout * Ho * Wo, wb = 4ll * Cout * (Cin / groups) * kh * kw;
  GE_REQUIRE(yb < 0xFFFFFFF0ll && wb < 0xFFFFFFF0ll, "conv2d_dgrad: tensors of 4 GiB or more are not supported");
  p.src_bytes = (uint32_t)yb;
  p.wp_bytes = (uint32_t)wb;

  if (stride == 2 && kh == 1 && kw == 1 && pad == 0) {
    // dx[:, :, 2u, 2v] = W^T dy[:, :, u, v]; every other position only receives the addend (or zero)
    ge_init_async(dx, addend, (long long)B * Cin * Hi * Wi, st);      // a kernel, never a memset / memcpy node (ge_common.h)
    GE_CHECK_LAUNCH("conv2d_dgrad_init");
    p.stride = 1;   // dense 1x1 "forward" over the Ho x Wo grid with the data-gradient operand
    p.os = 2;
    p.N = B * Ho * Wo;
    p.K = p.Cs_g;
    p.div_hw = make_fastdiv(Ho * Wo);
    p.div_w = make_fastdiv(Wo);
    return dispatch_conv_tile<1, 1, false>(p, groups, st);
  }
  if (stride == 2 && kh == 3 && kw == 3 && pad == 1) {
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int Hc = (Hi - py + 1) / 2, Wc = (Wi - px + 1) / 2;
        if (Hc <= 0 || Wc <= 0) continue;
        // ty = y + 1 - kh must be even: y even -> kh = 1; y odd -> kh in {0, 2}
        int khs[2], kws[2];
        const int nkh = py == 0 ? (khs[0] = 1, 1) : (khs[0] = 0, khs[1] = 2, 2);
        const int nkw = px == 0 ? (kws[0] = 1, 1) : (kws[0] = 0, kws[1] = 2, 2);
        p.ntaps = nkh * nkw;
        p.tap_shift = p.ntaps == 1 ? 0 : (p.ntaps == 2 ? 1 : 2);
        for (int a = 0; a < nkh; ++a)
          for (int b2 = 0; b2 < nkw; ++b2) p.taps[a * nkw + b2] = khs[a] * 3 + kws[b2];
        p.os = 2;
        p.ooy = py;
        p.oox = px;
        p.N = B * Hc * Wc;
        p.K = p.Cs_g * p.ntaps;
        p.div_hw = make_fastdiv(Hc * Wc);
        p.div_w = make_fastdiv(Wc);
        const int rc = dispatch_conv_tile<3, 3, true, true>(p, groups, st);
        if (rc) return rc;
      }
    return GE_OK;
  }
  p.N = B * Hi * Wi;
  p.K = p.Cs_g * kh * kw;
  p.div_hw = make_fastdiv(Hi * Wi);
  p.div_w = make_fastdiv(Wi);
  p.ws = stride == 1 ? workspace : nullptr;
  return dispatch_conv<true>(p, groups, st);
}

int ge_conv2d_dgrad(const float* dy, const float* wp, const float* addend, float* dx, int B, int Cin, int Hi, int Wi,
                    int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, void* stream) {
  return conv2d_dgrad_impl(dy, wp, addend, dx, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, nullptr, stream);
}

// Split-K plan of the stride-1 data gradient (reduction over Cout*kh*kw); 0 for strided layers and filled grids.
long long ge_conv2d_dgrad_workspace(int B, int Cin, int Hi, int Wi, int Cout, int kh, int kw, int stride, int groups) {
  if (stride != 1) return 0;
  int s = 1;
  (void)conv_split_plan(Cin / groups, (long long)B * Hi * Wi, (long long)(Cout / groups) * kh * kw, groups, s);
  return s > 1 ? (long long)(s - 1) * B * Cin * Hi * Wi : 0;
}

int ge_conv2d_dgrad_splitk(const float* dy, const float* wp, const float* addend, float* dx, int B, int Cin, int Hi,
                           int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups,
                           float* workspace, void* stream) {
  GE_REQUIRE(workspace && stride == 1, "conv2d_dgrad_splitk: needs a workspace and stride 1");
  return conv2d_dgrad_impl(dy, wp, addend, dx, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, workspace,
                           stream);
}

}  // extern "C"

template <class T, int KH, int KW>
static int launch_wgrad(WgradParams& p, int G, float* dw, hipStream_t st) {
  p.tiles_m = ge_cdiv(p.M, T::MT);
  p.tiles_j = ge_cdiv(p.J, T::NT);
  constexpr int dbg = 0;      // (phase-ablation bits: tuning builds edit this line)
  p.dbg = dbg;
  const size_t lds = (size_t)(T::MT + T::NT) * (T::KC + 1) * sizeof(float);
    GE_MAX_LDS((int)lds, (const void*)conv_wgrad_kernel<T, KH, KW>);
  dim3 grid(p.tiles_m * p.tiles_j * p.splits, 1, G);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, KH, KW>), grid, dim3(T::NTHREADS), lds, st, p);
  ge_note_kernel("conv_wgrad_kernel<TileCfg<%d, %d, %d, %d, %d>, %d, %d>", T::WM, T::WN, T::TM, T::TN, T::KC, KH, KW);
  GE_CHECK_LAUNCH("conv_wgrad");
  return GE_OK;
}

template <class T, int WC>
static int launch_wgrad3x3(WgradParams& p, int G, hipStream_t st) {
  p.tiles_m = ge_cdiv(p.M, T::MT);
  p.tiles_j = ge_cdiv(p.J, T::NT);
  p.dbg = 0;
  constexpr int ROWS = T::KC / WC + 2, LDC = 35, CS = ((ROWS * LDC - 9 + 31) / 32) * 32 + 9, NCH = T::NT / 9 + 2;
  static const bool db = getenv("GE_WGRAD_DB") && atoi(getenv("GE_WGRAD_DB")) != 0;   // measured: 1 stage is faster
  const size_t lds = (db ? 2 : 1) * ((size_t)T::MT * (T::KC + 1) + (size_t)NCH * CS) * sizeof(float);
  dim3 grid(p.tiles_m * p.tiles_j * p.splits, 1, G);
  if (db)
    hipLaunchKernelGGL((conv_wgrad3x3_kernel<T, WC, true>), grid, dim3(T::NTHREADS), lds, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad3x3_kernel<T, WC, false>), grid, dim3(T::NTHREADS), lds, st, p);
  ge_note_kernel("conv_wgrad3x3_kernel<TileCfg<%d, %d, %d, %d, %d>, %d, %s>", T::WM, T::WN, T::TM, T::TN, T::KC, WC,
                 db ? "true" : "false");
  GE_CHECK_LAUNCH("conv_wgrad3x3");
  return GE_OK;
}

static void wgrad_plan(int M, int J, int G, int Ktot, int& big, int& splits, int& klen) {
  const long long t128 = (long long)ge_cdiv(M, 128) * ge_cdiv(J, 128) * G;
  const int kc = 32;
  const int chunks = ge_cdiv(Ktot, kc);
  int max_splits = chunks / 8 > 0 ? chunks / 8 : 1;  // >= 8 chunks (256 positions) per split
  // 128x128 tiles need >= 8 of them: below that the 64x64 plan measured faster even though it re-reads each
  // operand twice as often (more K-splits of big tiles cost more in the slab reduce than the re-reads save).
  big = (M > 64 && J > 64 && t128 >= 8) ? 1 : 0;
  const long long tiles = big ? t128 : (long long)ge_cdiv(M, 64) * ge_cdiv(J, 64) * G;
  // Pick the split count whose workgroup total balances best over the 256 CUs (every CU should get the same
  // number of equally long workgroups: 792 workgroups = 3.09 per CU costs a 4th round on 24 CUs), among counts
  // that give roughly 3-4 workgroups per CU.
  int best_klen = chunks * kc, best_splits = 1;
  double best_score = -1.0;
  for (int s = 1; s <= max_splits; ++s) {
    const int kl = ge_cdiv(chunks, s) * kc;
    const int sa = ge_cdiv(Ktot, kl);              // split count actually produced by this chunking
    const long long blocks = tiles * sa;
    if (blocks > 1024 && s > 1) break;             // at most 4 co-resident workgroups per CU
    const double per_cu = (double)blocks / 256.0;
    const double rounds = (double)((blocks + 255) / 256);
    double score = per_cu / rounds;                // balance in (0, 1]
    if (per_cu < 1.0) score *= per_cu;             // do not leave CUs empty
    score *= (per_cu >= 2.5 ? 1.0 : 0.85 + 0.06 * per_cu);   // enough workgroups per CU to hide latency
    if (score > best_score + 1e-9) {
      best_score = score;
      best_klen = kl;
      best_splits = sa;
    }
  }
  klen = best_klen;
  splits = best_splits;
}

extern "C" {

// Workspace (floats) needed by ge_conv2d_wgrad.
long long ge_conv2d_wgrad_workspace(int B, int Cin, int Cout, int Ho, int Wo, int kh, int kw, int groups) {
  int big, splits, klen;
  wgrad_plan(Cout / groups, (Cin / groups) * kh * kw, groups, B * Ho * Wo, big, splits, klen);
  const long long gemm = (long long)splits * ((long long)Cout * (Cin / groups) * kh * kw + Cout);   // + bias row sums
  // one-output-channel 3x3 layers may take the reduction kernel of ge_conv_c1.hip: [B][Cin][9] partial sums
  const long long c1 = (Cout == 1 && groups == 1 && kh == 3 && kw == 3) ? ge_conv3x3_c1_wgrad_workspace(B, Cin) : 0;
  return gemm > c1 ? gemm : c1;
}

// dw[Cout, Cin/groups, kh, kw] (+)= conv2d weight gradient.  workspace: ge_conv2d_wgrad_workspace() floats.
static bool wgrad_takes_patch_kernel(int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad, int klen) {
  static const bool patch_on = !(getenv("GE_WGRAD_PATCH") && atoi(getenv("GE_WGRAD_PATCH")) == 0);
  const int wc = (Wo % 32 == 0) ? 32 : ((Wo == 16 || Wo == 8) ? Wo : 0);
  return kh == 3 && kw == 3 && patch_on && wc && stride == 1 && pad == 1 && Hi == Ho && Wi == Wo && (Ho * Wo) % 32 == 0 &&
         klen % 32 == 0;
}

static int conv2d_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int Cin, int Hi,
                             int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups,
                             int accumulate, void* stream);

int ge_conv2d_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int Hi, int Wi,
                    int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate,
                    void* stream) {
  return conv2d_wgrad_impl(x, dy, dw, nullptr, workspace, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups,
                           accumulate, stream);
}

// 1 when ge_conv2d_wgrad_bias computes the bias gradient inside the weight-gradient pass for this layer (the general MFMA
// kernel; not the 3x3 patch kernel -- its register budget is spent -- and not the one-output-channel reduction)
int ge_conv2d_wgrad_fuses_bias(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad,
                               int groups) {
  static const bool on = !(getenv("GE_WGRAD_BIAS") && atoi(getenv("GE_WGRAD_BIAS")) == 0);
  if (!on) return 0;
  if (ge_conv3x3_c1_applies(Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, pad, groups) && ge_conv3x3_c1_wgrad_applies(Hi, Wi))
    return 0;
  int big, splits, klen;
  wgrad_plan(Cout / groups, (Cin / groups) * kh * kw, groups, B * Ho * Wo, big, splits, klen);
  return wgrad_takes_patch_kernel(Hi, Wi, Ho, Wo, kh, kw, stride, pad, klen) ? 0 : 1;
}

// ge_conv2d_wgrad + db[Cout] (+)= sum over (b, y, x) of dy in the same launches (only where ge_conv2d_wgrad_fuses_bias
// says 1; `accumulate` bit 0 applies to dw and db alike; with bit 1 the slabs stay in the workspace for
// ge_slab_reduce_batched: stride Cout*J + Cout floats, the bias row sums in the last Cout of each)
int ge_conv2d_wgrad_bias(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int Cin, int Hi,
                         int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups, int accumulate,
                         void* stream) {
  GE_REQUIRE(db, "conv2d_wgrad_bias: db required");      // accumulate & 2: slabs (stride Cout*J + Cout, bias rows last) left to the caller
  GE_REQUIRE(ge_conv2d_wgrad_fuses_bias(B, Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, pad, groups),
             "conv2d_wgrad_bias: this layer's weight-gradient kernel does not produce the bias gradient");
  return conv2d_wgrad_impl(x, dy, dw, db, workspace, B, Cin, Hi, Wi, Cout, Ho, Wo, kh, kw, stride, pad, groups, accumulate,
                           stream);
}

static int conv2d_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int Cin, int Hi,
                             int Wi, int Cout, int Ho, int Wo, int kh, int kw, int stride, int pad, int groups,
                             int accumulate, void* stream) {
  GE_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad: null pointer");
  GE_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0 && stride > 0,
             "conv2d_wgrad: bad shape");
  GE_REQUIRE((long long)B * Ho * Wo < (1ll << 31), "conv2d_wgrad: B*Ho*Wo overflows int32");
  hipStream_t st = (hipStream_t)stream;
  if (ge_conv3x3_c1_applies(Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, pad, groups) && ge_conv3x3_c1_wgrad_applies(Hi, Wi))
    return ge_conv3x3_c1_wgrad(x, dy, dw, workspace, B, Cin, Hi, Wi, accumulate, st);
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
  const long long xb = 4ll * B * Cin * Hi * Wi, yb = 4ll * B * Cout * Ho * Wo;
  GE_REQUIRE(xb < 0xFFFFFFF0ll && yb < 0xFFFFFFF0ll, "conv2d_wgrad: tensors of 4 GiB or more are not supported");
  p.x_bytes = (uint32_t)xb;
  p.dy_bytes = (uint32_t)yb;
  int big;
  wgrad_plan(p.M, p.J, groups, p.Ktot, big, p.splits, p.klen);
  p.bias = db ? 1 : 0;
  p.bias_off = (long long)Cout * p.J;
  p.slab_stride = p.bias_off + (db ? Cout : 0);
  int rc;
  if (kh == 1 && kw == 1)
    rc = big ? launch_wgrad<WTile128, 1, 1>(p, groups, dw, st) : launch_wgrad<WTile64, 1, 1>(p, groups, dw, st);
  else if (kh == 3 && kw == 3) {
    const int wc = (Wo % 32 == 0) ? 32 : ((Wo == 16 || Wo == 8) ? Wo : 0);
    if (wgrad_takes_patch_kernel(Hi, Wi, Ho, Wo, kh, kw, stride, pad, p.klen)) {
      if (wc == 32)
        rc = big ? launch_wgrad3x3<WTile128, 32>(p, groups, st) : launch_wgrad3x3<WTile64, 32>(p, groups, st);
      else if (wc == 16)
        rc = big ? launch_wgrad3x3<WTile128, 16>(p, groups, st) : launch_wgrad3x3<WTile64, 16>(p, groups, st);
      else
        rc = big ? launch_wgrad3x3<WTile128, 8>(p, groups, st) : launch_wgrad3x3<WTile64, 8>(p, groups, st);
    } else {
      rc = big ? launch_wgrad<WTile128, 3, 3>(p, groups, dw, st) : launch_wgrad<WTile64, 3, 3>(p, groups, dw, st);
    }
  }
  else if (kh == 7 && kw == 7)
    rc = big ? launch_wgrad<WTile128, 7, 7>(p, groups, dw, st) : launch_wgrad<WTile64, 7, 7>(p, groups, dw, st);
  else
    rc = big ? launch_wgrad<WTile128, 0, 0>(p, groups, dw, st) : launch_wgrad<WTile64, 0, 0>(p, groups, dw, st);
  if (rc) return rc;
  ge_record_split_event(st);
  if (accumulate & 2) return GE_OK;      // the caller reduces the slabs later (ge_slab_reduce_batched)
  const long long n = p.slab_stride;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ge_stream_grid(n, 256)), dim3(256), 0, st, workspace, dw, n, p.splits,
                     accumulate & 1, db, p.bias_off);
  GE_CHECK_LAUNCH("slab_reduce");
  return GE_OK;
}

// Number of K-split slabs ge_conv2d_wgrad leaves in its workspace for this layer when called with accumulate | 2
// (0: the layer takes a kernel without slabs -- the one-output-channel reduction -- and cannot be deferred).
int ge_conv2d_wgrad_splits(int B, int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad,
                           int groups) {
  if (ge_conv3x3_c1_applies(Cin, Cout, Hi, Wi, Ho, Wo, kh, kw, stride, pad, groups) && ge_conv3x3_c1_wgrad_applies(Hi, Wi))
    return 0;
  int big, splits, klen;
  wgrad_plan(Cout / groups, (Cin / groups) * kh * kw, groups, B * Ho * Wo, big, splits, klen);
  return splits;
}

// The slab reduces of up to 16 weight-gradient calls in ONE launch (grid.y = call): out_i[j] (+)= sum_s slab_i[s][j] in
// split order -- the same sums, in the same order, as the reduce ge_conv2d_wgrad launches itself.  Host arrays.
int ge_slab_reduce_batched(const float* const* slabs, float* const* outs, const long long* ns, const long long* strides,
                           const int* splits, const int* accumulate, int count, void* stream) {
  GE_REQUIRE(slabs && outs && ns && splits && accumulate && count >= 1 && count <= SlabBatch::MAX,
             "slab_reduce_batched: 1..16 entries");
  SlabBatch b;
  long long nmax = 0;
  for (int i = 0; i < count; ++i) {
    GE_REQUIRE(slabs[i] && outs[i] && ns[i] > 0 && splits[i] >= 1, "slab_reduce_batched: bad entry");
    b.slab[i] = slabs[i];
    b.out[i] = outs[i];
    b.n[i] = ns[i];
    b.stride[i] = strides ? strides[i] : ns[i];
    GE_REQUIRE(b.stride[i] >= ns[i], "slab_reduce_batched: stride smaller than the reduced range");
    b.splits[i] = splits[i];
    b.accumulate[i] = accumulate[i];
    nmax = nmax > ns[i] ? nmax : ns[i];
  }
  const int gx = (int)std::min<long long>(256, (nmax + 1023) / 1024);
  hipLaunchKernelGGL(slab_reduce_batched_kernel, dim3(gx > 0 ? gx : 1, count), dim3(256), 0, (hipStream_t)stream, b);
  GE_CHECK_LAUNCH("slab_reduce_batched");
  return GE_OK;
}

// Strided batched GEMM (see GemmParams).  Either stride of each operand must be 1.
static bool gemm_small_applies(int M, int N, int K, int batch) {
  static const int small_on = getenv("GE_GEMM_SMALL") ? atoi(getenv("GE_GEMM_SMALL")) : 1;
  const long long t64 = (long long)ge_cdiv(M, Tile64::MT) * ge_cdiv(N, Tile64::NT) * batch;
  return small_on && t64 <= 128 && K >= 64;
}

static int gemm_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam,
                     long long sak, long long sbk, long long sbn, long long scm, long long scn, int batch, long long bsA,
                     long long bsB, long long bsC, float alpha, int bias_mode, int relu, int accumulate, float* asum,
                     int asum_accumulate, void* stream);

int ge_gemm(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam,
            long long sak, long long sbk, long long sbn, long long scm, long long scn, int batch, long long bsA,
            long long bsB, long long bsC, float alpha, int bias_mode, int relu, int accumulate, void* stream) {
  return gemm_impl(A, B, bias, C, M, N, K, sam, sak, sbk, sbn, scm, scn, batch, bsA, bsB, bsC, alpha, bias_mode, relu,
                   accumulate, nullptr, 0, stream);
}

// 1 when ge_gemm_rowsum takes this product (the 32 x 32-tile kernel: few output tiles, K >= 64)
int ge_gemm_rowsum_ok(int M, int N, int K, int batch) { return gemm_small_applies(M, N, K, batch) ? 1 : 0; }

// ge_gemm + asum[batch][M] (+)= sum_k op(A)[m][k] in the same launch (nn.Linear backward: dW = dY^T X and db = column sums
// of dY, models/transformer.py:14-38, models/graph_matching.py:148-162); only where ge_gemm_rowsum_ok says 1
int ge_gemm_rowsum(const float* A, const float* B, float* C, int M, int N, int K, long long sam, long long sak, long long sbk,
                   long long sbn, long long scm, long long scn, int batch, long long bsA, long long bsB, long long bsC,
                   float alpha, int accumulate, float* asum, int asum_accumulate, void* stream) {
  GE_REQUIRE(asum && gemm_small_applies(M, N, K, batch), "gemm_rowsum: asum required / product not on the small-tile kernel");
  return gemm_impl(A, B, nullptr, C, M, N, K, sam, sak, sbk, sbn, scm, scn, batch, bsA, bsB, bsC, alpha, 0, 0, accumulate, asum,
                   asum_accumulate, stream);
}

static int gemm_impl(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, long long sam,
                     long long sak, long long sbk, long long sbn, long long scm, long long scn, int batch, long long bsA,
                     long long bsB, long long bsC, float alpha, int bias_mode, int relu, int accumulate, float* asum,
                     int asum_accumulate, void* stream) {
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
  p.asum = asum;
  p.asum_accumulate = asum_accumulate;
  p.tiles_m = ge_cdiv(M, Tile64::MT);
  const long long ae = (batch - 1) * bsA + (M - 1) * sam + (K - 1) * sak + 1;
  const long long be = (batch - 1) * bsB + (K - 1) * sbk + (N - 1) * sbn + 1;
  GE_REQUIRE(sam >= 0 && sak >= 0 && sbk >= 0 && sbn >= 0 && bsA >= 0 && bsB >= 0, "gemm: negative strides");
  GE_REQUIRE(4 * ae < 0xFFFFFFF0ll && 4 * be < 0xFFFFFFF0ll, "gemm: operands of 4 GiB or more are not supported");
  p.a_bytes = (uint32_t)(4 * ae);
  p.b_bytes = (uint32_t)(4 * be);
  hipStream_t st = (hipStream_t)stream;
  const bool a_k = (sak == 1 && sam != 1), b_k = (sbk == 1 && sbn != 1);
  // few 64 x 64 tiles: 32 x 32 tiles with the four waves splitting K (GE_GEMM_SMALL=0 turns it off)
  if (gemm_small_applies(M, N, K, batch)) {
    p.tiles_m = ge_cdiv(M, 32);
    dim3 sgrid(p.tiles_m * ge_cdiv(N, 32), 1, batch);
    if (a_k && b_k)
      hipLaunchKernelGGL((gemm_small_kernel<true, true>), sgrid, dim3(256), 0, st, p);
    else if (a_k && !b_k)
      hipLaunchKernelGGL((gemm_small_kernel<true, false>), sgrid, dim3(256), 0, st, p);
    else if (!a_k && b_k)
      hipLaunchKernelGGL((gemm_small_kernel<false, true>), sgrid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL((gemm_small_kernel<false, false>), sgrid, dim3(256), 0, st, p);
    GE_CHECK_LAUNCH("gemm_small");
    return GE_OK;
  }
  dim3 grid(p.tiles_m * ge_cdiv(N, Tile64::NT), 1, batch);
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
