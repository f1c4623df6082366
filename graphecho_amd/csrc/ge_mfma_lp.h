// Helpers of the fp16-input MFMA conv kernels (ge_mfma_f16.hip): buffer-load helpers, tile configuration, accumulator
// layout, the conv parameter block and the epilogue (bias / skip addend / ReLU / fused BatchNorm moments) -- the same
// definitions as in ge_mfma.hip, kept in a header of their own because folding them into the fp32 kernels cost those
// 4.5 % (codegen of the tuned main loop shifts when its epilogue moves into a shared function).
#pragma once
#include "ge_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Gathers go through buffer loads: the descriptor's hardware range check returns 0 for an out-of-range offset,
// so padding / tile-edge handling needs no branches and all loads of a chunk issue back to back
// (a predicated `ok ? p[i] : 0` makes hipcc emit an exec-mask branch + s_waitcnt vmcnt(0) per load).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define GE_OOB 0xFFFFFFFFu
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
// Element offset -> byte offset, or the out-of-range sentinel when !ok.  The empty asm pins the offset
// computation as unconditional straight-line code; without it hipcc turns the select into a branch around
// the index arithmetic and duplicates the load into both arms, each followed by s_waitcnt vmcnt(0).
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
  int dbg;   // tuning only (GE_CONV_DEBUG): bit 0 = skip the epilogue, bit 1 = run a single K chunk
};

// Epilogue shared by every conv_gemm variant: the wave's TM x TN accumulators (32x32 MFMA layout) -> bias / skip
// addend / ReLU -> NCHW stores, plus the optional fused BatchNorm moments.  m0/n0: tile origin; a_off/b_off: the
// wave's offset inside the tile; tn/wn: n-tile index and the wave's column (slot of the stats partial).
template <class T>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, f32x16 (&acc)[T::TM][T::TN], int g, int m0,
                                              int n0, int a_off, int b_off, int lane, int tn, int wn) {
  // Epilogue: lanes walk n (contiguous x within an image row) -> coalesced 128 B segments.
  const int li = lane & 31, hi = lane >> 5;
  const size_t dplane = (size_t)p.Hd * p.Wd;
#pragma unroll
  for (int i = 0; i < T::TM; ++i) {
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + a_off + i * 32 + acc_row(r, hi);
      bias_r[r] = (p.bias && m < p.M) ? p.bias[g * p.M + m] : 0.f;
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
      float* dst = p.dst + dbase;
      if (p.addend) {   // gradient of a skip connection: issue the 16 loads before the dependent stores
        const float* add = p.addend + dbase;
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
