// fp16 ACTIVATION STORAGE for the conv -> BatchNorm -> ReLU (-> 2x2 max-pool) stacks of the VGG16 backbone
// (models/fpnseg.py:18-166 of the reference, the backbone train_cardiac_uda.py:73 builds): BASELINE.json config 5,
// "fp16 MFMA conv path + fp32 Sinkhorn".
//
// Inside such a stack every activation and every activation gradient lives in HBM as fp16 in the channel-blocked layout
//     h[b][c / 32][y][x][c % 32]            ("blocked-32": the 32 channels of a pixel are 64 contiguous bytes)
// so that (1) the HBM-bound passes between two convolutions (BatchNorm apply, its backward, pooling) move half the bytes,
// (2) a conv's operand tile is a run of whole 64-byte pixels: it goes global -> LDS with buffer_load_dwordx4 ... lds
// (no staging registers, no conversion, no ds_write), and (3) a 3x3 conv stages ONE input patch per 32-channel chunk
// and takes all nine taps from it (the fp32-storage f16 kernels of ge_mfma_f16.hip re-gather the input once per tap,
// 4 bytes per element, one load instruction per element).
// Statistics, affine parameters, weights (fp32 masters, fp16 packed copies), weight gradients and every reduction stay
// fp32.  Gradients inside the stack carry a constant loss scale (the caller multiplies when it enters the stack's
// backward and the weight / affine gradient kernels divide it out), because fp16 has 5 exponent bits.
//
// Kernels:
//   h_conv3x3_kernel       forward and data gradient (flipped taps, transposed packed weights), 3x3 / stride 1 / pad 1:
//                          workgroup tile = 128 pixels (a TR x TC rectangle of one image) x 128 output channels, or
//                          256 x 64; four waves of 64 x 64; v_mfma_f32_32x32x16_f16, fp32 accumulation; epilogue: bias,
//                          fp16 store, per-wave BatchNorm moments (count, mean, M2) of the fp32 results.
//   h_wgrad3x3_kernel      weight gradient: K = pixels.  Both operands are [pixel][32 channel] LDS images read with
//                          ds_read_b64_tr_b16 (the hardware transpose read of gfx950: a lane gets 4 consecutive PIXELS of
//                          its channel), so a tap is a pixel offset and never an alignment problem; nine 32 x 32
//                          accumulators per wave (one per tap); K split over workgroups, fp32 slabs, one reduce.
//   bnh_* / poolh_* / cast kernels: the HBM-bound passes in the blocked layout, 16 bytes per thread and access.
#include "ge_common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define LDS_AS __attribute__((address_space(3)))

#define H_OOB 0xFFFFFFFFu

// fp32 -> fp16 store that saturates instead of overflowing to inf (a loss-scaled gradient beyond fp16's range)
__device__ __forceinline__ _Float16 h_sat(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }

// 16 bytes per lane straight into LDS (lane l lands at lds_addr + 16 l); an out-of-range offset writes zeros.
__device__ __forceinline__ void h_dma16(u32x4 rs, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void h_dma_wait() {      // at most N of this wave's vector-memory operations still in flight
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
__device__ __forceinline__ u32x4 h_rsrc(const void* p, uint32_t bytes) {
  const unsigned long long ad = (unsigned long long)p;
  u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)ad);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(ad >> 32) & 0xFFFFu);
  rs.z = __builtin_amdgcn_readfirstlane(bytes);
  rs.w = 0x00020000u;
  return rs;
}
__device__ __forceinline__ int h_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__device__ __forceinline__ int h_acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// =========================================================================================
// forward / data gradient
// =========================================================================================
struct HConvParams {
  const void* x;      // blocked fp16 [B][C/32][H][W][32]
  const void* wp;     // fp16 [9][M][C]   (ge_conv2d_f16_pack_weight: forward [t][co][ci], data gradient [t][ci][co])
  const float* bias;  // [M] or null
  void* y;            // blocked fp16 [B][M/32][H][W][32], or (F32OUT) fp32 NCHW [B][M][H][W]
  const float* addend;   // F32OUT only: optional fp32 NCHW tensor added to the result (gradient of a skip connection)
  float out_scale;    // F32OUT only: the result is multiplied by this (1 / loss scale in a data gradient)
  const float* hs;    // F32OUT only, nullable: device-resident loss scale; the result is also multiplied by hs[1]
  float* stats;       // [M][parts][3] or null
  int B, C, M, H, W;
  int TR, TC, tcs;    // tile rectangle, TC = 1 << tcs
  int tiles_x, tiles_y, tiles_m;
  int flip;           // data gradient: tap t of the patch meets weight tap 8 - t
  int stats_parts;
  uint32_t x_bytes, wp_bytes;
  int dbg;            // tuning only (GE_H_DBG, wrong results): 1 = no operand DMA after the prologue, 2 = no epilogue, 4 = no MFMA phase
};

// PW x CW waves of 64 pixels x 64 output channels each.  LDS: two patch buffers of PPW * 4 KB (pixel-major, 64 B per pixel,
// the four 16-byte segments of a pixel XOR-swizzled by (pixel >> 2) & 3 on the SOURCE side so that the 16-lane groups of
// a ds_read_b128 hit 16 different 16-byte slots) + three weight stages of MT * 64 B (same swizzle by output channel).
// F32OUT: the result leaves the fp16 domain -- fp32 NCHW stores (operands in MFMA order weights x pixels, so that the 32 lanes
// of a half-wave hold 32 consecutive pixels of one output channel: 128-byte runs), times out_scale, plus the optional addend.
// WPX: 32-pixel blocks per wave (2: 64 x 64 wave tiles; 4: 128 pixels x 64 channels -- a 256 x 128 workgroup tile moves half the
// weight bytes per FLOP through the LDS-DMA path, which the phase split shows to be what the 128 x 128 form waits for, needs
// six fragment reads for eight MFMAs instead of four for four, and meets at a barrier after 16 MFMAs instead of 8)
template <int PW, int CW, int WPX, bool F32OUT>
__global__ __launch_bounds__(256, 2) void h_conv3x3_kernel(HConvParams p) {
  constexpr int NT = PW * WPX * 32, MT = CW * 64;
  constexpr int AI = MT / 64;                  // weight DMA instructions per wave and step (1 KB each)
  constexpr int PPW = NT == 128 ? 5 : 7;       // patch DMA instructions per wave and chunk (upper bound over tile shapes)
  constexpr uint32_t PBUF = PPW * 4096u, ASTAGE = MT * 64u;
  static_assert(PW * CW == 4, "four waves");
  extern __shared__ __attribute__((aligned(16))) char hsmem[];
  const uint32_t lds0 = (uint32_t)(size_t)(LDS_AS char*)hsmem;
  const uint32_t lds_patch = lds0, lds_a = lds0 + 2 * PBUF;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int wp_ = wave % PW, wc = wave / PW;
  const int lid = h_xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid % p.tiles_m, tp = lid / p.tiles_m;
  const int tiles_img = p.tiles_x * p.tiles_y;
  const int b = tp / tiles_img, trem = tp - b * tiles_img;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * p.TR, x0 = tx * p.TC, m0 = tm * MT;
  const int PC = p.TC + 2, PR = p.TR + 2;
  const int CBK = p.C >> 5;
  const uint32_t plane_bytes = (uint32_t)p.H * p.W * 64u;

  const u32x4 xrs = h_rsrc(p.x, p.x_bytes), wrs = h_rsrc(p.wp, p.wp_bytes);

  // ---- patch DMA: piece k of this wave covers slots ((k * 4 + wave) * 64 + lane) ----
  uint32_t poff[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int slot = (k * 4 + wave) * 64 + lane;
    const int pix = slot >> 2, sl = slot & 3;
    const int prow = pix / PC, pcol = pix - prow * PC;
    const int seg = sl ^ ((pix >> 2) & 3);
    const int iy = y0 - 1 + prow, ix = x0 - 1 + pcol;
    const bool ok = prow < PR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    uint32_t off = (((uint32_t)b * CBK * p.H + iy) * p.W + ix) * 64u + seg * 16u;
    asm volatile("" : "+v"(off));
    poff[k] = ok ? off : H_OOB;
  }
  // ---- weight DMA: instruction e of this wave covers slots ((e * 4 + wave) * 64 + lane) of the MT x 4 stage ----
  uint32_t aoff[AI];
#pragma unroll
  for (int e = 0; e < AI; ++e) {
    const int slot = (e * 4 + wave) * 64 + lane;
    const int m = slot >> 2, sl = slot & 3;
    const int seg = sl ^ ((m >> 2) & 3);
    aoff[e] = ((uint32_t)(m0 + m) * p.C + seg * 8) * 2u;
  }
  const uint32_t tap_bytes = (uint32_t)p.M * p.C * 2u;
  auto issue_a = [&](int cb, int t_static, int stage) {      // weights of tap t_static of chunk cb
    const int tw = p.flip ? 8 - t_static : t_static;
    const uint32_t uni = (uint32_t)tw * tap_bytes + (uint32_t)cb * 64u;
    const bool ok = cb < CBK;
#pragma unroll
    for (int e = 0; e < AI; ++e) {
      uint32_t off = aoff[e] + uni;
      asm volatile("" : "+v"(off));
      h_dma16(wrs, lds_a + (uint32_t)stage * ASTAGE + (uint32_t)(e * 4 + wave) * 1024u, ok ? off : H_OOB);
    }
  };
  auto issue_patch = [&](int cb, int k, int buf) {
    const bool ok = cb < CBK;
    const uint32_t add = ok ? (uint32_t)cb * plane_bytes : H_OOB;
    h_dma16(xrs, lds_patch + (uint32_t)buf * PBUF + (uint32_t)(k * 4 + wave) * 1024u,
            __builtin_elementwise_add_sat(poff[k], add));
  };

  // ---- fragment addresses ----
  int pbase[WPX];      // patch pixel index (tap (0, 0)) of this lane's pixel in pixel block i
#pragma unroll
  for (int i = 0; i < WPX; ++i) {
    const int n = wp_ * (WPX * 32) + i * 32 + li;
    pbase[i] = (n >> p.tcs) * PC + (n & (p.TC - 1));
  }
  uint32_t wfrag[2][2];   // [j][ks]: byte offset inside a weight stage
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = wc * 64 + j * 32 + li;
    const int f = (m >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wfrag[j][ks] = (uint32_t)m * 64u + (uint32_t)(((ks * 2 + hi) ^ f) * 16);
  }

  f32x16 acc[WPX][2];
#pragma unroll
  for (int i = 0; i < WPX; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto mma = [&](int t_static, int pbuf, int stage) {
    const int tapoff = (t_static / 3) * PC + (t_static % 3);
    const char* pb = hsmem + (uint32_t)pbuf * PBUF;
    const char* ab = hsmem + 2 * PBUF + (uint32_t)stage * ASTAGE;
    uint32_t pa[WPX][2];
#pragma unroll
    for (int i = 0; i < WPX; ++i) {
      const int pix = pbase[i] + tapoff;
      const int s0 = (hi ^ (pix >> 2)) & 3;
      pa[i][0] = (uint32_t)pix * 64u + (uint32_t)s0 * 16u;
      pa[i][1] = pa[i][0] ^ 32u;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 fp[WPX], fw[2];
#pragma unroll
      for (int i = 0; i < WPX; ++i) fp[i] = *(const half8*)(pb + pa[i][ks]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fw[j] = *(const half8*)(ab + wfrag[j][ks]);
#pragma unroll
      for (int i = 0; i < WPX; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = F32OUT ? __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j], fp[i], acc[i][j], 0, 0, 0)
                             : __builtin_amdgcn_mfma_f32_32x32x16_f16(fp[i], fw[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- prologue ----
#pragma unroll
  for (int k = 0; k < PPW; ++k) issue_patch(0, k, 0);
  issue_a(0, 0, 0);
  issue_a(0, 1, 1);
  h_dma_wait<0>();
  __syncthreads();

  int st_rd = 0, st_wr = 2;
  for (int cb = 0; cb < CBK; ++cb) {
    const int pbuf = cb & 1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (!(p.dbg & 1)) {
        issue_a(cb + (t + 2 >= 9 ? 1 : 0), (t + 2) % 9, st_wr);
        if (t < PPW) issue_patch(cb + 1, t, pbuf ^ 1);
      }
      if (!(p.dbg & 4)) mma(t, pbuf, st_rd);
      // the weights of step s + 1 have landed: behind them only this step's DMAs and the previous step's patch piece
      if (t == 0) h_dma_wait<AI + 1>();
      else if (t < PPW) h_dma_wait<AI + 2>();
      else if (t == PPW) h_dma_wait<AI + 1>();
      else h_dma_wait<AI>();
      __syncthreads();
      st_rd = st_rd == 2 ? 0 : st_rd + 1;
      st_wr = st_wr == 2 ? 0 : st_wr + 1;
    }
  }
  h_dma_wait<0>();
  if (p.dbg & 2) return;

  if constexpr (F32OUT) {
    // ---- epilogue, fp32 NCHW: rows of an accumulator are output channels, its column is the lane's pixel ----
    float* yo = (float*)p.y;
    const size_t HWs = (size_t)p.H * p.W;
    const float osc = p.out_scale * (p.hs ? p.hs[1] : 1.f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float bias_r[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bias_r[r] = p.bias ? p.bias[m0 + wc * 64 + j * 32 + h_acc_row(r, hi)] : 0.f;
#pragma unroll
      for (int i = 0; i < WPX; ++i) {
        const int n = wp_ * (WPX * 32) + i * 32 + li;
        const int oy = y0 + (n >> p.tcs), ox = x0 + (n & (p.TC - 1));
        const size_t base = ((size_t)b * p.M + m0 + wc * 64 + j * 32) * HWs + (size_t)oy * p.W + ox;
        if (p.addend) {
          float addv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) addv[r] = p.addend[base + (size_t)h_acc_row(r, hi) * HWs];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            yo[base + (size_t)h_acc_row(r, hi) * HWs] = (acc[i][j][r] + bias_r[r]) * osc + addv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) yo[base + (size_t)h_acc_row(r, hi) * HWs] = (acc[i][j][r] + bias_r[r]) * osc;
        }
      }
      if (p.stats)
#pragma unroll
      for (int ih = 0; ih < WPX / 2; ++ih) {
        // per-row moments over 64 of this wave's pixels: reduce-scatter butterfly over the 32 lanes of a half-wave
        float v[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t0 = acc[2 * ih][j][r] + bias_r[r], t1 = acc[2 * ih + 1][j][r] + bias_r[r];
          v[r] = t0 + t1;
          v[16 + r] = t0 * t0 + t1 * t1;
        }
#pragma unroll
        for (int hh = 16; hh > 0; hh >>= 1) {
          const bool up = (li & hh) != 0;
#pragma unroll
          for (int k = 0; k < hh; ++k) {
            const float send = up ? v[k] : v[k + hh];
            const float keep = up ? v[k + hh] : v[k];
            v[k] = keep + __shfl_xor(send, hh, 64);
          }
        }
        const float qsum = __shfl_down(v[0], 16, 64);
        if (li < 16) {
          const int co = m0 + wc * 64 + j * 32 + h_acc_row(li, hi);
          const float mean = v[0] * (1.f / 64.f);
          float* o3 = p.stats + ((size_t)co * p.stats_parts + (size_t)tp * (NT / 64) + wp_ * (WPX / 2) + ih) * 3;
          o3[0] = 64.f;
          o3[1] = mean;
          o3[2] = fmaxf(qsum - v[0] * mean, 0.f);
        }
      }
    }
    return;
  }
  // ---- epilogue: rows of an accumulator are pixels, its column is the lane's output channel ----
  _Float16* yo = (_Float16*)p.y;
  const int MB = p.M >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = m0 + wc * 64 + j * 32 + li;
    const float bv = p.bias ? p.bias[co] : 0.f;
    const size_t cbase = ((size_t)b * MB + (co >> 5)) * ((size_t)p.H * p.W) * 32 + (co & 31);
#pragma unroll
    for (int ih = 0; ih < WPX / 2; ++ih) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = 2 * ih + i2;
          const int n = wp_ * (WPX * 32) + i * 32 + h_acc_row(r, hi);
          const int oy = y0 + (n >> p.tcs), ox = x0 + (n & (p.TC - 1));
          const float v = acc[i][j][r] + bv;
          sv += v;
          qv += v * v;
          yo[cbase + ((size_t)oy * p.W + ox) * 32] = h_sat(v);
        }
      if (p.stats) {
        sv += __shfl_xor(sv, 32, 64);
        qv += __shfl_xor(qv, 32, 64);
        if (hi == 0) {
          const float mean = sv * (1.f / 64.f);
          float* o3 = p.stats + ((size_t)co * p.stats_parts + (size_t)tp * (NT / 64) + wp_ * (WPX / 2) + ih) * 3;
          o3[0] = 64.f;
          o3[1] = mean;
          o3[2] = fmaxf(qv - sv * mean, 0.f);
        }
      }
    }
  }
}

// =========================================================================================
// weight gradient
// =========================================================================================
struct HWgradParams {
  const void* x;     // blocked fp16 [B][C/32][H][W][32]
  const void* dz;    // blocked fp16 [B][M/32][H][W][32]
  float* slab;       // [slabs][9][M][C]
  int B, C, M, H, W, TR;
  int tiles_x, tiles_y, ntiles, tiles_per_split, tiles_m, cblocks;
  uint32_t x_bytes, dz_bytes;
};

// CB output-channel blocks of 32 per workgroup (4: wave w owns block w; 2: waves 0/1 own the blocks over the first half
// of every tile's pixels, waves 2/3 over the second half and write slabs of their own).  TC = 1 << TCS columns per tile,
// 128 pixels per tile.
// SB: ONE LDS buffer (load - barrier - MFMAs - barrier) and two workgroups per CU that cover for each other, instead of two
// buffers and one workgroup per CU (one wave per SIMD has nobody to hide its fragment-read latency behind)
template <int CB, int TCS, bool SB>
__global__ __launch_bounds__(256, SB ? 2 : 1) void h_wgrad3x3_kernel(HWgradParams p) {
  constexpr int TC = 1 << TCS, PC = TC + 2;
  constexpr int PPW = 5;
  constexpr uint32_t DZB = CB * 8192u, PBUF = PPW * 4096u, BUF = DZB + PBUF;
  constexpr int DI = 2 * CB;        // dZ DMA instructions per wave and tile
  constexpr int KSTEPS = CB == 4 ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) char hsmem[];
  const uint32_t lds0 = (uint32_t)(size_t)(LDS_AS char*)hsmem;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  // XCD-aware order: the tiles_m * cblocks workgroups of one K split (they share its dz tiles and x patches) are consecutive
  // logical ids = one XCD's L2 (measured before: 606 MB of fabric reads per launch for 201 MB of operands)
  int bid = h_xcd_remap(blockIdx.x, gridDim.x);
  const int tm = bid % p.tiles_m;
  bid /= p.tiles_m;
  const int cb = bid % p.cblocks, sp = bid / p.cblocks;
  const int t_beg = sp * p.tiles_per_split, t_end = min(t_beg + p.tiles_per_split, p.ntiles);
  const int CBK = p.C >> 5, MB = p.M >> 5;
  const int PR = p.TR + 2;
  const uint32_t plane_bytes = (uint32_t)p.H * p.W * 64u;
  const u32x4 xrs = h_rsrc(p.x, p.x_bytes), drs = h_rsrc(p.dz, p.dz_bytes);

  // dZ DMA: instruction q = k * 4 + wave (k < DI) fills slots q * 64 + lane of the CB x 512-slot image; block q / 8
  uint32_t dlane[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int slot = ((k * 4 + wave) & 7) * 64 + lane;
    const int n = slot >> 2, seg = slot & 3;
    dlane[k] = ((uint32_t)(n >> TCS) * p.W + (n & (TC - 1))) * 64u + seg * 16u;
  }
  // patch DMA: piece k covers slots (k * 4 + wave) * 64 + lane; (prow, pcol, seg) per piece
  int pr_[PPW], pc_[PPW];
  uint32_t pl_[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int slot = (k * 4 + wave) * 64 + lane;
    const int pix = slot >> 2;
    pr_[k] = pix / PC;
    pc_[k] = pix - pr_[k] * PC;
    pl_[k] = (uint32_t)(slot & 3) * 16u;
  }
  auto issue = [&](int tile, int buf) {
    const int tiles_img = p.tiles_x * p.tiles_y;
    const int b = tile / tiles_img, trem = tile - b * tiles_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * p.TR, x0 = tx * TC;
    const uint32_t base = lds0 + (uint32_t)buf * BUF;
    const uint32_t duni = (((uint32_t)b * MB + tm * CB) * p.H + y0) * p.W * 64u + (uint32_t)x0 * 64u;
#pragma unroll
    for (int k = 0; k < DI; ++k) {
      const int q = k * 4 + wave;
      uint32_t off = dlane[k & 1] + duni + (uint32_t)(q >> 3) * plane_bytes;
      asm volatile("" : "+v"(off));
      h_dma16(drs, base + (uint32_t)q * 1024u, off);
    }
    const uint32_t xuni = ((uint32_t)b * CBK + cb) * plane_bytes;
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int iy = y0 - 1 + pr_[k], ix = x0 - 1 + pc_[k];
      const bool ok = pr_[k] < PR && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      uint32_t off = xuni + ((uint32_t)iy * p.W + ix) * 64u + pl_[k];
      asm volatile("" : "+v"(off));
      h_dma16(xrs, base + DZB + (uint32_t)(k * 4 + wave) * 1024u, ok ? off : H_OOB);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transpose-read lane address: 16-lane group g, lane i = 4 j + q of it supplies (pixel j of the group's 4, channels 4 q ..)
  const int g = lane >> 4, i16 = lane & 15;
  const uint32_t lane_tr = (uint32_t)((g >> 1) * 8 + (i16 >> 2)) * 64u + (uint32_t)(16 * (g & 1) + 4 * (i16 & 3)) * 2u;
  const int myblk = CB == 4 ? wave : (wave & 1);
  const int k0 = CB == 4 ? 0 : (wave >> 1) * 4;

  if (!SB && t_beg < t_end) issue(t_beg, 0);
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int buf = SB ? 0 : (tile - t_beg) & 1;
    if (SB) {
      __syncthreads();          // everybody is done with the previous tile's image
      issue(tile, 0);
    }
    h_dma_wait<0>();
    __syncthreads();
    if (!SB && tile + 1 < t_end) issue(tile + 1, buf ^ 1);
    // k0 (0 or 4 k-steps = 64 pixels, a multiple of every TC) is wave-uniform; everything else below is an immediate
    const LDS_AS char* da = (const LDS_AS char*)(hsmem + (uint32_t)buf * BUF + (uint32_t)myblk * 8192u + lane_tr +
                                                 (uint32_t)k0 * 1024u);
    const LDS_AS char* xa = (const LDS_AS char*)(hsmem + (uint32_t)buf * BUF + DZB + lane_tr +
                                                 (uint32_t)(k0 * 16 + 2 * ((k0 * 16) >> TCS)) * 64u);
#pragma unroll
    for (int kq = 0; kq < KSTEPS; ++kq) {
      const fp16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(da + kq * 1024));
      const fp16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(da + kq * 1024 + 256));
      half8 fa;
      {
        const half4 h0 = __builtin_bit_cast(half4, a0), h1 = __builtin_bit_cast(half4, a1);
        fa = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int po = (kq * 16 + 2 * ((kq * 16) >> TCS) + (t / 3) * PC + (t % 3)) * 64;
        const fp16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(xa + po));
        const fp16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)(xa + po + 256));
        const half4 h0 = __builtin_bit_cast(half4, b0), h1 = __builtin_bit_cast(half4, b1);
        const half8 fb = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[t], 0, 0, 0);
      }
    }
  }

  // slab [slab index][tap][co][ci]: rows of an accumulator are output channels, its column is the lane's input channel
  const int slab_i = CB == 4 ? sp : sp * 2 + (wave >> 1);
  float* out = p.slab + (size_t)slab_i * 9 * p.M * p.C;
  const int co0 = (tm * CB + myblk) * 32, ci = cb * 32 + li;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[((size_t)t * p.M + co0 + h_acc_row(r, hi)) * p.C + ci] = acc[t][r];
}

// stage 1 of a long reduction: part[g][t][co][ci] = sum of the slabs of group g (gridDim.y groups of `per` slabs)
__global__ __launch_bounds__(256) void h_slab_group_kernel(const float* __restrict__ slab, float* __restrict__ part, int MC,
                                                           int nslabs, int per) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= MC) return;
  const int k0 = blockIdx.y * per, k1 = min(k0 + per, nslabs);
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float* sp = slab + (size_t)k * 9 * MC + idx;
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] += sp[(size_t)t * MC];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) part[((size_t)blockIdx.y * 9 + t) * MC + idx] = s[t];
}
// dw[co][ci][t] (+)= scale * sum over slabs of slab[s][t][co][ci]
__global__ __launch_bounds__(256) void h_slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M,
                                                            int C, int nslabs, float scale, int accumulate,
                                                            const float* __restrict__ hs) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // (co, ci)
  if (idx >= M * C) return;
  if (hs) scale *= hs[1];
  const size_t mc = (size_t)M * C;
  float s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = 0.f;
  for (int k = 0; k < nslabs; ++k) {
    const float* sp = slab + (size_t)k * 9 * mc + idx;
#pragma unroll
    for (int t = 0; t < 9; ++t) s[t] += sp[(size_t)t * mc];
  }
  float* o = dw + (size_t)idx * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) o[t] = (accumulate ? o[t] : 0.f) + s[t] * scale;
}

// =========================================================================================
// HBM-bound passes in the blocked layout.  A "vector" is 16 bytes = 8 channels of one pixel; vector v of a plane
// (b, channel block) covers pixel v >> 2, channels (v & 3) * 8 .. + 7 of the block.
// =========================================================================================
// fp32 NCHW -> blocked fp16 (times `scale`); lanes walk pixels (coalesced fp32 reads)
// hs (nullable): the device-resident loss scale {scale, 1 / scale, bits of the largest |x| seen since the last update}: the
// cast multiplies by hs[0] (times `scale`) and folds this tensor's largest magnitude into hs[2] (ge_h_scale_update)
// addend (nullable): a blocked fp16 tensor added to the scaled values IN fp32 before the one saturating rounding (the gradient
// of a tensor with two consumers, one of them outside the fp16 domain: an fp16 + fp16 add could overflow to inf)
__global__ __launch_bounds__(256) void h_from_f32_kernel(const float* __restrict__ x, _Float16* __restrict__ h, int C,
                                                         int HW, float scale, float* __restrict__ hs,
                                                         const _Float16* __restrict__ addend) {
  const int CBK = C >> 5;
  if (hs) scale *= hs[0];
  float amax = 0.f;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK;
  const float* xp = x + ((size_t)b * C + cblk * 32) * HW;
  half8* hp = (half8*)(h + (size_t)pl * HW * 32);
  const half8* ap = addend ? (const half8*)(addend + (size_t)pl * HW * 32) : nullptr;
  // (a 4-pixels-per-thread form with 16-byte fp32 loads was measured: 3.4 instead of 4.5 TB/s -- its fp16 stores are 16-byte pieces
  // 256 bytes apart; here the four stores of a thread complete one 64-byte pixel)
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      half8 v, av;
      if (ap) av = ap[(size_t)pix * 4 + cg];
#pragma unroll
      for (int e = 0; e < 8; ++e) {      // saturating: a scaled gradient beyond fp16's range must not become inf
        const float xv = xp[(size_t)(cg * 8 + e) * HW + pix];
        amax = fmaxf(amax, fabsf(xv));
        v[e] = h_sat(ap ? fmaf(xv, scale, (float)av[e]) : xv * scale);
      }
      hp[(size_t)pix * 4 + cg] = v;
    }
  }
  if (hs) {
    // one atomic per WORKGROUP, and only where it can change the record (the record is read first: a stale read only costs an
    // atomic that does nothing); one per wave, unconditionally, doubled this kernel's time -- 1e5 atomics on one address
    __shared__ float wmax[4];
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
      const unsigned bits = __float_as_uint(amax);
      if (amax < 3.0e38f && bits > __builtin_nontemporal_load((const unsigned*)(hs + 2))) atomicMax((unsigned*)(hs + 2), bits);
    }
  }
}
// blocked fp16 -> fp32 NCHW (times `scale`)
__global__ __launch_bounds__(256) void h_to_f32_kernel(const _Float16* __restrict__ h, float* __restrict__ x, int C, int HW,
                                                       float scale, const float* __restrict__ hs) {
  const int CBK = C >> 5;
  if (hs) scale *= hs[1];
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK;
  float* xp = x + ((size_t)b * C + cblk * 32) * HW;
  const half8* hp = (const half8*)(h + (size_t)pl * HW * 32);
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += gridDim.x * 256) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const half8 v = hp[(size_t)pix * 4 + cg];
#pragma unroll
      for (int e = 0; e < 8; ++e) xp[(size_t)(cg * 8 + e) * HW + pix] = (float)v[e] * scale;
    }
  }
}

// a = relu?(z * sc + sh), per-channel sc = gamma * invstd, sh = beta - mean * sc
__global__ __launch_bounds__(256) void bnh_apply_kernel(const half8* __restrict__ z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half8* __restrict__ a, int C, int HW,
                                                        int relu) {
  const int CBK = C >> 5;
  const int pl = blockIdx.y, cblk = pl % CBK;
  const int c0 = cblk * 32 + (threadIdx.x & 3) * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (gamma ? gamma[c0 + e] : 1.f) * invstd[c0 + e];
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mean[c0 + e] * sc[e];
  }
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    const half8 zi = z[base + v];
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf((float)zi[e], sc[e], sh[e]);
      if (relu) t = fmaxf(t, 0.f);
      o[e] = (_Float16)t;
    }
    a[base + v] = o;
  }
}

// MODE 0: partial[c][blk] = (sum g, sum g * (z - mean) * invstd), g = da masked by the recomputed ReLU;
// MODE 1: partial[c][blk] = (sum da, 0)  (bias gradient of the conv in front: a plain channel sum)
template <int MODE>
__global__ __launch_bounds__(256) void bnh_bwd_partial_kernel(const half8* __restrict__ da, const half8* __restrict__ z,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int relu, float* __restrict__ partial, int C, int HW, int S,
                                                              int NB) {
  __shared__ float red[256 * 17];
  const int CBK = C >> 5;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK;
  const int cg = threadIdx.x & 3, c0 = cblk * 32 + cg * 8;
  float sc[8], sh[8], mu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (MODE == 0) {
      mu[e] = mean[c0 + e];
      sc[e] = (gamma ? gamma[c0 + e] : 1.f) * invstd[c0 + e];
      sh[e] = (beta ? beta[c0 + e] : 0.f) - mu[e] * sc[e];
    } else {
      mu[e] = sc[e] = sh[e] = 0.f;
    }
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4, per = (nv + S - 1) / S;
  const int v0 = blockIdx.x * per, v1 = min(v0 + per, nv);
  for (int v = v0 + threadIdx.x; v < v1; v += 256) {
    const half8 g8 = da[base + v];
    if (MODE == 0) {
      const half8 z8 = z[base + v];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zf = (float)z8[e];
        float gf = (float)g8[e];
        if (relu) gf = fmaf(zf, sc[e], sh[e]) > 0.f ? gf : 0.f;
        s1[e] += gf;
        s2[e] += gf * (zf - mu[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[e] += (float)g8[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x * 17 + e] = s1[e];
    red[threadIdx.x * 17 + 8 + e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {      // thread (ch, which): ch = 0..31, which = 0 / 1
    const int ch = threadIdx.x & 31, which = threadIdx.x >> 5;
    const int cgi = ch >> 3, e = ch & 7;
    float t = 0.f;
    for (int k = 0; k < 64; ++k) t += red[(k * 4 + cgi) * 17 + which * 8 + e];
    const int c = cblk * 32 + ch;
    if (MODE == 0 && which == 1) t *= invstd[c];
    partial[((size_t)c * NB + (size_t)b * S + blockIdx.x) * 2 + which] = t;
  }
}
// sums[c] = (sum g, sum g xhat) over the NB partials times inv_scale, i.e. in TRUE units (what SyncBN all-reduces: the ranks'
// loss scales need not agree); dgamma / dbeta (+)= the same
__global__ __launch_bounds__(256) void bnh_bwd_finalize_kernel(const float* __restrict__ partial, int NB, int C,
                                                               float* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int accumulate, float inv_scale,
                                                               const float* __restrict__ hs) {
  if (hs) inv_scale *= hs[1];
  // one wave per channel: lanes stride over the channel's NB pairs (fixed order -> bit-reproducible), DPP wave sum
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  const float2* pp = (const float2*)partial + (size_t)c * NB;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < NB; i += 64) {
    const float2 v = pp[i];
    s1 += v.x;
    s2 += v.y;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    s1 *= inv_scale;
    s2 *= inv_scale;
    if (sums) {
      sums[c * 2 + 0] = s1;
      sums[c * 2 + 1] = s2;
    }
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
  }
}
// dz = gamma * invstd * (g - scale * (s1 / n + xhat * s2 / n)): g is loss-scaled, the sums are in true units
__global__ __launch_bounds__(256) void bnh_bwd_apply_kernel(const half8* __restrict__ da, const half8* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int relu, const float* __restrict__ sums, float inv_count,
                                                            half8* __restrict__ dz, int C, int HW, float scale,
                                                            const float* __restrict__ hs) {
  if (hs) scale *= hs[0];
  inv_count *= scale;
  const int CBK = C >> 5;
  const int pl = blockIdx.y, cblk = pl % CBK;
  const int c0 = cblk * 32 + (threadIdx.x & 3) * 8;
  float sc[8], sh[8], mu[8], kk[8], a1[8], a2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float is = invstd[c0 + e], gm = gamma ? gamma[c0 + e] : 1.f;
    mu[e] = mean[c0 + e];
    sc[e] = gm * is;
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu[e] * sc[e];
    kk[e] = gm * is;
    a1[e] = sums[(c0 + e) * 2] * inv_count;
    a2[e] = sums[(c0 + e) * 2 + 1] * inv_count * is;
  }
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    const half8 g8 = da[base + v], z8 = z[base + v];
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float zf = (float)z8[e];
      float gf = (float)g8[e];
      if (relu) gf = fmaf(zf, sc[e], sh[e]) > 0.f ? gf : 0.f;
      o[e] = h_sat(kk[e] * (gf - a1[e] - (zf - mu[e]) * a2[e]));
    }
    dz[base + v] = o;
  }
}

// ---- GroupNorm (+ ReLU) with 8 channels per group on blocked fp16 tensors (the discriminator towers, nn.GroupNorm(32, 256),
// fpnseg.py:465): a group is exactly one 16-byte vector of a pixel, so a thread's (scale, shift) pair is one (sample, group)
// statistic times its 8 channels' affine parameters.  The statistics come from the conv epilogue's per-64-pixel moments.
// mean / invstd: [B][C / 8].
__global__ __launch_bounds__(256) void gnh_finalize_kernel(const float* __restrict__ stats, int parts, int per, int C, float eps,
                                                           float* __restrict__ mean, float* __restrict__ invstd, int BG) {
  // one wave per (sample, group): the group's 8 channels x `per` triples (count 64 each) of this sample
  const int bg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (bg >= BG) return;
  const int G = C >> 3, b = bg / G, g = bg - b * G;
  const int n = 8 * per;
  float sm = 0.f;
  for (int i = lane; i < n; i += 64) {
    const int c = g * 8 + i / per, part = b * per + i % per;
    sm += stats[((size_t)c * parts + part) * 3 + 1];
  }
  const float mu = wave_sum(sm) / (float)n;
  float m2 = 0.f;
  for (int i = lane; i < n; i += 64) {
    const int c = g * 8 + i / per, part = b * per + i % per;
    const float* t = stats + ((size_t)c * parts + part) * 3;
    const float d = t[1] - mu;
    m2 += t[2] + 64.f * d * d;
  }
  m2 = wave_sum(m2);
  if (lane == 0) {
    mean[bg] = mu;
    invstd[bg] = rsqrtf(m2 / (64.f * (float)n) + eps);
  }
}
__global__ __launch_bounds__(256) void gnh_apply_kernel(const half8* __restrict__ z, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, half8* __restrict__ a, int C, int HW,
                                                        int relu) {
  const int CBK = C >> 5, G = C >> 3;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK, cg = threadIdx.x & 3;
  const int c0 = cblk * 32 + cg * 8, bg = b * G + cblk * 4 + cg;
  const float mu = mean[bg], is = invstd[bg];
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (gamma ? gamma[c0 + e] : 1.f) * is;
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu * sc[e];
  }
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    const half8 zi = z[base + v];
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = fmaf((float)zi[e], sc[e], sh[e]);
      if (relu) t = fmaxf(t, 0.f);
      o[e] = (_Float16)t;
    }
    a[base + v] = o;
  }
}
// partial[b][c][slice] = (sum g, sum g xhat), g = da masked by the recomputed ReLU, xhat = (z - mean_bg) invstd_bg
__global__ __launch_bounds__(256) void gnh_bwd_partial_kernel(const half8* __restrict__ da, const half8* __restrict__ z,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int relu, float* __restrict__ partial, int C, int HW, int S) {
  __shared__ float red[256 * 17];
  const int CBK = C >> 5, G = C >> 3;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK, cg = threadIdx.x & 3;
  const int c0 = cblk * 32 + cg * 8, bg = b * G + cblk * 4 + cg;
  const float mu = mean[bg], is = invstd[bg];
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (gamma ? gamma[c0 + e] : 1.f) * is;
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu * sc[e];
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4, per = (nv + S - 1) / S;
  const int v0 = blockIdx.x * per, v1 = min(v0 + per, nv);
  for (int v = v0 + threadIdx.x; v < v1; v += 256) {
    const half8 g8 = da[base + v], z8 = z[base + v];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float zf = (float)z8[e];
      float gf = (float)g8[e];
      if (relu) gf = fmaf(zf, sc[e], sh[e]) > 0.f ? gf : 0.f;
      s1[e] += gf;
      s2[e] += gf * (zf - mu);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x * 17 + e] = s1[e];
    red[threadIdx.x * 17 + 8 + e] = s2[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int ch = threadIdx.x & 31, which = threadIdx.x >> 5;
    const int cgi = ch >> 3, e = ch & 7;
    float t = 0.f;
    for (int k = 0; k < 64; ++k) t += red[(k * 4 + cgi) * 17 + which * 8 + e];
    if (which == 1) t *= invstd[b * G + cblk * 4 + cgi];
    partial[(((size_t)b * C + cblk * 32 + ch) * S + blockIdx.x) * 2 + which] = t;
  }
}
// sums[b][g] = (sum over the group's channels of gamma_c * sum g, of gamma_c * sum g xhat); one wave per (sample, group)
__global__ __launch_bounds__(256) void gnh_bwd_group_kernel(const float* __restrict__ partial, const float* __restrict__ gamma, int C,
                                                            int S, float* __restrict__ sums, int BG) {
  const int bg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (bg >= BG) return;
  const int G = C >> 3, b = bg / G, g = bg - b * G;
  float a = 0.f, q = 0.f;
  for (int i = lane; i < 8 * S; i += 64) {
    const int c = g * 8 + i / S;
    const float gm = gamma ? gamma[c] : 1.f;
    const float* t = partial + (((size_t)b * C + c) * S + i % S) * 2;
    a += gm * t[0];
    q += gm * t[1];
  }
  a = wave_sum(a);
  q = wave_sum(q);
  if (lane == 0) {
    sums[bg * 2] = a;
    sums[bg * 2 + 1] = q;
  }
}
// dgamma[c] (+)= inv_scale * sum over (b, slice) of partial[..][1], dbeta likewise from [..][0]; one wave per channel
__global__ __launch_bounds__(256) void gnh_bwd_affine_kernel(const float* __restrict__ partial, int B, int C, int S,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                             float inv_scale, const float* __restrict__ hs) {
  if (hs) inv_scale *= hs[1];
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < B * S; i += 64) {
    const float* t = partial + (((size_t)(i / S) * C + c) * S + i % S) * 2;
    s1 += t[0];
    s2 += t[1];
  }
  s1 = wave_sum(s1) * inv_scale;
  s2 = wave_sum(s2) * inv_scale;
  if (lane == 0) {
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + s2;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + s1;
  }
}
// dz = invstd_bg * (gamma_c g - A / n - xhat B / n), n = 8 HW
__global__ __launch_bounds__(256) void gnh_bwd_apply_kernel(const half8* __restrict__ da, const half8* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int relu, const float* __restrict__ sums, half8* __restrict__ dz, int C,
                                                            int HW) {
  const int CBK = C >> 5, G = C >> 3;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK, cg = threadIdx.x & 3;
  const int c0 = cblk * 32 + cg * 8, bg = b * G + cblk * 4 + cg;
  const float mu = mean[bg], is = invstd[bg];
  const float inv_n = 1.f / (8.f * (float)HW);
  const float a1 = sums[bg * 2] * inv_n, a2 = sums[bg * 2 + 1] * inv_n * is;
  float sc[8], sh[8], gm[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gm[e] = gamma ? gamma[c0 + e] : 1.f;
    sc[e] = gm[e] * is;
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu * sc[e];
  }
  const size_t base = (size_t)pl * HW * 4;
  const int nv = HW * 4;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    const half8 g8 = da[base + v], z8 = z[base + v];
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float zf = (float)z8[e];
      float gf = (float)g8[e];
      if (relu) gf = fmaf(zf, sc[e], sh[e]) > 0.f ? gf : 0.f;
      o[e] = h_sat(is * (gm[e] * gf - a1 - (zf - mu) * a2));
    }
    dz[base + v] = o;
  }
}

// ---- BatchNorm + ReLU + 2x2 max-pool as ONE pass each way (the last layer of a VGG16 stack: its full-resolution activation has no
// other reader than the pool, so it is never written: the forward reads z and writes the pooled map, the backward recomputes
// relu(z * sc + sh) per window to find the argmax -- first maximum in scan order, on the fp16-ROUNDED activations, i.e. exactly
// what bnh_apply + poolh_fwd/bwd do in two passes).  One thread per (output pixel, 8 channels); plane = (sample, channel block).
struct PoolWin {
  half8 z[4];
};
__device__ __forceinline__ PoolWin poolwin_load(const half8* __restrict__ z, long long pl, int oy, int ox, int cg, int H, int W) {
  const half8* p = z + ((pl * H + 2 * oy) * W + 2 * ox) * 4 + cg;
  PoolWin w;
  w.z[0] = p[0];
  w.z[1] = p[4];
  w.z[2] = p[(size_t)W * 4];
  w.z[3] = p[(size_t)W * 4 + 4];
  return w;
}
__global__ __launch_bounds__(256) void bnh_apply_pool_kernel(const half8* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, half8* __restrict__ y, int C, int H,
                                                             int W) {
  const int CBK = C >> 5, Ho = H >> 1, Wo = W >> 1;
  const int pl = blockIdx.y, cblk = pl % CBK, cg = threadIdx.x & 3;
  const int c0 = cblk * 32 + cg * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = (gamma ? gamma[c0 + e] : 1.f) * invstd[c0 + e];
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mean[c0 + e] * sc[e];
  }
  const int nv = Ho * Wo * 4;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
    const int op = v >> 2, oy = op / Wo, ox = op - oy * Wo;
    const PoolWin w = poolwin_load(z, pl, oy, ox, cg, H, W);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 m = (_Float16)fmaxf(fmaf((float)w.z[0][e], sc[e], sh[e]), 0.f);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const _Float16 a = (_Float16)fmaxf(fmaf((float)w.z[k][e], sc[e], sh[e]), 0.f);
        m = a > m ? a : m;
      }
      o[e] = m;
    }
    y[(size_t)pl * nv + v] = o;
  }
}
// MODE 0: partial[c][blk] = (sum g, sum g (z - mean) invstd) with g = the pooled gradient at the window's argmax where the
// activation is positive; MODE 1: dz for the four positions of every window.
template <int MODE>
__global__ __launch_bounds__(256) void bnh_pool_bwd_kernel(const half8* __restrict__ dy, const half8* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ partial, const float* __restrict__ sums,
                                                           float inv_count, half8* __restrict__ dz, int C, int H, int W, int S,
                                                           int NB, float scale, const float* __restrict__ hs) {
  __shared__ float red[MODE == 0 ? 256 * 17 : 1];
  const int CBK = C >> 5, Ho = H >> 1, Wo = W >> 1;
  const int pl = blockIdx.y, b = pl / CBK, cblk = pl - b * CBK, cg = threadIdx.x & 3;
  const int c0 = cblk * 32 + cg * 8;
  float sc[8], sh[8], mu[8], kk[8], a1[8], a2[8];
  if (MODE == 1) {
    if (hs) scale *= hs[0];
    inv_count *= scale;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float is = invstd[c0 + e], gm = gamma ? gamma[c0 + e] : 1.f;
    mu[e] = mean[c0 + e];
    sc[e] = gm * is;
    sh[e] = (beta ? beta[c0 + e] : 0.f) - mu[e] * sc[e];
    kk[e] = gm * is;
    a1[e] = MODE == 1 ? sums[(c0 + e) * 2] * inv_count : 0.f;
    a2[e] = MODE == 1 ? sums[(c0 + e) * 2 + 1] * inv_count * is : 0.f;
  }
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const int nv = Ho * Wo * 4;
  int v0 = blockIdx.x * 256, v1 = nv, step = gridDim.x * 256;
  if (MODE == 0) {
    const int per = (nv + S - 1) / S;
    v0 = blockIdx.x * per;
    v1 = min(v0 + per, nv);
    step = 256;
  }
  for (int v = v0 + threadIdx.x; v < v1; v += step) {
    const int op = v >> 2, oy = op / Wo, ox = op - oy * Wo;
    const PoolWin w = poolwin_load(z, pl, oy, ox, cg, H, W);
    const half8 g8 = dy[(size_t)pl * nv + v];
    half8 o[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int arg = 0;
      _Float16 m = (_Float16)fmaxf(fmaf((float)w.z[0][e], sc[e], sh[e]), 0.f);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const _Float16 a = (_Float16)fmaxf(fmaf((float)w.z[k][e], sc[e], sh[e]), 0.f);
        if (a > m) {
          m = a;
          arg = k;
        }
      }
      const float zarg = (float)(arg == 0 ? w.z[0][e] : (arg == 1 ? w.z[1][e] : (arg == 2 ? w.z[2][e] : w.z[3][e])));
      // ReLU mask from the fp32 pre-activation, as bnh_bwd_* recompute it
      const float gf = fmaf(zarg, sc[e], sh[e]) > 0.f ? (float)g8[e] : 0.f;
      if (MODE == 0) {
        s1[e] += gf;
        s2[e] += gf * (zarg - mu[e]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zf = (float)w.z[k][e];
          o[k][e] = h_sat(kk[e] * ((k == arg ? gf : 0.f) - a1[e] - (zf - mu[e]) * a2[e]));
        }
      }
    }
    if (MODE == 1) {
      half8* q = dz + ((pl * (long long)H + 2 * oy) * W + 2 * ox) * 4 + cg;
      q[0] = o[0];
      q[4] = o[1];
      q[(size_t)W * 4] = o[2];
      q[(size_t)W * 4 + 4] = o[3];
    }
  }
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[threadIdx.x * 17 + e] = s1[e];
      red[threadIdx.x * 17 + 8 + e] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int ch = threadIdx.x & 31, which = threadIdx.x >> 5;
      const int cgi = ch >> 3, e = ch & 7;
      float t = 0.f;
      for (int k = 0; k < 64; ++k) t += red[(k * 4 + cgi) * 17 + which * 8 + e];
      const int c = cblk * 32 + ch;
      if (which == 1) t *= invstd[c];
      partial[((size_t)c * NB + (size_t)b * S + blockIdx.x) * 2 + which] = t;
    }
  }
}

// 2x2 / stride 2 max-pool.  One thread per (output pixel, 8 channels).
__global__ __launch_bounds__(256) void poolh_fwd_kernel(const half8* __restrict__ x, half8* __restrict__ y, int H, int W,
                                                        long long total) {
  const int Ho = H >> 1, Wo = W >> 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cg = (int)(i & 3);
    long long r = i >> 2;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    const half8* xp = x + ((pl * H + 2 * oy) * W + 2 * ox) * 4 + cg;
    const half8 a = xp[0], b = xp[4], c = xp[(size_t)W * 4], d = xp[(size_t)W * 4 + 4];
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 m0 = a[e] > b[e] ? a[e] : b[e], m1 = c[e] > d[e] ? c[e] : d[e];
      o[e] = m0 > m1 ? m0 : m1;
    }
    y[i] = o;
  }
}
// dx: the FIRST maximum of the window in scan order takes the gradient (torch's choice)
__global__ __launch_bounds__(256) void poolh_bwd_kernel(const half8* __restrict__ x, const half8* __restrict__ dy,
                                                        half8* __restrict__ dx, int H, int W, long long total) {
  const int Ho = H >> 1, Wo = W >> 1;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int cg = (int)(i & 3);
    long long r = i >> 2;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const long long pl = r / Ho;
    const size_t o00 = ((pl * H + 2 * oy) * W + 2 * ox) * 4 + cg;
    const half8 a = x[o00], b = x[o00 + 4], c = x[o00 + (size_t)W * 4], d = x[o00 + (size_t)W * 4 + 4];
    const half8 g = dy[i];
    half8 ga, gb, gc, gd;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int arg = 0;
      _Float16 m = a[e];
      if (b[e] > m) { m = b[e]; arg = 1; }
      if (c[e] > m) { m = c[e]; arg = 2; }
      if (d[e] > m) { m = d[e]; arg = 3; }
      const _Float16 zero = (_Float16)0.f;
      ga[e] = arg == 0 ? g[e] : zero;
      gb[e] = arg == 1 ? g[e] : zero;
      gc[e] = arg == 2 ? g[e] : zero;
      gd[e] = arg == 3 ? g[e] : zero;
    }
    dx[o00] = ga;
    dx[o00 + 4] = gb;
    dx[o00 + (size_t)W * 4] = gc;
    dx[o00 + (size_t)W * 4 + 4] = gd;
  }
}

// Next step's loss scale from the largest gradient magnitude the casts saw since the last update: the power of two that puts
// that magnitude at `target` (4096: 16x below fp16's maximum; stores saturate anyway), clamped to [lo, hi]; no magnitude seen
// (no backward since): unchanged.
__global__ void h_scale_update_kernel(float* hs, float target, float lo, float hi) {
  const float m = __uint_as_float(((unsigned*)hs)[2]);
  if (m > 0.f) {
    float sc = exp2f(floorf(log2f(target / m)));
    sc = fminf(fmaxf(sc, lo), hi);
    hs[0] = sc;
    hs[1] = 1.f / sc;
    ((unsigned*)hs)[2] = 0u;
  }
}

// =========================================================================================
// The stem: 3x3 / s1 / p1 conv from an fp32 NCHW image with 1..4 channels (VGG16's first layer, fpnseg.py:28-31) straight into
// the blocked fp16 domain.  K = 9 * CIN is far too short for the matrix pipe and the layer is bound by writing its output
// (64 channels x 256 x 256 x 48 frames = 403 MB as fp16): plain fp32 FMAs, a lane = a pixel with its 3x3xCIN neighbourhood
// in registers, weights through the scalar cache (their index is wave-uniform), 16-byte fp16 stores, and the BatchNorm
// moments of the fp32 results per 64 pixels = per wave (DPP sums) in ge_bn_finalize's format.
// =========================================================================================
template <int CIN>
__global__ __launch_bounds__(256) void h_stem3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, half8* __restrict__ z,
                                                            float* __restrict__ stats, int M, int H, int W, int nwave_tiles,
                                                            int parts) {
  const int lane = threadIdx.x & 63;
  const int HW = H * W, per_img = HW >> 6, MB = M >> 5;
  for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < nwave_tiles; tile += gridDim.x * 4) {
    const int b = tile / per_img, p0 = (tile - b * per_img) << 6;      // 64 consecutive pixels of one row (W % 64 == 0)
    const int pix = p0 + lane, y = pix / W, xq = pix - y * W;
    float xin[CIN * 9];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = y + t / 3 - 1, ix = xq + t % 3 - 1;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        xin[ci * 9 + t] = ok ? x[((size_t)b * CIN + ci) * HW + (size_t)iy * W + ix] : 0.f;
      }
    for (int g = 0; g < MB * 4; ++g) {      // 8 output channels per pass
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int co = g * 8 + e;
        float a = bias ? bias[co] : 0.f;
#pragma unroll
        for (int k = 0; k < CIN * 9; ++k) a = fmaf(w[co * CIN * 9 + k], xin[k], a);
        acc[e] = a;
      }
      half8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = h_sat(acc[e]);
      z[(((size_t)b * MB + (g >> 2)) * HW + pix) * 4 + (g & 3)] = v;
      if (stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sv = wave_sum(acc[e]), qv = wave_sum(acc[e] * acc[e]);
          if (lane == 0) {
            const float mean = sv * (1.f / 64.f);
            float* o3 = stats + ((size_t)(g * 8 + e) * parts + tile) * 3;
            o3[0] = 64.f;
            o3[1] = mean;
            o3[2] = fmaxf(qv - sv * mean, 0.f);
          }
        }
      }
    }
  }
}
// Its weight gradient: slab[wg][t][co][ci] = sum over the workgroup's pixels of dz[co][pix] * x[ci][pix + t]  (h_slab_reduce's
// layout).  A lane = a pixel; per pass of 8 output channels 8 * 9 * CIN accumulators, summed over the wave by DPP at the end.
template <int CIN>
__global__ __launch_bounds__(256) void h_stem3x3_wgrad_kernel(const float* __restrict__ x, const half8* __restrict__ dz,
                                                              float* __restrict__ slab, int M, int H, int W, int nwave_tiles) {
  __shared__ float red[4][8 * 9 * CIN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int HW = H * W, per_img = HW >> 6, MB = M >> 5;
  float* out = slab + (size_t)blockIdx.x * 9 * M * CIN;
  for (int g = 0; g < MB * 4; ++g) {
    float acc[8][CIN * 9];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < CIN * 9; ++k) acc[e][k] = 0.f;
    for (int tile = blockIdx.x * 4 + wave; tile < nwave_tiles; tile += gridDim.x * 4) {
      const int b = tile / per_img, p0 = (tile - b * per_img) << 6;
      const int pix = p0 + lane, y = pix / W, xq = pix - y * W;
      const half8 d = dz[(((size_t)b * MB + (g >> 2)) * HW + pix) * 4 + (g & 3)];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y + t / 3 - 1, ix = xq + t % 3 - 1;
          const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
          const float xv = ok ? x[((size_t)b * CIN + ci) * HW + (size_t)iy * W + ix] : 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e][ci * 9 + t] = fmaf((float)d[e], xv, acc[e][ci * 9 + t]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < CIN * 9; ++k) {
        const float sv = wave_sum(acc[e][k]);
        if (lane == 0) red[wave][e * CIN * 9 + k] = sv;
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * 9 * CIN; i += 256) {
      const int e = i / (CIN * 9), k = i - e * CIN * 9, ci = k / 9, t = k - ci * 9;
      out[((size_t)t * M + g * 8 + e) * CIN + ci] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
    __syncthreads();
  }
}

// What ds_read_b64_tr_b16 returns: LDS holds halves 0..255 (value = index), lane l reads at byte 8 l.  out[l][0..3].
__global__ void h_probe_tr_kernel(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 buf[256];
  for (int i = threadIdx.x; i < 256; i += 64) buf[i] = (_Float16)(float)i;
  __syncthreads();
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS fp16x4_t*)((char*)buf + threadIdx.x * 8));
  const half4 h = __builtin_bit_cast(half4, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)h[j];
}

// =========================================================================================
// host side
// =========================================================================================
static bool h_tile_shape(int NT, int H, int W, int& TR, int& TC, int& tcs) {
  TC = W >= 64 ? 64 : W;
  if (TC != 16 && TC != 32 && TC != 64) return false;
  tcs = TC == 16 ? 4 : (TC == 32 ? 5 : 6);
  TR = NT / TC;
  return W % TC == 0 && H % TR == 0;
}
static bool h_conv_ok(int B, int C, int M, int H, int W) {
  if (B < 1 || C % 32 || M % 64 || C < 32) return false;
  int TR, TC, tcs;
  if (!h_tile_shape(M % 128 == 0 ? 128 : 256, H, W, TR, TC, tcs)) return false;
  const unsigned long long xb = 2ull * B * C * H * W, yb = 2ull * B * M * H * W;
  return xb < 0xFFFF0000ull && yb < 0xFFFF0000ull;
}

template <int PW, int CW, int WPX, bool F32OUT>
static void h_conv_launch_t(const HConvParams& p, int grid, hipStream_t st) {
  constexpr size_t smem = 2 * (PW * WPX * 32 == 128 ? 5 : 7) * 4096 + 3 * CW * 64 * 64;
    GE_MAX_LDS((int)smem, (const void*)h_conv3x3_kernel<PW, CW, WPX, F32OUT>);
  h_conv3x3_kernel<PW, CW, WPX, F32OUT><<<grid, 256, smem, st>>>(p);
  ge_note_kernel("h_conv3x3_kernel<%d, %d, %d, %s>", PW, CW, WPX, F32OUT ? "true" : "false");
}

static int h_conv_launch(const void* x, const void* wp, const float* bias, void* y, float* stats, int B, int C, int M, int H,
                         int W, int flip, hipStream_t st, bool f32out = false, const float* addend = nullptr,
                         float out_scale = 1.f, const float* hs = nullptr) {
  HConvParams p;
  p.addend = addend;
  p.out_scale = out_scale;
  p.hs = hs;
  p.dbg = 0;      // (phase-ablation bits of the tuning builds)
  p.x = x;
  p.wp = wp;
  p.bias = bias;
  p.y = y;
  p.stats = stats;
  p.B = B;
  p.C = C;
  p.M = M;
  p.H = H;
  p.W = W;
  p.flip = flip;
  p.x_bytes = (uint32_t)(2ull * B * C * H * W);
  p.wp_bytes = (uint32_t)(2ull * 9 * M * C);
  const bool big = M % 128 == 0;
  // 256-pixel x 128-channel tiles (GE_H_WIDE, default on) where the image splits into 256-pixel rectangles and the grid still
  // holds two workgroups per CU; otherwise 128 x 128 (M % 128 == 0) or 256 x 64
  static const int wide_env = []() {
    const char* e = getenv("GE_H_WIDE");
    return e ? atoi(e) : 1;
  }();
  int TRw, TCw, tcsw;
  const bool wide = big && wide_env && h_tile_shape(256, H, W, TRw, TCw, tcsw) &&
                    (wide_env == 2 || (long long)B * (H / TRw) * (W / TCw) * (M / 128) >= 512);      // 2: always (tests)
  const int NT = (big && !wide) ? 128 : 256, MT = big ? 128 : 64;
  h_tile_shape(NT, H, W, p.TR, p.TC, p.tcs);
  p.tiles_x = W / p.TC;
  p.tiles_y = H / p.TR;
  p.tiles_m = M / MT;
  const int ntp = B * p.tiles_x * p.tiles_y;
  p.stats_parts = ntp * (NT / 64);
  const int grid = ntp * p.tiles_m;
  if (wide) {
    if (f32out) h_conv_launch_t<2, 2, 4, true>(p, grid, st);
    else h_conv_launch_t<2, 2, 4, false>(p, grid, st);
  } else if (big) {
    if (f32out) h_conv_launch_t<2, 2, 2, true>(p, grid, st);
    else h_conv_launch_t<2, 2, 2, false>(p, grid, st);
  } else {
    if (f32out) h_conv_launch_t<4, 1, 2, true>(p, grid, st);
    else h_conv_launch_t<4, 1, 2, false>(p, grid, st);
  }
  GE_CHECK_LAUNCH("h_conv3x3");
  return GE_OK;
}

#define H_SLAB_GROUPS 32      // reductions over more slabs than this run in two stages (group sums, then the groups)
struct HWgradPlan {
  int TR, TC, tcs, tiles_x, tiles_y, ntiles, tiles_m, cblocks, splits, tiles_per_split, nslabs, CB;
};
static bool h_wgrad_plan(int B, int C, int M, int H, int W, HWgradPlan& q) {
  if (B < 1 || C % 32 || M % 64 || C < 32) return false;
  if (!h_tile_shape(128, H, W, q.TR, q.TC, q.tcs)) return false;
  q.CB = M % 128 == 0 ? 4 : 2;
  q.tiles_x = W / q.TC;
  q.tiles_y = H / q.TR;
  q.ntiles = B * q.tiles_x * q.tiles_y;
  q.tiles_m = M / (q.CB * 32);
  q.cblocks = C / 32;
  const int base = q.tiles_m * q.cblocks;
  int splits = (512 + base - 1) / base;             // about two workgroups per CU's worth of work items
  if (splits > q.ntiles) splits = q.ntiles;
  if (splits < 1) splits = 1;
  q.tiles_per_split = (q.ntiles + splits - 1) / splits;
  q.splits = (q.ntiles + q.tiles_per_split - 1) / q.tiles_per_split;
  q.nslabs = q.splits * (q.CB == 4 ? 1 : 2);
  return true;
}

template <int CB, int TCS, bool SB>
static void h_wgrad_launch_sb(const HWgradParams& p, int grid, hipStream_t st) {
  const size_t smem = (SB ? 1 : 2) * (CB * 8192 + 5 * 4096);
    GE_MAX_LDS((int)smem, (const void*)h_wgrad3x3_kernel<CB, TCS, SB>);
  h_wgrad3x3_kernel<CB, TCS, SB><<<grid, 256, smem, st>>>(p);
}
template <int CB, int TCS>
static void h_wgrad_launch_t(const HWgradParams& p, int grid, hipStream_t st) {
  // measured (tools/bench_half.py, 48 frames): maps of 32 and 16 columns gain 5 - 8 % (512 -> 512 @ 32 x 32: 1037 -> 1104 TFLOP/s),
  // maps of 64 columns and more lose 4 - 10 % (their instantiation needs 268 registers: squeezed into 256 it spills)
  static const int sb_env = []() {
    const char* e = getenv("GE_H_WGRAD_SB");
    return e ? atoi(e) : -1;
  }();
  if (sb_env < 0 ? TCS < 6 : sb_env != 0) h_wgrad_launch_sb<CB, TCS, true>(p, grid, st);
  else h_wgrad_launch_sb<CB, TCS, false>(p, grid, st);
}

extern "C" {

// 1 when the blocked-fp16 3x3 / stride 1 / pad 1 kernels cover a layer: Cin % 32 == 0, Cout % 64 == 0, W in {16, 32} or a
// multiple of 64, H a multiple of the tile rows (128 or 256 pixels per tile), tensors below 4 GB
int ge_h_conv3x3_supported(int B, int Cin, int Cout, int H, int W) {
  HWgradPlan q;
  return h_conv_ok(B, Cin, Cout, H, W) && h_conv_ok(B, Cout, Cin, H, W) && h_wgrad_plan(B, Cin, Cout, H, W, q) ? 1 : 0;
}
// number of (count, mean, M2) triples per output channel ge_h_conv3x3_fwd writes: one per 64 pixels
int ge_h_conv3x3_stat_parts(int B, int H, int W) { return B * H * W / 64; }

// z = conv3x3(x) (+ bias): x, z blocked fp16; wp = ge_conv2d_f16_pack_weight(w, .., transposed = 0): [9][Cout][Cin] fp16.
// stats (nullable): [Cout][ge_h_conv3x3_stat_parts][3] BatchNorm moments of z (of the fp32 results, before rounding).
int ge_h_conv3x3_fwd(const void* x, const void* wp, const float* bias, void* z, float* stats, int B, int Cin, int Cout, int H,
                     int W, void* stream) {
  GE_REQUIRE(x && wp && z, "h_conv3x3_fwd: null pointer");
  GE_REQUIRE(h_conv_ok(B, Cin, Cout, H, W), "h_conv3x3_fwd: unsupported geometry B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
  return h_conv_launch(x, wp, bias, z, stats, B, Cin, Cout, H, W, 0, (hipStream_t)stream);
}
// dx = data gradient: dz, dx blocked fp16; wp = ge_conv2d_f16_pack_weight(w, .., transposed = 1): [9][Cin][Cout] fp16
int ge_h_conv3x3_dgrad(const void* dz, const void* wp, void* dx, int B, int Cin, int Cout, int H, int W, void* stream) {
  GE_REQUIRE(dz && wp && dx, "h_conv3x3_dgrad: null pointer");
  GE_REQUIRE(h_conv_ok(B, Cout, Cin, H, W), "h_conv3x3_dgrad: unsupported geometry B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
  return h_conv_launch(dz, wp, nullptr, dx, nullptr, B, Cout, Cin, H, W, 1, (hipStream_t)stream);
}
// The same two passes LEAVING the fp16 domain: fp32 NCHW results.  y = conv3x3(x) (+ bias), stats as above;
// dx = out_scale * data gradient (+ addend, fp32 NCHW: the gradient arriving through a skip connection)
int ge_h_conv3x3_fwd_f32(const void* x, const void* wp, const float* bias, float* y, float* stats, int B, int Cin, int Cout,
                         int H, int W, void* stream) {
  GE_REQUIRE(x && wp && y, "h_conv3x3_fwd_f32: null pointer");
  GE_REQUIRE(h_conv_ok(B, Cin, Cout, H, W) && 4ull * B * Cout * H * W < (1ull << 40), "h_conv3x3_fwd_f32: unsupported geometry");
  return h_conv_launch(x, wp, bias, y, stats, B, Cin, Cout, H, W, 0, (hipStream_t)stream, true, nullptr, 1.f);
}
int ge_h_conv3x3_dgrad_f32(const void* dz, const void* wp, const float* addend, float* dx, float out_scale,
                           const float* dev_scale, int B, int Cin, int Cout, int H, int W, void* stream) {
  GE_REQUIRE(dz && wp && dx, "h_conv3x3_dgrad_f32: null pointer");
  GE_REQUIRE(h_conv_ok(B, Cout, Cin, H, W), "h_conv3x3_dgrad_f32: unsupported geometry");
  return h_conv_launch(dz, wp, nullptr, dx, nullptr, B, Cout, Cin, H, W, 1, (hipStream_t)stream, true, addend, out_scale, dev_scale);
}
// floats of workspace for ge_h_conv3x3_wgrad
long long ge_h_conv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W) {
  HWgradPlan q;
  if (!h_wgrad_plan(B, Cin, Cout, H, W, q)) return 0;
  return (long long)(q.nslabs + (q.nslabs > H_SLAB_GROUPS ? H_SLAB_GROUPS : 0)) * 9 * Cout * Cin;
}
// dw[Cout][Cin][3][3] (+)= scale * weight gradient; x, dz blocked fp16, dw fp32
int ge_h_conv3x3_wgrad(const void* x, const void* dz, float* dw, float* workspace, int B, int Cin, int Cout, int H, int W,
                       float scale, const float* dev_scale, int accumulate, void* stream) {
  GE_REQUIRE(x && dz && dw && workspace, "h_conv3x3_wgrad: null pointer");
  HWgradPlan q;
  GE_REQUIRE(h_wgrad_plan(B, Cin, Cout, H, W, q) && 2ull * B * Cin * H * W < 0xFFFF0000ull &&
                 2ull * B * Cout * H * W < 0xFFFF0000ull,
             "h_conv3x3_wgrad: unsupported geometry B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
  hipStream_t st = (hipStream_t)stream;
  HWgradParams p;
  p.x = x;
  p.dz = dz;
  p.slab = workspace;
  p.B = B;
  p.C = Cin;
  p.M = Cout;
  p.H = H;
  p.W = W;
  p.TR = q.TR;
  p.tiles_x = q.tiles_x;
  p.tiles_y = q.tiles_y;
  p.ntiles = q.ntiles;
  p.tiles_per_split = q.tiles_per_split;
  p.tiles_m = q.tiles_m;
  p.cblocks = q.cblocks;
  p.x_bytes = (uint32_t)(2ull * B * Cin * H * W);
  p.dz_bytes = (uint32_t)(2ull * B * Cout * H * W);
  const int grid = q.tiles_m * q.cblocks * q.splits;
  if (q.CB == 4) {
    if (q.tcs == 4) h_wgrad_launch_t<4, 4>(p, grid, st);
    else if (q.tcs == 5) h_wgrad_launch_t<4, 5>(p, grid, st);
    else h_wgrad_launch_t<4, 6>(p, grid, st);
  } else {
    if (q.tcs == 4) h_wgrad_launch_t<2, 4>(p, grid, st);
    else if (q.tcs == 5) h_wgrad_launch_t<2, 5>(p, grid, st);
    else h_wgrad_launch_t<2, 6>(p, grid, st);
  }
  ge_note_kernel("h_wgrad3x3_kernel<%d, %d>", q.CB, q.tcs);
  GE_CHECK_LAUNCH("h_wgrad3x3");
  ge_record_split_event(st);
  const int MC = Cout * Cin;
  if (q.nslabs > H_SLAB_GROUPS) {
    float* part = workspace + (size_t)q.nslabs * 9 * MC;
    const int per = ge_cdiv(q.nslabs, H_SLAB_GROUPS), groups = ge_cdiv(q.nslabs, per);
    h_slab_group_kernel<<<dim3(ge_cdiv(MC, 256), groups), 256, 0, st>>>(workspace, part, MC, q.nslabs, per);
    GE_CHECK_LAUNCH("h_slab_group");
    h_slab_reduce_kernel<<<ge_cdiv(MC, 256), 256, 0, st>>>(part, dw, Cout, Cin, groups, scale, accumulate, dev_scale);
  } else {
    h_slab_reduce_kernel<<<ge_cdiv(MC, 256), 256, 0, st>>>(workspace, dw, Cout, Cin, q.nslabs, scale, accumulate, dev_scale);
  }
  GE_CHECK_LAUNCH("h_slab_reduce");
  return GE_OK;
}

// fp32 NCHW <-> blocked fp16 (C % 32 == 0), values multiplied by `scale`
int ge_h_from_f32(const float* x, void* h, int B, int C, int HW, float scale, float* dev_scale, void* stream) {
  GE_REQUIRE(x && h && C % 32 == 0 && B > 0 && HW > 0, "h_from_f32: bad arguments");
  dim3 grid(min(ge_cdiv(HW, 256), 256), B * (C / 32));
  h_from_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (_Float16*)h, C, HW, scale, dev_scale, nullptr);
  GE_CHECK_LAUNCH("h_from_f32");
  return GE_OK;
}
// h = saturate(x * scale + addend): the cast with a blocked fp16 addend folded in, summed in fp32 (gradient of a blocked tensor that
// is read both inside the fp16 domain and, through ge_h_to_f32, outside it)
int ge_h_from_f32_add(const float* x, const void* addend, void* h, int B, int C, int HW, float scale, float* dev_scale,
                      void* stream) {
  GE_REQUIRE(x && h && addend && C % 32 == 0 && B > 0 && HW > 0, "h_from_f32_add: bad arguments");
  dim3 grid(min(ge_cdiv(HW, 256), 256), B * (C / 32));
  h_from_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, (_Float16*)h, C, HW, scale, dev_scale, (const _Float16*)addend);
  GE_CHECK_LAUNCH("h_from_f32_add");
  return GE_OK;
}
int ge_h_to_f32(const void* h, float* x, int B, int C, int HW, float scale, const float* dev_scale, void* stream) {
  GE_REQUIRE(x && h && C % 32 == 0 && B > 0 && HW > 0, "h_to_f32: bad arguments");
  dim3 grid(min(ge_cdiv(HW, 256), 256), B * (C / 32));
  h_to_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const _Float16*)h, x, C, HW, scale, dev_scale);
  GE_CHECK_LAUNCH("h_to_f32");
  return GE_OK;
}

// a = (z - mean) * invstd * gamma + beta (+ ReLU), blocked fp16 in and out, fp32 per-channel vectors
int ge_h_bn_apply(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* a, int B,
                  int C, int HW, int relu, void* stream) {
  GE_REQUIRE(z && a && mean && invstd && C % 32 == 0 && B > 0, "h_bn_apply: bad arguments");
  dim3 grid(min(ge_cdiv((long long)HW * 4, 256 * 4), 64), B * (C / 32));
  bnh_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8*)z, mean, invstd, gamma, beta, (half8*)a, C, HW, relu);
  GE_CHECK_LAUNCH("h_bn_apply");
  return GE_OK;
}
// slices per (sample, channel block) plane of the backward reductions; partial buffers hold C * B * slices * 2 floats
int ge_h_bn_slices(int HW) {
  const int s = HW * 4 / 8192;
  return s < 1 ? 1 : s;
}
// sums[C][2] = (sum g, sum g xhat) with g = da masked by the recomputed ReLU, times inv_scale (true units);
// dgamma / dbeta (nullable) (+)= the same
int ge_h_bn_bwd_reduce(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma,
                       const float* beta, int relu, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate,
                       float inv_scale, const float* dev_scale, int B, int C, int HW, void* stream) {
  GE_REQUIRE(da && z && mean && invstd && partial && sums && C % 32 == 0 && B > 0, "h_bn_bwd_reduce: bad arguments");
  const int S = ge_h_bn_slices(HW);
  dim3 grid(S, B * (C / 32));
  hipStream_t st = (hipStream_t)stream;
  bnh_bwd_partial_kernel<0><<<grid, 256, 0, st>>>((const half8*)da, (const half8*)z, mean, invstd, gamma, beta, relu, partial, C,
                                                  HW, S, B * S);
  GE_CHECK_LAUNCH("h_bn_bwd_partial");
  bnh_bwd_finalize_kernel<<<ge_cdiv(C, 4), 256, 0, st>>>(partial, B * S, C, sums, dgamma, dbeta, accumulate, inv_scale, dev_scale);
  GE_CHECK_LAUNCH("h_bn_bwd_finalize");
  return GE_OK;
}
// dz = gamma * invstd * (g - scale * (sums[0] * inv_count + xhat * sums[1] * inv_count)); scale = the gradients' loss scale
int ge_h_bn_bwd_apply(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, int relu, const float* sums, float inv_count, float scale, const float* dev_scale,
                      void* dz, int B, int C, int HW, void* stream) {
  GE_REQUIRE(da && z && dz && mean && invstd && sums && C % 32 == 0 && B > 0, "h_bn_bwd_apply: bad arguments");
  dim3 grid(min(ge_cdiv((long long)HW * 4, 256 * 4), 64), B * (C / 32));
  bnh_bwd_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8*)da, (const half8*)z, mean, invstd, gamma, beta, relu,
                                                              sums, inv_count, (half8*)dz, C, HW, scale, dev_scale);
  GE_CHECK_LAUNCH("h_bn_bwd_apply");
  return GE_OK;
}
// out[C] (+)= inv_scale * sum over (b, y, x) of dz  (bias gradient of the conv that produced z)
int ge_h_channel_sum(const void* dz, float* partial, float* out, int accumulate, float inv_scale, const float* dev_scale, int B,
                     int C, int HW, void* stream) {
  GE_REQUIRE(dz && partial && out && C % 32 == 0 && B > 0, "h_channel_sum: bad arguments");
  const int S = ge_h_bn_slices(HW);
  dim3 grid(S, B * (C / 32));
  hipStream_t st = (hipStream_t)stream;
  bnh_bwd_partial_kernel<1><<<grid, 256, 0, st>>>((const half8*)dz, nullptr, nullptr, nullptr, nullptr, nullptr, 0, partial, C, HW,
                                                  S, B * S);
  GE_CHECK_LAUNCH("h_channel_sum");
  bnh_bwd_finalize_kernel<<<ge_cdiv(C, 4), 256, 0, st>>>(partial, B * S, C, nullptr, nullptr, out, accumulate, inv_scale, dev_scale);
  GE_CHECK_LAUNCH("h_channel_sum_finalize");
  return GE_OK;
}

// nn.GroupNorm(C / 8, C) (+ ReLU) on blocked fp16 tensors (the discriminator towers, fpnseg.py:465): statistics [B][C / 8] from the
// conv epilogue's moments (stats [C][B * HW / 64][3]); backward: partial holds B * C * ge_h_bn_slices(HW) * 2 floats, sums B * C / 8 * 2
int ge_h_gn8_stats(const float* stats, float* mean, float* invstd, int B, int C, int HW, float eps, void* stream) {
  GE_REQUIRE(stats && mean && invstd && C % 32 == 0 && B > 0 && HW % 64 == 0, "h_gn8_stats: bad arguments");
  const int BG = B * (C / 8);
  gnh_finalize_kernel<<<ge_cdiv(BG, 4), 256, 0, (hipStream_t)stream>>>(stats, B * HW / 64, HW / 64, C, eps, mean, invstd, BG);
  GE_CHECK_LAUNCH("h_gn8_stats");
  return GE_OK;
}
int ge_h_gn8_apply(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* a, int B,
                   int C, int HW, int relu, void* stream) {
  GE_REQUIRE(z && a && mean && invstd && C % 32 == 0 && B > 0, "h_gn8_apply: bad arguments");
  dim3 grid(min(ge_cdiv((long long)HW * 4, 256 * 4), 64), B * (C / 32));
  gnh_apply_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8*)z, mean, invstd, gamma, beta, (half8*)a, C, HW, relu);
  GE_CHECK_LAUNCH("h_gn8_apply");
  return GE_OK;
}
int ge_h_gn8_bwd(const void* da, const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                 int relu, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate, float inv_scale,
                 const float* dev_scale, void* dz, int B, int C, int HW, void* stream) {
  GE_REQUIRE(da && z && dz && mean && invstd && partial && sums && C % 32 == 0 && B > 0, "h_gn8_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int S = ge_h_bn_slices(HW), BG = B * (C / 8);
  gnh_bwd_partial_kernel<<<dim3(S, B * (C / 32)), 256, 0, st>>>((const half8*)da, (const half8*)z, mean, invstd, gamma, beta, relu,
                                                              partial, C, HW, S);
  GE_CHECK_LAUNCH("h_gn8_bwd_partial");
  gnh_bwd_group_kernel<<<ge_cdiv(BG, 4), 256, 0, st>>>(partial, gamma, C, S, sums, BG);
  if (dgamma || dbeta)
    gnh_bwd_affine_kernel<<<ge_cdiv(C, 4), 256, 0, st>>>(partial, B, C, S, dgamma, dbeta, accumulate, inv_scale, dev_scale);
  dim3 grid(min(ge_cdiv((long long)HW * 4, 256 * 4), 64), B * (C / 32));
  gnh_bwd_apply_kernel<<<grid, 256, 0, st>>>((const half8*)da, (const half8*)z, mean, invstd, gamma, beta, relu, sums, (half8*)dz, C,
                                             HW);
  GE_CHECK_LAUNCH("h_gn8_bwd");
  return GE_OK;
}

// nn.BatchNorm2d + nn.ReLU + nn.MaxPool2d(2, 2) in one pass each way (the last layer of a VGG16 stack, fpnseg.py:40-44): y = the
// pooled blocked fp16 map; the full-resolution activation is never written.  Backward as ge_h_bn_bwd_reduce / _apply, with the
// pooled gradient dy in place of da: the window's argmax is recomputed from z.  partial: C * B * ge_h_bn_slices(H * W / 4) * 2.
int ge_h_bn_relu_pool_fwd(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y,
                          int B, int C, int H, int W, void* stream) {
  GE_REQUIRE(z && y && mean && invstd && C % 32 == 0 && B > 0 && H % 2 == 0 && W % 2 == 0, "h_bn_relu_pool_fwd: bad arguments");
  dim3 grid(min(ge_cdiv((long long)(H / 2) * (W / 2) * 4, 256 * 2), 64), B * (C / 32));
  bnh_apply_pool_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half8*)z, mean, invstd, gamma, beta, (half8*)y, C, H, W);
  GE_CHECK_LAUNCH("h_bn_relu_pool_fwd");
  return GE_OK;
}
int ge_h_bn_relu_pool_bwd_reduce(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, float* partial, float* sums, float* dgamma, float* dbeta, int accumulate,
                                 float inv_scale, const float* dev_scale, int B, int C, int H, int W, void* stream) {
  GE_REQUIRE(dy && z && mean && invstd && partial && sums && C % 32 == 0 && B > 0 && H % 2 == 0 && W % 2 == 0,
             "h_bn_relu_pool_bwd_reduce: bad arguments");
  const int S = ge_h_bn_slices((H / 2) * (W / 2));
  dim3 grid(S, B * (C / 32));
  hipStream_t st = (hipStream_t)stream;
  bnh_pool_bwd_kernel<0><<<grid, 256, 0, st>>>((const half8*)dy, (const half8*)z, mean, invstd, gamma, beta, partial, nullptr, 0.f,
                                               nullptr, C, H, W, S, B * S, 1.f, nullptr);
  GE_CHECK_LAUNCH("h_bn_relu_pool_bwd_partial");
  bnh_bwd_finalize_kernel<<<ge_cdiv(C, 4), 256, 0, st>>>(partial, B * S, C, sums, dgamma, dbeta, accumulate, inv_scale, dev_scale);
  GE_CHECK_LAUNCH("h_bn_relu_pool_bwd_finalize");
  return GE_OK;
}
int ge_h_bn_relu_pool_bwd_apply(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, const float* sums, float inv_count, float scale, const float* dev_scale,
                                void* dz, int B, int C, int H, int W, void* stream) {
  GE_REQUIRE(dy && z && dz && mean && invstd && sums && C % 32 == 0 && B > 0 && H % 2 == 0 && W % 2 == 0,
             "h_bn_relu_pool_bwd_apply: bad arguments");
  dim3 grid(min(ge_cdiv((long long)(H / 2) * (W / 2) * 4, 256 * 2), 64), B * (C / 32));
  bnh_pool_bwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>((const half8*)dy, (const half8*)z, mean, invstd, gamma, beta, nullptr,
                                                                sums, inv_count, (half8*)dz, C, H, W, 1, 1, scale, dev_scale);
  GE_CHECK_LAUNCH("h_bn_relu_pool_bwd_apply");
  return GE_OK;
}

// 2x2 / stride 2 max-pool of a blocked fp16 tensor (H, W even) and its backward (first maximum in scan order)
int ge_h_maxpool2_fwd(const void* x, void* y, int B, int C, int H, int W, void* stream) {
  GE_REQUIRE(x && y && C % 32 == 0 && H % 2 == 0 && W % 2 == 0 && B > 0, "h_maxpool2_fwd: bad arguments");
  const long long total = (long long)B * (C / 32) * (H / 2) * (W / 2) * 4;
  poolh_fwd_kernel<<<ge_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const half8*)x, (half8*)y, H, W, total);
  GE_CHECK_LAUNCH("h_maxpool2_fwd");
  return GE_OK;
}
int ge_h_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int C, int H, int W, void* stream) {
  GE_REQUIRE(x && dy && dx && C % 32 == 0 && H % 2 == 0 && W % 2 == 0 && B > 0, "h_maxpool2_bwd: bad arguments");
  const long long total = (long long)B * (C / 32) * (H / 2) * (W / 2) * 4;
  poolh_bwd_kernel<<<ge_stream_grid(total, 256), 256, 0, (hipStream_t)stream>>>((const half8*)x, (const half8*)dy, (half8*)dx, H, W,
                                                                             total);
  GE_CHECK_LAUNCH("h_maxpool2_bwd");
  return GE_OK;
}

// ---- the stem: fp32 NCHW image with 1..4 channels -> blocked fp16 (nn.Conv2d(in_channels, 64, 3, padding=1), fpnseg.py:28) ----
static bool h_stem_ok(int B, int Cin, int Cout, int H, int W) {
  return B > 0 && (Cin == 1 || Cin == 3) && Cout % 32 == 0 && Cout >= 32 && Cout <= 128 && W % 64 == 0 && H > 0 &&
         2ull * B * Cout * H * W < 0xFFFF0000ull;
}
static int h_stem_grid(int B, int H, int W) {
  const int tiles = B * H * W / 64;
  return min(ge_cdiv(tiles, 4), 1024);
}
int ge_h_stem3x3_supported(int B, int Cin, int Cout, int H, int W) { return h_stem_ok(B, Cin, Cout, H, W) ? 1 : 0; }
// z (blocked fp16) = conv3x3(x fp32 NCHW, w fp32 OIHW) (+ bias); stats: [Cout][B * H * W / 64][3] as ge_h_conv3x3_fwd writes them
int ge_h_stem3x3_fwd(const float* x, const float* w, const float* bias, void* z, float* stats, int B, int Cin, int Cout, int H,
                     int W, void* stream) {
  GE_REQUIRE(x && w && z, "h_stem3x3_fwd: null pointer");
  GE_REQUIRE(h_stem_ok(B, Cin, Cout, H, W), "h_stem3x3_fwd: unsupported geometry B=%d Cin=%d Cout=%d %dx%d", B, Cin, Cout, H, W);
  const int tiles = B * H * W / 64, grid = h_stem_grid(B, H, W);
  if (Cin == 1)
    h_stem3x3_fwd_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(x, w, bias, (half8*)z, stats, Cout, H, W, tiles, tiles);
  else
    h_stem3x3_fwd_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(x, w, bias, (half8*)z, stats, Cout, H, W, tiles, tiles);
  GE_CHECK_LAUNCH("h_stem3x3_fwd");
  return GE_OK;
}
long long ge_h_stem3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W) {
  if (!h_stem_ok(B, Cin, Cout, H, W)) return 0;
  const int nslabs = min(h_stem_grid(B, H, W), 256);
  return (long long)(nslabs + (nslabs > H_SLAB_GROUPS ? H_SLAB_GROUPS : 0)) * 9 * Cout * Cin;
}
// dw[Cout][Cin][3][3] (+)= scale * weight gradient from the fp32 image and the blocked fp16 dz
int ge_h_stem3x3_wgrad(const float* x, const void* dz, float* dw, float* workspace, int B, int Cin, int Cout, int H, int W,
                       float scale, const float* dev_scale, int accumulate, void* stream) {
  GE_REQUIRE(x && dz && dw && workspace, "h_stem3x3_wgrad: null pointer");
  GE_REQUIRE(h_stem_ok(B, Cin, Cout, H, W), "h_stem3x3_wgrad: unsupported geometry");
  hipStream_t st = (hipStream_t)stream;
  const int tiles = B * H * W / 64, nslabs = min(h_stem_grid(B, H, W), 256), MC = Cout * Cin;
  if (Cin == 1)
    h_stem3x3_wgrad_kernel<1><<<nslabs, 256, 0, st>>>(x, (const half8*)dz, workspace, Cout, H, W, tiles);
  else
    h_stem3x3_wgrad_kernel<3><<<nslabs, 256, 0, st>>>(x, (const half8*)dz, workspace, Cout, H, W, tiles);
  GE_CHECK_LAUNCH("h_stem3x3_wgrad");
  if (nslabs > H_SLAB_GROUPS) {
    float* part = workspace + (size_t)nslabs * 9 * MC;
    const int per = ge_cdiv(nslabs, H_SLAB_GROUPS), groups = ge_cdiv(nslabs, per);
    h_slab_group_kernel<<<dim3(ge_cdiv(MC, 256), groups), 256, 0, st>>>(workspace, part, MC, nslabs, per);
    h_slab_reduce_kernel<<<ge_cdiv(MC, 256), 256, 0, st>>>(part, dw, Cout, Cin, groups, scale, accumulate, dev_scale);
  } else {
    h_slab_reduce_kernel<<<ge_cdiv(MC, 256), 256, 0, st>>>(workspace, dw, Cout, Cin, nslabs, scale, accumulate, dev_scale);
  }
  GE_CHECK_LAUNCH("h_stem_slab_reduce");
  return GE_OK;
}

// Device-resident loss scale hs = {scale, 1 / scale, bits of the largest |gradient| cast since the last update, unused}: every
// entry point above that takes `dev_scale` multiplies by hs[0] (casts to fp16; they also record the magnitude) or hs[1] (kernels
// that leave the fp16 domain) ON TOP of its host-side factor; pass NULL for a host-side scale only.  ge_h_scale_init writes
// {scale, 1 / scale, 0, 0}; ge_h_scale_update (once per step, between a backward and the next forward) sets the scale to the
// power of two that puts the recorded magnitude at `target`, within [lo, hi], and clears the record.
int ge_h_scale_init(float* hs, float scale, void* stream) {
  GE_REQUIRE(hs && scale > 0.f, "h_scale_init: bad arguments");
  const float v[4] = {scale, 1.f / scale, 0.f, 0.f};
  if (hipMemcpyAsync(hs, v, sizeof(v), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) {
    ge_set_error("h_scale_init: copy failed");
    return GE_ERR_LAUNCH;
  }
  return GE_OK;
}
int ge_h_scale_update(float* hs, float target, float lo, float hi, void* stream) {
  GE_REQUIRE(hs && target > 0.f && lo > 0.f && hi >= lo, "h_scale_update: bad arguments");
  h_scale_update_kernel<<<1, 1, 0, (hipStream_t)stream>>>(hs, target, lo, hi);
  GE_CHECK_LAUNCH("h_scale_update");
  return GE_OK;
}

// Self-description of the hardware transpose read the weight-gradient kernel relies on (tests assert the lane mapping):
// out[64][4] floats, see h_probe_tr_kernel
int ge_h_probe_tr(float* out, void* stream) {
  GE_REQUIRE(out, "h_probe_tr: null pointer");
  h_probe_tr_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out);
  GE_CHECK_LAUNCH("h_probe_tr");
  return GE_OK;
}

}  // extern "C"
