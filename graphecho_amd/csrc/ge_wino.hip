// fp32 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on the fp32 matrix pipe (gfx950).
//
// Replaces, on the large layers, the direct implicit-GEMM kernels of ge_mfma.hip for the reference's 3x3 convolutions
// (/root/reference/models/fpnseg.py:182-187 Bottleneck.conv2, :340-352 the FPN smoothing / head convs): 16 multiplications per
// 2x2 outputs instead of 36.  gfx950 has no reduced-precision fp32 matrix mode, so for an fp32 convolution the only way past the
// 157 TFLOP/s of v_mfma_f32_32x32x2_f32 is to do fewer multiplications.
//
// One workgroup (4 waves) = 32 tiles (2 x 16 or 4 x 8 tiles = 4 x 32 or 8 x 16 output pixels) x 64 output channels.  Per chunk
// of 8 input channels: a thread loads the 4 x 4 patch of (tile, channel), transforms it (B^T d B, 32 adds) and writes the 16
// values into 16 LDS planes V[plane][k / 4][tile][k % 4]; the transformed weights (packed once per weight version by
// wino_pack_kernel in exactly the LDS order) arrive by LDS-DMA, one HALF chunk (16 KB) at a time, two half-chunks ahead, into three
// rotating buffers.  Wave w owns planes 4 w .. 4 w + 3: per half-chunk and plane one 8-byte A fragment (tile x 2 channels per
// k-lane) and two B fragments feed four 32x32x2 MFMAs.  Epilogue: the 16 planes meet in LDS and each thread applies A^T M A for
// its (tile, channel) pairs, adds bias / addend and stores 2 x 2 outputs.  History of the layout: DESIGN.md 4.
#include "ge_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int wn_u32x4 __attribute__((ext_vector_type(4)));

#define WN_OOB 0xFFFFFFFFu
constexpr int WN_KC = 8, WN_TILES = 32, WN_MC = 64;
constexpr int WN_VSTAGE = 16 * WN_KC * WN_TILES;      // 4096 floats
constexpr int WN_USTAGE = 16 * WN_KC * WN_MC;         // 8192 floats
constexpr int WN_STAGE = WN_VSTAGE + WN_USTAGE;       // 12288 floats = 48 KB
constexpr int WN_MROW = 33;                           // padded tile row of the epilogue exchange
constexpr int WN_LDS_FLOATS = 2 * WN_VSTAGE + 3 * (WN_USTAGE / 2);   // 80 KB: 2 x V + 3 x half-chunk U; the epilogue exchange (66 KB) reuses it
static_assert(16 * 32 * WN_MROW <= WN_LDS_FLOATS, "exchange buffer");

__device__ __forceinline__ void wn_dma16(wn_u32x4 rs, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}
template <int N>
__device__ __forceinline__ void wn_vm_wait() {
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
}
__device__ __forceinline__ wn_u32x4 wn_rsrc(const void* p, uint32_t bytes) {
  const unsigned long long ad = (unsigned long long)p;
  wn_u32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)ad);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)(ad >> 32) & 0xFFFFu);
  rs.z = __builtin_amdgcn_readfirstlane(bytes);
  rs.w = 0x00020000u;
  return rs;
}
typedef __amdgpu_buffer_rsrc_t wn_rsrc_t;
__device__ __forceinline__ float wn_load(wn_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ int wn_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct WinoParams {
  const float* x;        // [B][C][H][W]
  const float* u;        // [M / 64][C / 8][2 halves][16 planes][2][64][2]
  const float* bias;     // [M] or null
  const float* addend;   // [B][M][H][W] or null
  float* y;              // [B][M][H][W]
  int B, C, M, H, W;
  int blocks_x, blocks_y, tiles_m;
  uint32_t u_bytes;
  int order;
};

// TXT: tiles per block row (16: 2 x 16 tiles, 8: 4 x 8 tiles)
//
// LDS: V of a chunk of 8 channels (16 KB) twice + U of HALF a chunk (16 KB: MFMA steps 2 s, 2 s + 1 = channels 4 hi + 2 s + {0, 1})
// three times (the DMA of half-chunk h + 2 is issued when h starts: a half-chunk is 1024 MFMA cycles per wave = 0.43 us, less than
// one trip to L2 / HBM) = 80 KB, the epilogue exchange (66 KB) inside it: two workgroups per CU -- with one, every serial section of a workgroup (first
// loads, the output transform and its stores) left the matrix pipe idle: 1.07 ms on 256 -> 256 @ 64 x 64 x 32 against 0.44 of MFMA work.
template <int TXT, int DBG = 0>
__global__ __launch_bounds__(256, 2) void wino3x3_kernel(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sV = lds;                       // [2][16 planes][2 hi][32 tiles][4]
  float* sU = lds + 2 * WN_VSTAGE;       // [3][16 planes][2 hi][64 m][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int lid = wn_xcd_remap(blockIdx.x, gridDim.x);
  // order 0: the channel tiles of one spatial block are neighbours (they share the input patch in L2); 1: one channel tile's
  // spatial blocks are neighbours (they share its 16 * C * 64 transformed filters).  GE_WN_ORDER=1, measured: 0.750 - 0.766 vs
  // 0.762 - 0.774 ms on 256 -> 256 @ 64 x 64 x 32, nothing on the other layers -- neither operand's locality bounds the kernel
  const int nsp = gridDim.x / p.tiles_m;
  const int tm = p.order ? lid / nsp : lid % p.tiles_m, sp = p.order ? lid - tm * nsp : lid / p.tiles_m;
  const int per_img = p.blocks_x * p.blocks_y;
  const int b = sp / per_img, srem = sp - b * per_img;
  const int by = srem / p.blocks_x, bx = srem - by * p.blocks_x;
  constexpr int TYT = WN_TILES / TXT;
  const int y0 = by * (2 * TYT), x0 = bx * (2 * TXT), m0 = tm * WN_MC;
  const int HW = p.H * p.W;
  const int nch = p.C / WN_KC;

  const wn_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)b * p.C * HW), 0,
                                                          (uint32_t)p.C * (uint32_t)HW * 4u, 0x00020000);
  const wn_u32x4 urs = wn_rsrc(p.u, p.u_bytes);
  const uint32_t lds_u = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)sU;

  // ---- loader role: (tile, channel of the chunk)
  const int lt = tid & 31, lc = tid >> 5;
  const int ltx = lt % TXT, lty = lt / TXT;
  uint32_t poff[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy = y0 + 2 * lty - 1 + i, ix = x0 + 2 * ltx - 1 + j;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      poff[i * 4 + j] = ok ? (uint32_t)((lc * p.H + iy) * p.W + ix) * 4u : WN_OOB;
    }
  const uint32_t chunk_step = (uint32_t)WN_KC * (uint32_t)HW * 4u;
  const uint32_t u_block = (uint32_t)(tm * nch) * (WN_USTAGE * 4u) + (uint32_t)lane * 16u;
  const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane(wave);

  float d[16];
  auto load_patch = [&](int ch) {
    const uint32_t add = (uint32_t)ch * chunk_step;
#pragma unroll
    for (int e = 0; e < 16; ++e) d[e] = wn_load(xrs, __builtin_elementwise_add_sat(poff[e], add));
  };
  // half-chunk h = 2 * chunk + s of the transformed filters -> U buffer h & 1 (16 KB = 16 pieces of 1 KB, four per wave)
  auto issue_u = [&](int h, int ub) {
    const uint32_t gbase = u_block + (uint32_t)h * (WN_USTAGE * 2u);
    const uint32_t lbase = lds_u + (uint32_t)ub * (WN_USTAGE * 2u);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t piece = (uint32_t)e * 4u + wu;
      wn_dma16(urs, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + piece * 1024u)), gbase + piece * 1024u);
    }
  };
  auto stage_v = [&](int st) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    float* v = sV + st * WN_VSTAGE + (lc >> 2) * 128 + lt * 4 + (lc & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(i * 4 + 0) * 256] = t[i * 4 + 0] - t[i * 4 + 2];
      v[(i * 4 + 1) * 256] = t[i * 4 + 1] + t[i * 4 + 2];
      v[(i * 4 + 2) * 256] = t[i * 4 + 2] - t[i * 4 + 1];
      v[(i * 4 + 3) * 256] = t[i * 4 + 1] - t[i * 4 + 3];
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][nb][r] = 0.f;

  const int nhalf = 2 * nch;
  issue_u(0, 0);
  issue_u(1, 1);
  load_patch(0);
  stage_v(0);
  wn_vm_wait<0>();
  __syncthreads();

  int ub = 0;      // U buffer of half-chunk h = h % 3
  // One half-chunk: 12 fragment reads + 16 MFMAs.  The blocks are straight-line on purpose and carry an explicit issue order
  // (sched_group_barrier): a wave issues in order, so VALU / memory instructions placed behind the sixteen MFMAs would only start
  // when the last one has been issued -- measured: the patch loads + transform + V writes then ADD their 0.25 ms to the 0.58 ms of
  // the MFMA loop instead of hiding under it.
  auto frags_mfma = [&](int ch, int sh, int ubuf) {
    const float* sv = sV + (ch & 1) * WN_VSTAGE + (4 * wave) * 256 + hi * 128 + li * 4 + 2 * sh;
    const float* su = sU + ubuf * (WN_USTAGE / 2) + (4 * wave) * 256 + hi * 128 + li * 2;
    f32x2 fa[4], fb0[4], fb1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[q] = *(const f32x2*)(sv + q * 256);
      fb0[q] = *(const f32x2*)(su + q * 256);
      fb1[q] = *(const f32x2*)(su + q * 256 + 64);
    }
    if (!(DBG & 1)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][kk], fb0[q][kk], acc[q][0], 0, 0, 0);
          acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][kk], fb1[q][kk], acc[q][1], 0, 0, 0);
        }
    } else {
      asm volatile("" ::"v"(fa[0]), "v"(fb0[0]), "v"(fb1[3]));
    }
  };
  for (int ch = 0; ch + 1 < nch; ++ch) {
    // ---- first half: DMA of half-chunk 2 ch + 2, the next chunk's sixteen patch loads between the MFMAs
    if (!(DBG & 8)) issue_u(2 * ch + 2, ub == 0 ? 2 : ub - 1);
    if (!(DBG & 2)) load_patch(ch + 1);
    frags_mfma(ch, 0, ub);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);      // the fragment reads first
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);    // one address add
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one patch load
    }
    __builtin_amdgcn_sched_barrier(0);
    wn_vm_wait<20>();      // half-chunk 2 ch + 1 has landed; the four DMA instructions and sixteen loads just issued may fly on
    __syncthreads();
    ub = ub == 2 ? 0 : ub + 1;
    // ---- second half: DMA of half-chunk 2 ch + 3, transform + V writes of the next chunk between the MFMAs
    if (!(DBG & 8)) issue_u(2 * ch + 3, ub == 0 ? 2 : ub - 1);
    frags_mfma(ch, 1, ub);
    if (!(DBG & 4)) stage_v((ch + 1) & 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // one (paired) V write
    }
    __builtin_amdgcn_sched_barrier(0);
    wn_vm_wait<4>();
    __syncthreads();
    ub = ub == 2 ? 0 : ub + 1;
  }
  // ---- last chunk: nothing left to fetch
  frags_mfma(nch - 1, 0, ub);
  __builtin_amdgcn_sched_barrier(0);
  wn_vm_wait<0>();
  __syncthreads();
  ub = ub == 2 ? 0 : ub + 1;
  frags_mfma(nch - 1, 1, ub);
  __syncthreads();

  // ---- epilogue: the 16 planes of a (tile, channel) meet in LDS, 32 channels at a time
  if (DBG & 16) return;
  float* sM = lds;
  const int et = lane & 31;
  const int etx = et % TXT, ety = et / TXT;
  const int oy = y0 + 2 * ety, ox = x0 + 2 * etx;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    if (nb) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tile = (r & 3) + 8 * (r >> 2) + 4 * hi;
        sM[((4 * wave + q) * 32 + li) * WN_MROW + tile] = acc[q][nb][r];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int cl = wave * 8 + it * 2 + hi;
      float m[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) m[xi] = sM[(xi * 32 + cl) * WN_MROW + et];
      float r0[4], r1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r0[j] = m[0 * 4 + j] + m[1 * 4 + j] + m[2 * 4 + j];
        r1[j] = m[1 * 4 + j] - m[2 * 4 + j] - m[3 * 4 + j];
      }
      f32x2 o0, o1;
      o0.x = r0[0] + r0[1] + r0[2];
      o0.y = r0[1] - r0[2] - r0[3];
      o1.x = r1[0] + r1[1] + r1[2];
      o1.y = r1[1] - r1[2] - r1[3];
      const int mch = m0 + nb * 32 + cl;
      if (p.bias) {
        const float bv = p.bias[mch];
        o0.x += bv;
        o0.y += bv;
        o1.x += bv;
        o1.y += bv;
      }
      const size_t o = ((size_t)b * p.M + mch) * HW + (size_t)oy * p.W + ox;
      if (p.addend) {
        const f32x2 a0 = *(const f32x2*)(p.addend + o), a1 = *(const f32x2*)(p.addend + o + p.W);
        o0 += a0;
        o1 += a1;
      }
      *(f32x2*)(p.y + o) = o0;
      *(f32x2*)(p.y + o + p.W) = o1;
    }
  }
}

// u[m / 64][c / 8][s][plane][hi][m % 64][e] = (G g G^T)[plane] with c % 8 = 4 hi + 2 s + e, g = w[m][c] (transposed = 0) or the data gradient's
// filter w[c][m] rotated by 180 degrees (transposed = 1: m runs over the ORIGINAL input channels, c over the original output channels)
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int M, int C,
                                                        int transposed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * C) return;
  const int m = idx / C, c = idx - m * C;
  float g[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) g[a] = transposed ? w[((size_t)c * M + m) * 9 + (8 - a)] : w[((size_t)m * C + c) * 9 + a];
  float t[12];      // G g: 4 x 3
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0 * 3 + j] = g[0 * 3 + j];
    t[1 * 3 + j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
    t[2 * 3 + j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
    t[3 * 3 + j] = g[2 * 3 + j];
  }
  const int nch = C / WN_KC;
  const int c8 = c % WN_KC, hi = c8 >> 2, sh = (c8 >> 1) & 1, e = c8 & 1;
  float* o = u + ((size_t)(m / WN_MC) * nch + c / WN_KC) * WN_USTAGE + sh * (WN_USTAGE / 2) + hi * 128 + (m % WN_MC) * 2 + e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[(i * 4 + 0) * 256] = t[i * 3 + 0];
    o[(i * 4 + 1) * 256] = 0.5f * (t[i * 3 + 0] + t[i * 3 + 1] + t[i * 3 + 2]);
    o[(i * 4 + 2) * 256] = 0.5f * (t[i * 3 + 0] - t[i * 3 + 1] + t[i * 3 + 2]);
    o[(i * 4 + 3) * 256] = t[i * 3 + 2];
  }
}

static int wn_txt(int H, int W) {
  if (W % 32 == 0 && H % 4 == 0) return 16;
  if (W % 16 == 0 && H % 8 == 0) return 8;
  return 0;
}

extern "C" {

// 1 when ge_wino3x3_fwd covers the layer (C = reduction channels, M = output channels of the pass) AND the grid fills the chip
int ge_wino3x3_supported(int B, int C, int M, int H, int W) {
  if (B <= 0 || C % WN_KC || M % WN_MC || !wn_txt(H, W)) return 0;
  if (4ull * C * H * W >= 0xFFFF0000ull || 64ull * C * M >= 0xFFFF0000ull) return 0;
  const long long blocks = (long long)B * (H * W / 128) * (M / WN_MC);
  return blocks >= 512 ? 1 : 0;      // (256 -> 256 @ 16 x 16 x 32 = 256 workgroups: 1.05x the direct kernel, and it loses the moments epilogue)
}
long long ge_wino3x3_weight_floats(int C, int M) { return 16ll * C * M; }
// transformed filters of a pass with M output and C reduction channels from w (OIHW, 3 x 3): transposed = 0: w is [M][C][3][3]
// (forward); transposed = 1: w is [C][M][3][3] (data gradient: M = the layer's input channels, C = its output channels)
int ge_wino3x3_pack_weight(const float* w, float* u, int M, int C, int transposed, void* stream) {
  GE_REQUIRE(w && u && M % WN_MC == 0 && C % WN_KC == 0, "wino3x3_pack_weight: bad arguments");
  wino_pack_kernel<<<ge_cdiv((long long)M * C, 256), 256, 0, (hipStream_t)stream>>>(w, u, M, C, transposed);
  GE_CHECK_LAUNCH("wino_pack");
  return GE_OK;
}
// y = conv3x3(x; stride 1, pad 1) (+ bias) (+ addend): x [B][C][H][W], y / addend [B][M][H][W], u from ge_wino3x3_pack_weight
int ge_wino3x3_fwd(const float* x, const float* u, const float* bias, const float* addend, float* y, int B, int C, int M, int H,
                   int W, void* stream) {
  GE_REQUIRE(x && u && y, "wino3x3_fwd: null pointer");
  const int txt = wn_txt(H, W);
  GE_REQUIRE(B > 0 && C % WN_KC == 0 && M % WN_MC == 0 && txt && 4ull * C * H * W < 0xFFFF0000ull && 64ull * C * M < 0xFFFF0000ull,
             "wino3x3_fwd: unsupported geometry B=%d C=%d M=%d %dx%d", B, C, M, H, W);
  WinoParams p;
  p.x = x;
  p.u = u;
  p.bias = bias;
  p.addend = addend;
  p.y = y;
  p.B = B;
  p.C = C;
  p.M = M;
  p.H = H;
  p.W = W;
  p.blocks_x = W / (2 * txt);
  p.blocks_y = H / (2 * (WN_TILES / txt));
  p.tiles_m = M / WN_MC;
  p.u_bytes = (uint32_t)(64ull * C * M);
  static const int order_env = []() {
    const char* e = getenv("GE_WN_ORDER");
    return e ? atoi(e) : 0;
  }();
  p.order = order_env;
  const int grid = B * p.blocks_x * p.blocks_y * p.tiles_m;
  const size_t smem = WN_LDS_FLOATS * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)wino3x3_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)wino3x3_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  static const int dbg = []() {
    const char* e = getenv("GE_WN_DBG");
    return e ? atoi(e) : 0;
  }();
  if (dbg && txt == 16) {      // tuning only (wrong results): 1 no MFMAs, 2 no patch loads, 4 no transform / V writes, 8 no U DMA, 16 no epilogue
#define WN_DBG_CASE(D)                                                                                                    \
  case D:                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)wino3x3_kernel<16, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    wino3x3_kernel<16, D><<<grid, 256, smem, (hipStream_t)stream>>>(p);                                                   \
    break;
    switch (dbg) {
      WN_DBG_CASE(1) WN_DBG_CASE(6) WN_DBG_CASE(14) WN_DBG_CASE(15) WN_DBG_CASE(16) WN_DBG_CASE(31)
      default: {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)wino3x3_kernel<16, 0>, 256, smem);
        fprintf(stderr, "wino3x3_kernel<16>: %d workgroups per CU, %zu bytes of LDS\n", nb, smem);
        wino3x3_kernel<16><<<grid, 256, smem, (hipStream_t)stream>>>(p);
      }
    }
#undef WN_DBG_CASE
  } else if (txt == 16) wino3x3_kernel<16><<<grid, 256, smem, (hipStream_t)stream>>>(p);
  else wino3x3_kernel<8><<<grid, 256, smem, (hipStream_t)stream>>>(p);
  ge_note_kernel("wino3x3_kernel<%d, 0>", txt);      // as rocprofv3 prints it
  GE_CHECK_LAUNCH("wino3x3");
  return GE_OK;
}

}  // extern "C"
