// fp32 3x3 / stride 1 / pad 1 WEIGHT GRADIENT as Winograd F(3x3, 2x2) on the fp32 matrix pipe (gfx950).
//
// The third pass of the reference's 3x3 convolutions (/root/reference/models/fpnseg.py:182-187 Bottleneck.conv2, :340-352 the FPN
// smoothing / head convs, :457-473 the discriminator towers): after the forward and the data gradient (ge_wino.hip, F(2x2, 3x3)) the
// direct weight-gradient kernel was the largest kernel family of the training step.  The mirrored algorithm: for a 2 x 2 tile e of
// dy (one output channel m) and the 4 x 4 patch d of x under it (one input channel c; the forward's patch),
//
//     dw[m][c] (3 x 3)  +=  A'^T [ (G' e G'^T)  (.)  (B^T d B) ] A'
//
// with B^T as in the forward, G' = the forward's A = [[1,0],[1,1],[1,-1],[0,-1]] and A'^T = the forward's G^T =
// [[1,1/2,1/2,0],[0,1/2,-1/2,0],[0,1/2,1/2,1]] (transposing the forward's trilinear form; checked against the direct correlation in
// numpy to 4e-16): 16 multiplications per tile and (m, c) pair instead of 36.  Per plane p of the 16 this is a GEMM
// S_p[m][c] += sum over tiles E_p[tile][m] * V_p[tile][c] with K = tiles = B * H * W / 4.
//
// One workgroup (4 waves) = 64 output channels x 32 input channels x all 16 planes, a K range of the tiles (split-K: slabs, reduced in
// split order by wnw_reduce_kernel: bit-reproducible).  Wave w owns planes 4 w .. 4 w + 3 (one ROW of the 4 x 4 plane matrix): 4 planes
// x 2 halves of the 64 channels = 8 accumulators of 32 x 32 = 128 registers, the forward's shape.  A chunk is a strip of EIGHT tiles
// along x (16 output pixels of one tile row of one image):
//   * every thread loads the 2 x 4 pixels of dy under two tiles of one output channel (two 16-byte loads: lanes = 4 tile pairs x 16
//     channels, a wave instruction covers whole 64-byte runs) and the 4 x 4 patch of x of one (tile, input channel) (16 loads, out of
//     image = out-of-range offset = 0 = the padding; lanes = 8 tiles x 8 channels),
//   * transforms both (G' e G'^T: 12 adds per tile; B^T d B: 32 adds) and writes the 16 plane values of each into LDS images
//     E[plane][tile pair][m][2] / V[plane][tile pair][c][2] whose tile-pair pitches (144 / 80 words) make the 8-byte writes of the E
//     role, the 4-byte writes of the V role and the 8-byte fragment reads conflict-free,
//   * per plane and half, two 8-byte A fragments (E) and one B fragment (V) feed the 32x32x2 MFMAs: 32 MFMAs per wave and chunk.
// ONE LDS stage of 56 KB (two workgroups per CU cover for each other; a second stage would leave one workgroup per CU, the
// configuration the forward kernel lost 25 % in): barrier - transform + write chunk k - barrier - issue the loads of chunk k + 1 -
// MFMAs of chunk k.  Epilogue: every wave folds its row of planes with A' (no exchange), the four rows meet in LDS one tap column at a
// time, A'^T, and the nine taps of the 64 x 32 block go to the split's slab in OIHW order.
#include "ge_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t wnw_rsrc_t;

#define WNW_OOB 0xFFFFFFFFu
constexpr int WNW_MT = 64, WNW_CT = 32, WNW_KT = 8;
constexpr int WNW_EPITCH = 144, WNW_VPITCH = 80;                   // words between tile pairs of a plane
constexpr int WNW_EPLANE = 4 * WNW_EPITCH, WNW_VPLANE = 4 * WNW_VPITCH;
constexpr int WNW_LDS_FLOATS = 16 * (WNW_EPLANE + WNW_VPLANE);     // 14 336 floats = 56 KB
constexpr int WNW_XROW = 33;                                       // padded row of the epilogue exchange [4 rows][64 m][33]
static_assert(4 * WNW_MT * WNW_XROW <= WNW_LDS_FLOATS, "exchange buffer");

__device__ __forceinline__ float wnw_load(wnw_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ f32x4 wnw_load4(wnw_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ int wnw_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct WinoWgradParams {
  const float* x;      // [B][C][H][W]
  const float* dy;     // [B][M][H][W]
  float* slab;         // [splits][M][C][9]
  int B, C, M, H, W;
  int tiles_m, tiles_c;      // M / 64, C / 32
  int splits, chunks, split_chunks;      // chunks = B * (H / 2) * (W / 16); split s takes chunks [s * split_chunks, ...)
  int th, sx;                // H / 2 tile rows, W / 16 strips per tile row
};

__global__ __launch_bounds__(256, 2) void wino3x3_wgrad_kernel(WinoWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sE = lds;                            // [16][4 tile pairs][144]: (m, tile & 1) at 2 m + (tile & 1)
  float* sV = lds + 16 * WNW_EPLANE;          // [16][4 tile pairs][80]:  (c, tile & 1) at 2 c + (tile & 1)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  // split-major: the tile workgroups of one K range are neighbours after the XCD remap and share its x / dy in one L2
  const int gid = wnw_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_m * p.tiles_c;
  const int split = gid / ntile, trem = gid - split * ntile;
  const int tm = trem / p.tiles_c, tc = trem - tm * p.tiles_c;
  const int m0 = tm * WNW_MT, c0 = tc * WNW_CT;
  const int HW = p.H * p.W;
  const int g_begin = split * p.split_chunks;
  const int g_end = min(p.chunks, g_begin + p.split_chunks);

  const wnw_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (uint32_t)((size_t)p.B * p.C * HW * 4u), 0x00020000);
  const wnw_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (uint32_t)((size_t)p.B * p.M * HW * 4u), 0x00020000);

  // ---- E role: output channel em, tile pair eq (tiles 2 eq, 2 eq + 1 of the strip);  V role: input channel vc, tile vt
  const int eq = tid & 3, em = tid >> 2;
  const int vt = tid & 7, vc = tid >> 3;
  const uint32_t e_thread = (uint32_t)(((m0 + em) * p.H) * p.W + 4 * eq);      // + image / row / strip part of the chunk
  const int v_thread = ((c0 + vc) * p.H) * p.W + 2 * vt - 1;                    // (may be negative: resolved per chunk)

  f32x4 er0, er1;      // raw dy rows of the thread's two tiles
  float d[16];         // raw 4 x 4 patch of x
  auto issue_loads = [&](int g) {
    // chunk g -> (image b, tile row ty, strip sxi); past the end of the split: everything out of range = zeros
    const bool live = g < g_end;
    const int b = g / (p.th * p.sx), r1 = g - b * (p.th * p.sx);
    const int ty = r1 / p.sx, sxi = r1 - ty * p.sx;
    const uint32_t eoff = live ? (e_thread + (uint32_t)((b * p.M * p.H + 2 * ty) * p.W + 16 * sxi)) * 4u : WNW_OOB;
    er0 = wnw_load4(yrs, eoff);
    er1 = wnw_load4(yrs, live ? eoff + (uint32_t)p.W * 4u : WNW_OOB);
    const int vbase = v_thread + (b * p.C * p.H + 2 * ty - 1) * p.W + 16 * sxi;
    const int iy0 = 2 * ty - 1, ix0 = 16 * sxi + 2 * vt - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rok = live && (unsigned)(iy0 + r) < (unsigned)p.H;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = rok && (unsigned)(ix0 + s) < (unsigned)p.W;
        d[r * 4 + s] = wnw_load(xrs, ok ? (uint32_t)(vbase + r * p.W + s) * 4u : WNW_OOB);
      }
    }
  };
  // G' e G'^T of one tile e = [[a, b], [c, d]]: rows (a, b), (a + c, b + d), (a - c, b - d), (-c, -d), then the same on the columns
  auto stage = [&]() {
    // E: tiles 2 eq (pixels .x .y of both rows) and 2 eq + 1 (.z .w) -> [plane][eq][em][0 / 1]
    float* e = sE + eq * WNW_EPITCH + em * 2;
    float ea[16], eb[16];
    {
      const float a = er0.x, b = er0.y, c = er1.x, dd = er1.y;
      const float rx[4] = {a, a + c, a - c, -c}, ry[4] = {b, b + dd, b - dd, -dd};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ea[i * 4 + 0] = rx[i];
        ea[i * 4 + 1] = rx[i] + ry[i];
        ea[i * 4 + 2] = rx[i] - ry[i];
        ea[i * 4 + 3] = -ry[i];
      }
    }
    {
      const float a = er0.z, b = er0.w, c = er1.z, dd = er1.w;
      const float rx[4] = {a, a + c, a - c, -c}, ry[4] = {b, b + dd, b - dd, -dd};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        eb[i * 4 + 0] = rx[i];
        eb[i * 4 + 1] = rx[i] + ry[i];
        eb[i * 4 + 2] = rx[i] - ry[i];
        eb[i * 4 + 3] = -ry[i];
      }
    }
#pragma unroll
    for (int pl = 0; pl < 16; ++pl) {
      f32x2 v2;
      v2.x = ea[pl];
      v2.y = eb[pl];
      *(f32x2*)(e + pl * WNW_EPLANE) = v2;
    }
    // V: B^T d B of the thread's (tile, channel) -> [plane][vt >> 1][vc][vt & 1]
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
      t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
      t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
      t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
    }
    float* v = sV + (vt >> 1) * WNW_VPITCH + vc * 2 + (vt & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[(i * 4 + 0) * WNW_VPLANE] = t[i * 4 + 0] - t[i * 4 + 2];
      v[(i * 4 + 1) * WNW_VPLANE] = t[i * 4 + 1] + t[i * 4 + 2];
      v[(i * 4 + 2) * WNW_VPLANE] = t[i * 4 + 2] - t[i * 4 + 1];
      v[(i * 4 + 3) * WNW_VPLANE] = t[i * 4 + 1] - t[i * 4 + 3];
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][mh][r] = 0.f;

  issue_loads(g_begin);
  for (int g = g_begin; g < g_end; ++g) {
    __syncthreads();      // every wave has read the fragments of the previous chunk
    stage();
    __syncthreads();
    issue_loads(g + 1);   // (past the end: out of range, zeros, no traffic)
    const float* ea = sE + (4 * wave) * WNW_EPLANE + hi * WNW_EPITCH + li * 2;
    const float* va = sV + (4 * wave) * WNW_VPLANE + hi * WNW_VPITCH + li * 2;
#pragma unroll
    for (int P = 0; P < 2; ++P) {      // tile pairs 2 P + hi
      f32x2 fa[4][2], fb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[q][0] = *(const f32x2*)(ea + q * WNW_EPLANE + 2 * P * WNW_EPITCH);
        fa[q][1] = *(const f32x2*)(ea + q * WNW_EPLANE + 2 * P * WNW_EPITCH + 64);
        fb[q] = *(const f32x2*)(va + q * WNW_VPLANE + 2 * P * WNW_VPITCH);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0][s], fb[q][s], acc[q][0], 0, 0, 0);
          acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1][s], fb[q][s], acc[q][1], 0, 0, 0);
        }
    }
  }
  __syncthreads();

  // ---- epilogue.  acc[q][mh][r] = S[plane 4 wave + q][m = 32 mh + (r & 3) + 8 (r >> 2) + 4 hi][c = li].  Right factor A' inside the
  // wave (its row of planes): R[b] for the three tap columns; then the four rows meet in LDS, one tap column at a time
  float* sX = lds;      // [4 rows][64 m][33]
  float* out = p.slab + ((size_t)split * p.M + m0) * (size_t)p.C * 9 + (size_t)c0 * 9;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (b) __syncthreads();
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s0 = acc[0][mh][r], s1 = acc[1][mh][r], s2 = acc[2][mh][r], s3 = acc[3][mh][r];
        const float v = b == 0 ? s0 + 0.5f * (s1 + s2) : b == 1 ? 0.5f * (s1 - s2) : 0.5f * (s1 + s2) + s3;
        const int m = 32 * mh + (r & 3) + 8 * (r >> 2) + 4 * hi;
        sX[(wave * WNW_MT + m) * WNW_XROW + li] = v;
      }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int e = tid + 256 * n, m = e >> 5, c = e & 31;
      const float r0 = sX[(0 * WNW_MT + m) * WNW_XROW + c], r1 = sX[(1 * WNW_MT + m) * WNW_XROW + c];
      const float r2 = sX[(2 * WNW_MT + m) * WNW_XROW + c], r3 = sX[(3 * WNW_MT + m) * WNW_XROW + c];
      float* o = out + ((size_t)m * p.C + c) * 9 + b;
      o[0] = r0 + 0.5f * (r1 + r2);
      o[3] = 0.5f * (r1 - r2);
      o[6] = 0.5f * (r1 + r2) + r3;
    }
  }
}

// dw (+)= slab[0] + slab[1] + ...  (split order: bit-reproducible)
__global__ __launch_bounds__(256) void wnw_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int splits,
                                                         int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s0 = accumulate ? dw[i] : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      s0 += slab[(size_t)k * n + i];
      s1 += slab[(size_t)(k + 1) * n + i];
      s2 += slab[(size_t)(k + 2) * n + i];
      s3 += slab[(size_t)(k + 3) * n + i];
    }
    for (; k < splits; ++k) s0 += slab[(size_t)k * n + i];
    dw[i] = (s0 + s1) + (s2 + s3);
  }
}

static bool wnw_covered(int B, int C, int M, int H, int W) {
  if (B <= 0 || C <= 0 || M <= 0 || C % WNW_CT || M % WNW_MT || W % 16 || H % 2) return false;
  return (unsigned long long)B * C * H * W * 4ull < 0xFFFF0000ull && (unsigned long long)B * M * H * W * 4ull < 0xFFFF0000ull;
}
static int wnw_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// K splits: enough workgroups for two per CU, at least WNW_MIN_CHUNKS chunks of 8 tiles each (the prologue + the 64 x 32 x 9 slab
// write are per workgroup); 0: the direct kernel keeps the layer.  Measured (tools/bench_wino_wgrad.py, profiles/r05_wino_wgrad_
// microbench.txt): every layer that reaches 512 workgroups wins (x1.06 .. x1.52), the 128-workgroup layers of an 8-frame step
// (64 -> 64 @ 64 x 64, 128 -> 128 @ 32 x 32, 256 -> 256 @ 16 x 16) lose (x0.7): routed from WNW_MIN_GRID workgroups.
static int wnw_plan(int B, int C, int M, int H, int W, int& chunks, bool routing) {
  if (!wnw_covered(B, C, M, H, W)) return 0;
  static const int target = wnw_env("GE_WNW_TARGET", 512), min_chunks = wnw_env("GE_WNW_MIN_CHUNKS", 16),
                   min_grid = wnw_env("GE_WNW_MIN_GRID", 384), forced = wnw_env("GE_WNW_SPLITS", 0);
  chunks = B * (H / 2) * (W / 16);
  const int tiles = (M / WNW_MT) * (C / WNW_CT);
  int s = forced > 0 ? forced : (target + tiles - 1) / tiles;
  if (s > chunks / min_chunks) s = chunks / min_chunks;
  if (s > 1024) s = 1024;
  if (s > chunks) s = chunks;
  if (routing && forced <= 0 && (s < 1 || (long long)s * tiles < min_grid)) return 0;
  return s < 1 ? 1 : s;      // (a covered layer the routing plan would leave to the direct kernel: the caller insists)
}

extern "C" {

// 1 when ge_wino3x3_wgrad covers the layer (x [B][C][H][W], dy [B][M][H][W]: C % 32 == 0, M % 64 == 0, W % 16 == 0, H even) and it has
// enough tiles to fill the chip
int ge_wino3x3_wgrad_supported(int B, int C, int M, int H, int W) {
  int chunks = 0;
  return wnw_plan(B, C, M, H, W, chunks, true) > 0 ? 1 : 0;
}
// covered geometry, whatever the grid size (tests / microbenches); the splits / workspace the entry point then uses
int ge_wino3x3_wgrad_covered(int B, int C, int M, int H, int W) { return wnw_covered(B, C, M, H, W) ? 1 : 0; }
int ge_wino3x3_wgrad_splits(int B, int C, int M, int H, int W) {
  int chunks = 0;
  return wnw_plan(B, C, M, H, W, chunks, false);
}
// floats of workspace (the K-split slabs)
long long ge_wino3x3_wgrad_workspace(int B, int C, int M, int H, int W) {
  int chunks = 0;
  const int s = wnw_plan(B, C, M, H, W, chunks, false);
  return (long long)s * M * C * 9;
}
// dw[M][C][3][3] (+)= weight gradient of y = conv3x3(x; stride 1, pad 1) from x [B][C][H][W] and dy [B][M][H][W]
// accumulate bit 0: add to dw; bit 1: leave the slabs in the workspace (stride M * C * 9) for ge_slab_reduce_batched
int ge_wino3x3_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int C, int M, int H, int W, int accumulate,
                     void* stream) {
  GE_REQUIRE(x && dy && dw && workspace, "wino3x3_wgrad: null pointer");
  int chunks = 0;
  const int splits0 = wnw_plan(B, C, M, H, W, chunks, false);
  GE_REQUIRE(splits0 > 0, "wino3x3_wgrad: unsupported geometry B=%d C=%d M=%d %dx%d", B, C, M, H, W);
  hipStream_t st = (hipStream_t)stream;
  WinoWgradParams p;
  p.x = x;
  p.dy = dy;
  p.slab = workspace;
  p.B = B;
  p.C = C;
  p.M = M;
  p.H = H;
  p.W = W;
  p.tiles_m = M / WNW_MT;
  p.tiles_c = C / WNW_CT;
  p.chunks = chunks;
  p.split_chunks = (chunks + splits0 - 1) / splits0;
  p.splits = (chunks + p.split_chunks - 1) / p.split_chunks;      // no empty split
  p.th = H / 2;
  p.sx = W / 16;
  const int grid = p.tiles_m * p.tiles_c * p.splits;
  const size_t smem = WNW_LDS_FLOATS * sizeof(float);
  static GeLdsAttr attr;
  const int rc = ge_set_max_lds(attr, (const void*)wino3x3_wgrad_kernel, (int)smem, "wino3x3_wgrad_kernel");
  if (rc != GE_OK) return rc;
  wino3x3_wgrad_kernel<<<grid, 256, smem, st>>>(p);
  ge_note_kernel("wino3x3_wgrad_kernel");
  GE_CHECK_LAUNCH("wino3x3_wgrad");
  ge_record_split_event(st);
  if (accumulate & 2) return GE_OK;      // the caller reduces the slabs later (ge_slab_reduce_batched)
  const long long n = (long long)M * C * 9;
  wnw_reduce_kernel<<<ge_stream_grid(n, 256), 256, 0, st>>>(workspace, dw, n, p.splits, accumulate & 1);
  GE_CHECK_LAUNCH("wino3x3_wgrad_reduce");
  return GE_OK;
}

}  // extern "C"
