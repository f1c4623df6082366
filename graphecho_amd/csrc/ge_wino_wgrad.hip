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
//     channels, a wave instruction covers whole 64-byte runs) and the 4 x 4 patch of x of one (tile, input channel) as four 16-byte
//     rows (4-byte aligned; out of image = out-of-range offset = 0 = the padding; lanes = 8 tiles x 8 channels),
//   * transforms both (G' e G'^T: 12 adds per tile; B^T d B: 32 adds) and writes the 16 plane values of each into LDS images
//     E[plane][tile pair][m][2] / V[plane][tile pair][c][2] whose tile-pair pitches (144 / 80 words) suit the 8-byte writes of the E
//     role, the 4-byte writes of the V role and the 8-byte fragment reads,
//   * per plane and half, two 8-byte A fragments (E) and one B fragment (V) feed the 32x32x2 MFMAs: 32 MFMAs per wave and chunk.
// Two kernels, chosen per layer by wnw_plan():
//   wino3x3_wgrad_kernel     256 threads, ONE LDS stage of 56 KB, two workgroups per CU: barrier - transform + write chunk k - barrier -
//                            MFMAs of chunk k with the loads of chunk k + 2 issued among them;
//   wino3x3_wgrad_ws_kernel  768 threads, one workgroup per CU, two LDS stages: four waves only read fragments and issue MFMAs, eight
//                            only load / transform / write the next chunk; one barrier per chunk.
// Epilogue: every wave folds its row of planes with A' (no exchange), the four rows meet in LDS one tap column at a time, A'^T, and the
// nine taps of the 64 x 32 block go to the split's slab in OIHW order.
// Where the time goes (profiles/r05_wino_wgrad_ablation.txt, 256 -> 256 @ 64 x 64 x 32, 0.83 ms = 0.52 of the MFMA peak): the MFMAs with
// their fragment reads and barriers alone 0.56 ms, everything else alone 0.44 ms; transform + LDS writes hide behind the MFMAs (0.57 ms
// together), the global loads do not -- with them 0.83 ms whatever their width (4 or 16 bytes), prefetch distance (1 or 2 chunks), place
// in the issue order, or the waves that issue them (the specialised kernel: 0.57 -> 0.88 ms, 0.83 with cache-hot addresses).
#include "ge_common.h"
#include "ge_wino_plan.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t wnw_rsrc_t;

#define WNW_OOB 0xFFFFFFFFu
constexpr int WNW_KT = 8;      // (WNW_MT, WNW_CT: ge_wino_plan.h)
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
  float* bslab;        // nullable: [splits][M] row sums of dy over the split's K range (the bias gradient), written by the c-tile-0 workgroups
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

  // Raw operands of one chunk as they come from memory: two 16-byte rows of dy (the thread's two tiles) and the 4 x 4 patch of x as
  // FOUR 16-byte loads (one per row; 4-byte aligned: the patch starts one pixel left of an even column) -- six load instructions per
  // thread and chunk instead of eighteen.  Left image border (strip 0, tile 0): the row is read from its first pixel and shifted by
  // one at use; right border (last strip, tile 7): the fourth pixel is masked at use (the load itself stays inside the buffer, or has
  // that dword range-checked to 0 at the buffer's end).  WNW_AHEAD register sets: the loads of chunk g + WNW_AHEAD are issued during
  // the MFMA phase of chunk g, one behind each of its first six MFMA pairs.
#ifndef WNW_DBG
#define WNW_DBG 0      // tuning builds only (WRONG results): 1 no loads inside the loop, 2 no transform / LDS writes, 4 no MFMAs, 8 no barriers
#endif
  struct Raw {
    f32x4 e0, e1, v[4];
    bool left, right;
  };
#ifndef WNW_AHEAD
#define WNW_AHEAD 2
#endif
  Raw raw[WNW_AHEAD];
  uint32_t pc_eoff = WNW_OOB;
  int pc_vbase = 0, pc_iy0 = 0;
  bool pc_live = false;
  auto load_piece = [&](int k, int g, Raw& w) {      // k = 0 .. 5
    if (k == 0) {
      // chunk g -> (image b, tile row ty, strip sxi); past the end of the split: everything out of range = zeros, no traffic
      pc_live = g < g_end;
      const int b = g / (p.th * p.sx), r1 = g - b * (p.th * p.sx);
      const int ty = r1 / p.sx, sxi = r1 - ty * p.sx;
      pc_eoff = pc_live ? (e_thread + (uint32_t)((b * p.M * p.H + 2 * ty) * p.W + 16 * sxi)) * 4u : WNW_OOB;
      pc_iy0 = 2 * ty - 1;
      w.left = sxi == 0 && vt == 0;
      w.right = sxi == p.sx - 1 && vt == 7;
      pc_vbase = v_thread + (b * p.C * p.H + pc_iy0) * p.W + 16 * sxi + (w.left ? 1 : 0);
      w.e0 = wnw_load4(yrs, pc_eoff);
    } else if (k == 1) {
      w.e1 = wnw_load4(yrs, pc_live ? pc_eoff + (uint32_t)p.W * 4u : WNW_OOB);
    } else {
      const int r = k - 2;
      const bool rok = pc_live && (unsigned)(pc_iy0 + r) < (unsigned)p.H;
      if ((WNW_DBG & 32) && r >= 2) {      // tuning build: half the bytes of the x patch
        w.v[r] = w.v[r - 2];
        return;
      }
      if (WNW_DBG & 64) {                  // tuning build: 8 bytes per row instead of 16 (a de-duplicated strip would load that)
        const f32x2 h = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, rok ? (uint32_t)(pc_vbase + r * p.W) * 4u : WNW_OOB, 0, 0));
        w.v[r].x = h.x; w.v[r].y = h.y; w.v[r].z = h.x; w.v[r].w = h.y;
        return;
      }
      w.v[r] = wnw_load4(xrs, rok ? (uint32_t)(pc_vbase + r * p.W) * 4u : WNW_OOB);
    }
  };
  // G' e G'^T of one tile e = [[a, b], [c, d]]: rows (a, b), (a + c, b + d), (a - c, b - d), (-c, -d), then the same on the columns
  float bsum = 0.f;      // bias gradient: this thread's share of sum(dy[m0 + em]) over the split (used by the c-tile-0 workgroups)
  auto stage = [&](const Raw& w) {
    bsum += ((w.e0.x + w.e0.y) + (w.e0.z + w.e0.w)) + ((w.e1.x + w.e1.y) + (w.e1.z + w.e1.w));
    // E: tiles 2 eq (pixels .x .y of both rows) and 2 eq + 1 (.z .w) -> [plane][eq][em][0 / 1]
    float* e = sE + eq * WNW_EPITCH + em * 2;
    float ea[16], eb[16];
    {
      const float a = w.e0.x, b = w.e0.y, c = w.e1.x, dd = w.e1.y;
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
      const float a = w.e0.z, b = w.e0.w, c = w.e1.z, dd = w.e1.w;
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
    float d[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d[r * 4 + 0] = w.left ? 0.f : w.v[r].x;
      d[r * 4 + 1] = w.left ? w.v[r].x : w.v[r].y;
      d[r * 4 + 2] = w.left ? w.v[r].y : w.v[r].z;
      d[r * 4 + 3] = w.right ? 0.f : (w.left ? w.v[r].z : w.v[r].w);
    }
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

  const float* ea = sE + (4 * wave) * WNW_EPLANE + hi * WNW_EPITCH + li * 2;
  const float* va = sV + (4 * wave) * WNW_VPLANE + hi * WNW_VPITCH + li * 2;
  // One chunk: barrier - transform + write - barrier - 32 MFMAs.  The issue order of the MFMA phase is placed by hand and pinned with
  // sched_barrier(0): behind each of the first eight MFMA pairs one load of chunk g + WNW_AHEAD (six of them) and a share of the
  // second tile pair's fragment reads.  (hipcc's own order put the loads behind the LAST MFMAs of the chunk.)
  auto chunk = [&](int g, Raw& w) {
    if (!(WNW_DBG & 8)) __syncthreads();      // every wave has read the fragments of the previous chunk
    if (!(WNW_DBG & 2) || g == g_begin) stage(w);
    if (!(WNW_DBG & 8)) __syncthreads();
    f32x2 fa[2][4][2], fb[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[0][q][0] = *(const f32x2*)(ea + q * WNW_EPLANE);
      fa[0][q][1] = *(const f32x2*)(ea + q * WNW_EPLANE + 64);
      fb[0][q] = *(const f32x2*)(va + q * WNW_VPLANE);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int P = 0; P < 2; ++P)      // tile pairs 2 P + hi
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (WNW_DBG & 4) {
            acc[q][0][0] += fa[P][q][0][s] * fb[P][q][s];
            acc[q][1][0] += fa[P][q][1][s] * fb[P][q][s];
          } else {
            acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][q][0][s], fb[P][q][s], acc[q][0], 0, 0, 0);
            acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[P][q][1][s], fb[P][q][s], acc[q][1], 0, 0, 0);
          }
          if (P == 0) {
            const int k = s * 4 + q;      // slot 0 .. 7
            if (k < 6 && !(WNW_DBG & 1)) load_piece(k, g + WNW_AHEAD, w);
            const int q1 = k >> 1;
            if (k & 1) {
              fb[1][q1] = *(const f32x2*)(va + q1 * WNW_VPLANE + 2 * WNW_VPITCH);
            } else {
              fa[1][q1][0] = *(const f32x2*)(ea + q1 * WNW_EPLANE + 2 * WNW_EPITCH);
              fa[1][q1][1] = *(const f32x2*)(ea + q1 * WNW_EPLANE + 2 * WNW_EPITCH + 64);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
  };

#pragma unroll
  for (int a = 0; a < WNW_AHEAD; ++a)
#pragma unroll
    for (int k = 0; k < 6; ++k) load_piece(k, g_begin + a, raw[a]);
  for (int g = g_begin; g < g_end; g += WNW_AHEAD) {
    chunk(g, raw[0]);
#if WNW_AHEAD == 2
    if (g + 1 < g_end) chunk(g + 1, raw[1]);
#endif
  }
  __syncthreads();
  if (p.bslab && tc == 0) {      // the four tile-pair lanes of a channel are a DPP quad
    bsum += dpp_f32<0xB1>(bsum, 0.f);
    bsum += dpp_f32<0x4E>(bsum, 0.f);
    if (eq == 0) p.bslab[(size_t)split * p.M + m0 + em] = bsum;
  }

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

// ---- the same computation with the two phases of a chunk on DIFFERENT waves ---------------------------------------------------------
// Round-5 ablation of the kernel above (profiles/r05_wino_wgrad_microbench.txt): everything but the MFMAs takes 0.44 ms on
// 256 -> 256 @ 64 x 64 x 32, the MFMAs alone 0.44-0.49 ms, the kernel 0.83 ms -- barrier-separated phases of one workgroup do not
// overlap, and the second workgroup of the CU covers for little of it (prefetch distance, load width and issue order all measured: no
// change).  Here one workgroup of EIGHT waves owns the CU: waves 0-3 (one per SIMD) only read fragments and issue MFMAs, waves 4-7 (one
// per SIMD) only load, transform and write the NEXT chunk into the other of two LDS stages (2 x 56 KB); one barrier per chunk.  The
// producer's VALU / LDS-write / load instructions issue beside the consumer's MFMAs on the same SIMD by construction.
#ifndef WNW_PW
#define WNW_PW 8
#endif
constexpr int WNW_WS_THREADS = 256 + 64 * WNW_PW;
__global__ __launch_bounds__(WNW_WS_THREADS, 1) void wino3x3_wgrad_ws_kernel(WinoWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // [2 stages][E: 16 planes | V: 16 planes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int gid = wnw_xcd_remap(blockIdx.x, gridDim.x);
  const int ntile = p.tiles_m * p.tiles_c;
  const int split = gid / ntile, trem = gid - split * ntile;
  const int tm = trem / p.tiles_c, tc = trem - tm * p.tiles_c;
  const int m0 = tm * WNW_MT, c0 = tc * WNW_CT;
  const int HW = p.H * p.W;
  const int g_begin = split * p.split_chunks;
  const int g_end = min(p.chunks, g_begin + p.split_chunks);
  const int n = g_end - g_begin;

  f32x16 acc[4][2];
  if (wave >= 4) {
    // ================= producer: thread pt of 256 in the E / V roles of the kernel above =================
    const wnw_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (uint32_t)((size_t)p.B * p.C * HW * 4u), 0x00020000);
    const wnw_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (uint32_t)((size_t)p.B * p.M * HW * 4u), 0x00020000);
    const int pt = (tid - 256) & 255;
    const bool do_e = WNW_PW == 4 || wave < 8, do_v = WNW_PW == 4 || wave >= 8;      // (wave-uniform)
    const int eq = pt & 3, em = pt >> 2;
    const int vt = pt & 7, vc = pt >> 3;
    const uint32_t e_thread = (uint32_t)(((m0 + em) * p.H) * p.W + 4 * eq);
    const int v_thread = ((c0 + vc) * p.H) * p.W + 2 * vt - 1;
    struct Raw {
      f32x4 e0, e1, v[4];
      bool left, right;
    };
    auto load = [&](int g, Raw& w) {
      if (WNW_DBG & 16) g = g_begin + (g & 1);      // tuning build: the same two chunks over and over (cache-hot loads)
      const bool live = g < g_end;
      const int b = g / (p.th * p.sx), r1 = g - b * (p.th * p.sx);
      const int ty = r1 / p.sx, sxi = r1 - ty * p.sx;
      if (do_e) {
        const uint32_t eoff = live ? (e_thread + (uint32_t)((b * p.M * p.H + 2 * ty) * p.W + 16 * sxi)) * 4u : WNW_OOB;
        w.e0 = wnw_load4(yrs, eoff);
        w.e1 = wnw_load4(yrs, live ? eoff + (uint32_t)p.W * 4u : WNW_OOB);
      }
      if (do_v) {
        const int iy0 = 2 * ty - 1;
        w.left = sxi == 0 && vt == 0;
        w.right = sxi == p.sx - 1 && vt == 7;
        const int vbase = v_thread + (b * p.C * p.H + iy0) * p.W + 16 * sxi + (w.left ? 1 : 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool rok = live && (unsigned)(iy0 + r) < (unsigned)p.H;
          w.v[r] = wnw_load4(xrs, rok ? (uint32_t)(vbase + r * p.W) * 4u : WNW_OOB);
        }
      }
    };
    float bsum = 0.f;
    auto stage = [&](const Raw& w, float* sE, float* sV) {
     if (do_e) {
      bsum += ((w.e0.x + w.e0.y) + (w.e0.z + w.e0.w)) + ((w.e1.x + w.e1.y) + (w.e1.z + w.e1.w));
      float* e = sE + eq * WNW_EPITCH + em * 2;
      float ea[16], eb[16];
      {
        const float a = w.e0.x, b = w.e0.y, c = w.e1.x, dd = w.e1.y;
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
        const float a = w.e0.z, b = w.e0.w, c = w.e1.z, dd = w.e1.w;
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
     }
     if (do_v) {
      float d[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d[r * 4 + 0] = w.left ? 0.f : w.v[r].x;
        d[r * 4 + 1] = w.left ? w.v[r].x : w.v[r].y;
        d[r * 4 + 2] = w.left ? w.v[r].y : w.v[r].z;
        d[r * 4 + 3] = w.right ? 0.f : (w.left ? w.v[r].z : w.v[r].w);
      }
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
     }
    };
    float* sE0 = lds;
    float* sV0 = lds + 16 * WNW_EPLANE;
    float* sE1 = lds + WNW_LDS_FLOATS;
    float* sV1 = lds + WNW_LDS_FLOATS + 16 * WNW_EPLANE;
    // chunk i (relative) lives in register set i & 1 and goes to LDS stage i & 1; its loads are issued two chunks ahead
    Raw r0, r1;
    load(g_begin, r0);
    load(g_begin + 1, r1);
    stage(r0, sE0, sV0);
    load(g_begin + 2, r0);
    __syncthreads();
    for (int i = 0; i < n; i += 2) {
      // consumers are on chunk i (stage 0): write chunk i + 1 into stage 1
      if (i + 1 < n && !(WNW_DBG & 2)) stage(r1, sE1, sV1);
      if (!(WNW_DBG & 1)) load(g_begin + i + 3, r1);
      __syncthreads();
      if (i + 1 < n) {
        // consumers are on chunk i + 1 (stage 1): write chunk i + 2 into stage 0
        if (i + 2 < n && !(WNW_DBG & 2)) stage(r0, sE0, sV0);
        if (!(WNW_DBG & 1)) load(g_begin + i + 4, r0);
        __syncthreads();
      }
    }
    if (p.bslab && tc == 0 && do_e) {
      bsum += dpp_f32<0xB1>(bsum, 0.f);
      bsum += dpp_f32<0x4E>(bsum, 0.f);
      if (eq == 0) p.bslab[(size_t)split * p.M + m0 + em] = bsum;
    }
  } else {
    // ================= consumer: wave w owns planes 4 w .. 4 w + 3 =================
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][mh][r] = 0.f;
    const int foff_e = (4 * wave) * WNW_EPLANE + hi * WNW_EPITCH + li * 2;
    const int foff_v = 16 * WNW_EPLANE + (4 * wave) * WNW_VPLANE + hi * WNW_VPITCH + li * 2;
    auto mma = [&](const float* base) {
      const float* ea = base + foff_e;
      const float* va = base + foff_v;
      f32x2 fa[4][2], fb[4];      // one set: plane q's fragments of the second tile pair replace the first pair's behind their last use
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[q][0] = *(const f32x2*)(ea + q * WNW_EPLANE);
        fa[q][1] = *(const f32x2*)(ea + q * WNW_EPLANE + 64);
        fb[q] = *(const f32x2*)(va + q * WNW_VPLANE);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int P = 0; P < 2; ++P)      // tile pairs 2 P + hi
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (WNW_DBG & 4) {
              acc[q][0][0] += fa[q][0][s] * fb[q][s];
              acc[q][1][0] += fa[q][1][s] * fb[q][s];
            } else {
#ifdef WNW_AGPR      // tuning build: accumulators in AccVGPRs
              asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[q][0]) : "v"(fa[q][0][s]), "v"(fb[q][s]));
              asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[q][1]) : "v"(fa[q][1][s]), "v"(fb[q][s]));
#else
              acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][0][s], fb[q][s], acc[q][0], 0, 0, 0);
              acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1][s], fb[q][s], acc[q][1], 0, 0, 0);
#endif
            }
            if (P == 0 && s == 1) {
              fa[q][0] = *(const f32x2*)(ea + q * WNW_EPLANE + 2 * WNW_EPITCH);
              fa[q][1] = *(const f32x2*)(ea + q * WNW_EPLANE + 2 * WNW_EPITCH + 64);
              fb[q] = *(const f32x2*)(va + q * WNW_VPLANE + 2 * WNW_VPITCH);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
    };
    __syncthreads();
    for (int i = 0; i < n; i += 2) {
      mma(lds);
      __syncthreads();
      if (i + 1 < n) {
        mma(lds + WNW_LDS_FLOATS);
        __syncthreads();
      }
    }
  }

  // ---- epilogue (as above; the consumers hold the accumulators, all 512 threads store)
  float* sX = lds;      // [4 rows][64 m][33]
  float* out = p.slab + ((size_t)split * p.M + m0) * (size_t)p.C * 9 + (size_t)c0 * 9;
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    if (b) __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s0 = acc[0][mh][r], s1 = acc[1][mh][r], s2 = acc[2][mh][r], s3 = acc[3][mh][r];
          const float v = b == 0 ? s0 + 0.5f * (s1 + s2) : b == 1 ? 0.5f * (s1 - s2) : 0.5f * (s1 + s2) + s3;
          const int m = 32 * mh + (r & 3) + 8 * (r >> 2) + 4 * hi;
          sX[(wave * WNW_MT + m) * WNW_XROW + li] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 2048; e += WNW_WS_THREADS) {
      const int m = e >> 5, c = e & 31;
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
// the same with the bias slabs behind the weights: db[m] (+)= bslab[0][m] + bslab[1][m] + ... for the grid's elements n .. n + M - 1
__global__ __launch_bounds__(256) void wnw_reduce_bias_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n, int splits,
                                                              int accumulate, const float* __restrict__ bslab, float* __restrict__ db, int M) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n + M; i += (long long)gridDim.x * 256) {
    const bool w = i < n;
    const float* src = w ? slab + i : bslab + (i - n);
    const size_t stride = w ? (size_t)n : (size_t)M;
    float* dst = w ? dw + i : db + (i - n);
    float s0 = accumulate ? *dst : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
      s0 += src[(size_t)k * stride];
      s1 += src[(size_t)(k + 1) * stride];
      s2 += src[(size_t)(k + 2) * stride];
      s3 += src[(size_t)(k + 3) * stride];
    }
    for (; k < splits; ++k) s0 += src[(size_t)k * stride];
    *dst = (s0 + s1) + (s2 + s3);
  }
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
// the routing decision in full (tests, tools): *splits as ge_wino3x3_wgrad_splits, *ws_kernel = 1 for the warp-specialised kernel,
// return value = ge_wino3x3_wgrad_supported
int ge_wino3x3_wgrad_plan(int B, int C, int M, int H, int W, int* splits, int* ws_kernel) {
  int chunks = 0;
  bool ws = false;
  const int s = wnw_plan(B, C, M, H, W, chunks, false, &ws);
  if (splits) *splits = s;
  if (ws_kernel) *ws_kernel = ws ? 1 : 0;
  return wnw_plan(B, C, M, H, W, chunks, true) > 0 ? 1 : 0;
}
// floats of workspace: the K-split weight slabs [splits][M][C][9], then the bias slabs [splits][M] (ge_wino3x3_wgrad_bias)
long long ge_wino3x3_wgrad_workspace(int B, int C, int M, int H, int W) {
  int chunks = 0;
  const int s = wnw_plan(B, C, M, H, W, chunks, false);
  return (long long)s * M * C * 9 + (long long)s * M;
}
// dw[M][C][3][3] (+)= weight gradient of y = conv3x3(x; stride 1, pad 1) from x [B][C][H][W] and dy [B][M][H][W]
// accumulate bit 0: add to dw; bit 1: leave the slabs in the workspace (stride M * C * 9) for ge_slab_reduce_batched
static int wino3x3_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int C, int M, int H, int W,
                              int accumulate, void* stream);
int ge_wino3x3_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int C, int M, int H, int W, int accumulate,
                     void* stream) {
  return wino3x3_wgrad_impl(x, dy, dw, nullptr, workspace, B, C, M, H, W, accumulate, stream);
}
// the same, plus the bias gradient db[M] (+)= sum over batch and positions of dy (the weight-gradient pass loads every dy element anyway:
// the c-tile-0 workgroups add them up, the slab reduce folds the splits) -- no separate ge_channel_sum pass over dy.  accumulate bit 0
// applies to dw and db alike; bit 1 (slabs left to the caller) is not offered here.
int ge_wino3x3_wgrad_bias(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int C, int M, int H, int W,
                          int accumulate, void* stream) {
  GE_REQUIRE(db, "wino3x3_wgrad_bias: db required");
  GE_REQUIRE(!(accumulate & 2), "wino3x3_wgrad_bias: the deferred-slab form has no bias part");
  return wino3x3_wgrad_impl(x, dy, dw, db, workspace, B, C, M, H, W, accumulate, stream);
}
static int wino3x3_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* workspace, int B, int C, int M, int H, int W,
                              int accumulate, void* stream) {
  GE_REQUIRE(x && dy && dw && workspace, "wino3x3_wgrad: null pointer");
  int chunks = 0;
  bool ws = false;
  const int splits0 = wnw_plan(B, C, M, H, W, chunks, false, &ws);
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
  p.bslab = db ? workspace + (size_t)p.splits * M * C * 9 : nullptr;
  const int grid = p.tiles_m * p.tiles_c * p.splits;
  if (ws) {
    const size_t smem = 2 * WNW_LDS_FLOATS * sizeof(float);
    static GeLdsAttr attr;
    const int rc = ge_set_max_lds(attr, (const void*)wino3x3_wgrad_ws_kernel, (int)smem, "wino3x3_wgrad_ws_kernel");
    if (rc != GE_OK) return rc;
    wino3x3_wgrad_ws_kernel<<<grid, WNW_WS_THREADS, smem, st>>>(p);
    ge_note_kernel("wino3x3_wgrad_ws_kernel");
  } else {
    const size_t smem = WNW_LDS_FLOATS * sizeof(float);
    static GeLdsAttr attr;
    const int rc = ge_set_max_lds(attr, (const void*)wino3x3_wgrad_kernel, (int)smem, "wino3x3_wgrad_kernel");
    if (rc != GE_OK) return rc;
    wino3x3_wgrad_kernel<<<grid, 256, smem, st>>>(p);
    ge_note_kernel("wino3x3_wgrad_kernel");
  }
  GE_CHECK_LAUNCH("wino3x3_wgrad");
  ge_record_split_event(st);
  if (accumulate & 2) return GE_OK;      // the caller reduces the slabs later (ge_slab_reduce_batched)
  const long long n = (long long)M * C * 9;
  if (db)
    wnw_reduce_bias_kernel<<<ge_stream_grid(n + M, 256), 256, 0, st>>>(workspace, dw, n, p.splits, accumulate & 1, p.bslab, db, M);
  else
    wnw_reduce_kernel<<<ge_stream_grid(n, 256), 256, 0, st>>>(workspace, dw, n, p.splits, accumulate & 1);
  GE_CHECK_LAUNCH("wino3x3_wgrad_reduce");
  return GE_OK;
}

}  // extern "C"
