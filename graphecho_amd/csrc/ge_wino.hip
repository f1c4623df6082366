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
// its (tile, channel) pairs, adds bias / addend and stores 2 x 2 outputs (optionally: the BatchNorm moments of the 128 outputs of
// each channel, one (count, mean, M2) triple per workgroup and channel in ge_bn_finalize's format).  Layers whose grid would
// leave most of the chip idle (16 x 16 / 8 x 8 maps, small per-GPU batches) are split over the input channels: split 0 writes
// the destination, the others slabs that wn_slab_reduce_kernel adds in split order (bit-reproducible).
#include "ge_common.h"
#include "ge_wino_plan.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int wn_u32x4 __attribute__((ext_vector_type(4)));

#define WN_OOB 0xFFFFFFFFu
// tuning builds only (tools/build_wino_variants.sh; results are WRONG with any bit set): 1 no MFMAs, 2 no patch loads, 4 no transform /
// V writes, 8 no filter DMA, 16 no epilogue.  The production library is compiled with WN_DBG = 0.
#ifndef WN_DBG
#define WN_DBG 0
#endif
#ifndef WN_DMA_SPREAD
#define WN_DMA_SPREAD 1
#endif
constexpr int WN_VSTAGE = 16 * WN_KC * WN_TILES;      // 4096 floats
constexpr int WN_USTAGE = 16 * WN_KC * WN_MC;         // 8192 floats
constexpr int WN_STAGE = WN_VSTAGE + WN_USTAGE;       // 12288 floats = 48 KB
constexpr int WN_MROW = 33;                           // padded tile row of the epilogue exchange
constexpr int WN_LDS_FLOATS = 2 * WN_VSTAGE + 3 * (WN_USTAGE / 2);   // 80 KB: 2 x V + 3 x half-chunk U; the epilogue exchange (66 KB) reuses it
static_assert(16 * 32 * WN_MROW <= WN_LDS_FLOATS, "exchange buffer");

// voff: the lane's 16 bytes inside a 1 KB piece (constant: lane * 16); soff: everything wave-uniform -- the piece's offset in the
// packed filters -- in the SGPR-offset field: issuing a piece costs no vector instruction (round 6, see wn_load)
__device__ __forceinline__ void wn_dma16(wn_u32x4 rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
#ifndef WN_SAFE_WAIT
#define WN_SAFE_WAIT 0      // tuning builds: 1 = every hand-counted wait becomes vmcnt(0)
#endif
template <int N0>
__device__ __forceinline__ void wn_vm_wait() {
  constexpr int N = WN_SAFE_WAIT ? 0 : N0;
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
// Cache policy of the input patch loads and the output stores (tuning builds: -DWN_NT_X=0/1, -DWN_NT_Y=0/1).  gfx950 aux bits:
// 1 = sc0, 2 = nt, 16 = sc1.  nt = streaming: the line is the first to be replaced in the XCD's L2.  Why (round 6, PMC per block
// order, profiles/r06_wino_traffic.txt): at 256 -> 256 the transformed filters are 4 MB = one XCD's whole L2, the input and the
// output stream through the same L2 (LRU), and every generation of workgroups fetched the filters again -- 412 MB of the 546 MB a
// launch read were filter re-reads.  Measured on 256 -> 256 @ 64 x 64 x 32 (reads per launch; 134 MB written either way; time):
//   stores plain, order 0: 545 MB, 0.695 ms (round 5)      nt stores, order 0: 479 MB, 0.69 ms
//   stores plain, order 2: 432 MB, 0.68-0.70 ms            nt stores, order 2: 386 MB, 0.68-0.69 ms   <- default
//   nt loads (either order): 420-610 MB, 0.74 ms
// i.e. traffic 2.48x -> 1.9x of the algorithmic 273 MB; the time does not follow the bytes (0.75 TB/s of fabric reads; the filter
// re-reads hit the 256 MB Infinity Cache), which is what the round-5 cache-hot experiment said.
#ifndef WN_NT_X
#define WN_NT_X 0      // measured: nt patch loads are SLOWER (0.74 vs 0.69 ms): the four channel tiles of a spatial block share these lines
#endif
#ifndef WN_NT_Y
#define WN_NT_Y 1
#endif
// voff: the lane's byte offset inside one chunk (WN_OOB for padding positions), soff: the chunk's byte offset, wave-uniform, in the
// instruction's SGPR-offset field.  The hardware range check covers voff + soff without wrapping (tools/microbench/soffset_check.hip:
// the all-ones voff stays out of range whatever soff is, offsets past num_records read 0) -- so stepping through the chunks costs NO
// vector instruction.  Rounds 4 - 5 did it with a saturating v_add_u32 per load: sixteen VALU instructions per chunk and wave, and on
// this part every VALU instruction is 4 cycles its SIMD's fp32 MFMAs do not run (profiles/r06_mfma_valu_microbench.txt).
__device__ __forceinline__ float wn_load(wn_rsrc_t rs, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, WN_NT_X ? 2 : 0));
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
  float* stats;          // [M][B * H * W / 128][3] (count, mean, M2) of y per workgroup and channel, or null
  float* ws;             // slabs of splits 1 .. splits - 1, each [B][M][H][W]
  int B, C, M, H, W;
  int blocks_x, blocks_y, tiles_m;
  int splits, split_chunks;      // K split: split s reduces chunks [s * split_chunks, min(C / 8, (s + 1) * split_chunks))
  uint32_t u_bytes;
  int order;
};

// TXT: tiles per block row (16: 2 x 16 tiles, 8: 4 x 8 tiles); STATS: BatchNorm moments of the result (splits == 1 only)
//
// LDS: V of a chunk of 8 channels (16 KB) twice + U of HALF a chunk (16 KB: MFMA steps 2 s, 2 s + 1 = channels 4 hi + 2 s + {0, 1})
// three times = 80 KB, the epilogue exchange (66 KB) inside it: two workgroups per CU -- with one, every serial section of a
// workgroup (first loads, the output transform and its stores) left the matrix pipe idle.
//
// Memory pipeline of the main loop (round 5; the wave counts its own vmcnt, the compiler does not see the LDS-DMA instructions):
//   first half of chunk c : DMA of U half-chunk 2 c + 2, the sixteen patch loads of chunk c + 2 (TWO chunks ahead: a patch is
//                           first touched in HBM -- the four channel tiles of a spatial block run side by side on one XCD and
//                           all wait for the same lines -- and one half-chunk, 0.4 us of MFMA work, does not cover that trip)
//   second half of chunk c: transform + V writes of chunk c + 1 (loaded during chunk c - 1), THEN the DMA of U half-chunk 2 c + 3.
// Round 4 issued that DMA at the start of the half: hipcc, blind to it, waited for "all my patch loads" with vmcnt(0) a few MFMAs
// later, i.e. for the DMA issued a moment ago as well -- one full trip to L2 of stall per chunk and wave (ISA: s_waitcnt vmcnt(1) /
// vmcnt(0) between the MFMAs of the second half).  Issued after the last use of the patch registers nothing younger than the patch
// loads is in flight when they are consumed.
template <int TXT, bool STATS>
__global__ __launch_bounds__(256, 2) void wino3x3_kernel(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sV = lds;                       // [2][16 planes][4 channel pairs][32 tiles][2]
  float* sU = lds + 2 * WN_VSTAGE;       // [3][16 planes][2 hi][64 m][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int nblk = (int)gridDim.x / p.splits;
  const int gid = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int split = gid / nblk, lid = gid - split * nblk;
  // order 0: the channel tiles of one spatial block are neighbours (they share the input patch in L2); 1: one channel tile's
  // spatial blocks are neighbours (they share its 16 * C * 64 transformed filters).  GE_WN_ORDER=1, measured: 0.750 - 0.766 vs
  // 0.762 - 0.774 ms on 256 -> 256 @ 64 x 64 x 32, nothing on the other layers -- neither operand's locality bounds the kernel
  // order 2 (round 6): the grid in tiles_m / 2 consecutive ranges, range r = channel tiles 2 r, 2 r + 1 of EVERY spatial block
  // (the two neighbours share the patch).  After the XCD remap a range is a set of whole XCDs: each XCD's L2 then holds the
  // transformed filters of TWO channel tiles (2 MB at 256 -> 256) instead of cycling through all of them (4 MB = the whole L2)
  // once per generation of workgroups; the price is the input read by tiles_m / 2 XCD groups instead of one.
  const int nsp = nblk / p.tiles_m;
  int tm, sp;
  if (p.order == 2) {
    const int per = 2 * nsp, r = lid / per, l2 = lid - r * per;
    tm = 2 * r + (l2 & 1);
    sp = l2 >> 1;
  } else {
    tm = p.order ? lid / nsp : lid % p.tiles_m;
    sp = p.order ? lid - tm * nsp : lid / p.tiles_m;
  }
  const int per_img = p.blocks_x * p.blocks_y;
  const int b = sp / per_img, srem = sp - b * per_img;
  const int by = srem / p.blocks_x, bx = srem - by * p.blocks_x;
  constexpr int TYT = WN_TILES / TXT;
  const int y0 = by * (2 * TYT), x0 = bx * (2 * TXT), m0 = tm * WN_MC;
  const int HW = p.H * p.W;
  const int nch_all = p.C / WN_KC;
  const int c_begin = split * p.split_chunks;
  const int nch = min(nch_all - c_begin, p.split_chunks);      // chunks of this split (>= 1 by construction)

  const wn_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + ((size_t)b * p.C + (size_t)c_begin * WN_KC) * HW), 0, (uint32_t)(nch * WN_KC) * (uint32_t)HW * 4u,
      0x00020000);
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
  const uint32_t u_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(tm * nch_all + c_begin) * (WN_USTAGE * 4u)));
  const uint32_t u_lane = (uint32_t)lane * 16u;
  const uint32_t wu = (uint32_t)__builtin_amdgcn_readfirstlane(wave);

  float d[2][16];      // patch registers of two chunks in flight (chunk c in set c & 1)
  // chunks past the split's range: the offset runs out of the buffer's range = zeros, no memory traffic
  auto patch_add = [&](int ch) {
    const unsigned long long a64 = (unsigned long long)ch * chunk_step;
    return a64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)a64;
  };
  auto load_patch = [&](int ch, float* dd) {
    const uint32_t add = (uint32_t)__builtin_amdgcn_readfirstlane((int)patch_add(ch));
#pragma unroll
    for (int e = 0; e < 16; ++e) dd[e] = wn_load(xrs, poff[e], add);
  };
  auto load_patch_row = [&](uint32_t add, float* dd, int i) {      // row i of the 4 x 4 patch
#pragma unroll
    for (int e = 4 * i; e < 4 * i + 4; ++e) dd[e] = wn_load(xrs, poff[e], add);
  };
  // half-chunk h = 2 * chunk + s of the transformed filters -> U buffer ub (16 KB = 16 pieces of 1 KB, four per wave)
  auto issue_u = [&](int h, int ub) {
    const uint32_t gbase = u_block + (uint32_t)h * (WN_USTAGE * 2u);
    const uint32_t lbase = lds_u + (uint32_t)ub * (WN_USTAGE * 2u);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t piece = (uint32_t)e * 4u + wu;
      wn_dma16(urs, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + piece * 1024u)), u_lane,
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(gbase + piece * 1024u)));
    }
  };
  // B^T d B of the thread's (tile, channel) -> V[plane][channel pair][tile][channel & 1]: a wave writes, and a half-wave reads
  // (8-byte A fragments), 64 consecutive words -- no bank conflicts (round 4's [k / 4][tile][k % 4] put tiles t and t + 16 on
  // one bank for both)
  auto stage_v = [&](int st, const float* dd) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0 * 4 + j] = dd[0 * 4 + j] - dd[2 * 4 + j];
      t[1 * 4 + j] = dd[1 * 4 + j] + dd[2 * 4 + j];
      t[2 * 4 + j] = dd[2 * 4 + j] - dd[1 * 4 + j];
      t[3 * 4 + j] = dd[1 * 4 + j] - dd[3 * 4 + j];
    }
    float* v = sV + st * WN_VSTAGE + (lc >> 1) * 64 + lt * 2 + (lc & 1);
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

  issue_u(0, 0);
  issue_u(1, 1);
  load_patch(0, d[0]);
  load_patch(1, d[1]);
  stage_v(0, d[0]);
  wn_vm_wait<16>();      // both U half-chunks and chunk 0's patch have landed; chunk 1's sixteen loads may fly on
  __syncthreads();

  int ub = 0;      // U buffer of half-chunk h = h % 3
  // One half-chunk: 12 fragment reads + 16 MFMAs.  The blocks are straight-line on purpose and carry an explicit issue order
  // (sched_group_barrier): a wave issues in order, so VALU / memory instructions placed behind the sixteen MFMAs would only start
  // when the last one has been issued -- measured (round 4): the patch loads + transform + V writes then ADD their 0.25 ms to the
  // 0.58 ms of the MFMA loop instead of hiding under it.
  // chunk ch with the patch registers of chunk ch + 1 in `dn` (loaded one chunk ago) and those of chunk ch + 2 going to `df`
  // (LOAD = false: chunk ch + 2 does not exist -- no loads are issued, and the vmcnt arithmetic below must not count on them:
  // hipcc drops loads whose registers are dead, a `wait until at most 20 are in flight` then waits for nothing)
  //
  // The four DMA pieces of a half are spread over its MFMA sequence, one per four MFMAs (WN_DMA_SPREAD): a 1 KiB LDS-DMA piece
  // costs its wave 60 - 185 issue cycles (MI355X_MICROARCH.md), and four of them back to back in front of the fragment reads, as
  // round 4 placed them, were ~0.5 us per chunk in which the wave fed the matrix pipe nothing.
  auto mfma_group = [&](const f32x2* fa, const f32x2* fb0, const f32x2* fb1, int kk, int q0) {
#pragma unroll
    for (int q = q0; q < q0 + 2; ++q) {
      if (!(WN_DBG & 1)) {
        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][kk], fb0[q][kk], acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][kk], fb1[q][kk], acc[q][1], 0, 0, 0);
      }
    }
  };
  auto issue_u_piece = [&](int h, int ubuf, int e) {
    if (WN_DBG & 8) return;
    const uint32_t gbase = u_block + (uint32_t)h * (WN_USTAGE * 2u);
    const uint32_t lbase = lds_u + (uint32_t)ubuf * (WN_USTAGE * 2u);
    const uint32_t piece = (uint32_t)e * 4u + wu;
    wn_dma16(urs, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lbase + piece * 1024u)), u_lane,
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(gbase + piece * 1024u)));
  };
  auto load_frags = [&](int ch, int sh, int ubuf, f32x2* fa, f32x2* fb0, f32x2* fb1) {
    const float* sv = sV + (ch & 1) * WN_VSTAGE + (4 * wave) * 256 + (2 * hi + sh) * 64 + li * 2;
    const float* su = sU + ubuf * (WN_USTAGE / 2) + (4 * wave) * 256 + hi * 128 + li * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      fa[q] = *(const f32x2*)(sv + q * 256);
      fb0[q] = *(const f32x2*)(su + q * 256);
      fb1[q] = *(const f32x2*)(su + q * 256 + 64);
    }
    if (WN_DBG & 1) asm volatile("" ::"v"(fa[0]), "v"(fb0[0]), "v"(fb1[3]));
  };
  auto chunk = [&](auto load_tag, int ch, float* df, const float* dn) {
    constexpr bool LOAD = decltype(load_tag)::value;
    f32x2 fa[4], fb0[4], fb1[4];
    // ---- first half: DMA of half-chunk 2 ch + 2, the sixteen patch loads of chunk ch + 2 between the MFMAs
    const int ub_n = ub == 0 ? 2 : ub - 1;
    const uint32_t padd = (uint32_t)__builtin_amdgcn_readfirstlane((int)patch_add(ch + 2));
    load_frags(ch, 0, ub, fa, fb0, fb1);
#if WN_DMA_SPREAD
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      issue_u_piece(2 * ch + 2, ub_n, g);
      if (LOAD && !(WN_DBG & 2)) load_patch_row(padd, df, g);
      mfma_group(fa, fb0, fb1, g >> 1, (g & 1) * 2);
      if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);      // the fragment reads first
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
        if (LOAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one patch load (its chunk offset rides in the SGPR field)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int g = 0; g < 4; ++g) issue_u_piece(2 * ch + 2, ub_n, g);
    if (LOAD && !(WN_DBG & 2)) load_patch(ch + 2, df);
#pragma unroll
    for (int g = 0; g < 4; ++g) mfma_group(fa, fb0, fb1, g >> 1, (g & 1) * 2);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (LOAD) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#endif
    // half-chunk 2 ch + 1 has landed (everything older too: the patch of chunk ch + 1); in flight at most: the four DMA pieces
    // and (LOAD) the sixteen loads just issued
    if (LOAD) wn_vm_wait<20>(); else wn_vm_wait<4>();
    __syncthreads();
    ub = ub == 2 ? 0 : ub + 1;
    // ---- second half: transform + V writes of chunk ch + 1 between the first eight MFMAs, the DMA of half-chunk 2 ch + 3 between
    // the last eight -- AFTER the last use of the patch registers: hipcc, blind to the DMA, waits for "all my patch loads" with a
    // vmcnt that would include pieces issued before that point (round 4: vmcnt(0) a few MFMAs after issuing them = one trip to L2
    // of stall per chunk)
    const int ub_f = ub == 0 ? 2 : ub - 1;      // the buffer the FIRST half read (free since its barrier)
    load_frags(ch, 1, ub, fa, fb0, fb1);
    mfma_group(fa, fb0, fb1, 0, 0);
    mfma_group(fa, fb0, fb1, 0, 2);
    if (!(WN_DBG & 4)) stage_v((ch + 1) & 1, dn);
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // one (paired) V write
    }
    __builtin_amdgcn_sched_barrier(0);
#if WN_DMA_SPREAD
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      issue_u_piece(2 * ch + 3, ub_f, g);
      if (!(WN_DBG & 1)) {
        const int q = g;
        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb0[q][1], acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q][1], fb1[q][1], acc[q][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    mfma_group(fa, fb0, fb1, 1, 0);
    mfma_group(fa, fb0, fb1, 1, 2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) issue_u_piece(2 * ch + 3, ub_f, g);
#endif
    // half-chunk 2 ch + 2 has landed.  Spread order of the first half: piece 0, 4 loads, piece 1, 4 loads, ..., piece 3, 4 loads --
    // behind its LAST piece came four loads and the four pieces just issued: at most 8 younger operations may stay in flight (a
    // "20" here let pieces 1 .. 3 of the half-chunk the next MFMAs read fly on: wrong results under load, found by bench_wino.py)
    if (LOAD) wn_vm_wait<WN_DMA_SPREAD ? 8 : 20>(); else wn_vm_wait<4>();
    __syncthreads();
    ub = ub == 2 ? 0 : ub + 1;
  };
  const std::integral_constant<bool, true> kLoad;
  const std::integral_constant<bool, false> kNoLoad;
  // Chunks in pairs (the two patch register sets swap roles); the second chunk of the last pair may fetch a patch that does not
  // exist -- zeros from beyond the buffer's range, no traffic, and the vmcnt arithmetic stays the same.  What is left is the last
  // chunk alone, or one chunk without loads and the last.  (Peeling more cases made hipcc reconcile the register roles of the
  // variants through scratch memory -- spills whose loads share the vmcnt this kernel counts by hand.)
  int ch = 0;
  for (; ch + 2 < nch; ch += 2) {
    chunk(kLoad, ch, d[0], d[1]);
    chunk(kLoad, ch + 1, d[1], d[0]);
  }
  if (ch + 1 < nch) {
    chunk(kNoLoad, ch, d[0], d[1]);
    ++ch;
  }
  // ---- last chunk: nothing left to fetch
  {
    f32x2 fa[4], fb0[4], fb1[4];
    load_frags(nch - 1, 0, ub, fa, fb0, fb1);
#pragma unroll
    for (int g = 0; g < 4; ++g) mfma_group(fa, fb0, fb1, g >> 1, (g & 1) * 2);
    __builtin_amdgcn_sched_barrier(0);
    wn_vm_wait<0>();
    __syncthreads();
    ub = ub == 2 ? 0 : ub + 1;
    load_frags(nch - 1, 1, ub, fa, fb0, fb1);
#pragma unroll
    for (int g = 0; g < 4; ++g) mfma_group(fa, fb0, fb1, g >> 1, (g & 1) * 2);
  }
  __syncthreads();

  // ---- epilogue: the 16 planes of a (tile, channel) meet in LDS, 32 channels at a time
  if (WN_DBG & 16) return;
  float* sM = lds;
  const int et = lane & 31;
  const int etx = et % TXT, ety = et / TXT;
  const int oy = y0 + 2 * ety, ox = x0 + 2 * etx;
  const bool first = split == 0;
  float* __restrict__ dst = first ? p.y : p.ws + (size_t)(split - 1) * ((size_t)p.B * p.M * HW);
  const float* __restrict__ bias = first ? p.bias : nullptr;
  const float* __restrict__ addend = first ? p.addend : nullptr;
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
      if (bias) {
        const float bv = bias[mch];
        o0.x += bv;
        o0.y += bv;
        o1.x += bv;
        o1.y += bv;
      }
      const size_t o = ((size_t)b * p.M + mch) * HW + (size_t)oy * p.W + ox;
      if (addend) {
        const f32x2 a0 = *(const f32x2*)(addend + o), a1 = *(const f32x2*)(addend + o + p.W);
        o0 += a0;
        o1 += a1;
      }
      if (WN_NT_Y) {
        __builtin_nontemporal_store(o0, (f32x2*)(dst + o));
        __builtin_nontemporal_store(o1, (f32x2*)(dst + o + p.W));
      } else {
        *(f32x2*)(dst + o) = o0;
        *(f32x2*)(dst + o + p.W) = o1;
      }
      if (STATS) {
        // moments of the channel's 128 outputs of this workgroup: 4 per lane, then equal-count Chan merges across the 32 lanes that
        // hold the channel (DPP inside the 16-lane rows, the two rows of a half-wave through readlane)
        float mean = 0.25f * ((o0.x + o0.y) + (o1.x + o1.y));
        const float e0 = o0.x - mean, e1 = o0.y - mean, e2 = o1.x - mean, e3 = o1.y - mean;
        float m2 = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        float half_n = 2.f;      // count / 2 of each side of the merge
#define WN_MERGE(CTRL)                                        \
  {                                                           \
    const float mb = dpp_f32<CTRL>(mean, 0.f), qb = dpp_f32<CTRL>(m2, 0.f); \
    const float dl = mean - mb;                               \
    m2 = (m2 + qb) + dl * dl * half_n;                        \
    mean = 0.5f * (mean + mb);                                \
    half_n += half_n;                                         \
  }
        WN_MERGE(0xB1) WN_MERGE(0x4E) WN_MERGE(0x141) WN_MERGE(0x140)
#undef WN_MERGE
        const float ma = readlane_f32(mean, 0), qa = readlane_f32(m2, 0), mb_ = readlane_f32(mean, 16), qb_ = readlane_f32(m2, 16);
        const float mc = readlane_f32(mean, 32), qc = readlane_f32(m2, 32), md = readlane_f32(mean, 48), qd = readlane_f32(m2, 48);
        const float mlo = 0.5f * (ma + mb_), qlo = (qa + qb_) + (ma - mb_) * (ma - mb_) * 32.f;
        const float mhi = 0.5f * (mc + md), qhi = (qc + qd) + (mc - md) * (mc - md) * 32.f;
        if (et == 0) {
          float* sp_ = p.stats + ((size_t)mch * (size_t)(p.B * per_img) + (size_t)sp) * 3;
          sp_[0] = 128.f;
          sp_[1] = hi ? mhi : mlo;
          sp_[2] = hi ? qhi : qlo;
        }
      }
    }
  }
}

// y += slab[0] + slab[1] + ... (slab order: bit-reproducible); n4 = elements / 4
__global__ __launch_bounds__(256) void wn_slab_reduce_kernel(const f32x4* __restrict__ ws, f32x4* __restrict__ y, long long n4,
                                                             int slabs) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    f32x4 s = y[i];
    for (int k = 0; k < slabs; ++k) s += ws[(size_t)k * n4 + i];
    y[i] = s;
  }
}

// u[m / 64][c / 8][s][plane][hi][m % 64][e] = (G g G^T)[plane] with c % 8 = 4 hi + 2 s + e, g = w[m][c] (transposed = 0) or the data gradient's
// filter w[c][m] rotated by 180 degrees (transposed = 1: m runs over the ORIGINAL input channels, c over the original output channels)
__device__ __forceinline__ void wn_pack_one(const float* __restrict__ w, float* __restrict__ u, int M, int C, int transposed, int idx) {
  // idx enumerates (m tile, chunk, (hi, sh), m % 64, e): the 64 lanes of a wave write 64 CONSECUTIVE floats of every plane (the
  // (m, c)-major enumeration of round 4 wrote pairs 512 bytes apart: 0.36 ms per step for the FPN's layers, 0.9 TB/s)
  const int nchk = C / WN_KC;
  const int e_ = idx & 1, ml_ = (idx >> 1) & 63, hs_ = (idx >> 7) & 3, blk_ = idx >> 9;
  const int mt_ = blk_ / nchk, chk_ = blk_ - mt_ * nchk;
  const int m = mt_ * WN_MC + ml_, c = chk_ * WN_KC + 4 * (hs_ >> 1) + 2 * (hs_ & 1) + e_;
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
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int M, int C,
                                                        int transposed) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < M * C) wn_pack_one(w, u, M, C, transposed, idx);
}
// every Winograd operand of a model in one launch (after the optimizer step): table rows (int64) = (offset of the OIHW weight in
// `flat` in floats, destination pointer, M, C, transposed); grid.y = row
__global__ __launch_bounds__(256) void wino_pack_batched_kernel(const float* __restrict__ flat, const long long* __restrict__ table) {
  const long long* row = table + (size_t)blockIdx.y * 5;
  const float* w = flat + row[0];
  float* u = (float*)(uintptr_t)row[1];
  const int M = (int)row[2], C = (int)row[3], tr = (int)row[4];
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < M * C; idx += gridDim.x * 256) wn_pack_one(w, u, M, C, tr, idx);
}

extern "C" {

// 1 when ge_wino3x3_fwd covers the layer (C = reduction channels, M = output channels of the pass) and the Winograd route is the
// faster one there (ge_wino3x3_splits > 0)
int ge_wino3x3_supported(int B, int C, int M, int H, int W) { return wn_plan_splits(B, C, M, H, W) > 0 ? 1 : 0; }
// covered geometry, whatever the grid size (tests / microbenches)
int ge_wino3x3_covered(int B, int C, int M, int H, int W) { return wn_covered(B, C, M, H, W) ? 1 : 0; }
// number of K splits ge_wino3x3_fwd uses for the layer (0: not routed) and the workspace it then needs, in floats
int ge_wino3x3_splits(int B, int C, int M, int H, int W) { return wn_plan_splits(B, C, M, H, W); }
long long ge_wino3x3_workspace(int B, int C, int M, int H, int W) {
  const int s = wn_plan_splits(B, C, M, H, W);
  return s > 1 ? (long long)(s - 1) * B * M * H * W : 0;
}
long long ge_wino3x3_weight_floats(int C, int M) { return 16ll * C * M; }
// [M][parts][3] BatchNorm moments per launch: parts = B * H * W / 128 (one per workgroup and channel)
int ge_wino3x3_stat_parts(int B, int H, int W) { return B * (H * W / 128); }
// transformed filters of a pass with M output and C reduction channels from w (OIHW, 3 x 3): transposed = 0: w is [M][C][3][3]
// (forward); transposed = 1: w is [C][M][3][3] (data gradient: M = the layer's input channels, C = its output channels)
int ge_wino3x3_pack_weight(const float* w, float* u, int M, int C, int transposed, void* stream) {
  GE_REQUIRE(w && u && M % WN_MC == 0 && C % WN_KC == 0, "wino3x3_pack_weight: bad arguments");
  wino_pack_kernel<<<ge_cdiv((long long)M * C, 256), 256, 0, (hipStream_t)stream>>>(w, u, M, C, transposed);
  GE_CHECK_LAUNCH("wino_pack");
  return GE_OK;
}
// table: device int64 [n][5] rows (weight offset in `flat` in floats, destination device pointer, M, C, transposed)
int ge_wino3x3_pack_weights_batched(const float* flat, const long long* table, int n, void* stream) {
  GE_REQUIRE(flat && table && n > 0 && n <= 65535, "wino3x3_pack_weights_batched: bad arguments");
  wino_pack_batched_kernel<<<dim3(64, n), 256, 0, (hipStream_t)stream>>>(flat, table);
  GE_CHECK_LAUNCH("wino_pack_batched");
  return GE_OK;
}
// y = conv3x3(x; stride 1, pad 1) (+ bias) (+ addend): x [B][C][H][W], y / addend [B][M][H][W], u from ge_wino3x3_pack_weight;
// stats (nullable): [M][ge_wino3x3_stat_parts][3] moments of y; workspace: ge_wino3x3_workspace floats (null when that is 0)
int ge_wino3x3_fwd(const float* x, const float* u, const float* bias, const float* addend, float* y, float* stats, float* workspace,
                   int B, int C, int M, int H, int W, void* stream) {
  GE_REQUIRE(x && u && y, "wino3x3_fwd: null pointer");
  const int txt = wn_txt(H, W);
  GE_REQUIRE(wn_covered(B, C, M, H, W), "wino3x3_fwd: unsupported geometry B=%d C=%d M=%d %dx%d", B, C, M, H, W);
  int splits = wn_plan_splits(B, C, M, H, W);
  if (splits < 1) splits = 1;      // a covered layer the plan would not route: the caller insists (tests, microbenches)
  GE_REQUIRE(splits == 1 || workspace, "wino3x3_fwd: this layer runs split over its input channels and needs its workspace");
  GE_REQUIRE(splits == 1 || !stats, "wino3x3_fwd: no moments epilogue on the split path");
  hipStream_t st = (hipStream_t)stream;
  WinoParams p;
  p.x = x;
  p.u = u;
  p.bias = bias;
  p.addend = addend;
  p.y = y;
  p.stats = stats;
  p.ws = workspace;
  p.B = B;
  p.C = C;
  p.M = M;
  p.H = H;
  p.W = W;
  p.blocks_x = W / (2 * txt);
  p.blocks_y = H / (2 * (WN_TILES / txt));
  p.tiles_m = M / WN_MC;
  const int nch = C / WN_KC;
  p.split_chunks = (nch + splits - 1) / splits;
  splits = (nch + p.split_chunks - 1) / p.split_chunks;      // no empty split
  p.splits = splits;
  p.u_bytes = (uint32_t)(64ull * C * M);
  static const int order_env = wino_env("GE_WN_ORDER", 2);
  p.order = (order_env == 2 && (p.tiles_m & 1)) ? 0 : order_env;
  const int grid = B * p.blocks_x * p.blocks_y * p.tiles_m * splits;
  const size_t smem = WN_LDS_FLOATS * sizeof(float);
  static GeLdsAttr attr[4];
#define WN_LAUNCH(T, S, SLOT)                                                                                   \
  {                                                                                                             \
    const int rc = ge_set_max_lds(attr[SLOT], (const void*)wino3x3_kernel<T, S>, (int)smem, "wino3x3_kernel"); \
    if (rc != GE_OK) return rc;                                                                                 \
    wino3x3_kernel<T, S><<<grid, 256, smem, st>>>(p);                                                           \
  }
  if (txt == 16) {
    if (stats) WN_LAUNCH(16, true, 0) else WN_LAUNCH(16, false, 1)
  } else {
    if (stats) WN_LAUNCH(8, true, 2) else WN_LAUNCH(8, false, 3)
  }
#undef WN_LAUNCH
  ge_note_kernel("wino3x3_kernel<%d, %s>", txt, stats ? "true" : "false");      // as rocprofv3 prints it
  GE_CHECK_LAUNCH("wino3x3");
  if (splits > 1) {
    const long long n4 = (long long)B * M * H * W / 4;
    wn_slab_reduce_kernel<<<ge_stream_grid(n4, 256), 256, 0, st>>>((const f32x4*)workspace, (f32x4*)y, n4, splits - 1);
    GE_CHECK_LAUNCH("wino3x3_slab_reduce");
  }
  return GE_OK;
}

}  // extern "C"
