// Routing of the fp32 3x3 / stride 1 / pad 1 convolutions onto the Winograd kernels (ge_wino.hip: forward and data gradient,
// ge_wino_wgrad.hip: weight gradient) -- ONE place for every threshold (round 6; rounds 4 - 5 had nine of them as GE_WN_* / GE_WNW_*
// environment switches spread over two files; seven constants are left).  Each constant below has a test that pins the decision on either side of it
// (tests/test_abi.py::test_winograd_routing_thresholds, CPU: these are host functions), and the measurement it comes from.
//
//   forward / data gradient (workgroup = 32 tiles of 2 x 2 outputs x 64 output channels; blocks = B * H * W / 128 * M / 64):
//     blocks >= WN_SPLIT_TARGET (256)     one pass, no K split: half a wave of the chip's 512 workgroup slots or more
//     WN_MIN_BLOCKS (32) <= blocks < 256  split over the input channels: ceil(256 / blocks) splits (<= 8 by construction), each >=
//                                         WN_SPLIT_MIN_CHUNKS (4) chunks of 8 channels; slabs + ordered reduce
//     blocks < 32                         the direct kernels keep the layer (their split-K path)
//     measured: tools/bench_wino.py at 32 / 16 / 8 frames, profiles/r05_wino_microbench.txt -- every covered layer down to 32
//     workgroups is x1.3 - 1.9 the direct route; below, the reduce of > 8 slabs costs more than the multiplications saved.
//     (Rounds 4 - 5 also had a "full grid" threshold of 384 and a cap of 8 splits: both implied by the two constants above.)
//
//   weight gradient (workgroup = 64 output x 32 input channels x 16 planes over a K range of 8-tile chunks; tiles = M / 64 * C / 32,
//   chunks = B * H / 2 * W / 16):
//     splits = ceil(WNW_TARGET (512) / tiles), each >= WNW_MIN_CHUNKS (16) chunks        (two workgroups per CU)
//     a split of <= WNW_WS_CHUNKS (48) chunks  -> the warp-specialised kernel (768 threads, one workgroup per CU) with
//                                                 ceil(256 / tiles) splits: 3 - 16 % faster on those layers, 4 - 6 % slower on long K
//     routed when splits * tiles >= WNW_MIN_GRID (192), else the direct kernel (only the warp-specialised plan can fall under it:
//     a plan with > 48 chunks per split always has >= 512 workgroups)
//     measured: tools/bench_wino_wgrad.py at 8 / 16 / 32 frames, profiles/r05_wino_wgrad_microbench.txt, r05_wino_wgrad_ablation.txt.
//
// Still switchable (microbenches and tests force ONE decision, never a threshold): GE_WN_SPLITS, GE_WNW_SPLITS (number of K splits),
// GE_WNW_WS (0 / 1: which weight-gradient kernel), GE_WN_ORDER (block order, ge_wino.hip).
#pragma once
#include <stdlib.h>

constexpr int WN_KC = 8, WN_TILES = 32, WN_MC = 64;      // forward / data gradient: channels per chunk, tiles and channels per workgroup
constexpr int WN_SPLIT_TARGET = 256, WN_MIN_BLOCKS = 32, WN_SPLIT_MIN_CHUNKS = 4;
constexpr int WNW_MT = 64, WNW_CT = 32;                  // weight gradient: output / input channels per workgroup
constexpr int WNW_TARGET = 512, WNW_MIN_CHUNKS = 16, WNW_WS_CHUNKS = 48, WNW_MIN_GRID = 192;

static inline int wino_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static inline int wn_txt(int H, int W) {      // tiles per block row: 2 x 16 or 4 x 8 tiles per workgroup, 0: map not covered
  if (W % 32 == 0 && H % 4 == 0) return 16;
  if (W % 16 == 0 && H % 8 == 0) return 8;
  return 0;
}
static inline bool wn_covered(int B, int C, int M, int H, int W) {
  if (B <= 0 || C <= 0 || M <= 0 || C % WN_KC || M % WN_MC || !wn_txt(H, W)) return false;
  return 4ull * C * H * W < 0xFFFF0000ull && 64ull * C * M < 0xFFFF0000ull && (long long)B * M * H * W < (1ll << 40);
}
// K splits of a forward / data-gradient pass with C reduction and M output channels: 1 = one pass, > 1 = split, 0 = not routed
static inline int wn_plan_splits(int B, int C, int M, int H, int W) {
  if (!wn_covered(B, C, M, H, W)) return 0;
  static const int forced = wino_env("GE_WN_SPLITS", 0);
  const long long blocks = (long long)B * (H * W / 128) * (M / WN_MC);
  const int nch = C / WN_KC;
  if (forced > 0) return forced <= nch ? forced : nch;
  if (blocks >= WN_SPLIT_TARGET) return 1;
  if (blocks < WN_MIN_BLOCKS) return 0;
  int s = (int)((WN_SPLIT_TARGET + blocks - 1) / blocks);      // 2 .. 8
  const int smax = nch / WN_SPLIT_MIN_CHUNKS;
  if (s > smax) s = smax;
  return s < 1 ? 1 : s;
}

static inline bool wnw_covered(int B, int C, int M, int H, int W) {
  if (B <= 0 || C <= 0 || M <= 0 || C % WNW_CT || M % WNW_MT || W % 16 || H % 2) return false;
  return (unsigned long long)B * C * H * W * 4ull < 0xFFFF0000ull && (unsigned long long)B * M * H * W * 4ull < 0xFFFF0000ull;
}
// K splits of the weight gradient (x [B][C][H][W], dy [B][M][H][W]) and the kernel (*use_ws: the warp-specialised one).
// routing = true: 0 when the direct kernel keeps the layer; false: what a caller that insists gets (>= 1 on covered layers).
static inline int wnw_plan(int B, int C, int M, int H, int W, int& chunks, bool routing, bool* use_ws = nullptr) {
  if (!wnw_covered(B, C, M, H, W)) return 0;
  static const int ws_mode = wino_env("GE_WNW_WS", -1), forced = wino_env("GE_WNW_SPLITS", 0);
  chunks = B * (H / 2) * (W / 16);
  const int tiles = (M / WNW_MT) * (C / WNW_CT);
  auto plan = [&](int tgt) {
    int s = forced > 0 ? forced : (tgt + tiles - 1) / tiles;
    if (s > chunks / WNW_MIN_CHUNKS) s = chunks / WNW_MIN_CHUNKS;
    if (s > chunks) s = chunks;
    return s;
  };
  int s = plan(WNW_TARGET);
  const bool ws = ws_mode >= 0 ? ws_mode != 0 : (s < 1 || (chunks + s - 1) / s <= WNW_WS_CHUNKS);
  if (ws) s = plan(WNW_TARGET / 2);
  if (use_ws) *use_ws = ws;
  if (routing && forced <= 0 && (s < 1 || (long long)s * tiles < WNW_MIN_GRID)) return 0;
  return s < 1 ? 1 : s;
}
