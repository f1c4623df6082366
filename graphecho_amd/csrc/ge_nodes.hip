// Front end of GModule's graph construction (reference models/graph_matching.py):
//   * ge_mask_boxes      -- masks_to_boxes (:702-740): tight (x1, y1, x2, y2) of the non-zero pixels of each mask
//   * ge_fcos_labels     -- PrototypeComputation.compute_targets_for_locations (:874-959): class of every pyramid
//                           location (smallest containing box whose largest side distance lies in the level's range)
//   * ge_gather_nodes_*  -- the rows PrototypeComputation samples from the NCHW pyramid levels (:961-1013) and the
//                           scatter of their gradients
// The labels are one byte per location (num_classes < 256) so that the whole label set of a step -- 5440 locations per
// 256x256 frame -- goes to the host in one small copy; the host plans the sampling (counts, ranks, class histograms)
// and sends back one index table.  That replaces ~250 small ATen launches and the second device->host read per call.
#include "ge_common.h"

// One workgroup of 1024 threads per mask.  Column / row extents of the non-zero pixels; an all-zero mask yields
// (0, 0, W, H).  16-byte loads, four of them in flight per thread (round 4: the 256-thread scalar loop took 83-100 us for
// 32 masks of 256 x 256 -- a dependent chain of 256 loads per thread -- in front of GModule's blocking label read).
__global__ __launch_bounds__(1024) void mask_boxes_kernel(const float* __restrict__ masks, float* __restrict__ boxes,
                                                          int H, int W) {
  const float* m = masks + (size_t)blockIdx.x * H * W;
  int x1 = W, x2 = -1, y1 = H, y2 = -1;
  const int n = H * W;
  auto see = [&](float v, int e) {
    if (v != 0.f) {
      const int y = e / W, x = e - y * W;
      x1 = min(x1, x);
      x2 = max(x2, x);
      y1 = min(y1, y);
      y2 = max(y2, y);
    }
  };
  if ((W & 3) == 0 && (((size_t)m) & 15) == 0) {
    const float4* m4 = (const float4*)m;
    const int n4 = n >> 2;
    int e = threadIdx.x;
    for (; e + 3 * 1024 < n4; e += 4 * 1024) {
      const float4 a = m4[e], b = m4[e + 1024], c = m4[e + 2048], d = m4[e + 3072];
      const float4 q[4] = {a, b, c, d};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int base = 4 * (e + u * 1024);
        if (q[u].x != 0.f || q[u].y != 0.f || q[u].z != 0.f || q[u].w != 0.f) {
          see(q[u].x, base);
          see(q[u].y, base + 1);
          see(q[u].z, base + 2);
          see(q[u].w, base + 3);
        }
      }
    }
    for (; e < n4; e += 1024) {
      const float4 a = m4[e];
      see(a.x, 4 * e);
      see(a.y, 4 * e + 1);
      see(a.z, 4 * e + 2);
      see(a.w, 4 * e + 3);
    }
  } else {
    for (int e = threadIdx.x; e < n; e += 1024) see(m[e], e);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    x1 = min(x1, __shfl_xor(x1, o));
    y1 = min(y1, __shfl_xor(y1, o));
    x2 = max(x2, __shfl_xor(x2, o));
    y2 = max(y2, __shfl_xor(y2, o));
  }
  __shared__ int red[16][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[wave][0] = x1;
    red[wave][1] = y1;
    red[wave][2] = x2;
    red[wave][3] = y2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      x1 = min(x1, red[w][0]);
      y1 = min(y1, red[w][1]);
      x2 = max(x2, red[w][2]);
      y2 = max(y2, red[w][3]);
    }
    float* o = boxes + (size_t)blockIdx.x * 4;
    if (x2 < 0) {
      o[0] = 0.f;
      o[1] = 0.f;
      o[2] = (float)W;
      o[3] = (float)H;
    } else {
      o[0] = (float)x1;
      o[1] = (float)y1;
      o[2] = (float)x2;
      o[3] = (float)y2;
    }
  }
}

#define GE_MAX_LEVELS 5

struct LevelTable {
  int off[GE_MAX_LEVELS + 1];  // first location of each level in the concatenated list
  int w[GE_MAX_LEVELS];
  int stride[GE_MAX_LEVELS];
  float lo[GE_MAX_LEVELS], hi[GE_MAX_LEVELS];
  int levels;
};

// One thread per (frame, location).  The arithmetic is the reference's, in fp32: side distances l/t/r/b of the location
// to every class box, "inside" = min > 0, "cared" = lo <= max <= hi, area = (y2 - y1) * (x2 - x1); the valid class of
// smallest area wins (first one on ties), label 0 when none is valid.  Note that the label IS the class index, so class
// 0 and "no class" coincide, as in the reference.
__global__ __launch_bounds__(256) void fcos_labels_kernel(const float* __restrict__ boxes,
                                                          unsigned char* __restrict__ labels, int B, int nc,
                                                          LevelTable tab) {
  const int L = tab.off[tab.levels];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * L) return;
  const int b = e / L, loc = e - b * L;
  int lvl = 0;
  while (lvl + 1 < tab.levels && loc >= tab.off[lvl + 1]) ++lvl;
  const int i = loc - tab.off[lvl];
  const int iy = i / tab.w[lvl], ix = i - iy * tab.w[lvl];
  const float half = (float)(tab.stride[lvl] / 2);
  const float x = (float)(ix * tab.stride[lvl]) + half, y = (float)(iy * tab.stride[lvl]) + half;
  const float lo = tab.lo[lvl], hi = tab.hi[lvl];
  float best = 100000000.f;
  int arg = 0;
  const float* bx = boxes + (size_t)b * nc * 4;
  for (int c = 0; c < nc; ++c) {
    const float x1 = bx[c * 4 + 0], y1 = bx[c * 4 + 1], x2 = bx[c * 4 + 2], y2 = bx[c * 4 + 3];
    const float l = x - x1, t = y - y1, r = x2 - x, bt = y2 - y;
    const float mn = fminf(fminf(l, t), fminf(r, bt)), mx = fmaxf(fmaxf(l, t), fmaxf(r, bt));
    const bool ok = mn > 0.f && mx >= lo && mx <= hi;
    const float area = ok ? (y2 - y1) * (x2 - x1) : 100000000.f;
    if (area < best) {
      best = area;
      arg = c;
    }
  }
  labels[e] = (unsigned char)(best == 100000000.f ? 0 : arg);
}

struct LevelPtrs {
  const float* f[GE_MAX_LEVELS];
  float* d[GE_MAX_LEVELS];
  int hw[GE_MAX_LEVELS];
};

// out[n][c] = feat_level[b][c][p] with (b, p) = divmod(index[n], hw).  One workgroup per row, threads over channels.
__global__ __launch_bounds__(256) void gather_nodes_fwd_kernel(LevelPtrs lp, const long long* __restrict__ level,
                                                               const long long* __restrict__ index, float* __restrict__ out,
                                                               int C) {
  const int n = blockIdx.x, lvl = (int)level[n], idx = (int)index[n];
  const int hw = lp.hw[lvl];
  const int b = idx / hw, p = idx - b * hw;
  const float* src = lp.f[lvl] + (size_t)b * C * hw + p;
  for (int c = threadIdx.x; c < C; c += 256) out[(size_t)n * C + c] = src[(size_t)c * hw];
}

template <bool ATOMIC>
__global__ __launch_bounds__(256) void gather_nodes_bwd_kernel(LevelPtrs lp, const long long* __restrict__ level,
                                                               const long long* __restrict__ index,
                                                               const float* __restrict__ dout, int C) {
  const int n = blockIdx.x, lvl = (int)level[n], idx = (int)index[n];
  float* base = lp.d[lvl];
  if (base == nullptr) return;
  const int hw = lp.hw[lvl];
  const int b = idx / hw, p = idx - b * hw;
  float* dst = base + (size_t)b * C * hw + p;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float g = dout[(size_t)n * C + c];
    if (ATOMIC)
      atomicAdd(dst + (size_t)c * hw, g);
    else
      dst[(size_t)c * hw] = g;
  }
}


// Momentum update of the per-class seed bank (GModule.update_seed, models/graph_matching.py:532-567) in one launch per bank:
//   mean_c = mean of the kept rows of class c;  m = cosine_similarity(mean_c, bank_c) (each vector divided by its norm clamped at 1e-8,
//   as ATen does);  bank_c <- bank_c * m + mean_c * (1 - m)  for the classes present (has[c] != 0).
// One workgroup per class; cls[r] = class of node row r, -1 for rows the clustering dropped; thread = feature columns d, d + 256, ...
// The reference spells this as ~13 element-wise / reduce ops and three small host-to-device copies per bank.
__global__ __launch_bounds__(256) void seed_bank_update_kernel(float* __restrict__ bank, const float* __restrict__ nodes,
                                                               const int* __restrict__ cls, const int* __restrict__ has, int N, int D) {
  __shared__ float red[16];
  const int c = blockIdx.x;
  if (!has[c]) return;
  float cnt = 0.f;
  for (int r = 0; r < N; ++r) cnt += cls[r] == c ? 1.f : 0.f;      // (wave-uniform walk: N is a few hundred)
  float dot_mm = 0.f, dot_bb = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int r = 0; r < N; ++r)
      if (cls[r] == c) s += nodes[(size_t)r * D + d];
    const float mean = s / cnt;      // empty cluster -> NaN, as the reference
    const float b = bank[(size_t)c * D + d];
    dot_mm += mean * mean;
    dot_bb += b * b;
  }
  const float nm = fmaxf(sqrtf(block_sum(dot_mm, red)), 1e-8f);
  const float nb = fmaxf(sqrtf(block_sum(dot_bb, red)), 1e-8f);
  float dot = 0.f;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int r = 0; r < N; ++r)
      if (cls[r] == c) s += nodes[(size_t)r * D + d];
    dot += (s / cnt / nm) * (bank[(size_t)c * D + d] / nb);
  }
  const float m = block_sum(dot, red);
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int r = 0; r < N; ++r)
      if (cls[r] == c) s += nodes[(size_t)r * D + d];
    const float mean = s / cnt;
    const float b = bank[(size_t)c * D + d];
    bank[(size_t)c * D + d] = b * m + mean * (1.f - m);
  }
}

extern "C" {

int ge_mask_boxes(const float* masks, float* boxes, int n, int h, int w, void* stream) {
  GE_REQUIRE(n >= 0 && h > 0 && w > 0, "mask_boxes: bad shape n=%d h=%d w=%d", n, h, w);
  if (n == 0) return GE_OK;
  hipLaunchKernelGGL(mask_boxes_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, masks, boxes, h, w);
  GE_CHECK_LAUNCH("mask_boxes");
  return GE_OK;
}

// hws: host array [levels][3] = (h, w, stride) of each pyramid level; ranges: host array [levels][2] = (lo, hi) of the
// level's regression range.  labels: [batch][sum h*w] bytes.
int ge_fcos_labels(const float* boxes, unsigned char* labels, int batch, int num_class, int levels, const int* hws,
                   const float* ranges, void* stream) {
  GE_REQUIRE(levels >= 1 && levels <= GE_MAX_LEVELS, "fcos_labels: %d levels (1..%d supported)", levels, GE_MAX_LEVELS);
  GE_REQUIRE(num_class >= 1 && num_class < 256, "fcos_labels: %d classes do not fit the byte labels", num_class);
  LevelTable tab;
  tab.levels = levels;
  tab.off[0] = 0;
  for (int l = 0; l < levels; ++l) {
    GE_REQUIRE(hws[3 * l] > 0 && hws[3 * l + 1] > 0 && hws[3 * l + 2] > 0, "fcos_labels: bad level %d", l);
    tab.off[l + 1] = tab.off[l] + hws[3 * l] * hws[3 * l + 1];
    tab.w[l] = hws[3 * l + 1];
    tab.stride[l] = hws[3 * l + 2];
    tab.lo[l] = ranges[2 * l];
    tab.hi[l] = ranges[2 * l + 1];
  }
  const long long total = (long long)batch * tab.off[levels];
  GE_REQUIRE(total < (1ll << 31), "fcos_labels: too many locations");
  if (total == 0) return GE_OK;
  hipLaunchKernelGGL(fcos_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes,
                     labels, batch, num_class, tab);
  GE_CHECK_LAUNCH("fcos_labels");
  return GE_OK;
}

// Rows of up to five NCHW pyramid levels (f0..f4, hw0..hw4 = H*W of each; unused levels: null / 0).  level[n] selects the
// tensor, index[n] = b * hw + y * w + x the location.  out: [n][channels].
int ge_gather_nodes_fwd(const float* f0, const float* f1, const float* f2, const float* f3, const float* f4, int hw0,
                        int hw1, int hw2, int hw3, int hw4, int channels, const long long* level, const long long* index, float* out,
                        int n, void* stream) {
  GE_REQUIRE(n >= 0 && channels > 0, "gather_nodes_fwd: bad shape");
  if (n == 0) return GE_OK;
  LevelPtrs lp;
  const float* f[GE_MAX_LEVELS] = {f0, f1, f2, f3, f4};
  const int hw[GE_MAX_LEVELS] = {hw0, hw1, hw2, hw3, hw4};
  for (int l = 0; l < GE_MAX_LEVELS; ++l) {
    lp.f[l] = f[l];
    lp.d[l] = nullptr;
    lp.hw[l] = hw[l] > 0 ? hw[l] : 1;
  }
  hipLaunchKernelGGL(gather_nodes_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, lp, level, index, out,
                     channels);
  GE_CHECK_LAUNCH("gather_nodes_fwd");
  return GE_OK;
}

// Scatter of dout [n][channels] into the (pre-zeroed) gradients d0..d4 of the levels; a null pointer skips the level.
// atomic != 0: several rows may name the same location (the background ranks can repeat), accumulate with atomics.
int ge_gather_nodes_bwd(const float* dout, const long long* level, const long long* index, float* d0, float* d1, float* d2, float* d3,
                        float* d4, int hw0, int hw1, int hw2, int hw3, int hw4, int channels, int n, int atomic,
                        void* stream) {
  GE_REQUIRE(n >= 0 && channels > 0, "gather_nodes_bwd: bad shape");
  if (n == 0) return GE_OK;
  LevelPtrs lp;
  float* d[GE_MAX_LEVELS] = {d0, d1, d2, d3, d4};
  const int hw[GE_MAX_LEVELS] = {hw0, hw1, hw2, hw3, hw4};
  for (int l = 0; l < GE_MAX_LEVELS; ++l) {
    lp.f[l] = nullptr;
    lp.d[l] = d[l];
    lp.hw[l] = hw[l] > 0 ? hw[l] : 1;
  }
  if (atomic)
    hipLaunchKernelGGL(gather_nodes_bwd_kernel<true>, dim3(n), dim3(256), 0, (hipStream_t)stream, lp, level, index, dout,
                       channels);
  else
    hipLaunchKernelGGL(gather_nodes_bwd_kernel<false>, dim3(n), dim3(256), 0, (hipStream_t)stream, lp, level, index,
                       dout, channels);
  GE_CHECK_LAUNCH("gather_nodes_bwd");
  return GE_OK;
}

// bank [nc][D] updated in place from nodes [N][D]; cls [N] int32 (class of a kept row, -1: dropped), has [nc] int32
int ge_seed_bank_update(float* bank, const float* nodes, const int* cls, const int* has, int nc, int N, int D, void* stream) {
  GE_REQUIRE(bank && nodes && cls && has && nc > 0 && N > 0 && D > 0, "seed_bank_update: bad arguments");
  hipLaunchKernelGGL(seed_bank_update_kernel, dim3(nc), dim3(256), 0, (hipStream_t)stream, bank, nodes, cls, has, N, D);
  GE_CHECK_LAUNCH("seed_bank_update");
  return GE_OK;
}

}  // extern "C"
