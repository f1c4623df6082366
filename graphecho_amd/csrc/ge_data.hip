// Input formatting in front of FPN.forward and the validation metric behind it:
//   * frames: nearest resize -> crop -> / 255.0 (IEEE division, as the reference) -> fold clips (C,H,W,T) into the batch       (datasets/cardiac_uda.py:248-286,
//     datasets/camus.py:121-159: AddChannel, Resized(mode='nearest'), Rand/CenterSpatialCrop; `/ 255.0` at :155 / :103;
//     train_camus_echo.py:247-251: permute(0,4,1,2,3).reshape(-1,c,h,w))
//   * label maps: same geometry + one-hot over a class-value list                         (cardiac_uda.py:128-151,
//     camus.py:98-101: np.where(mask == v, 1, 0) stacked)
//   * overlap counts TP/FP/FN/TN of (logit > 0) against a binary mask, per class          (train_camus_echo.py:402-417)
// All HBM-bound byte/index work: one thread per output element, coalesced stores, exact integer arithmetic.
#include "ge_common.h"

// torch's legacy 'nearest' (what MONAI's Resized(mode='nearest') evaluates): src = min(floor(dst * in/out), in - 1)
__device__ __forceinline__ int nearest_src(int d, float scale, int in) {
  const int s = (int)floorf((float)d * scale);
  return s < in - 1 ? s : in - 1;
}

struct PrepGeom {
  int N, C, Hs, Ws, Ts;   // source: [N][C][Hs][Ws][Ts] (Ts = 1 for single frames)
  int S, To;              // resize target S x S (x To along time)
  int crop;               // output crop x crop
  int oy, ox;             // crop origin when offsets == null
  float sy, sx, st;       // Hs/S, Ws/S, Ts/To
};

// dst [N*To][C][crop][crop] fp32 = src[n][c][ny(y+oy)][nx(x+ox)][nt(t)] / divisor
template <typename SrcT>
__global__ __launch_bounds__(256) void frames_prepare_kernel(const SrcT* __restrict__ src, float* __restrict__ dst,
                                                             const int* __restrict__ offsets, PrepGeom g, float divisor,
                                                             FastDiv fd_x, FastDiv fd_y, FastDiv fd_c, FastDiv fd_t) {
  const long long total = (long long)g.N * g.To * g.C * g.crop * g.crop;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    uint32_t r = (uint32_t)i, x, y, c, t, n;   // total < 2^31 is enforced by the host
    fd_divmod(r, fd_x, r, x);
    fd_divmod(r, fd_y, r, y);
    fd_divmod(r, fd_c, r, c);
    fd_divmod(r, fd_t, n, t);
    const int oy = offsets ? offsets[2 * n] : g.oy, ox = offsets ? offsets[2 * n + 1] : g.ox;
    const int ys = nearest_src((int)y + oy, g.sy, g.Hs), xs = nearest_src((int)x + ox, g.sx, g.Ws);
    const int ts = nearest_src((int)t, g.st, g.Ts);
    const size_t so = ((((size_t)n * g.C + c) * g.Hs + ys) * g.Ws + xs) * g.Ts + ts;
    dst[i] = (float)src[so] / divisor;
  }
}

// dst [N*To][NC][crop][crop] fp32 = (label[n][ny][nx][nt] == values[k]); labels: [N][Hs][Ws][Ts] uint8
__global__ __launch_bounds__(256) void labels_onehot_kernel(const unsigned char* __restrict__ lab,
                                                            float* __restrict__ dst, const int* __restrict__ offsets,
                                                            const int* __restrict__ values, PrepGeom g, FastDiv fd_x,
                                                            FastDiv fd_y, FastDiv fd_c, FastDiv fd_t) {
  const long long total = (long long)g.N * g.To * g.C * g.crop * g.crop;   // g.C = number of classes here
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    uint32_t r = (uint32_t)i, x, y, c, t, n;
    fd_divmod(r, fd_x, r, x);
    fd_divmod(r, fd_y, r, y);
    fd_divmod(r, fd_c, r, c);
    fd_divmod(r, fd_t, n, t);
    const int oy = offsets ? offsets[2 * n] : g.oy, ox = offsets ? offsets[2 * n + 1] : g.ox;
    const int ys = nearest_src((int)y + oy, g.sy, g.Hs), xs = nearest_src((int)x + ox, g.sx, g.Ws);
    const int ts = nearest_src((int)t, g.st, g.Ts);
    const int v = lab[(((size_t)n * g.Hs + ys) * g.Ws + xs) * g.Ts + ts];
    dst[i] = v == values[c] ? 1.f : 0.f;
  }
}

// counts [C][4] (TP, FP, FN, TN) as int64, accumulated (+=) with one atomic per workgroup and class plane.
// prediction = logit > 0  (== sigmoid(logit) > 0.5); target = mask != 0.  logits, masks: [B][C][HW].
__global__ __launch_bounds__(256) void overlap_counts_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ masks,
                                                             unsigned long long* __restrict__ counts, int C, int HW) {
  __shared__ unsigned int red[4][4];
  const int c = blockIdx.y, b = blockIdx.z;
  const float* lp = logits + ((size_t)b * C + c) * HW;
  const float* mp = masks + ((size_t)b * C + c) * HW;
  unsigned int tp = 0, fp = 0, fn = 0, tn = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    const bool o = lp[i] > 0.f, t = mp[i] != 0.f;
    tp += o && t;
    fp += o && !t;
    fn += !o && t;
    tn += !o && !t;
  }
  unsigned int v[4] = {tp, fp, fn, tn};
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    unsigned int s = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[w][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long s = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                 red[3][threadIdx.x];
    if (s) atomicAdd(&counts[c * 4 + threadIdx.x], s);
  }
}

static int prep_geom(PrepGeom& g, int N, int C, int Hs, int Ws, int Ts, int S, int To, int crop, int oy, int ox,
                     const char* what) {
  GE_REQUIRE(N > 0 && C > 0 && Hs > 0 && Ws > 0 && Ts > 0 && S > 0 && To > 0 && crop > 0 && crop <= S, what);
  GE_REQUIRE(oy >= 0 && ox >= 0 && oy + crop <= S && ox + crop <= S, what);
  GE_REQUIRE((long long)N * To * C * crop * crop < (1ll << 31), what);
  g.N = N;
  g.C = C;
  g.Hs = Hs;
  g.Ws = Ws;
  g.Ts = Ts;
  g.S = S;
  g.To = To;
  g.crop = crop;
  g.oy = oy;
  g.ox = ox;
  g.sy = (float)Hs / (float)S;
  g.sx = (float)Ws / (float)S;
  g.st = (float)Ts / (float)To;
  return GE_OK;
}

extern "C" {

// src: [N][C][Hs][Ws][Ts] uint8 (src_is_float = 0) or fp32 (1); dst: [N*To][C][crop][crop] fp32.
// offsets: device int [N][2] crop origins (y, x) in the resized S x S frame, or null -> (oy, ox) for every sample.
int ge_frames_prepare(const void* src, int src_is_float, float* dst, const int* offsets, int N, int C, int Hs, int Ws,
                      int Ts, int S, int To, int crop, int oy, int ox, float divisor, void* stream) {
  GE_REQUIRE(src && dst && divisor != 0.f, "frames_prepare: null pointer or zero divisor");
  PrepGeom g;
  const int rc = prep_geom(g, N, C, Hs, Ws, Ts, S, To, crop, offsets ? 0 : oy, offsets ? 0 : ox,
                           "frames_prepare: bad geometry");
  if (rc) return rc;
  const long long total = (long long)N * To * C * crop * crop;
  const dim3 grid(ge_stream_grid(total, 256));
  const FastDiv fx = make_fastdiv(crop), fy = make_fastdiv(crop), fc = make_fastdiv(C), ft = make_fastdiv(To);
  if (src_is_float)
    hipLaunchKernelGGL(frames_prepare_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst,
                       offsets, g, divisor, fx, fy, fc, ft);
  else
    hipLaunchKernelGGL(frames_prepare_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, dst, offsets, g, divisor, fx, fy, fc, ft);
  GE_CHECK_LAUNCH("frames_prepare");
  return GE_OK;
}

// labels: [N][Hs][Ws][Ts] uint8 class ids; values: device int [NC]; dst: [N*To][NC][crop][crop] fp32 one-hot.
int ge_labels_onehot(const unsigned char* labels, float* dst, const int* offsets, const int* values, int N, int NC,
                     int Hs, int Ws, int Ts, int S, int To, int crop, int oy, int ox, void* stream) {
  GE_REQUIRE(labels && dst && values, "labels_onehot: null pointer");
  PrepGeom g;
  const int rc = prep_geom(g, N, NC, Hs, Ws, Ts, S, To, crop, offsets ? 0 : oy, offsets ? 0 : ox,
                           "labels_onehot: bad geometry");
  if (rc) return rc;
  const long long total = (long long)N * To * NC * crop * crop;
  hipLaunchKernelGGL(labels_onehot_kernel, dim3(ge_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, labels,
                     dst, offsets, values, g, make_fastdiv(crop), make_fastdiv(crop), make_fastdiv(NC),
                     make_fastdiv(To));
  GE_CHECK_LAUNCH("labels_onehot");
  return GE_OK;
}

// counts: device int64 [C][4] = (TP, FP, FN, TN), incremented (zero it before the first batch of a validation pass).
int ge_overlap_counts(const float* logits, const float* masks, long long* counts, int B, int C, int HW, void* stream) {
  GE_REQUIRE(logits && masks && counts && B > 0 && C > 0 && HW > 0, "overlap_counts: bad arguments");
  GE_REQUIRE(C <= 65535 && B <= 65535, "overlap_counts: too many planes");
  int gx = ge_cdiv(HW, 256 * 8);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(overlap_counts_kernel, dim3(gx, C, B), dim3(256), 0, (hipStream_t)stream, logits, masks,
                     (unsigned long long*)counts, C, HW);
  GE_CHECK_LAUNCH("overlap_counts");
  return GE_OK;
}

}  // extern "C"
