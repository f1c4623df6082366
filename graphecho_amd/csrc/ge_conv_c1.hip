// 3x3 / stride 1 / pad 1 convolution with ONE output channel: the `cls_logits` layer that ends every Discriminator tower
// (reference models/fpnseg.py:332-352, nn.Conv2d(256, 1, 3, padding=1)).  As an implicit GEMM it has M = 1: a 64-row MFMA
// tile does 1/64 useful work and the layer ran at 1.7 TFLOP/s (0.66 ms forward + 0.71 ms weight gradient per step at
// 64 frames).  It is a memory-bound reduction over the input channels -- each kernel here reads the activation once:
//   forward        y[b,y,x]  = bias + sum_c sum_t w[c][t] * x[b,c,y+dy_t,x+dx_t]        (HBM bound: |x| bytes)
//   weight grad    dw[c][t]  = sum_{b,y,x} dy[b,y,x] * x[b,c,y+dy_t,x+dx_t]             (HBM bound: |x| bytes)
// The data gradient (K = 9) stays on the GEMM path, where it is write-bound already.
#include "ge_common.h"

// Forward.  A lane owns a 2 x 4 block of outputs (two rows, four consecutive columns; W % 4 == 0): per channel it loads
// the 4 x 6 patch under it as four 16-byte loads plus the eight edge values -- 1.5 loads per output where one load per
// tap would be 9.  Workgroup = 64 such blocks x 8 waves; wave w reduces over the w-th eighth of the input channels (a
// serial walk over all channels is bound by load latency), the eight partial sums meet in LDS in a fixed order.  The
// weights are read through the scalar unit where the compiler can prove them wave-uniform, broadcast loads otherwise.
__global__ __launch_bounds__(512) void conv3x3_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int C, int H, int W, int units_per_image) {
  __shared__ float part[8][8][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W4 = W >> 2;
  const int u = blockIdx.x * 64 + lane;
  const bool live = u < units_per_image;
  const int py = live ? u / W4 : 0, xq = live ? u - py * W4 : 0;
  const int oy = 2 * py, x0 = 4 * xq;
  const size_t plane = (size_t)H * W;
  const float* xb = x + (size_t)b * C * plane;
  int roff[4];
  bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int iy = oy - 1 + r;
    rok[r] = iy >= 0 && iy < H;
    roff[r] = (rok[r] ? iy : 0) * W + x0;
  }
  const bool lok = x0 > 0, rgt = x0 + 4 < W;
  const int lo = lok ? -1 : 0, ro = rgt ? 4 : 3;       // clamped addresses of the two edge columns
  const int per = (C + 7) / 8;
  const int cb = wave * per, ce = min(C, cb + per);
  float acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 2
  for (int c = cb; c < ce; ++c) {
    const float* xc = xb + (size_t)c * plane;
    const float* wc = w + c * 9;
    float v[4][6];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 m = *(const float4*)(xc + roff[r]);
      const float l = xc[roff[r] + lo], g = xc[roff[r] + ro];
      v[r][0] = (rok[r] && lok) ? l : 0.f;
      v[r][1] = rok[r] ? m.x : 0.f;
      v[r][2] = rok[r] ? m.y : 0.f;
      v[r][3] = rok[r] ? m.z : 0.f;
      v[r][4] = rok[r] ? m.w : 0.f;
      v[r][5] = (rok[r] && rgt) ? g : 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float wv = wc[ky * 3 + kx];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv, v[ky + i][kx + j], acc[i][j]);
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][i * 4 + j][lane] = acc[i][j];
  __syncthreads();
  // wave q finishes output q of every block (q = row * 4 + column), summing the eight channel groups in order
  {
    float sum = bias ? bias[0] : 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) sum += part[g][wave][lane];
    const int yy = oy + (wave >> 2), xx = x0 + (wave & 3);
    if (live && yy < H) y[(size_t)b * plane + (size_t)yy * W + xx] = sum;
  }
}

// Weight gradient.  Workgroup = (frame b, group of CPB channels): the frame's dy plane is staged once in LDS with a
// zero halo, then every thread walks its share of each channel plane and keeps nine running sums
// acc[t] += x[iy][ix] * dy[iy - dy_t][ix - dx_t]; a block reduction leaves partial[b][c][9].  ge_conv3x3_c1_wgrad adds
// the partials over the frames in a second small launch (fixed order: bit-reproducible).
#define C1_CPB 8
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ partial, int C, int H, int W) {
  extern __shared__ float sdy[];                 // (H + 2) x (W + 2), zero border
  __shared__ float red[4][9];
  const int b = blockIdx.y, c0 = blockIdx.x * C1_CPB;
  const int Wp = W + 2, plane = H * W;
  for (int e = threadIdx.x; e < (H + 2) * Wp; e += 256) {
    const int r = e / Wp, q = e - r * Wp;
    const bool in = r >= 1 && r <= H && q >= 1 && q <= W;
    sdy[e] = in ? dy[(size_t)b * plane + (size_t)(r - 1) * W + (q - 1)] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int cc = 0; cc < C1_CPB && c0 + cc < C; ++cc) {
    const float* xc = x + ((size_t)b * C + c0 + cc) * plane;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    // four consecutive elements of a row per lane and step (W % 4 == 0): one 16-byte load of x, and the 3 x 6 window of
    // dy they meet read once from LDS (18 reads for 36 multiply-adds)
#pragma unroll 2
    for (int e = 4 * threadIdx.x; e < plane; e += 1024) {
      const int iy = e / W, ix = e - iy * W;
      const float4 xv = *(const float4*)(xc + e);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // x[iy][ix + j] meets tap (ky, kx) at padded dy index (iy - ky + 2, ix + j - kx + 2): columns ix .. ix + 5
        const float* d = sdy + (iy + 2 - ky) * Wp + ix;
        float dv[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) dv[q] = d[q];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = fmaf(xs[j], dv[j + 2 - kx], acc[ky * 3 + kx]);
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = acc[t];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9)
      partial[((size_t)b * C + c0 + cc) * 9 + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

// dw[i] (+)= sum_b partial[b][i], i < n
__global__ __launch_bounds__(256) void c1_partial_sum_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                             int n, int B, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = accumulate ? dw[i] : 0.f, s1 = 0.f;
  int b = 0;
  for (; b + 2 <= B; b += 2) {
    s0 += partial[(size_t)b * n + i];
    s1 += partial[(size_t)(b + 1) * n + i];
  }
  if (b < B) s0 += partial[(size_t)b * n + i];
  dw[i] = s0 + s1;
}

static int c1_min_plane() {
  constexpr int v = 1024;
  return v;
}

// ---- host side (called from ge_conv2d_fwd / ge_conv2d_wgrad in ge_mfma.hip) ---------------------------------------
bool ge_conv3x3_c1_applies(int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad,
                           int groups) {
  static const bool on = !(getenv("GE_CONV_C1") && atoi(getenv("GE_CONV_C1")) == 0);
  return on && Cout == 1 && groups == 1 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && Hi == Ho && Wi == Wo &&
         Cin >= 8 && Wi % 4 == 0;
}

// LDS of the weight-gradient kernel: the padded dy plane must fit the 64 KB a workgroup gets without an attribute
bool ge_conv3x3_c1_wgrad_applies(int H, int W) {
  return (size_t)(H + 2) * (W + 2) * sizeof(float) <= 48 * 1024 && W % 4 == 0 && H * W >= c1_min_plane();
}

// Forward: maps of >= 1024 positions and >= 32768 outputs in all (tools/bench_conv_c1.py: 64 x 64 maps from 8 frames on,
// 32 x 32 maps from 32 frames on); below that the GEMM path is as fast
bool ge_conv3x3_c1_fwd_applies(int B, int H, int W) {
  constexpr int total = 32768;
  return H * W >= c1_min_plane() && (long long)B * H * W >= total;
}

long long ge_conv3x3_c1_wgrad_workspace(int B, int Cin) { return (long long)B * Cin * 9; }

int ge_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                      hipStream_t st) {
  const int units = ((H + 1) / 2) * (W / 4);     // 2 x 4 output blocks per frame
  hipLaunchKernelGGL(conv3x3_c1_fwd_kernel, dim3(ge_cdiv(units, 64), B), dim3(512), 0, st, x, w, bias, y, Cin, H, W,
                     units);
  ge_note_kernel("conv3x3_c1_fwd_kernel");
  GE_CHECK_LAUNCH("conv3x3_c1_fwd");
  return GE_OK;
}

int ge_conv3x3_c1_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int H, int W,
                        int accumulate, hipStream_t st) {
  const size_t lds = (size_t)(H + 2) * (W + 2) * sizeof(float);
  hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel, dim3(ge_cdiv(Cin, C1_CPB), B), dim3(256), lds, st, x, dy, workspace, Cin, H,
                     W);
  ge_note_kernel("conv3x3_c1_wgrad_kernel");
  GE_CHECK_LAUNCH("conv3x3_c1_wgrad");
  ge_record_split_event(st);
  const int n = Cin * 9;
  hipLaunchKernelGGL(c1_partial_sum_kernel, dim3(ge_cdiv(n, 256)), dim3(256), 0, st, workspace, dw, n, B, accumulate);
  GE_CHECK_LAUNCH("conv3x3_c1_partial_sum");
  return GE_OK;
}
