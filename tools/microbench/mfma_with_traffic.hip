// Does the fp32 matrix pipe keep its rate while the same waves stream HBM?  Every wave runs ITERS x 32 independent-accumulator
// v_mfma_f32_32x32x2_f32 (the MFMAs of one 16-deep K chunk of a 64x64 wave tile) and, per iteration, L 16-byte loads and S
// 16-byte stores per lane to addresses nobody else touches (no LDS, no barriers, no dependence of the MFMAs on the loads).
// Prints TFLOP/s, TB/s and the shader clock (clock64 ticks / wall_clock64 ticks x 100 MHz) for a sweep of (L, S).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mwt tools/microbench/mfma_with_traffic.hip && /tmp/mwt
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int L, int S, int D = 1, int DEP = 1>
__global__ __launch_bounds__(256, 4) void k(const float4* __restrict__ src, float4* __restrict__ dst, float* sink, int iters,
                                            unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = lane * 1e-3f, b = 1.f;
  float4 keep = make_float4(0, 0, 0, 0);
  float cnt = lane;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float4 v[L > 0 ? L : 1], v2[L > 0 ? L : 1];
  if (D == 2) {
    const long long blk = wave * 64 + lane;
#pragma unroll
    for (int q = 0; q < L; ++q) v2[q] = src[blk * L + q * 64 - (long long)lane * (L - 1)];
  }
  for (int it = 0; it < iters; ++it) {
    const long long blk = ((long long)(it + (D == 2 ? 1 : 0)) * nwaves + wave) * 64 + lane;
    float early[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) early[i] = acc[i][0];
    if (D == 2) {      // the loads issued one iteration ago are the ones consumed at the end of this one
#pragma unroll
      for (int q = 0; q < L; ++q) {
        const float4 t = v2[q];
        v2[q] = src[blk * L + q * 64 - (long long)lane * (L - 1)];
        v[q] = t;
      }
    } else {
#pragma unroll
      for (int q = 0; q < L; ++q) v[q] = src[blk * L + q * 64 - (long long)lane * (L - 1)];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < L; ++q) {
      keep.x += v[q].x;
      keep.y += v[q].w;
    }
#pragma unroll
    for (int q = 0; q < S; ++q) {
      // DEP 0: loop-invariant data; 1: an MFMA result of this iteration; 2: an MFMA result copied at the top of the iteration
      // (i.e. of the previous iteration); 3: data that changes every iteration but does not come from the matrix pipe
      const float d = DEP == 1 ? acc[q & 3][0] : DEP == 2 ? early[q & 3] : DEP == 3 ? cnt : b;
      if (DEP == 4)      // four accumulator registers as they are: no VALU instruction touches the result
        dst[blk * S + q * 64 - (long long)lane * (S - 1)] = make_float4(acc[q & 3][0], acc[q & 3][1], acc[q & 3][2], acc[q & 3][3]);
      else
        dst[blk * S + q * 64 - (long long)lane * (S - 1)] = make_float4(d, keep.x, a, b);
    }
    cnt += 1.25f;
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = keep.x + keep.y;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

template <int L, int S, int D = 1, int DEP = 1>
static void run(const float4* src, float4* dst, float* sink, unsigned long long* clk, int wgs, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<L, S, D, DEP>), dim3(wgs), dim3(256), 0, 0, src, dst, sink, iters, clk);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<L, S, D, DEP>), dim3(wgs), dim3(256), 0, 0, src, dst, sink, iters, clk);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  unsigned long long h[2];
  (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double fl = 2.0 * 32 * 32 * 2 * 32 * (double)iters * wgs * 4;
  const double by = 1024.0 * (L + S) * (double)iters * wgs * 4;
  printf("L %d S %d depth %d stores %s: %8.1f us  %6.1f TFLOP/s  %5.2f TB/s  shader clock %4.0f MHz\n", L, S, D, DEP == 1 ? "of results" : DEP == 2 ? "of last iteration's results" : DEP == 4 ? "of result registers directly" : DEP == 3 ? "changing data" : "constant data", ms * 1e3, fl / ms / 1e9, by / ms / 1e9,
         (double)h[0] / (double)h[1] * 100.0);
}

int main() {
  const int wgs = 1024, iters = 128;   // 4 workgroups per CU, one round
  const size_t cap = (size_t)1024 * 8 * (iters + 1) * wgs * 4;
  float4 *src, *dst;
  float* sink;
  unsigned long long* clk;
  (void)hipMalloc(&src, cap);
  (void)hipMalloc(&dst, cap);
  (void)hipMalloc(&sink, 64);
  (void)hipMalloc(&clk, 64);
  (void)hipMemset(src, 0, cap);
  (void)hipMemset(dst, 0, cap);
  run<0, 0>(src, dst, sink, clk, wgs, iters);
  run<1, 0>(src, dst, sink, clk, wgs, iters);
  run<2, 0>(src, dst, sink, clk, wgs, iters);
  run<4, 0>(src, dst, sink, clk, wgs, iters);
  run<8, 0>(src, dst, sink, clk, wgs, iters);
  run<0, 1>(src, dst, sink, clk, wgs, iters);
  run<0, 2>(src, dst, sink, clk, wgs, iters);
  run<0, 4>(src, dst, sink, clk, wgs, iters);
  run<4, 1>(src, dst, sink, clk, wgs, iters);
  run<4, 4>(src, dst, sink, clk, wgs, iters);
  run<2, 2>(src, dst, sink, clk, wgs, iters);
  run<4, 0, 2>(src, dst, sink, clk, wgs, iters);
  run<8, 0, 2>(src, dst, sink, clk, wgs, iters);
  run<0, 2, 1, 0>(src, dst, sink, clk, wgs, iters);
  run<0, 4, 1, 0>(src, dst, sink, clk, wgs, iters);
  run<4, 1, 2>(src, dst, sink, clk, wgs, iters);
  run<0, 2, 1, 2>(src, dst, sink, clk, wgs, iters);
  run<0, 4, 1, 2>(src, dst, sink, clk, wgs, iters);
  run<0, 2, 1, 4>(src, dst, sink, clk, wgs, iters);
  run<0, 4, 1, 4>(src, dst, sink, clk, wgs, iters);
  run<0, 2, 1, 3>(src, dst, sink, clk, wgs, iters);
  run<0, 4, 1, 3>(src, dst, sink, clk, wgs, iters);
  run<4, 1, 2, 0>(src, dst, sink, clk, wgs, iters);
  run<2, 2, 2, 0>(src, dst, sink, clk, wgs, iters);
  return 0;
}
