// Round 6: what does a PURE fp32-MFMA stream sustain on this part, as a function of how long it runs and of the operand data?
// 1024 workgroups x 4 waves, every wave `iters` x 32 independent-accumulator v_mfma_f32_32x32x2_f32, no memory traffic at all.
// Prints TFLOP/s and the shader clock (clock64 ticks per wall_clock64 tick x 100 MHz) for kernels of ~0.5 ms to ~40 ms, on
// all-zero operands and on lane-dependent non-trivial operands.  The 157.3 TFLOP/s of the data sheet is 256 CUs x 4 SIMDs x 64
// FLOP/cycle x 2.4 GHz; the training step keeps the matrix pipe busy for tens of milliseconds at a time.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_sustained mfma_sustained.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(256, 4) void k(float* sink, int iters, float a0, float b0, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = a0 * (1.f + lane * 1e-3f), b = b0;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = c1 - c0;
    clk[1] = w1 - w0;
  }
}

int main() {
  float* sink;
  unsigned long long *clk, h[2];
  hipMalloc(&sink, 64);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nwg = 1024;
  for (int data = 0; data < 2; ++data)
    for (int iters : {128, 512, 2048, 8192}) {
      const float a = data ? 0.731f : 0.f, b = data ? 1.0001f : 0.f;
      k<<<nwg, 256>>>(sink, iters, a, b, clk);      // warm
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<nwg, 256>>>(sink, iters, a, b, clk);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double fl = (double)nwg * 4 * iters * 32 * 4096.0;
      printf("%s operands, %5d iterations: %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)  shader clock %.0f MHz\n",
             data ? "non-trivial" : "all-zero   ", iters, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, 100.0 * (double)h[0] / (double)h[1]);
    }
  // occupancy: 1 .. 4 workgroups of 4 waves per CU = 1 .. 4 MFMA waves per SIMD (256 CUs), the same loop
  for (int per_cu = 1; per_cu <= 4; ++per_cu) {
    const int g = 256 * per_cu, iters = 4096;
    k<<<g, 256>>>(sink, iters, 0.731f, 1.0001f, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<g, 256>>>(sink, iters, 0.731f, 1.0001f, clk);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float t = 0;
    hipEventElapsedTime(&t, e0, e1);
    const double fl = (double)g * 4 * iters * 32 * 4096.0;
    printf("%d MFMA wave(s) per SIMD (grid %4d): %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)\n", per_cu, g, t, fl / t / 1e9, fl / t / 1e9 / 157.3);
  }
  // back-to-back kernels of the long form: the sustained regime of a training step
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) k<<<nwg, 256>>>(sink, 2048, 0.731f, 1.0001f, clk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("10 x non-trivial 2048 iterations back to back: %.3f ms  %.1f TFLOP/s\n", ms, 10.0 * nwg * 4 * 2048 * 32 * 4096.0 / ms / 1e9);
  return 0;
}
