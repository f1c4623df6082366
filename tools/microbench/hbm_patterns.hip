// HBM throughput of the access patterns a 1x1-conv GEMM produces, without the GEMM: what the memory side alone allows.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_patterns tools/microbench/hbm_patterns.hip && /tmp/hbm_patterns
// Patterns (B = 32 images, HW = 4096 positions, fp32):
//   stream r:w   -- grid-stride float4 copy reading R planes and writing W planes per position block
//   tile  r:w    -- a wave reads Cin rows and writes Cout rows of a TN-position tile (row segments of TN*4 bytes, 16 KB apart),
//                   the order conv_gemm_kernel / a wave-tile kernel touches memory in
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n_src,
                                                     long long n_dst) {
  const long long stride = (long long)gridDim.x * 256;
  float4 acc = make_float4(0, 0, 0, 0);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_src; i += stride) {
    const float4 v = src[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_dst; i += stride) dst[i] = acc;
}

// interleaved: every workgroup alternates reading a block and writing W/R blocks (what a tile kernel does in time)
__global__ __launch_bounds__(256) void tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cin, int Cout,
                                                   int HW, int TN, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_row = TN / 4;              // lanes per row segment
  const int rows_per_pass = 64 / per_row;
  for (int t = blockIdx.x * 4 + wave; t < tiles; t += gridDim.x * 4) {
    const long long n0 = (long long)t * TN;
    const long long b = n0 / HW, hw0 = n0 % HW;
    float4 acc = make_float4(0, 0, 0, 0);
    const float* s = src + (b * Cin) * HW + hw0 + (lane % per_row) * 4;
    for (int k = lane / per_row; k < Cin; k += rows_per_pass) {
      const float4 v = *(const float4*)(s + (long long)k * HW);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* d = dst + (b * Cout) * HW + hw0 + (lane % per_row) * 4;
    for (int m = lane / per_row; m < Cout; m += rows_per_pass) *(float4*)(d + (long long)m * HW) = acc;
  }
}

static float time_it(void (*launch)(void*), void* ctx) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch(ctx);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) launch(ctx);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20;
}

struct Ctx { float *src, *dst; int Cin, Cout, HW, TN, B, wgs; };
static void launch_stream(void* p) {
  Ctx* c = (Ctx*)p;
  hipLaunchKernelGGL(stream_kernel, dim3(c->wgs), dim3(256), 0, 0, (const float4*)c->src, (float4*)c->dst,
                     (long long)c->B * c->Cin * c->HW / 4, (long long)c->B * c->Cout * c->HW / 4);
}
static void launch_tile(void* p) {
  Ctx* c = (Ctx*)p;
  hipLaunchKernelGGL(tile_kernel, dim3(c->wgs), dim3(256), 0, 0, c->src, c->dst, c->Cin, c->Cout, c->HW, c->TN,
                     (int)((long long)c->B * c->HW / c->TN));
}

int main() {
  const int B = 32;
  float *src, *dst;
  const size_t cap = (size_t)B * 1024 * 4096 * 4;
  hipMalloc(&src, cap); hipMalloc(&dst, cap);
  hipMemset(src, 0, cap); hipMemset(dst, 0, cap);
  const int shapes[][3] = {{64, 256, 4096}, {256, 64, 4096}, {256, 256, 4096}, {64, 64, 4096}, {128, 512, 1024}, {512, 128, 1024}, {1024, 256, 256}};
  for (auto& s : shapes) {
    Ctx c{src, dst, s[0], s[1], s[2], 64, B, 2048};
    const double bytes = 4.0 * B * s[2] * (s[0] + s[1]);
    float ms = time_it(launch_stream, &c);
    printf("Cin %4d Cout %4d HW %4d (%6.1f MB): stream %6.1f us %5.2f TB/s", s[0], s[1], s[2], bytes / 1e6, ms * 1e3, bytes / ms / 1e9);
    for (int tn : {64, 128, 256}) {
      for (int wgs : {1024, 4096}) {
        c.TN = tn; c.wgs = wgs;
        ms = time_it(launch_tile, &c);
        printf(" | tile%d/%d %6.1f us %5.2f", tn, wgs, ms * 1e3, bytes / ms / 1e9);
      }
    }
    printf("\n");
  }
  return 0;
}
