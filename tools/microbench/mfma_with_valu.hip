// Round 6: do fp32 MFMAs of one wave and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?  (The k-NN kernel's distance
// GEMM and its 64-bit compare / select top-k phases add up even when co-resident workgroups are started out of phase)
// A workgroup = 8 waves = 2 per SIMD.  Role by wave parity: even waves run `m` x 32 v_mfma_f32_32x32x2_f32 per iteration, odd waves
// `v` x 64 dependent-free VALU operations (fp32 FMAs, or 64-bit integer compare + select pairs like the top-k merge).  Cases: MFMA
// waves alone, VALU waves alone, both.  Prints the wall time of each; "both" == max(alone) means full overlap, == sum means none.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_with_valu mfma_with_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* sink, int iters, int do_mfma, int do_valu) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (!(wave & 1)) {
    if (!do_mfma) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = lane * 1e-3f, b = 1.0001f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.f) sink[0] = s;
  } else {
    if (!do_valu) return;
    if (KIND == 0) {      // fp32 FMAs, 16 independent chains
      float x[16];
      for (int j = 0; j < 16; ++j) x[j] = lane * 0.01f + j;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) x[j] = fmaf(x[j], 1.0001f, 0.5f);
      float s = 0.f;
      for (int j = 0; j < 16; ++j) s += x[j];
      if (s == 12345.f) sink[1] = s;
    } else if (KIND == 2) {      // 32-bit compare + select chains
      unsigned x[16], y[16];
      for (int j = 0; j < 16; ++j) {
        x[j] = lane * 977u + j;
        y[j] = x[j] ^ 0x55555u;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const unsigned m = x[j] < y[j] ? x[j] : y[j];
            y[j] = (x[j] < y[j] ? y[j] : x[j]) + 3u;
            x[j] = m + 1u;
          }
      unsigned s = 0;
      for (int j = 0; j < 16; ++j) s += x[j] ^ y[j];
      if (s == 12345u) sink[3] = (float)s;
    } else if (KIND == 3) {      // the same 64-bit min as KIND 1 with the comparison spelled in 32-bit operations
      unsigned xh[8], xl[8], yh[8], yl[8];
      for (int j = 0; j < 8; ++j) {
        xh[j] = lane * 977u + j;
        xl[j] = lane * 31u + j;
        yh[j] = xh[j] ^ 0x5u;
        yl[j] = xl[j] ^ 0x55555u;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool lt = (xh[j] < yh[j]) | ((xh[j] == yh[j]) & (xl[j] < yl[j]));
            const unsigned mh = lt ? xh[j] : yh[j], ml = lt ? xl[j] : yl[j];
            const unsigned Mh = lt ? yh[j] : xh[j], Ml = lt ? yl[j] : xl[j];
            yh[j] = Mh + 1u;      // (no carry chains: 32-bit adds on each half)
            yl[j] = Ml + 3u;
            xh[j] = mh;
            xl[j] = ml + 1u;
          }
      unsigned s = 0;
      for (int j = 0; j < 8; ++j) s += xh[j] ^ yl[j] ^ xl[j] ^ yh[j];
      if (s == 12345u) sink[4] = (float)s;
    } else if (KIND == 5) {      // integer add / xor / shift chains: no compare, no select
      unsigned x[16];
      for (int j = 0; j < 16; ++j) x[j] = lane * 977u + j;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) x[j] = ((x[j] + 0x9E3779B9u) ^ (x[j] >> 7)) + (x[j] << 3);
      unsigned s = 0;
      for (int j = 0; j < 16; ++j) s += x[j];
      if (s == 12345u) sink[6] = (float)s;
    } else if (KIND == 6) {      // v_min_u32 / v_max_u32 chains (no VCC)
      unsigned x[16], y[16];
      for (int j = 0; j < 16; ++j) {
        x[j] = lane * 977u + j;
        y[j] = x[j] ^ 0x55555u;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const unsigned m = min(x[j], y[j]), M = max(x[j], y[j]);
            y[j] = M + 3u;
            x[j] = m + 1u;
          }
      unsigned s = 0;
      for (int j = 0; j < 16; ++j) s += x[j] ^ y[j];
      if (s == 12345u) sink[7] = (float)s;
    } else if (KIND == 7) {      // v_min_f64 / v_max_f64 chains on positive doubles (64-bit keys ordered as doubles)
      double x[8], y[8];
      for (int j = 0; j < 8; ++j) {
        x[j] = 1.0 + lane * 1e-3 + j;
        y[j] = x[j] * 1.0000001 + 0.25;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double m = fmin(x[j], y[j]), M = fmax(x[j], y[j]);
            y[j] = M;
            x[j] = m;
            asm volatile("" : "+v"(x[j]), "+v"(y[j]));
          }
      double s = 0;
      for (int j = 0; j < 8; ++j) s += x[j] + y[j];
      if (s == 12345.0) sink[8] = (float)s;
    } else if (KIND == 8) {      // v_cndmask chains with a LOOP-INVARIANT mask (no compare inside the loop)
      unsigned x[16], y[16];
      const bool pick = (lane * 2654435761u) & 0x10000u;
      for (int j = 0; j < 16; ++j) {
        x[j] = lane * 977u + j;
        y[j] = x[j] ^ 0x55555u;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const unsigned a = pick ? x[j] : y[j], b = pick ? y[j] : x[j];
            x[j] = a + 1u;
            y[j] = b + 3u;
          }
      unsigned s = 0;
      for (int j = 0; j < 16; ++j) s += x[j] ^ y[j];
      if (s == 12345u) sink[9] = (float)s;
    } else if (KIND == 9) {      // saturating 32-bit adds (v_add_u32 ... clamp): the conv loaders' offset + chunk step
      unsigned x[16];
      for (int j = 0; j < 16; ++j) x[j] = lane * 977u + j;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            x[j] = __builtin_elementwise_add_sat(x[j], 0x01000193u + j);
            asm volatile("" : "+v"(x[j]));
          }
      unsigned s = 0;
      for (int j = 0; j < 16; ++j) s += x[j];
      if (s == 12345u) sink[10] = (float)s;
    } else if (KIND == 10) {     // packed fp32 adds (v_pk_add_f32): the Winograd input transform
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 x[16];
      for (int j = 0; j < 16; ++j) x[j] = f2{lane * 0.01f + j, lane * 0.02f - j};
      const f2 d = f2{1.0001f, 0.5f};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            x[j] = x[j] + d;
            asm volatile("" : "+v"(x[j]));
          }
      float s = 0.f;
      for (int j = 0; j < 16; ++j) s += x[j].x + x[j].y;
      if (s == 12345.f) sink[11] = s;
    } else if (KIND == 11) {     // 64-bit address arithmetic (v_lshl_add_u64)
      unsigned long long x[16];
      for (int j = 0; j < 16; ++j) x[j] = ((unsigned long long)(lane * 977 + j) << 28) + j;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            x[j] = (x[j] << 1) + (unsigned long long)(j + 3);
            asm volatile("" : "+v"(x[j]));
          }
      unsigned long long s = 0;
      for (int j = 0; j < 16; ++j) s += x[j];
      if (s == 12345ull) sink[12] = (float)s;
    } else if (KIND == 4) {      // 64-bit adds only (v_add_co + v_addc carry pairs), no compares
      unsigned long long x[16];
      for (int j = 0; j < 16; ++j) x[j] = ((unsigned long long)(lane * 977 + j) << 28) + j;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) x[j] += 0x100000001ull * (j + 1);
      unsigned long long s = 0;
      for (int j = 0; j < 16; ++j) s += x[j];
      if (s == 12345ull) sink[5] = (float)s;
    } else {              // 64-bit unsigned min chains (compare + two selects), like the top-k merge
      unsigned long long x[8], y[8];
      for (int j = 0; j < 8; ++j) {
        x[j] = (unsigned long long)(lane * 977 + j) << 20;
        y[j] = x[j] ^ 0x5555555555ull;
      }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const unsigned long long m = x[j] < y[j] ? x[j] : y[j];
            y[j] = (x[j] < y[j] ? y[j] : x[j]) + 3;
            x[j] = m + 1;
          }
      unsigned long long s = 0;
      for (int j = 0; j < 8; ++j) s += x[j] ^ y[j];
      if (s == 12345ull) sink[2] = (float)s;
    }
  }
}

template <int KIND>
static float run(float* sink, int iters, int m, int v) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k<KIND><<<256 * 4, 512>>>(sink, iters, m, v);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k<KIND><<<256 * 4, 512>>>(sink, iters, m, v);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* sink;
  hipMalloc(&sink, 64);
  const int iters = 2000;
  {
    const float tm = run<0>(sink, iters, 1, 0), tv = run<0>(sink, iters, 0, 1), tb = run<0>(sink, iters, 1, 1);
    const double fl = 256.0 * 4 * 4 * iters * 32 * 2.0 * 32 * 32 * 2;      // 4 MFMA waves per workgroup
    printf("fp32 FMA VALU waves : MFMA alone %.3f ms (%.1f TFLOP/s), VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm,
           fl / tm / 1e9, tv, tb, tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<1>(sink, iters, 1, 0), tv = run<1>(sink, iters, 0, 1), tb = run<1>(sink, iters, 1, 1);
    printf("64-bit min VALU waves: MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<2>(sink, iters, 1, 0), tv = run<2>(sink, iters, 0, 1), tb = run<2>(sink, iters, 1, 1);
    printf("32-bit cmp + select   : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<3>(sink, iters, 1, 0), tv = run<3>(sink, iters, 0, 1), tb = run<3>(sink, iters, 1, 1);
    printf("64-bit min in 32-bit ops: MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<4>(sink, iters, 1, 0), tv = run<4>(sink, iters, 0, 1), tb = run<4>(sink, iters, 1, 1);
    printf("64-bit adds (carry)   : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<5>(sink, iters, 1, 0), tv = run<5>(sink, iters, 0, 1), tb = run<5>(sink, iters, 1, 1);
    printf("int add / xor / shift : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<6>(sink, iters, 1, 0), tv = run<6>(sink, iters, 0, 1), tb = run<6>(sink, iters, 1, 1);
    printf("v_min_u32 / v_max_u32 : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<7>(sink, iters, 1, 0), tv = run<7>(sink, iters, 0, 1), tb = run<7>(sink, iters, 1, 1);
    printf("v_min_f64 / v_max_f64 : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<8>(sink, iters, 1, 0), tv = run<8>(sink, iters, 0, 1), tb = run<8>(sink, iters, 1, 1);
    printf("v_cndmask, fixed mask : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<9>(sink, iters, 1, 0), tv = run<9>(sink, iters, 0, 1), tb = run<9>(sink, iters, 1, 1);
    printf("v_add_u32 clamp (sat) : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<10>(sink, iters, 1, 0), tv = run<10>(sink, iters, 0, 1), tb = run<10>(sink, iters, 1, 1);
    printf("v_pk_add_f32          : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  {
    const float tm = run<11>(sink, iters, 1, 0), tv = run<11>(sink, iters, 0, 1), tb = run<11>(sink, iters, 1, 1);
    printf("v_lshl_add_u64        : MFMA alone %.3f ms, VALU alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", tm, tv, tb,
           tm > tv ? tm : tv, tm + tv);
  }
  return 0;
}
