// Round 6: do fp32 MFMAs of one wave and VALU work of ANOTHER wave ON THE SAME SIMD overlap on gfx950?
// Placement is made explicit: workgroups of 4 waves with 72 KB of LDS (at most two per CU), grid = 2 x CUs, every workgroup reads its
// physical (XCC, SE, CU) from the hardware-id registers and takes a ticket from that CU's counter: ticket 0 = MFMA role, ticket 1 =
// VALU role.  Each wave also records its SIMD; the host checks that every CU got one workgroup of each role and every SIMD one wave
// of each.  MFMA role: `iters` x 32 v_mfma_f32_32x32x2_f32.  VALU role: chains of ONE instruction class (template KIND).
// Cases: MFMA role alone (VALU workgroups exit), VALU role alone, both.  both ~ max(alone): the classes overlap; ~ sum: they exclude.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_with_valu mfma_with_valu.hip        (-DMFMA_F16 -o mfma16_with_valu: fp16 MFMAs in the MFMA role)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned cu_key(unsigned& simd) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID, 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // HW_REG_XCC_ID, bits 3:0
  simd = (hw >> 4) & 3;
  const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  return ((xcc & 15) << 8) | (se << 5) | (sh << 4) | cu;
}

template <int KIND>
__device__ __forceinline__ void valu_role(float* sink, int iters, int lane) {
  if (KIND == 0) {             // v_fma_f32
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
  } else if (KIND == 1) {      // v_cmp_lt_u64 / v_cmp_gt_u64 + v_cndmask: the top-k merge's 64-bit key min
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
  } else if (KIND == 2) {      // v_min_u32 / v_max_u32 + v_add_u32
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
    if (s == 12345u) sink[3] = (float)s;
  } else if (KIND == 3) {      // v_add_u32 / v_xad_u32 / shifts
    unsigned x[16];
    for (int j = 0; j < 16; ++j) x[j] = lane * 977u + j;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = ((x[j] + 0x9E3779B9u) ^ (x[j] >> 7)) + (x[j] << 3);
    unsigned s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 12345u) sink[4] = (float)s;
  } else if (KIND == 4) {      // v_add_u32 ... clamp: the conv loaders' saturating offset step
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
    if (s == 12345u) sink[5] = (float)s;
  } else {                     // v_pk_add_f32: the Winograd input transform
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
    if (s == 12345.f) sink[6] = s;
  }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void k(float* sink, int iters, int do_mfma, int do_valu, int* tickets, unsigned* place) {
  extern __shared__ float pad[];      // 72 KB: two workgroups per CU at most
  __shared__ int s_role;
  unsigned simd;
  const unsigned key = cu_key(simd);
  if (threadIdx.x == 0) s_role = atomicAdd(tickets + key, 1);
  __syncthreads();
  const int role = s_role, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) place[blockIdx.x * 4 + wave] = (key << 8) | (simd << 4) | (unsigned)role;
  if (pad[0] == 12345.f) sink[7] = 1.f;
  if (role == 0) {
    if (!do_mfma) return;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#ifdef MFMA_F16      // -DMFMA_F16: the MFMA role issues v_mfma_f32_32x32x16_f16 (config 5's kernels) -- twice as many, half as long each
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
      a[j] = (_Float16)(0.731f * (1.f + lane * 1e-3f) + j * 0.01f);
      b[j] = (_Float16)(1.0001f - j * 0.01f);
    }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 16; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#else
    const float a = 0.731f * (1.f + lane * 1e-3f), b = 1.0001f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#endif
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.f) sink[0] = s;
  } else {
    if (!do_valu) return;
    valu_role<KIND>(sink, iters, lane);
  }
}

static int ncu = 256;
template <int KIND>
static float run(float* sink, int iters, int m, int v, int* tickets, unsigned* place, bool check) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(tickets, 0, 4096 * 4);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    k<KIND><<<2 * ncu, 256, 72 * 1024>>>(sink, iters, m, v, tickets, place);
    (void)hipEventRecord(b);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  if (check) {
    static unsigned h[8192];
    (void)hipMemcpy(h, place, 2 * ncu * 4 * 4, hipMemcpyDeviceToHost);
    static int per[4096][2][4];
    memset(per, 0, sizeof(per));
    for (int i = 0; i < 2 * ncu * 4; ++i) per[(h[i] >> 8) & 4095][h[i] & 1][(h[i] >> 4) & 3]++;
    int good = 0, cus = 0;
    for (int c = 0; c < 4096; ++c) {
      int n = 0, ok = 1;
      for (int r = 0; r < 2; ++r)
        for (int s = 0; s < 4; ++s) {
          n += per[c][r][s];
          ok &= per[c][r][s] == 1;
        }
      if (n) {
        ++cus;
        good += ok;
      }
    }
    printf("placement: %d CUs seen, %d of them hold exactly one MFMA wave and one VALU wave on each of their 4 SIMDs\n", cus, good);
  }
  return best;
}

template <int KIND>
static void report(const char* name, float* sink, int* tickets, unsigned* place, bool check) {
  const int iters = 2000;
  const float tm = run<KIND>(sink, iters, 1, 0, tickets, place, check), tv = run<KIND>(sink, iters, 0, 1, tickets, place, false),
              tb = run<KIND>(sink, iters, 1, 1, tickets, place, false);
#ifdef MFMA_F16
  const double fl = (double)ncu * 4 * iters * 64 * 32768.0;
#else
  const double fl = (double)ncu * 4 * iters * 32 * 4096.0;
#endif
  printf("%-34s MFMA alone %6.3f ms (%5.1f TFLOP/s)  VALU alone %6.3f ms  both %6.3f ms  | max %6.3f  sum %6.3f  hidden %3.0f %% of the shorter\n",
         name, tm, fl / tm / 1e9, tv, tb, tm > tv ? tm : tv, tm + tv, 100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main() {
  float* sink;
  int* tickets;
  unsigned* place;
  (void)hipMalloc(&sink, 64);
  (void)hipMalloc(&tickets, 4096 * 4);
  (void)hipMalloc(&place, 8192 * 4);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  report<0>("v_fma_f32", sink, tickets, place, true);
  report<5>("v_pk_add_f32", sink, tickets, place, false);
  report<3>("v_add_u32 / v_xad_u32 / shifts", sink, tickets, place, false);
  report<4>("v_add_u32 clamp (saturating)", sink, tickets, place, false);
  report<2>("v_min_u32 / v_max_u32 (+ adds)", sink, tickets, place, false);
  report<1>("v_cmp_*_u64 + v_cndmask (key min)", sink, tickets, place, false);
  return 0;
}
