// Shared helpers for libgraphecho_hip.so (gfx950 / CDNA4 only).
// Every extern "C" entry point returns 0 on success or a negative error code and never
// throws, allocates or synchronises; the caller owns all buffers including workspaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <initializer_list>

#define GE_OK 0
#define GE_ERR_ARG -1
#define GE_ERR_LAUNCH -2
#define GE_ERR_UNSUPPORTED -3

void ge_set_error(const char* fmt, ...);
// Records the name (as rocprofv3 prints it) of the conv/GEMM kernel instantiation just launched by this thread.
void ge_note_kernel(const char* fmt, ...);
// Records the event set through ge_set_wgrad_split_event() (if any) on `st`.
void ge_record_split_event(hipStream_t st);

// 3x3 / stride 1 / pad 1 convolutions with one output channel (ge_conv_c1.hip), dispatched from ge_conv2d_fwd / _wgrad.
bool ge_conv3x3_c1_applies(int Cin, int Cout, int Hi, int Wi, int Ho, int Wo, int kh, int kw, int stride, int pad,
                           int groups);
bool ge_conv3x3_c1_wgrad_applies(int H, int W);
bool ge_conv3x3_c1_fwd_applies(int B, int H, int W);
long long ge_conv3x3_c1_wgrad_workspace(int B, int Cin);
int ge_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int H, int W,
                      hipStream_t st);
int ge_conv3x3_c1_wgrad(const float* x, const float* dy, float* dw, float* workspace, int B, int Cin, int H, int W,
                        int accumulate, hipStream_t st);

#define GE_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ge_set_error(__VA_ARGS__);         \
      return GE_ERR_ARG;                 \
    }                                    \
  } while (0)

#define GE_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      ge_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return GE_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the function's image on ONE device: a process that touches a second GPU
// must set it there too (a process-global "done" flag made every launch with over 64 KB of LDS fail on the second device).
// One GeLdsAttr per kernel instantiation; the return code of hipFuncSetAttribute is checked.
struct GeLdsAttr {
  int bytes[64] = {};      // largest dynamic-LDS size granted so far, per device (0: never set)
};
static inline int ge_set_max_lds(GeLdsAttr& a, const void* fn, int bytes, const char* name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    ge_set_error("%s: cannot query the current device", name);
    return GE_ERR_LAUNCH;
  }
  if (a.bytes[dev] < bytes) {      // also when a later launch of the same instantiation needs MORE than the first one did
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      ge_set_error("%s: %d bytes of dynamic LDS refused on device %d: %s", name, bytes, dev, hipGetErrorString(e));
      return GE_ERR_LAUNCH;
    }
    a.bytes[dev] = bytes;
  }
  return GE_OK;
}

// the same for several instantiations launched from one site (and for helpers that return void): a refusal is recorded through
// ge_set_error and surfaces as the launch failure GE_CHECK_LAUNCH reports right after
static inline void ge_set_max_lds_all(GeLdsAttr& a, std::initializer_list<const void*> fns, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || a.bytes[dev] >= bytes) return;
  for (const void* fn : fns) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      ge_set_error("%d bytes of dynamic LDS refused on device %d: %s", bytes, dev, hipGetErrorString(e));
      return;
    }
  }
  a.bytes[dev] = bytes;
}
#define GE_MAX_LDS(bytes, ...)                          \
  do {                                                  \
    static GeLdsAttr ge_lds_attr__;                     \
    ge_set_max_lds_all(ge_lds_attr__, {__VA_ARGS__}, (bytes)); \
  } while (0)

// Division by a runtime constant n / d for 0 <= n < 2^31 (mul-hi + shift).
struct FastDiv {
  uint32_t d, mul, shr;
};

static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) {
    f.mul = 0;
    f.shr = 0;
  } else {
    uint32_t lg = 0;
    while ((1ull << lg) < d) ++lg;  // ceil(log2 d)
    uint32_t p = 31 + lg;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
  }
  return f;
}

__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}
__device__ __forceinline__ void fd_divmod(uint32_t n, const FastDiv& f, uint32_t& q, uint32_t& r) {
  q = fd_div(n, f);
  r = n - q * f.d;
}

static inline int ge_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Grid size for grid-stride streaming kernels: enough workgroups to fill 256 CUs x 8.
static inline int ge_stream_grid(long long n, int per_block) {
  long long g = (n + per_block - 1) / per_block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// dst[0..n) = src[0..n) (src != nullptr) or 0, as a KERNEL.  The entry points never use hipMemsetAsync / hipMemcpyAsync for
// this: captured into a HIP graph those become memset / memcpy nodes, and on ROCm 7.2 a replayed memset node was observed
// to run unordered with the kernel nodes around it (the stride-2 1x1 data gradient kept stale pool memory at the positions
// only the memset writes; DESIGN.md, round-5 ledger).  n in floats; both pointers 4-byte aligned.
static __global__ void __launch_bounds__(256) ge_init_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                             int vec) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const long long n4 = n >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst);
    const float4* s4 = reinterpret_cast<const float4*>(src);
    for (long long j = i; j < n4; j += stride) d4[j] = src ? s4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long j = (n4 << 2) + i; j < n; j += stride) dst[j] = src ? src[j] : 0.f;
  } else {
    for (; i < n; i += stride) dst[i] = src ? src[i] : 0.f;
  }
}
static inline void ge_init_async(float* dst, const float* src, long long n, hipStream_t st) {
  if (n <= 0) return;
  const int vec = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  hipLaunchKernelGGL(ge_init_kernel, dim3(ge_stream_grid(vec ? (n + 3) / 4 : n, 256)), dim3(256), 0, st, dst, src, n, vec);
}

// Launch of a kernel whose workgroups MEET at a device-scope barrier (ge_sinkhorn.hip: rpm_coop_*, sd_fused_kernel with B > 1):
// every workgroup of the grid must get a slot while the others spin, so the launch goes through hipLaunchCooperativeKernel --
// the runtime refuses a grid that cannot be co-resident on the CURRENT device (CU-masked or partitioned devices included) and
// keeps two such launches from interleaving half-resident.  Probed on ROCm 7.2 / gfx950 (tools/microbench/coop_capture.hip,
// profiles/r06_coop_launch.txt): accepted inside a stream capture and replayed correctly from the graph, runs beside other
// streams' kernels, ~20 us more per launch than <<<>>>.  GE_COOP_LAUNCH=0: plain launch after the occupancy test below alone
// (the round-5 behaviour).  Either way the grid is first held to a QUARTER of occupancy x CUs of this kernel at this LDS size.
static inline int ge_launch_coresident(const void* fn, dim3 grid, dim3 block, void** args, size_t lds, hipStream_t st,
                                       const char* name) {
  static int cus[64];
  static const bool coop = [] {
    const char* e = getenv("GE_COOP_LAUNCH");
    return !(e && e[0] == '0');
  }();
  int dev = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    ge_set_error("%s: cannot query the current device", name);
    return GE_ERR_LAUNCH;
  }
  if (!cus[dev]) {
    int n = 0, ok = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
      ge_set_error("%s: cannot query the CU count", name);
      return GE_ERR_LAUNCH;
    }
    if (coop && (hipDeviceGetAttribute(&ok, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !ok)) {
      ge_set_error("%s: device %d does not support co-operative launches (GE_COOP_LAUNCH=0 for the plain launch)", name, dev);
      return GE_ERR_UNSUPPORTED;
    }
    cus[dev] = n;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, (int)(block.x * block.y * block.z), lds) != hipSuccess) {
    ge_set_error("%s: occupancy query failed", name);
    return GE_ERR_LAUNCH;
  }
  const long long slots = (long long)per_cu * cus[dev], need = (long long)grid.x * grid.y * grid.z;
  if (need * 4 > slots) {
    ge_set_error("%s: %lld workgroups must be co-resident, a quarter of this device holds %lld", name, need, slots / 4);
    return GE_ERR_UNSUPPORTED;
  }
  const hipError_t e = coop ? hipLaunchCooperativeKernel(fn, grid, block, args, (unsigned)lds, st)
                            : hipLaunchKernel(fn, grid, block, args, lds, st);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    ge_set_error("%s: %s launch failed: %s", name, coop ? "co-operative" : "plain", hipGetErrorString(e));
    return GE_ERR_LAUNCH;
  }
  return GE_OK;
}

// ---- wave64 reductions -----------------------------------------------------------------
// DPP butterflies inside each 16-lane row (quad_perm xor-1, xor-2, row_half_mirror, row_mirror: VALU-rate, no LDS
// crossbar round trips like ds_bpermute), then the four row totals are combined through v_readlane.  Every lane
// returns the same value.  Call with the whole wave active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v, float identity) {
  return __int_as_float(
      __builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f32(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1>(v, 0.f);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E>(v, 0.f);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141>(v, 0.f);   // row_half_mirror
  v += dpp_f32<0x140>(v, 0.f);   // row_mirror: every lane of a row now holds the row total
  return (readlane_f32(v, 0) + readlane_f32(v, 16)) + (readlane_f32(v, 32) + readlane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1>(v, v));
  v = fmaxf(v, dpp_f32<0x4E>(v, v));
  v = fmaxf(v, dpp_f32<0x141>(v, v));
  v = fmaxf(v, dpp_f32<0x140>(v, v));
  return fmaxf(fmaxf(readlane_f32(v, 0), readlane_f32(v, 16)), fmaxf(readlane_f32(v, 32), readlane_f32(v, 48)));
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Chan et al. merge of (count, mean, M2) moments.
__device__ __forceinline__ void moments_merge(float& n, float& mean, float& m2, float nb, float meanb,
                                              float m2b) {
  if (nb == 0.f) return;
  if (n == 0.f) {
    n = nb;
    mean = meanb;
    m2 = m2b;
    return;
  }
  const float nt = n + nb;
  const float delta = meanb - mean;
  mean += delta * (nb / nt);
  m2 += m2b + delta * delta * (n * nb / nt);
  n = nt;
}
template <int CTRL>
__device__ __forceinline__ void moments_dpp_step(float& n, float& mean, float& m2) {
  const float nb = dpp_f32<CTRL>(n, 0.f), mb = dpp_f32<CTRL>(mean, 0.f), qb = dpp_f32<CTRL>(m2, 0.f);
  moments_merge(n, mean, m2, nb, mb, qb);
}
__device__ __forceinline__ void wave_moments(float& n, float& mean, float& m2) {
  moments_dpp_step<0xB1>(n, mean, m2);
  moments_dpp_step<0x4E>(n, mean, m2);
  moments_dpp_step<0x141>(n, mean, m2);
  moments_dpp_step<0x140>(n, mean, m2);
  // combine the four 16-lane rows (wave-uniform from here on)
  float n0 = readlane_f32(n, 0), a0 = readlane_f32(mean, 0), q0 = readlane_f32(m2, 0);
  float n2 = readlane_f32(n, 32), a2 = readlane_f32(mean, 32), q2 = readlane_f32(m2, 32);
  moments_merge(n0, a0, q0, readlane_f32(n, 16), readlane_f32(mean, 16), readlane_f32(m2, 16));
  moments_merge(n2, a2, q2, readlane_f32(n, 48), readlane_f32(mean, 48), readlane_f32(m2, 48));
  moments_merge(n0, a0, q0, n2, a2, q2);
  n = n0;
  mean = a0;
  m2 = q0;
}

// GELU, erf form (nn.GELU default), and its derivative: shared by the element-wise activation kernels (ge_spatial.hip)
// and the BatchNorm kernels that fuse the activation (ge_norm.hip), so that fused and unfused give the same bits.
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float v) {
  const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
  return cdf + v * pdf;
}
