// Probe (round 6): what hipLaunchCooperativeKernel does on this stack (ROCm 7.2, gfx950) for the hand-barrier kernels of
// ge_sinkhorn.hip -- (1) eager launch time next to a plain <<<>>> launch of the same kernel, (2) whether it is accepted inside
// a stream capture and replays correctly from the instantiated graph, (3) whether two co-operative launches on two streams
// overlap each other and a long plain kernel on a third stream, (4) what hipOccupancyMaxActiveBlocksPerMultiprocessor reports.
// Build: hipcc --offload-arch=gfx950 -O2 -o coop_capture coop_capture.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("FAIL %s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
      fails++;                                                                      \
    }                                                                               \
  } while (0)
static int fails = 0;

// G workgroups meet `rounds` times at a monotonic counter; out[g] = number of rounds in which every partial was visible
__global__ __launch_bounds__(256) void meet_kernel(int* counter, float* part, int* out, int rounds, int spin) {
  const int g = blockIdx.x, G = gridDim.x;
  int ok = 0;
  for (int t = 0; t < rounds; ++t) {
    if (threadIdx.x == 0) __hip_atomic_store(part + (t & 1) * G + g, (float)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (t + 1) * G) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    int good = 1;
    for (int j = threadIdx.x; j < G; j += 256)
      good &= __hip_atomic_load(part + (t & 1) * G + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (float)(t + 1);
    ok += __syncthreads_and(good);
    for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) out[g] = ok;
}
__global__ void zero_kernel(int* p, int n) {
  if (threadIdx.x < n) p[threadIdx.x] = 0;
}
__global__ __launch_bounds__(256) void busy_kernel(float* p, int iters) {
  float v = p[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
  p[blockIdx.x * 256 + threadIdx.x] = v;
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const int G = 16, R = 20;
  int *counter[2], *out[2];
  float *part[2], *busy;
  for (int i = 0; i < 2; ++i) {
    CK(hipMalloc(&counter[i], 64));
    CK(hipMalloc(&out[i], G * 4));
    CK(hipMalloc(&part[i], 2 * G * 4));
  }
  CK(hipMalloc(&busy, 4096 * 256 * 4));
  CK(hipMemset(busy, 0, 4096 * 256 * 4));
  hipStream_t s[3];
  for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  int per_cu = 0, cus = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)meet_kernel, 256, 0));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  int coop = 0;
  CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
  printf("occupancy %d blocks/CU x %d CUs; cooperativeLaunch attribute %d\n", per_cu, cus, coop);

  int rounds = R, spin = 0;
  auto check = [&](int i, const char* what) {
    std::vector<int> h(G);
    CK(hipMemcpy(h.data(), out[i], G * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int v : h) bad += v != R;
    printf("  %-44s %s\n", what, bad ? "WRONG" : "ok");
    if (bad) fails++;
  };
  auto launch = [&](int i, hipStream_t st, bool co) -> hipError_t {
    hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, st, counter[i], 1);
    void* args[] = {&counter[i], &part[i], &out[i], &rounds, &spin};
    if (co) return hipLaunchCooperativeKernel((const void*)meet_kernel, dim3(G), dim3(256), args, 0, st);
    hipLaunchKernelGGL(meet_kernel, dim3(G), dim3(256), 0, st, counter[i], part[i], out[i], rounds, spin);
    return hipGetLastError();
  };
  // (1) eager, plain vs co-operative: host time per launch and device time per launch
  for (int co = 0; co < 2; ++co) {
    CK(launch(0, s[0], co));
    CK(hipStreamSynchronize(s[0]));
    check(0, co ? "eager co-operative launch" : "eager plain launch");
    const int N = 200;
    double t0 = now_us();
    for (int k = 0; k < N; ++k) CK(launch(0, s[0], co));
    double t1 = now_us();
    CK(hipStreamSynchronize(s[0]));
    double t2 = now_us();
    printf("  %s: host %.1f us / launch pair, end-to-end %.1f us / launch pair\n", co ? "co-operative" : "plain", (t1 - t0) / N,
           (t2 - t0) / N);
  }
  // (2) inside a capture
  for (int co = 0; co < 2; ++co) {
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    hipError_t e = launch(0, s[0], co);
    hipError_t e2 = hipStreamEndCapture(s[0], &g);
    printf("  capture of a %s launch: launch -> %s, end capture -> %s\n", co ? "co-operative" : "plain", hipGetErrorString(e),
           hipGetErrorString(e2));
    if (e == hipSuccess && e2 == hipSuccess && g) {
      hipError_t e3 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      printf("  instantiate -> %s\n", hipGetErrorString(e3));
      if (e3 == hipSuccess) {
        CK(hipMemset(out[0], 0, G * 4));
        for (int k = 0; k < 50; ++k) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        check(0, co ? "replayed co-operative node x50" : "replayed plain node x50");
        double t0 = now_us();
        for (int k = 0; k < 200; ++k) CK(hipGraphLaunch(ge, s[0]));
        CK(hipStreamSynchronize(s[0]));
        printf("  graph replay (%s): %.1f us per replay\n", co ? "co-operative" : "plain", (now_us() - t0) / 200);
        CK(hipGraphExecDestroy(ge));
      }
    } else {
      (void)hipGetLastError();
    }
    if (g) CK(hipGraphDestroy(g));
  }
  // (3) two barrier kernels on two streams beside a chip-filling plain kernel on a third; long rounds so that they must overlap
  spin = 200;
  for (int co = 0; co < 2; ++co) {
    hipEvent_t e0, e1, b0, b1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&b0));
    CK(hipEventCreate(&b1));
    CK(hipDeviceSynchronize());
    double t0 = now_us();
    CK(hipEventRecord(b0, s[2]));
    hipLaunchKernelGGL(busy_kernel, dim3(4096), dim3(256), 0, s[2], busy, 400000);
    CK(hipEventRecord(b1, s[2]));
    CK(hipEventRecord(e0, s[0]));
    CK(launch(0, s[0], co));
    CK(launch(1, s[1], co));
    CK(hipEventRecord(e1, s[0]));
    CK(hipDeviceSynchronize());
    double t1 = now_us();
    float mb = 0, mm = 0;
    CK(hipEventElapsedTime(&mb, b0, b1));
    CK(hipEventElapsedTime(&mm, e0, e1));
    check(0, "stream 0 beside busy kernel");
    check(1, "stream 1 beside busy kernel");
    printf("  %s: busy kernel %.2f ms, barrier kernel on stream 0 %.2f ms, wall %.2f ms (sum would be the serial case)\n",
           co ? "co-operative" : "plain", mb, mm, (t1 - t0) / 1000);
  }
  printf("%s\n", fails ? "PROBE HAD FAILURES" : "probe done");
  return 0;
}
