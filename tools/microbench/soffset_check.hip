// Round 6: range-check semantics of raw buffer loads with a non-zero SGPR offset on gfx950: is `soffset` part of the bounds test?
// A buffer of NREC bytes; lanes load with (voffset, soffset) combinations around the end of the range and with the all-ones sentinel.
//   hipcc --offload-arch=gfx950 -O2 -o soffset_check soffset_check.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__global__ void k(const float* src, float* out, unsigned nrec, unsigned soff) {
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
  const unsigned voffs[8] = {0u, 4u, nrec - 8u, nrec - 4u, nrec, nrec + 4u, 0xFFFFFFF0u, 0xFFFFFFFFu};
  const unsigned v = voffs[threadIdx.x & 7];
  const unsigned s = __builtin_amdgcn_readfirstlane(soff);
  out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v, s, 0));
}
int main() {
  const unsigned n = 64, nrec = 32 * 4;      // the buffer descriptor covers the first 32 floats; 64 are allocated and filled
  float h[64], *d, *o, r[8];
  for (unsigned i = 0; i < n; ++i) h[i] = 100.f + i;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  for (unsigned soff : {0u, 16u, 120u, 124u, 128u, 256u}) {
    k<<<1, 8>>>(d, o, nrec, soff);
    hipMemcpy(r, o, 8 * 4, hipMemcpyDeviceToHost);
    printf("num_records %u, soffset %3u | voffset 0: %g  4: %g  nrec-8: %g  nrec-4: %g  nrec: %g  nrec+4: %g  0xFFFFFFF0: %g  0xFFFFFFFF: %g\n", nrec, soff,
           r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
  }
  return 0;
}
