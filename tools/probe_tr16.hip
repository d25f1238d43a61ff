// Probe of ds_read_b64_tr_b16 semantics on gfx950 (prints which LDS element each lane/slot receives).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  // test 0: natural addresses (lane l -> elements 4l..4l+3)
  s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
  // test 1: 64-element rows; group g = l>>4 reads rows 8g + (p>>2), cols 4*(p&3) .. +3
  const int g = l >> 4, p = l & 15;
  s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(lds + (g * 8 + (p >> 2)) * 64 + (p & 3) * 4));
  for (int j = 0; j < 4; ++j) { out[l * 4 + j] = r0[j]; out[256 + l * 4 + j] = r1[j]; }
}
int main() {
  short* d; hipMalloc(&d, 512 * sizeof(short));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int t = 0; t < 2; ++t) {
    printf("test %d\n", t);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[t*256+l*4], h[t*256+l*4+1], h[t*256+l*4+2], h[t*256+l*4+3]);
  }
  return 0;
}
