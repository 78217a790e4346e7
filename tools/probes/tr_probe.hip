// Probe of ds_read_b64_tr_b16 (gfx950): every lane l passes the address of its own 8-byte chunk l (LDS halves hold their own
// index), the output says which (source lane, element) each (lane, j) received.   hipcc --offload-arch=gfx950 -o tr_probe tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s4*)((__attribute__((address_space(3))) char*)lds + threadIdx.x * 8));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf("  (src lane %2d, e %d)", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
    printf("\n");
  }
  return 0;
}
