// probe: what ds_swizzle(SWAP,16) and v_permlane32_swap deliver per lane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float x = (float)l;
  float a = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), (0x10 << 10) | 0x1f));
  unsigned u = __builtin_bit_cast(unsigned, x), cpy;
  asm volatile("v_mov_b32 %0, %1" : "=v"(cpy) : "v"(u));
  auto r = __builtin_amdgcn_permlane32_swap(u, cpy, false, false);
  out[l] = a; out[64 + l] = __builtin_bit_cast(float, r[0]); out[128 + l] = __builtin_bit_cast(float, r[1]);
}
int main() {
  float* d; hipMalloc(&d, 192 * 4); k<<<1, 64>>>(d);
  float h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 3; i++) { printf("%s:", i == 0 ? "swz16" : i == 1 ? "r0" : "r1"); for (int l = 0; l < 64; l++) printf(" %g", h[i * 64 + l]); printf("\n"); }
  return 0;
}
