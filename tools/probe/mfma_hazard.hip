// probe: dependent MFMA chain 16x16x32_bf16 -> 16x16x16_bf16 (SrcC = previous vDst), back to back,
// against the same chain with explicit wait states before the opcode change.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NOPS>
__global__ void k(const bf16x8* a, const bf16x8* b, const bf16x4* a4, const bf16x4* b4, f32x4* out) {
  const int l = threadIdx.x;
  bf16x8 A[9], B[9];
  for (int t = 0; t < 9; t++) { A[t] = a[t * 64 + l]; B[t] = b[t * 64 + l]; }
  bf16x4 At = a4[l], Bt = b4[l];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 9; t++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[t], B[t], acc, 0, 0, 0);
  if (NOPS) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n s_nop 15" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(At, Bt, acc, 0, 0, 0);
  out[l] = acc;
}
int main() {
  std::vector<short> a(9 * 64 * 8), b(9 * 64 * 8), a4(64 * 4), b4(64 * 4);
  srand(3);
  auto rb = []() { float f = (float)rand() / RAND_MAX - 0.5f; unsigned u = __builtin_bit_cast(unsigned, f); return (short)(u >> 16); };
  for (auto &x : a) x = rb(); for (auto &x : b) x = rb(); for (auto &x : a4) x = rb(); for (auto &x : b4) x = rb();
  void *da, *db, *da4, *db4; f32x4 *o0, *o1;
  hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&da4, a4.size() * 2); hipMalloc(&db4, b4.size() * 2);
  hipMalloc(&o0, 64 * 16); hipMalloc(&o1, 64 * 16);
  hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(da4, a4.data(), a4.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db4, b4.data(), b4.size() * 2, hipMemcpyHostToDevice);
  k<0><<<1, 64>>>((bf16x8*)da, (bf16x8*)db, (bf16x4*)da4, (bf16x4*)db4, o0);
  k<1><<<1, 64>>>((bf16x8*)da, (bf16x8*)db, (bf16x4*)da4, (bf16x4*)db4, o1);
  std::vector<float> r0(256), r1(256);
  hipMemcpy(r0.data(), o0, 1024, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), o1, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; i++) if (r0[i] != r1[i]) { if (bad < 5) printf("i %d back-to-back %g waited %g\n", i, r0[i], r1[i]); bad++; }
  printf("mismatches %d / 256\n", bad);
  return 0;
}
