// What v_permlane32_swap / v_permlane16_swap do on gfx950, and whether the results may be read at once:
//   hipcc --offload-arch=gfx950 -O3 -o permlane_swap tools/probe/permlane_swap.hip && ./permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(int *o) {
	const int lane = threadIdx.x;
	{   // builtin
		const auto r = __builtin_amdgcn_permlane32_swap((unsigned)(100 + lane), (unsigned)(200 + lane), false, false);
		o[0 * 64 + lane] = r[0]; o[1 * 64 + lane] = r[1];
		const auto q = __builtin_amdgcn_permlane16_swap((unsigned)(100 + lane), (unsigned)(200 + lane), false, false);
		o[2 * 64 + lane] = q[0]; o[3 * 64 + lane] = q[1];
	}
	{   // asm, operands written just before, results read just after, no nops
		int a = o[8 * 64 + lane] + 100 + lane, b = o[8 * 64 + lane] + 200 + lane;
		asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
		o[4 * 64 + lane] = a; o[5 * 64 + lane] = b;
		int c = o[8 * 64 + lane] + 100 + lane, d = o[8 * 64 + lane] + 200 + lane;
		asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
		o[6 * 64 + lane] = c; o[7 * 64 + lane] = d;
	}
}

// the helpers as the kernels use them: wait states before (operands fresh from the VALU) and after (results read at once)
template <int BEFORE, int AFTER>
__device__ int xor32(int x, int lane) {
	int a = x, b = x;
	if constexpr (BEFORE == 0 && AFTER == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	else if constexpr (BEFORE == 1 && AFTER == 0) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	else if constexpr (BEFORE == 0 && AFTER == 1) asm volatile("v_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	else asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	return lane < 32 ? b : a;
}
template <int BEFORE, int AFTER>
__device__ int xor16(int x, int lane) {
	int a = x, b = x;
	if constexpr (BEFORE == 0 && AFTER == 0) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	else if constexpr (BEFORE == 1 && AFTER == 0) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
	else if constexpr (BEFORE == 0 && AFTER == 1) asm volatile("v_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	return (lane & 16) ? a : b;
}

template <int BEFORE, int AFTER>
__global__ void k2(int *o) {
	const int lane = threadIdx.x;
	int bad = 0;
	int x = o[lane] + lane;   // (o is zero)
	for (int i = 0; i < 1000; i++) {
		x = x * 3 + 1;                                    // fresh VALU result
		const int y = xor32<BEFORE, AFTER>(x, lane) + 5;  // consumed at once
		const int want = __shfl_xor(x, 32, 64) + 5;
		bad += y != want;
		x = x + 7;
		const int z = xor16<BEFORE, AFTER>(x, lane) ^ 3;
		const int want2 = __shfl_xor(x, 16, 64) ^ 3;
		bad += z != want2;
		x += y & 1;
	}
	o[lane] = bad;
}

template <int BEFORE, int AFTER>
static void run2(int *d) {
	int h[64];
	hipMemset(d, 0, sizeof h);
	k2<BEFORE, AFTER><<<1, 64>>>(d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	int bad = 0;
	for (int l = 0; l < 64; l++) bad += h[l];
	printf("lane ^ 32 / lane ^ 16 through the swaps, s_nop 1 before: %d after: %d -> %d mismatches of 128000\n", BEFORE, AFTER, bad);
}

int main() {
	int *d, h[9 * 64];
	hipMalloc(&d, sizeof h);
	hipMemset(d, 0, sizeof h);
	k<<<1, 64>>>(d);
	hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	const char *names[8] = {"builtin32 [0] (vdst)", "builtin32 [1] (src0)", "builtin16 [0]", "builtin16 [1]", "asm32 vdst", "asm32 src0", "asm16 vdst", "asm16 src0"};
	for (int r = 0; r < 8; r++) {
		printf("%-22s", names[r]);
		for (int l = 0; l < 64; l += 8) printf(" %d:%d", l, h[r * 64 + l]);
		printf("\n");
	}
	run2<0, 0>(d); run2<1, 0>(d); run2<0, 1>(d); run2<1, 1>(d);
	return 0;
}
