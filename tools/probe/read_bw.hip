// probe: what a kernel that does nothing but read reaches on this GPU -- the ceiling the scoring kernel is measured against in
// DESIGN 5 (8 TB/s is the HBM3E spec; MI355X_MICROARCH.md quotes 6.29 TB/s for a copy).  A buffer of --gb GB is read once per
// launch with 16-byte loads per lane, U loads in flight per lane, W workgroups of 256 threads per CU, temporal / nontemporal;
// every lane folds what it read into one word that is written only if it has an impossible value.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/read_bw tools/probe/read_bw.hip && tools/probe/read_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool RUNS>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *__restrict__ src, size_t n16, unsigned *__restrict__ sink) {
	// RUNS = false: consecutive workgroups read consecutive 4 KB x U chunks (grid stride)
	// RUNS = true: a wave owns a contiguous 16 KB x U run at a time (as the scoring kernel's waves own their groups of slices)
	unsigned acc = 0;
	if (!RUNS) {
		const size_t stride = (size_t)gridDim.x * 256 * U;
		for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base + (size_t)(U - 1) * 256 < n16; base += stride) {
			u32x4 v[U];
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + base + (size_t)u * 256) : src[base + (size_t)u * 256];
#pragma unroll
			for (int u = 0; u < U; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
		}
	} else {
		const int lane = threadIdx.x & 63;
		const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
		const size_t run = (size_t)64 * U * 16;   // 16 batches of U loads of 1 KB
		for (size_t r = wave; (r + 1) * run <= n16; r += nwaves) {
			for (int b = 0; b < 16; b++) {
				const size_t base = r * run + (size_t)b * 64 * U + lane;
				u32x4 v[U];
#pragma unroll
				for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + base + (size_t)u * 64) : src[base + (size_t)u * 64];
#pragma unroll
				for (int u = 0; u < U; u++) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
			}
		}
	}
	if (acc == 0x9e3779b9u) sink[0] = acc;
}

__global__ void fill_random(u32x4 *dst, size_t n16) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
		unsigned x = (unsigned)i * 2654435761u + 12345u;
		u32x4 v;
		for (int k = 0; k < 4; k++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[k] = x; }
		dst[i] = v;
	}
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// the same runs, every 1 KB batch fed to one v_mfma_f32_16x16x32_bf16 (what the similarity tiles do with the rows), SPIN extra
// dependent VALU operations per batch standing in for a DP
// WRITES: 16 lanes store 64 bytes per batch to each of two arrays (the scores of 16 slices), as vk_span_kernel does
template <int U, int SPIN, int WRITES = 0>
__global__ __launch_bounds__(256) void read_mfma_kernel(const u32x4 *__restrict__ src, size_t n16, unsigned *__restrict__ sink, float *__restrict__ out = nullptr) {
	const int lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
	const size_t run = (size_t)64 * U * 16;
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
	bf16x8 q;
	for (int i = 0; i < 8; i++) q[i] = (__bf16)(0.01f * (lane + i));
	float spin = 0.0f, keep0 = 0.0f, keep1 = 0.0f;
	float k4a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, k4b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
	for (size_t r = wave; (r + 1) * run <= n16; r += nwaves) {
		for (int b = 0; b < 16; b++) {
			const size_t base = r * run + (size_t)b * 64 * U + lane;
			u32x4 v[U];
#pragma unroll
			for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(src + base + (size_t)u * 64);
#pragma unroll
			for (int u = 0; u < U; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, __builtin_bit_cast(bf16x8, v[u]), acc, 0, 0, 0);
#pragma unroll
			for (int k = 0; k < SPIN; k++) spin = __builtin_fmaf(spin, 1.0001f, acc[k & 3]);
			// WRITES 1 / 3: 64 bytes per batch to each of two arrays (3: nontemporal); 2 / 4: 256 bytes per four batches (4: nontemporal)
			if ((WRITES == 1 || WRITES == 3) && lane < 16) {
				const size_t idx = (r * 16 + b) * 16 + lane;
				if (WRITES == 1) { out[idx] = acc[0]; out[(n16 / (64 * U)) * 16 + idx] = acc[1]; }
				else { __builtin_nontemporal_store(acc[0], out + idx); __builtin_nontemporal_store(acc[1], out + (n16 / (64 * U)) * 16 + idx); }
				acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			}
			// WRITES 6: what vk_score_kernel writes -- 4 lanes x 4 bytes to each of two arrays per 8 batches (80 KB read)
			if (WRITES == 6) {
				if ((b & 7) == 7 && (lane & 15) == 15) {
					const size_t idx = (r * 2 + (b >> 3)) * 4 + (lane >> 4);
					out[idx] = acc[0];
					out[(n16 / (64 * U)) * 16 + idx] = acc[1];
				}
			}
			// WRITES 5: once per run of 16 batches, 1 KB per array (four values per lane, nontemporal)
			if (WRITES == 5) {
				if ((lane >> 4) == (b & 3)) { k4a[b >> 2] = acc[0]; k4b[b >> 2] = acc[1]; }
				if (b == 15) {
#pragma unroll
					for (int g = 0; g < 4; g++) {
						const size_t idx = (r * 16 + g * 4) * 16 + lane;
						__builtin_nontemporal_store(k4a[g], out + idx);
						__builtin_nontemporal_store(k4b[g], out + (n16 / (64 * U)) * 16 + idx);
					}
				}
				acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			}
			if (WRITES == 2 || WRITES == 4) {
				if ((lane >> 4) == (b & 3)) { keep0 = acc[0]; keep1 = acc[1]; }
				if ((b & 3) == 3) {
					const size_t idx = (r * 16 + (b & ~3)) * 16 + lane;
					if (WRITES == 2) { out[idx] = keep0; out[(n16 / (64 * U)) * 16 + idx] = keep1; }
					else { __builtin_nontemporal_store(keep0, out + idx); __builtin_nontemporal_store(keep1, out + (n16 / (64 * U)) * 16 + idx); }
				}
				acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			}
		}
	}
	if (acc[0] + spin == 12345.678f) sink[0] = 1;
}

// the 300-d tile: 9728 bytes = nine full 1 KB K-steps and a half one (lanes 0..31), tiles back to back; a wave takes runs of 8
// tiles (a group of four 32-token slices), then SPIN dependent operations (the DP), 16 B x 2 written per run
template <int SPIN>
__global__ __launch_bounds__(256) void read_tiles_kernel(const unsigned char *__restrict__ src, size_t n_tiles, unsigned *__restrict__ sink, float *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
	bf16x8 q;
	for (int i = 0; i < 8; i++) q[i] = (__bf16)(0.01f * (lane + i));
	float spin = 0.0f;
	for (size_t r = wave; (r + 1) * 8 <= n_tiles; r += nwaves) {
		f32x4 tot = {0.0f, 0.0f, 0.0f, 0.0f};
		for (int b = 0; b < 8; b++) {
			const unsigned char *tp = src + (r * 8 + b) * 9728;
			u32x4 v[10];
#pragma unroll
			for (int u = 0; u < 9; u++) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tp + u * 1024 + lane * 16));
			v[9] = u32x4{0, 0, 0, 0};
			if (lane < 32) v[9] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(tp + 9 * 1024 + lane * 16));
			f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
			for (int u = 0; u < 10; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, __builtin_bit_cast(bf16x8, v[u]), acc, 0, 0, 0);
			tot += acc;
		}
#pragma unroll 16
		for (int k = 0; k < SPIN; k++) spin = __builtin_fmaf(spin, 1.0001f, tot[k & 3]);
		if ((lane & 15) == 15) {
			const size_t idx = r * 4 + (lane >> 4);
			out[idx] = tot[0] + spin;
			out[n_tiles + idx] = tot[1];
		}
	}
	if (spin == 12345.678f) sink[0] = 1;
}

template <int SPIN>
static double run_tiles(const u32x4 *src, size_t n16, unsigned *sink, int wg_per_cu, int cus, int reps, float *out) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const int grid = cus * wg_per_cu;
	const size_t n_tiles = n16 * 16 / 9728 / 8 * 8;
	read_tiles_kernel<SPIN><<<grid, 256>>>(reinterpret_cast<const unsigned char *>(src), n_tiles, sink, out);
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int i = 0; i < reps; i++) {
		hipEventRecord(a);
		read_tiles_kernel<SPIN><<<grid, 256>>>(reinterpret_cast<const unsigned char *>(src), n_tiles, sink, out);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	return (double)n_tiles * 9728 / (best * 1e-3) / 1e12;
}

template <int U, int SPIN, int WRITES = 0>
static double run_mfma(const u32x4 *src, size_t n16, unsigned *sink, int wg_per_cu, int cus, int reps, float *out = nullptr) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const int grid = cus * wg_per_cu;
	read_mfma_kernel<U, SPIN, WRITES><<<grid, 256>>>(src, n16, sink, out);
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int i = 0; i < reps; i++) {
		hipEventRecord(a);
		read_mfma_kernel<U, SPIN, WRITES><<<grid, 256>>>(src, n16, sink, out);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	return (double)n16 * 16 / (best * 1e-3) / 1e12;
}

template <int U, bool NT, bool RUNS>
static double run(const u32x4 *src, size_t n16, unsigned *sink, int wg_per_cu, int cus, int reps) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	const int grid = cus * wg_per_cu;
	read_kernel<U, NT, RUNS><<<grid, 256>>>(src, n16, sink);
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int i = 0; i < reps; i++) {
		hipEventRecord(a);
		read_kernel<U, NT, RUNS><<<grid, 256>>>(src, n16, sink);
		hipEventRecord(b);
		hipEventSynchronize(b);
		float ms = 0; hipEventElapsedTime(&ms, a, b);
		if (ms < best) best = ms;
	}
	return (double)n16 * 16 / (best * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
	double gb = 19.2;
	if (argc > 1) gb = atof(argv[1]);
	const size_t n16 = (size_t)(gb * 1e9 / 16) / (64 * 16 * 16) * (64 * 16 * 16);
	u32x4 *src; unsigned *sink;
	if (hipMalloc(&src, n16 * 16) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
	if (argc > 2 && atoi(argv[2]) == 0) hipMemset(src, 1, n16 * 16);   // constant bytes
	else fill_random<<<4096, 256>>>(src, n16);                          // pseudo-random words, as real rows are
	hipDeviceSynchronize();
	float *out; hipMalloc(&out, (n16 / 640 + 64) * 16 * 2 * 4);
	int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	printf("{\"probe\": \"read_bw\", \"GB\": %.2f, \"cus\": %d, \"data\": \"%s\", \"TBps\": {", (double)n16 * 16 / 1e9, cus, (argc > 2 && atoi(argv[2]) == 0) ? "constant" : "random");
	bool first = true;
	for (int w : {2, 3, 4}) {
#define ROW(U, NT, RUNS, name) { double t = run<U, NT, RUNS>(src, n16, sink, w, cus, 5); printf("%s\"%s_w%d\": %.3f", first ? "" : ", ", name, w, t); first = false; }
		ROW(4, true, false, "stride_u4_nt") ROW(8, true, false, "stride_u8_nt") ROW(4, false, false, "stride_u4") ROW(8, false, false, "stride_u8")
		{ double t = run_mfma<10, 0>(src, n16, sink, w, cus, 5); printf(", \"mfma_u10_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 1>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w64B_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 3>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w64Bnt_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 2>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w256B_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 4>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w256Bnt_w%d\": %.3f", w, t); }
		{ double t = run_tiles<0>(src, n16, sink, w, cus, 5, out); printf(", \"tiles9728_w%d\": %.3f", w, t); }
		{ double t = run_tiles<512>(src, n16, sink, w, cus, 5, out); printf(", \"tiles9728_spin512_w%d\": %.3f", w, t); }
		{ double t = run_tiles<2048>(src, n16, sink, w, cus, 5, out); printf(", \"tiles9728_spin2048_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 6>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w16B_per_80KB_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 0, 5>(src, n16, sink, w, cus, 5, out); printf(", \"mfma_u10_w1KBnt_per_run_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 64>(src, n16, sink, w, cus, 5); printf(", \"mfma_u10_spin64_w%d\": %.3f", w, t); }
		{ double t = run_mfma<10, 256>(src, n16, sink, w, cus, 5); printf(", \"mfma_u10_spin256_w%d\": %.3f", w, t); }
		ROW(4, true, true, "runs_u4_nt") ROW(8, true, true, "runs_u8_nt") ROW(10, true, true, "runs_u10_nt") ROW(8, false, true, "runs_u8")
	}
	printf("}}\n");
	return 0;
}
