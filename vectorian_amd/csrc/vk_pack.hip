// vk_pack.hip -- corpus upload (normalise, round, tile order) and the per-query table of the static layout.
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// corpus upload: L2-normalise rows (Vectors.normalized, vectorian/embedding/vectors.py:71-86),
// round to bf16 (RNE) and store in tile order.  One wave per row.
// ---------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ float load_elem(const T *p);
template <> __device__ __forceinline__ float load_elem<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load_elem<uint16_t>(const uint16_t *p) {
	return __builtin_bit_cast(float, ((uint32_t)*p) << 16);
}

template <typename T>
__global__ __launch_bounds__(256) void vk_pack_rows_kernel(
	const T *__restrict__ in, int64_t n_rows, int32_t d, int32_t d_pad, int64_t row0,
	uint8_t *__restrict__ tiles, float *__restrict__ mag_out, int32_t normalize, int32_t prec) {

	const int lane = threadIdx.x & 63;
	const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_rows) return;
	const T *row = in + r * (int64_t)d;

	float m = 1.0f;
	if (normalize || mag_out) {
		double acc = 0.0;
		for (int k = lane; k < d; k += 64) {
			const double x = (double)load_elem<T>(row + k);
			acc += x * x;
		}
		for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
		m = (float)sqrt(acc);
		if (m != m) m = 0.0f;
		if (mag_out && lane == 0) mag_out[row0 + r] = m;
	}

	const int64_t grow = row0 + r;
	const int64_t tile = grow >> 4;
	const int i = (int)(grow & 15);
	const int nk32 = d_pad >> 5;
	const int tile_bytes = prec ? d_pad * 64 : d_pad * 32;
	uint8_t *tp = tiles + tile * (int64_t)tile_bytes;
	if (prec) {
		// fp32 tiles (operand order of v_mfma_f32_16x16x4_f32): a block of 16 features is 1 KiB; lane 16 g + i owns
		// row i, features 16 b + 4 s + g for s = 0..3 (element s feeds MFMA step s)
		for (int k = lane; k < d_pad; k += 64) {
			float x = 0.0f;
			if (k < d) {
				x = load_elem<T>(row + k);
				if (normalize) {
					x = x / m;
					if (x != x) x = 0.0f;
				}
			}
			const int b = k >> 4, sidx = (k & 15) >> 2, g = k & 3;
			*reinterpret_cast<float *>(tp + b * 1024 + (g * 16 + i) * 16 + sidx * 4) = x;
		}
		return;
	}

	const int n8 = d_pad >> 3;                  // 8-element chunks; chunk c: K-step c>>2, lane group c&3
	(void)nk32;
	for (int c = lane; c < n8; c += 64) {
		const int k0 = c * 8;
		const int off = (c >> 2) * 1024 + ((c & 3) * 16 + i) * 16;
		uint16_t v[8];
		for (int j = 0; j < 8; j++) {
			float x = 0.0f;
			if (k0 + j < d) {
				x = load_elem<T>(row + k0 + j);
				if (normalize) {
					x = x / m;
					if (x != x) x = 0.0f;
				}
			}
			v[j] = f32_to_bf16_rne(x);
		}
		uint4 w;
		w.x = v[0] | ((uint32_t)v[1] << 16); w.y = v[2] | ((uint32_t)v[3] << 16);
		w.z = v[4] | ((uint32_t)v[5] << 16); w.w = v[6] | ((uint32_t)v[7] << 16);
		*reinterpret_cast<uint4 *>(tp + off) = w;
	}
}

// ---------------------------------------------------------------------------
// static layout: per-query similarity table [V_pad x 16] (metric/static.cpp:9-78)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void vk_table_kernel(const uint8_t *__restrict__ etiles, const uint8_t *__restrict__ qtile,
	int32_t n_tiles, int32_t nk32, int32_t tail, int32_t tile_bytes, float *__restrict__ table, int32_t prec) {
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (tile >= n_tiles) return;
	const f32x4 acc = sim_tile_generic(qtile, etiles + (int64_t)tile * tile_bytes, nk32, tail, lane, prec);
	*reinterpret_cast<f32x4 *>(table + ((int64_t)tile * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;
}

// sim[id(t_j)][j] = 1 (metric/static.cpp:58-67); runs after vk_table_kernel
__global__ void vk_table_fix_kernel(float *__restrict__ table, const int32_t *__restrict__ q_ids, int32_t len_t, int32_t V) {
	const int j = threadIdx.x;
	if (j < len_t) {
		const int id = q_ids[j];
		if (id >= 0 && id < V) table[(int64_t)id * 16 + j] = 1.0f;
	}
}

extern "C" hipError_t vk_launch_pack(const void *in, int32_t dtype_bf16, int64_t n_rows, int32_t d, int32_t d_pad, int64_t row0,
	uint8_t *tiles, float *mag_out, int32_t normalize, int32_t prec, hipStream_t stream) {
	if (n_rows <= 0) return hipSuccess;
	const unsigned grid = (unsigned)((n_rows + 3) / 4);
	if (dtype_bf16)
		vk_pack_rows_kernel<uint16_t><<<grid, 256, 0, stream>>>((const uint16_t *)in, n_rows, d, d_pad, row0, tiles, mag_out, normalize, prec);
	else
		vk_pack_rows_kernel<float><<<grid, 256, 0, stream>>>((const float *)in, n_rows, d, d_pad, row0, tiles, mag_out, normalize, prec);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_table(const uint8_t *etiles, const uint8_t *qtile, int32_t n_tiles, int32_t nk32, int32_t tail,
	int32_t tile_bytes, float *table, const int32_t *q_ids, int32_t len_t, int32_t V, int32_t prec, hipStream_t stream) {
	vk_table_kernel<<<(n_tiles + 3) / 4, 256, 0, stream>>>(etiles, qtile, n_tiles, nk32, tail, tile_bytes, table, prec);
	if (q_ids) vk_table_fix_kernel<<<1, 64, 0, stream>>>(table, q_ids, len_t, V);
	return hipGetLastError();
}
