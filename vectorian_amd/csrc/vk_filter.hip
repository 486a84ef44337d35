// vk_filter.hip -- token filters (TokenFilter, vectorian/core/cpp/query.h:8-28; FilteredSliceFactory,
// slice/static.h:366-416): a filtered corpus is the source corpus with the tokens that do not pass removed and the
// slices re-indexed, built on the device once per distinct filter.  Every scoring kernel then runs on it unchanged
// (the reference compacts each slice again for every query and document).
#include "vk_common.hip.h"

#include <hipcub/hipcub.hpp>

// pass(t) = !((pos_mask >> t.pos) & 1 || (tag_mask >> t.tag) & 1)   (query.h:21-27)
__global__ __launch_bounds__(256) void vk_filter_keep_kernel(const int8_t *__restrict__ pos, const int8_t *__restrict__ tag,
	uint64_t pos_mask, uint64_t tag_mask, int64_t n, int32_t *__restrict__ keep) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t > n) return;
	int k = 0;
	if (t < n) {
		const int p = pos ? pos[t] : 0, g = tag ? tag[t] : 0;
		const bool drop = ((unsigned)p < 64u && ((pos_mask >> p) & 1)) || ((unsigned)g < 64u && ((tag_mask >> g) & 1));   // codes 0..63 have a bit
		k = drop ? 0 : 1;
	}
	keep[t] = k;   // keep[n] = 0: the scan runs over n + 1 items so that new_index[n] is the number of tokens kept
}

__global__ __launch_bounds__(256) void vk_filter_scatter_kernel(const int32_t *__restrict__ keep, const int32_t *__restrict__ new_index,
	int64_t n, int32_t *__restrict__ src_of) {
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (t < n && keep[t]) src_of[new_index[t]] = (int32_t)t;
}

// slice table: start / end in token units -> positions in the filtered token stream
__global__ __launch_bounds__(256) void vk_filter_slices_kernel(const int32_t *__restrict__ start, const int32_t *__restrict__ end,
	const int32_t *__restrict__ new_index, int64_t n_entries, int32_t *__restrict__ out_start, int32_t *__restrict__ out_end) {
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= n_entries) return;
	out_start[e] = new_index[start[e]];
	out_end[e] = new_index[end[e]];
}

// token rows in tile order: a tile is a sequence of 256-byte slabs, slab q holding 16 bytes of each of its 16 rows
// (vk_pack_rows_kernel).  One workgroup per destination tile; 16 consecutive threads write one slab.
__global__ __launch_bounds__(256) void vk_filter_rows_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
	const int32_t *__restrict__ src_of, int64_t n_kept, int32_t tile_bytes) {
	const int64_t tile = blockIdx.x;
	const int i = threadIdx.x & 15;
	const int64_t j = tile * 16 + i;
	if (j >= n_kept) return;
	const int64_t s = src_of[j];
	const uint8_t *sp = src + (s >> 4) * (int64_t)tile_bytes + (s & 15) * 16;
	uint8_t *dp = dst + tile * (int64_t)tile_bytes + i * 16;
	const int n_slabs = tile_bytes >> 8;
	for (int q = threadIdx.x >> 4; q < n_slabs; q += 16)
		*reinterpret_cast<uint4 *>(dp + q * 256) = *reinterpret_cast<const uint4 *>(sp + q * 256);
}

template <typename T>
__global__ __launch_bounds__(256) void vk_filter_gather_kernel(const T *__restrict__ src, T *__restrict__ dst,
	const int32_t *__restrict__ src_of, int64_t n_kept) {
	const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (j < n_kept) dst[j] = src[src_of[j]];
}

static inline unsigned blocks_of(int64_t n) { return (unsigned)((n + 255) / 256); }

// keep flags and their exclusive scan over n + 1 items: new_index[t] = tokens kept before t
extern "C" hipError_t vk_launch_filter_scan(const int8_t *pos, const int8_t *tag, uint64_t pos_mask, uint64_t tag_mask, int64_t n,
	int32_t *keep, int32_t *new_index, void *temp, size_t *temp_bytes, hipStream_t stream) {
	if (!temp) return hipcub::DeviceScan::ExclusiveSum(nullptr, *temp_bytes, keep, new_index, (int)(n + 1), stream);
	vk_filter_keep_kernel<<<blocks_of(n + 1), 256, 0, stream>>>(pos, tag, pos_mask, tag_mask, n, keep);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	return hipcub::DeviceScan::ExclusiveSum(temp, *temp_bytes, keep, new_index, (int)(n + 1), stream);
}

extern "C" hipError_t vk_launch_filter_maps(const int32_t *keep, const int32_t *new_index, int64_t n, int32_t *src_of,
	const int32_t *start, const int32_t *end, int64_t n_entries, int32_t *out_start, int32_t *out_end, hipStream_t stream) {
	if (n > 0) vk_filter_scatter_kernel<<<blocks_of(n), 256, 0, stream>>>(keep, new_index, n, src_of);
	if (n_entries > 0) vk_filter_slices_kernel<<<blocks_of(n_entries), 256, 0, stream>>>(start, end, new_index, n_entries, out_start, out_end);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_filter_rows(const uint8_t *src, uint8_t *dst, const int32_t *src_of, int64_t n_kept, int32_t tile_bytes,
	hipStream_t stream) {
	if (n_kept > 0) vk_filter_rows_kernel<<<(unsigned)((n_kept + 15) / 16), 256, 0, stream>>>(src, dst, src_of, n_kept, tile_bytes);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_filter_gather(const void *src, void *dst, int32_t elem_bytes, const int32_t *src_of, int64_t n_kept,
	hipStream_t stream) {
	if (n_kept <= 0) return hipSuccess;
	if (elem_bytes == 4)
		vk_filter_gather_kernel<int32_t><<<blocks_of(n_kept), 256, 0, stream>>>((const int32_t *)src, (int32_t *)dst, src_of, n_kept);
	else
		vk_filter_gather_kernel<int8_t><<<blocks_of(n_kept), 256, 0, stream>>>((const int8_t *)src, (int8_t *)dst, src_of, n_kept);
	return hipGetLastError();
}
