// vk_rwmd_batch.hip -- batched relaxed WMD as an MFMA GEMM (BASELINE config 4).
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// Batched relaxed Word Mover's Distance (BASELINE config 4: 256 queries x 1M sentences):
// a GEMM [T x d] . [d x (B * 16)] on MFMA with the row / column minima and the RWMD score as
// epilogue.  MFMA-bound (intensity ~2.5 kFLOP per corpus byte), so the corpus tokens stay in
// registers and the queries stream past them:
//   workgroup = 4 waves; each wave loads TPW token tiles (64 tokens for TPW = 4) ONCE into
//   registers as MFMA B operands; the B query tiles (A operands, one 16-row tile per query) are
//   staged one after the other into a double-buffered LDS slot shared by the 4 waves
//   (global -> registers -> LDS while the previous query's MFMAs run), so every query byte is
//   fetched from L2 once per 256 tokens and every corpus byte from HBM once per batch.
// Requires sentences of one length L = 16 * TPS (the config's shape); other corpora take the
// per-query path.  Scores: scores[q * n_sent + s] = Score::value as in rwmd_rows.
// ---------------------------------------------------------------------------

__device__ __forceinline__ float xor16_f(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), (0x10 << 10) | 0x1f));
}

// DPP row broadcast used by wave reductions: CTRL 0x142 = row_bcast:15 (lane 15 of each row to the next
// row), 0x143 = row_bcast:31; ROWS = row_mask of the rows that receive.  Lanes outside get 0.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_bcast(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, false));
}

// partner lane l ^ 32 through ds_bpermute.  (v_permlane32_swap would be cheaper, but the builtin's
// second result did not deliver the upper halves here -- tools/probe/xlane_probe.hip -- so it is not used.)
__device__ __forceinline__ float xor32_f(float x, int lane) {
	return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
}

// Ragged corpora take the same kernel once per length bucket of a padded copy of the corpus (vk_batch.cpp: sentences of 1..16,
// 17..32, 33..48, 49..64 tokens, each padded with zero rows to 16 TPS tokens): p.sent_len masks the padding out of the sums over
// a sentence's tokens (a zero row gives D = 1: it never lowers a column minimum), p.sent_id sends the score to the sentence's own
// place.  NK > 12 (768-d rows): one wave per SIMD -- the 64 tokens of a wave then take 384 registers.
template <int NK, bool HALF, int TPS>
__global__ __launch_bounds__(256, NK > 12 ? 1 : 2) void vk_rwmd_batch_kernel(VkRwmdBatchParams p) {
	constexpr int TPW = TPS == 3 ? 3 : 4;          // token tiles per wave
	constexpr int SPW = TPW / TPS;                 // sentences per wave
	extern __shared__ float4 vk_smem4[];
	uint8_t *qbuf = reinterpret_cast<uint8_t *>(vk_smem4);   // 2 x tile_bytes
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int g4 = lane >> 4;
	const int n16 = p.tile_bytes >> 4;             // 16-byte pieces of one query tile
	const int64_t n_chunks = (p.n_tiles + TPW * 4 - 1) / (TPW * 4);
	const int64_t stride = p.score_stride > 0 ? p.score_stride : p.n_sent;

	for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const int64_t tile0 = chunk * (TPW * 4) + (int64_t)wv * TPW;
		// ---- corpus tiles of this wave -> registers (read once per batch)
		bf16x8 x[TPW][NK];
#pragma unroll
		for (int tt = 0; tt < TPW; tt++) {
			const int64_t tile = tile0 + tt < p.n_tiles ? tile0 + tt : p.n_tiles;   // one zero tile follows the corpus
			const uint8_t *tp = p.tiles + tile * p.tile_bytes;
#pragma unroll
			for (int t = 0; t < NK; t++) {
				if (HALF && t == NK - 1) x[tt][t] = load_half_block(tp + t * 1024, lane, true);
				else x[tt][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + t * 1024 + lane * 16));
			}
		}
		// this wave's sentences: length (ragged corpora) and where their scores go
		int slen[SPW];
		int64_t sout[SPW];
#pragma unroll
		for (int sw = 0; sw < SPW; sw++) {
			const int64_t sent = (tile0 + sw * TPS) / TPS;
			const bool ok = sent < p.n_sent;
			slen[sw] = ok ? (p.sent_len ? p.sent_len[sent] : TPS * 16) : 0;
			sout[sw] = ok ? (p.sent_id ? (int64_t)p.sent_id[sent] : sent) : -1;
		}
		// ---- stage query 0
		__syncthreads();   // previous chunk's readers are done with the LDS slots
		for (int i = threadIdx.x; i < n16; i += 256)
			vk_smem4[i] = *reinterpret_cast<const float4 *>(p.qtiles + i * 16);
		__syncthreads();

		for (int q = 0; q < p.n_queries; q++) {
			const uint8_t *cur = qbuf + (q & 1) * p.tile_bytes;
			float4 *nxt = vk_smem4 + ((q + 1) & 1) * n16;
			// prefetch the next query tile into registers (<= 3 pieces per thread for d <= 384, 6 for d = 768)
			// (named registers: an array indexed in two separately unrolled loops stays in scratch with hipcc)
			constexpr bool WIDE = NK > 12;
			const float4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
			float4 st0 = z4, st1 = z4, st2 = z4, st3 = z4, st4 = z4, st5 = z4;
			const bool more = q + 1 < p.n_queries;
			if (more) {
				const uint8_t *src = p.qtiles + (int64_t)(q + 1) * p.tile_bytes;
				const int i0 = threadIdx.x;
				auto piece = [&](int i) { return *reinterpret_cast<const float4 *>(src + (i < n16 ? i : 0) * 16); };   // clamped: no branch around a load
				st0 = piece(i0); st1 = piece(i0 + 256); st2 = piece(i0 + 512);
				if (WIDE) { st3 = piece(i0 + 768); st4 = piece(i0 + 1024); st5 = piece(i0 + 1280); }
			}
			// ---- S^T = Q X^T for TPW tiles; query fragments DEPTH steps ahead of their MFMAs
			f32x4 acc[TPW];
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) acc[tt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			// query fragments: the ds_read_b128 of a chunk of K-steps (all of them for d <= 384: 40 VGPRs at d = 300) are issued up
			// front, so that the MFMAs of step t never wait for the LDS latency of step t
			constexpr int KC = NK > 12 ? 12 : NK;          // K-steps per chunk
#pragma unroll
			for (int t0 = 0; t0 < NK; t0 += KC) {
				bf16x8 af[KC];
#pragma unroll
				for (int t = 0; t < KC; t++) {
					if (t0 + t < NK) {
						af[t] = *reinterpret_cast<const bf16x8 *>(cur + (t0 + t) * 1024 + ((HALF && t0 + t == NK - 1) ? (lane & 31) : lane) * 16);
						if (HALF && t0 + t == NK - 1) {
							const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
							af[t] = lane < 32 ? af[t] : z;
						}
					}
				}
#pragma unroll
				for (int t = 0; t < KC; t++) {
					if (t0 + t < NK) {
#pragma unroll
						for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], x[tt][t0 + t], acc[tt], 0, 0, 0);
					}
				}
			}
			if (more) {
				const int i0 = threadIdx.x;
				if (i0 < n16) nxt[i0] = st0;
				if (i0 + 256 < n16) nxt[i0 + 256] = st1;
				if (i0 + 512 < n16) nxt[i0 + 512] = st2;
				if (WIDE) {
					if (i0 + 768 < n16) nxt[i0 + 768] = st3;
					if (i0 + 1024 < n16) nxt[i0 + 1024] = st4;
					if (i0 + 1280 < n16) nxt[i0 + 1280] = st5;
				}
			}
			// ---- epilogue: D = 1 - clip(S); lane holds token (lane & 15) x query columns 4*g4 .. +3.
			// Lane exchanges across the four 16-lane rows go through the LDS crossbar (ds_swizzle /
			// ds_bpermute); the exchanges of all tiles are issued back to back so their latencies overlap.
			// (reciprocals instead of the oracle's divisions: a batch epilogue runs per (query, sentence) and
			// is VALU-bound; the results differ from the per-query kernel by <= 1 ulp, far inside 1e-4)
			const int len_t = p.q_len[q];
			const float inv_t = 1.0f / (float)len_t;
			float rm[TPW], cm[SPW][4];
#pragma unroll
			for (int sw = 0; sw < SPW; sw++)
#pragma unroll
				for (int r = 0; r < 4; r++) cm[sw][r] = 3.0f;
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) {
				const f32x4 a4 = acc[tt];
				float dd[4];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					dd[r] = 1.0f - clip01(a4[r]);
					cm[tt / TPS][r] = fminf(cm[tt / TPS][r], dd[r]);
				}
				// padded query columns have S = 0, D = 1: they never lower a minimum
				rm[tt] = fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3]));
			}
			float ex[TPW];
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) ex[tt] = xor16_f(rm[tt]);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) rm[tt] = fminf(rm[tt], ex[tt]);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) ex[tt] = xor32_f(rm[tt], lane);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) rm[tt] = fminf(rm[tt], ex[tt]);   // row minimum of token (lane & 15) of tile tt
#pragma unroll
			for (int sw = 0; sw < SPW; sw++) {
				float rsum = 0.0f;
#pragma unroll
				for (int ts = 0; ts < TPS; ts++) rsum += (ts * 16 + (lane & 15) < slen[sw]) ? rm[sw * TPS + ts] : 0.0f;   // padding rows stay out
				// column minima over the sentence's tokens: reduce over the 16 lanes of the DPP row
#pragma unroll
				for (int r = 0; r < 4; r++) {
					float x = cm[sw][r];
					x = fminf(x, dpp_f<DPP_ROW_SHR1>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR2>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR4>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR8>(x, x));
					cm[sw][r] = x;
				}
				float c0 = 0.0f;
#pragma unroll
				for (int r = 0; r < 4; r++) c0 += (4 * g4 + r < len_t) ? cm[sw][r] : 0.0f;   // valid in lane 15 of each row
				// sum of the four rows' lane 15 -> lane 63 (row_bcast:15 into rows 1, 3; row_bcast:31 into rows 2, 3)
				c0 += dpp_bcast<0x142, 0xa>(c0);
				c0 += dpp_bcast<0x143, 0xc>(c0);
				rsum += dpp_f<DPP_ROW_SHR1>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR2>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR4>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR8>(0.0f, rsum);   // lane 15 of every row: sum over the sentence's tokens
				const float inv_s = 1.0f / (float)(slen[sw] > 0 ? slen[sw] : 1);
				const float acc0 = inv_t * c0, acc1 = inv_s * rsum;      // nbow and bow/len agree up to rounding
				const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(acc0, acc1)) : acc0;
				const float raw = p.nbow ? 1.0f - cost : ((float)len_t - cost) * inv_t;
				if (lane == 63 && sout[sw] >= 0) {
					const float boost = p.boost ? p.boost[sout[sw]] : 1.0f;
					p.scores[(int64_t)q * stride + sout[sw]] = (raw * inv_t) * boost;
				}
			}
			__syncthreads();   // next query tile is in place; this one may be overwritten
		}
	}
}

// padded copy of a length bucket: destination row j = token j % (16 tps) of the bucket's sentence j / (16 tps), a zero row
// beyond the sentence's length.  One 16-lane group per row, the row's 16-byte pieces (4 per K-step block) spread over 16 groups.
__global__ __launch_bounds__(256) void vk_batch_pack_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int32_t *__restrict__ ids,
	const int32_t *__restrict__ sent_start, const int32_t *__restrict__ sent_end, int64_t n, int32_t tps, int32_t tile_bytes) {
	const int64_t tile = blockIdx.x;
	const int i = threadIdx.x & 15;
	const int64_t j = tile * 16 + i;
	const int64_t sb = j / (16 * tps);
	if (sb >= n) return;
	const int t = (int)(j - sb * 16 * tps);
	const int32_t id = ids[sb];
	const int64_t s = (int64_t)sent_start[id] + t;
	const bool real = s < sent_end[id];
	const uint8_t *sp = src + (s >> 4) * (int64_t)tile_bytes + (s & 15) * 16;
	uint8_t *dp = dst + tile * (int64_t)tile_bytes + i * 16;
	const int n_slabs = tile_bytes >> 8;
	const uint4 zero = {0u, 0u, 0u, 0u};
	for (int q = threadIdx.x >> 4; q < n_slabs; q += 16)
		*reinterpret_cast<uint4 *>(dp + q * 256) = real ? *reinterpret_cast<const uint4 *>(sp + q * 256) : zero;
}

extern "C" hipError_t vk_launch_batch_pack(const uint8_t *src_tiles, uint8_t *dst_tiles, const int32_t *ids, const int32_t *sent_start, const int32_t *sent_end,
	int64_t n, int32_t tps, int32_t tile_bytes, hipStream_t stream) {
	if (n > 0) vk_batch_pack_kernel<<<(unsigned)(n * tps), 256, 0, stream>>>(src_tiles, dst_tiles, ids, sent_start, sent_end, n, tps, tile_bytes);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Batched RWMD for 32-token sentences (the shape of BASELINE config 4) on v_mfma_f32_32x32x16_bf16.
// Two inefficiencies of the 16-row kernel above go away: a 10-token query no longer occupies a
// 16-row tile (QPT = 3 queries share the 32 rows of one A tile: 30 / 32 rows used, 10 / 16 before)
// and K is padded to 16, not 32 (d = 300: 19 steps of 16 = 304, not 320).
//   workgroup = 8 waves; a wave keeps its 2 sentences (4 token tiles) in registers as B operands of
//   two MFMA chains: chain 0 takes tokens 0..15 of both sentences (columns 0..15 = sentence 0,
//   16..31 = sentence 1), chain 1 tokens 16..31.  So a lane (n = lane & 31, h = lane >> 5) holds, in
//   acc0[i] and acc1[i], the similarities of ONE query row with tokens n & 15 and 16 + (n & 15) of
//   sentence n >> 4: the maximum over a sentence's tokens is one in-lane max and a reduction over the
//   16 lanes of a DPP row, never across rows.
//   A rows: M = 8 (i >> 2) + 4 h + (i & 3) for accumulator register i of half h (hardware layout of
//   the 32x32 result).  QPT = 3: half h, i < 10 = token i of query 3 qt + h; 10 <= i < 15 = token
//   5 h + i - 10 of query 3 qt + 2.  QPT = 2: half h = query 2 qt + h, i = token.  The host packs the
//   A tiles accordingly (vk_query_batch in vk_batch.cpp).
//   The query tiles stream through a double-buffered LDS slot shared by the 8 waves.
// ---------------------------------------------------------------------------

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8n;

__device__ __forceinline__ float row_sum_to_lane15(float x) {
	x += dpp_f<DPP_ROW_SHR1>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR2>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR4>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR8>(0.0f, x);
	return x;
}

// Maxima of similarities that end in clip01() may be taken on the raw bit patterns as signed integers:
// non-negative floats order like their bits, negative floats are negative integers and lose against any
// non-negative one, and a maximum that stays negative is clipped to 0 whichever negative value it is.
// Integer maxima need no canonicalisation of their inputs and fuse with DPP (v_max_i32_dpp).
__device__ __forceinline__ int fbits(float x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
template <int CTRL>
__device__ __forceinline__ int dpp_imax(int x) {
	return imax(x, __builtin_amdgcn_update_dpp((int)0x80000000, x, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ int row_imax_to_lane15(int x) {
	x = dpp_imax<DPP_ROW_SHR1>(x);
	x = dpp_imax<DPP_ROW_SHR2>(x);
	x = dpp_imax<DPP_ROW_SHR4>(x);
	x = dpp_imax<DPP_ROW_SHR8>(x);
	return x;
}
__device__ __forceinline__ float clip01_bits(int x) { return __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, x), 0.0f, 1.0f); }
// U16: the accumulator registers hold clipped similarities as 16-bit fixed point (0 .. 65535 = 0 .. 1; the static layout's table,
// vk_table_batch_kernel) instead of float bits: the integer maxima are the same code, only the reduced values are converted
template <bool U16>
__device__ __forceinline__ float sim_of_bits(int x) { return U16 ? (float)x * (1.0f / 65535.0f) : clip01_bits(x); }

// Maxima of 16 registers over the 16 lanes of each DPP row, transposed: lane v of the row ends up with the row
// maximum of register v.  Halving exchange: at the step for lane bit b a lane keeps the registers whose index bit
// equals its own lane bit and hands the others to its partner (lane ^ (1 << b)), so the register count halves while
// the lane span doubles: 8 + 4 + 2 + 1 exchanges (47 instructions) instead of 16 four-step DPP reductions (64+).
#ifndef VK_TR_OLD
// Round 4: the same halving exchange, started with the lane bits a DPP write mask can select.  bank_mask enables the four quads
// of a row separately (bank = lane bits 2 and 3), so the steps for lane ^ 4 and lane ^ 8 need no selects at all: two
// v_max_i32_dpp per register pair, each writing only the quads of one parity -- banks 0, 2 take max(v[2k], v[2k] of lane + 4),
// banks 1, 3 take max(v[2k + 1], v[2k + 1] of lane - 4); likewise lane ^ 8 through row_ror:8 with banks {0, 1} / {2, 3}.  Those
// two steps come first, while there are 8 and 4 pairs; the steps inside a quad (lane ^ 1, lane ^ 2; 2 + 1 pairs) take both maxima and
// one select.  16 + 8 + 6 + 3 = 33 vector instructions instead of 24 + 12 + 10 + 3 = 49, and the selects' results no longer feed
// DPP reads (each cost two wait states).  Inline asm: the compiler has no partial-write form of update_dpp + max.
// The register a lane ends up with: bit 0 of its index = lane bit 2, bit 1 = lane bit 3, bit 2 = lane bit 0, bit 3 = lane bit 1.
__device__ __forceinline__ int row_transpose_reg(int lane) {
	return ((lane >> 2) & 1) | (((lane >> 3) & 1) << 1) | ((lane & 1) << 2) | (((lane >> 1) & 1) << 3);
}
__device__ __forceinline__ int row_transpose_imax16(int (&v)[16], int lane) {
	// ONE asm block for both masked steps, IN PLACE (the kernel sits at the VGPR cap): the result of pair (2k, 2k + 1) lands in the
	// register of v[2k] -- banks 0, 2 of it first (they read v[2k] of lane + 4, in banks 1, 3, still untouched), then banks 1, 3 from
	// v[2k + 1]; the second step pairs the results (0, 2), (4, 6), .. into v[0], v[4], v[8], v[12] likewise.  The compiler's hazard
	// recogniser does not look into asm, so the distances are kept by hand -- a DPP read needs two wait states behind the vector
	// write of its source: `s_nop 1` in front (the inputs may have been written just before) and behind (the results are read by DPP
	// next); inside, every source is at least four instructions old.
	asm("s_nop 1\n\t"
		"v_max_i32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %5, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %6, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %7, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
		"v_max_i32_dpp %0, %8, %8 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %2, %10, %10 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %3, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %4, %12, %12 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %5, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %6, %14, %14 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %7, %15, %15 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
		"v_max_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
		"v_max_i32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
		"v_max_i32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
		"v_max_i32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
		"v_max_i32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
		"v_max_i32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
		"v_max_i32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
		"v_max_i32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
		"s_nop 1"
		: "+v"(v[0]), "+v"(v[2]), "+v"(v[4]), "+v"(v[6]), "+v"(v[8]), "+v"(v[10]), "+v"(v[12]), "+v"(v[14])
		: "v"(v[1]), "v"(v[3]), "v"(v[5]), "v"(v[7]), "v"(v[9]), "v"(v[11]), "v"(v[13]), "v"(v[15]));
	const int x[4] = {v[0], v[4], v[8], v[12]};
	const bool b0 = lane & 1, b1 = lane & 2;
	int y[2];
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int u0 = dpp_imax<0xB1>(x[2 * k]), u1 = dpp_imax<0xB1>(x[2 * k + 1]);   // quad_perm [1,0,3,2]: lane ^ 1
		y[k] = b0 ? u1 : u0;
	}
	const int u0 = dpp_imax<0x4E>(y[0]), u1 = dpp_imax<0x4E>(y[1]);                     // quad_perm [2,3,0,1]: lane ^ 2
	return b1 ? u1 : u0;
}
#else
__device__ __forceinline__ int row_transpose_reg(int lane) { return lane & 15; }
__device__ __forceinline__ int row_transpose_imax16(int (&v)[16], int lane) {
	const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
	const int NEG = (int)0x80000000;
	int w[8], x[4], y[2];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const int keep = b0 ? v[2 * k + 1] : v[2 * k], send = b0 ? v[2 * k] : v[2 * k + 1];
		w[k] = imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]: lane ^ 1
	}
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int keep = b1 ? w[2 * k + 1] : w[2 * k], send = b1 ? w[2 * k] : w[2 * k + 1];
		x[k] = imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]: lane ^ 2
	}
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int keep = b2 ? x[2 * k + 1] : x[2 * k], send = b2 ? x[2 * k] : x[2 * k + 1];
		int t = __builtin_amdgcn_update_dpp(NEG, send, 0x104, 0xf, 0x5, false);                // row_shl:4 into banks 0, 2: lane + 4
		t = __builtin_amdgcn_update_dpp(t, send, 0x114, 0xf, 0xa, false);                      // row_shr:4 into banks 1, 3: lane - 4
		y[k] = imax(keep, t);
	}
	const int keep = b3 ? y[1] : y[0], send = b3 ? y[0] : y[1];
	return imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0x128, 0xf, 0xf, false));          // row_ror:8: lane ^ 8
}
#endif

// S tiles of one query tile against the wave's two sentences: 2 x NK16 MFMAs, A fragments DEPTH steps ahead.
// The wave's share of the NEXT query tile (1 KiB pieces wv, wv + 8, wv + 16) leaves as LDS-DMA right after the first MFMA
// pair (global_load_lds_dwordx4: no staging registers -- the kernel sits at the VGPR cap), in flight during this tile's MFMAs
// and retired by the barrier that ends the iteration.  (Fetched into registers after the first MFMA pairs and written with
// ds_write_b128 before the last ones, the same bytes cost the same time and 8 registers: measured, not kept.  Without the
// staging at all the kernel takes 10 % less: what costs is moving 100 GB per batch from L2 into the CUs, not issuing it.)
template <int NK16, int NW = 8>
__device__ __forceinline__ void batch32_mfma(const uint8_t *cur, const bf16x8 (&x)[2][NK16], f32x16 &acc0, f32x16 &acc1,
	const uint8_t *next_src /* wave-uniform */, unsigned lane16, const uint8_t *next_dst, int wv) {
#ifdef VK_DEPTH
	constexpr int DEPTH = VK_DEPTH;
#else
	constexpr int DEPTH = 2;                           // A fragments in flight (ds_read_b128 ahead of their MFMAs)
#endif
#ifdef VK_PRIO
	__builtin_amdgcn_s_setprio(VK_PRIO);
#endif
#pragma unroll
	for (int i = 0; i < 16; i++) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
	bf16x8 a[DEPTH];
#pragma unroll
	for (int t = 0; t < DEPTH && t < NK16; t++) a[t] = *reinterpret_cast<const bf16x8 *>(cur + t * 1024);
#pragma unroll
	for (int t = 0; t < NK16; t++) {
		const bf16x8 at = a[t % DEPTH];
#if defined(VK_ABL) && VK_ABL == 3
		if (t > 0) { acc0[t & 15] += (float)at[0]; acc1[t & 15] += (float)x[1][t][0] + (float)x[0][t][0]; continue; }
#endif
		acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, at), __builtin_bit_cast(bf16x8n, x[0][t]), acc0, 0, 0, 0);
		acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, at), __builtin_bit_cast(bf16x8n, x[1][t]), acc1, 0, 0, 0);
#if defined(VK_ABL) && VK_ABL >= 5
		if (t + DEPTH < NK16) a[t % DEPTH] = a[(t + 1) % DEPTH];
#else
		if (t + DEPTH < NK16) a[t % DEPTH] = *reinterpret_cast<const bf16x8 *>(cur + (t + DEPTH) * 1024);
#endif
#if !defined(VK_ABL) || VK_ABL < 4
		if (t == 0) {
#pragma unroll
			for (int b = wv; b < NK16; b += NW)   // wave-uniform piece base (scalar registers) + the lane's 32-bit offset: no 64-bit vector addresses
				__builtin_amdgcn_global_load_lds((vk_glb_ptr)(next_src + b * 1024 + lane16), (vk_lds_ptr)(next_dst + b * 1024), 16, 0, 0);
		}
#endif
#ifndef VK_NO_SCHED_BARRIER
		__builtin_amdgcn_sched_barrier(0);
#endif
	}
#ifdef VK_PRIO
	__builtin_amdgcn_s_setprio(0);
#endif
}

// RWMD scores of one query tile for the wave's two sentences from the accumulators.
// D = 1 - clip(S) is monotone in S: reduce S (as integers, see above), convert the reduced values only.
// Rows of absent query tokens are zero: S = 0, clip = 0, no masks.
// The parameters of the tile's queries (length as float, its reciprocal; 0 for an absent query) sit in LDS, staged once per
// workgroup: four ds_read_b32 at the top of the epilogue, consumed at its end.  (Fetched from global memory under the
// lane-15 branch, each tile paid the latency of two dependent vector loads inside its epilogue.)
struct B32Vals { float main, third; };
// the lane's sentence: 1 / its real length and the number of padding tokens (ragged corpora run on a padded copy, vk_batch.cpp:
// a zero row has S = 0 exactly, so each padding token adds exactly 1 to the sum of the token distances and nothing to a maximum)
struct B32Sent { float inv_s, pad; };
// W64: the wave holds ONE sentence of 64 (padded) tokens instead of two of 32 -- columns 0..15 of the two MFMA chains are its
// tokens 0..31, columns 16..31 its tokens 32..63; sums over its tokens add the two DPP rows of a lane half (row_bcast:15),
// maxima over its tokens meet through one ds_swizzle (lane ^ 16); its scores sit in lanes 31 and 63.
template <bool W64>
__device__ __forceinline__ float tokens_sum(float x) {
	x = row_sum_to_lane15(x);
	if (W64) x += dpp_bcast<0x142, 0xa>(x);
	return x;
}
constexpr int B32_MAX_QTILES = 256;   // query tiles per launch (their parameters: 8 KiB of LDS); larger batches take several launches

template <int QPT, bool W64, bool U16 = false>
__device__ __forceinline__ B32Vals batch32_epilogue(const VkRwmdBatchParams &p, int lane, const f32x16 &acc0, const f32x16 &acc1,
	const float *tile_param, float boost, const B32Sent &sn) {
	constexpr int NMAIN = QPT == 3 ? 10 : 16;          // rows of the half's own query
	const int h = lane >> 5;
	const float inv_s = sn.inv_s;
	const float len_main = tile_param[h], inv_main = tile_param[4 + h];
	const float len_third = tile_param[2], inv_third = tile_param[6];
#if defined(VK_ABL) && (VK_ABL == 1 || VK_ABL >= 4)
	{
		// no epilogue: the accumulators stay live, nothing is computed from them
#pragma unroll
		for (int i = 0; i < 16; i++) asm volatile("" :: "v"(acc0[i]), "v"(acc1[i]));
		return {0.0f, 0.0f};
	}
#endif
	// (a) per token: max over the query's rows (in-lane) -> this lane's two tokens' distances -> sum
	//     over the sentence's 32 tokens = 16 lanes x 2 chains
	int ca0 = fbits(acc0[0]), ca1 = fbits(acc1[0]);
#pragma unroll
	for (int i = 1; i < NMAIN; i++) { ca0 = imax(ca0, fbits(acc0[i])); ca1 = imax(ca1, fbits(acc1[i])); }
	const float ts_main = tokens_sum<W64>((2.0f - sim_of_bits<U16>(ca0)) - sim_of_bits<U16>(ca1)) - sn.pad;
	float ts_third = 0.0f;
	if (QPT == 3) {
		int cb0 = fbits(acc0[10]), cb1 = fbits(acc1[10]);
#pragma unroll
		for (int i = 11; i < 15; i++) { cb0 = imax(cb0, fbits(acc0[i])); cb1 = imax(cb1, fbits(acc1[i])); }
		const int e0 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, cb0);
		const int e1 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, cb1);
		cb0 = imax(cb0, e0); cb1 = imax(cb1, e1);
		ts_third = tokens_sum<W64>((2.0f - sim_of_bits<U16>(cb0)) - sim_of_bits<U16>(cb1)) - sn.pad;
	}
	// (b) per query row: max over the sentence's tokens -> lane 15 of the DPP row
	float s_main = 0.0f, s_third = 0.0f;
	{
		int m[16];
#pragma unroll
		for (int i = 0; i < 16; i++) m[i] = imax(fbits(acc0[i]), fbits(acc1[i]));
		float z = sim_of_bits<U16>(row_transpose_imax16(m, lane));    // this lane: maximum of query row v = row_transpose_reg(lane) over the sentence
		if (W64) z = fmaxf(z, xor16_f(z));                          // ... over both halves of a 64-token sentence
		const int v = row_transpose_reg(lane);
		s_main = row_sum_to_lane15(v < NMAIN ? z : 0.0f);
		if (QPT == 3) {
			s_third = row_sum_to_lane15((v >= 10 && v < 15) ? z : 0.0f);
			s_third += xor32_f(s_third, lane);
		}
	}
	// (c) scores (the expressions of vk_rwmd_batch_kernel; sum (1 - x) over len rows = len - sum x); valid in lane 15 of each row
	B32Vals out;
	{
		const float len_t = len_main, inv_t = inv_main;
		const float a0 = inv_t * (len_t - s_main), a1 = inv_s * ts_main;
		const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
		const float raw = p.nbow ? 1.0f - cost : (len_t - cost) * inv_t;
		out.main = (raw * inv_t) * boost;
	}
	out.third = 0.0f;
	if (QPT == 3) {
		const float len_t = len_third, inv_t = inv_third;
		const float a0 = inv_t * (len_t - s_third), a1 = inv_s * ts_third;
		const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
		const float raw = p.nbow ? 1.0f - cost : (len_t - cost) * inv_t;
		out.third = (raw * inv_t) * boost;
	}
	return out;
}

// the scores of tile qt: lane 15 of each DPP row holds them (row = sentence x lane half).  Rows of absent queries (the
// last tile may hold fewer than QPT) are written too: the score array has n_qtiles * QPT rows.
template <int QPT>
__device__ __forceinline__ void batch32_store(const VkRwmdBatchParams &p, int qt, bool store_lane, unsigned lane_off, int lane, const B32Vals &v, int64_t stride) {
	// row base wave-uniform (scalar registers), the lane's part a 32-bit offset: no 64-bit address arithmetic in vector registers
	float *row = p.scores + (int64_t)(qt * QPT) * stride;
	if (store_lane) {
		row[lane_off] = v.main;
		if (QPT == 3 && lane < 32) (row + (int64_t)2 * stride)[lane_off] = v.third;
	}
}

// NW: waves per workgroup -- 8 (two per SIMD, 256 registers each) for rows of up to 304 features; 4 (one per SIMD, up to 512
// registers) for 768-d rows, whose 64 token columns per wave take 384 registers
template <int NK16, int QPT, bool W64, int NW = 8>
__global__ __launch_bounds__(64 * NW) void vk_rwmd_batch32_kernel(VkRwmdBatchParams p) {
	constexpr int QT_BYTES = NK16 * 1024;
	extern __shared__ float4 vk_smem4[];
	const uint8_t *qbuf = reinterpret_cast<const uint8_t *>(vk_smem4);
	float *param = reinterpret_cast<float *>(vk_smem4) + 2 * QT_BYTES / 4;   // [n_qtiles][8]: len of the tile's 3 queries, pad, 1 / len, pad
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n32 = lane & 31, h = lane >> 5;
	constexpr int SPC = W64 ? NW : 2 * NW;         // sentences per chunk (workgroup)
	const int64_t n_chunks = ((int64_t)p.n_sent + SPC - 1) / SPC;
	const int64_t stride = p.score_stride > 0 ? p.score_stride : p.n_sent;
	for (int i = threadIdx.x; i < p.n_qtiles * 8; i += 64 * NW) param[i] = p.q_param[i];
	// The two waves that share a SIMD (w and w + 4 of the workgroup) run the two halves of an interval in
	// opposite order: the "late" wave first finishes the epilogue of the previous tile (VALU) while the
	// other one issues its MFMAs, then they swap.  Barriers would otherwise keep all waves in phase:
	// every matrix core idle during the epilogues, every VALU idle during the MFMAs.
	const bool late = (wv & p.late_mask) != 0;

	for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const int64_t sent = W64 ? chunk * NW + wv : chunk * (2 * NW) + wv * 2 + (n32 >> 4);   // this lane's sentence
		const bool have = sent < p.n_sent;
		const int slen = have ? (p.sent_len ? p.sent_len[sent] : (W64 ? 64 : 32)) : 1;
		const int64_t sout = have ? (p.sent_id ? (int64_t)p.sent_id[sent] : sent) : 0;   // where its scores go
		const B32Sent sn{1.0f / (float)slen, (float)((W64 ? 64 : 32) - slen)};
		const float boost = (p.boost && have) ? p.boost[sout] : 1.0f;
		const bool store_lane = (W64 ? (lane & 31) == 31 : (lane & 15) == 15) && have;   // the last lane of a sentence's lanes holds its scores
		const unsigned lane_off = (unsigned)h * (unsigned)stride + (unsigned)sout;       // its place in the score rows of its half's query
		// ---- the wave's token tiles -> registers (read once per batch)
		bf16x8 x[2][NK16];
#pragma unroll
		for (int m = 0; m < 2; m++) {
			const int64_t tile = have ? (W64 ? sent * 4 + 2 * (n32 >> 4) + m : sent * 2 + m) : p.n_tiles;   // one zero tile follows the corpus
			const uint8_t *tp = p.tiles + tile * p.tile_bytes + (n32 & 15) * 16;
#pragma unroll
			for (int t = 0; t < NK16; t++)
				x[m][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (t >> 1) * 1024 + (2 * (t & 1) + h) * 256));
		}
		__syncthreads();   // previous chunk's readers are done with the LDS slots
#pragma unroll
		for (int b = wv; b < NK16; b += NW)
			__builtin_amdgcn_global_load_lds((vk_glb_ptr)(p.qtiles + b * 1024 + lane * 16), (vk_lds_ptr)(qbuf + b * 1024), 16, 0, 0);
		__syncthreads();

		f32x16 acc0, acc1;
		B32Vals pend{0.0f, 0.0f};
		for (int qt = 0; qt < p.n_qtiles; qt++) {
			const uint8_t *cur = qbuf + (qt & 1) * QT_BYTES + lane * 16;
			// next query tile: LDS-DMA from inside the MFMA sequence (batch32_mfma) into the other slot, complete at the barrier
			// that ends the iteration.  The tile after the last one is zero padding.
			const uint8_t *next_src = p.qtiles + (int64_t)(qt + 1) * QT_BYTES;
			const uint8_t *next_dst = qbuf + ((qt + 1) & 1) * QT_BYTES;
			if (!late) {
				// the scores of the previous tile leave now, not at the end of its iteration: the barrier there waits for
				// every outstanding memory operation of the wave (vmcnt(0) for the LDS-DMA), stores to HBM included
				if (qt > 0) batch32_store<QPT>(p, qt - 1, store_lane, lane_off, lane, pend, stride);
				batch32_mfma<NK16, NW>(cur, x, acc0, acc1, next_src, (unsigned)lane * 16u, next_dst, wv);
				pend = batch32_epilogue<QPT, W64>(p, lane, acc0, acc1, param + qt * 8, boost, sn);
			} else {
				if (qt > 0) batch32_store<QPT>(p, qt - 1, store_lane, lane_off, lane, batch32_epilogue<QPT, W64>(p, lane, acc0, acc1, param + (qt - 1) * 8, boost, sn), stride);
				batch32_mfma<NK16, NW>(cur, x, acc0, acc1, next_src, (unsigned)lane * 16u, next_dst, wv);
			}
#if !defined(VK_ABL) || VK_ABL < 6
			__syncthreads();   // next query tile is in place; this one may be overwritten
#endif
		}
		if (late) pend = batch32_epilogue<QPT, W64>(p, lane, acc0, acc1, param + (p.n_qtiles - 1) * 8, boost, sn);
		batch32_store<QPT>(p, p.n_qtiles - 1, store_lane, lane_off, lane, pend, stride);
	}
}

// ---------------------------------------------------------------------------
// Dense form of the 32-token kernel for batches of ten-token queries: 16 queries = 160 rows = FIVE A tiles exactly (3 queries
// per tile leave 2 of 32 rows empty: 86 tiles per 256 queries against 80 here, 7 % fewer MFMAs, query bytes and epilogues).
// A lane half h (rows 8 (i >> 2) + 4 h + (i & 3), i = accumulator register) serves 8 of the 16 queries of a super tile and
// every query stays inside one half, so the maxima over a query's rows are in-lane, with no exchange across the halves:
//   registers 0..9 of tile k: query F_k = 8 h + k of the super tile (tokens 0..9);
//   registers 10..15 of the five tiles (30 slots): queries G_0..G_2 = 8 h + 5 + g, ten tokens each, in slot order --
//     G_0: tile 0 slots 10..15, tile 1 slots 10..13;  G_1: tile 1 slots 14, 15, tile 2 slots 10..15, tile 3 slots 10, 11;
//     G_2: tile 3 slots 12..15, tile 4 slots 10..15.
// A split query carries its in-lane maxima (both chains) and its per-lane sum of row maxima from tile to tile (3 registers).
// Shorter queries leave zero rows (S = 0, clip 0, no masks); the host packs the tiles (vk_batch.cpp).
// ---------------------------------------------------------------------------

struct B32DState { int cb0, cb1; float zg; };

template <int K, bool W64, bool U16 = false>
__device__ __forceinline__ B32Vals batch32d_epilogue(const VkRwmdBatchParams &p, int lane, const f32x16 &acc0, const f32x16 &acc1,
	const float *super_param /* [16][2]: len, 1 / len of the super tile's queries */, float boost, B32DState &st, const B32Sent &sn) {
	constexpr int C_HI = K == 0 ? 10 : K == 1 ? 14 : K == 3 ? 12 : 16;   // slots [10, C_HI) continue the split query in progress
	constexpr bool CLOSES = K == 1 || K == 3 || K == 4;                  // ... and complete it
	constexpr bool STARTS = K == 0 || K == 1 || K == 3;                  // slots [C_HI, 16) start the next one
	constexpr int G = K == 1 ? 0 : K == 3 ? 1 : 2;
	const int h = lane >> 5, v = row_transpose_reg(lane);   // v: the query row whose maximum over the sentence this lane holds after the transposing exchange
	const float inv_s = sn.inv_s;
	const float len_f = super_param[2 * (8 * h + K)], inv_f = super_param[2 * (8 * h + K) + 1];
	float len_g = 0.0f, inv_g = 0.0f;
	if (CLOSES) { len_g = super_param[2 * (8 * h + 5 + G)]; inv_g = super_param[2 * (8 * h + 5 + G) + 1]; }
#if defined(VK_ABL) && (VK_ABL == 1 || VK_ABL >= 4)
	{
#pragma unroll
		for (int i = 0; i < 16; i++) asm volatile("" :: "v"(acc0[i]), "v"(acc1[i]));
		return {0.0f, 0.0f};
	}
#endif
	// (a) per token: maximum over the query's rows, in-lane; distances of this lane's two tokens; sum over the sentence
	int ca0 = fbits(acc0[0]), ca1 = fbits(acc1[0]);
#pragma unroll
	for (int i = 1; i < 10; i++) { ca0 = imax(ca0, fbits(acc0[i])); ca1 = imax(ca1, fbits(acc1[i])); }
	const float ts_f = tokens_sum<W64>((2.0f - sim_of_bits<U16>(ca0)) - sim_of_bits<U16>(ca1)) - sn.pad;
#pragma unroll
	for (int i = 10; i < C_HI; i++) { st.cb0 = imax(st.cb0, fbits(acc0[i])); st.cb1 = imax(st.cb1, fbits(acc1[i])); }
	float ts_g = 0.0f;
	if (CLOSES) ts_g = tokens_sum<W64>((2.0f - sim_of_bits<U16>(st.cb0)) - sim_of_bits<U16>(st.cb1)) - sn.pad;
	if (STARTS) {
		st.cb0 = fbits(acc0[C_HI]); st.cb1 = fbits(acc1[C_HI]);
#pragma unroll
		for (int i = C_HI + 1; i < 16; i++) { st.cb0 = imax(st.cb0, fbits(acc0[i])); st.cb1 = imax(st.cb1, fbits(acc1[i])); }
	}
	// (b) per query row: maximum over the sentence's tokens, transposed: lane v of the DPP row holds the one of register v
	__builtin_amdgcn_sched_barrier(0);   // (a) is done with the accumulators: m[] may take their registers (the kernel sits at the VGPR cap)
	int m[16];
#pragma unroll
	for (int i = 0; i < 16; i++) m[i] = imax(fbits(acc0[i]), fbits(acc1[i]));
	float z = sim_of_bits<U16>(row_transpose_imax16(m, lane));
	if (W64) z = fmaxf(z, xor16_f(z));
	const float s_f = row_sum_to_lane15(v < 10 ? z : 0.0f);
	if (C_HI > 10) st.zg += (v >= 10 && v < C_HI) ? z : 0.0f;
	float s_g = 0.0f;
	if (CLOSES) s_g = row_sum_to_lane15(st.zg);
	if (STARTS) st.zg = v >= C_HI ? z : 0.0f;
	// (c) scores, valid in lane 15 of each row (the expressions of batch32_epilogue)
	B32Vals out;
	{
		const float a0 = inv_f * (len_f - s_f), a1 = inv_s * ts_f;
		const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
		const float raw = p.nbow ? 1.0f - cost : (len_f - cost) * inv_f;
		out.main = (raw * inv_f) * boost;
	}
	out.third = 0.0f;
	if (CLOSES) {
		const float a0 = inv_g * (len_g - s_g), a1 = inv_s * ts_g;
		const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
		const float raw = p.nbow ? 1.0f - cost : (len_g - cost) * inv_g;
		out.third = (raw * inv_g) * boost;
	}
	return out;
}

constexpr int B32D_MAX_SUPER = 48;   // super tiles per launch: 768 queries, their parameters 6 KiB of LDS

// epilogue / store of tile K (0..4) of a super tile; K is wave-uniform, the five forms are the arms of one switch (unrolling the
// five tiles into the loop body instead let hipcc hoist their addresses and parameters: 30 more registers, scratch)
template <bool W64, bool U16 = false>
__device__ __forceinline__ B32Vals batch32d_epilogue_k(int K, const VkRwmdBatchParams &p, int lane, const f32x16 &acc0, const f32x16 &acc1,
	const float *super_param, float boost, B32DState &st, const B32Sent &sn) {
	switch (K) {
	case 0: return batch32d_epilogue<0, W64, U16>(p, lane, acc0, acc1, super_param, boost, st, sn);
	case 1: return batch32d_epilogue<1, W64, U16>(p, lane, acc0, acc1, super_param, boost, st, sn);
	case 2: return batch32d_epilogue<2, W64, U16>(p, lane, acc0, acc1, super_param, boost, st, sn);
	case 3: return batch32d_epilogue<3, W64, U16>(p, lane, acc0, acc1, super_param, boost, st, sn);
	default: return batch32d_epilogue<4, W64, U16>(p, lane, acc0, acc1, super_param, boost, st, sn);
	}
}

__device__ __forceinline__ void batch32d_store_k(int K, const VkRwmdBatchParams &p, int super, bool store_lane, unsigned lane_off, const B32Vals &v, int64_t stride) {
	// tiles 1, 3 and 4 complete a split query: G = 0, 1, 2; its row is 5 + G - K rows below the tile's own
	const bool closes = K == 1 || K == 3 || K == 4;
	const int g_row = K == 1 ? 4 : K == 3 ? 3 : 3;   // 5 + G - K
	float *row = p.scores + (int64_t)(super * 16 + K) * stride;   // wave-uniform; lane_off = (8 h) stride + sentence
	if (store_lane) {
		row[lane_off] = v.main;
		if (closes) (row + (int64_t)g_row * stride)[lane_off] = v.third;
	}
}

template <int NK16, bool W64, int NW = 8>
__global__ __launch_bounds__(64 * NW) void vk_rwmd_batch32d_kernel(VkRwmdBatchParams p) {
	constexpr int QT_BYTES = NK16 * 1024;
	extern __shared__ float4 vk_smem4[];
	const uint8_t *qbuf = reinterpret_cast<const uint8_t *>(vk_smem4);
	float *param = reinterpret_cast<float *>(vk_smem4) + 2 * QT_BYTES / 4;   // [n_super * 16][2]: len, 1 / len per query
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n32 = lane & 31, h = lane >> 5;
	const int n_super = p.n_qtiles / 5;
	constexpr int SPC = W64 ? NW : 2 * NW;         // sentences per chunk (workgroup)
	const int64_t n_chunks = ((int64_t)p.n_sent + SPC - 1) / SPC;
	const int64_t stride = p.score_stride > 0 ? p.score_stride : p.n_sent;
	for (int i = threadIdx.x; i < n_super * 32; i += 64 * NW) param[i] = p.q_param[i];
	const bool late = (wv & p.late_mask) != 0;   // see vk_rwmd_batch32_kernel

	for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const int64_t sent = W64 ? chunk * NW + wv : chunk * (2 * NW) + wv * 2 + (n32 >> 4);   // this lane's sentence
		const bool have = sent < p.n_sent;
		const int slen = have ? (p.sent_len ? p.sent_len[sent] : (W64 ? 64 : 32)) : 1;
		const int64_t sout = have ? (p.sent_id ? (int64_t)p.sent_id[sent] : sent) : 0;
		const B32Sent sn{1.0f / (float)slen, (float)((W64 ? 64 : 32) - slen)};
		const float boost = (p.boost && have) ? p.boost[sout] : 1.0f;
		const bool store_lane = (W64 ? (lane & 31) == 31 : (lane & 15) == 15) && have;
		const unsigned lane_off = (unsigned)(8 * h) * (unsigned)stride + (unsigned)sout;
		bf16x8 x[2][NK16];
#pragma unroll
		for (int m = 0; m < 2; m++) {
			const int64_t tile = have ? (W64 ? sent * 4 + 2 * (n32 >> 4) + m : sent * 2 + m) : p.n_tiles;   // one zero tile follows the corpus
			const uint8_t *tp = p.tiles + tile * p.tile_bytes + (n32 & 15) * 16;
#pragma unroll
			for (int t = 0; t < NK16; t++)
				x[m][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (t >> 1) * 1024 + (2 * (t & 1) + h) * 256));
		}
		__syncthreads();   // previous chunk's readers are done with the LDS slots
#pragma unroll
		for (int b = wv; b < NK16; b += NW)
			__builtin_amdgcn_global_load_lds((vk_glb_ptr)(p.qtiles + b * 1024 + lane * 16), (vk_lds_ptr)(qbuf + b * 1024), 16, 0, 0);
		__syncthreads();

		f32x16 acc0, acc1;
		B32Vals pend{0.0f, 0.0f};
		B32DState st{0, 0, 0.0f};
		int K = 0, super = 0;          // tile qt = 5 * super + K
		for (int qt = 0; qt < p.n_qtiles; qt++) {
			const int KP = K == 0 ? 4 : K - 1, super_p = K == 0 ? super - 1 : super;   // the tile before
			const uint8_t *cur = qbuf + (qt & 1) * QT_BYTES + lane * 16;
			const uint8_t *next_src = p.qtiles + (int64_t)(qt + 1) * QT_BYTES;
			const uint8_t *next_dst = qbuf + ((qt + 1) & 1) * QT_BYTES;
			if (!late) {
				if (qt > 0) batch32d_store_k(KP, p, super_p, store_lane, lane_off, pend, stride);
				batch32_mfma<NK16, NW>(cur, x, acc0, acc1, next_src, (unsigned)lane * 16u, next_dst, wv);
				pend = batch32d_epilogue_k<W64>(K, p, lane, acc0, acc1, param + super * 32, boost, st, sn);
			} else {
				if (qt > 0) batch32d_store_k(KP, p, super_p, store_lane, lane_off, batch32d_epilogue_k<W64>(KP, p, lane, acc0, acc1, param + super_p * 32, boost, st, sn), stride);
				batch32_mfma<NK16, NW>(cur, x, acc0, acc1, next_src, (unsigned)lane * 16u, next_dst, wv);
			}
#if !defined(VK_ABL) || VK_ABL < 6
			__syncthreads();   // next query tile is in place; this one may be overwritten
#endif
			if (++K == 5) { K = 0; super++; }
		}
		if (late) pend = batch32d_epilogue<4, W64>(p, lane, acc0, acc1, param + (n_super - 1) * 32, boost, st, sn);
		batch32d_store_k(4, p, n_super - 1, store_lane, lane_off, pend, stride);
	}
}

// ---------------------------------------------------------------------------
// The static layout (token ids + vocabulary; StaticEmbeddingSlice, vectorian/core/cpp/slice/static.h:71-75) under a batch of
// relaxed-WMD queries.  The reference builds ONE similarity table per query over the vocabulary (metric/static.cpp:9-78) and
// gathers a slice's rows from it by token id; so does this path, for the whole batch at once:
//   1. vk_table_batch_kernel: table[word][query tile][lane half][16] = clip(cosine), the packed 32-row query tiles (as the GEMM
//      kernels above take them) against the vocabulary on MFMA -- V x 32 n_qtiles x d, 77 GFLOP for 50,000 words and 256 queries
//      where the contextual GEMM issues 49 TFLOP -- stored in the accumulator layout of a 32x32 MFMA result, so that
//   2. vk_rwmd_static32_kernel fills acc0 / acc1 of a lane with two 16-byte loads each -- the 16 query rows of its lane half
//      against its token, gathered by token id -- and runs the SAME epilogues as the GEMM kernels (batch32_epilogue,
//      batch32d_epilogue): maxima over the query's rows in-lane, over the sentence's tokens by the transposing exchange, scores.
//      No LDS tiles, no barriers, waves independent; a sentence's padding points at a row of zeros (S = 0 exactly, as the zero
//      rows of the padded copy a ragged contextual corpus runs on).
// Every token id is read once per launch for all queries; what streams is the table, and it decides the time (the first form kept
// fp32 cells, 128 bytes per (token, query tile): 328 GB of gathers per 256 x 1 M x 32, a quarter of them past the L2 for Zipf(1.1)
// words -- 35.2 ms, no faster than the GEMM).  The cells are therefore 16-bit fixed point, round(65535 clip(S)): 64 bytes per
// (token, tile).  The similarities are clipped to [0, 1] anyway, the maxima of the epilogues are integer maxima either way, and a
// cell is within 7.7e-6 of its fp32 value -- the scores of this pass rank the slices (within 1e-5 of the fp32 scores: the margin of
// the canonical restating covers it, vk_batch.cpp), the scores a caller with flows receives are restated from canonical rows.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void vk_table_batch_kernel(const uint8_t *__restrict__ etiles, int64_t n_vtiles, int tile_bytes, int nk16,
	const uint8_t *__restrict__ qtiles, int n_qtiles, uint16_t *__restrict__ table) {
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int n32 = lane & 31, h = lane >> 5;
	const int64_t pair = (int64_t)blockIdx.x * 4 + wv;           // vocabulary tiles 2 pair, 2 pair + 1: 32 words = the 32 columns
	if (pair * 2 >= n_vtiles) return;
	int64_t tile = pair * 2 + (n32 >> 4);
	if (tile >= n_vtiles) tile = n_vtiles - 1;                  // (an odd count: the zero tile behind the vocabulary once more)
	const uint8_t *tp = etiles + tile * tile_bytes + (n32 & 15) * 16;
	uint16_t *out = table + (pair * 32 + n32) * ((int64_t)n_qtiles * 32) + h * 16;
	for (int qt = 0; qt < n_qtiles; qt++) {
		f32x16 acc;
#pragma unroll
		for (int i = 0; i < 16; i++) acc[i] = 0.0f;
		const uint8_t *a = qtiles + (int64_t)qt * nk16 * 1024 + lane * 16;
#pragma unroll 2
		for (int t = 0; t < nk16; t++) {
			const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a + t * 1024);
			const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(tp + (t >> 1) * 1024 + (2 * (t & 1) + h) * 256);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, av), __builtin_bit_cast(bf16x8n, bv), acc, 0, 0, 0);
		}
		// clip to [0, 1] (SimilarityMatrix::clip, metric/metric.h:28-30), then 16-bit fixed point: slots 2 k, 2 k + 1 in one dword
		uint32_t w[8];
#pragma unroll
		for (int k = 0; k < 8; k++)
			w[k] = (uint32_t)__builtin_rintf(clip01(acc[2 * k]) * 65535.0f) | ((uint32_t)__builtin_rintf(clip01(acc[2 * k + 1]) * 65535.0f) << 16);
		*reinterpret_cast<uint4 *>(out + qt * 32) = make_uint4(w[0], w[1], w[2], w[3]);
		*reinterpret_cast<uint4 *>(out + qt * 32 + 8) = make_uint4(w[4], w[5], w[6], w[7]);
	}
}

__global__ void vk_table_batch_fix_kernel(uint16_t *__restrict__ table, const int64_t *__restrict__ offsets, int n) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) table[offsets[i]] = 65535;   // sim[id(t_j)][j] = 1
}

// the lane's two tokens of its sentence: chain m holds token 16 m + (n & 15) of sentence n >> 4, or (W64) token
// 16 (2 (n >> 4) + m) + (n & 15) of the wave's one sentence -- the column order of the GEMM kernels above
template <int QPT, bool DENSE, bool W64>
__global__ __launch_bounds__(256) void vk_rwmd_static32_kernel(VkRwmdBatchParams p) {
	extern __shared__ float4 vk_smem4[];
	float *param = reinterpret_cast<float *>(vk_smem4);
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n32 = lane & 31, h = lane >> 5;
	constexpr int SPW = W64 ? 1 : 2;                 // sentences per wave
	const int64_t n_chunks = ((int64_t)p.n_sent + SPW - 1) / SPW;
	const int64_t stride = p.score_stride > 0 ? p.score_stride : p.n_sent;
	const int n_super = p.n_qtiles / 5;
	for (int i = threadIdx.x; i < (DENSE ? n_super * 32 : p.n_qtiles * 8); i += 256) param[i] = p.q_param[i];
	__syncthreads();
	for (int64_t chunk = (int64_t)blockIdx.x * 4 + wv; chunk < n_chunks; chunk += (int64_t)gridDim.x * 4) {
		const int64_t sent = W64 ? chunk : chunk * 2 + (n32 >> 4);
		const bool have = sent < p.n_sent;
		const int64_t srow = have ? (p.sent_id ? (int64_t)p.sent_id[sent] : sent) : 0;   // row of the slice table = where its scores go
		const int s_a = have ? p.sent_start[srow] : 0;
		const int slen = have ? p.sent_end[srow] - s_a : 1;
		const B32Sent sn{1.0f / (float)slen, (float)((W64 ? 64 : 32) - slen)};
		const float boost = (p.boost && have) ? p.boost[srow] : 1.0f;
		const bool store_lane = (W64 ? (lane & 31) == 31 : (lane & 15) == 15) && have;
		const unsigned lane_off = (unsigned)((DENSE ? 8 : 1) * h) * (unsigned)stride + (unsigned)srow;
		const uint16_t *r0, *r1;
		{
			const int pos0 = (W64 ? 32 * (n32 >> 4) : 0) + (n32 & 15), pos1 = pos0 + 16;
			const int64_t w0 = (have && pos0 < slen) ? p.tok_id[s_a + pos0] : p.zero_row;
			const int64_t w1 = (have && pos1 < slen) ? p.tok_id[s_a + pos1] : p.zero_row;
			r0 = p.table + w0 * p.table_row + h * 16;
			r1 = p.table + w1 * p.table_row + h * 16;
		}
		f32x16 acc0, acc1;   // (holding the cells' integers as bit patterns: the epilogues' U16 form)
		uint4 nx0[2], nx1[2];
#pragma unroll
		for (int j = 0; j < 2; j++) { nx0[j] = *reinterpret_cast<const uint4 *>(r0 + 8 * j); nx1[j] = *reinterpret_cast<const uint4 *>(r1 + 8 * j); }
		B32DState st{0, 0, 0.0f};
		int K = 0, super = 0;
		for (int qt = 0; qt < p.n_qtiles; qt++) {
#pragma unroll
			for (int j = 0; j < 2; j++) {
				const uint32_t a[4] = {nx0[j].x, nx0[j].y, nx0[j].z, nx0[j].w}, b[4] = {nx1[j].x, nx1[j].y, nx1[j].z, nx1[j].w};
#pragma unroll
				for (int e = 0; e < 4; e++) {
					acc0[8 * j + 2 * e] = __builtin_bit_cast(float, a[e] & 0xffffu); acc0[8 * j + 2 * e + 1] = __builtin_bit_cast(float, a[e] >> 16);
					acc1[8 * j + 2 * e] = __builtin_bit_cast(float, b[e] & 0xffffu); acc1[8 * j + 2 * e + 1] = __builtin_bit_cast(float, b[e] >> 16);
				}
			}
			// the next tile's cells are requested before this tile's epilogue runs (the last request re-reads tile 0: in range, unused)
			const int qn = qt + 1 < p.n_qtiles ? qt + 1 : 0;
#pragma unroll
			for (int j = 0; j < 2; j++) { nx0[j] = *reinterpret_cast<const uint4 *>(r0 + qn * 32 + 8 * j); nx1[j] = *reinterpret_cast<const uint4 *>(r1 + qn * 32 + 8 * j); }
			if constexpr (DENSE) {
				const B32Vals v = batch32d_epilogue_k<W64, true>(K, p, lane, acc0, acc1, param + super * 32, boost, st, sn);
				batch32d_store_k(K, p, super, store_lane, lane_off, v, stride);
				if (++K == 5) { K = 0; super++; }
			} else {
				const B32Vals v = batch32_epilogue<QPT, W64, true>(p, lane, acc0, acc1, param + qt * 8, boost, sn);
				batch32_store<QPT>(p, qt, store_lane, lane_off, lane, v, stride);
			}
		}
	}
}

extern "C" hipError_t vk_launch_table_batch(const uint8_t *etiles, int64_t n_vtiles, int32_t tile_bytes, int32_t nk16, const uint8_t *qtiles, int32_t n_qtiles,
	uint16_t *table, hipStream_t stream) {
	const int64_t pairs = (n_vtiles + 1) / 2;
	if (pairs < 1 || n_qtiles < 1) return hipSuccess;
	vk_table_batch_kernel<<<(unsigned)((pairs + 3) / 4), 256, 0, stream>>>(etiles, n_vtiles, tile_bytes, nk16, qtiles, n_qtiles, table);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_table_batch_fix(uint16_t *table, const int64_t *offsets, int32_t n, hipStream_t stream) {
	if (n < 1) return hipSuccess;
	vk_table_batch_fix_kernel<<<(n + 255) / 256, 256, 0, stream>>>(table, offsets, n);
	return hipGetLastError();
}

// p->n_qtiles tiles of the table (dense: 5 per super tile of 16 queries), p->table_row floats per word; p->sent_id: the rows of the
// slice table this launch takes (null: 0 .. n_sent), each of at most 32 tokens (w64: 64); scores as the GEMM kernels write them
extern "C" hipError_t vk_launch_rwmd_static32(const VkRwmdBatchParams *pp, int32_t w64, hipStream_t stream) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (pp->qpt != 2 && pp->qpt != 3) return hipErrorNotSupported;
	const int64_t n_chunks = ((int64_t)pp->n_sent + (w64 ? 0 : 1)) / (w64 ? 1 : 2);
	const int64_t blocks = (n_chunks + 3) / 4;
	const int grid = (int)(blocks < (int64_t)cus * 8 ? blocks : (int64_t)cus * 8);
	if (grid < 1) return hipSuccess;
	const int64_t stride = pp->score_stride > 0 ? pp->score_stride : pp->n_sent;
	void (*kernel)(VkRwmdBatchParams);
	if (pp->dense) kernel = w64 ? vk_rwmd_static32_kernel<3, true, true> : vk_rwmd_static32_kernel<3, true, false>;
	else if (pp->qpt == 3) kernel = w64 ? vk_rwmd_static32_kernel<3, false, true> : vk_rwmd_static32_kernel<3, false, false>;
	else kernel = w64 ? vk_rwmd_static32_kernel<2, false, true> : vk_rwmd_static32_kernel<2, false, false>;
	// the tiles' parameters are staged in LDS (as the GEMM kernels do): launches of at most B32D_MAX_SUPER super tiles / B32_MAX_QTILES tiles
	// 40 query tiles per pass over the token ids (2.5 KB of a word's table row per visit): 256 queries in two passes took 19.1 ms where one
	// pass over all 80 tiles took 20.3 and four passes of 20 tiles 19.5 (MI355X, 1 M x 32 tokens, Zipf(1.1) over 50,000 words)
	int step = 40;
	if (const char *e = getenv("VK_STATIC_TILES")) {   // tuning aid
		const int v = atoi(e);
		if (v > 0) step = pp->dense ? (v + 4) / 5 * 5 : v;
		if (step > (pp->dense ? B32D_MAX_SUPER * 5 : B32_MAX_QTILES)) step = pp->dense ? B32D_MAX_SUPER * 5 : B32_MAX_QTILES;
	}
	for (int t0 = 0; t0 < pp->n_qtiles; t0 += step) {
		VkRwmdBatchParams p = *pp;
		p.n_qtiles = pp->n_qtiles - t0 < step ? pp->n_qtiles - t0 : step;
		p.table = pp->table + (size_t)t0 * 32;
		p.q_param = pp->q_param + (pp->dense ? (size_t)(t0 / 5) * 32 : (size_t)t0 * 8);
		p.scores = pp->scores + (size_t)(pp->dense ? (t0 / 5) * 16 : t0 * pp->qpt) * stride;
		const size_t smem = (size_t)(pp->dense ? (p.n_qtiles / 5) * 32 : p.n_qtiles * 8) * 4 + 16;
		kernel<<<grid, 256, smem, stream>>>(p);
		const hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

template <int NK, bool HALF>
static hipError_t launch_rwmd_batch_tps(const VkRwmdBatchParams &p, size_t smem, hipStream_t stream) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int tpw = p.tiles_per_sent == 3 ? 3 : 4;
	const int per_cu = NK > 12 ? 1 : 2;     // workgroups of 4 waves per CU (768-d rows: one wave per SIMD)
	const int64_t n_chunks = (p.n_tiles + tpw * 4 - 1) / (tpw * 4);
	const int grid = (int)(n_chunks < (int64_t)cus * per_cu ? n_chunks : (int64_t)cus * per_cu);
	if (grid < 1) return hipSuccess;
	switch (p.tiles_per_sent) {
	case 1: vk_rwmd_batch_kernel<NK, HALF, 1><<<grid, 256, smem, stream>>>(p); break;
	case 2: vk_rwmd_batch_kernel<NK, HALF, 2><<<grid, 256, smem, stream>>>(p); break;
	case 3: vk_rwmd_batch_kernel<NK, HALF, 3><<<grid, 256, smem, stream>>>(p); break;
	default: vk_rwmd_batch_kernel<NK, HALF, 4><<<grid, 256, smem, stream>>>(p); break;
	}
	return hipGetLastError();
}

// 32-token sentences: p->qtiles holds p->n_qtiles A tiles of 32 rows (p->qpt queries each); p->scores has n_qtiles * qpt rows
template <int NK16, int NW>
static hipError_t launch_batch32(const VkRwmdBatchParams *pp, bool w64, int grid, hipStream_t stream) {
	if (pp->dense) {
		// p->n_qtiles = 5 x super tiles of 16 queries; q_param [n_super * 16][2]; scores n_super * 16 rows
		const int n_super = pp->n_qtiles / 5;
		const int64_t stride = pp->score_stride > 0 ? pp->score_stride : pp->n_sent;
		const size_t smem = (size_t)2 * NK16 * 1024 + (size_t)B32D_MAX_SUPER * 128;
		void (*kernel)(VkRwmdBatchParams) = w64 ? vk_rwmd_batch32d_kernel<NK16, true, NW> : vk_rwmd_batch32d_kernel<NK16, false, NW>;
		if (smem > 64 * 1024) {
			const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
			if (e != hipSuccess) return e;
		}
		for (int s0 = 0; s0 < n_super; s0 += B32D_MAX_SUPER) {
			VkRwmdBatchParams p = *pp;
			const int ns = n_super - s0 < B32D_MAX_SUPER ? n_super - s0 : B32D_MAX_SUPER;
			p.n_qtiles = ns * 5;
			p.qtiles = pp->qtiles + (size_t)s0 * 5 * NK16 * 1024;
			p.q_param = pp->q_param + (size_t)s0 * 32;
			p.scores = pp->scores + (size_t)s0 * 16 * stride;
			kernel<<<grid, 64 * NW, smem, stream>>>(p);
			const hipError_t e = hipGetLastError();
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	}
	const int64_t stride = pp->score_stride > 0 ? pp->score_stride : pp->n_sent;
	const size_t smem = (size_t)2 * NK16 * 1024 + (size_t)B32_MAX_QTILES * 32;
	void (*kernel)(VkRwmdBatchParams) = pp->qpt == 3 ? (w64 ? vk_rwmd_batch32_kernel<NK16, 3, true, NW> : vk_rwmd_batch32_kernel<NK16, 3, false, NW>)
		: (w64 ? vk_rwmd_batch32_kernel<NK16, 2, true, NW> : vk_rwmd_batch32_kernel<NK16, 2, false, NW>);
	if (smem > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	for (int t0 = 0; t0 < pp->n_qtiles; t0 += B32_MAX_QTILES) {
		VkRwmdBatchParams p = *pp;
		p.n_qtiles = pp->n_qtiles - t0 < B32_MAX_QTILES ? pp->n_qtiles - t0 : B32_MAX_QTILES;
		p.qtiles = pp->qtiles + (size_t)t0 * NK16 * 1024;
		p.q_param = pp->q_param + (size_t)t0 * 8;
		p.scores = pp->scores + (size_t)t0 * pp->qpt * stride;
		kernel<<<grid, 64 * NW, smem, stream>>>(p);
		const hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	return hipSuccess;
}

// 32-token sentences: p->qtiles holds p->n_qtiles A tiles of 32 rows (p->qpt queries each); p->scores has n_qtiles * qpt rows
extern "C" hipError_t vk_launch_rwmd_batch32(const VkRwmdBatchParams *pp, hipStream_t stream) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if ((pp->tiles_per_sent != 2 && pp->tiles_per_sent != 4) || (pp->qpt != 2 && pp->qpt != 3)) return hipErrorNotSupported;
	const bool w64 = pp->tiles_per_sent == 4;   // 64-token (padded) sentences: one per wave
	const bool wide = pp->nk == 24 && pp->half == 0;   // 768-d rows: four waves per workgroup, one per SIMD
	const int spc = (w64 ? 1 : 2) * (wide ? 4 : 8);
	const int64_t n_chunks = ((int64_t)pp->n_sent + spc - 1) / spc;
	const int grid = (int)(n_chunks < (int64_t)cus ? n_chunks : (int64_t)cus);
	if (grid < 1) return hipSuccess;
	if (pp->nk == 10 && pp->half == 1) return launch_batch32<19, 8>(pp, w64, grid, stream);
	if (pp->nk == 4 && pp->half == 0) return launch_batch32<8, 8>(pp, w64, grid, stream);
	if (wide) return launch_batch32<48, 4>(pp, w64, grid, stream);
	return hipErrorNotSupported;
}

// row widths the batched kernels are built for: 300-d (10 K-steps, the last half filled), 128-d, 768-d
extern "C" int vk_rwmd_batch_supported(int32_t nk, int32_t half) {
	return (nk == 10 && half == 1) || (nk == 4 && half == 0) || (nk == 24 && half == 0);
}

// returns hipErrorNotSupported when no batched kernel exists for this corpus shape
extern "C" hipError_t vk_launch_rwmd_batch(const VkRwmdBatchParams *p, hipStream_t stream) {
	const size_t smem = (size_t)p->tile_bytes * 2;
	if (p->nk == 10 && p->half == 1) return launch_rwmd_batch_tps<10, true>(*p, smem, stream);
	if (p->nk == 4 && p->half == 0) return launch_rwmd_batch_tps<4, false>(*p, smem, stream);
	if (p->nk == 24 && p->half == 0) return launch_rwmd_batch_tps<24, false>(*p, smem, stream);
	return hipErrorNotSupported;
}
