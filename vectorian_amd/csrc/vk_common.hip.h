// vk_common.hip.h -- gfx950 (MI355X, CDNA4) kernels of the brute-force alignment search.
//
// Written for wave64 / MFMA / LDS of gfx950 only (no portability layer).
// Compiled with -ffp-contract=off: the DP recurrences must be the literal fp32
// add / subtract / max sequence of the oracle (oracle/vk_oracle.c).
//
// Reference functions realised here (paths relative to the reference tree):
//   similarity  : vectorian/sim/vector.py:66-78 (cosine of unit rows),
//                 vectorian/core/cpp/metric/metric.h:28-30 (clip),
//                 metric/contextual.cpp:26-63, metric/static.cpp:9-78
//   slices      : slice/contextual.h:65-67, slice/static.h:71-75, document.h:147-169
//   alignment   : metric/alignment.h:247-294 (make_match -> pyalign solve), :84-106 (reference_score),
//                 match/match.h:295-307 (Score)
//   result set  : result_set.h:32-93, match/match_impl.h:8-42
//
// Data layout in HBM ("tiles"): rows (token vectors, or vocabulary vectors) are stored
// as unit-norm bf16 in MFMA operand order.  A tile is 16 consecutive rows; for each
// K-step t of 32 features the tile holds one 1 KiB block in which lane l
// (l = 16*g + i) owns the 16 bytes  row i, features 32t + 8g .. 32t + 8g + 7.
// If d_pad % 32 == 16 the last block is a half block (512 bytes, g = 0, 1 only); lanes
// 32..63 feed zeros to that K-step.  A wave therefore reads a tile with NK perfectly
// coalesced global_load_dwordx4 and feeds the registers to v_mfma_f32_16x16x32_bf16
// without any shuffle or LDS staging.
//
// One MFMA opcode only on the accumulator chain: on gfx950 (ROCm 7.2 hipcc) a
// v_mfma_f32_16x16x16_bf16 whose SrcC is the vDst of the immediately preceding
// v_mfma_f32_16x16x32_bf16 reads stale accumulator registers -- the compiler inserts no
// wait states for that opcode change (tools/probe/mfma_hazard.hip reproduces it:
// 128 of 256 results differ).  Hence the K=16 tail is issued as a K=32 step.
//
// Shared by the translation units vk_*.hip: wave helpers, the similarity tile (MFMA) and the DP / transport row
// sweeps.  Everything here is a template or __forceinline__ device function; the kernels live in the .hip files.

#ifndef VK_COMMON_HIP_H
#define VK_COMMON_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vk_device.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VK_NEG_INF (-__builtin_inff())
#define VK_RUN 4   // consecutive groups of 4 slices per wave turn (vk_score_kernel); 1, 2 and 8 measured the same on aligned 32-token slices
#ifdef VK_DBG_NOINLINE
#define VK_DP_INLINE __attribute__((noinline))
#else
#define VK_DP_INLINE __forceinline__
#endif

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
	// lanes without a source lane keep `old` (bound_ctrl = 0)
	return __builtin_bit_cast(float,
		__builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
// lanes without a source lane read zero (bound_ctrl): lets the compiler fold the move into the consuming VALU op
template <int CTRL>
__device__ __forceinline__ float dpp_zero(float src) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, true));
}
#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHR2 0x112
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_SHR8 0x118

// max over each 16-lane DPP row; result valid in lane 15 of the row
__device__ __forceinline__ float row_max_to_lane15(float x) {
	x = fmaxf(x, dpp_f<DPP_ROW_SHR1>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR2>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR4>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR8>(x, x));
	return x;
}

__device__ __forceinline__ void wave_lds_fence() {
	// one wave's LDS operations execute in order; this only stops the compiler
	// from moving LDS accesses across the point and drains lgkmcnt
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The value of the lane 32 / 16 away at VALU speed: gfx950's v_permlane32_swap / v_permlane16_swap exchange the halves / the odd and
// even DPP rows of two registers (tools/probe/permlane_swap.hip prints what lands where).  Written as asm: given the same value
// twice the builtin came back folded to one of its two results (vk_doc_kernel lost three of its four shares), and operands fresh
// from the VALU were swapped stale in the probe -- wait states either side.
__device__ __forceinline__ float lane_xor32(float x, int lane) {
	float a = x, b = x;
	asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	return lane < 32 ? b : a;
}
__device__ __forceinline__ float lane_xor16(float x, int lane) {
	float a = x, b = x;
	asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
	return (lane & 16) ? a : b;
}

__device__ __forceinline__ float clip01(float x) {
	// xt::clip(sim, 0, 1); NaN -> 0 as the oracle does
	return fminf(fmaxf(x, 0.0f), 1.0f);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
	uint32_t u = __builtin_bit_cast(uint32_t, x);
	if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
	u += 0x7fffu + ((u >> 16) & 1u);
	return (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------------------
// similarity of one 16-row tile against the query: S^T = Q * X^T on MFMA.
// A operand = query fragment (rows = query tokens), B operand = token tile
// (columns = tokens).  Result: lane l holds S[token l&15][query 4*(l>>4) + r], r=0..3.
// ---------------------------------------------------------------------------

// NK = number of K=32 steps (the last one half filled when HALF)
template <int NK, bool HALF>
struct QFrag {
	bf16x8 q[NK > 0 ? NK : 1];
};

__device__ __forceinline__ bf16x8 load_half_block(const uint8_t *__restrict__ p, int lane, bool nt) {
	// half block: 32 x 16 bytes; lanes 32..63 contribute zeros to the K-step
	const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p + (lane & 31) * 16);
	bf16x8 x = nt ? __builtin_nontemporal_load(src) : *src;
	const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
	return lane < 32 ? x : z;
}

template <int NK, bool HALF>
__device__ __forceinline__ void load_qfrag(QFrag<NK, HALF> &f, const uint8_t *__restrict__ qtile, int lane) {
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) f.q[t] = load_half_block(qtile + t * 1024, lane, false);
		else f.q[t] = *reinterpret_cast<const bf16x8 *>(qtile + t * 1024 + lane * 16);
	}
}

// (The scheduler keeps a rolling four of the NK loads in flight here, not all of them; with 12 waves per CU that is enough for
// bf16 rows -- a sched_barrier that forces all ten costs 1.5 % at 300-d, DESIGN 5.0 -- unlike the fp32 form below.)
template <int NK, bool HALF, bool BARRIER = false>
__device__ __forceinline__ f32x4 sim_tile(const QFrag<NK, HALF> &f, const uint8_t *__restrict__ tile, int lane) {
	bf16x8 x[NK > 0 ? NK : 1];
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) x[t] = load_half_block(tile + t * 1024, lane, true);
		else x[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
	}
	if constexpr (BARRIER) __builtin_amdgcn_sched_barrier(0);
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int t = 0; t < NK; t++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.q[t], x[t], acc, 0, 0, 0);
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// Large d (e.g. 768): the query tile is staged once per workgroup in LDS (NK KiB) and its fragments
// are re-read per K-step with ds_read_b128 (conflict-free: 64 consecutive 16-byte slots); the token
// tile's NK loads are all issued up front.  Same MFMA sequence as sim_tile.
template <int NK, bool HALF>
__device__ __forceinline__ f32x4 sim_tile_qlds(const uint8_t *__restrict__ qlds, const uint8_t *__restrict__ tile, int lane) {
	bf16x8 x[NK];
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) x[t] = load_half_block(tile + t * 1024, lane, true);
		else x[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
	}
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int t = 0; t < NK; t++) {
		bf16x8 q = *reinterpret_cast<const bf16x8 *>(qlds + t * 1024 + ((HALF && t == NK - 1) ? (lane & 31) : lane) * 16);
		if (HALF && t == NK - 1) {
			const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
			q = lane < 32 ? q : z;
		}
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x[t], acc, 0, 0, 0);
	}
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// fp32 unit rows (VK_PREC_F32, the reference's own precision), NB16 blocks of 16 features known at compile time: all the
// tile's loads are issued up front (NB16 KiB in flight per wave, as the bf16 form keeps its whole tile in flight; the generic
// loop below holds 4 KiB), the query tile is read from LDS block by block, four v_mfma_f32_16x16x4_f32 per block -- exact fp32
// products, fp32 accumulation in k order.  Same MFMA sequence as sim_tile_generic(prec = 1): bit-identical similarities.
template <int NB16>
__device__ __forceinline__ f32x4 sim_tile_f32_qlds(const uint8_t *__restrict__ qlds, const uint8_t *__restrict__ tile, int lane) {
	f32x4 x[NB16];
#pragma unroll
	for (int b = 0; b < NB16; b++) x[b] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + b * 1024 + lane * 16));
	// without the barrier the machine scheduler sinks every load to its MFMAs to save registers and leaves 2 - 3 KiB in flight
	__builtin_amdgcn_sched_barrier(0);
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int b = 0; b < NB16; b++) {
		const f32x4 q = *reinterpret_cast<const f32x4 *>(qlds + b * 1024 + lane * 16);
#pragma unroll
		for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[b][e], acc, 0, 0, 0);
	}
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// any d: query fragments re-read per K-step (L1/L2 resident), runtime trip count.
// Same MFMA sequence as sim_tile, hence bit-identical similarities.
// DEEP: eight K-steps (8 KiB per wave) in flight instead of four -- rows of 256 features and more, where the tile loop of a
// ragged corpus otherwise runs out of loads in flight (1024-d, U{8..64} tokens: 4.9 -> 5.5 TB/s; tools/sweep_dims.py); costs
// 24 VGPRs, so the narrower rows keep the four-deep form and their occupancy.
// PREC: -1 the row type is the runtime argument `prec`; 0 / 1: bf16 / fp32 rows known at compile time (the scoring kernel
// instantiates one form per row type, so that the registers of the fp32 batches do not count against the bf16 kernels).
template <bool DEEP = false, int PREC = -1>
__device__ __forceinline__ f32x4 sim_tile_generic(const uint8_t *__restrict__ qtile, const uint8_t *__restrict__ tile,
	int nk, int half, int lane, int prec = 0) {
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
	if (PREC == 1 || (PREC < 0 && prec)) {
		// fp32 rows (the reference's own precision): nk blocks of 16 features, four v_mfma_f32_16x16x4_f32 per block
		int b = 0;
		if constexpr (DEEP)
		for (; b + 8 <= nk; b += 8) {   // eight blocks (8 KiB per wave) in flight
			f32x4 q[8], x[8];
#pragma unroll
			for (int i = 0; i < 8; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + (b + i) * 1024 + lane * 16));
#pragma unroll
			for (int i = 0; i < 8; i++) q[i] = *reinterpret_cast<const f32x4 *>(qtile + (b + i) * 1024 + lane * 16);
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int i = 0; i < 8; i++) {
#pragma unroll
				for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[i][e], x[i][e], acc, 0, 0, 0);
			}
		}
		for (; b + 4 <= nk; b += 4) {   // four blocks in flight
			f32x4 q[4], x[4];
#pragma unroll
			for (int i = 0; i < 4; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + (b + i) * 1024 + lane * 16));
#pragma unroll
			for (int i = 0; i < 4; i++) q[i] = *reinterpret_cast<const f32x4 *>(qtile + (b + i) * 1024 + lane * 16);
			__builtin_amdgcn_sched_barrier(0);   // keep the batch of loads in front of its MFMAs (the scheduler sinks them otherwise)
#pragma unroll
			for (int i = 0; i < 4; i++) {
#pragma unroll
				for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[i][e], x[i][e], acc, 0, 0, 0);
			}
		}
		for (; b < nk; b++) {
			const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + b * 1024 + lane * 16));
			const f32x4 q = *reinterpret_cast<const f32x4 *>(qtile + b * 1024 + lane * 16);
#pragma unroll
			for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[e], acc, 0, 0, 0);
		}
		acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
		return acc;
	}
	const int nfull = half ? nk - 1 : nk;
	int t = 0;
	if constexpr (DEEP)
	for (; t + 8 <= nfull; t += 8) {
		bf16x8 q[8], x[8];
#pragma unroll
		for (int i = 0; i < 8; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + (t + i) * 1024 + lane * 16));
#pragma unroll
		for (int i = 0; i < 8; i++) q[i] = *reinterpret_cast<const bf16x8 *>(qtile + (t + i) * 1024 + lane * 16);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < 8; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[i], x[i], acc, 0, 0, 0);
	}
	for (; t + 4 <= nfull; t += 4) {   // four K-steps in flight
		bf16x8 q[4], x[4];
#pragma unroll
		for (int i = 0; i < 4; i++) {
			x[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + (t + i) * 1024 + lane * 16));
			q[i] = *reinterpret_cast<const bf16x8 *>(qtile + (t + i) * 1024 + lane * 16);
		}
#pragma unroll
		for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[i], x[i], acc, 0, 0, 0);
	}
	for (; t < nfull; t++) {
		const bf16x8 q = *reinterpret_cast<const bf16x8 *>(qtile + t * 1024 + lane * 16);
		const bf16x8 x = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x, acc, 0, 0, 0);
	}
	if (half) {
		const bf16x8 q = load_half_block(qtile + nfull * 1024, lane, false);
		const bf16x8 x = load_half_block(tile + nfull * 1024, lane, true);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x, acc, 0, 0, 0);
	}
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// ---------------------------------------------------------------------------
// Canonical similarities: what the k winners' tracebacks, flows and exact transport plans are computed from.
// The million-row scoring pass takes its cosines on MFMA (fp32 accumulation in the matrix pipe's own order); a traceback
// that must equal the oracle's index for index cannot hang on the last bit of that sum (co-optimal alignments tie exactly
// in real arithmetic when a word repeats, and the rounding of S then picks the path).  So everything that is computed for
// the winners only restates the cosine in ONE arithmetic that the oracle shares bit for bit (oracle/vk_oracle.c dot_f32):
// four partial sums over k mod 4 in double, k ascending, the d % 4 trailing features all into the first, then
// (s0 + s1) + (s2 + s3) rounded to float once.  Products of two bf16 (or fp32) values are exact in double, so a fused
// multiply-add and the oracle's multiply-then-add round identically.  Cost: k x |s| x |q| x d double FMAs, microseconds.
// One lane computes NC consecutive query columns (c0 a multiple of NC <= 4, all inside one 16-row query tile) for row `row`
// of the token tile; both tiles in operand order (bf16: DESIGN 3; fp32: block of 16 features, lane 16 g + i owns row i,
// features 16 b + 4 s + g).
// ---------------------------------------------------------------------------

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

__device__ __forceinline__ float tile_elem(const uint8_t *__restrict__ tile, int row, int k, int prec) {
	if (prec) return *reinterpret_cast<const float *>(tile + (int64_t)(k >> 4) * 1024 + ((k & 3) * 16 + row) * 16 + ((k & 15) >> 2) * 4);
	return bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(tile + (int64_t)(k >> 5) * 1024 + (((k & 31) >> 3) * 16 + row) * 16 + (k & 7) * 2));
}

// element k of a row given by the address of its first 16 bytes in block 0 (canon_row_ptr)
__device__ __forceinline__ float row_elem(const uint8_t *__restrict__ rowp, int k, int prec) {
	if (prec) return *reinterpret_cast<const float *>(rowp + (int64_t)(k >> 4) * 1024 + (k & 3) * 256 + ((k & 15) >> 2) * 4);
	return bf16_bits_to_f32(*reinterpret_cast<const uint16_t *>(rowp + (int64_t)(k >> 5) * 1024 + ((k & 31) >> 3) * 256 + (k & 7) * 2));
}

template <int NC>
__device__ __forceinline__ void sim_canon_cols(const uint8_t *__restrict__ tile, int row, const uint8_t *__restrict__ qtile, int c0,
	int d, int prec, float *out) {
	double acc[NC][4];
#pragma unroll
	for (int c = 0; c < NC; c++) { acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0; acc[c][3] = 0.0; }
	const int d4 = d & ~3;
	int k0 = 0;
	if (prec) {
#pragma unroll 1
		for (; k0 + 16 <= d4; k0 += 16) {
			const int64_t off = (int64_t)(k0 >> 4) * 1024;
#pragma unroll
			for (int g = 0; g < 4; g++) {   // lane 16 g + i of the block: features k0 + 4 s + g, s = 0 .. 3 (k mod 4 = g, ascending in s)
				const f32x4 x = *reinterpret_cast<const f32x4 *>(tile + off + (g * 16 + row) * 16);
#pragma unroll
				for (int c = 0; c < NC; c++) {
					const f32x4 q = *reinterpret_cast<const f32x4 *>(qtile + off + (g * 16 + c0 + c) * 16);
#pragma unroll
					for (int s = 0; s < 4; s++) acc[c][g] = __builtin_fma((double)x[s], (double)q[s], acc[c][g]);
				}
			}
		}
	} else {
#pragma unroll 1
		for (; k0 + 8 <= d4; k0 += 8) {
			const int64_t off = (int64_t)(k0 >> 5) * 1024 + ((k0 & 31) >> 3) * 256;
			const bf16x8 x = *reinterpret_cast<const bf16x8 *>(tile + off + row * 16);
#pragma unroll
			for (int c = 0; c < NC; c++) {
				const bf16x8 q = *reinterpret_cast<const bf16x8 *>(qtile + off + (c0 + c) * 16);
#pragma unroll
				for (int e = 0; e < 8; e++)
					acc[c][e & 3] = __builtin_fma((double)bf16_bits_to_f32((uint16_t)x[e]), (double)bf16_bits_to_f32((uint16_t)q[e]), acc[c][e & 3]);
			}
		}
	}
#pragma unroll 1
	for (int k = k0; k < d; k++) {   // what is left of the last chunk; features from d4 on all go to the first sum
		const double xv = (double)tile_elem(tile, row, k, prec);
#pragma unroll
		for (int c = 0; c < NC; c++) {
			const double p = xv * (double)tile_elem(qtile, c0 + c, k, prec);
			const int a = k < d4 ? (k & 3) : 0;
			acc[c][0] += a == 0 ? p : 0.0;
			acc[c][1] += a == 1 ? p : 0.0;
			acc[c][2] += a == 2 ? p : 0.0;
			acc[c][3] += a == 3 ? p : 0.0;
		}
	}
#pragma unroll
	for (int c = 0; c < NC; c++) out[c] = (float)((acc[c][0] + acc[c][1]) + (acc[c][2] + acc[c][3]));
}

// (two columns at a time: eight double accumulators; the kernels that call this run beside the scoring kernel and must stay small)
template <int NC>
__device__ __forceinline__ void sim_canon(const uint8_t *__restrict__ tile, int row, const uint8_t *__restrict__ qtile, int c0,
	int d, int prec, float (&out)[NC]) {
	if constexpr (NC == 4) {
		sim_canon_cols<2>(tile, row, qtile, c0, d, prec, out);
		__builtin_amdgcn_sched_barrier(0);
		sim_canon_cols<2>(tile, row, qtile, c0 + 2, d, prec, out + 2);
	} else sim_canon_cols<NC>(tile, row, qtile, c0, d, prec, out);
}

// The same sums for a whole 16 x 16 block of similarities, one wave: rows = 16 token rows (lane l's row l & 15 given by its own
// pointer `xrow` -- the address of the row's first 16 bytes in block 0: a tile row of the contextual layout, or a gathered
// vocabulary row of the static one), columns = the 16 rows of a query tile.  Blocks of both sides are staged through LDS
// VK_CANON_RB KiB at a time with the coalesced loads of the scoring kernel, all of a round in flight together (the direct form above
// waits out one memory round trip per 16 bytes: 1.4 ms per query for the winners of config 2 beside a busy scoring kernel, this
// form 0.2 ms); every lane then reads its row's chunks and its four columns' from LDS.  out[r]: row l & 15 x column 4 (l >> 4) + r,
// unclipped.  nblk: 1 KiB blocks of a tile (bf16: K-steps of 32 features, the last one a half block when `half`; fp32: blocks of 16).
// `lds`: VK_CANON_LDS bytes private to the wave.
#define VK_CANON_RB 4
#define VK_CANON_LDS (2 * VK_CANON_RB * 1024)
typedef __attribute__((address_space(3))) void *vk_lds_ptr;
typedef __attribute__((address_space(1))) const void *vk_glb_ptr;

template <int NC>
__device__ __forceinline__ void sim_canon16_cols(const uint8_t *__restrict__ xrow, const uint8_t *__restrict__ qtile, int nblk, int half,
	int d, int prec, uint8_t *__restrict__ lds, int lane, int c0, float *out) {
	double acc[NC][4];
#pragma unroll
	for (int c = 0; c < NC; c++) { acc[c][0] = 0.0; acc[c][1] = 0.0; acc[c][2] = 0.0; acc[c][3] = 0.0; }
	const int d4 = d & ~3;
	const int lim = d4;   // features below it take the sums k mod 4 (from d on the rows hold zeros: d4 bounds the copies in a half block too)
	const int row = lane & 15, g_l = lane >> 4;
	uint8_t *Xl = lds, *Ql = lds + VK_CANON_RB * 1024;
#pragma unroll 1
	for (int t0 = 0; t0 < nblk; t0 += VK_CANON_RB) {
		const int nb = nblk - t0 < VK_CANON_RB ? nblk - t0 : VK_CANON_RB;
#pragma unroll
		for (int i = 0; i < VK_CANON_RB; i++) {
			// LDS-DMA (global_load_lds_dwordx4: per-lane source, lane l lands at base + 16 l): no staging registers -- the callers run
			// beside the scoring kernel and must stay under 96 VGPRs.  Addresses clamped to the last block (no loads under branches);
			// a half block holds chunks 0 and 1 only: lanes of chunks 2, 3 fetch those again, their LDS slots are zeroed below.
			const int t = t0 + i < nblk ? t0 + i : nblk - 1;
			const int gl = (half && t == nblk - 1) ? (g_l & 1) : g_l;
			__builtin_amdgcn_global_load_lds((vk_glb_ptr)(xrow + (int64_t)t * 1024 + gl * 256), (vk_lds_ptr)(Xl + i * 1024), 16, 0, 0);
			__builtin_amdgcn_global_load_lds((vk_glb_ptr)(qtile + (int64_t)t * 1024 + (gl * 16 + row) * 16), (vk_lds_ptr)(Ql + i * 1024), 16, 0, 0);
		}
		__builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): an LDS-DMA is a pending LDS write on the VM counter
		wave_lds_fence();
		// Rolled loops on purpose (one 16-byte chunk of the row against the NC columns per trip): unrolled, the compiler widens the
		// chunks of a whole block to double before the first multiply (64 more registers).  Features from `lim` on (the d % 4
		// trailing ones, which belong to the first sum) are zeroed here and added after the loop.
		if (prec) {
#pragma unroll 1
			for (int is = 0; is < nb * 4; is++) {   // block i, element s of the lanes' float4: features kb + 4 s + g (k mod 4 = g), s ascending
				const int i = is >> 2, sidx = is & 3, kb = (t0 + i) * 16;
				const bool live = kb + 4 * sidx < lim;
#pragma unroll
				for (int g = 0; g < 4; g++) {
					float x = *reinterpret_cast<const float *>(Xl + i * 1024 + (g * 16 + row) * 16 + sidx * 4);
					x = live ? x : 0.0f;
#pragma unroll
					for (int c = 0; c < NC; c++)
						acc[c][g] = __builtin_fma((double)x, (double)*reinterpret_cast<const float *>(Ql + i * 1024 + (g * 16 + c0 + c) * 16 + sidx * 4), acc[c][g]);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		} else {
#pragma unroll 1
			for (int ig = 0; ig < nb * 8; ig++) {   // half a 16-byte chunk per trip: features k0 .. k0 + 3 (k mod 4 = e)
				const int i = ig >> 3, g = (ig >> 1) & 3, hh = ig & 1;
				const int k0 = (t0 + i) * 32 + g * 8 + hh * 4;   // (chunks 2, 3 of a half block hold copies of chunks 0, 1: k0 >= d_pad >= lim zeroes them)
				const bf16x4 xb = *reinterpret_cast<const bf16x4 *>(Xl + i * 1024 + (g * 16 + row) * 16 + hh * 8);
				float xf[4];
#pragma unroll
				for (int e = 0; e < 4; e++) xf[e] = k0 < lim ? bf16_bits_to_f32((uint16_t)xb[e]) : 0.0f;   // lim is a multiple of 4
#pragma unroll
				for (int c = 0; c < NC; c++) {
					const bf16x4 q = *reinterpret_cast<const bf16x4 *>(Ql + i * 1024 + (g * 16 + c0 + c) * 16 + hh * 8);
#pragma unroll
					for (int e = 0; e < 4; e++)
						acc[c][e] = __builtin_fma((double)xf[e], (double)bf16_bits_to_f32((uint16_t)q[e]), acc[c][e]);
				}
			}
		}
		wave_lds_fence();
	}
#pragma unroll 1
	for (int k = d4; k < d; k++) {   // the d % 4 trailing features, in k order, into the first sum (oracle: `for (; k < d; k++) s0 += ..`)
		const double xv = (double)row_elem(xrow, k, prec);
#pragma unroll
		for (int c = 0; c < NC; c++) acc[c][0] += xv * (double)row_elem(qtile + (c0 + c) * 16, k, prec);
	}
#pragma unroll
	for (int c = 0; c < NC; c++) out[c] = (float)((acc[c][0] + acc[c][1]) + (acc[c][2] + acc[c][3]));
}

__device__ __forceinline__ void sim_canon16(const uint8_t *__restrict__ xrow, const uint8_t *__restrict__ qtile, int nblk, int half,
	int d, int prec, uint8_t *__restrict__ lds, int lane, float (&out)[4]) {
	sim_canon16_cols<4>(xrow, qtile, nblk, half, d, prec, lds, lane, (lane >> 4) * 4, out);
}

// this lane's row pointer for sim_canon16: row (lane & 15) of a contextual tile, or vocabulary entry `id` of the static layout
__device__ __forceinline__ const uint8_t *canon_row_ptr(const uint8_t *__restrict__ tile, int lane) { return tile + (lane & 15) * 16; }
__device__ __forceinline__ const uint8_t *canon_row_ptr_static(const uint8_t *__restrict__ etiles, int tile_bytes, int id) {
	return etiles + (int64_t)(id >> 4) * tile_bytes + (id & 15) * 16;
}

// static layout: the cell of the per-query table (metric/static.cpp:9-78) for vocabulary entry `id` and query columns
// c0 .. c0 + NC - 1 in the canonical arithmetic: cosine, then sim[id(t_j)][j] = 1 (:58-67), then the clip (:75)
template <int NC>
__device__ __forceinline__ void static_sim_canon(const uint8_t *__restrict__ etiles, int tile_bytes, int id, const uint8_t *__restrict__ qtiles,
	int c0, int d, int prec, const int32_t *__restrict__ q_ids, float (&out)[NC]) {
	sim_canon<NC>(etiles + (int64_t)(id >> 4) * tile_bytes, id & 15, qtiles + (int64_t)(c0 >> 4) * tile_bytes, c0 & 15, d, prec, out);
#pragma unroll
	for (int c = 0; c < NC; c++) out[c] = (q_ids && q_ids[c0 + c] == id) ? 1.0f : clip01(out[c]);
}

// TagWeightedSlice::similarity (vectorian/core/cpp/slice/static.h:237-264): S * weight(i, j) with
// weight = t_pos_weights[j] * (pos_s != pos_t ? 1 - penalty : 1); values <= threshold become 0.
__device__ __forceinline__ float tag_weighted(float s, float w, int pos_s, int pos_t, float keep, float thr) {
	float wgt = w;
	if (pos_s != pos_t) wgt *= keep;
	const float sc = s * wgt;
	return sc <= thr ? 0.0f : sc;
}

// Vocabulary transports (both RWMD forms, full WMD) with the tag-weighted similarity over the STATIC layout.  Upstream fills its
// distance matrix over the joint vocabulary with dist(u, v) = dist(v, u) = d, u over the slice's entries, v over the query's
// (alignment/wmd.h:121-133): a cell whose two entries occur in BOTH documents is written twice, the later write -- the larger u --
// wins.  The modifier is asymmetric (the tag weight belongs to the query side), so the two values differ: for a slice token with
// key a and a query token with key b, both keys in both documents and a < b, the surviving value is the similarity of the slice's
// b with the query's a.  With keys (token id, tag) and the universal POS a function of the tag (TaggedTokenFactory,
// bow.h:150-176; spaCy's tag map) that is this cell's own cosine and POS penalty under the tag weight of a instead of b.
// One slice, rows S[u * stride + j] already weighted; the NL lanes l = 0 .. NL - 1 share the rows.  Every lane finds the shared
// query columns by itself (a bitmap of the query's ids keeps the scan to the few tokens that can match): no exchange.
// CANON: the cell's cosine restated in the canonical arithmetic (the winners' and candidates' rows) instead of read from the table.
// RESTORE: the inverse -- the same cells under their own tag weight again.  The scoring kernels keep the rows of the slices of one
// wave in ONE strip: where slices overlap (sliding windows) a row belongs to several slices, and a cell rewritten for one of them
// must not be seen by the others; those waves take their slices in turns: rewrite, evaluate, restore (static_vocab_turns).

// the query columns whose (id, tag) key occurs in the slice; 0 unless there are at least two (with fewer nothing is written twice
// with different values)
__device__ __forceinline__ unsigned long long static_vocab_shared(int len_s, int len_t, const int32_t *__restrict__ tok,
	const int8_t *__restrict__ tag, const uint32_t *__restrict__ bits, const int32_t *qkey) {
	unsigned long long mask = 0;
	for (int u = 0; u < len_s; u++) {
		const int id = tok[u];
		if (!((bits[id >> 5] >> (id & 31)) & 1u)) continue;
		const int key = id * 256 + (tag[u] & 255);
		for (int j = 0; j < len_t; j++) mask |= qkey[j] == key ? 1ull << j : 0ull;
	}
	return (mask & (mask - 1)) ? mask : 0ull;
}

template <int NL, bool CANON = false, bool RESTORE = false>
__device__ __forceinline__ void static_vocab_fixup(float *__restrict__ S, int stride, int len_s, int len_t,
	const int32_t *__restrict__ tok, const int8_t *__restrict__ tag, const int8_t *__restrict__ pos,
	const float *__restrict__ table, int64_t table_stride, const uint32_t *__restrict__ bits, const int32_t *qkey,
	const float *tw, const int32_t *tpos, float keep, float thr, int l,
	const uint8_t *__restrict__ etiles = nullptr, int tile_bytes = 0, const uint8_t *__restrict__ qtiles = nullptr, int d = 0, int prec = 0,
	const int32_t *__restrict__ q_ids = nullptr) {
	const unsigned long long mask = static_vocab_shared(len_s, len_t, tok, tag, bits, qkey);
	if (!mask) return;
	for (int u = l; u < len_s; u += NL) {
		const int id = tok[u];
		if (!((bits[id >> 5] >> (id & 31)) & 1u)) continue;
		const int key = id * 256 + (tag[u] & 255);
		int ft = -1;
		for (int j = len_t - 1; j >= 0; j--) ft = qkey[j] == key ? j : ft;
		if (ft < 0) continue;
		for (int j = 0; j < len_t; j++)
			if (((mask >> j) & 1ull) && qkey[j] > key) {
				float raw;
				if constexpr (CANON) {
					float v1[1];
					static_sim_canon<1>(etiles, tile_bytes, id, qtiles, j, d, prec, q_ids, v1);
					raw = v1[0];
				} else raw = table[(int64_t)(j >> 4) * table_stride + (int64_t)id * 16 + (j & 15)];
				S[u * stride + j] = tag_weighted(raw, tw[RESTORE ? j : ft], pos[u], tpos[j], keep, thr);
			}
	}
}


// ---------------------------------------------------------------------------
// DP over a group of 4 sentences: DPP row sigma = lane >> 4 is one sentence, lane
// v = lane & 15 is query column v + 1.  Rows (sentence tokens) are swept serially;
// within a row the left-to-right dependency is resolved by a monotone fixpoint chain
// on DPP row_shr:1, which reproduces the sequential recurrence bit for bit (max is
// exact and x -> x - g is monotone).
//
// S: wave-private LDS [rows][LT] (the LT = 4/8/12/16 padded query columns); sentence sigma's token i is row rowbase + i.
// Lanes v >= LT read into the next row: their values never reach a lower lane.
// Returns the aligner score (raw) in lane 15 of each DPP row.
// ---------------------------------------------------------------------------

struct DpArgs {
	int32_t locality;
	int32_t len_t;
	float gs, gt;          // linear: w(k) = g*k ; affine: extension cost b
	float a_s, a_t;        // affine: a
	float open_s, open_t;  // affine: a + b
	const float *ws;       // general: w_s[0..max_len]
	const float *wt;       // general: w_t[0..16] for the in-row candidates (the closure of the caller's table, vk_query.cpp)
	const float *wt0;      // general: the caller's table (border row H[0][j] = -w_t(j), set directly, never chained)
	int32_t rwmd_symmetric, rwmd_normalize_bow, wmd_bound;
	float wrd_raw_total;   // WRD on raw magnitudes: sum of the query's magnitudes (0: masses are normalised)
};

// In-row dependency of the linear recurrence H[u][j] = max(c[j], H[u][j-1] - gt): unrolled,
// H[u][j] = max_k (c[j-k] - k gt), a prefix maximum with decay.  It is taken in log2(LT) doubling steps
// x <- max(x, row_shr:s(x) - s gt), s = 1, 2, 4, 8 (s gt is exact for powers of two), ten instructions
// instead of a chain of LT dependent subtractions.  The value may differ from the sequential
// recurrence in the last bit (c - 2 gt is rounded once, (c - gt) - gt twice); scores are compared at
// 1e-4, and the tracebacks of the winners come from vk_flow_kernel, which walks the recurrence
// sequentially.  The border column enters as H[u][0] - (v + 1) gt (non-LOCAL only).
// The decays per lane: s g where column v - s exists, +inf where it does not, so that each step is one
// v_sub_f32_dpp with zero fill (0 - inf = -inf drops out of the maximum) and one v_max_f32.
struct DecaySteps { float g1, g2, g4, g8; };

__device__ __forceinline__ DecaySteps decay_steps(float g, int v) {
	const float inf = __builtin_inff();
	return {v >= 1 ? g : inf, v >= 2 ? 2.0f * g : inf, v >= 4 ? 4.0f * g : inf, v >= 8 ? 8.0f * g : inf};
}

template <int LT>
__device__ __forceinline__ float decay_scan(float x, const DecaySteps &d) {
	x = fmaxf(x, dpp_zero<0x111>(x) - d.g1);
	if (LT > 2) x = fmaxf(x, dpp_zero<0x112>(x) - d.g2);
	if (LT > 4) x = fmaxf(x, dpp_zero<0x114>(x) - d.g4);
	if (LT > 8) x = fmaxf(x, dpp_zero<0x118>(x) - d.g8);
	return x;
}

template <int LT>
__device__ __forceinline__ float dp_linear(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const float gsb = is_global ? a.gs : 0.0f;   // border H[u][0] = -(gs*u) (GLOBAL) else 0
	const float gs = a.gs, gt = a.gt;
	const bool last_col = v == a.len_t - 1;
	const float gt_v1 = gt * (float)(v + 1);     // distance of this column from the border column

	const DecaySteps dt = decay_steps(gt, v);

	float h = is_global ? -gt_v1 : 0.0f;  // H[0][v+1]
	float best = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = -(gsb * (float)(u - 1));
		const float bcur = -(gsb * (float)u);
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		float c = fmaxf(diag + s, floor0);
		c = fmaxf(c, h - gs);
		float hn = decay_scan<LT>(c, dt);
		if (!is_local) hn = fmaxf(hn, bcur - gt_v1);
		h = act ? hn : h;
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;   // last row U last column U border 0
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// Gotoh, w(k) = a + b*k: E (gap over s tokens) lives in the lane, F (gap over query
// tokens) is resolved with the same fixpoint chain as H.
template <int LT>
__device__ VK_DP_INLINE float dp_affine(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const float bs = a.gs, bt = a.gt, open_s = a.open_s, open_t = a.open_t;
	const float a_s = a.a_s, a_t = a.a_t;
	const bool last_col = v == a.len_t - 1;

	// borders (GLOBAL): H[0][j] = -(a_t + bt*j), H[u][0] = -(a_s + bs*u)
	float h = is_global ? -(a_t + bt * (float)(v + 1)) : 0.0f;
	float e = VK_NEG_INF;                       // E[0][j]
	float best = 0.0f;
	const DecaySteps dt = decay_steps(bt, v);
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = (is_global && u > 1) ? -(a_s + bs * (float)(u - 1)) : 0.0f;
		const float bcur = is_global ? -(a_s + bs * (float)u) : 0.0f;
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		const float en = fmaxf(h - open_s, e - bs);     // E[u][j]
		float c = fmaxf(fmaxf(diag + s, floor0), en);
		// F[u][j] = max(H[u][j-1] - open_t, F[u][j-1] - bt); F[u][0] = -inf
		float f = VK_NEG_INF;
		float hc = c;
		if (a_t >= 0.0f) {
			// opening costs at least an extension, so extending a gap never loses against reopening it from the same
			// cell: F[u][j] = max_k (c[j-k] - open_t - (k-1) bt), the decayed prefix maximum of c shifted by one
			// column (the border column enters at lane 0).  Last-bit differences as in dp_linear.
			f = decay_scan<LT>(dpp_f<DPP_ROW_SHR1>(bcur, c) - open_t, dt);
			hc = fmaxf(c, f);
		} else {
#pragma unroll
			for (int i = 0; i < LT; i++) {
				const float fl = fmaxf(dpp_f<DPP_ROW_SHR1>(bcur, hc) - open_t, dpp_f<DPP_ROW_SHR1>(VK_NEG_INF, f) - bt);
				f = fl;
				hc = fmaxf(c, f);
			}
		}
		if (act) { h = hc; e = en; }
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// General gap costs (Waterman-Smith-Beyer): the column history H[0..u-1][j] of each
// lane is staged in wave-private LDS (Hh[sigma][u][v]); the in-row dependency walks
// the source columns left to right with ds_bpermute broadcasts.
template <int LT>
__device__ __forceinline__ float dp_general(const float *__restrict__ S, float *__restrict__ Hh, int hstride,
	int rowbase, int len, int maxlen, int lane, const DpArgs &a) {

	const int v = lane & 15, sigma = lane >> 4;
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = v == a.len_t - 1;
	float *hist = Hh + sigma * hstride * 16 + v;   // hist[u*16] = H[u][v+1]

	// wtl[p] = w_t(j - p) for source column p < j = v + 1, else +inf (no candidate)
	float wtl[LT];
#pragma unroll
	for (int p = 0; p < LT; p++) wtl[p] = (p <= v) ? a.wt[v + 1 - p] : __builtin_inff();

	float h = is_global ? -a.wt0[v + 1] : 0.0f;
	hist[0] = h;
	float best = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = is_global ? -a.ws[u - 1] : 0.0f;    // ws[0] = 0
		const float bcur = is_global ? -a.ws[u] : 0.0f;
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		float c = fmaxf(diag + s, floor0);
		// gaps over s tokens: H[u-k][j] - w_s(k), k = 1..u (each lane re-reads its own column)
		for (int k = 1; k <= u; k++) c = fmaxf(c, hist[(u - k) * 16] - a.ws[k]);
		// gaps over query tokens: H[u][p] - w_t(j - p), p = 0 (border) .. j-1
		c = fmaxf(c, bcur - wtl[0]);
#pragma unroll
		for (int p = 1; p < LT; p++) {
			const float hp = __builtin_bit_cast(float,
				__builtin_amdgcn_ds_bpermute((sigma * 16 + p - 1) * 4, __builtin_bit_cast(int, c)));
			c = fmaxf(c, hp - wtl[p]);
		}
		if (act) { h = c; hist[u * 16] = c; }
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// General gap costs, fast form: sentences of at most MAXLEN tokens, the column history
// H[0..u-1][j] lives in registers (rows fully unrolled, w_s in scalar registers), and
// the in-row step takes its candidates from the row's values *before* in-row gaps,
//     H[u][j] = max(c[j], max_k c[j-k] - w_t(k)),   c = max(zero, diagonal, gaps over s tokens)
// which needs no serial chain.  In the sequential recurrence an in-row candidate H[u][j-k] - w_t(k) may itself end in an in-row
// gap: two gaps in a row cost w_t(a) + w_t(b), so the recurrence works with the cheapest way of composing a gap,
//     w*(k) = min(w_t(k), min_{a+b=k} w*(a) + w*(b))    (the subadditive closure of w_t),
// and H[u][j] = max(c[j], max_k c[j-k] - w*(k)) exactly.  The host hands over w* as `wt` (vk_query.cpp; it equals w_t for
// subadditive costs such as 1 - 2^(-k/c)); a composed gap is rounded once here and twice in the chain, a difference of an ulp,
// far inside the 1e-4 of the scores -- the tracebacks of the winners walk the caller's table sequentially (vk_flow_kernel).
// The border row is set directly: H[0][j] = -w_t(j), one gap, from `wt0`.
template <int LT, int MAXLEN>
__device__ __forceinline__ float dp_general_reg(const float *__restrict__ S, int rowbase, int len, int maxlen, int v,
	const DpArgs &a, const float (&wsr)[MAXLEN + 1], const float (&wtr)[LT]) {

	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = v == a.len_t - 1;
	const float wt_border = a.wt[v + 1];             // distance from the border column to column v + 1 (chains of gaps allowed: closure)
	const float wt_border0 = a.wt0[v + 1];           // ... as one gap: the border row

	// in-row gap costs per lane: w_t(k) where column v - k exists, +inf where it does not.  The candidate is then ONE
	// v_sub_f32_dpp (zero fill for the missing source lanes: 0 - inf = -inf drops out of the maximum) instead of a
	// preset, a DPP move and a subtract.
	float wtv[16];
#pragma unroll
	for (int k = 1; k < LT; k++) wtv[k] = v >= k ? wtr[k] : __builtin_inff();

	float hreg[MAXLEN + 1];
	float h = is_global ? -wt_border0 : 0.0f;
	hreg[0] = h;
	float best = 0.0f;
#pragma unroll
	for (int u = 1; u <= MAXLEN; u++) {
		if (u <= maxlen) {
			const bool act = u <= len;
			const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
			const float bprev = is_global ? -wsr[u - 1] : 0.0f;
			const float bcur = is_global ? -wsr[u] : 0.0f;
			const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hreg[u - 1]);
			float c = fmaxf(diag + s, floor0);
#pragma unroll
			for (int k = 1; k <= u; k++) c = fmaxf(c, hreg[u - k] - wsr[k]);
			float hc = fmaxf(c, bcur - wt_border);
			if (LT > 1) hc = fmaxf(hc, dpp_zero<0x111>(c) - wtv[1]);
			if (LT > 2) hc = fmaxf(hc, dpp_zero<0x112>(c) - wtv[2]);
			if (LT > 3) hc = fmaxf(hc, dpp_zero<0x113>(c) - wtv[3]);
			if (LT > 4) hc = fmaxf(hc, dpp_zero<0x114>(c) - wtv[4]);
			if (LT > 5) hc = fmaxf(hc, dpp_zero<0x115>(c) - wtv[5]);
			if (LT > 6) hc = fmaxf(hc, dpp_zero<0x116>(c) - wtv[6]);
			if (LT > 7) hc = fmaxf(hc, dpp_zero<0x117>(c) - wtv[7]);
			if (LT > 8) hc = fmaxf(hc, dpp_zero<0x118>(c) - wtv[8]);
			if (LT > 9) hc = fmaxf(hc, dpp_zero<0x119>(c) - wtv[9]);
			if (LT > 10) hc = fmaxf(hc, dpp_zero<0x11a>(c) - wtv[10]);
			if (LT > 11) hc = fmaxf(hc, dpp_zero<0x11b>(c) - wtv[11]);
			if (LT > 12) hc = fmaxf(hc, dpp_zero<0x11c>(c) - wtv[12]);
			if (LT > 13) hc = fmaxf(hc, dpp_zero<0x11d>(c) - wtv[13]);
			if (LT > 14) hc = fmaxf(hc, dpp_zero<0x11e>(c) - wtv[14]);
			if (LT > 15) hc = fmaxf(hc, dpp_zero<0x11f>(c) - wtv[15]);
			hreg[u] = hc;
			h = act ? hc : h;
			if (is_local || last_col) best = fmaxf(best, h);
		}
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// Relaxed Word Mover's Distance, injective form (vectorian/core/cpp/alignment/wmd.h:287-416 with
// the BOW builders of alignment/bow.h:204-333).  For the injective solver every vocabulary entry
// moves its whole mass to its nearest partner, so repeated tokens contribute the same distance
// once per occurrence and the joint-vocabulary formulation reduces to positions:
//   acc0 = sum_j w_t * min_i D[i][j]   (t -> s, wmd.h:303-306 computes this direction first)
//   acc1 = sum_i w_s * min_j D[i][j]   (s -> t)
// D = max(1 - S, 0) (wmd.h:107-135); nbow: w = 1/len, bow: w = 1 and acc /= len (wmd.h:379-381);
// cost = acc0, or max(acc0, acc1) when symmetric (wmd.h:383-390); score = (max_cost - cost)/max_cost
// with max_cost = 1 (nbow) or len_t (wmd.h:411-415).  Sums run in position order in fp32.
template <int LT>
__device__ __forceinline__ float rwmd_rows(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const int len_t = a.len_t;
	const bool nbow = a.rwmd_normalize_bow != 0;
	const float w_t = nbow ? 1.0f / (float)len_t : 1.0f;
	const float w_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;
	const bool col_ok = v < len_t;
	float colmin = 3.402823466e+38F;
	float acc1 = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float dist = fmaxf(1.0f - s, 0.0f);
		if (act) colmin = fminf(colmin, dist);
		// min over the query columns of this row -> lane 15
		float m = col_ok ? dist : 3.402823466e+38F;
		m = fminf(m, dpp_f<DPP_ROW_SHR1>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR2>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR4>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR8>(m, m));
		if (act) acc1 += w_s * m;
	}
	// acc0: sequential sum over the query columns, lane by lane
	const float x = col_ok ? w_t * colmin : 0.0f;
	float sum = x;
#pragma unroll
	for (int i = 1; i < LT; i++) {
		const float t = dpp_f<DPP_ROW_SHR1>(0.0f, sum);
		if (v == i) sum = t + x;
	}
	// lane len_t - 1 holds acc0; move it to lane 15 with a max-reduction over one valid lane
	float acc0 = (v == len_t - 1) ? sum : VK_NEG_INF;
	acc0 = row_max_to_lane15(acc0);
	if (!nbow) {
		acc0 = acc0 / (float)len_t;
		acc1 = acc1 / (float)(len > 0 ? len : 1);
	}
	if (a.wmd_bound) {
		// stage 1 of the full WMD: every unit of the side that is shipped completely travels at least to
		// its nearest partner, so 1 - (that side's relaxed cost) bounds the score from above.  nbow: both
		// sides ship everything; bow (unit masses): the shorter side does.
		float lb;
		if (a.wmd_bound == 1) lb = fmaxf(acc0, acc1);
		else lb = len_t <= len ? acc0 : acc1;
		return fminf(1.0f - lb + 3e-5f, 1.0f);
	}
	float cost = 0.0f;
	if (a.rwmd_symmetric) {
		if (acc0 > cost) cost = acc0;
		if (acc1 > cost) cost = acc1;
	} else {
		cost = acc0;
	}
	const float max_cost = nbow ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

// Relaxed Word Mover's Distance, 1:n form (rwmd('nbow/distributed'); RelaxedSolver with injective = false,
// vectorian/core/cpp/alignment/wmd.h:339-376): every source ships its mass to the targets in order of
// ascending distance, each target taking at most its own mass; as upstream is written, whatever the last
// (partial) shipment carried is charged once more at the maximum distance 1 (wmd.h:373-375 keeps
// `remaining` after the break).  Sources and targets are vocabulary entries: q_mass / smass hold count / len
// at the first occurrence of a token and 0 at its repetitions, which then neither ship nor receive
// (contextual layout: every position is an entry of its own).
//   direction 0 (t -> s): lane = source (query token); its column is consumed in ascending (distance, row)
//     order, one selection sweep over the rows per target taken;
//   direction 1 (s -> t): row = source; the 16 lanes of the DPP row are the targets, selected by a row
//     minimum over keys (distance bits with the lane in the low 4 bits: distances closer than 2^-19
//     relative may swap, which moves the cost by less than 1e-7).
template <int LT>
__device__ __forceinline__ float rwmd_fill_rows(const float *__restrict__ S, const float *__restrict__ smass, int rowbase, int len, int maxlen,
	int lane, const DpArgs &a, float q_mass) {
	const int v = lane & 15, sigma = lane >> 4;
	const int len_t = a.len_t;
	const bool nbow = a.rwmd_normalize_bow != 0;
	const bool col_ok = v < len_t;
	const float cap_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;   // capacity of a slice position
	const float INF = __builtin_inff();

	// ---- direction 0
	float rem = col_ok ? q_mass : 0.0f, cost0 = 0.0f, last_d = -1.0f;
	int last_i = -1;
	bool fin = !(rem > 0.0f) || len < 1;
	for (int round = 0; round < maxlen && __any(!fin); round++) {
		float bd = INF;
		int bi = -1;
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * LT + v], 0.0f);
			const bool later = dist > last_d || (dist == last_d && u - 1 > last_i);
			if (act && later && dist < bd) { bd = dist; bi = u - 1; }
		}
		if (!fin) {
			// capacity of the target: the mass of its vocabulary entry (0 at the repetitions of a token)
			const float cap = (smass && bi >= 0) ? smass[bi] : cap_s;
			if (bi < 0) fin = true;
			else if (rem <= cap) { cost0 += rem * bd; fin = true; }
			else { rem -= cap; cost0 += cap * bd; last_d = bd; last_i = bi; }
		}
	}
	if (rem > 0.0f) cost0 += rem;
	// sum over the query columns in order, lane by lane (as rwmd_rows)
	float sum = cost0;
#pragma unroll
	for (int i = 1; i < LT; i++) {
		const float t = dpp_f<DPP_ROW_SHR1>(0.0f, sum);
		if (v == i) sum = t + cost0;
	}
	float acc0 = (v == len_t - 1) ? sum : VK_NEG_INF;
	acc0 = row_max_to_lane15(acc0);

	// ---- direction 1
	float acc1 = 0.0f;
	if (a.rwmd_symmetric) {
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * LT + v], 0.0f);
			float r1 = act ? (smass ? smass[u - 1] : cap_s) : 0.0f;
			float cost = 0.0f;
			bool used = !col_ok, done = !(r1 > 0.0f);
			for (int r = 0; r < LT && __any(!done); r++) {
				int key = used ? 0x7fffffff : ((__builtin_bit_cast(int, dist) & ~15) | v);
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x141, 0xf, 0xf, false));   // row_half_mirror
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x140, 0xf, 0xf, false));   // row_mirror
				const int tl = key & 15;
				const float td = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((sigma * 16 + tl) * 4, __builtin_bit_cast(int, dist)));
				const float tc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((sigma * 16 + tl) * 4, __builtin_bit_cast(int, q_mass)));
				if (!done) {
					if (key == 0x7fffffff) done = true;
					else if (r1 <= tc) { cost += r1 * td; done = true; }
					else { r1 -= tc; cost += tc * td; }
				}
				if (v == tl) used = true;
			}
			if (r1 > 0.0f) cost += r1;
			acc1 += cost;
		}
	}
	if (!nbow) {
		acc0 = acc0 / (float)len_t;
		acc1 = acc1 / (float)(len > 0 ? len : 1);
	}
	float cost = acc0;
	if (a.rwmd_symmetric) { cost = 0.0f; if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
	const float max_cost = nbow ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

// Word Rotator's Distance, stage 1: an upper bound of the score for every sentence.
// WRD = 1 - EMD of the transport problem with masses |x| / sum|x| and costs max(0, 1 - S)
// (vectorian/core/cpp/alignment/wrd.h:62-146).  Every unit of mass travels at least to its nearest
// partner, so both  sum_j m_t[j] min_i C[j][i]  and  sum_i m_s[i] min_j C[j][i]  bound the EMD from
// below; the larger of the two (minus a margin for fp32 rounding) gives score <= 1 - LB.
// Stage 2 (vk_wrd_exact_kernel) solves the survivors exactly.
template <int LT>
__device__ __forceinline__ float wrd_bound_rows(const float *__restrict__ S, int rowbase, int len, int maxlen, int v,
	const DpArgs &a, const float *__restrict__ mag, const int32_t *__restrict__ ids, float q_mass) {
	const int len_t = a.len_t;
	const bool col_ok = v < len_t;
	// one sweep: sum of the slice's magnitudes and sum of magnitude x nearest distance (the quotient is the
	// bound of the s -> t direction; it is only a bound, so the order of the float operations is free)
	// t -> s direction: a query token's mass cannot all go to its nearest slice token -- that one takes at most its
	// own mass.  The lane keeps the four nearest (distance, mass) pairs of its column; filling them in order and
	// charging what is left at the fourth distance bounds the cost of any feasible plan from below (a capacity-
	// constrained relaxation, ICT / ACT of Atasu & Mittelholzer): far tighter than the nearest-neighbour bound when a
	// query token weighs several slice tokens (10 against 32 tokens), and the exact stage sees that many fewer rows.
	float sum_s = 0.0f;
	float lb1n = 0.0f;
	const float BIG = 3.402823466e+38F;
	float d1 = BIG, d2 = BIG, d3 = BIG, d4 = BIG, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, c4 = 0.0f;
	// the magnitudes of the slice (at most 64 tokens on this path) are fetched once, lane v holding rows v, v + 16,
	// v + 32, v + 48, and handed out row by row with ds_bpermute: a global load per row inside the sweep left the wave
	// waiting on memory latency 64 times per group (static layout: magnitude of the vocabulary entry; no magnitudes:
	// unit masses -- bags of words, full WMD)
	// (slices of more than 64 tokens -- the one-slice-per-wave pass -- go through the same sweep 64 rows at a time)
	const int lane_row = (int)(threadIdx.x & 48);
	for (int base = 0; base < maxlen; base += 64) {
		float mreg[4];
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const int u0 = base + c * 16 + v;
			float mg = 0.0f;
			if (u0 < len) mg = mag ? (ids ? mag[ids[u0]] : mag[u0]) : 1.0f;
			mreg[c] = mg;
		}
#pragma unroll
		for (int c = 0; c < 4; c++) {
			const int u_end = maxlen < base + c * 16 + 16 ? maxlen : base + c * 16 + 16;
			for (int u = base + c * 16 + 1; u <= u_end; u++) {
				const bool act = u <= len;
				const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
				const float dist = fmaxf(1.0f - s, 0.0f);
				// minimum over the query columns -> lane 15; distances are >= 0, so the integer order of the bits is theirs
				// (v_min_i32 takes the DPP operand directly, fminf would canonicalise first)
				int mi = __builtin_bit_cast(int, col_ok ? dist : BIG);
				mi = min(mi, __builtin_amdgcn_update_dpp(mi, mi, DPP_ROW_SHR1, 0xf, 0xf, false));
				mi = min(mi, __builtin_amdgcn_update_dpp(mi, mi, DPP_ROW_SHR2, 0xf, 0xf, false));
				mi = min(mi, __builtin_amdgcn_update_dpp(mi, mi, DPP_ROW_SHR4, 0xf, 0xf, false));
				mi = min(mi, __builtin_amdgcn_update_dpp(mi, mi, DPP_ROW_SHR8, 0xf, 0xf, false));
				const float m = __builtin_bit_cast(float, mi);
				const float mg = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane_row | ((u - 1) & 15)) * 4, __builtin_bit_cast(int, mreg[c])));
				sum_s += mg;
				lb1n += mg * m;
				// insert (dist, mg) into the sorted four
				float nd = act ? dist : BIG, nc = mg;
				bool sw;
				sw = nd < d4; d4 = sw ? nd : d4; c4 = sw ? nc : c4;
				sw = d4 < d3; nd = d3; nc = c3; d3 = sw ? d4 : d3; c3 = sw ? c4 : c3; d4 = sw ? nd : d4; c4 = sw ? nc : c4;
				sw = d3 < d2; nd = d2; nc = c2; d2 = sw ? d3 : d2; c2 = sw ? c3 : c2; d3 = sw ? nd : d3; c3 = sw ? nc : c3;
				sw = d2 < d1; nd = d1; nc = c1; d1 = sw ? d2 : d1; c1 = sw ? c2 : c1; d2 = sw ? nd : d2; c2 = sw ? nc : c2;
			}
		}
	}
	const float lb1 = lb1n / sum_s * (1.0f - 2e-6f);
	// masses on the scale of q_mass: normalised (shares of sum_s) unless the magnitudes are used as they are
	const float scale = a.wrd_raw_total > 0.0f ? 1.0f : 1.0f / sum_s;
	float rem = col_ok ? q_mass : 0.0f, x = 0.0f, last = 0.0f;
	{
		float amt;
		if (d1 < BIG) { amt = fminf(rem, c1 * scale * (1.0f + 2e-6f)); x += amt * d1; rem -= amt; last = d1; }
		if (d2 < BIG) { amt = fminf(rem, c2 * scale * (1.0f + 2e-6f)); x += amt * d2; rem -= amt; last = d2; }
		if (d3 < BIG) { amt = fminf(rem, c3 * scale * (1.0f + 2e-6f)); x += amt * d3; rem -= amt; last = d3; }
		if (d4 < BIG) { amt = fminf(rem, c4 * scale * (1.0f + 2e-6f)); x += amt * d4; rem -= amt; last = d4; }
		x += fmaxf(rem, 0.0f) * last;
	}
	// sum over lanes, any order: it is only a bound
	x += dpp_f<DPP_ROW_SHR1>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR2>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR4>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR8>(0.0f, x);
	float lb;
	if (a.wrd_raw_total > 0.0f) {
		// magnitudes as they are (normalize_magnitudes = false): min(sum_t, sum_s) units are shipped and the score
		// is 1 - cost / shipped; only the lighter side ships everything, so only its relaxed cost is a bound
		const float sum_t = a.wrd_raw_total;
		lb = (sum_t <= sum_s ? x / sum_t : lb1) * (1.0f - 2e-6f);
		if (!(sum_s > 0.0f)) lb = 0.0f;
	} else lb = fmaxf(x, lb1);
	return fminf(1.0f - lb + 3e-5f, 1.0f);
}

#endif
