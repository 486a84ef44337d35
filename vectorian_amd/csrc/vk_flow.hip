// vk_flow.hip -- traceback of the winners (vk_flow_kernel) and the one-wave-per-slice kernel for long queries.
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// flow of the winners: one wave per winner restates the similarity rows in the canonical
// arithmetic (sim_canon, vk_common.hip.h: the oracle's own sums, bit for bit -- the scoring
// kernel's MFMA cosines differ from them in the last bit, which is enough to flip the
// traceback between co-optimal alignments), then runs the sequential DP with traceback
// exactly as the oracle states it (vko_align in oracle/vk_oracle.c): the aligner score,
// the mapping and the edge similarities of a winner are the oracle's, bit for bit.
// candidates zero (LOCAL), diagonal, gap in s (k = 1..), gap in t (k = 1..), replace
// on strictly greater; start cell = first maximum in row-major order.
// Output: mapping[j] = matched sentence token or -1 (InjectiveFlow,
// metric/alignment.h:194-196), edge_sim[j] = S[mapping[j]][j] (metric/alignment.h:335-345).
// ---------------------------------------------------------------------------

#define VK_TB_W 17

// dynamic LDS of vk_flow_kernel for slices of at most max_len tokens (bytes); the carve-up below must match
static inline size_t vk_flow_lds_bytes(int max_len, bool tagged) {
	const size_t rows = (size_t)max_len + 1, srows = (size_t)max_len + 32;
	size_t b = VK_CANON_LDS;                           // staging of sim_canon16
	b += srows * 16 * 4 * (tagged ? 2 : 1);            // S (+ SW)
	b += rows * VK_TB_W * 4;                           // H
	b += (rows + 3) / 4 * 4 * 4 + 32 * 4;              // wsl, wtl
	b += rows * VK_TB_W * 2;                           // dk
	b += rows * VK_TB_W;                               // flags
	return (b + 15) / 16 * 16;
}

// PREC: bf16 / fp32 rows at compile time -- the kernel has to fit into the registers the next query's scoring kernel leaves free
// (three waves per SIMD of 136 VGPRs leave 104; the fp32 scoring kernel's 126 leave 134): 64 VGPRs for bf16 rows, the fp32 form
// likewise (five waves per SIMD asked of the register allocator: at most 96).
template <int PREC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6))) void vk_flow_kernel(VkFlowParams p) {
	extern __shared__ float4 vk_smem4[];
	const int rows = p.max_len + 1, srows = p.max_len + 32;
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem4);
	float *S = reinterpret_cast<float *>(canon + VK_CANON_LDS);
	float *SW = p.pos_s ? S + srows * 16 : S;            // tag-weighted copy the DP runs on (else S itself)
	float *H = SW + srows * 16;
	float *wsl = H + rows * VK_TB_W;
	float *wtl = wsl + (rows + 3) / 4 * 4;
	int16_t *dk = reinterpret_cast<int16_t *>(wtl + 32);
	uint8_t *flags = reinterpret_cast<uint8_t *>(dk + rows * VK_TB_W);   // bits 0-1 direction, 2 E extended, 3 F extended

	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	const uint64_t key = p.keys[w];
	if (key == 0) return;   // fewer than k admitted
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int len_s = t_b - t_a, len_t = p.len_t;

	// S: unmodified similarities (reported per edge, metric/alignment.h:339); SW: what the DP runs on
	const int is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	const int tile0 = is_static ? 0 : t_a >> 4;
	const int ntiles = is_static ? (len_s + 15) >> 4 : ((t_b + 15) >> 4) - tile0;
	for (int ti = 0; ti < ntiles; ti++) {
		// 16 rows x 16 columns per step: lane l -> row l & 15, columns 4 (l >> 4) .. + 3 (static layout: the rows of 16
		// consecutive tokens of the slice, gathered from the vocabulary; past the end: the slice's first token again, never read)
		const int tok = is_static ? t_a + (ti * 16 + (lane & 15) < len_s ? ti * 16 + (lane & 15) : 0) : (tile0 + ti) * 16 + (lane & 15);
		const int id = is_static ? p.tok_id[tok] : 0;
		const uint8_t *xrow = is_static ? canon_row_ptr_static(p.tiles, p.tile_bytes, id) : canon_row_ptr(p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, lane);
		const int cb = (lane >> 4) * 4;
		float val[4];
		sim_canon16(xrow, p.qtile, p.nk32, p.tail, p.d, PREC, canon, lane, val);
#pragma unroll
		for (int r = 0; r < 4; r++) val[r] = (is_static && p.q_ids && p.q_ids[cb + r] == id) ? 1.0f : clip01(val[r]);   // sim[id(t_j)][j] = 1 (metric/static.cpp:58-67)
		*reinterpret_cast<float4 *>(S + (ti * 16 + (lane & 15)) * 16 + cb) = make_float4(val[0], val[1], val[2], val[3]);
		if (p.pos_s) {
			const int ps = p.pos_s[tok];
#pragma unroll
			for (int r = 0; r < 4; r++) val[r] = tag_weighted(val[r], p.tw[cb + r], ps, p.tpos[cb + r], p.tw_keep, p.tw_threshold);
			*reinterpret_cast<float4 *>(SW + (ti * 16 + (lane & 15)) * 16 + cb) = make_float4(val[0], val[1], val[2], val[3]);
		}
	}
	const int rowbase = is_static ? 0 : t_a - tile0 * 16;
	// gap tables into LDS (uniform broadcast reads in the candidate loops)
	for (int i = lane; i <= p.max_len; i += 64) wsl[i] = p.ws[i];
	if (lane <= VK_DEV_MAX_QUERY_LEN) wtl[lane] = p.wt[lane];
	__syncthreads();

	const float *Sm = SW + rowbase * 16;      // DP input
	const float *Su = S + rowbase * 16;       // unmodified, for the edges
	const int W = VK_TB_W;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const int gap = p.gap_mode;
	const float gs = p.gs, gt = p.gt, open_s = p.open_s, open_t = p.open_t, a_s = p.a_s, a_t = p.a_t;

	// ---- fill: lane l owns query column v = l + 1 (lanes 0..len_t-1, one DPP row).  Per row the
	// zero / diagonal / gap-in-s candidates of all columns are evaluated in parallel; the gap-in-t
	// candidates need the final cells to the left, which become final one column per step and are
	// broadcast with v_readlane (the wave holds ONE sentence, so the column index is wave-uniform).
	// Candidate order and strict-greater replacement are the oracle's (vko_align): zero, diagonal,
	// gap in s with k = 1, 2, .., gap in t with k = 1, 2, ..  In-row candidates arrive with k
	// descending, so among them ">=" keeps the smallest k of the maximum, and the winner replaces
	// the earlier candidates only if strictly greater.
	const int v = lane + 1;
	const bool col = v <= len_t;
	float wrel[16];   // general: w_t(v - p) for source column p < v
#pragma unroll
	for (int pp = 0; pp < 16; pp++) wrel[pp] = (col && pp < v) ? wtl[v - pp] : __builtin_inff();

	float hprev = 0.0f, eprev = VK_NEG_INF;
	if (global && col) hprev = gap == 0 ? -(gt * (float)v) : gap == 1 ? -(a_t + gt * (float)v) : -wtl[v];
	if (col) H[v] = hprev;
	for (int u = 1; u <= len_s; u++) {
		float bprev = 0.0f, bcur = 0.0f;
		if (global) {
			bprev = u == 1 ? 0.0f : (gap == 0 ? -(gs * (float)(u - 1)) : gap == 1 ? -(a_s + gs * (float)(u - 1)) : -wsl[u - 1]);
			bcur = gap == 0 ? -(gs * (float)u) : gap == 1 ? -(a_s + gs * (float)u) : -wsl[u];
		}
		const float sv = Sm[(u - 1) * 16 + (col ? v - 1 : 0)];
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hprev);
		float best, e = VK_NEG_INF;
		uint8_t d, ee = 0, fe = 0;
		int16_t kk = 0;
		float c = diag + sv;
		if (local) { best = 0.0f; d = 0; if (c > best) { best = c; d = 1; } }
		else { best = c; d = 1; }
		if (gap == 0) {
			c = hprev - gs;
			if (c > best) { best = c; d = 2; kk = 1; }
		} else if (gap == 1) {
			e = hprev - open_s;
			c = eprev - gs;
			if (c > e) { e = c; ee = 1; }
			if (e > best) { best = e; d = 2; }
		} else {
			for (int k = 1; k <= u; k++) {
				c = H[(u - k) * W + (col ? v : 1)] - wsl[k];
				if (c > best) { best = c; d = 2; kk = (int16_t)k; }
			}
		}
		// in-row candidates
		float left_best = VK_NEG_INF, f = VK_NEG_INF, fin = best, ffin = VK_NEG_INF;
		int16_t left_k = 0;
#pragma unroll
		for (int pp = 0; pp < 16; pp++) {
			if (pp < len_t) {
				const float sp = pp == 0 ? bcur : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), pp - 1));
				const float fp = pp == 0 ? VK_NEG_INF : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ffin), pp - 1));
				if (gap == 2) {
					const float cc = sp - wrel[pp];
					if (v > pp && cc >= left_best) { left_best = cc; left_k = (int16_t)(v - pp); }
				} else if (v == pp + 1) {
					if (gap == 0) { left_best = sp - gt; left_k = 1; }
					else {
						f = sp - open_t;
						const float c2 = fp - gt;
						if (c2 > f) { f = c2; fe = 1; }
					}
				}
				if (v == pp + 1) {   // all sources of this column are in: finalise it
					if (gap == 1) { if (f > best) { best = f; d = 3; } ffin = f; }
					else if (left_best > best) { best = left_best; d = 3; kk = left_k; }
					fin = best;
				}
			}
		}
		if (col) {
			H[u * W + v] = best;
			dk[u * W + v] = kk;
			flags[u * W + v] = (uint8_t)(d | (ee << 2) | (fe << 3));
		}
		hprev = best;
		eprev = e;
	}

	// ---- start cell: first maximum in row-major order (u outer, v inner), borders (0) first
	float bv = 0.0f;
	int bu = 0;
	if (col && !global) {
		for (int uu = 1; uu <= len_s; uu++) {
			if (!local && !(uu == len_s || v == len_t)) continue;
			const float hv = H[uu * W + v];
			if (hv > bv) { bv = hv; bu = uu; }
		}
	}
	wave_lds_fence();
	int u = len_s, vq = len_t;
	float raw;
	if (global) {
		raw = H[len_s * W + len_t];
	} else {
		raw = 0.0f; u = 0; vq = 0;
		for (int j = 0; j < len_t; j++) {
			const float vj = __shfl(bv, j, 64);
			const int uj = __shfl(bu, j, 64);
			if (vj > raw || (vj == raw && vj > 0.0f && uj < u)) { raw = vj; u = uj; vq = j + 1; }
		}
	}
	if (lane != 0) return;
	int v2 = vq;
	int16_t *mp = p.mapping + (int64_t)w * 16;
	float *es = p.edge_sim + (int64_t)w * 16;
	for (int j = 0; j < 16; j++) { mp[j] = -1; es[j] = 0.0f; }
	int state = 0;
	while (u > 0 && v2 > 0) {
		const int idx = u * W + v2;
		const uint8_t fl = flags[idx];
		if (gap == 1 && state == 1) { if (!(fl & 4)) state = 0; u--; continue; }
		if (gap == 1 && state == 2) { if (!(fl & 8)) state = 0; v2--; continue; }
		const uint8_t d = fl & 3;
		if (d == 0) break;
		if (d == 1) { mp[v2 - 1] = (int16_t)(u - 1); es[v2 - 1] = Su[(u - 1) * 16 + v2 - 1]; u--; v2--; }
		else if (gap == 1) state = (d == 2) ? 1 : 2;
		else if (d == 2) u -= dk[idx];
		else v2 -= dk[idx];
	}
	p.raw_out[w] = raw;
}

// ---------------------------------------------------------------------------
// Queries of 17 .. 64 tokens: one wave per slice, lane = query column (the fill of vk_flow_kernel
// widened to the whole wave).  SCORE mode walks all slices and writes Score::value / raw like
// vk_score_kernel; FLOW mode retraces the k winners.  The similarity rows are produced 16 tokens at
// a time (one MFMA tile per 16 query rows) into a small LDS strip and consumed by the row-serial DP
// at once, so LDS holds only the column history (general gaps) and, in FLOW mode, the traceback.
// Candidate order, strict-greater replacement and start-cell rule: as vk_flow_kernel / the oracle.
// Roughly 10 us of issue time per (32-token slice, 32-token query): a fallback that keeps long
// queries on the device, not a roofline kernel.
// ---------------------------------------------------------------------------

static inline size_t vk_wide_lds_bytes(int max_len, int nq, int gap_mode, bool tagged, bool flow, bool gstate = false, int ring = 0) {
	const size_t LQ = (size_t)nq * 16, W = LQ + 1, rows = (size_t)max_len + 1;
	size_t fl = 16 * LQ * (tagged ? 2 : 1);            // Sx (+ SWx)
	fl += (gstate ? 0 : (rows + 3) / 4 * 4) + LQ + 4;  // wsl, wtl
	fl += 64 + 64;                                     // twl, tposl
	if (gap_mode == 2 && !gstate) fl += rows * W;      // H
	if (gap_mode == 2 && gstate && ring > 0) fl += (size_t)ring * W + (size_t)ring + 4;   // the last `ring` rows of H, w_s[0 .. ring)
	size_t b = fl * 4;
	if (flow) b += 64 * 2 + (gstate ? 0 : rows * W * 2 + rows * W);   // mapl, dk, flags
	b = (b + 15) / 16 * 16;
	if (flow) b += VK_CANON_LDS;                       // staging of sim_canon16, behind everything else
	return b;
}

// The state of a slice that grows with its length -- the column history of general gaps (H), the traceback's step lengths and
// flags -- in global memory (GS form, one region per workgroup): slices of any length the mapping's int16 can name (whole documents
// as slices), and queries of more than 16 tokens over slices whose state does not fit the LDS.  Layout of a region: H [rows x W]
// floats (general gaps), dk [rows x W] int16 and flags [rows x W] bytes (FLOW), each 16-byte aligned.
static inline size_t vk_wide_scratch_bytes_impl(int max_len, int nq, int gap_mode, bool flow, int ring = 0) {
	const size_t W = (size_t)nq * 16 + 1, rows = (size_t)max_len + 1;
	size_t b = 0;
	if (gap_mode == 2 && ring == 0) b += (rows * W * 4 + 15) / 16 * 16;   // (a saturated gap table: the history is a ring in LDS)
	if (flow) b += (rows * W * 2 + 15) / 16 * 16 + (rows * W + 15) / 16 * 16;
	return b < 16 ? 16 : b;
}

__device__ __forceinline__ float wave_min64(float m) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
	return m;
}

// the fence between a wave's writes of its slice state and lane 0's walk over it: LDS form, or global memory (GS)
template <bool GS>
__device__ __forceinline__ void wide_state_fence() {
	wave_lds_fence();
	if constexpr (GS) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
}

// GAPT: the gap mode at compile time (0 linear, 1 affine, 2 general, 4 relaxed WMD) -- the generic body took 106 scalar registers
// and spilled more into a vector register's lanes, read back inside the row loop
// GSM: 0 the state of a slice in LDS, 1 in global memory, 2 (general gaps with a saturated table, ws_tail) traceback in global
// memory and the column history as a ring of the last p.h_ring rows in LDS -- candidates further back than ws_tail rows are one
// running maximum, so nothing older is read, and the scan's loads come back in an LDS round trip instead of one to the L2
template <bool FLOW, int GSM, int GAPT>
__global__ __launch_bounds__(64) void vk_wide_kernel(VkWideParams p) {
	constexpr bool GS = GSM != 0, HR = GSM == 2;
	extern __shared__ float4 vk_smem4[];
	const int lane = threadIdx.x;
	const int LQ = p.nq * 16, W = LQ + 1, rows = p.max_len + 1;
	float *Sx = reinterpret_cast<float *>(vk_smem4);       // [16][LQ] similarities of the current 16 tokens
	float *SWx = p.pos_s ? Sx + 16 * LQ : Sx;              // tag-weighted copy the DP runs on
	float *wsl_lds = SWx + 16 * LQ;
	float *wtl = GS ? wsl_lds : wsl_lds + (rows + 3) / 4 * 4;
	float *twl = wtl + LQ + 4;
	int *tposl = reinterpret_cast<int *>(twl + 64);
	float *H_lds = reinterpret_cast<float *>(tposl + 64);  // general gaps: H[u][v], row stride W
	const int ring = HR ? p.h_ring : 0, rmask = ring - 1;
	float *wsk_lds = H_lds + ring * W;                     // HR: w_s[0 .. ring)
	float *after = (GAPT == 2 && !GS) ? H_lds + rows * W : HR ? wsk_lds + ring + 4 : H_lds;
	int16_t *mapl = reinterpret_cast<int16_t *>(after);    // FLOW: mapping of the winner
	int16_t *dk_lds = mapl + 64;
	uint8_t *flags_lds = reinterpret_cast<uint8_t *>(dk_lds + rows * W);
	uint8_t *lds_end = GS ? reinterpret_cast<uint8_t *>(dk_lds) : flags_lds + rows * W;
	uint8_t *canon = FLOW ? reinterpret_cast<uint8_t *>(vk_smem4) + (((size_t)(lds_end - reinterpret_cast<uint8_t *>(vk_smem4)) + 15) / 16 * 16) : nullptr;
	// GS: this workgroup's region of the scratch (vk_wide_scratch_bytes_impl); the gap table of the slices is read where it lies
	uint8_t *region = GS ? p.scratch + (int64_t)blockIdx.x * p.scratch_stride : nullptr;
	const size_t h_bytes = (GAPT == 2 && !HR) ? ((size_t)rows * W * 4 + 15) / 16 * 16 : 0;
	const size_t dk_bytes = ((size_t)rows * W * 2 + 15) / 16 * 16;
	float *H;
	int16_t *dk;
	uint8_t *flags;
	const float *wsl;
	if constexpr (GS) {
		H = reinterpret_cast<float *>(region);
		dk = reinterpret_cast<int16_t *>(region + h_bytes);
		flags = region + h_bytes + dk_bytes;
		wsl = p.ws;
		if constexpr (HR) for (int i = lane; i < ring && i <= p.max_len; i += 64) wsk_lds[i] = p.ws[i];
	} else {
		H = H_lds; dk = dk_lds; flags = flags_lds; wsl = wsl_lds;
		for (int i = lane; i <= p.max_len; i += 64) wsl_lds[i] = p.ws[i];
	}
	if (lane <= LQ) wtl[lane] = p.wt[lane];
	if (lane == 0) wtl[LQ] = p.wt[LQ <= 64 ? LQ : 64];
	twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane];
	wave_lds_fence();

	const int len_t = p.len_t;
	const int v = lane + 1;
	const bool col = v <= len_t;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	constexpr int gap = GAPT;
	const float gs = p.gs, gt = p.gt, open_s = p.open_s, open_t = p.open_t, a_s = p.a_s, a_t = p.a_t;
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	// general gaps, one block of 16 query columns: this lane's in-row costs w_t[v - pp], pp = 0 .. 15
	float wreg[16];
	if (gap == 2) {
#pragma unroll
		for (int pp = 0; pp < 16; pp++) {
			const int k = v - pp;
			wreg[pp] = wtl[k < 0 ? 0 : k > 16 ? 16 : k];
		}
	}
	// ... and, for the scoring pass, the closure of w_t: wcl[s] = w*(s) (uniform), wcb = w*(v) (the border column is v columns away)
	float wcl[16], wcb = 0.0f;
	if (gap == 2 && !FLOW) {
#pragma unroll
		for (int sh = 0; sh < 16; sh++) wcl[sh] = p.wt[80 + sh];
		wcb = p.wt[80 + (v <= 64 ? v : 64)];
	}
	auto Hrow = [&](int r) -> float * {
		// (24-bit multiplies: rows < 2^15, W <= 65 -- v_mul_u32_u24 issues at full rate, v_mul_lo_u32 at a quarter, eight per batch of the scan)
		if constexpr (HR) return H_lds + __umul24((unsigned)(r & rmask), (unsigned)W);
		else return H + __umul24((unsigned)r, (unsigned)W);
	};
	const float *wsk;   // the gap table as the candidate scan reads it
	if constexpr (HR) wsk = wsk_lds;
	else wsk = wsl;

	// SCORE: every row of the slice table, or -- p.order -- the non-empty ones, longest first (one wave per slice: a wave's share of
	// the work is then what the longest-processing-time rule deals it; the scores of the empty rows are preset by the host)
	const int64_t n_items = FLOW ? (int64_t)gridDim.x : p.order ? (int64_t)p.n_order : (int64_t)p.n_sent;
	for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
		int64_t g = (!FLOW && p.order) ? (int64_t)p.order[item] : item;
		if (FLOW) {
			const uint64_t key = p.keys[item];
			if (key == 0) return;   // fewer than k admitted
			g = (int64_t)(uint32_t)(key & 0xffffffffu);
		}
		const int t_a = p.sent_start[g], t_b = p.sent_end[g];
		const int len_s = t_b - t_a;
		if (len_s < 1) {
			if (!FLOW && lane == 0) { p.scores[g] = VK_NEG_INF; if (p.raw) p.raw[g] = VK_NEG_INF; }
			continue;
		}
		// tag-weighted vocabulary transport over the static layout: the query columns whose (id, tag) key occurs in this slice
		// (cells upstream writes twice take the tag weight of the slice token's key, static_vocab_fixup in vk_common.hip.h)
		unsigned long long vmask = 0;
		if (is_static && p.qid_bits && p.pos_s && gap == 4) {
			for (int tok = t_a; tok < t_b; tok++) {
				const int id = p.tok_id[tok];
				if (!((p.qid_bits[id >> 5] >> (id & 31)) & 1u)) continue;
				const int key = id * 256 + (p.tag_s[tok] & 255);
				for (int j = 0; j < len_t; j++) vmask |= p.qkey[j] == key ? 1ull << j : 0ull;
			}
			if (!(vmask & (vmask - 1))) vmask = 0;
		}
		// similarities of tokens base .. base + 15 (contextual: one tile, 16-aligned; static: gather)
		auto fill = [&](int base) {
			if constexpr (FLOW) {
				// winners: the canonical arithmetic (sim_canon16), as vk_flow_kernel -- 16 rows x 16 columns per query tile, lane l ->
				// row l & 15, columns 4 (l >> 4) .. + 3 (static layout: rows gathered from the vocabulary; past the slice's end its
				// first token again, never read).  (FLOW runs alignments only: no vocabulary fixup here.)
				const int trow = base + (lane & 15);
				const int tok = is_static ? (trow < t_b ? trow : t_a) : trow;
				const int id = is_static ? p.tok_id[tok] : 0;
				const uint8_t *xrow = is_static ? canon_row_ptr_static(p.tiles, p.tile_bytes, id) : canon_row_ptr(p.tiles + (int64_t)(base >> 4) * p.tile_bytes, lane);
				const int ps = p.pos_s ? p.pos_s[tok] : 0;
				for (int qt = 0; qt < p.nq; qt++) {
					float val[4];
					sim_canon16(xrow, p.qtile + (int64_t)qt * p.tile_bytes, p.nk32, p.tail, p.d, p.prec, canon, lane, val);
					const int c0 = qt * 16 + (lane >> 4) * 4;
#pragma unroll
					for (int r = 0; r < 4; r++) {
						val[r] = (is_static && p.q_ids && p.q_ids[c0 + r] == id) ? 1.0f : clip01(val[r]);
						Sx[(lane & 15) * LQ + c0 + r] = val[r];
						if (p.pos_s) SWx[(lane & 15) * LQ + c0 + r] = tag_weighted(val[r], twl[c0 + r], ps, tposl[c0 + r], p.tw_keep, p.tw_threshold);
					}
				}
				return;
			}
			if (is_static && !vmask) {
				// 16 tokens at once: lane -> token lane >> 2, four query columns 4 (lane & 3) .. + 3 of each 16-column table (one id
				// load and one 16-byte row load per lane and table, all in flight together; token by token the two dependent loads of
				// each cost a round trip to the L2, sixteen times per tile)
				const int r = lane >> 2, c4 = (lane & 3) * 4;
				const int tok = base + r;
				if (tok < t_b) {
					const int id = p.tok_id[tok];
					const int ps = p.pos_s ? p.pos_s[tok] : 0;
					for (int qt = 0; qt < p.nq; qt++) {
						const float4 sv = *reinterpret_cast<const float4 *>(p.table + (int64_t)qt * p.table_stride + (int64_t)id * 16 + c4);
						const int c0 = qt * 16 + c4;
						*reinterpret_cast<float4 *>(Sx + r * LQ + c0) = sv;
						if (p.pos_s) {
							float4 wv;
							wv.x = tag_weighted(sv.x, twl[c0 + 0], ps, tposl[c0 + 0], p.tw_keep, p.tw_threshold);
							wv.y = tag_weighted(sv.y, twl[c0 + 1], ps, tposl[c0 + 1], p.tw_keep, p.tw_threshold);
							wv.z = tag_weighted(sv.z, twl[c0 + 2], ps, tposl[c0 + 2], p.tw_keep, p.tw_threshold);
							wv.w = tag_weighted(sv.w, twl[c0 + 3], ps, tposl[c0 + 3], p.tw_keep, p.tw_threshold);
							*reinterpret_cast<float4 *>(SWx + r * LQ + c0) = wv;
						}
					}
				}
			} else if (is_static) {
				for (int r = 0; r < 16; r++) {
					const int tok = base + r;
					if (tok < t_b && lane < LQ) {
						const int id = p.tok_id[tok];
						const float sv = p.table[(int64_t)(lane >> 4) * p.table_stride + (int64_t)id * 16 + (lane & 15)];
						Sx[r * LQ + lane] = sv;
						if (p.pos_s) {
							float w = twl[lane];
							if (vmask && ((p.qid_bits[id >> 5] >> (id & 31)) & 1u)) {
								const int key = id * 256 + (p.tag_s[tok] & 255);
								int ft = -1;
								for (int j = len_t - 1; j >= 0; j--) ft = p.qkey[j] == key ? j : ft;
								if (ft >= 0 && lane < len_t && ((vmask >> lane) & 1ull) && p.qkey[lane] > key) w = twl[ft];
							}
							SWx[r * LQ + lane] = tag_weighted(sv, w, p.pos_s[tok], tposl[lane], p.tw_keep, p.tw_threshold);
						}
					}
				}
			} else {
				const uint8_t *tp = p.tiles + (int64_t)(base >> 4) * p.tile_bytes;
				for (int qt = 0; qt < p.nq; qt++) {
					f32x4 acc = sim_tile_generic(p.qtile + (int64_t)qt * p.tile_bytes, tp, p.nk32, p.tail, lane, p.prec);
					const int c0 = qt * 16 + (lane >> 4) * 4;
					*reinterpret_cast<f32x4 *>(Sx + (lane & 15) * LQ + c0) = acc;
					if (p.pos_s) {
						const int ps = p.pos_s[base + (lane & 15)];
#pragma unroll
						for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[c0 + r], ps, tposl[c0 + r], p.tw_keep, p.tw_threshold);
						*reinterpret_cast<f32x4 *>(SWx + (lane & 15) * LQ + c0) = acc;
					}
				}
			}
		};
		const int base0 = is_static ? t_a : (t_a >> 4) * 16;
		// Scoring passes whose rows are cheap (linear / affine gaps on the fast rows, relaxed WMD) over bf16 contextual rows of up to
		// 12 K-steps: the token tile of the NEXT 16 rows is requested before the rows of the current tile are walked -- three dependent
		// batches of loads per tile were 60 % of such a pass once its rows took 0.1 us each -- and every query tile reuses the one copy
		// in registers.  (Tried under slow rows first, and dropped there: DESIGN 9.)
		constexpr int NKP = 12;
		constexpr bool PIPE = GSM == 1 && !FLOW && (GAPT == 0 || GAPT == 1 || GAPT == 4);
		const int nfull = p.tail ? p.nk32 - 1 : p.nk32;
		const bool pipe = PIPE && !is_static && p.prec == 0 && p.nk32 <= NKP;
		bf16x8 xn[NKP], xh = {0, 0, 0, 0, 0, 0, 0, 0};
		auto tile_load = [&](int base) {
			if constexpr (PIPE) {
				const uint8_t *tp = p.tiles + (int64_t)(base >> 4) * p.tile_bytes;
#pragma unroll
				for (int i = 0; i < NKP; i++)   // (K-steps the row does not have re-read its first one: unconditional loads, no copies of xn kept alive)
					xn[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (i < nfull ? i : 0) * 1024 + lane * 16));
				xh = load_half_block(tp + (p.tail ? nfull : 0) * 1024, lane, true);
			}
		};
		auto tile_mma = [&](int base) {   // the MFMA sequence of sim_tile_generic over the tile in xn / xh
			if constexpr (PIPE)
			for (int qt = 0; qt < p.nq; qt++) {
				const uint8_t *qp = p.qtile + (int64_t)qt * p.tile_bytes;
				f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
				for (int i = 0; i < NKP; i++)
					if (i < nfull) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(qp + i * 1024 + lane * 16), xn[i], acc, 0, 0, 0);
				if (p.tail) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(load_half_block(qp + nfull * 1024, lane, false), xh, acc, 0, 0, 0);
				acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
				const int c0 = qt * 16 + (lane >> 4) * 4;
				*reinterpret_cast<f32x4 *>(Sx + (lane & 15) * LQ + c0) = acc;
				if (p.pos_s) {
					const int ps = p.pos_s[base + (lane & 15)];
#pragma unroll
					for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[c0 + r], ps, tposl[c0 + r], p.tw_keep, p.tw_threshold);
					*reinterpret_cast<f32x4 *>(SWx + (lane & 15) * LQ + c0) = acc;
				}
			}
		};
		auto fill_next = [&](int base) {   // the rows of tile `base` into the strip; on the pipelined form: and the next tile requested
			if (pipe) {
				tile_mma(base);
				if (base + 16 < t_b) tile_load(base + 16);
			} else fill(base);
		};
		if (pipe) tile_load(base0);
		// FLOW over long winners: the rows the recurrence runs on were restated beforehand (vk_canon_rows_kernel: the same canonical
		// arithmetic, sim[id(t_j)][j] = 1 and tag weights applied), [dp_rows_len][LQ] per winner, row 0 = the slice's first token
		auto fill_dp = [&](int base) {
			const float *rows = p.dp_rows + (int64_t)item * p.dp_rows_len * LQ;
			for (int i = lane; i < 4 * LQ; i += 64) {   // 16 rows of LQ floats, a float4 per step
				const int r = i / (LQ >> 2), c4 = i % (LQ >> 2);
				const int rel = base + r - t_a;
				float4 val = {0.0f, 0.0f, 0.0f, 0.0f};
				if (rel >= 0 && rel < len_s) val = *reinterpret_cast<const float4 *>(rows + (int64_t)rel * LQ + c4 * 4);
				*reinterpret_cast<float4 *>(SWx + r * LQ + c4 * 4) = val;
			}
		};

		float raw;
		int u_start = 0, v_start = 0;
		if (gap == 4) {
			// ---- relaxed word mover's distance (rwmd_rows of vk_score_kernel over <= 64 columns)
			const bool nbow = p.rwmd_normalize_bow != 0;
			const float w_t = nbow ? 1.0f / (float)len_t : 1.0f, w_s = nbow ? 1.0f / (float)len_s : 1.0f;
			float colmin = 3.402823466e+38F, acc1 = 0.0f;
			for (int base = base0; base < t_b; base += 16) {
				fill_next(base);
				wave_lds_fence();
				const int r0 = t_a > base ? t_a - base : 0, r1 = t_b - base < 16 ? t_b - base : 16;
				for (int r = r0; r < r1; r++) {
					const float dist = fmaxf(1.0f - SWx[r * LQ + (col ? v - 1 : 0)], 0.0f);   // the tag-weighted similarity (SWx == Sx without tag weights); round 2 read Sx here
					colmin = fminf(colmin, dist);
					// the row's minimum over the query columns (the s -> t direction: read by the symmetric form only); one block of 16
					// columns: four DPP steps within the lanes' row and one v_readlane instead of six cross-lane round trips
					if (p.rwmd_symmetric) {
						float m = col ? dist : 3.402823466e+38F;
						if (LQ == 16) {
							m = fminf(m, dpp_f<DPP_ROW_SHR1>(3.402823466e+38F, m));
							m = fminf(m, dpp_f<DPP_ROW_SHR2>(3.402823466e+38F, m));
							m = fminf(m, dpp_f<DPP_ROW_SHR4>(3.402823466e+38F, m));
							m = fminf(m, dpp_f<DPP_ROW_SHR8>(3.402823466e+38F, m));
							m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 15));
						} else m = wave_min64(m);
						acc1 += w_s * m;
					}
				}
				wave_lds_fence();
			}
			const float x = col ? w_t * colmin : 0.0f;
			float acc0 = 0.0f;
			for (int j = 0; j < len_t; j++) {
				const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
				acc0 = j == 0 ? xj : acc0 + xj;
			}
			if (!nbow) { acc0 = acc0 / (float)len_t; acc1 = acc1 / (float)len_s; }
			float cost = 0.0f;
			if (p.rwmd_symmetric) { if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
			else cost = acc0;
			const float max_cost = nbow ? 1.0f : (float)len_t;
			raw = (max_cost - cost) / max_cost;
		} else {
			// ---- alignment: fill, lane = column
			float hprev = 0.0f, eprev = VK_NEG_INF;
			if (global && col) hprev = gap == 0 ? -(gt * (float)v) : gap == 1 ? -(a_t + gt * (float)v) : -wtl[v];
			if (gap == 2 && col) Hrow(0)[v] = hprev;
			float bv = 0.0f;
			int bu = 0, u = 0;
			const bool fast16 = !FLOW && LQ == 16 && (gap == 0 || (gap == 1 && a_t >= 0.0f));
			const DecaySteps dt16 = decay_steps(gt, lane & 15);
			const int ktail = (gap == 2 && p.ws_tail > 0) ? p.ws_tail : 0x7fffffff;
			const float wtail = ktail <= p.max_len ? wsk[ktail] : 0.0f;
			float tmax = VK_NEG_INF;
			int trow = 0;
			for (int base = base0; base < t_b; base += 16) {
				if (FLOW && p.dp_rows) fill_dp(base);
				else fill_next(base);
				wave_lds_fence();
				const int r0 = t_a > base ? t_a - base : 0, r1 = t_b - base < 16 ? t_b - base : 16;
				if (fast16) {
					// the fast rows of a tile: its sixteen similarities read from the strip up front (the rolled loop waits out an LDS round
					// trip at the head of every row's dependent chain), then the rows one after the other
					float svv[16];
#pragma unroll
					for (int r = 0; r < 16; r++) svv[r] = SWx[r * LQ + (col ? v - 1 : 0)];
					const float floor0 = local ? 0.0f : VK_NEG_INF;
#pragma unroll
					for (int r = 0; r < 16; r++) {
						if (r >= r0 && r < r1) {
							u++;
							float bprev = 0.0f, bcur = 0.0f;
							if (global) {
								bprev = u == 1 ? 0.0f : (gap == 0 ? -(gs * (float)(u - 1)) : -(a_s + gs * (float)(u - 1)));
								bcur = gap == 0 ? -(gs * (float)u) : -(a_s + gs * (float)u);
							}
							const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hprev);
							float best, e = VK_NEG_INF;
							if (gap == 0) {
								const float c = fmaxf(fmaxf(diag + svv[r], floor0), hprev - gs);
								best = decay_scan<16>(c, dt16);
								if (!local) best = fmaxf(best, bcur - gt * (float)v);
							} else {
								e = fmaxf(hprev - open_s, eprev - gs);
								const float c = fmaxf(fmaxf(diag + svv[r], floor0), e);
								best = fmaxf(c, decay_scan<16>(dpp_f<DPP_ROW_SHR1>(bcur, c) - open_t, dt16));
							}
							if (col && !global && (local || u == len_s || v == len_t) && best > bv) { bv = best; bu = u; }
							hprev = best;
							eprev = e;
						}
					}
				} else
				for (int r = r0; r < r1; r++) {
					u++;
					float bprev = 0.0f, bcur = 0.0f;
					if (global) {
						bprev = u == 1 ? 0.0f : (gap == 0 ? -(gs * (float)(u - 1)) : gap == 1 ? -(a_s + gs * (float)(u - 1)) : -wsl[u - 1]);
						bcur = gap == 0 ? -(gs * (float)u) : gap == 1 ? -(a_s + gs * (float)u) : -wsl[u];
					}
					const float sv = SWx[r * LQ + (col ? v - 1 : 0)];
					float best, e = VK_NEG_INF;
					uint8_t d = 0, ee = 0, fe = 0;
					int16_t kk = 0;
					if (fast16) {
						// The scoring pass of a query of at most 16 tokens under linear / affine gaps: the row as the fused kernel takes it
						// (dp_linear / dp_affine, vk_common.hip.h) -- the in-row gaps as a decayed prefix maximum, four DPP steps within the
						// lanes' row of 16, instead of a chain of len_t dependent v_readlane steps; values agree to the last bits (the
						// winners are restated by the serial form, FLOW)
						const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hprev);
						const float floor0 = local ? 0.0f : VK_NEG_INF;
						if (gap == 0) {
							float c = fmaxf(fmaxf(diag + sv, floor0), hprev - gs);
							best = decay_scan<16>(c, dt16);
							if (!local) best = fmaxf(best, bcur - gt * (float)v);
						} else {
							e = fmaxf(hprev - open_s, eprev - gs);
							const float c = fmaxf(fmaxf(diag + sv, floor0), e);
							const float fsc = decay_scan<16>(dpp_f<DPP_ROW_SHR1>(bcur, c) - open_t, dt16);
							best = fmaxf(c, fsc);
						}
					} else {
					const float up = __shfl_up(hprev, 1, 64);
					const float diag = lane == 0 ? bprev : up;
					float c = diag + sv;
					if (local) { best = 0.0f; d = 0; if (c > best) { best = c; d = 1; } }
					else { best = c; d = 1; }
					if (gap == 0) {
						c = hprev - gs;
						if (c > best) { best = c; d = 2; kk = 1; }
					} else if (gap == 1) {
						e = hprev - open_s;
						c = eprev - gs;
						if (c > e) { e = c; ee = 1; }
						if (e > best) { best = e; d = 2; }
					} else {
						// gaps of ktail tokens and more cost the same (a saturated table, e.g. 1 - 2^(-k/5) = 1.0f from k = 126 on): their
						// candidates H[r][v] - w_tail, r <= u - ktail, are a running maximum over the rows as they leave the window -- kept
						// with the largest row among equal values, i.e. the smallest k, which is the one the k-ascending scan with its strict
						// comparison keeps -- and the scan stops at ktail - 1: O(len_s ktail) per column instead of O(len_s^2)
						const int kmax = u < ktail ? u : ktail - 1;
						// The scan over k is dealt out over the lanes a short query leaves idle: with 16 (32) columns four (two) lanes share a
						// column, lane group kg takes k = 1 + kg, 1 + kg + KG, ...; each keeps its best candidate and the smallest k that
						// reached it, the groups are merged by value, then by k -- what the k-ascending scan with its strict comparison
						// keeps -- and the result meets `best` once, strictly.  Eight loads are in flight per lane.
						const int KG = LQ == 16 ? 4 : LQ == 32 ? 2 : 1;
						const int kg = KG == 4 ? lane >> 4 : KG == 2 ? lane >> 5 : 0;
						const int vs = (lane & (LQ - 1)) + 1;
						const int cv = KG == 1 ? (col ? v : 1) : (vs <= len_t ? vs : 1);
						float cb = VK_NEG_INF;
						int ck = 0;
						int k = 1 + kg;
						for (; k + 7 * KG <= kmax; k += 8 * KG) {
							float hv[8], wv[8];
#pragma unroll
							for (int i = 0; i < 8; i++) { hv[i] = Hrow(u - k - i * KG)[cv]; wv[i] = wsk[k + i * KG]; }
#pragma unroll
							for (int i = 0; i < 8; i++) {
								c = hv[i] - wv[i];
								if constexpr (FLOW) { if (c > cb) { cb = c; ck = k + i * KG; } }
								else cb = fmaxf(cb, c);   // the scoring pass needs the value only
							}
						}
						for (; k <= kmax; k += KG) {
							c = Hrow(u - k)[cv] - wsk[k];
							if constexpr (FLOW) { if (c > cb) { cb = c; ck = k; } }
							else cb = fmaxf(cb, c);
						}
						if (KG >= 2) {
							for (int off = 32; off >= LQ; off >>= 1) {
								const float oc = __shfl_xor(cb, off, 64);
								if constexpr (FLOW) {
									const int ok = __shfl_xor(ck, off, 64);
									if (oc > cb || (oc == cb && ok < ck)) { cb = oc; ck = ok; }
								} else cb = fmaxf(cb, oc);
							}
						}
						if (cb > best) { best = cb; d = 2; kk = (int16_t)ck; }
						if (u >= ktail) {
							c = Hrow(u - ktail)[col ? v : 1] - wtail;
							if (c >= tmax) { tmax = c; trow = u - ktail; }
							if (tmax > best) { best = tmax; d = 2; kk = (int16_t)(u - trow); }
						}
					}
					// in-row candidates: columns become final left to right
					float left_best = VK_NEG_INF, f = VK_NEG_INF, fin = best, ffin = VK_NEG_INF;
					int16_t left_k = 0;
					if (!FLOW && gap == 2 && LQ == 16) {
						// The scoring pass, one block of 16 columns: the in-row candidates taken from the row's values BEFORE in-row gaps, with
						// w_t replaced by its subadditive closure (the host's wt[80 ..]; dp_general_reg of the fused kernel: the sequential
						// recurrence chains gaps, which is what the closure prices) -- fifteen independent DPP shifts instead of a chain of
						// len_t dependent steps; values agree to the last bits, the winners are restated by the serial form (FLOW)
						const float arow = best;
						float x = fmaxf(arow, bcur - wcb);
#define VK_CLOSURE_STEP(SH) x = fmaxf(x, dpp_f<0x110 + SH>(VK_NEG_INF, arow) - wcl[SH]);   /* row_shr:SH, lanes without a source drop out */
						VK_CLOSURE_STEP(1) VK_CLOSURE_STEP(2) VK_CLOSURE_STEP(3) VK_CLOSURE_STEP(4) VK_CLOSURE_STEP(5)
						VK_CLOSURE_STEP(6) VK_CLOSURE_STEP(7) VK_CLOSURE_STEP(8) VK_CLOSURE_STEP(9) VK_CLOSURE_STEP(10)
						VK_CLOSURE_STEP(11) VK_CLOSURE_STEP(12) VK_CLOSURE_STEP(13) VK_CLOSURE_STEP(14) VK_CLOSURE_STEP(15)
#undef VK_CLOSURE_STEP
						best = x;
					} else if (gap == 2 && LQ == 16) {
						// one block of 16 columns: the same chain unrolled, the costs w_t[v - pp] in registers (wreg: read once per kernel; the
						// rolled form reads LDS inside every step of the dependent chain) and the lane of every v_readlane a constant
#pragma unroll
						for (int pp = 0; pp < 16; pp++) {
							if (pp < len_t) {
								const float sp = pp == 0 ? bcur : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), pp > 0 ? pp - 1 : 0));
								if (col && v > pp) {
									const float cc = sp - wreg[pp];
									if (cc >= left_best) { left_best = cc; left_k = (int16_t)(v - pp); }
								}
								if (v == pp + 1) {
									if (left_best > best) { best = left_best; d = 3; kk = left_k; }
									fin = best;
								}
							}
						}
					} else
					for (int pp = 0; pp < len_t; pp++) {
						const float sp = pp == 0 ? bcur : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), pp - 1));
						const float fp = pp == 0 ? VK_NEG_INF : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ffin), pp - 1));
						if (gap == 2) {
							if (col && v > pp) {
								const float cc = sp - wtl[v - pp];
								if (cc >= left_best) { left_best = cc; left_k = (int16_t)(v - pp); }
							}
						} else if (v == pp + 1) {
							if (gap == 0) { left_best = sp - gt; left_k = 1; }
							else {
								f = sp - open_t;
								const float c2 = fp - gt;
								if (c2 > f) { f = c2; fe = 1; }
							}
						}
						if (v == pp + 1) {
							if (gap == 1) { if (f > best) { best = f; d = 3; } ffin = f; }
							else if (left_best > best) { best = left_best; d = 3; kk = left_k; }
							fin = best;
						}
					}
					}
					if (col) {
						if (gap == 2) Hrow(u)[v] = best;
						if (FLOW) {
							dk[u * W + v] = kk;
							flags[u * W + v] = (uint8_t)(d | (ee << 2) | (fe << 3));
						}
						// start cell: first maximum in row-major order; per column the first row wins (strict >)
						if (!global && (local || u == len_s || v == len_t) && best > bv) { bv = best; bu = u; }
					}
					hprev = best;
					eprev = e;
				}
				wave_lds_fence();
				if constexpr (GS && !HR && GAPT == 2) {
					// the history row just stored lives in global memory and is read by OTHER lanes in the next row's candidate scan:
					// same-wave vector memory is in order, this pins the compiler to it (wavefront scope: no instructions)
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
					__builtin_amdgcn_wave_barrier();
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				}
			}
			if (global) {
				raw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hprev), len_t - 1));
				u_start = len_s; v_start = len_t;
			} else {
				raw = 0.0f;
				for (int j = 0; j < len_t; j++) {
					const float vj = __shfl(bv, j, 64);
					const int uj = __shfl(bu, j, 64);
					if (vj > raw || (vj == raw && vj > 0.0f && uj < u_start)) { raw = vj; u_start = uj; v_start = j + 1; }
				}
			}
		}

		if (!FLOW) {
			if (lane == 0) {
				const float boost = p.boost ? p.boost[g] : 1.0f;
				p.scores[g] = (raw / p.ref_total) * boost;
				if (p.raw) p.raw[g] = raw;
			}
			continue;
		}
		// ---- FLOW: traceback by lane 0, then the edge similarities from a second sweep over the tiles
		mapl[lane] = -1;
		wide_state_fence<GS>();
		if (lane == 0 && gap != 4) {
			int u = u_start, v2 = v_start, state = 0;
			while (u > 0 && v2 > 0) {
				const int idx = u * W + v2;
				const uint8_t fl = flags[idx];
				if (gap == 1 && state == 1) { if (!(fl & 4)) state = 0; u--; continue; }
				if (gap == 1 && state == 2) { if (!(fl & 8)) state = 0; v2--; continue; }
				const uint8_t d = fl & 3;
				if (d == 0) break;
				if (d == 1) { mapl[v2 - 1] = (int16_t)(u - 1); u--; v2--; }
				else if (gap == 1) state = (d == 2) ? 1 : 2;
				else if (d == 2) u -= dk[idx];
				else v2 -= dk[idx];
			}
		}
		wave_lds_fence();
		const int mine = mapl[lane];
		float es = 0.0f;
		// only the tiles that hold a mapped row (the mapping ascends with the query column: each such tile is restated once)
		int done = -1;
		for (int j = 0; j < len_t; j++) {
			const int mj = mapl[j];
			if (mj < 0) continue;
			const int base = is_static ? t_a + ((mj >> 4) << 4) : ((t_a + mj) >> 4) << 4;
			if (base == done) continue;
			done = base;
			fill(base);
			wave_lds_fence();
			const int row = t_a + mine - base;
			if (mine >= 0 && row >= 0 && row < 16) es = Sx[row * LQ + lane];
			wave_lds_fence();
		}
		p.mapping[item * 64 + lane] = (int16_t)mine;
		p.edge_sim[item * 64 + lane] = es;
		if (lane == 0) p.raw_out[item] = raw;
	}
}

// Residency of the global-state form: a region of scratch per workgroup, so the grid is what the scratch allows
static const int64_t kWideScratchCap = 4ll << 30;

extern "C" size_t vk_wide_scratch_bytes(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t flow, int32_t ring) {
	return vk_wide_scratch_bytes_impl(max_len, nq, gap_mode, flow != 0, ring);
}

// rows of the LDS ring that holds the column history of general gaps when the gap table is constant from ws_tail on: a power of
// two above ws_tail, 0 when there is no such tail or the ring would not leave room for several workgroups per CU
extern "C" int32_t vk_wide_ring_rows(int32_t nq, int32_t gap_mode, int32_t ws_tail) {
	if (gap_mode != 2 || ws_tail < 1) return 0;
	int ring = 16;
	while (ring < ws_tail + 1) ring <<= 1;
	return (size_t)ring * (16 * (size_t)nq + 2) * 4 <= 40 * 1024 ? ring : 0;
}

// workgroups of a global-state launch: SCORE walks the slices with a grid stride (every CU filled, within the scratch cap);
// FLOW takes one workgroup per winner
extern "C" int32_t vk_wide_gs_blocks(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t flow_k, int64_t n_sent, int32_t ring) {
	if (flow_k > 0) return flow_k;
	const int64_t per = (int64_t)vk_wide_scratch_bytes_impl(max_len, nq, gap_mode, false, ring);
	int64_t blocks = (gap_mode == 2 && ring == 0) ? kWideScratchCap / per : 2048;
	if (blocks > 2048) blocks = 2048;
	if (blocks > n_sent) blocks = n_sent;
	return (int32_t)(blocks < 1 ? 1 : blocks);
}

template <bool FLOW, int GSM>
static hipError_t launch_wide_gap(const VkWideParams &p, int blocks, size_t smem, hipStream_t stream, bool occupancy_grid) {
	auto go = [&](auto kernel) -> hipError_t {
		if (smem > 64 * 1024) {
			hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
			if (e != hipSuccess) return e;
		}
		if (occupancy_grid) {   // SCORE, state in LDS: the slices are walked with a grid stride by as many workgroups as are resident
			int occ = 0, dev = 0, cus = 256;
			hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 64, smem);
			if (e != hipSuccess) return e;
			if (occ < 1) occ = 1;
			if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			const int64_t want = p.order ? p.n_order : p.n_sent, cap = (int64_t)cus * occ;
			blocks = (int)(want < cap ? want : cap);
		}
		kernel<<<blocks, 64, smem, stream>>>(p);
		return hipGetLastError();
	};
	if constexpr (GSM == 2) return p.gap_mode == 2 ? go(vk_wide_kernel<FLOW, 2, 2>) : hipErrorInvalidValue;
	else switch (p.gap_mode) {
	case 0: return go(vk_wide_kernel<FLOW, GSM, 0>);
	case 1: return go(vk_wide_kernel<FLOW, GSM, 1>);
	case 2: return go(vk_wide_kernel<FLOW, GSM, 2>);
	case 4: return go(vk_wide_kernel<FLOW, GSM, 4>);
	default: return hipErrorInvalidValue;
	}
}

extern "C" hipError_t vk_launch_wide(const VkWideParams *p, int32_t flow_k, hipStream_t stream) {
	const bool flow = flow_k > 0;
	const bool gs = p->scratch != nullptr;
	const int ring = (gs && p->gap_mode == 2) ? p->h_ring : 0;
	if (ring != 0 && (ring != vk_wide_ring_rows(p->nq, p->gap_mode, p->ws_tail) || (ring & (ring - 1)))) return hipErrorInvalidValue;
	const size_t smem = vk_wide_lds_bytes(p->max_len, p->nq, p->gap_mode, p->pos_s != nullptr, flow, gs, ring);
	if (smem > 160 * 1024) return hipErrorInvalidValue;
	if (gs) {
		// the host sized p->scratch for vk_wide_gs_blocks regions of scratch_stride bytes
		const int blocks = vk_wide_gs_blocks(p->max_len, p->nq, p->gap_mode, flow_k, p->order ? p->n_order : p->n_sent, ring);
		if (p->scratch_stride < (int64_t)vk_wide_scratch_bytes_impl(p->max_len, p->nq, p->gap_mode, flow, ring)) return hipErrorInvalidValue;
		if (ring) return flow ? launch_wide_gap<true, 2>(*p, blocks, smem, stream, false) : launch_wide_gap<false, 2>(*p, blocks, smem, stream, false);
		return flow ? launch_wide_gap<true, 1>(*p, blocks, smem, stream, false) : launch_wide_gap<false, 1>(*p, blocks, smem, stream, false);
	}
	if (flow) return launch_wide_gap<true, 0>(*p, flow_k, smem, stream, false);
	return launch_wide_gap<false, 0>(*p, 0, smem, stream, true);
}

extern "C" size_t vk_wide_lds_demand(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t tagged, int32_t flow) {
	return vk_wide_lds_bytes(max_len, nq, gap_mode, tagged != 0, flow != 0);
}

extern "C" hipError_t vk_launch_flow(const VkFlowParams *p, int32_t k, hipStream_t stream) {
	const size_t smem = vk_flow_lds_bytes(p->max_len, p->pos_s != nullptr);
	if (smem > 64 * 1024) {
		hipError_t e = hipFuncSetAttribute(p->prec ? reinterpret_cast<const void *>(vk_flow_kernel<1>) : reinterpret_cast<const void *>(vk_flow_kernel<0>),
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	if (p->prec) vk_flow_kernel<1><<<k, 64, smem, stream>>>(*p);
	else vk_flow_kernel<0><<<k, 64, smem, stream>>>(*p);
	return hipGetLastError();
}
