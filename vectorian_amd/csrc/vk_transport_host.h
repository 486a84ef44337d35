// vk_transport_host.h -- the relaxed word mover's distance of ONE slice restated on the host from its similarity rows.
//
// The scoring kernels rank a million slices on MFMA cosines (within 2e-6 of the reference's, never bit-equal); the rows of the few
// winners come back from vk_rows_kernel in the canonical arithmetic (sim_canon16, DESIGN.md 7.1: the bits a scalar fp32 dot product
// summed as the reference sums it would give).  From those rows the score of a winner is computed here once more, operation by
// operation in the order of the reference's RelaxedSolver, so that the result set carries the reference's own floats:
//   vocabulary and bag of words  vectorian/core/cpp/alignment/bow.h:204-275 (one entry per position) and :281-333 (one entry per
//                                distinct token -- the static layout; entries in token order, a count per entry)
//   distances                    alignment/wmd.h:107-135 (1 - similarity, clamped at 0, between the FIRST positions of two entries)
//   relaxed costs                alignment/wmd.h:287-416 (t -> s, and s -> t when symmetric; 1:1 nearest entry, or 1:n by ascending
//                                distance until the mass is placed), cost_to_score :138-140
// The rows handed in are the slice's modified similarities (tag weights applied).  The cells that upstream's symmetric distance matrix
// holds twice are resolved here, from the rows' cells of the later write (the rows of vk_rows_kernel carry that value in the
// earlier cell too, static_vocab_fixup: the host states flows from them; this function never reads a rewritten cell).
#ifndef VK_TRANSPORT_HOST_H
#define VK_TRANSPORT_HOST_H

#include <algorithm>
#include <cstdint>
#include <vector>

namespace vk_host {

struct BowEntry {
	int32_t first;   // first position of the entry in its document
	float mass;      // occurrences (divided by the document's length when the bags are normalised)
	int32_t key;     // vocabulary key (position vocabularies: unused)
};

// entries of one document: key == nullptr -- every position is an entry, in position order; else one entry per distinct key, in
// ascending key order (upstream sorts the tokens of both documents by id to build the joint vocabulary)
inline void bag_of_words(const int32_t *key, int n, bool normalise, std::vector<BowEntry> &out) {
	out.clear();
	if (!key) {
		for (int i = 0; i < n; i++) out.push_back({i, 1.0f, 0});
	} else {
		std::vector<int32_t> by_key((size_t)n);
		for (int i = 0; i < n; i++) by_key[(size_t)i] = i;
		std::sort(by_key.begin(), by_key.end(), [key](int32_t a, int32_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
		for (int i = 0; i < n; i++) {
			const int32_t pos = by_key[(size_t)i];
			if (i > 0 && key[pos] == key[by_key[(size_t)i - 1]]) out.back().mass += 1.0f;
			else out.push_back({pos, 1.0f, key[pos]});
		}
	}
	if (normalise) {
		const float total = (float)n;
		for (auto &e : out) e.mass /= total;
	}
}

// S[i * ld + j]: similarity of slice token i and query token j.  key_s / key_t: vocabulary keys of the tokens (both null: every
// position is its own entry).  Returns the score (max_cost - cost) / max_cost.
inline float rwmd_from_rows(const float *S, int ld, int len_s, int len_t, const int32_t *key_s, const int32_t *key_t,
	bool injective, bool symmetric, bool normalise) {
	if (len_s <= 0 || len_t <= 0) return 0.0f;
	const bool vocab = key_s && key_t;
	std::vector<BowEntry> doc[2];   // 0: the slice, 1: the query
	bag_of_words(vocab ? key_s : nullptr, len_s, normalise, doc[0]);
	bag_of_words(vocab ? key_t : nullptr, len_t, normalise, doc[1]);
	const int len[2] = {len_s, len_t};
	// first position of a key in a document, -1: absent (entries are in key order)
	auto first_of = [&](int d, int32_t key) {
		auto it = std::lower_bound(doc[d].begin(), doc[d].end(), key, [](const BowEntry &e, int32_t k) { return e.key < k; });
		return it != doc[d].end() && it->key == key ? it->first : -1;
	};
	// Upstream fills a SYMMETRIC matrix over the joint vocabulary, dist(u, v) = dist(v, u) = d(first position of u in s, of v in t),
	// u over the slice's entries in key order, v over the query's (wmd.h:121-133): a cell whose two keys occur in BOTH documents is
	// written twice and the later write -- the larger slice-side key -- stands.  With a similarity that is not symmetric in the two
	// sides (tag weights belong to the query side) the two values differ.
	auto distance = [&](int from_doc, const BowEntry &a, const BowEntry &b) {   // a: entry of from_doc, b: entry of the other one
		const BowEntry &es = from_doc == 0 ? a : b, &et = from_doc == 0 ? b : a;   // the slice's entry, the query's entry
		int i = es.first, j = et.first;
		if (vocab && et.key > es.key) {
			const int i2 = first_of(0, et.key), j2 = first_of(1, es.key);
			if (i2 >= 0 && j2 >= 0) { i = i2; j = j2; }
		}
		const float d = 1.0f - S[(size_t)i * ld + j];
		return d > 0.0f ? d : 0.0f;
	};
	struct Candidate { float d; int32_t pos, entry; };
	std::vector<Candidate> cand;
	float cost = 0.0f;
	for (int pass = 0; pass < 2; pass++) {
		const int from = pass == 0 ? 1 : 0, to = 1 - from;   // first the query's mass moves to the slice
		float acc = 0.0f;
		for (const BowEntry &a : doc[from]) {
			if (injective) {
				float best = 3.402823466e+38F;
				bool found = false;
				for (const BowEntry &b : doc[to]) {
					const float d = distance(from, a, b);
					if (d < best) { best = d; found = true; }
				}
				acc += a.mass * (found ? best : 1.0f);
			} else {
				cand.clear();
				for (size_t e = 0; e < doc[to].size(); e++) cand.push_back({distance(from, a, doc[to][e]), doc[to][e].first, (int32_t)e});
				// upstream pops a heap keyed on the distance alone; among equal distances the order is the oracle's: by first position
				std::sort(cand.begin(), cand.end(), [](const Candidate &x, const Candidate &y) { return x.d != y.d ? x.d < y.d : x.pos < y.pos; });
				float remaining = a.mass;
				for (const Candidate &cd : cand) {
					const float room = doc[to][(size_t)cd.entry].mass;
					if (remaining <= room) {
						acc += remaining * cd.d;
						break;
					}
					remaining -= room;
					acc += room * cd.d;
				}
				// (wmd.h:373-375 as written: `remaining` keeps its value when the loop breaks, and is charged once more at distance 1)
				if (remaining > 0.0f) acc += remaining * 1.0f;
			}
		}
		if (!normalise) acc /= (float)len[from];
		if (!symmetric) { cost = acc; break; }
		if (acc > cost) cost = acc;
	}
	const float max_cost = normalise ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

} // namespace vk_host

#endif
