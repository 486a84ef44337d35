"""Option dicts of the reference -> vk_query arguments (Query::initialize, vectorian/core/cpp/query.cpp:32-154;
create_alignment_matcher, metric/alignment.h:780-920)."""

from vectorian_amd import core
from vectorian_amd.alignment import GapCost

_QUERY_OPTION_WHITELIST = {
	# Query::initialize (vectorian/core/cpp/query.cpp:45-55)
	"metric", "pos_filter", "tag_filter", "submatch_weight", "bidirectional",
	"max_matches", "min_score", "partition", "debug"}


def _split_gap(gap):
	"""gap_cost is one GapCost or {'s': .., 't': ..} (metric/alignment.h:365-370; vectorian/alignment.py:78-83)"""
	if isinstance(gap, dict):
		from vectorian_amd.alignment import ConstantGapCost
		return gap.get("s", ConstantGapCost(0)), gap.get("t", ConstantGapCost(0))
	return gap, gap



def backend_args(options):
	"""option dicts of the reference -> vk_query arguments (Query::initialize,
	vectorian/core/cpp/query.cpp:32-154; create_alignment_matcher, metric/alignment.h:780-920)"""
	for k in options:
		if k not in _QUERY_OPTION_WHITELIST:
			raise RuntimeError(f"illegal option {k}")   # query.cpp:60-63
	metric = options.get("metric")
	if not isinstance(metric, dict) or metric.get("metric") not in ("alignment-isolated", "alignment-tag-weighted"):
		raise RuntimeError(f"unknown sentence metric type {metric.get('metric') if isinstance(metric, dict) else metric}")  # instantiate.cpp:191-196
	alignment = metric["alignment"]
	args = dict(
		max_matches=int(options.get("max_matches", 100)),      # query.cpp:87-89
		min_score=float(options.get("min_score", 0.2)),        # query.cpp:91-93
		submatch_weight=float(options.get("submatch_weight", 0.0)),
		bidirectional=bool(options.get("bidirectional", False)))
	algorithm = alignment.get("algorithm")
	if algorithm == "pyalign":
		o = alignment.get("options", {})
		gs, gt = _split_gap(o.get("gap_cost"))
		for g in (gs, gt):
			if not isinstance(g, GapCost):
				raise TypeError(f"gap cost {g!r} is not a GapCost")
		args.update(algorithm=core.VK_ALG_ALIGN, locality=int(o.get("locality", core.Locality.LOCAL)), gap_s=gs, gap_t=gt)
		gaps = (gs, gt)
	elif algorithm == "word-movers-distance":
		args.update(algorithm=core.VK_ALG_RWMD, wmd_full=not alignment.get("relaxed", True),
			rwmd=(alignment["injective"], alignment["symmetric"], alignment["normalize_bow"]))
		gaps = (lambda k: 0.0, lambda k: 0.0)   # gap_cost_s/t of WordMoversDistance return 0 (metric/alignment.h:632-638)
	elif algorithm == "word-rotators-distance":
		args.update(algorithm=core.VK_ALG_WRD, wrd_normalize=alignment.get("normalize_magnitudes", True))
		gaps = (lambda k: 0.0, lambda k: 0.0)
	else:
		raise RuntimeError(f"unknown alignment algorithm {algorithm}")   # metric/alignment.h:914-919
	if metric["metric"] == "alignment-tag-weighted":   # any matcher: TagWeightedSlice wraps the slice (match/instantiate.cpp:173-189)
		args["tag_weighted"] = dict(
			tag_weights=metric["tag_weights"],
			pos_mismatch_penalty=float(metric.get("pos_mismatch_penalty", 0)),
			similarity_threshold=float(metric.get("similarity_threshold", 0)))
	return args, gaps
