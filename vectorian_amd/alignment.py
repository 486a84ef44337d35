"""Span-similarity optimizers: the operator surface of `vectorian.alignment`.

Mirrors vectorian/alignment.py (same class names, constructor signatures and
`to_args` dictionaries) plus the gap-cost families of `pyalign.gaps`, which the
reference re-exports with `from pyalign.gaps import *` (vectorian/alignment.py:6).
pyalign itself is absent offline; the families are restated per SURVEY A.4:
  ConstantGapCost(u)         w(k) = u  (k >= 1)
  LinearGapCost(u)           w(k) = u k
  AffineGapCost(u, v)        w(k) = u + v k
  ExponentialGapCost(a, b)   w(k) = 1 - a^(-k b)
  smooth_gap_cost(k)         ExponentialGapCost(2, 1/k)  -- vectorian/interact.py:559-565;
                             reproduces the documented gap penalty 0.12944944 of
                             mkdocs/docs/introduction.md:174 for k = 5.
Gap tables are computed in float64 and cast to float32 (that is what reproduces the
documented known answer bit for bit).
"""

from typing import Dict, Union

import numpy as np

from vectorian_amd import core


# ---------------------------------------------------------------------------
# pyalign.gaps (restated)
# ---------------------------------------------------------------------------

class GapCost:
	def costs(self, n):
		"""float32 array w(0..n-1), w(0) = 0"""
		raise NotImplementedError()

	def to_special_case(self):
		"""{'linear': u} / {'affine': (u, v)} when an O(nm) solver applies, else {}"""
		return {}

	def to_tuple(self):
		raise NotImplementedError()

	def __call__(self, k):
		return float(self.costs(int(k) + 1)[int(k)])


class ConstantGapCost(GapCost):
	def __init__(self, u):
		self._u = float(u)

	def costs(self, n):
		c = np.full(n, self._u, dtype=np.float32)
		c[0] = 0
		return c

	def to_special_case(self):
		return {"linear": 0.0} if self._u == 0 else {}

	def to_tuple(self):
		return ("constant", self._u)


class LinearGapCost(GapCost):
	def __init__(self, u):
		self._u = float(u)

	def costs(self, n):
		return (self._u * np.arange(n, dtype=np.float64)).astype(np.float32)

	def to_special_case(self):
		return {"linear": self._u}

	def to_tuple(self):
		return ("linear", self._u)


class AffineGapCost(GapCost):
	def __init__(self, u, v):
		self._u, self._v = float(u), float(v)

	def costs(self, n):
		c = (self._u + self._v * np.arange(n, dtype=np.float64)).astype(np.float32)
		c[0] = 0
		return c

	def to_special_case(self):
		return {"affine": (self._u, self._v)}

	def to_tuple(self):
		return ("affine", self._u, self._v)


class ExponentialGapCost(GapCost):
	def __init__(self, base, rate):
		self._base, self._rate = float(base), float(rate)

	def costs(self, n):
		return (1.0 - np.power(self._base, -np.arange(n, dtype=np.float64) * self._rate)).astype(np.float32)

	def to_tuple(self):
		return ("exponential", self._base, self._rate)


def smooth_gap_cost(k):
	"""gap cost that reaches 0.5 after skipping k tokens (vectorian/interact.py:559-565)"""
	if k == 0:
		return ConstantGapCost(0)
	return ExponentialGapCost(2, 1.0 / k)


# ---------------------------------------------------------------------------
# vectorian.alignment
# ---------------------------------------------------------------------------

class Optimizer:
	def to_description(self, partition):
		raise NotImplementedError()

	def to_args(self, partition):
		raise NotImplementedError()


def coalesce_default_gap(gap):
	# vectorian/alignment.py:17-21
	return ConstantGapCost(0) if gap is None else gap


class Alignment(Optimizer):
	"""order-preserving matching of two token sequences (vectorian/alignment.py:24-38)"""

	def __init__(self, gap: Union[GapCost, Dict[str, GapCost]] = None):
		gap = coalesce_default_gap(gap)
		self._gap = gap
		if isinstance(gap, dict):
			if not all(k in ("s", "t") for k in gap.keys()):
				raise ValueError(gap)

	@property
	def gap(self):
		return self._gap

	def _make_args(self, locality, gaps):
		return {
			"algorithm": "pyalign",
			"options": {
				"locality": locality,
				"gap_cost": gaps
			}
		}


class OptimalTransport(Optimizer):
	pass


class GlobalAlignment(Alignment):
	"""Needleman-Wunsch (vectorian/alignment.py:50-97)"""

	def to_description(self, partition):
		return {"GlobalAlignment": {"gap": self._gap}}

	def to_args(self, partition):
		return self._make_args(core.pyalign.Locality.GLOBAL, self._gap)


class SemiGlobalAlignment(Alignment):
	"""end gaps free (vectorian/alignment.py:100-130)"""

	def to_description(self, partition):
		return {"SemiGlobalAlignment": {"gap": self._gap}}

	def to_args(self, partition):
		return self._make_args(core.pyalign.Locality.SEMIGLOBAL, self._gap)


class LocalAlignment(Alignment):
	"""Smith-Waterman / Waterman-Smith-Beyer (vectorian/alignment.py:133-187)"""

	def to_description(self, partition):
		return {"LocalAlignment": {"gap": self._gap}}

	def to_args(self, partition):
		return self._make_args(core.pyalign.Locality.LOCAL, self._gap)


class WordMoversDistance(OptimalTransport):
	"""variants of the (relaxed) Word Mover's Distance (vectorian/alignment.py:190-283)"""

	@staticmethod
	def wmd(variant="nbow", **kwargs):
		kwargs["builtin"] = f"wmd/{variant}"
		if variant == "bow":
			return WordMoversDistance(False, False, False, True, **kwargs)
		elif variant == "nbow":
			return WordMoversDistance(False, False, False, False, **kwargs)
		else:
			raise ValueError(variant)

	@staticmethod
	def rwmd(variant="nbow", **kwargs):
		kwargs["builtin"] = f"rwmd/{variant}"
		if variant == "nbow":
			return WordMoversDistance(True, True, True, True, **kwargs)
		elif variant == "nbow/distributed":
			return WordMoversDistance(True, False, True, True, **kwargs)
		elif variant == "bow/fast":
			return WordMoversDistance(True, True, False, False, **kwargs)
		else:
			raise ValueError(variant)

	def __init__(
		self, relaxed=True, injective=True, symmetric=False, normalize_bow=False,
		extra_mass_penalty=-1, builtin=None):

		self._options = {
			"relaxed": relaxed,
			"injective": injective,
			"normalize_bow": normalize_bow,
			"symmetric": symmetric,
			"extra_mass_penalty": extra_mass_penalty
		}
		self._builtin_name = builtin

	@property
	def builtin_name(self):
		return self._builtin_name

	def to_description(self, partition):
		return {"WordMoversDistance": self._options}

	def to_args(self, partition):
		return {
			"algorithm": "word-movers-distance",
			"relaxed": self._options["relaxed"],
			"injective": self._options["injective"],
			"symmetric": self._options["symmetric"],
			"normalize_bow": self._options["normalize_bow"],
			"extra_mass_penalty": self._options["extra_mass_penalty"]
		}


class WordRotatorsDistance(OptimalTransport):
	"""Word Rotator's Distance (vectorian/alignment.py:286-313)"""

	def __init__(self, normalize_magnitudes=True, extra_mass_penalty=-1):
		self._normalize_magnitudes = normalize_magnitudes
		self._extra_mass_penalty = extra_mass_penalty

	def to_description(self, partition):
		return {
			"WordRotatorsDistance": {
				"normalize_magnitudes": self._normalize_magnitudes,
				"extra_mass_penalty": self._extra_mass_penalty
			}
		}

	def to_args(self, partition):
		return {
			"algorithm": "word-rotators-distance",
			"normalize_magnitudes": self._normalize_magnitudes,
			"extra_mass_penalty": self._extra_mass_penalty
		}
