"""Generates tests/golden/*.npz from the reference's own Python similarity code.

Runs ONLY in the build container (needs /root/reference).  The reference package
cannot be imported as a whole (gensim/h5py/spacy absent), so the
three numpy-only modules on the hot path are loaded by file path:
  vectorian/sim/kernel.py, vectorian/embedding/vectors.py, vectorian/sim/vector.py
with two harness stubs (cached_property -> functools.cached_property, empty h5py).
Only inputs and outputs (data) are written; no reference source is copied.
"""

import functools
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load(name, rel):
	spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
	mod = importlib.util.module_from_spec(spec)
	sys.modules[name] = mod
	spec.loader.exec_module(mod)
	return mod


def load_reference():
	cp = types.ModuleType("cached_property")
	cp.cached_property = functools.cached_property
	sys.modules.setdefault("cached_property", cp)
	sys.modules.setdefault("h5py", types.ModuleType("h5py"))
	for pkg in ("vectorian", "vectorian.sim", "vectorian.embedding"):
		if pkg not in sys.modules:
			m = types.ModuleType(pkg)
			m.__path__ = []
			sys.modules[pkg] = m
	_load("vectorian.sim.kernel", "vectorian/sim/kernel.py")
	vectors = _load("vectorian.embedding.vectors", "vectorian/embedding/vectors.py")
	vector = _load("vectorian.sim.vector", "vectorian/sim/vector.py")
	return vectors, vector


def main():
	vectors, vector = load_reference()
	rng = np.random.default_rng(20261003)
	os.makedirs(OUT, exist_ok=True)

	cases = {}
	for name, (n, m, d) in {"small": (7, 5, 300), "wide": (33, 10, 768), "tiny": (3, 2, 8)}.items():
		a = rng.standard_normal((n, d)).astype(np.float32)
		b = rng.standard_normal((m, d)).astype(np.float32)
		if name == "small":
			# clustered so that cosines spread over [-0.2, 0.9]; plus a duplicate row (cos = 1)
			b = (a[:m] + 0.6 * rng.standard_normal((m, d))).astype(np.float32)
			b[0] = a[0]
		if name == "tiny":
			a[1] = 0.0  # zero row: 0/0 -> NaN -> 0 (vectors.py:79)
		va, vb = vectors.Vectors(a), vectors.Vectors(b)
		out = np.zeros((n, m), dtype=np.float32)
		old = np.seterr(all="ignore")
		vector.CosineSim()(va, vb, out)
		np.seterr(**old)
		cases[name + "_a"] = a
		cases[name + "_b"] = b
		cases[name + "_a_mag"] = va.magnitudes
		cases[name + "_a_norm"] = va.normalized
		cases[name + "_b_norm"] = vb.normalized
		cases[name + "_cos"] = out
		# clip as SimilarityMatrix::clip does (metric/metric.h:28-30)
		cases[name + "_cos_clipped"] = np.clip(out, 0, 1)
	np.savez_compressed(os.path.join(OUT, "cosine_reference.npz"), **cases)
	print("wrote", os.path.join(OUT, "cosine_reference.npz"), {k: v.shape for k, v in cases.items()})


if __name__ == "__main__":
	main()
