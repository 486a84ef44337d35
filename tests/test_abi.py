"""CPU tier: the C-ABI library loads without a GPU, exports every symbol the header
declares, its pure-host entry point works, and compute entry points fail loudly."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
	text = open(os.path.join(ROOT, "include", "vectorian_hip.h")).read()
	text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
	return sorted(set(re.findall(r"\b(vk_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
	from vectorian_amd import core
	lib = core.lib()
	syms = declared_symbols()
	assert len(syms) >= 15
	missing = [s for s in syms if not hasattr(lib, s)]
	assert not missing, missing
	assert sorted(core.EXPORTS) == syms
	assert lib.vk_abi_version() == 5


def test_no_gpu_means_loud_failure():
	import torch
	if torch.cuda.is_available():
		pytest.skip("a GPU is present")
	from vectorian_amd import core
	assert core.device_count() == 0
	with pytest.raises(core.VkError):
		core.init(0)
	with pytest.raises(core.VkError):
		core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=8, n_tokens=4, n_sentences=1)


def test_merge_topk_is_resultset_extend():
	# ResultSet::extend (result_set.h:70-93): bounded merge, order score desc then sentence desc
	from vectorian_amd import core
	a, b = core.TopK(4, 3), core.TopK(4, 3)
	a.n = 3
	a.score[:3] = [0.9, 0.5, 0.5]; a.sentence[:3] = [7, 12, 3]
	a.mapping[:3] = [[0, 1, 2], [1, -1, 2], [-1, -1, 0]]
	b.n = 2
	b.score[:2] = [0.7, 0.5]; b.sentence[:2] = [100, 101]
	b.mapping[:2] = [[5, 6, 7], [9, -1, -1]]
	m = core.merge_topk([a, b], 3, 4)
	assert m.n == 4
	assert list(m.sentence[:4]) == [7, 100, 101, 12]
	assert list(m.mapping[1]) == [5, 6, 7] and list(m.mapping[3]) == [1, -1, 2]
	np.testing.assert_array_equal(m.score[:4], np.array([0.9, 0.7, 0.5, 0.5], np.float32))
