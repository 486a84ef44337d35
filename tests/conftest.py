import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
	"""CPU restatement of the reference algorithm -- the checker, never the product."""
	from oracle import vk_oracle
	vk_oracle.lib()
	return vk_oracle


@pytest.fixture(scope="session")
def hip():
	"""the product's C-ABI binding, initialised on device 0 (fails loudly without a GPU)"""
	from vectorian_amd import core
	core.init(0)
	return core
