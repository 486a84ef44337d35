import faulthandler
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)


# A fault of the test process (a native crash on the GPU box) must leave a record: faulthandler dumps the Python stack of every thread
# into a file under gpurun_out/incidents/ (merged back from the GPU box by gpurun; tools/keep_incidents.sh copies what is there into
# profiles/incidents/, which is tracked) as well as to stderr.  Round 3 lost the only evidence of a segmentation fault because
# nothing of the kind was kept.
_INCIDENT_DIR = os.path.join(ROOT, "gpurun_out", "incidents")
_incident_file = None


def pytest_configure(config):
	global _incident_file
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
	# tests let the garbage collector find their corpora; the product parks such handles and warns (core.Corpus.__del__)
	config.addinivalue_line("filterwarnings", "ignore:vectorian_amd Corpus was garbage-collected:ResourceWarning")
	try:
		os.makedirs(_INCIDENT_DIR, exist_ok=True)
		path = os.path.join(_INCIDENT_DIR, time.strftime("fault_%Y%m%d_%H%M%S") + f"_{os.getpid()}.log")
		_incident_file = open(path, "w")
		_incident_file.write(f"# faulthandler record of pytest pid {os.getpid()}, argv {sys.argv}\n")
		_incident_file.flush()
		faulthandler.enable(file=_incident_file, all_threads=True)
	except OSError:
		faulthandler.enable(all_threads=True)


def pytest_runtest_logstart(nodeid, location):
	# the test that was running when the process died: the last line of the record
	if _incident_file is not None:
		_incident_file.write(f"running {nodeid}\n")
		_incident_file.flush()


def pytest_sessionfinish(session, exitstatus):
	# a session that ends by itself leaves no incident: drop the record
	global _incident_file
	if _incident_file is not None:
		faulthandler.disable()
		name = _incident_file.name
		_incident_file.close()
		_incident_file = None
		try:
			os.remove(name)
		except OSError:
			pass


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
