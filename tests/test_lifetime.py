"""CPU tier: who frees a corpus handle, and when.

Round 3 recorded one segmentation fault inside CPython's garbage collector while another thread was inside vk_query.
Hazards removed since: finalizers no longer make GPU calls (core.Corpus.__del__ parks the handle for core.reap()), every native call
on a handle holds the handle's lock, and the library reference-counts what handles share (tests/test_gpu_lifetime.py).  The
reference keeps native results alive through shared_ptrs (vectorian/core/cpp/result_set.h:17-30)."""

import ctypes as C
import gc
import threading
import warnings

import numpy as np
import pytest

from vectorian_amd import core


class FakeLib:
	"""stands in for libvectorian_hip.so: records every vk_corpus_free with the thread it came from and whether the collector was running"""

	def __init__(self):
		self.freed = []
		self.in_gc = False

	def vk_corpus_free(self, h):
		self.freed.append((h.value, threading.get_ident(), self.in_gc))
		return 0


def _handle(value):
	c = core.Corpus.__new__(core.Corpus)
	c.lock = threading.RLock()
	c._h = C.c_void_p(value)
	return c


class Holder:
	"""an index-like object in a reference cycle with its corpus handles"""


@pytest.fixture
def fake(monkeypatch):
	lib = FakeLib()
	monkeypatch.setattr(core, "_lib", lib)
	monkeypatch.setattr(core, "lib", lambda: lib)
	core._graveyard.clear()
	yield lib
	core._graveyard.clear()


def test_cyclic_garbage_holding_an_index_makes_no_native_call(fake):
	gc.collect()
	h = Holder()
	h.corpus, h.views = _handle(0x1000), [_handle(0x2000), _handle(0x3000)]
	h.corpus.index = h           # the cycle: only the collector can free these
	for v in h.views:
		v.owner = h.corpus
	del h, v

	def cb(phase, info):
		fake.in_gc = phase == "start"
	gc.callbacks.append(cb)
	try:
		with warnings.catch_warnings(record=True) as seen:
			warnings.simplefilter("always")
			gc.collect()
	finally:
		gc.callbacks.remove(cb)
	assert fake.freed == []                                             # the finalizers parked the handles: no call into the library
	assert sorted(core._graveyard) == [0x1000, 0x2000, 0x3000]
	assert sum(issubclass(w.category, ResourceWarning) for w in seen) == 3   # ... and said so
	# the next safe point of a calling thread frees them, each once, outside the collector
	assert core.reap() == 3
	assert sorted(f[0] for f in fake.freed) == [0x1000, 0x2000, 0x3000] and not any(f[2] for f in fake.freed)
	assert core.reap() == 0 and len(fake.freed) == 3


def test_close_frees_once_and_finalizer_after_close_is_silent(fake):
	c = _handle(0x4000)
	c.close()
	c.close()
	assert [f[0] for f in fake.freed] == [0x4000]
	with warnings.catch_warnings(record=True) as seen:
		warnings.simplefilter("always")
		del c
		gc.collect()
	assert not seen and not core._graveyard and len(fake.freed) == 1


def test_close_waits_for_the_call_in_progress_on_its_handle(fake):
	"""close() takes the handle's lock: a native call in progress on the SAME handle (another thread) ends first"""
	c = _handle(0x5000)
	order = []
	inside, release = threading.Event(), threading.Event()

	def call():
		with c.lock:                       # what Corpus.query holds around vk_query
			inside.set()
			release.wait(5)
			order.append("call done")
	t = threading.Thread(target=call)
	t.start()
	inside.wait(5)
	closer = threading.Thread(target=lambda: (c.close(), order.append("closed")))
	closer.start()
	closer.join(0.2)
	assert closer.is_alive() and fake.freed == []      # blocked behind the call
	release.set()
	t.join(5); closer.join(5)
	assert order == ["call done", "closed"] and [f[0] for f in fake.freed] == [0x5000]


def test_context_managers_close():
	from tests.fake_backend import OracleCorpus
	from tests.test_host_api import toy_session
	from vectorian_amd import alignment
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	session, emb, words, rng = toy_session(n_docs=1, sents_per_doc=5, V=50, d=16)
	closed = []

	class Counting(OracleCorpus):
		def close(self):
			closed.append(self)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.2)))
	with session.partition("sentence").index(sim, corpus_factory=Counting) as index:
		assert len(index.find(" ".join(session.documents[0].tokens[:3]), n=2)) > 0
	assert len(closed) == 1
