"""Load order check (run on the GPU box): the HIP library first, torch afterwards -- both must see the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vectorian_amd import core, synth
core.init(0)
corpus = synth.make_contextual_corpus(500, 2, 30, 300, 48)
c = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=48, n_tokens=corpus["X"].shape[0], n_sentences=500)
c.append_vectors(corpus["X"]); c.set_sentences(corpus["sent_off"]); c.finalize()
top = c.query(synth.make_queries(corpus, 1, 5)[0]["vectors"], gap_s=0.1, gap_t=0.1)
import torch
assert torch.cuda.is_available(), "torch lost the GPU"
x = torch.ones(8, device="cuda").sum().item()
top2 = c.query(synth.make_queries(corpus, 1, 5)[0]["vectors"], gap_s=0.1, gap_t=0.1)
assert (top.score == top2.score).all() and x == 8.0
print("library first, torch second: ok")
