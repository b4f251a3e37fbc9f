"""Rows and parameters of the sampling kept-set fixtures.  TEST INFRASTRUCTURE (see oracle/__init__.py).

The rows are drawn from numpy's PCG64 stream and rounded to the bfloat16 grid, so that the
generator (oracle/gen_golden.py, which asks HF transformers' TopKLogitsWarper / TopPLogitsWarper
which tokens survive), the numpy oracle and the GPU kernel all see the same values without the rows
being stored: tests/golden/sampling_kept_hf.npz holds only the survivors' bit masks."""
from __future__ import annotations

import numpy as np

from .ops_ref import round_to

# name -> (vocabulary, row kind, temperature, top_k, top_p)
CASES = {
    "flat_nucleus_beyond_1024": (16384, "flat", 1.5, -1, 0.9),
    "flat_topk5000_topp": (16384, "flat", 1.5, 5000, 0.95),
    "flat_topk2000": (16384, "flat", 1.0, 2000, 1.0),
    "peaked_topp": (16384, "peaked", 0.7, -1, 0.9),
    "peaked_topk50_topp": (16384, "peaked", 0.7, 50, 0.9),
    "peaked_topk1": (16384, "peaked", 1.0, 1, 1.0),
    "peaked_tiny_topp": (16384, "peaked", 1.0, -1, 0.01),
    "tied_topk100_keeps_every_tie": (16384, "tied", 1.0, 100, 1.0),
    "tied_topk3000_keeps_every_tie": (16384, "tied", 1.0, 3000, 1.0),
    "topk_is_the_vocabulary": (4096, "peaked", 1.0, 4096, 0.8),
    "llama3_vocab_flat_topp": (128256, "flat", 2.0, -1, 0.9),
    "llama3_vocab_peaked_topk_topp": (128256, "peaked", 0.8, 40, 0.95),
}


def row_of(name: str) -> np.ndarray:
    """float32 values on the bfloat16 grid."""
    import zlib
    V, kind, *_ = CASES[name]
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))
    x = rng.standard_normal(V).astype(np.float32)
    if kind == "flat":
        x = x * np.float32(0.3)
    elif kind == "peaked":
        x = x * np.float32(3.0)
    elif kind == "tied":                                   # a handful of levels: large groups of equal logits
        x = np.round(x * np.float32(2.0)) / np.float32(2.0)
    return round_to(x, "bfloat16")


def kept_sets_agree(row, a, b) -> bool:
    """Two kept sets of one row are the same set up to WHICH of the equal logits at the cut are in
    it: the same number of tokens, the same tokens wherever the logit differs from the smallest
    kept one, and the same number of tokens at that smallest value.  (Rows on the bfloat16 grid tie
    in dozens at any cut; HF takes the tied tokens in its sort's unspecified order, this engine
    lowest index first.)"""
    if int(a.sum()) != int(b.sum()) or not a.any():
        return False
    cut = row[a].min()
    if cut != row[b].min():
        return False
    off = row != cut
    return bool((a[off] == b[off]).all()) and int(a[~off].sum()) == int(b[~off].sum())
