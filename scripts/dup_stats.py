#!/usr/bin/env python3
"""How many of the K + K^2 table rows a pair's two-level tree gathers are DISTINCT (CPU, numpy): the reference's sampler draws K
neighbours WITH replacement whenever an entity has fewer than K edges (data_loader_user_set.py:375-388), so low-degree entities
repeat their neighbours.  usage: scripts/dup_stats.py   (the three dataset shapes of bench.py, synthetic KG with the notebook's degrees)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvin_amd import synth

for name, K in (("last-fm_50core", 32), ("MovieLens-1M", 16), ("amazon-book_20core", 64)):
    c = synth.dataset_case(name, K, 65536, seed=0)
    A = np.asarray(c.adj_entity)
    S = np.sort(A, axis=1)
    distinct = 1 + (S[:, 1:] != S[:, :-1]).sum(1)                 # distinct neighbours per adjacency row
    items = np.asarray(c.items)
    ch = A[items]
    within = (distinct[items] + distinct[ch].sum(1)).mean()        # every list deduplicated on its own
    both = np.mean([len(u) + distinct[u].sum() for u in (np.unique(ch[b]) for b in range(4000))])   # duplicate children once, too
    print("%-20s K=%3d  distinct neighbours per adjacency row %5.1f | rows per pair: faithful %5d, lists deduplicated %6.0f, "
          "duplicate children once %6.0f" % (name, K, distinct.mean(), K + K * K, within, both))
