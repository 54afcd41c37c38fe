"""Utterance sharding across GPUs.

The reference's only parallelism is ``Pool.imap_unordered(process_audio, data)`` over
independent utterances (/root/reference/src/preprocess.py:155-161).  Here the same independent
units are dealt to one process per GPU; there is no collective on the data path.
"""
from __future__ import annotations

import heapq
from typing import List, Sequence

import numpy as np


def lpt_shards(costs: Sequence[int], n_ranks: int) -> List[np.ndarray]:
    """Longest-processing-time-first assignment: sort units by cost (frame count) descending and
    give each to the least-loaded rank.  Returns, per rank, the unit indices in ascending order.
    Deterministic: ties are broken by unit index and rank index."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    costs = np.asarray(costs, dtype=np.int64)
    order = np.lexsort((np.arange(len(costs)), -costs))
    heap = [(0, r) for r in range(n_ranks)]
    heapq.heapify(heap)
    out: List[List[int]] = [[] for _ in range(n_ranks)]
    for i in order:
        load, r = heapq.heappop(heap)
        out[r].append(int(i))
        heapq.heappush(heap, (load + int(costs[i]), r))
    return [np.asarray(sorted(x), dtype=np.int64) for x in out]


def shard_loads(costs: Sequence[int], shards: Sequence[np.ndarray]) -> np.ndarray:
    costs = np.asarray(costs, dtype=np.int64)
    return np.asarray([int(costs[s].sum()) for s in shards], dtype=np.int64)


def pack_batches(lengths: Sequence[int], max_samples: int) -> List[np.ndarray]:
    """Greedy packing of a rank's utterances, in order, into resident batches of at most
    ``max_samples`` samples (an utterance longer than that gets a batch of its own)."""
    batches: List[List[int]] = []
    cur: List[int] = []
    tot = 0
    for i, n in enumerate(lengths):
        n = int(n)
        if cur and tot + n > max_samples:
            batches.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += n
    if cur:
        batches.append(cur)
    return [np.asarray(b, dtype=np.int64) for b in batches]
