"""Host ingest: which utterances a rank takes, and a bounded decode-ahead pipeline.

The reference walks its dataset lazily: ``Pool.imap_unordered(process_audio, self.data)``
decodes item after item in the parent and hands each to a worker
(/root/reference/src/preprocess.py:155-161); the serial path does the same one item at a time
(:165-167).  Here ``data[i]`` (the FLAC decode, on the host CPU) runs in a small thread pool, a
bounded number of samples ahead of the GPU, and results are taken in completion order, as
``imap_unordered`` yields them.  Nothing is decoded twice and nothing is decoded that another
rank owns.
"""
from __future__ import annotations

import queue
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np

from .shard import lpt_shards


def shard_indices(n_items: int, rank: int, world_size: int, lengths: Optional[Sequence[int]] = None,
                  hop_length: int = 512) -> np.ndarray:
    """Indices of the utterances rank ``rank`` of ``world_size`` processes, without touching the
    audio.  With a length manifest (``lengths[i]`` = samples of item i: file sizes, a metadata
    table, a previous pass) the shards are balanced by frame count, longest first; without one
    the items are dealt round-robin, which balances a shuffled or length-sorted corpus to within
    one utterance per rank.  Every rank computes the same partition."""
    if not 0 <= rank < world_size:
        raise ValueError("rank must be in [0, world_size)")
    if lengths is not None:
        lengths = np.asarray(lengths, dtype=np.int64)
        if lengths.shape != (n_items,):
            raise ValueError("the length manifest must have one entry per item")
        if world_size == 1:
            return np.arange(n_items, dtype=np.int64)
        return lpt_shards(1 + lengths // hop_length, world_size)[rank]
    return np.arange(rank, n_items, world_size, dtype=np.int64)


class DecodeAhead:
    """Iterates ``(index, fetch(index))`` over ``indices`` in completion order.  ``fetch`` runs in
    ``workers`` threads; a new fetch starts only while the samples of the results that have been
    fetched but not yet taken stay below ``budget_samples`` (so at most that many samples, plus
    one chunk of utterances per worker, wait in host memory).  ``size_of(result)`` gives a result's samples.

    Indices are handed to the workers ``chunk`` at a time (one task, one hand-over per chunk: for
    cheap fetches the hand-over, not the fetch, is the cost).

    ``peak_ahead`` records the most samples that ever waited; ``n_fetched`` the number of fetches."""

    _END = object()

    def __init__(self, fetch: Callable[[int], object], indices: Iterable[int], size_of: Callable[[object], int],
                 workers: int = 4, budget_samples: int = 1 << 26, chunk: int = 8):
        self._fetch, self._size_of = fetch, size_of
        self._indices = iter(indices)
        self._workers = max(1, int(workers))
        self._budget = max(1, int(budget_samples))
        self._chunk = max(1, int(chunk))
        self._out: "queue.Queue" = queue.Queue()
        self._cv = threading.Condition()
        self._ahead = 0            # samples fetched and not yet taken
        self._inflight = 0
        self._error: Optional[BaseException] = None
        self._stop = False
        self.peak_ahead = 0
        self.n_fetched = 0
        self._pool = ThreadPoolExecutor(max_workers=self._workers, thread_name_prefix="decode")
        self._feeder = threading.Thread(target=self._feed, name="decode-feeder", daemon=True)
        self._feeder.start()

    def _feed(self) -> None:
        try:
            todo = []
            for i in self._indices:
                todo.append(int(i))
                if len(todo) < self._chunk:
                    continue
                if not self._submit(todo):
                    todo = []
                    break
                todo = []
            if todo:
                self._submit(todo)
        except BaseException as e:                       # the index iterable itself failed
            with self._cv:
                self._error = self._error or e
        finally:
            with self._cv:
                while self._inflight > 0:
                    self._cv.wait()
            self._out.put(self._END)

    def _submit(self, idx) -> bool:
        with self._cv:
            while not self._stop and (self._inflight >= self._workers or self._ahead >= self._budget):
                self._cv.wait()
            if self._stop:
                return False
            self._inflight += 1
        self._pool.submit(self._run, idx)
        return True

    def _run(self, idx) -> None:
        try:
            res = [self._fetch(i) for i in idx]
            sizes = [int(self._size_of(r)) for r in res]
            with self._cv:
                # The result is queued BEFORE the chunk stops counting as in flight: the feeder
                # queues its end marker once nothing is in flight, and a result queued after that
                # marker would be lost.
                self._out.put((idx, res, sizes))
                self._inflight -= 1
                self._ahead += sum(sizes)
                self.peak_ahead = max(self.peak_ahead, self._ahead)
                self.n_fetched += len(idx)
                self._cv.notify_all()
        except BaseException as e:                       # a worker's exception reaches the consumer,
            with self._cv:                               # as it does from Pool.imap_unordered
                self._inflight -= 1
                self._error = self._error or e
                self._stop = True
                self._cv.notify_all()

    def __iter__(self) -> Iterator[Tuple[int, object]]:
        try:
            while True:
                got = self._out.get()
                if got is self._END:
                    break
                idx, res, sizes = got
                for i, r, n in zip(idx, res, sizes):
                    with self._cv:
                        self._ahead -= n
                        if self._ahead < self._budget:
                            self._cv.notify_all()
                    yield i, r
            if self._error is not None:
                raise self._error
        finally:
            self.close()

    def close(self) -> None:
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        self._pool.shutdown(wait=True)


def iter_batches(decoded: Iterable[Tuple[int, object]], size_of: Callable[[object], int], max_samples: int):
    """Cuts a stream of ``(index, item)`` into batches of at most ``max_samples`` samples (an item
    longer than that forms a batch of its own).  Yields ``(indices, items)``."""
    idx, items, tot = [], [], 0
    for i, it in decoded:
        n = int(size_of(it))
        if items and tot + n > max_samples:
            yield idx, items
            idx, items, tot = [], [], 0
        idx.append(i)
        items.append(it)
        tot += n
    if items:
        yield idx, items
