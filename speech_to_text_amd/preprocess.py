"""Host-side mirror of the reference's preprocessing driver.

``Preprocessor`` keeps the method names, argument meaning and file schema of
``SpeechToTextTrainer`` (/root/reference/src/preprocess.py:53-176) for the MFCC path:

    process_audio(item)           :131-144   one utterance -> mfcc/{spk}_{chap}_{utt}.hdf5
    preprocess(multiprocess)      :146-167   the loop over the dataset
    read_preprocessed_data()      :169-176   walk mfcc/ and print datasets + attrs

The arithmetic runs on the GPU in ragged batches; audio decode and resampling stay on the host
(the dataset is any sequence or iterable of LibriSpeech-style items, e.g.
torchaudio.datasets.LIBRISPEECH as the reference builds it at :75).  The dataset is walked lazily,
as the reference walks it: a rank decodes only its own utterances (``data[i]`` in a few threads, a
bounded number of samples ahead of the GPU, results taken in completion order like
``imap_unordered`` at :160), cuts them into batches, and while the GPU works on one batch the next
is being decoded and the previous one's files are being written.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import hdf5_min
from .ingest import DecodeAhead, iter_batches, shard_indices

Item = Tuple  # (waveform (1, T) float32 tensor-or-ndarray, sample_rate, transcript, speaker, chapter, utterance)


def _waveform_to_numpy(samples) -> np.ndarray:
    if hasattr(samples, "detach"):          # torch.Tensor, as torchaudio yields
        samples = samples.detach().cpu().numpy()
    return np.asarray(samples)


def to_int16_pcm(samples: np.ndarray) -> Optional[np.ndarray]:
    """The int16 PCM a float waveform was decoded from (``value * 32768``), or None if the
    waveform is not on that grid.  torchaudio decodes 16-bit FLAC (LibriSpeech) to exactly such
    floats, so uploading the int16 form loses nothing and halves the PCIe traffic."""
    flat = np.asarray(samples).reshape(-1)
    if flat.dtype == np.int16:
        return flat
    scaled = flat.astype(np.float64) * 32768.0
    r = np.rint(scaled)
    if not np.array_equal(r, scaled) or r.size and (r.min() < -32768 or r.max() > 32767):
        return None
    return r.astype(np.int16)


def feature_file_name(speaker_id, chapter_id, utterance_id) -> str:
    """/root/reference/src/preprocess.py:138."""
    return f"{speaker_id}_{chapter_id}_{utterance_id}.hdf5"


def write_feature_file(path, mfcc_frames: np.ndarray, *, speaker_id, chapter_id, utterance_id,
                       sample_rate, transcript, backend: Optional[str] = None) -> str:
    """Dataset ``mfccs`` + the five attributes of /root/reference/src/preprocess.py:139-144.
    backend: 'h5py' (if importable), 'native' (the writer inside libmfcc_hip.so), 'min' (the same
    bytes from plain Python) or None = the first of these that is available."""
    attrs = dict(speaker_id=int(speaker_id), chapter_id=int(chapter_id), utterance_id=int(utterance_id),
                 sample_rate=int(sample_rate), transcript=str(transcript))
    if backend in (None, "h5py"):
        try:
            import h5py  # noqa: WPS433

            with h5py.File(path, "w") as f:
                ds = f.create_dataset("mfccs", data=mfcc_frames)
                for k, v in attrs.items():
                    ds.attrs[k] = v
            return "h5py"
        except ImportError:
            if backend == "h5py":
                raise
    if backend in (None, "native"):
        try:
            hdf5_min.write_features_native(path, mfcc_frames, attrs, name="mfccs")
            return "native"
        except ImportError:           # libmfcc_hip.so not built
            if backend == "native":
                raise
    hdf5_min.write_features(path, mfcc_frames, attrs, name="mfccs")
    return "min"


def read_feature_file(path):
    """(mfccs array, attrs dict) -- the reads of /root/reference/src/trainer.py:110-114."""
    try:
        import h5py  # noqa: WPS433

        with h5py.File(path, "r") as f:
            ds = f["mfccs"]
            attrs = {k: (v.decode() if isinstance(v, bytes) else (int(v) if np.ndim(v) == 0 and not isinstance(v, str) else v))
                     for k, v in ds.attrs.items()}
            return ds[:], attrs
    except ImportError:
        return hdf5_min.read_features(path, "mfccs")


class Preprocessor:
    """Arguments
    - data: iterable (ideally a sequence) of items
      ``(waveform, sample_rate, transcript, speaker_id, chapter_id, utterance_id)``
      (/root/reference/src/preprocess.py:132)
    - out_dir: where the per-utterance files go ('mfcc', relative to the CWD, in the reference)
    - rank / world_size: this process's shard of the utterances (one process per GPU)
    - ingest: 'float32' uploads the waveforms as they are; 'int16' uploads 16-bit PCM whenever
      every waveform of a batch lies on the int16 grid (bit-identical features, half the upload:
      what torchaudio yields for the 16-bit FLAC of /root/reference/src/preprocess.py:75 does);
      'auto' (the default) does that until a batch is found off the grid, then stops trying
    - lengths: optional manifest, samples per item (from file sizes or a metadata table): the
      ranks' shards are then balanced by frame count; without it items are dealt round-robin.
      Either way no audio is decoded to decide who takes what.
    - decode_workers: threads that run ``data[i]`` ahead of the GPU (default: up to 8)
    - max_batch_seconds: audio per GPU batch, counted at 16 kHz; at most about two batches of
      waveforms are alive at any time (the one on the GPU and the one being decoded)
    """

    NOMINAL_SR = 16000

    def __init__(self, data: Optional[Iterable[Item]] = None, out_dir: str = "mfcc", *, device=None,
                 rank: int = 0, world_size: int = 1, max_batch_seconds: float = 1800.0,
                 hdf5_backend: Optional[str] = None, extractor=None, ingest: str = "auto",
                 lengths: Optional[Sequence[int]] = None, decode_workers: Optional[int] = None):
        # MFCC meta parameters, as /root/reference/src/preprocess.py:60-62
        self.hop_length = 512
        self.n_fft = 2048
        self.n_mfcc = 13
        self.data = data
        self.out_dir = Path(out_dir)
        self.device = device
        self.rank, self.world_size = int(rank), int(world_size)
        self.max_batch_seconds = float(max_batch_seconds)
        self.hdf5_backend = hdf5_backend
        self._extract = extractor          # injectable for host-only tests
        if ingest not in ("auto", "float32", "int16"):
            raise ValueError("ingest must be 'auto', 'float32' or 'int16'")
        self.ingest = ingest
        self.lengths = lengths
        self.decode_workers = decode_workers
        self.stats: dict = {}

    # ------------------------------------------------------------------------------------
    def _features(self, waveforms: Sequence[np.ndarray]) -> List[np.ndarray]:
        if self._extract is not None:
            return self._extract(waveforms, n_fft=self.n_fft, hop_length=self.hop_length, n_mfcc=self.n_mfcc)
        from .feature import mfcc_batch

        return mfcc_batch(waveforms, n_fft=self.n_fft, hop_length=self.hop_length, n_mfcc=self.n_mfcc,
                          device=self.device, int16_if_exact=self.ingest != "float32")

    def _write(self, item: Item, frames: np.ndarray) -> None:
        _, sample_rate, transcript, speaker_id, chapter_id, utterance_id = item
        path = self.out_dir / feature_file_name(speaker_id, chapter_id, utterance_id)
        write_feature_file(path, frames, speaker_id=speaker_id, chapter_id=chapter_id,
                           utterance_id=utterance_id, sample_rate=sample_rate, transcript=transcript,
                           backend=self.hdf5_backend)

    def _write_many(self, pairs) -> None:
        if self.hdf5_backend in (None, "native") and pairs:
            use_native = self.hdf5_backend == "native"
            if not use_native:
                try:
                    import h5py  # noqa: F401,WPS433
                except ImportError:
                    use_native = True
            if use_native:
                try:
                    hdf5_min.write_many_features_native(
                        [self.out_dir / feature_file_name(it[3], it[4], it[5]) for it, _ in pairs],
                        [fr for _, fr in pairs],
                        [dict(speaker_id=int(it[3]), chapter_id=int(it[4]), utterance_id=int(it[5]),
                              sample_rate=int(it[1]), transcript=str(it[2])) for it, _ in pairs])
                    return
                except ImportError:           # libmfcc_hip.so not built
                    if self.hdf5_backend == "native":
                        raise
        for item, frames in pairs:
            self._write(item, frames)

    def process_audio(self, item: Item) -> None:
        """/root/reference/src/preprocess.py:131-144 for one item."""
        self.out_dir.mkdir(parents=True, exist_ok=True)
        samples = _waveform_to_numpy(item[0])
        frames = self._features([samples])[0]          # == librosa.feature.mfcc(y=samples)[0].T
        self._write(item, frames)

    def _decoded_stream(self, multiprocess: bool):
        """(index, item) pairs of this rank's utterances, plus the number it will yield (None if
        unknown) and the DecodeAhead behind it (None for a plain iterable)."""
        data = self.data
        budget = int(self.max_batch_seconds * self.NOMINAL_SR)
        if hasattr(data, "__len__") and hasattr(data, "__getitem__"):
            mine = shard_indices(len(data), self.rank, self.world_size, self.lengths, self.hop_length)
            workers = self.decode_workers or max(1, min(8, (os.cpu_count() or 2) - 1))
            if multiprocess:                       # the reference's pool size, /root/reference/src/preprocess.py:156
                workers = max(workers, min(32, (os.cpu_count() or 2) - 1 or 1))
            ahead = DecodeAhead(data.__getitem__, mine, lambda it: it[0].shape[-1], workers=workers,
                                budget_samples=budget)
            return ahead, len(mine), ahead
        # a plain iterable can only be walked in order: every rank walks it and keeps its share
        return ((i, it) for i, it in enumerate(data) if i % self.world_size == self.rank), None, None

    def preprocess(self, multiprocess: bool = False, verbose: bool = True, packed: bool = False) -> int:
        """/root/reference/src/preprocess.py:146-167: every item of ``self.data`` -> one file.
        Returns the number of utterances this rank wrote.  The dataset is never materialised:
        decode threads run ahead of the GPU by at most one batch of samples, the GPU takes one
        batch at a time, and a writer thread stores batch k while batch k + 1 is on the GPU.
        ``multiprocess=True`` raises the decode threads to the reference's pool size.
        ``packed=True`` writes this rank's utterances to one file
        ``out_dir/packed_{rank:05d}.hdf5`` instead (sink.write_packed)."""
        if self.data is None:
            print("Data is empty. Aborted.")
            return 0
        self.out_dir.mkdir(parents=True, exist_ok=True)
        self.stats = {}
        stream, n_mine, ahead = self._decoded_stream(multiprocess)
        budget = int(self.max_batch_seconds * self.NOMINAL_SR)
        if verbose:
            print(f"Processing audio data... rank {self.rank}/{self.world_size}: "
                  f"{n_mine if n_mine is not None else '?'} utterances, GPU batches of up to "
                  f"{self.max_batch_seconds:g} s of audio")
        batches = iter_batches(stream, lambda it: it[0].shape[-1], budget)
        if self._extract is None and self._native_files():
            done, n_batches, peak_batch, kept, kept_meta = self._run_overlapped(batches, packed, verbose, n_mine)
        else:
            done, n_batches, peak_batch, kept, kept_meta = self._run_simple(batches, packed, verbose, n_mine)
        if packed:
            from .sink import write_packed

            metas = [dict(sample_rate=m[0], transcript=m[1], speaker_id=m[2], chapter_id=m[3], utterance_id=m[4])
                     for m in kept_meta]
            write_packed(self.out_dir / f"packed_{self.rank:05d}.hdf5", kept, metas, backend=self.hdf5_backend)
        self.stats.update({"utterances": done, "batches": n_batches, "peak_batch_samples": peak_batch,
                           "batch_budget_samples": budget,
                           "decode_calls": ahead.n_fetched if ahead is not None else None,
                           "peak_decoded_ahead_samples": ahead.peak_ahead if ahead is not None else None})
        if verbose:
            print()
        return done

    def _native_files(self) -> bool:
        """Whether the per-utterance files come from the writer inside libmfcc_hip.so (no h5py, or
        ``hdf5_backend='native'``)."""
        if self.hdf5_backend == "native":
            return True
        if self.hdf5_backend is not None:
            return False
        try:
            import h5py  # noqa: F401,WPS433
        except ImportError:
            return True
        return False

    def _run_simple(self, batches, packed, verbose, n_mine):
        """One batch at a time through ``_features``; a writer thread stores batch k while batch
        k + 1 is being extracted.  (Injected extractors, h5py and the pure-Python writer.)"""
        # One writer thread: per file the writers spend more time in the interpreter (arguments,
        # byte strings) than in I/O, so more threads only add hand-overs of its lock.
        done = n_batches = peak_batch = 0
        kept, kept_meta = [], []
        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = None
            for _idx, items in batches:
                waves = [_waveform_to_numpy(it[0]) for it in items]
                peak_batch = max(peak_batch, sum(int(w.shape[-1]) for w in waves))
                feats = self._features(waves)
                metas = [it[1:] for it in items]                 # the waveforms can go now
                del waves, items
                if pending is not None:
                    pending.result()
                if packed:
                    kept.extend(feats)
                    kept_meta.extend(metas)
                else:
                    pairs = [((None,) + tuple(m), fr) for m, fr in zip(metas, feats)]
                    pending = pool.submit(self._write_many, pairs)
                done += len(feats)
                n_batches += 1
                if verbose:
                    print(f"Progress: {done}/{n_mine if n_mine is not None else '?'}", end="\r")
            if pending is not None:
                pending.result()
        return done, n_batches, peak_batch, kept, kept_meta

    def _run_overlapped(self, batches, packed, verbose, n_mine):
        """The GPU path: a staging thread packs and uploads batch k + 1 (pipeline.FeatureStream)
        while this thread computes batch k and a writer hands batch k - 1's pinned feature buffer to
        the native file writer, which creates the per-utterance files on several host threads
        without the interpreter."""
        import time

        from . import staging
        from .feature import make_config
        from .pipeline import features_of_batches

        cfg = make_config(n_fft=self.n_fft, hop_length=self.hop_length, n_mfcc=self.n_mfcc)
        # file creation is bound by the file system's per-directory lock, not by CPU: two threads
        # take what there is to take and leave the cores to the packing threads
        write_threads = int(os.environ.get("MFCC_WRITE_THREADS", "0")) or min(2, staging.host_threads())
        done = n_batches = peak_batch = 0
        kept, kept_meta = [], []
        t_write = [0.0]
        timeline: dict = {"write": []}

        peak = [0]

        def tagged():
            for _idx, items in batches:
                waves = [_waveform_to_numpy(it[0]) for it in items]
                peak[0] = max(peak[0], sum(int(w.shape[-1]) for w in waves))
                yield [it[1:] for it in items], waves

        def write(metas, host, fo, release):
            t0 = time.perf_counter()
            try:
                out = os.fspath(self.out_dir)
                paths = [f"{out}/{m[2]}_{m[3]}_{m[4]}.hdf5" for m in metas]          # feature_file_name
                ints = [(int(m[2]), int(m[3]), int(m[4]), int(m[0])) for m in metas]
                hdf5_min.write_feature_batch_native(
                    paths, host, fo, ("speaker_id", "chapter_id", "utterance_id", "sample_rate"), ints, "transcript",
                    [str(m[1]) for m in metas], n_threads=write_threads)
            finally:
                release()
                t_write[0] += time.perf_counter() - t0
                timeline["write"].append((t0, time.perf_counter()))

        with ThreadPoolExecutor(max_workers=1) as pool:
            pending = []
            for metas, host, fo, release in features_of_batches(tagged(), cfg, self.device,
                                                                try_int16={"auto": "auto", "int16": True, "float32": False}[self.ingest], timeline=timeline):
                if packed:
                    block = host.copy()
                    release()
                    kept.extend(block[int(a):int(b)] for a, b in zip(fo[:-1], fo[1:]))
                    kept_meta.extend(metas)
                else:
                    pending.append(pool.submit(write, metas, host, fo, release))
                    while len(pending) > 1:
                        pending.pop(0).result()
                done += len(metas)
                n_batches += 1
                if verbose:
                    print(f"Progress: {done}/{n_mine if n_mine is not None else '?'}", end="\r")
            for f in pending:
                f.result()
        self.stats["file_writer"] = {"threads": write_threads, "seconds_in_writer": t_write[0]}
        # seconds each stage was busy, and the wall clock from the first stage call to the last write
        self.stats["uploaded_as"] = dict(timeline.pop("uploads", {}))
        spans = [x for v in timeline.values() for x in v]
        self.stats["stage_seconds"] = {k: sum(b - a for a, b in v) for k, v in timeline.items()}
        self.stats["stage_seconds"]["wall"] = (max(b for _, b in spans) - min(a for a, _ in spans)) if spans else 0.0
        self.timeline = timeline
        return done, n_batches, peak[0], kept, kept_meta

    def read_preprocessed_data(self):
        """/root/reference/src/preprocess.py:169-176."""
        out = []
        for root, _dirs, files in os.walk(self.out_dir):
            for file in sorted(files):
                arr, attrs = read_feature_file(Path(root) / file)
                print(f'<HDF5 dataset "mfccs": shape {arr.shape}, type "<f4">')
                for k, v in attrs.items():
                    print(f"{k}: {v}")
                out.append((file, arr, attrs))
        return out
