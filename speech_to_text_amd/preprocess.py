"""Host-side mirror of the reference's preprocessing driver.

``Preprocessor`` keeps the method names, argument meaning and file schema of
``SpeechToTextTrainer`` (/root/reference/src/preprocess.py:53-176) for the MFCC path:

    process_audio(item)           :131-144   one utterance -> mfcc/{spk}_{chap}_{utt}.hdf5
    preprocess(multiprocess)      :146-167   the loop over the dataset
    read_preprocessed_data()      :169-176   walk mfcc/ and print datasets + attrs

The arithmetic runs on the GPU in ragged batches; audio decode and resampling stay on the host
(the dataset is any iterable of LibriSpeech-style items, e.g. torchaudio.datasets.LIBRISPEECH as
the reference builds it at :75).  Where the reference parallelises over worker processes
(``multiprocess=True``), this driver parallelises the file writing over threads instead.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import hdf5_min
from .shard import lpt_shards, pack_batches

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
      every waveform of a batch lies on the int16 grid (bit-identical features, half the upload)
    """

    def __init__(self, data: Optional[Iterable[Item]] = None, out_dir: str = "mfcc", *, device=None,
                 rank: int = 0, world_size: int = 1, max_batch_seconds: float = 1800.0,
                 hdf5_backend: Optional[str] = None, extractor=None, ingest: str = "float32"):
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
        if ingest not in ("float32", "int16"):
            raise ValueError("ingest must be 'float32' or 'int16'")
        self.ingest = ingest

    # ------------------------------------------------------------------------------------
    def _features(self, waveforms: Sequence[np.ndarray]) -> List[np.ndarray]:
        if self._extract is not None:
            return self._extract(waveforms, n_fft=self.n_fft, hop_length=self.hop_length, n_mfcc=self.n_mfcc)
        from .feature import mfcc_batch

        return mfcc_batch(waveforms, n_fft=self.n_fft, hop_length=self.hop_length, n_mfcc=self.n_mfcc,
                          device=self.device, int16_if_exact=self.ingest == "int16")

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

    def preprocess(self, multiprocess: bool = False, verbose: bool = True, packed: bool = False) -> int:
        """/root/reference/src/preprocess.py:146-167: every item of ``self.data`` -> one file.
        Returns the number of utterances this rank wrote.  While the writer threads store batch
        k the GPU already works on batch k + 1.  ``packed=True`` writes this rank's utterances to
        one file ``out_dir/packed_{rank:05d}.hdf5`` instead (sink.write_packed)."""
        if self.data is None:
            print("Data is empty. Aborted.")
            return 0
        self.out_dir.mkdir(parents=True, exist_ok=True)
        items = list(self.data)
        lengths = np.asarray([_waveform_to_numpy(it[0]).shape[-1] for it in items], dtype=np.int64)
        costs = 1 + lengths // self.hop_length
        mine = lpt_shards(costs, self.world_size)[self.rank] if self.world_size > 1 else np.arange(len(items))
        sr = int(items[0][1]) if items else 16000
        batches = pack_batches(lengths[mine], int(self.max_batch_seconds * sr))
        # One writer thread: per file the writers spend more time in the interpreter (arguments,
        # byte strings) than in I/O, so more threads only add hand-overs of its lock -- measured
        # slower with 4 threads than with 1 for all three backends.
        n_threads = 1
        if verbose:
            print(f"Processing audio data... rank {self.rank}/{self.world_size}: {len(mine)} of {len(items)} "
                  f"utterances in {len(batches)} GPU batch(es), {n_threads} writer thread(s)")
        done = 0
        kept, kept_idx = [], []
        with ThreadPoolExecutor(max_workers=n_threads) as pool:
            pending = []
            for b in batches:
                idx = mine[b]
                feats = self._features([_waveform_to_numpy(items[i][0]) for i in idx])
                for f in pending:
                    f.result()
                if packed:
                    kept.extend(feats)
                    kept_idx.extend(int(i) for i in idx)
                    pending = []
                else:
                    # one task per writer thread: the writers are Python code, so a task per file
                    # costs more in hand-over than the file itself
                    pairs = [(items[i], fr) for i, fr in zip(idx, feats)]
                    step = -(-len(pairs) // n_threads)
                    pending = [pool.submit(self._write_many, pairs[k:k + step]) for k in range(0, len(pairs), step)]
                done += len(idx)
                if verbose:
                    print(f"Progress: {done}/{len(mine)}", end="\r")
            for f in pending:
                f.result()
        if packed:
            from .sink import write_packed

            metas = [dict(sample_rate=items[i][1], transcript=items[i][2], speaker_id=items[i][3],
                          chapter_id=items[i][4], utterance_id=items[i][5]) for i in kept_idx]
            write_packed(self.out_dir / f"packed_{self.rank:05d}.hdf5", kept, metas, backend=self.hdf5_backend)
        if verbose:
            print()
        return done

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
