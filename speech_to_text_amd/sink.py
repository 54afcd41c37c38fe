"""Feature sinks and sources around the GPU path: the packed single-file variant of the
reference's per-utterance HDF5 files, readers that hand the trainer what it expects, and a
dataset that feeds device-resident features straight to a training loop.

The reference writes one file per utterance (/root/reference/src/preprocess.py:138-144) and its
trainer reads them back one by one into ``[mfccs, sample_rate, transcript, speaker_id]`` lists
(/root/reference/src/trainer.py:104-114).  At 10^5 utterances file creation dominates the wall
time once the features come off a GPU, so a shard can also be written as ONE file:

    mfccs          (sum F_u, n_mfcc) float32   all utterances back to back
    frame_offsets  (U + 1,) int64              utterance u owns rows [offsets[u], offsets[u+1])
    speaker_id, chapter_id, utterance_id, sample_rate   (U,) int64
    transcript     (U,) variable-length UTF-8
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from . import hdf5_min

PACKED_FIELDS = ("speaker_id", "chapter_id", "utterance_id", "sample_rate")


def write_packed(path, feats: Sequence[np.ndarray], metas: Sequence[dict], backend: Optional[str] = None) -> str:
    """feats[u]: (F_u, n_mfcc) float32; metas[u]: dict with the five attributes the reference
    stores per file.  backend: 'h5py', 'min' or None (= h5py when importable)."""
    if len(feats) != len(metas):
        raise ValueError("one metadata record per utterance")
    n_mfcc = feats[0].shape[1] if len(feats) else 0
    fo = np.zeros(len(feats) + 1, dtype=np.int64)
    np.cumsum([f.shape[0] for f in feats], out=fo[1:])
    allf = np.concatenate(feats, axis=0).astype(np.float32, copy=False) if len(feats) else np.zeros((0, 0), np.float32)
    cols = {k: np.asarray([int(m[k]) for m in metas], dtype=np.int64) for k in PACKED_FIELDS}
    texts = [str(m["transcript"]) for m in metas]
    if backend in (None, "h5py"):
        try:
            import h5py  # noqa: WPS433

            with h5py.File(path, "w") as f:
                ds = f.create_dataset("mfccs", data=allf)
                ds.attrs["n_mfcc"] = int(n_mfcc)
                f.create_dataset("frame_offsets", data=fo)
                for k, v in cols.items():
                    f.create_dataset(k, data=v)
                f.create_dataset("transcript", data=np.array(texts, dtype=object), dtype=h5py.string_dtype("utf-8"))
            return "h5py"
        except ImportError:
            if backend == "h5py":
                raise
    hdf5_min.write_datasets(path, [("mfccs", allf, {"n_mfcc": int(n_mfcc)}), ("frame_offsets", fo, None)] +
                            [(k, v, None) for k, v in cols.items()] + [("transcript", texts, None)])
    return "min"


class PackedReader:
    """Random access to a packed feature file: ``reader[u]`` is ``(mfccs (F_u, n_mfcc), attrs)``,
    the pair ``file['mfccs'][:]`` / ``file['mfccs'].attrs`` gives for a per-utterance file."""

    def __init__(self, path):
        self.path = str(path)
        try:
            import h5py  # noqa: WPS433

            with h5py.File(self.path, "r") as f:
                self._mfccs = f["mfccs"][:]
                self._fo = f["frame_offsets"][:]
                self._cols = {k: f[k][:] for k in PACKED_FIELDS}
                self._texts = [t.decode("utf-8") if isinstance(t, bytes) else str(t) for t in f["transcript"][:]]
        except ImportError:
            self._mfccs = hdf5_min.read_dataset(self.path, "mfccs")[0]
            self._fo = hdf5_min.read_dataset(self.path, "frame_offsets")[0]
            self._cols = {k: hdf5_min.read_dataset(self.path, k)[0] for k in PACKED_FIELDS}
            self._texts = hdf5_min.read_dataset(self.path, "transcript")[0]

    def __len__(self) -> int:
        return len(self._fo) - 1

    def __getitem__(self, u: int):
        if not -len(self) <= u < len(self):
            raise IndexError(u)
        u %= len(self)
        attrs = {k: int(self._cols[k][u]) for k in PACKED_FIELDS}
        attrs["transcript"] = self._texts[u]
        return self._mfccs[int(self._fo[u]): int(self._fo[u + 1])], attrs

    def __iter__(self):
        return (self[u] for u in range(len(self)))


def load_training_items(source, num_files: Optional[int] = None) -> List[list]:
    """The list the reference's trainer builds at /root/reference/src/trainer.py:104-114:
    ``[mfccs, sample_rate, transcript, speaker_id]`` per utterance, from a directory of
    per-utterance files (as the reference writes them) or from one packed file."""
    from .preprocess import read_feature_file

    src = Path(source)
    data = []
    if src.is_dir():
        for root, _dirs, files in os.walk(src):
            files = sorted(files)
            if num_files is not None:
                files = files[:num_files]
            for file in files:
                arr, attrs = read_feature_file(Path(root) / file)
                data.append([arr, attrs["sample_rate"], attrs["transcript"], attrs["speaker_id"]])
    else:
        for u, (arr, attrs) in enumerate(PackedReader(src)):
            if num_files is not None and u >= num_files:
                break
            data.append([arr, attrs["sample_rate"], attrs["transcript"], attrs["speaker_id"]])
    return data


class FeatureDataset:
    """Map-style dataset that skips the HDF5 round trip: the utterances of a LibriSpeech-style
    dataset go through the GPU once, the features stay on the device, and item ``i`` is the
    ``[mfccs, sample_rate, transcript, speaker_id]`` list the reference's training loop indexes
    (``item[0]`` and ``item[2]`` at /root/reference/src/trainer.py:158-160) with ``mfccs`` a
    ``(F, n_mfcc)`` device tensor.  ``extractor`` is injectable for host-only tests."""

    def __init__(self, data, *, device=None, n_fft=2048, hop_length=512, n_mfcc=13, max_batch_seconds=1800.0,
                 ingest="float32", extractor=None, decode_workers=None):
        from .ingest import DecodeAhead, iter_batches
        from .preprocess import Preprocessor, _waveform_to_numpy

        # the dataset is walked lazily, a batch of waveforms at a time (the features are 40 times
        # smaller than the audio and stay)
        budget = int(max_batch_seconds * Preprocessor.NOMINAL_SR)
        if hasattr(data, "__len__") and hasattr(data, "__getitem__"):
            workers = decode_workers or max(1, min(8, (os.cpu_count() or 2) - 1))
            stream = DecodeAhead(data.__getitem__, range(len(data)), lambda it: it[0].shape[-1], workers=workers,
                                 budget_samples=budget)
        else:
            stream = enumerate(data)
        feats, meta = {}, {}
        for idx, items in iter_batches(stream, lambda it: it[0].shape[-1], budget):
            chunk = [_waveform_to_numpy(it[0]).reshape(-1) for it in items]
            if extractor is not None:
                outs = extractor(chunk, n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc)
            else:
                from .feature import mfcc_batch

                outs = mfcc_batch(chunk, n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc, device=device,
                                  return_tensor=True, int16_if_exact=ingest == "int16")
            for i, it, o in zip(idx, items, outs):
                feats[int(i)] = o
                meta[int(i)] = [int(it[1]), str(it[2]), int(it[3])]
        order = sorted(feats)
        self._feats = [feats[i] for i in order]
        self._meta = [meta[i] for i in order]

    def __len__(self) -> int:
        return len(self._feats)

    def __getitem__(self, i: int) -> list:
        sr, text, spk = self._meta[i]
        return [self._feats[i], sr, text, spk]
