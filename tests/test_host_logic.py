"""Host-side logic without a GPU: sharding, HDF5 schema round trips, the preprocessing driver
(with the oracle injected as the feature extractor)."""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import mfcc_ref as R
from speech_to_text_amd import hdf5_min
from speech_to_text_amd.preprocess import Preprocessor, feature_file_name, read_feature_file, write_feature_file
from speech_to_text_amd.shard import lpt_shards, pack_batches, shard_loads
from tests.signals import ragged_lengths, speech_like


def oracle_extractor(waveforms, *, n_fft, hop_length, n_mfcc):
    return [R.process_audio_features(np.asarray(w), n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc) for w in waveforms]


def make_items(n, seed=0):
    lens = ragged_lengths(n, 0.2, 1.5, seed)
    return [(speech_like(int(L), seed=seed * 100 + i)[None, :], 16000, f"TRANSCRIPT NUMBER {i} ü", 1000 + i, 2000 + i, i)
            for i, L in enumerate(lens)]


# ---- sharding -----------------------------------------------------------------------------------
@pytest.mark.parametrize("n_ranks", [1, 2, 4, 8])
def test_lpt_shards_partition_and_balance(n_ranks):
    costs = 1 + ragged_lengths(1946, 2, 35, seed=1234) // 512
    shards = lpt_shards(costs, n_ranks)
    allidx = np.sort(np.concatenate(shards))
    assert np.array_equal(allidx, np.arange(len(costs)))          # a partition: nothing lost or doubled
    loads = shard_loads(costs, shards)
    assert loads.max() - loads.min() <= costs.max()               # LPT bound
    assert loads.max() <= 1.01 * costs.sum() / n_ranks + costs.max()
    again = lpt_shards(costs, n_ranks)
    assert all(np.array_equal(a, b) for a, b in zip(shards, again))   # deterministic


def test_lpt_edge_cases():
    assert [len(s) for s in lpt_shards([], 4)] == [0, 0, 0, 0]
    s = lpt_shards([5], 3)
    assert sorted(len(x) for x in s) == [0, 0, 1]
    with pytest.raises(ValueError):
        lpt_shards([1, 2], 0)


def test_pack_batches():
    b = pack_batches([10, 20, 30, 100, 5, 5], 50)
    assert [x.tolist() for x in b] == [[0, 1], [2], [3], [4, 5]]
    assert pack_batches([], 10) == []


# ---- HDF5 schema ----------------------------------------------------------------------------------
def test_hdf5_roundtrip_schema(tmp_path):
    a = np.random.default_rng(0).standard_normal((327, 13)).astype(np.float32)
    p = tmp_path / feature_file_name(1272, 128104, 3)
    assert p.name == "1272_128104_3.hdf5"
    write_feature_file(p, a, speaker_id=1272, chapter_id=128104, utterance_id=3, sample_rate=16000,
                       transcript="MISTER QUILTER IS THE APOSTLE OF THE MIDDLE CLASSES", backend="min")
    arr, attrs = hdf5_min.read_features(p)
    assert arr.dtype == np.float32 and arr.shape == (327, 13) and np.array_equal(arr, a)
    assert attrs == dict(speaker_id=1272, chapter_id=128104, utterance_id=3, sample_rate=16000,
                         transcript="MISTER QUILTER IS THE APOSTLE OF THE MIDDLE CLASSES")
    arr2, attrs2 = read_feature_file(p)
    assert np.array_equal(arr2, a) and attrs2["transcript"] == attrs["transcript"]


def test_hdf5_non_contiguous_input_and_empty(tmp_path):
    m = np.random.default_rng(1).standard_normal((13, 50)).astype(np.float32)
    write_feature_file(tmp_path / "t.hdf5", m.T, speaker_id=1, chapter_id=2, utterance_id=3, sample_rate=16000,
                       transcript="", backend="min")          # mfccs[0].T is a non-contiguous view (preprocess.py:136)
    arr, attrs = hdf5_min.read_features(tmp_path / "t.hdf5")
    assert np.array_equal(arr, m.T) and attrs["transcript"] == ""
    hdf5_min.write_features(tmp_path / "e.hdf5", np.zeros((0, 13), np.float32), {"speaker_id": -5})
    arr, attrs = hdf5_min.read_features(tmp_path / "e.hdf5")
    assert arr.shape == (0, 13) and attrs == {"speaker_id": -5}
    with pytest.raises(TypeError):
        hdf5_min.write_features(tmp_path / "x.hdf5", np.zeros((1, 1), np.float32), {"bad": 1.5})


def test_hdf5_is_readable_by_libhdf5(tmp_path):
    """When the HDF5 command-line tools exist, libhdf5 itself must read the file back."""
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if shutil.os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        pytest.skip("no h5dump on this machine")
    a = (np.arange(26, dtype=np.float32) / 4).reshape(2, 13)
    p = tmp_path / "1_2_3.hdf5"
    write_feature_file(p, a, speaker_id=1, chapter_id=2, utterance_id=3, sample_rate=16000, transcript="HELLO WORLD", backend="min")
    out = subprocess.run([h5dump, str(p)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    txt = out.stdout
    assert 'DATASET "mfccs"' in txt and "H5T_IEEE_F32LE" in txt and "( 2, 13 ) / ( 2, 13 )" in txt
    for name in ("speaker_id", "chapter_id", "utterance_id", "sample_rate", "transcript"):
        assert f'ATTRIBUTE "{name}"' in txt
    assert "H5T_STD_I64LE" in txt and "H5T_VARIABLE" in txt and "H5T_CSET_UTF8" in txt and '"HELLO WORLD"' in txt
    assert "6.25" in txt          # last element


# ---- the driver -------------------------------------------------------------------------------------------
def test_preprocessor_writes_reference_schema(tmp_path, capsys):
    items = make_items(7)
    pre = Preprocessor(items, out_dir=tmp_path / "mfcc", extractor=oracle_extractor, hdf5_backend="min")
    assert (pre.hop_length, pre.n_fft, pre.n_mfcc) == (512, 2048, 13)      # preprocess.py:60-62
    assert pre.preprocess(multiprocess=True, verbose=False) == 7
    files = sorted(p.name for p in (tmp_path / "mfcc").iterdir())
    assert files == sorted(f"{1000 + i}_{2000 + i}_{i}.hdf5" for i in range(7))
    for i, it in enumerate(items):
        arr, attrs = read_feature_file(tmp_path / "mfcc" / f"{1000 + i}_{2000 + i}_{i}.hdf5")
        want = R.process_audio_features(it[0])
        assert arr.shape == (1 + it[0].shape[-1] // 512, 13) and np.array_equal(arr, want)
        assert attrs == dict(speaker_id=1000 + i, chapter_id=2000 + i, utterance_id=i, sample_rate=16000,
                             transcript=f"TRANSCRIPT NUMBER {i} ü")
    listed = pre.read_preprocessed_data()
    assert len(listed) == 7 and "transcript: TRANSCRIPT NUMBER" in capsys.readouterr().out


def test_preprocessor_process_audio_and_sharding(tmp_path):
    items = make_items(9, seed=3)
    Preprocessor(out_dir=tmp_path / "one", extractor=oracle_extractor, hdf5_backend="min").process_audio(items[0])
    assert (tmp_path / "one" / "1000_2000_0.hdf5").exists()
    n = 0
    for r in range(2):
        n += Preprocessor(items, out_dir=tmp_path / "two", rank=r, world_size=2, extractor=oracle_extractor,
                          hdf5_backend="min", max_batch_seconds=1.0).preprocess(verbose=False)
    assert n == 9 and len(list((tmp_path / "two").iterdir())) == 9
    assert Preprocessor(None, out_dir=tmp_path / "none").preprocess(verbose=False) == 0


def test_int16_ingest_conversion():
    """Preprocessor(ingest='int16'): only waveforms on the int16 grid are converted."""
    from speech_to_text_amd.preprocess import to_int16_pcm

    pcm = np.array([-32768, -1, 0, 1, 32767], dtype=np.int16)
    assert np.array_equal(to_int16_pcm(pcm.astype(np.float32) / 32768.0), pcm)
    assert to_int16_pcm(pcm) is not None and to_int16_pcm(pcm).dtype == np.int16
    assert to_int16_pcm(np.array([0.1], dtype=np.float32)) is None            # off the grid
    assert to_int16_pcm(np.array([1.0], dtype=np.float32)) is None            # 32768 does not fit
    assert to_int16_pcm(np.zeros((1, 0), dtype=np.float32)).size == 0


# ---- packed file, trainer-side readers, direct dataset -------------------------------------------------
def test_packed_file_round_trip_and_h5dump(tmp_path):
    from speech_to_text_amd import PackedReader, load_training_items

    items = make_items(6, seed=5)
    n = Preprocessor(items, out_dir=tmp_path / "p", extractor=oracle_extractor, hdf5_backend="min",
                     max_batch_seconds=1.0).preprocess(verbose=False, packed=True)
    assert n == 6
    path = tmp_path / "p" / "packed_00000.hdf5"
    rd = PackedReader(path)
    assert len(rd) == 6
    seen = {}
    for arr, attrs in rd:
        seen[attrs["utterance_id"]] = (arr, attrs)
    for i, it in enumerate(items):
        arr, attrs = seen[i]
        assert np.array_equal(arr, R.process_audio_features(it[0])) and arr.dtype == np.float32
        assert attrs == dict(speaker_id=1000 + i, chapter_id=2000 + i, utterance_id=i, sample_rate=16000,
                             transcript=f"TRANSCRIPT NUMBER {i} ü")
    data = load_training_items(path)          # what /root/reference/src/trainer.py:104-114 builds
    assert len(data) == 6 and all(len(d) == 4 and d[1] == 16000 and d[2].startswith("TRANSCRIPT") for d in data)
    with pytest.raises(IndexError):
        rd[6]
    import shutil, subprocess
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if shutil.os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is not None:                    # libhdf5's own reader accepts the file
        out = subprocess.run([h5dump, "-H", str(path)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        for name in ("mfccs", "frame_offsets", "speaker_id", "chapter_id", "utterance_id", "sample_rate", "transcript"):
            assert f'DATASET "{name}"' in out.stdout
        assert "H5T_IEEE_F32LE" in out.stdout and "H5T_VARIABLE" in out.stdout


def test_load_training_items_from_per_utterance_files(tmp_path):
    from speech_to_text_amd import load_training_items

    items = make_items(4, seed=6)
    Preprocessor(items, out_dir=tmp_path / "mfcc", extractor=oracle_extractor, hdf5_backend="min").preprocess(verbose=False)
    data = load_training_items(tmp_path / "mfcc")
    assert len(data) == 4
    by_spk = {d[3]: d for d in data}
    for i, it in enumerate(items):
        d = by_spk[1000 + i]
        assert np.array_equal(d[0], R.process_audio_features(it[0])) and d[2] == f"TRANSCRIPT NUMBER {i} ü"
    assert len(load_training_items(tmp_path / "mfcc", num_files=2)) == 2


def test_feature_dataset_matches_trainer_item_layout():
    from speech_to_text_amd import FeatureDataset

    items = make_items(5, seed=8)
    ds = FeatureDataset(items, extractor=oracle_extractor, max_batch_seconds=1.0)
    assert len(ds) == 5
    for i, it in enumerate(items):
        mf, sr, text, spk = ds[i]
        assert np.array_equal(np.asarray(mf), R.process_audio_features(it[0]))
        assert (sr, text, spk) == (16000, f"TRANSCRIPT NUMBER {i} ü", 1000 + i)


def test_hdf5_min_many_strings_span_collections(tmp_path):
    texts = [f"utterance {i} " + "x" * (i % 50) for i in range(70000)]     # > 65535 heap objects
    hdf5_min.write_datasets(tmp_path / "s.hdf5", [("transcript", texts, None), ("ids", np.arange(70000), {"k": 3})])
    got, _ = hdf5_min.read_dataset(tmp_path / "s.hdf5", "transcript")
    assert got == texts
    ids, attrs = hdf5_min.read_dataset(tmp_path / "s.hdf5", "ids")
    assert np.array_equal(ids, np.arange(70000)) and attrs == {"k": 3}
    with pytest.raises(ValueError):
        hdf5_min.write_datasets(tmp_path / "n.hdf5", [(f"d{i}", np.zeros(1, np.float32), None) for i in range(9)])


def test_native_feature_file_writer_matches_python_writer(tmp_path):
    """mfcc_host_write_feature_file (inside libmfcc_hip.so) writes byte for byte what hdf5_min's
    Python writer does, for the reference's schema and the corner cases (no attributes, empty
    dataset, non-ASCII and long strings); libhdf5's h5dump reads the result when it is installed."""
    import shutil
    import subprocess

    from speech_to_text_amd import hdf5_min

    rng = np.random.default_rng(5)
    cases = [
        (rng.random((700, 13), dtype=np.float32), dict(speaker_id=1089, chapter_id=134686, utterance_id=7,
                                                       sample_rate=16000, transcript="HE HOPED THERE WOULD BE STEW")),
        (rng.random((1, 13), dtype=np.float32), dict(transcript="ünïcödé ✓ text", speaker_id=-5)),
        (np.zeros((0, 13), np.float32), dict(a=1)),
        (rng.random((33, 7), dtype=np.float32), {}),
        (rng.random((5, 3), dtype=np.float32), dict(t1="a", t2="", t3="x" * 5000, n=2 ** 40)),
    ]
    for i, (arr, attrs) in enumerate(cases):
        a, b = tmp_path / f"py{i}.hdf5", tmp_path / f"native{i}.hdf5"
        hdf5_min.write_features(a, arr, attrs)
        hdf5_min.write_features_native(b, arr, attrs)
        assert a.read_bytes() == b.read_bytes()
        back, at = hdf5_min.read_features(b, "mfccs")
        assert np.array_equal(back, arr) and at == attrs
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if Path("/opt/conda/bin/h5dump").exists() else None)
    if h5dump:
        out = subprocess.run([h5dump, "-H", str(tmp_path / "native0.hdf5")], capture_output=True, text=True)
        assert out.returncode == 0 and "DATASET \"mfccs\"" in out.stdout and "transcript" in out.stdout
    with pytest.raises(OSError):
        hdf5_min.write_features_native(tmp_path / "no_such_dir" / "x.hdf5", cases[0][0], cases[0][1])
    # many files in one call: the same bytes as one call per file
    arrays = [rng.random((n, 13), dtype=np.float32) for n in (3, 0, 250)]
    rows = [dict(speaker_id=10 + i, chapter_id=20, utterance_id=i, sample_rate=16000, transcript=f"TEXT {i} é")
            for i in range(3)]
    hdf5_min.write_many_features_native([tmp_path / f"many{i}.hdf5" for i in range(3)], arrays, rows)
    for i in range(3):
        hdf5_min.write_features(tmp_path / f"one{i}.hdf5", arrays[i], rows[i])
        assert (tmp_path / f"many{i}.hdf5").read_bytes() == (tmp_path / f"one{i}.hdf5").read_bytes()
    with pytest.raises(TypeError):
        hdf5_min.write_many_features_native([tmp_path / "a.hdf5", tmp_path / "b.hdf5"], arrays[:2],
                                            [dict(x=1), dict(x="s")])


class _SlowDataset:
    """Sequence whose __getitem__ stands in for a FLAC decode: it sleeps, counts its calls and
    hands out a fresh waveform every time (so nothing can be shared between calls)."""

    def __init__(self, n, seed=0, delay=0.002):
        import threading

        self.lens = ragged_lengths(n, 0.2, 1.5, seed=seed)
        self.calls = np.zeros(n, dtype=np.int64)
        self.delay = delay
        self._lock = threading.Lock()

    def __len__(self):
        return len(self.lens)

    def __getitem__(self, i):
        import time

        time.sleep(self.delay)
        with self._lock:
            self.calls[i] += 1
        return (speech_like(int(self.lens[i]), seed=1000 + i)[None, :], 16000, f"TEXT {i}", 3 + i, 7 + i, i)


def _oracle_extract(waveforms, *, n_fft, hop_length, n_mfcc):
    return [R.process_audio_features(np.asarray(w).reshape(-1), n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc)
            for w in waveforms]


@pytest.mark.parametrize("manifest", [False, True])
def test_streaming_preprocess_fetches_each_item_once_with_bounded_memory(tmp_path, manifest):
    """The driver walks its dataset lazily (as /root/reference/src/preprocess.py:155-167 does): over
    a whole world every item is decoded exactly once, a rank never holds more than about two
    batches of waveforms, and the files equal those of the one-item-at-a-time path."""
    n, world = 40, 2
    ds = _SlowDataset(n)
    budget_s = 4.0
    written = 0
    for rank in range(world):
        pre = Preprocessor(ds, out_dir=tmp_path / "mfcc", rank=rank, world_size=world, extractor=_oracle_extract,
                           hdf5_backend="min", max_batch_seconds=budget_s, decode_workers=3,
                           lengths=ds.lens if manifest else None)
        written += pre.preprocess(verbose=False)
        st = pre.stats
        assert st["batches"] >= 3                                    # really streamed in pieces
        assert st["peak_batch_samples"] <= st["batch_budget_samples"]
        # decoded ahead of the GPU: at most one batch, plus one chunk (8 utterances) per decode thread
        assert st["peak_decoded_ahead_samples"] <= st["batch_budget_samples"] + 3 * 8 * int(ds.lens.max())
        assert st["decode_calls"] == st["utterances"]
    assert written == n and (ds.calls == 1).all()
    files = sorted(os.listdir(tmp_path / "mfcc"))
    assert len(files) == n and not any(f.endswith(".tmp") for f in files)
    one = Preprocessor(None, out_dir=tmp_path / "single", extractor=_oracle_extract, hdf5_backend="min")
    for i in (0, 17, n - 1):
        item = _SlowDataset(n)[i]
        one.process_audio(item)
        name = feature_file_name(item[3], item[4], item[5])
        assert (tmp_path / "mfcc" / name).read_bytes() == (tmp_path / "single" / name).read_bytes()


def test_streaming_preprocess_plain_iterable_and_worker_error(tmp_path):
    """A generator (no len, no indexing) is walked in order by every rank, each keeping its share;
    an exception in a decode thread reaches the caller, as it does from imap_unordered."""
    ds = _SlowDataset(9, delay=0.0)
    total = 0
    for rank in range(2):
        pre = Preprocessor((ds[i] for i in range(len(ds))), out_dir=tmp_path / "gen", rank=rank, world_size=2,
                           extractor=_oracle_extract, hdf5_backend="min", max_batch_seconds=2.0)
        total += pre.preprocess(verbose=False)
    assert total == 9 and len(os.listdir(tmp_path / "gen")) == 9

    class Broken(_SlowDataset):
        def __getitem__(self, i):
            if i == 5:
                raise OSError("corrupt file")
            return super().__getitem__(i)

    pre = Preprocessor(Broken(9, delay=0.0), out_dir=tmp_path / "bad", extractor=_oracle_extract, hdf5_backend="min",
                       max_batch_seconds=2.0)
    with pytest.raises(OSError, match="corrupt file"):
        pre.preprocess(verbose=False)


def test_feature_dataset_streams_its_source():
    from speech_to_text_amd.sink import FeatureDataset

    ds = _SlowDataset(12, delay=0.0)
    fd = FeatureDataset(ds, extractor=_oracle_extract, max_batch_seconds=3.0, decode_workers=2)
    assert len(fd) == 12 and (ds.calls == 1).all()
    for i in (0, 5, 11):
        want = R.process_audio_features(speech_like(int(ds.lens[i]), seed=1000 + i))
        assert np.array_equal(fd[i][0], want) and fd[i][1:] == [16000, f"TEXT {i}", 3 + i]


def test_decode_ahead_never_loses_the_last_chunk(monkeypatch):
    """The end marker must not overtake a result: every worker's hand-over to the output queue is
    delayed (so the last chunk's put is still pending when nothing looks in flight any more, if
    the two were ordered wrongly) and every index must still come out, exactly once."""
    import queue
    import time

    from speech_to_text_amd.ingest import DecodeAhead

    real_put = queue.Queue.put

    def slow_put(self, item, *a, **kw):
        if isinstance(item, tuple):          # a worker's result (the end marker is a bare object)
            time.sleep(0.02)
        return real_put(self, item, *a, **kw)

    monkeypatch.setattr(queue.Queue, "put", slow_put)
    for n, workers, chunk in [(37, 8, 8), (5, 3, 2), (1, 4, 8), (64, 8, 3)]:
        da = DecodeAhead(lambda i: np.full(3, i), range(n), size_of=len, workers=workers, chunk=chunk)
        seen = sorted(i for i, r in da if int(r[0]) == i)
        assert seen == list(range(n)), (n, workers, chunk, seen)


def test_native_batch_file_writer_from_one_buffer(tmp_path):
    """mfcc_host_write_feature_files_mt through hdf5_min.write_feature_batch_native: the files of a
    batch written from the batch's one host buffer on several threads are byte for byte the files
    the one-file writer makes (the schema of /root/reference/src/preprocess.py:138-144, attributes
    in that order), an empty utterance and non-ASCII transcripts included; a bad directory is an
    OSError and leaves no partial directory behind."""
    from speech_to_text_amd import hdf5_min

    rng = np.random.default_rng(11)
    rows = [5, 0, 130, 1, 77, 300, 2, 9]
    fo = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    base = rng.random((int(fo[-1]) + 4, 13), dtype=np.float32)        # the buffer may be larger than the batch
    metas = [(16000, f"TEXT {i} é✓", 100 + i, 2000 + i, i) for i in range(len(rows))]
    for threads in (1, 4):
        out = tmp_path / f"mfcc{threads}"
        out.mkdir()
        paths = [f"{out}/{m[2]}_{m[3]}_{m[4]}.hdf5" for m in metas]
        hdf5_min.write_feature_batch_native(paths, base, fo, ("speaker_id", "chapter_id", "utterance_id", "sample_rate"),
                                            [(m[2], m[3], m[4], m[0]) for m in metas], "transcript",
                                            [m[1] for m in metas], n_threads=threads)
        assert sorted(p.name for p in out.iterdir()) == sorted(Path(p).name for p in paths)
        assert not (tmp_path / f"mfcc{threads}.partial").exists()
        for i, m in enumerate(metas):
            one = tmp_path / "one.hdf5"
            hdf5_min.write_features(one, base[fo[i]:fo[i + 1]],
                                    dict(speaker_id=m[2], chapter_id=m[3], utterance_id=m[4], sample_rate=m[0], transcript=m[1]))
            assert Path(paths[i]).read_bytes() == one.read_bytes()
    with pytest.raises(OSError):
        hdf5_min.write_feature_batch_native([str(tmp_path / "nowhere" / "a.hdf5")], base, fo[:2], ("speaker_id",), [(1,)],
                                            "transcript", ["x"], n_threads=2)
    with pytest.raises(ValueError):
        hdf5_min.write_feature_batch_native(["a"], base[:, ::2], fo[:2], ("speaker_id",), [(1,)], "transcript", ["x"])
    with pytest.raises(ValueError):
        hdf5_min.write_feature_batch_native(["a"], base, np.array([0, 10 ** 9]), ("speaker_id",), [(1,)], "transcript", ["x"])


def test_native_pack_range_matches_concatenation():
    """mfcc_host_pack_range: any sub-range [a, b) of the concatenation of ragged utterances (empty
    ones included) lands where numpy's concatenate puts it, for float32, int16 and float32 -> int16;
    a sample off the int16 grid is reported (0), not converted."""
    from speech_to_text_amd import _lib

    lib = _lib.lib()
    rng = np.random.default_rng(3)
    lens = [7, 0, 1000, 1, 0, 333, 4096, 5]
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    f32 = [(rng.integers(-32768, 32768, n).astype(np.float32) / np.float32(32768.0)) for n in lens]
    i16 = [np.rint(x * 32768.0).astype(np.int16) for x in f32]
    total = int(so[-1])
    for mode, src, want, dt in ((0, f32, np.concatenate(f32), np.float32), (1, i16, np.concatenate(i16), np.int16),
                                (2, f32, np.concatenate(i16), np.int16)):
        ptrs = np.array([x.__array_interface__["data"][0] for x in src], dtype=np.uint64)
        for c0, a, b in ((0, 0, total), (0, 5, 9), (6, 7, 1008), (1000, 1007, 1007), (7, 1007, total), (100, 1500, 5441)):
            dst = np.full(total - c0 + 3, 77, dtype=dt)
            rc = lib.mfcc_host_pack_range(ptrs.ctypes.data, so.ctypes.data, len(lens), dst.ctypes.data, c0, a, b, mode)
            assert rc == 1
            assert np.array_equal(dst[a - c0:b - c0], want[a:b])
            assert (dst[:a - c0] == 77).all() and (dst[b - c0:] == 77).all()
    bad = [x.copy() for x in f32]
    bad[2][500] += np.float32(1e-6)
    ptrs = np.array([x.__array_interface__["data"][0] for x in bad], dtype=np.uint64)
    dst = np.zeros(total, dtype=np.int16)
    assert lib.mfcc_host_pack_range(ptrs.ctypes.data, so.ctypes.data, len(lens), dst.ctypes.data, 0, 0, total, 2) == 0
    assert lib.mfcc_host_pack_range(ptrs.ctypes.data, so.ctypes.data, len(lens), dst.ctypes.data, 0, 0, 400, 2) == 1
    assert lib.mfcc_host_pack_range(ptrs.ctypes.data, so.ctypes.data, len(lens), dst.ctypes.data, 10, 5, 20, 0) < 0


def test_host_threads_respects_the_jobs_share(monkeypatch):
    from speech_to_text_amd import staging

    monkeypatch.setenv("OMP_NUM_THREADS", "4")
    assert staging.host_threads() == 3
    monkeypatch.setenv("OMP_NUM_THREADS", "1")
    assert staging.host_threads() == 1
    monkeypatch.delenv("OMP_NUM_THREADS")
    assert 1 <= staging.host_threads(5) <= 5
