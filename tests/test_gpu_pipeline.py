"""GPU tests of everything around the kernels: the raw C ABI (ctypes + device pointers), the
preprocessing driver writing the reference's HDF5 schema, and the CMVN extension kernels."""
import ctypes as C

import numpy as np
import pytest

from oracle import mfcc_ref as R
from tests.signals import ragged_lengths, speech_like

pytestmark = pytest.mark.gpu


def test_raw_c_abi(gpu):
    """Calls include/mfcc_hip.h directly: config -> plan -> layout -> workspace -> execute."""
    import torch

    from speech_to_text_amd import _lib

    lib = _lib.lib()
    c = _lib.MfccConfigC()
    lib.mfcc_config_default(C.byref(c))
    c.n_mfcc = 13
    plan = C.c_void_p()
    _lib.check(lib.mfcc_plan_create(C.byref(c), C.byref(plan)))
    try:
        ys = [speech_like(n, seed=i) for i, n in enumerate([30000, 2048, 51234])]
        so = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
        fo = np.zeros_like(so)
        _lib.check(lib.mfcc_batch_layout(C.byref(c), so.ctypes.data_as(C.POINTER(C.c_int64)), 3,
                                         fo.ctypes.data_as(C.POINTER(C.c_int64))))
        F = int(fo[-1])
        pcm = torch.from_numpy(np.concatenate(ys)).to(gpu)
        d_so, d_fo = torch.from_numpy(so).to(gpu), torch.from_numpy(fo).to(gpu)
        out = torch.full((F, 13), float("nan"), device=gpu)
        mel = torch.empty((F, 128), device=gpu)
        need = lib.mfcc_workspace_bytes(plan, F, 3)
        ws = torch.empty(need, dtype=torch.uint8, device=gpu)
        # too small a workspace is an error, not a fault
        rc = lib.mfcc_execute(plan, pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), None, 3, F,
                              out.data_ptr(), None, ws.data_ptr(), 16, None)
        assert rc == _lib.MFCC_E_WORKSPACE and "workspace" in _lib.last_error()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.mfcc_execute(plan, pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), None, 3, F,
                                    out.data_ptr(), mel.data_ptr(), ws.data_ptr(), need, stream))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i, y in enumerate(ys):
            np.testing.assert_allclose(got[fo[i]:fo[i + 1]], R.process_audio_features(y), rtol=0, atol=1e-2)
        melh = mel.cpu().numpy()
        want = R.melspectrogram(y=ys[0], n_fft=2048, hop_length=512).T
        assert (np.abs(melh[:fo[1]] - want) <= 1e-4 * want + 2e-7 * want.max()).all()
        # an empty batch is a no-op
        assert lib.mfcc_execute(plan, None, None, None, None, 0, 0, None, None, None, 0, None) == 0
    finally:
        lib.mfcc_plan_destroy(plan)


def test_driver_writes_reference_files(gpu, tmp_path):
    """Preprocessor.preprocess on the GPU: same files, names, dataset and attributes as
    /root/reference/src/preprocess.py:131-167 would write."""
    import speech_to_text_amd as S

    lens = ragged_lengths(12, 0.5, 4.0, seed=8)
    items = [(speech_like(int(L), seed=40 + i, int16_grid=True)[None, :], 16000, f"UTTERANCE {i}", 100 + i, 200 + i, i)
             for i, L in enumerate(lens)]
    pre = S.Preprocessor(items, out_dir=tmp_path / "mfcc", hdf5_backend="min", max_batch_seconds=10.0)
    assert pre.preprocess(multiprocess=True, verbose=False) == 12
    for i, it in enumerate(items):
        arr, attrs = S.read_feature_file(tmp_path / "mfcc" / f"{100 + i}_{200 + i}_{i}.hdf5")
        want = R.process_audio_features(it[0])
        assert arr.dtype == np.float32 and arr.shape == want.shape
        np.testing.assert_allclose(arr, want, rtol=0, atol=1e-2)
        assert attrs == dict(speaker_id=100 + i, chapter_id=200 + i, utterance_id=i, sample_rate=16000,
                             transcript=f"UTTERANCE {i}")
    pre.process_audio(items[0])      # the single-item entry point overwrites in place ('w' mode)


def test_cmvn_kernels(gpu):
    import torch

    import speech_to_text_amd as S

    rng = np.random.default_rng(3)
    x = (rng.standard_normal((50_001, 13)) * np.linspace(1, 30, 13) + np.linspace(-400, 40, 13)).astype(np.float32)
    d = torch.from_numpy(x).to(gpu)
    stats = S.cmvn.new_stats(13, gpu)
    S.cmvn.accumulate(d[:20_000].contiguous(), stats)
    S.cmvn.accumulate(d[20_000:].contiguous(), stats)
    S.cmvn.allreduce(stats)                                  # single process: no-op
    s = stats.cpu().numpy()
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(s[:13], x64.sum(0), rtol=1e-12)
    np.testing.assert_allclose(s[13:26], (x64 ** 2).sum(0), rtol=1e-12)
    assert s[26] == x.shape[0]
    mean, std = S.cmvn.mean_std(stats)
    S.cmvn.apply(d, stats)
    got = d.cpu().numpy()
    np.testing.assert_allclose(got, (x64 - mean) / std, rtol=0, atol=2e-5)
    assert abs(got.mean(0)).max() < 1e-4 and abs(got.std(0) - 1).max() < 1e-4


def test_asr_frontend_with_cmvn(gpu):
    """BASELINE config 4 end to end on one GPU: n_fft=400 (and the n_fft=512 / win_length=400
    spelling), hop=160, 80 mels, 13 MFCC, global CMVN."""
    import torch

    import speech_to_text_amd as S

    ys = [speech_like(int(n), seed=70 + i) for i, n in enumerate(ragged_lengths(6, 0.5, 3.0, seed=2))]
    for kw in (dict(n_fft=400), dict(n_fft=512, win_length=400)):
        feats = S.mfcc_batch(ys, sr=16000, hop_length=160, n_mels=80, n_mfcc=13, return_tensor=True, **kw)
        want = [R.mfcc(y=y, sr=16000, hop_length=160, n_mels=80, n_mfcc=13, **kw).T for y in ys]
        for f, w in zip(feats, want):
            assert f.shape == w.shape == (1 + len(ys[0]) // 160, 13) or f.shape == w.shape
            np.testing.assert_allclose(f.cpu().numpy(), w, rtol=0, atol=1e-2)
        allf = torch.cat(feats).contiguous()
        stats = S.cmvn.new_stats(13, allf.device)
        S.cmvn.accumulate(allf, stats)
        S.cmvn.apply(allf, stats)
        w = np.concatenate(want).astype(np.float64)
        np.testing.assert_allclose(allf.cpu().numpy(), (w - w.mean(0)) / w.std(0), rtol=0, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512, n_mfcc=13),
    dict(n_fft=2048, hop_length=512, n_mfcc=13, pad_mode="reflect"),
    dict(n_fft=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=1024, hop_length=256, n_mfcc=13),
    dict(n_fft=256, hop_length=80, n_mels=40, n_mfcc=13, sr=8000),
    dict(n_fft=640, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=960, hop_length=480, n_mels=80, n_mfcc=13, sr=48000),
    # wide filters (4 Hz bins): the mel table no longer fits beside 16 wavefronts, and at hop 512
    # the float path takes the variant that keeps samples in registers, which has no int16 build
    dict(n_fft=2048, hop_length=512, n_mels=128, n_mfcc=13, sr=8000),
    dict(n_fft=2048, hop_length=512, n_mels=80, n_mfcc=13, sr=8000),
    dict(n_fft=2048, hop_length=512, n_mels=128, n_mfcc=13, sr=16000),
])
def test_int16_pcm_ingest_is_bit_identical(gpu, kw):
    """16-bit PCM in (value / 32768, the waveform torchaudio decodes a LibriSpeech FLAC into for
    /root/reference/src/preprocess.py:132-134) gives exactly the float32 path's output."""
    import warnings

    import speech_to_text_amd as S

    rng = np.random.default_rng(7)
    lens = [40000, 3, 2047, 2048, 70001, 1]
    ys16 = [np.round(np.clip(speech_like(n, seed=20 + i), -0.999, 0.999) * 32767).astype(np.int16)
            for i, n in enumerate(lens)]
    ys16[1][:] = [-32768, 32767, 0]                     # the extremes of the grid
    ysf = [y.astype(np.float32) / 32768.0 for y in ys16]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = S.mfcc_batch(ys16, **kw)
        b = S.mfcc_batch(ysf, **kw)
    assert len(a) == len(b) == len(lens)
    for x, y in zip(a, b):
        assert x.shape == y.shape and np.array_equal(x, y)
    del rng


@pytest.mark.gpu
@pytest.mark.parametrize("kw,name", [
    (dict(n_fft=2048, hop_length=512), "fft2048"),
    (dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, sr=16000), "fft512"),
    (dict(n_fft=1024, hop_length=256), "fft1024"),
    (dict(n_fft=256, hop_length=80, n_mels=40, sr=8000), "fft256"),
    (dict(n_fft=400, hop_length=160, n_mels=80, sr=16000), "fft400"),
    (dict(n_fft=320, hop_length=160, n_mels=40, sr=16000), "fft320"),
    (dict(n_fft=480, hop_length=160, n_mels=80, sr=16000), "fft480"),
    (dict(n_fft=640, hop_length=160, n_mels=80, sr=16000), "fft640"),
    (dict(n_fft=800, hop_length=320, n_mels=80, sr=32000), "fft800"),
    (dict(n_fft=960, hop_length=480, n_mels=80, sr=48000), "fft960"),
    (dict(n_fft=600, hop_length=150, n_mels=80, sr=16000), "dense_dft"),
    (dict(n_fft=128, hop_length=64, n_mels=20, sr=8000), "fft128"),
    (dict(n_fft=160, hop_length=80, n_mels=23, sr=8000), "fft160"),
    (dict(n_fft=192, hop_length=96, n_mels=32, sr=16000), "fft192"),
    (dict(n_fft=384, hop_length=160, n_mels=80, sr=16000), "fft384"),
    (dict(n_fft=576, hop_length=160, n_mels=80, sr=22050), "fft576"),
    (dict(n_fft=768, hop_length=256, n_mels=128, sr=48000), "fft768"),
    (dict(n_fft=224, hop_length=112, n_mels=40, sr=16000), "fft224"),
    (dict(n_fft=448, hop_length=160, n_mels=80, sr=16000), "fft448"),
    (dict(n_fft=992, hop_length=480, n_mels=80, sr=48000), "fft992"),
    (dict(n_fft=240, hop_length=80, n_mels=40, sr=8000), "fft240"),         # 16 x an odd number
    (dict(n_fft=976, hop_length=480, n_mels=128, sr=44100), "fft976"),
    (dict(n_fft=200, hop_length=80, n_mels=40, sr=8000), "dense_dft"),      # not a multiple of 16
    (dict(n_fft=4096, hop_length=1024), None),
    (dict(n_fft=2047, hop_length=512), None),
])
def test_front_kernel_selection(gpu, kw, name):
    import speech_to_text_amd as S

    if name is None:       # lengths no kernel takes: refused as unsupported, not as invalid parameters
        from speech_to_text_amd import _lib

        with pytest.raises(_lib.LibraryError) as e:
            S.MfccPlan(S.make_config(n_mfcc=13, **kw))
        assert e.value.code == _lib.MFCC_E_UNSUPPORTED
        return
    plan = S.MfccPlan(S.make_config(n_mfcc=13, **kw))
    assert plan.front_kernel == name
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=1024, hop_length=200, n_mels=128, n_mfcc=20, pad_mode="reflect"),
    dict(n_fft=512, hop_length=512, n_mels=40, n_mfcc=13, center=False, power=1.0),
    dict(n_fft=256, hop_length=80, n_mels=40, n_mfcc=13, sr=8000),
    dict(n_fft=256, win_length=200, hop_length=64, n_mels=128, n_mfcc=20, sr=8000, pad_mode="reflect"),
    dict(n_fft=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=400, hop_length=100, n_mels=128, n_mfcc=20, sr=16000, pad_mode="reflect", window="hamming"),
    dict(n_fft=400, hop_length=400, n_mels=23, n_mfcc=13, sr=16000, center=False, power=1.0),
    dict(n_fft=480, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=480, hop_length=120, n_mels=128, n_mfcc=20, sr=16000, pad_mode="reflect", window="hamming"),
    dict(n_fft=480, hop_length=480, n_mels=23, n_mfcc=13, sr=16000, center=False, power=1.0),
    dict(n_fft=320, hop_length=160, n_mels=40, n_mfcc=13, sr=16000),
    dict(n_fft=320, hop_length=77, n_mels=80, n_mfcc=20, sr=16000, pad_mode="reflect"),
    dict(n_fft=640, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=640, hop_length=200, n_mels=128, n_mfcc=20, sr=22050, pad_mode="reflect", window="hamming"),
    dict(n_fft=640, hop_length=640, n_mels=23, n_mfcc=13, sr=16000, center=False, power=1.0),
    dict(n_fft=800, hop_length=200, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=800, hop_length=333, n_mels=128, n_mfcc=20, sr=32000, pad_mode="reflect", window="hamming"),
    dict(n_fft=960, hop_length=480, n_mels=80, n_mfcc=13, sr=48000),
    dict(n_fft=960, hop_length=960, n_mels=23, n_mfcc=13, sr=16000, center=False, power=1.0),
    dict(n_fft=128, hop_length=32, n_mels=20, n_mfcc=13, sr=8000),
    dict(n_fft=128, hop_length=128, n_mels=16, n_mfcc=13, sr=8000, center=False, power=1.0),
    dict(n_fft=160, hop_length=80, n_mels=23, n_mfcc=13, sr=8000, pad_mode="reflect"),
    dict(n_fft=192, hop_length=50, n_mels=40, n_mfcc=20, sr=16000, window="hamming"),
    dict(n_fft=384, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=384, hop_length=97, n_mels=128, n_mfcc=20, sr=16000, pad_mode="reflect", window="hamming"),
    dict(n_fft=576, hop_length=144, n_mels=80, n_mfcc=13, sr=22050),
    dict(n_fft=576, hop_length=576, n_mels=23, n_mfcc=13, sr=16000, center=False, power=1.0),
    dict(n_fft=768, hop_length=256, n_mels=128, n_mfcc=20, sr=48000, pad_mode="reflect"),
    dict(n_fft=768, hop_length=192, n_mels=80, n_mfcc=13, sr=16000, win_length=600),
])
def test_fft_and_dense_dft_kernels_agree(gpu, kw, monkeypatch):
    """The in-register FFT kernel and the dense-DFT kernel are independent implementations of the
    same transform; on a ragged batch (with edge frames, a partly filled last pass and utterances
    shorter than a frame) their MFCCs agree to the parity tolerance."""
    import torch

    import speech_to_text_amd as S

    cfg = S.make_config(**kw)
    lens = [1, 159, 160, 511, 512, 513, 4000, 16000, 33333, 7]
    if not kw.get("center", True):
        lens = [n for n in lens if n >= kw["n_fft"]]
    ys = [speech_like(n, seed=40 + i) for i, n in enumerate(lens)]
    so = np.zeros(len(ys) + 1, dtype=np.int64)
    np.cumsum([len(y) for y in ys], out=so[1:])
    pcm = torch.from_numpy(np.concatenate(ys)).cuda()
    outs = {}
    for dense in (False, True):
        if dense:
            monkeypatch.setenv("MFCC_HIP_DENSE_DFT", "1")
        plan = S.MfccPlan(cfg)
        assert plan.front_kernel == ("dense_dft" if dense else f"fft{kw['n_fft']}")
        fo = plan.frame_offsets(so)
        out = plan.execute(pcm, torch.from_numpy(so).cuda(), torch.from_numpy(fo).cuda(), int(fo[-1]))
        outs[dense] = out.cpu().numpy()
        plan.close()
    assert outs[False].shape == outs[True].shape and outs[False].shape[0] > 0
    assert np.abs(outs[False] - outs[True]).max() <= 1e-2          # dB, as for the librosa parity


def test_packed_sink_int16_ingest_and_device_dataset(gpu, tmp_path):
    """The packed single-file sink (fed through the int16 ingest) and the dataset that hands the
    trainer device-resident features give the same numbers as the per-utterance path."""
    import torch

    import speech_to_text_amd as S

    lens = ragged_lengths(10, 0.5, 4.0, seed=9)
    items = [(speech_like(int(L), seed=60 + i, int16_grid=True)[None, :], 16000, f"UTTERANCE {i}", 100 + i, 200 + i, i)
             for i, L in enumerate(lens)]
    pre = S.Preprocessor(items, out_dir=tmp_path / "p", hdf5_backend="min", max_batch_seconds=10.0, ingest="int16")
    assert pre.preprocess(verbose=False, packed=True) == 10
    rd = S.PackedReader(tmp_path / "p" / "packed_00000.hdf5")
    ref = S.mfcc_batch([it[0] for it in items], n_fft=2048, hop_length=512, n_mfcc=13)     # float32 path
    by_utt = {attrs["utterance_id"]: arr for arr, attrs in rd}
    for i, it in enumerate(items):
        assert np.array_equal(by_utt[i], ref[i])                       # int16 ingest is bit-identical
        np.testing.assert_allclose(by_utt[i], R.process_audio_features(it[0]), rtol=0, atol=1e-2)
    ds = S.FeatureDataset(items, device=gpu, max_batch_seconds=10.0)
    assert len(ds) == 10
    for i in range(10):
        mf, sr, text, spk = ds[i]
        assert isinstance(mf, torch.Tensor) and mf.is_cuda and tuple(mf.shape) == ref[i].shape
        assert np.array_equal(mf.cpu().numpy(), ref[i]) and (sr, text, spk) == (16000, f"UTTERANCE {i}", 100 + i)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512, n_mfcc=13),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
])
def test_execute_is_graph_capturable(gpu, kw):
    """mfcc_execute launches its kernels on the caller's stream and neither allocates nor
    synchronises, so a whole batch can be captured in a HIP graph and replayed on new audio."""
    import torch

    import speech_to_text_amd as S

    dev = torch.device("cuda", 0)
    plan = S.MfccPlan(S.make_config(**kw), dev)
    ys = [speech_like(int(n), seed=80 + i) for i, n in enumerate(ragged_lengths(12, 0.3, 3.0, seed=4))]
    so = np.zeros(len(ys) + 1, dtype=np.int64)
    np.cumsum([len(y) for y in ys], out=so[1:])
    fo = plan.frame_offsets(so)
    F = int(fo[-1])
    pcm = torch.from_numpy(np.concatenate(ys)).to(dev)
    d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
    out = torch.empty((F, kw["n_mfcc"]), dtype=torch.float32, device=dev)
    ws = plan.workspace(F, len(ys))
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):                                  # eager reference (and warm-up)
        plan.execute(pcm, d_so, d_fo, F, out=out, workspace=ws)
    torch.cuda.synchronize()
    eager = out.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.execute(pcm, d_so, d_fo, F, out=out, workspace=ws)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    pcm.mul_(0.25)                                                  # new audio in the captured buffer
    graph.replay()
    torch.cuda.synchronize()
    replayed = out.clone()
    plan.execute(pcm, d_so, d_fo, F, out=out, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(out, replayed) and not torch.equal(replayed, eager)
    plan.close()


@pytest.mark.gpu
def test_random_configurations_cross_check(gpu):
    """150 random configurations and ragged batches (tools/soak.py, fixed seed): every batch runs
    twice bit-identically, the in-register FFT kernels agree with the dense-DFT kernel and the
    2048-point kernel with the oracle."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import soak

    n, worst, where = soak.soak(seconds=60.0, seed=12345, max_batches=150)
    assert n == 150 and worst <= 1e-2, (n, worst, where)


def test_out_of_range_group_is_reported_not_dereferenced(gpu):
    """A d_group value outside [0, n_utts) must not index the per-utterance maxima: the kernels
    give that utterance a maximum of its own and mfcc_workspace_status returns MFCC_E_INVALID."""
    import torch

    import speech_to_text_amd as S
    from speech_to_text_amd import _lib

    lib = _lib.lib()
    plan = S.get_plan(S.MfccConfig.reference_default(), gpu)
    ys = [speech_like(n, seed=70 + i) for i, n in enumerate([20000, 31000, 9000])]
    so = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    fo = plan.frame_offsets(so)
    F = int(fo[-1])
    pcm = torch.from_numpy(np.concatenate(ys)).to(gpu)
    d_so, d_fo = torch.from_numpy(so).to(gpu), torch.from_numpy(fo).to(gpu)
    ws = plan.workspace(F, 3)
    stream = torch.cuda.current_stream().cuda_stream
    good = plan.execute(pcm, d_so, d_fo, F, workspace=ws).cpu().numpy()
    assert lib.mfcc_workspace_status(ws.data_ptr(), stream) == _lib.MFCC_OK
    bad = torch.tensor([0, 1 << 30, -5], dtype=torch.int32, device=gpu)
    out = plan.execute(pcm, d_so, d_fo, F, group=bad, workspace=ws).cpu().numpy()
    assert lib.mfcc_workspace_status(ws.data_ptr(), stream) == _lib.MFCC_E_INVALID
    assert "d_group" in _lib.last_error()
    assert np.array_equal(out, good)              # own maximum each == the default grouping
    with pytest.raises(S.ParameterError):         # the Python layer checks its host copy first
        from speech_to_text_amd.feature import _run_batch
        _run_batch(ys, S.MfccConfig.reference_default(), groups=np.array([0, 7, 1]), device=gpu)


def test_timing_history_and_per_stream_workspaces(gpu):
    """The plan's event ring returns one (front, dB+DCT) pair per execute, oldest first, without a
    host synchronisation between the executes; executes on two streams use two workspaces."""
    import torch

    import speech_to_text_amd as S

    plan = S.MfccPlan(S.MfccConfig.reference_default(), gpu)
    y = speech_like(200000, seed=3)
    so = np.array([0, len(y)], dtype=np.int64)
    fo = plan.frame_offsets(so)
    pcm = torch.from_numpy(y).to(gpu)
    d_so, d_fo = torch.from_numpy(so).to(gpu), torch.from_numpy(fo).to(gpu)
    plan.set_profiling(True)
    outs = [plan.execute(pcm, d_so, d_fo, int(fo[-1])) for _ in range(5)]
    front, dct = plan.timing_history(3)
    assert front.shape == dct.shape == (3,) and (front > 0).all() and (dct > 0).all()
    front, dct = plan.timing_history()
    assert len(front) == 5
    a, b = plan.last_timings()
    assert a == pytest.approx(float(front[-1])) and b == pytest.approx(float(dct[-1]))
    plan.set_profiling(True)
    assert len(plan.timing_history()[0]) == 0
    plan.set_profiling(False)
    ws_main = plan.workspace(int(fo[-1]), 1)
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ws_side = plan.workspace(int(fo[-1]), 1)
        other = plan.execute(pcm, d_so, d_fo, int(fo[-1]))
    assert ws_side.data_ptr() != ws_main.data_ptr()
    side.synchronize()
    torch.cuda.synchronize()
    assert all(torch.equal(o, other) for o in outs)
    plan.close()


def test_bench_under_torchrun_initialises_rccl(gpu):
    """The driver's launch form for a scaling run, `python -m torch.distributed.run
    --nproc-per-node N bench.py --gpus N`, at the N this box allows (1): the rank sets up its
    process group with the nccl backend (RCCL), runs the barriers and the max / sum all-reduces of
    the N > 1 path on the GPU, and prints one line."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_REHEARSAL")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(root / "bench.py"), "--gpus", "1",
                        "--steps", "5", "--warmup", "2", "--hours", "0.5", "--no-cpu-baseline",
                        "--strong-hours", "20", "--batch-gb", "0.5", "--sustain-seconds", "0.2"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert len(line["per_rank_ms"]) == 1
    # the legs a launcher adds: the fixed-corpus (strong-scaling) workload in several batches through
    # one workspace, the CMVN statistics through the process group, the sustained run
    st = line["strong"]
    assert st["scaling"] == "strong" and abs(st["audio_hours"] - 20.0) < 0.05 and st["per_rank_batches"][0] >= 9
    assert st["audio_hours_per_s"] > 0 and len(st["per_rank_seconds"]) == 1
    assert line["rccl"]["ranks_seen"] == 1 and line["rccl"]["cmvn_allreduce_ok"]
    assert line["sustained"]["steps"] >= 5 and line["sustained"]["audio_hours_per_s"] > 0


def test_overlapped_batches_match_single_batches(gpu):
    """pipeline.features_of_batches (a staging thread packing and uploading ahead, the download
    into pinned slots behind) gives, batch for batch, exactly what mfcc_batch gives for the same
    utterances -- float32 waveforms, float32 converted to 16-bit PCM while packing, and int16
    sources; the slots are recycled (more batches than slots), one batch spans several staging
    chunks, and the features of an utterance match the oracle."""
    import speech_to_text_amd as S
    from speech_to_text_amd import pipeline, staging

    lens = ragged_lengths(40, 0.3, 4.0, seed=9)
    ys = [speech_like(int(n), seed=300 + i, int16_grid=True) for i, n in enumerate(lens)]
    big = speech_like(staging.CHUNK_BYTES // 4 + 70001, seed=77, int16_grid=True)       # > one chunk of float32
    batches = [("a", ys[:7]), ("b", [big] + ys[7:9]), ("c", ys[9:10]), ("d", ys[10:25]), ("e", ys[25:26]),
               ("f", ys[26:]), ("g", [ys[3][:100], ys[4][:1]])]
    want = {tag: S.mfcc_batch(b, n_fft=2048, hop_length=512, n_mfcc=13) for tag, b in batches}
    for kind in ("f32", "f32_to_i16", "i16"):
        src = [(tag, [np.rint(y * 32768.0).astype(np.int16) for y in b] if kind == "i16" else b) for tag, b in batches]
        with pytest.warns(UserWarning, match="too large for input signal"):
            got = pipeline.mfcc_of_batches(iter(src), n_fft=2048, hop_length=512, n_mfcc=13, device=gpu,
                                           try_int16=kind == "f32_to_i16")
        assert [t for t, _ in got] == [t for t, _ in batches]
        for tag, feats in got:
            assert len(feats) == len(want[tag])
            for a, b in zip(feats, want[tag]):
                assert a.shape == b.shape and np.array_equal(a, b), (kind, tag)
    np.testing.assert_allclose(want["a"][2], R.process_audio_features(ys[2]), rtol=0, atol=1e-2)


def test_overlapped_batches_report_bad_audio_and_stager_errors(gpu):
    """A non-finite sample raises ParameterError (librosa's valid_audio) from the batch that holds
    it; an exception in the iterable that feeds the staging thread reaches the consumer."""
    import speech_to_text_amd as S
    from speech_to_text_amd import pipeline

    ys = [speech_like(9000 + 100 * i, seed=i) for i in range(6)]
    bad = ys[4].copy()
    bad[17] = np.inf
    cfg = S.make_config(n_fft=2048, hop_length=512, n_mfcc=13)
    seen = []
    with pytest.raises(S.ParameterError, match="not finite"):
        for tag, host, fo, release in pipeline.features_of_batches([(0, ys[:2]), (1, ys[2:4]), (2, [bad, ys[5]])], cfg, gpu):
            seen.append(tag)
            release()
    assert seen == [0, 1]

    def feed():
        yield 0, ys[:3]
        raise KeyError("decode failed")

    with pytest.raises(KeyError, match="decode failed"):
        for tag, host, fo, release in pipeline.features_of_batches(feed(), cfg, gpu):
            release()
    with pytest.raises(S.ParameterError, match="floating-point"):
        for _ in pipeline.features_of_batches([(0, [np.arange(5000, dtype=np.int32)])], cfg, gpu):
            pass


def test_preprocess_overlapped_files_match_oracle_and_stats(gpu, tmp_path):
    """Preprocessor.preprocess on the GPU path (several small batches so that staging, compute and
    the native multi-threaded file writer overlap): every file holds the oracle's features and the
    reference's attributes, no partial directory is left, and the stage times are reported."""
    import speech_to_text_amd as S

    lens = ragged_lengths(23, 0.4, 2.5, seed=21)
    items = [(speech_like(int(n), seed=500 + i, int16_grid=True)[None, :], 16000, f"SENTENCE {i} é", 10 + i, 20 + i % 3, i)
             for i, n in enumerate(lens)]
    for ingest in ("float32", "int16"):
        out = tmp_path / f"mfcc_{ingest}"
        pre = S.Preprocessor(items, str(out), device=gpu, max_batch_seconds=6.0, ingest=ingest, hdf5_backend="native")
        assert pre.preprocess(verbose=False) == len(items)
        assert pre.stats["batches"] >= 5 and pre.stats["stage_seconds"]["wall"] > 0
        assert not (tmp_path / f"mfcc_{ingest}.partial").exists()
        assert len(list(out.iterdir())) == len(items)
        for i, it in enumerate(items):
            arr, attrs = S.read_feature_file(out / f"{it[3]}_{it[4]}_{it[5]}.hdf5")
            assert attrs == dict(speaker_id=it[3], chapter_id=it[4], utterance_id=it[5], sample_rate=16000, transcript=it[2])
            assert list(attrs) == ["speaker_id", "chapter_id", "utterance_id", "sample_rate", "transcript"]
            if i % 5 == 0:
                np.testing.assert_allclose(arr, R.process_audio_features(it[0][0]), rtol=0, atol=1e-2)


def test_preprocess_ingest_auto_follows_the_data(gpu, tmp_path):
    """The default ingest: waveforms on the int16 grid (what a 16-bit FLAC decodes to) go up as 16-bit
    PCM, batch after batch; waveforms off the grid are found out on the first batch and the
    conversion is not tried again.  Either way the files hold the float32 path's features bit for bit."""
    import speech_to_text_amd as S

    lens = ragged_lengths(17, 0.4, 2.0, seed=31)
    for grid in (True, False):
        items = [(speech_like(int(n), seed=700 + i, int16_grid=grid)[None, :], 16000, f"T{i}", 1, 2, i) for i, n in enumerate(lens)]
        outs = {}
        for ingest in ("auto", "float32"):
            out = tmp_path / f"{ingest}_{grid}"
            pre = S.Preprocessor(items, str(out), device=gpu, max_batch_seconds=5.0, hdf5_backend="native",
                                 **({} if ingest == "auto" else {"ingest": ingest}))
            assert pre.ingest == ingest and pre.preprocess(verbose=False) == len(items)
            up = pre.stats["uploaded_as"]
            assert up["i16"] + up["f32"] == pre.stats["batches"] >= 4
            if ingest == "auto":
                assert (up["f32"] == 0) if grid else (up["i16"] == 0)
            else:
                assert up["i16"] == 0
            outs[ingest] = out
        for it in items:
            name = f"{it[3]}_{it[4]}_{it[5]}.hdf5"
            a, _ = S.read_feature_file(outs["auto"] / name)
            b, _ = S.read_feature_file(outs["float32"] / name)
            assert np.array_equal(a, b)
