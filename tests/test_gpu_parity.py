"""GPU parity: libmfcc_hip (through the C ABI via ctypes) against the CPU oracle on the same
seeded inputs.

Tolerances are the north-star's: <= 1e-4 relative on mel power and <= 1e-2 dB absolute on
log-mel / MFCC.  An fp32 transform's error is relative to the FRAME's energy, not to each
band, so the mel-power check carries the absolute floor that implies (MEL_FLOOR x the largest
mel power of the utterance); bands that far below the peak are clamped by top_db anyway.
"""
import numpy as np
import pytest

from oracle import mfcc_ref as R
from tests.signals import SR, speech_like

pytestmark = pytest.mark.gpu

DB_TOL = 1e-2          # dB, absolute, on log-mel / MFCC
MEL_RTOL = 1e-4        # relative, on mel power
MEL_FLOOR = 2e-7       # x max mel power of the utterance
MEL_FLOOR_MAX_FRACTION = 1e-3     # of an utterance's mel values may exceed the relative bound alone ...
MEL_FLOOR_MIN_DEPTH_DB = 60.0     # ... and only bands at least this far below its largest band
FLOOR_USE = []                    # (elements that needed the floor, elements, highest level in dB re peak) per call


def check_mel(got, want, amin=1e-10, top_db=80.0):
    """Mel power against the oracle, two criteria, both asserted:
    (1) <= 1e-4 relative, with the absolute floor an fp32 transform needs (its error scales with
        the frame's energy, not with the band's);
    (2) the log-mel criterion over the whole range top_db keeps: wherever the oracle's band lies
        within top_db of the utterance's largest band, 10 log10 of the two values differ by
        <= 1e-2 dB -- so the floor of (1) never hides an error that would survive into the
        features.
    The floor's use is bounded too: at most MEL_FLOOR_MAX_FRACTION of the elements may need it,
    and none that sits less than MEL_FLOOR_MIN_DEPTH_DB below the utterance's largest band."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    peak = np.abs(want).max()
    err = np.abs(got - want)
    tol = MEL_RTOL * np.abs(want) + MEL_FLOOR * peak
    assert (err <= tol).all(), f"mel power: max err/tol = {(err / np.maximum(tol, 1e-300)).max():.3g}"
    need_floor = err > MEL_RTOL * np.abs(want)
    if need_floor.any() and peak > 0:
        level = 10.0 * np.log10(np.maximum(want[need_floor], 1e-300) / peak)
        FLOOR_USE.append((int(need_floor.sum()), int(want.size), float(level.max())))
        assert need_floor.mean() <= MEL_FLOOR_MAX_FRACTION, \
            f"{need_floor.mean():.2%} of the mel values need the absolute floor"
        assert level.max() <= -MEL_FLOOR_MIN_DEPTH_DB, \
            f"a band only {-level.max():.1f} dB below the peak needs the absolute floor"
    db_got = 10.0 * np.log10(np.maximum(amin, got))
    db_want = 10.0 * np.log10(np.maximum(amin, want))
    kept = db_want >= db_want.max() - top_db
    worst = np.abs(db_got - db_want)[kept].max()
    assert worst <= DB_TOL, f"log-mel within top_db of the peak differs by {worst:.3g} dB"


@pytest.fixture(scope="module")
def S(gpu):
    import speech_to_text_amd as pkg

    return pkg


def test_reference_default_one_utterance(S):
    """BASELINE config 1: the call of /root/reference/src/preprocess.py:134 on one utterance."""
    y = speech_like(166960, seed=2)
    want = R.process_audio_features(y)                      # (327, 13)
    got = S.mfcc(y=y[None, :], n_fft=2048, hop_length=512, n_mfcc=13)
    assert got.shape == (1, 13, 327) and got.dtype == np.float32
    np.testing.assert_allclose(got[0].T, want, rtol=0, atol=DB_TOL)


@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512),
    dict(n_fft=2048, hop_length=160, n_mels=80),
    dict(n_fft=2048, hop_length=512, pad_mode="reflect"),
    dict(n_fft=2048, hop_length=512, center=False),
    dict(n_fft=2048, hop_length=300, win_length=1200),
    dict(n_fft=2048, hop_length=512, htk=True, fmin=50.0, fmax=7000.0, sr=16000),
    dict(n_fft=2048, hop_length=512, window="hamming"),
    dict(n_fft=2048, hop_length=512, power=1.0),
    dict(n_fft=2048, hop_length=512, mel_norm=None, n_mels=40),
    dict(n_fft=2048, hop_length=512, n_mels=10),                 # wide filters: shared mel cells
    dict(n_fft=2048, hop_length=256, n_mels=256, sr=44100),      # four groups of 64 filters
    dict(n_fft=2048, hop_length=512, n_mels=1),
    dict(n_fft=400, hop_length=160, n_mels=80, sr=16000),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, sr=16000),
    dict(n_fft=1024, hop_length=256),
    dict(n_fft=512, hop_length=128, pad_mode="reflect", n_mels=64),
    dict(n_fft=400, hop_length=160, n_mels=80, sr=16000, center=False),
    dict(n_fft=254, hop_length=77, n_mels=20, sr=8000),
    dict(n_fft=320, hop_length=160, n_mels=40, sr=16000),          # 20 x 16 in-register kernel
    dict(n_fft=480, hop_length=160, n_mels=80, sr=16000),          # 30 x 16 in-register kernel
    dict(n_fft=480, hop_length=240, n_mels=128, sr=22050, pad_mode="reflect"),
    dict(n_fft=640, hop_length=160, n_mels=40, sr=16000),          # 40 x 16 in-register kernel
    dict(n_fft=640, hop_length=320, n_mels=128, sr=22050, pad_mode="reflect"),
    dict(n_fft=800, hop_length=200, n_mels=80, sr=16000),          # 50 x 16
    dict(n_fft=960, hop_length=480, n_mels=64, sr=48000, pad_mode="reflect"),   # 60 x 16
    dict(n_fft=600, hop_length=150, n_mels=40, sr=16000),          # dense-DFT kernel
    dict(n_fft=128, hop_length=32, n_mels=20, sr=8000),            # 8 x 16
    dict(n_fft=160, hop_length=80, n_mels=23, sr=8000, pad_mode="reflect"),     # 10 x 16
    dict(n_fft=192, hop_length=48, n_mels=32, sr=16000),           # 12 x 16
    dict(n_fft=384, hop_length=160, n_mels=80, sr=16000),          # 24 x 16
    dict(n_fft=576, hop_length=144, n_mels=64, sr=22050, pad_mode="reflect"),   # 36 x 16
    dict(n_fft=768, hop_length=256, n_mels=128, sr=48000),         # 48 x 16
    dict(n_fft=512, hop_length=160, n_mels=10, sr=16000),          # wide filters on the 512-point kernel
    dict(n_fft=1024, hop_length=256, n_mels=256, sr=44100),        # 16 mel rounds of 32 filters
    dict(n_fft=400, hop_length=160, n_mels=1, sr=16000),
    dict(n_fft=256, hop_length=128, n_mels=64, sr=8000, htk=True),
])
def test_melspectrogram_parity(S, kw):
    y = speech_like(41234, seed=3)
    okw = dict(kw)
    if "mel_norm" in okw:
        okw["norm"] = okw.pop("mel_norm")
    want = R.melspectrogram(y=y, **okw)
    got = S.melspectrogram(y=y, **kw)
    assert got.shape == want.shape and got.dtype == np.float32
    check_mel(got, want)


@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512, n_mfcc=13),
    dict(n_fft=2048, hop_length=512, n_mfcc=20),
    dict(n_fft=2048, hop_length=512, n_mfcc=40, lifter=22),
    dict(n_fft=2048, hop_length=512, n_mfcc=13, dct_type=3),
    dict(n_fft=2048, hop_length=512, n_mfcc=13, dct_type=1),
    dict(n_fft=2048, hop_length=512, n_mfcc=13, norm=None),
    dict(n_fft=2048, hop_length=512, n_mfcc=13, pad_mode="reflect"),
    dict(n_fft=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=1024, hop_length=256, n_mfcc=13, n_mels=80),
    dict(n_fft=256, hop_length=80, n_mfcc=13, n_mels=40, sr=8000),
    dict(n_fft=480, hop_length=160, n_mfcc=13, n_mels=80, sr=16000),
    dict(n_fft=320, hop_length=160, n_mfcc=13, n_mels=40, sr=16000),
    dict(n_fft=384, hop_length=160, n_mfcc=13, n_mels=80, sr=16000),
    dict(n_fft=768, hop_length=384, n_mfcc=20, n_mels=128, sr=48000),
    dict(n_fft=160, hop_length=80, n_mfcc=13, n_mels=23, sr=8000),
])
def test_mfcc_parity(S, kw):
    y = speech_like(52001, seed=5)
    okw = dict(kw)
    want = R.mfcc(y=y, **okw)
    got = S.mfcc(y=y, **kw)
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=0, atol=DB_TOL)


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (400, 160), (512, 128), (1024, 256), (256, 64), (480, 160), (320, 80), (640, 160), (800, 200), (960, 240),
                                       (128, 32), (160, 40), (192, 64), (384, 96), (576, 192), (768, 256)])
def test_spectrogram_parity(S, n_fft, hop):
    y = speech_like(30000, seed=4)
    want = np.abs(R.stft(y, n_fft=n_fft, hop_length=hop)) ** 2
    got = S.spectrogram(y=y, n_fft=n_fft, hop_length=hop, power=2.0)
    assert got.shape == want.shape
    tol = 1e-4 * want + 2e-6 * want.max(axis=0, keepdims=True)
    assert (np.abs(got - want) <= tol).all()
    mag = S.spectrogram(y=y, n_fft=n_fft, hop_length=hop)          # power=1, as preprocess.py:113
    np.testing.assert_allclose(mag, np.sqrt(want), rtol=2e-4, atol=2e-4 * np.sqrt(want).max())


@pytest.mark.parametrize("T", [1, 2, 511, 512, 513, 1023, 1024, 2047, 2048, 2049, 4096, 5121])
@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_short_and_boundary_lengths(S, T, pad_mode):
    """Frame count F = 1 + T // hop and the padded head/tail frames, incl. T < n_fft (librosa
    warns and still computes)."""
    y = speech_like(T, seed=100 + T)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = R.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13, pad_mode=pad_mode)
        got = S.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13, pad_mode=pad_mode)
    assert got.shape == want.shape == (13, 1 + T // 512)
    np.testing.assert_allclose(got, want, rtol=0, atol=DB_TOL)


def test_ragged_batch_matches_per_utterance(S):
    """The batch path gives each utterance its own top_db maximum and is bit-identical to
    running the utterances one at a time."""
    lens = [32000, 5, 70001, 2048, 166960, 999, 40000, 12345]
    ys = [speech_like(n, seed=10 + i) * (10.0 ** (-i / 4.0)) for i, n in enumerate(lens)]
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        batch = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=13)
        for y, f in zip(ys, batch):
            want = R.process_audio_features(y)
            assert f.shape == want.shape and f.flags["C_CONTIGUOUS"] and f.dtype == np.float32
            np.testing.assert_allclose(f, want, rtol=0, atol=DB_TOL)
            single = S.mfcc_batch([y], n_fft=2048, hop_length=512, n_mfcc=13)[0]
            assert np.array_equal(single, f)


def test_multichannel_shares_top_db_max(S):
    """librosa takes log_spec.max() over the whole array: a quiet channel next to a loud one is
    clamped against the loud channel's maximum."""
    loud = speech_like(30000, seed=1)
    quiet = speech_like(30000, seed=2) * 1e-4
    y = np.stack([loud, quiet])
    want = R.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13)
    got = S.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13)
    assert got.shape == want.shape == (2, 13, 59)
    np.testing.assert_allclose(got, want, rtol=0, atol=DB_TOL)
    alone = S.mfcc(y=quiet, n_fft=2048, hop_length=512, n_mfcc=13)
    assert np.abs(alone - got[1]).max() > 1.0   # the shared maximum really changed the result


@pytest.mark.parametrize("name", ["zeros", "square", "noise_m60", "dc", "chirp", "click_in_silence"])
def test_value_distributions(S, name):
    n = 48000
    t = np.arange(n) / SR
    rng = np.random.default_rng(7)
    y = {
        "zeros": np.zeros(n),
        "square": np.sign(np.sin(2 * np.pi * 100 * t)) * 0.999,
        "noise_m60": 1e-3 * rng.standard_normal(n),
        "dc": np.full(n, 0.25) + 0.01 * rng.standard_normal(n),
        "chirp": 0.5 * np.sin(2 * np.pi * (50 + 3900 * t / t[-1]) * t),
        "click_in_silence": np.where(np.arange(n) == 20000, 0.9, 0.0) + 1e-5 * rng.standard_normal(n),
    }[name].astype(np.float32)
    want = R.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13)
    got = S.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13)
    np.testing.assert_allclose(got, want, rtol=0, atol=DB_TOL)
    if name == "zeros":
        assert np.allclose(got[0], -100.0 * np.sqrt(128.0), atol=1e-3) and np.allclose(got[1:], 0, atol=1e-3)


def test_top_db_clamp_property(S):
    """log-mel min >= max - 80 for every utterance (checked through the mel tap + oracle's dB)."""
    y = speech_like(60000, seed=9)
    mel = S.melspectrogram(y=y, n_fft=2048, hop_length=512)
    db = R.power_to_db(mel)
    assert db.min() >= db.max() - 80.0 - 1e-4


def test_nonfinite_audio_is_rejected(S):
    y = speech_like(5000, seed=1)
    y[100] = np.nan
    with pytest.raises(S.ParameterError):
        S.mfcc(y=y, n_fft=2048, hop_length=512, n_mfcc=13)
    y[100] = np.inf
    with pytest.raises(S.ParameterError):
        S.mfcc_batch([y], n_fft=2048, hop_length=512, n_mfcc=13)
    with pytest.raises(S.ParameterError):
        S.mfcc(y=(y * 100).astype(np.int16), n_fft=2048)


def test_invalid_parameters(S):
    y = speech_like(5000, seed=1)
    for kw in [dict(lifter=-1), dict(hop_length=0), dict(pad_mode="wrap"), dict(n_mfcc=200),
               dict(dct_type=4), dict(center=False)]:
        with pytest.raises((S.ParameterError, ValueError)):
            S.mfcc(y=y[:1000] if "center" in kw else y, n_fft=2048, **kw)


def test_deterministic(S):
    ys = [speech_like(n, seed=i) for i, n in enumerate([50000, 30000, 80000])]
    a = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=13)
    b = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=13)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_full_size_batch_properties(S):
    """BASELINE config 2 scale (a 2-hour slice of it, sized for the test budget): frame counts,
    per-utterance independence (spot utterances equal the oracle), scaling property
    mfcc(c * y)[0] = mfcc(y)[0] + 20 log10(c) sqrt(n_mels) away from the top_db floor."""
    import torch

    from bench import make_batch

    dev = torch.device("cuda", 0)
    cfg = S.MfccConfig.reference_default()
    plan = S.get_plan(cfg, dev)
    pcm, so = make_batch(2.0, seed=99, device=dev)
    fo = plan.frame_offsets(so)
    assert np.array_equal(np.diff(fo), 1 + np.diff(so) // 512)
    d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
    out = plan.execute(pcm, d_so, d_fo, int(fo[-1])).cpu().numpy()
    assert np.isfinite(out).all()
    for u in (0, len(so) // 2, len(so) - 2):
        y = pcm[int(so[u]):int(so[u + 1])].cpu().numpy()
        np.testing.assert_allclose(out[int(fo[u]):int(fo[u + 1])], R.process_audio_features(y), rtol=0, atol=DB_TOL)
    out2 = plan.execute(pcm * 0.5, d_so, d_fo, int(fo[-1])).cpu().numpy()
    shift = 20 * np.log10(0.5) * np.sqrt(128.0)
    # coefficient 0 shifts by 20 log10(c) sqrt(n_mels) where no band sits on the top_db floor;
    # the floor moves with the maximum, so the shift holds for clamped bands too.
    np.testing.assert_allclose(out2[:, 0], out[:, 0] + shift, rtol=0, atol=5e-3)
    np.testing.assert_allclose(out2[:, 1:], out[:, 1:], rtol=0, atol=5e-3)


@pytest.mark.parametrize("kw", [
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
    dict(n_fft=400, hop_length=160, n_mels=80, n_mfcc=13, sr=16000),
])
def test_full_size_batch_properties_asr_front_ends(S, kw):
    """The 16 kHz front ends (BASELINE config 4) at a 2-hour slice of the bench batch: frame
    counts, spot utterances against the oracle, and the scaling property of the other test."""
    import torch

    from bench import make_batch

    dev = torch.device("cuda", 0)
    cfg = S.make_config(**kw)
    plan = S.get_plan(cfg, dev)
    assert plan.front_kernel == f"fft{kw['n_fft']}"
    pcm, so = make_batch(2.0, seed=98, device=dev)
    fo = plan.frame_offsets(so)
    assert np.array_equal(np.diff(fo), 1 + np.diff(so) // kw["hop_length"])
    d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
    out = plan.execute(pcm, d_so, d_fo, int(fo[-1])).cpu().numpy()
    assert np.isfinite(out).all()
    for u in (0, len(so) // 3, len(so) - 2):
        y = pcm[int(so[u]):int(so[u + 1])].cpu().numpy()
        np.testing.assert_allclose(out[int(fo[u]):int(fo[u + 1])], R.mfcc(y=y, **kw).T, rtol=0, atol=DB_TOL)
    out2 = plan.execute(pcm * 0.5, d_so, d_fo, int(fo[-1])).cpu().numpy()
    shift = 20 * np.log10(0.5) * np.sqrt(80.0)
    np.testing.assert_allclose(out2[:, 0], out[:, 0] + shift, rtol=0, atol=5e-3)
    np.testing.assert_allclose(out2[:, 1:], out[:, 1:], rtol=0, atol=5e-3)


@pytest.mark.parametrize("ref,top_db", [(1.0, 80.0), (np.max, 80.0), (0.5, None), (np.max, 30.0)])
def test_amplitude_and_power_to_db(S, ref, top_db):
    """The spectrogram display path of /root/reference/src/preprocess.py:111-114:
    amplitude_to_db(np.abs(stft)) and power_to_db against the oracle's restatement."""
    y = speech_like(30000, seed=11)
    mag = S.spectrogram(y=y, n_fft=2048, hop_length=512)               # power=1, as :113
    want = R.amplitude_to_db(R.spectrogram(y=y, n_fft=2048, hop_length=512, power=1.0) if hasattr(R, "spectrogram")
                             else np.abs(R.stft(y, n_fft=2048, hop_length=512)), ref=ref, top_db=top_db)
    got = S.amplitude_to_db(mag, ref=ref, top_db=top_db)
    assert got.shape == want.shape and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=0, atol=DB_TOL)
    powr = mag.astype(np.float64) ** 2
    np.testing.assert_allclose(S.power_to_db(powr.astype(np.float32), ref=ref, top_db=top_db),
                               R.power_to_db(powr, ref=ref, top_db=top_db), rtol=0, atol=DB_TOL)
    import torch

    dev_in = torch.from_numpy(mag).cuda()
    dev_out = S.amplitude_to_db(dev_in, ref=ref, top_db=top_db)
    assert dev_out.is_cuda and np.array_equal(dev_out.cpu().numpy(), got)
    with pytest.raises(S.ParameterError):
        S.power_to_db(mag, amin=0.0)


@pytest.mark.parametrize("width,order", [(9, 1), (9, 2), (5, 1), (21, 3)])
def test_delta_features(S, width, order):
    """The delta extension against librosa.feature.delta's definition (scipy's savgol_filter in
    the oracle), for one array and for a ragged batch whose utterances must not leak into each
    other; an utterance shorter than the window is an error, as in librosa."""
    import torch

    ys = [speech_like(n, seed=60 + i) for i, n in enumerate([40000, 12000, 25000])]
    feats = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=13)
    want = [R.delta(f.T.astype(np.float64), width=width, order=order).T for f in feats]
    tol = 2e-4 * max(1.0, float(np.abs(np.concatenate(want)).max()))
    got0 = S.delta(feats[0].T, width=width, order=order)                 # librosa layout (n_mfcc, F)
    assert got0.shape == feats[0].T.shape and got0.dtype == np.float32
    np.testing.assert_allclose(got0.T, want[0], rtol=0, atol=tol)
    fo = np.zeros(len(feats) + 1, dtype=np.int64)
    np.cumsum([f.shape[0] for f in feats], out=fo[1:])
    dev = torch.from_numpy(np.concatenate(feats)).cuda()
    out = S.delta_batch(dev, fo, width=width, order=order).cpu().numpy()
    for u, w in enumerate(want):
        np.testing.assert_allclose(out[int(fo[u]):int(fo[u + 1])], w, rtol=0, atol=tol)
    with pytest.raises(S.ParameterError):
        S.delta(feats[0].T[:, :width - 1], width=width, order=order)
    with pytest.raises(S.ParameterError):
        S.delta(feats[0].T, width=4)


# ---- log-mel (power_to_db) itself, not only its first DCT coefficients ---------------------------
def _idct_ortho(coefs, n_mels):
    """Rows of an orthonormal DCT-II are an orthonormal basis: with all n_mels coefficients the
    log-mel spectrogram is basis.T @ coefficients."""
    k = np.arange(n_mels)[:, None]
    n = np.arange(n_mels)[None, :]
    basis = np.sqrt(2.0 / n_mels) * np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_mels))
    basis[0] *= np.sqrt(0.5)
    return np.moveaxis(np.tensordot(basis.T, coefs.astype(np.float64), axes=([1], [-2])), 0, -2)


@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512, n_mels=128),                              # the reference's shape
    dict(n_fft=2048, hop_length=512, n_mels=80),
    dict(n_fft=2048, hop_length=512, n_mels=40, pad_mode="reflect"),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, sr=16000),      # BASELINE config 4
    dict(n_fft=400, hop_length=160, n_mels=80, sr=16000),                      # BASELINE config 4, n_fft 400
    dict(n_fft=1024, hop_length=256, n_mels=128),
])
@pytest.mark.parametrize("signal", ["speech", "quiet_tail", "two_channels"])
def test_log_mel_parity(S, kw, signal):
    """power_to_db with the per-call maximum and the top_db clamp, checked element by element:
    with n_mfcc = n_mels the orthonormal DCT loses nothing, so the log-mel spectrogram is
    recovered from the coefficients and compared with the oracle's
    power_to_db(melspectrogram(y)) at 1e-2 dB (librosa.feature.mfcc at
    /root/reference/src/preprocess.py:134 computes exactly that before its DCT)."""
    n_mels = kw["n_mels"]
    if signal == "speech":
        y = speech_like(47001, seed=21)
    elif signal == "quiet_tail":          # the second half sits on or near the top_db floor
        y = speech_like(47001, seed=22)
        y[24000:] *= np.float32(3e-5)
    else:                                 # channels of one call share the maximum
        y = np.stack([speech_like(30000, seed=23), speech_like(30000, seed=24) * np.float32(2e-4)])
    want = R.power_to_db(R.melspectrogram(y=y, **kw))
    got = _idct_ortho(S.mfcc(y=y, n_mfcc=n_mels, **kw), n_mels)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= DB_TOL, np.abs(got - want).max()
    # the clamp is really exercised by the two stress signals
    if signal != "speech":
        assert (want <= want.max() - 80.0 + 1e-3).mean() > 0.05


@pytest.mark.parametrize("kw", [
    dict(n_fft=2048, hop_length=512, n_mels=128),
    dict(n_fft=512, win_length=400, hop_length=160, n_mels=80, sr=16000),
    dict(n_fft=400, hop_length=160, n_mels=40, sr=16000),
])
def test_small_db_dct_kernel_matches_general_kernel(S, kw):
    """<= 16 coefficients run the register-resident dB+DCT kernel, more run the general one
    (whose log-mel test_log_mel_parity checks element by element).  Both start from the same mel
    power, so their common coefficients may differ only by float32 rounding: 1e-3 dB here.  The
    first basis row weighs every band by 1 / sqrt(n_mels), so an error in any single log-mel
    element of the small kernel above ~1e-2 dB would show."""
    y = speech_like(47001, seed=25)
    y[30000:] *= np.float32(1e-4)
    for dct_type in (2, 3):
        small = S.mfcc(y=y, n_mfcc=16, dct_type=dct_type, **kw)
        full = S.mfcc(y=y, n_mfcc=kw["n_mels"], dct_type=dct_type, **kw)
        np.testing.assert_allclose(small, full[:16], rtol=0, atol=1e-3)


def test_config5_long_utterances(S):
    """BASELINE config 5: utterances of 30-60 s (938-1876 frames each, many tiles per utterance,
    one top_db maximum across all of them), n_fft 2048, hop 512, 128 mels, top_db 80 -- plus one
    45 s utterance that is noise at the 2e-6 level (about -99 dB per mel band) with a single
    full-scale click (-11.5 dB), so that almost every band of it sits on the top_db floor set by
    one frame."""
    rng = np.random.default_rng(5)
    lens = (rng.uniform(30.0, 60.0, size=6) * SR).astype(np.int64)
    ys = [speech_like(int(n), seed=500 + i) * np.float32(10.0 ** (-i / 3.0)) for i, n in enumerate(lens)]
    click = (2e-6 * rng.standard_normal(45 * SR)).astype(np.float32)
    click[17 * SR + 123] = 0.999
    ys.insert(3, click)
    feats = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=13)
    mels = S.mfcc_batch(ys, n_fft=2048, hop_length=512, n_mfcc=128)
    for y, f, m in zip(ys, feats, mels):
        want = R.process_audio_features(y)
        assert f.shape == want.shape == (1 + len(y) // 512, 13)
        np.testing.assert_allclose(f, want, rtol=0, atol=DB_TOL)
        # and the log-mel itself, through the full DCT
        logmel = _idct_ortho(m.T, 128)
        want_db = R.power_to_db(R.melspectrogram(y=y, n_fft=2048, hop_length=512))
        assert np.abs(logmel - want_db).max() <= DB_TOL
    want_db = R.power_to_db(R.melspectrogram(y=click, n_fft=2048, hop_length=512))
    assert (want_db <= want_db.max() - 80.0 + 1e-3).mean() > 0.9     # the click sets the floor


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 57), (2, 40, 33), (80, 1), (3, 2, 23, 16)])
@pytest.mark.parametrize("dct_type,norm,lifter", [(2, "ortho", 0), (2, None, 22), (1, None, 0), (3, "ortho", 5)])
def test_mfcc_of_a_given_log_mel_spectrogram(S, shape, dct_type, norm, lifter):
    """librosa.feature.mfcc(S=...): the DCT and lifter of a dB-scaled mel spectrogram the caller
    already has (the spelling of /root/reference/src/preprocess.py:134's function that takes S)."""
    rng = np.random.default_rng(11)
    logmel = (rng.standard_normal(shape) * 20.0 - 30.0).astype(np.float32)
    n_mfcc = min(13, shape[-2])
    got = S.mfcc(S=logmel, n_mfcc=n_mfcc, dct_type=dct_type, norm=norm, lifter=lifter)
    want = R.mfcc(S=logmel.astype(np.float64), n_mfcc=n_mfcc, dct_type=dct_type, norm=norm, lifter=lifter)
    assert got.shape == want.shape == shape[:-2] + (n_mfcc, shape[-1]) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * np.abs(want).max())
    # the same numbers as the y= form gives on its own log-mel spectrogram
    if dct_type == 2 and norm == "ortho" and lifter == 0 and tuple(shape) == (128, 57):
        y = speech_like(57 * 512 - 1, seed=5)
        full = S.mfcc(y=y, n_mfcc=13)
        via = S.mfcc(S=R.power_to_db(R.melspectrogram(y=y)), n_mfcc=13)
        np.testing.assert_allclose(via, full, atol=DB_TOL)
    # more coefficients asked for than there are bands: one row per band, as scipy's slice gives
    if dct_type == 2 and lifter == 0:
        many = S.mfcc(S=logmel, n_mfcc=shape[-2] + 7, dct_type=dct_type, norm=norm)
        want_many = R.mfcc(S=logmel.astype(np.float64), n_mfcc=shape[-2] + 7, dct_type=dct_type, norm=norm)
        assert many.shape == want_many.shape == shape
        np.testing.assert_allclose(many, want_many, rtol=2e-5, atol=2e-5 * np.abs(want_many).max())


# What fp32 transforms can promise on utterances shorter than one frame under reflect padding: a
# few samples repeated many times give a line spectrum, and the bands between the lines hold only
# window side lobes 50-80 dB below the peak -- where an fp32 transform's error, which is relative
# to the frame's energy, is a visible fraction of the band.  The bounds below are asserted, not
# assumed: log-mel inside top_db of the utterance's peak, and the features themselves.
SHORT_REFLECT_LOGMEL_TOL = 1e-2      # dB, bands within top_db (80 dB) of the peak
SHORT_REFLECT_LOGMEL_55_TOL = 1e-3   # dB, bands within 55 dB of the peak
SHORT_REFLECT_MFCC_TOL = 1e-2        # dB, the features (orthonormal DCT, no lifter)


def test_short_utterances_reflect_padding(S):
    """Utterances shorter than n_fft with pad_mode='reflect' (librosa only warns), over every
    front-end kernel and the rarely used parameters, against the oracle: 240 seeded cases."""
    import warnings

    rng = np.random.default_rng(20241112)
    worst_mel = worst_mel55 = worst_mfcc = 0.0
    where = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for case in range(240):
            n_fft = int(rng.choice([2048, 2048, 2048, 1024, 960, 800, 640, 600, 512, 480, 400, 320, 256]))
            kw = dict(n_fft=n_fft, hop_length=int(rng.integers(n_fft // 8, n_fft + 1)),
                      n_mels=int(rng.choice([20, 40, 64, 80, 128])), sr=int(rng.choice([8000, 16000, 22050])),
                      pad_mode="reflect")
            if rng.random() < 0.3:
                kw["win_length"] = int(rng.integers(n_fft // 2, n_fft + 1))
            if rng.random() < 0.2:
                kw["power"] = float(rng.choice([1.0, 3.0]))
            if rng.random() < 0.2:
                kw["htk"] = True
            if rng.random() < 0.15:
                kw["window"] = str(rng.choice(["hamming", "blackman"]))
            L = int(rng.integers(2, n_fft))
            kind = int(rng.integers(0, 3))
            scale = float(rng.choice([1e-3, 0.1, 1.0]))
            if kind == 0:
                y = (rng.standard_normal(L) * scale).astype(np.float32)
            elif kind == 1:
                y = (scale * np.sin(2 * np.pi * rng.uniform(50, 3000) * np.arange(L) / 16000.0)).astype(np.float32)
            else:
                y = (scale * speech_like(L, seed=case)).astype(np.float32)
            want = R.melspectrogram(y=y, **kw).astype(np.float64)
            got = S.melspectrogram(y=y, **kw).astype(np.float64)
            assert got.shape == want.shape
            dbw = 10.0 * np.log10(np.maximum(1e-10, want))
            dbg = 10.0 * np.log10(np.maximum(1e-10, got))
            d = np.abs(dbw - dbg)
            e80 = float(d[dbw >= dbw.max() - 80.0].max())
            e55 = float(d[dbw >= dbw.max() - 55.0].max())
            f_want = R.mfcc(y=y, n_mfcc=13, **kw)
            f_got = S.mfcc(y=y, n_mfcc=13, **kw)
            ef = float(np.abs(f_want - f_got).max())
            if e80 > worst_mel:
                where = dict(kw, L=L, kind=kind, scale=scale)
            worst_mel, worst_mel55, worst_mfcc = max(worst_mel, e80), max(worst_mel55, e55), max(worst_mfcc, ef)
    print(f"short/reflect: worst log-mel diff {worst_mel:.2e} dB inside top_db, {worst_mel55:.2e} dB within 55 dB of the "
          f"peak, worst MFCC diff {worst_mfcc:.2e} dB; worst case {where}")
    assert worst_mel <= SHORT_REFLECT_LOGMEL_TOL
    assert worst_mel55 <= SHORT_REFLECT_LOGMEL_55_TOL
    assert worst_mfcc <= SHORT_REFLECT_MFCC_TOL


@pytest.mark.parametrize("n_fft,hop,around", [(2048, 1024, 9000), (2048, 1500, 20000), (512, 400, 1990), (1024, 1000, 7000),
                                              (480, 240, 3000), (960, 480, 5000), (600, 310, 4000), (256, 250, 1130)])
def test_reflect_padding_of_tail_frames_follows_librosa_slices(S, n_fft, hop, around):
    """hop >= n_fft / 2 with pad_mode='reflect' and a signal length that is a multiple of the hop:
    librosa 0.10 pads the last frame from a slice of exactly n_fft/2 samples, one fewer than numpy
    needs to reflect n_fft/2 samples without turning round, so the frame's last sample is
    y[lo + 1] rather than y[lo - 1].  Every front-end kernel indexes the same way: with a loud
    sample at lo - 1 (which a reflection of the whole signal would fold into the frame) the |STFT|
    tap still equals the oracle's to fp32 accuracy."""
    from tests.test_oracle_known_answers import _short_tail_length

    T = _short_tail_length(n_fft, hop, around)
    half = n_fft // 2
    lo = ((T + half - n_fft) // hop + 1) * hop - half           # the slice the last frame is padded from: y[lo:]
    assert T - lo == half
    y = (1e-3 * np.random.default_rng(n_fft + hop).standard_normal(T)).astype(np.float32)
    y[lo - 1] = 0.9
    want = np.abs(R.stft(y, n_fft=n_fft, hop_length=hop, pad_mode="reflect")).astype(np.float64)
    got = S.spectrogram(y=y, n_fft=n_fft, hop_length=hop, pad_mode="reflect", power=1.0).astype(np.float64)
    assert got.shape == want.shape
    assert np.abs(got[:, -1] - want[:, -1]).max() / want[:, -1].max() < 1e-5
    assert np.abs(got - want).max() / want.max() < 1e-5
    sr = 16000 if n_fft < 2048 else 22050
    n_mels = 40 if n_fft < 2048 else 128
    np.testing.assert_allclose(S.mfcc(y=y, n_fft=n_fft, hop_length=hop, pad_mode="reflect", n_mfcc=13, sr=sr, n_mels=n_mels),
                               R.mfcc(y=y, n_fft=n_fft, hop_length=hop, pad_mode="reflect", n_mfcc=13, sr=sr, n_mels=n_mels),
                               rtol=0, atol=DB_TOL)


# n_fft = 16 A for every A from 8 to 63 that has no power-of-two kernel: every multiple of 16 from
# 128 to 1008 except 256 and 512
MULTIPLES_OF_16 = [n for n in range(128, 1009, 16) if n not in (256, 512)]


@pytest.mark.parametrize("n_fft", MULTIPLES_OF_16)
def test_every_multiple_of_16_runs_in_registers(S, n_fft, monkeypatch):
    """Each of these lengths has an in-register kernel of its own ('fft<n_fft>'), which matches the
    oracle on a ragged batch (odd hop, edge frames, an utterance shorter than a frame, both pad
    modes by turns, a window shorter than the transform every third length) to the parity
    tolerance, the |X|^2 tap included, and the dense-DFT kernel on the same batch."""
    import warnings

    import torch

    i = MULTIPLES_OF_16.index(n_fft)
    kw = dict(n_fft=n_fft, hop_length=n_fft // 4 + 1, n_mels=40 if n_fft < 320 else 80, n_mfcc=13, sr=16000,
              pad_mode="reflect" if i % 2 else "constant")
    if i % 3 == 2:
        kw["win_length"] = n_fft - (i + 1) - 1
    cfg = S.make_config(**kw)
    plan = S.MfccPlan(cfg)
    assert plan.front_kernel == f"fft{n_fft}"
    ys = [speech_like(n, seed=300 + i + j) for j, n in enumerate([20011, n_fft // 3, 4 * n_fft + 5, n_fft])]
    so = np.zeros(len(ys) + 1, dtype=np.int64)
    np.cumsum([len(y) for y in ys], out=so[1:])
    pcm = torch.from_numpy(np.concatenate(ys)).cuda()
    fo = plan.frame_offsets(so)
    got = plan.execute(pcm, torch.from_numpy(so).cuda(), torch.from_numpy(fo).cuda(), int(fo[-1])).cpu().numpy()
    plan.close()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for u, y in enumerate(ys):
            want = R.mfcc(y=y, **kw).T
            np.testing.assert_allclose(got[int(fo[u]):int(fo[u + 1])], want, rtol=0, atol=DB_TOL)
        y = ys[0]
        skw = {k: v for k, v in kw.items() if k in ("n_fft", "hop_length", "win_length", "pad_mode")}
        want = np.abs(R.stft(y, **skw)) ** 2
        spec = S.spectrogram(y=y, power=2.0, **skw)
    assert spec.shape == want.shape
    assert (np.abs(spec - want) <= 1e-4 * want + 2e-6 * want.max(axis=0, keepdims=True)).all()
    monkeypatch.setenv("MFCC_HIP_DENSE_DFT", "1")
    dense = S.MfccPlan(cfg)
    assert dense.front_kernel == "dense_dft"
    other = dense.execute(pcm, torch.from_numpy(so).cuda(), torch.from_numpy(fo).cuda(), int(fo[-1])).cpu().numpy()
    dense.close()
    assert np.abs(got - other).max() <= DB_TOL
