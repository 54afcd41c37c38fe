"""GPU parity at the sizes the kernels' 64-bit index arithmetic exists for: one batch of more than
2^31 samples (float32 and 16-bit PCM), and the batch turnover of BASELINE.json configs[2]
(several resident batches of up to 8 GB with different ragged layouts through one plan and one
workspace).  Inputs are generated on the device, as bench.py does; the oracle checks utterances
at the far end of the buffers, where a 32-bit offset would have wrapped.
"""
import numpy as np
import pytest

from oracle import mfcc_ref as R

pytestmark = pytest.mark.gpu

DB_TOL = 1e-2          # dB, absolute, on MFCC


@pytest.fixture(scope="module")
def S(gpu):
    import speech_to_text_amd as pkg

    return pkg


def _corpus_slice(hours, first_id=0):
    """Utterance ids and lengths of a slice of the bench's fixed corpus (ragged 2-35 s)."""
    from bench import corpus_lengths

    lens = corpus_lengths(hours)
    return np.arange(first_id, first_id + len(lens)), lens


@pytest.mark.parametrize("dtype_name", ["float32", "int16"])
def test_batch_beyond_2p31_samples(S, dtype_name):
    """One resident batch of 2.2e9 samples (8.8 GB as float32, 4.4 GB as 16-bit PCM).  The last three
    utterances, the one that straddles sample 2^31 and the first one equal the oracle; both input
    types give the same features bit for bit (the samples are on the int16 grid)."""
    import torch

    from bench import generate_batch

    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 24e9:
        pytest.skip("needs 24 GB of free device memory")
    dtype = torch.float32 if dtype_name == "float32" else torch.int16
    ids, lens = _corpus_slice(38.0)
    assert int(lens.sum()) > 2 ** 31 + 10 * 35 * 16000
    cfg = S.MfccConfig.reference_default()
    plan = S.get_plan(cfg, dev)
    pcm, so = generate_batch(ids, lens, dev, dtype=dtype)
    fo = plan.frame_offsets(so)
    F = int(fo[-1])
    d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
    out = plan.execute(pcm, d_so, d_fo, F)
    torch.cuda.synchronize()
    straddle = int(np.searchsorted(so, 2 ** 31, side="right")) - 1
    assert so[straddle] < 2 ** 31 < so[straddle + 1]
    n = len(lens)
    for u in (0, straddle, straddle + 1, n - 3, n - 2, n - 1):
        y = pcm[int(so[u]):int(so[u + 1])].cpu().numpy()
        if dtype == torch.int16:
            y = y.astype(np.float32) / 32768.0
        got = out[int(fo[u]):int(fo[u + 1])].cpu().numpy()
        want = R.process_audio_features(y)
        assert got.shape == want.shape
        err = float(np.abs(got - want).max())
        assert err <= DB_TOL, f"utterance {u} (samples {so[u]}..{so[u + 1]}): {err} dB"
    # the far end of the output is written and finite; a checksum ties the two input types together
    assert bool(torch.isfinite(out[-100000:]).all())
    chk = float(out.double().sum().item())
    key = "_chk_2p31"
    prev = getattr(test_batch_beyond_2p31_samples, key, None)
    if prev is not None:
        assert prev == chk, "float32 and int16 input gave different features"
    setattr(test_batch_beyond_2p31_samples, key, chk)
    del pcm, out
    torch.cuda.empty_cache()


def test_batch_turnover_one_plan_one_workspace(S):
    """BASELINE configs[2]'s turnover at full batch size: a large (6.9 GB), a small and another large
    batch with different ragged layouts through ONE plan, ONE workspace and ONE output buffer;
    each batch equals a run with a fresh workspace bit for bit, and two utterances of each equal
    the oracle."""
    import torch

    from bench import generate_batch

    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 30e9:
        pytest.skip("needs 30 GB of free device memory")
    cfg = S.MfccConfig.reference_default()
    plan = S.MfccPlan(cfg, dev)
    ids_all, lens_all = _corpus_slice(60.0)
    cuts = np.searchsorted(np.cumsum(lens_all), [int(30 * 3600 * 16000), int(32 * 3600 * 16000)])
    parts = [(ids_all[:cuts[0]], lens_all[:cuts[0]]), (ids_all[cuts[0]:cuts[1]], lens_all[cuts[0]:cuts[1]]),
             (ids_all[cuts[1]:], lens_all[cuts[1]:])]
    fmax = max(int((1 + l // 512).sum()) for _, l in parts)
    umax = max(len(l) for _, l in parts)
    ws = plan.workspace(fmax, umax)
    out = torch.empty((fmax, cfg.n_mfcc), dtype=torch.float32, device=dev)
    out.fill_(float("nan"))
    for k, (ids, lens) in enumerate(parts):
        pcm, so = generate_batch(ids, lens, dev)
        assert 4 * int(so[-1]) <= 8e9
        fo = plan.frame_offsets(so)
        F = int(fo[-1])
        d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
        plan.execute(pcm, d_so, d_fo, F, out=out[:F], workspace=ws)
        fresh_ws = torch.empty(ws.numel(), dtype=torch.uint8, device=dev)
        fresh = plan.execute(pcm, d_so, d_fo, F, workspace=fresh_ws)
        torch.cuda.synchronize()
        assert torch.equal(out[:F], fresh), f"batch {k}: the reused workspace changed the result"
        for u in (len(lens) // 2, len(lens) - 1):
            y = pcm[int(so[u]):int(so[u + 1])].cpu().numpy()
            err = float(np.abs(out[int(fo[u]):int(fo[u + 1])].cpu().numpy() - R.process_audio_features(y)).max())
            assert err <= DB_TOL, f"batch {k}, utterance {u}: {err} dB"
        del pcm, fresh, fresh_ws
        torch.cuda.empty_cache()
    plan.close()
