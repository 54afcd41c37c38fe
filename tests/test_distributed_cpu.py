"""The N > 1 path on CPU: two gloo ranks shard a set of utterances with no data-path collective
and all-reduce only the CMVN statistics (the one exchange step the engine has)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mfcc_ref as R
from speech_to_text_amd import cmvn
from speech_to_text_amd.preprocess import Preprocessor, read_feature_file
from speech_to_text_amd.shard import lpt_shards
from tests.signals import ragged_lengths, speech_like


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _items(n):
    lens = ragged_lengths(n, 0.1, 0.6, seed=5)
    return [(speech_like(int(L), seed=i)[None, :], 16000, f"T{i}", 10 + i, 20 + i, i) for i, L in enumerate(lens)]


def _extract(waveforms, *, n_fft, hop_length, n_mfcc):
    return [R.process_audio_features(np.asarray(w), n_fft=n_fft, hop_length=hop_length, n_mfcc=n_mfcc) for w in waveforms]


def _worker(rank, world, port, out_dir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = _items(11)
        pre = Preprocessor(items, out_dir=out_dir, rank=rank, world_size=world, extractor=_extract, hdf5_backend="min")
        n = pre.preprocess(verbose=False)
        # CMVN statistics of this rank's shard (host arithmetic stands in for the HIP kernel here)
        costs = 1 + np.asarray([it[0].shape[-1] for it in items]) // 512
        mine = lpt_shards(costs, world)[rank]
        feats = np.concatenate([_extract([items[i][0]], n_fft=2048, hop_length=512, n_mfcc=13)[0] for i in mine]) \
            if len(mine) else np.zeros((0, 13), np.float32)
        stats = torch.zeros(27, dtype=torch.float64)
        stats[:13] = torch.from_numpy(feats.astype(np.float64).sum(0))
        stats[13:26] = torch.from_numpy((feats.astype(np.float64) ** 2).sum(0))
        stats[26] = feats.shape[0]
        cmvn.allreduce(stats)
        q.put((rank, n, stats.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_and_allreduce(tmp_path):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path / "mfcc"), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[1] for r in results) == 11                      # every utterance written exactly once
    files = sorted(f.name for f in (tmp_path / "mfcc").iterdir())
    assert len(files) == 11
    items = _items(11)
    allf = np.concatenate([_extract([it[0]], n_fft=2048, hop_length=512, n_mfcc=13)[0] for it in items]).astype(np.float64)
    for _, _, st in results:                                     # both ranks hold the global sums
        np.testing.assert_allclose(st[:13], allf.sum(0), rtol=1e-12)
        np.testing.assert_allclose(st[13:26], (allf ** 2).sum(0), rtol=1e-12)
        assert st[26] == allf.shape[0]
    mean, std = cmvn.mean_std(results[0][2])
    np.testing.assert_allclose(mean, allf.mean(0), rtol=1e-10)
    np.testing.assert_allclose(std, allf.std(0), rtol=1e-8)
    arr, attrs = read_feature_file(tmp_path / "mfcc" / "13_23_3.hdf5")
    assert attrs["transcript"] == "T3" and np.array_equal(arr, _extract([items[3][0]], n_fft=2048, hop_length=512, n_mfcc=13)[0])


def _run_bench(args, extra_env):
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, str(root / "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=240)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no WORLD_SIZE) starts two fresh rank processes, relays rank 0's
    single JSON line and aggregates over both ranks; BENCH_STUB=1 swaps the GPU step for a sleep so
    that the launcher, the rendezvous and the max-over-ranks timing run on CPU (gloo)."""
    import json

    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--strong-hours", "5"], {"BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0
    assert len(line["per_rank_ms"]) == 2 and all(t > 0 for t in line["per_rank_ms"])
    # the fixed-corpus leg: one corpus, sharded longest-first, every hour counted once
    st = line["strong"]
    assert st["scaling"] == "strong" and len(st["per_rank_seconds"]) == 2
    assert abs(sum(st["per_rank_audio_hours"]) - st["audio_hours"]) < 1e-9 and abs(st["audio_hours"] - 5.0) < 0.02
    assert 1.0 <= st["lpt_imbalance"] < 1.01
    # the engine's one collective: both ranks seen, all-reduced statistics equal the gathered sum
    assert line["rccl"]["ranks_seen"] == 2 and line["rccl"]["cmvn_allreduce_ok"] and line["rccl"]["bytes"] == 216
    assert line["rccl"]["frames_all_ranks"] == 300.0


def test_bench_launcher_reports_a_failed_rank():
    r = _run_bench(["--gpus", "2", "--steps", "2"], {"BENCH_STUB": "1", "BENCH_STUB_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""          # no result line when a rank failed
    assert "rank exit codes" in r.stderr
