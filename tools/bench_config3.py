#!/usr/bin/env python3
"""BASELINE.json configs[2] as written: 1000 hours of DISTINCT synthetic 16 kHz audio, ragged 2-35 s
utterances, sharded by utterance over the ranks (LPT on frame count) and pushed through ONE plan
and ONE workspace in resident batches of at most 8 GB of PCM, every batch generated on the device
with a counter-based generator keyed by (seed, utterance id, sample index) and every batch with
its own ragged layout.

    python tools/bench_config3.py [--hours 1000] [--batch-gb 8] [--out file.json]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_config3.py   (one rank per GPU)

Per batch: HIP-event times of the two kernels (the library's event ring) and the wall time of the
execute call with a synchronisation either side; generation is outside the timed region.  One
utterance of a batch in the middle of the run is checked against the CPU oracle.  The looped
variant (one 10 h batch, 100 passes) is tools/bench_configs.py."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import speech_to_text_amd as S  # noqa: E402
from speech_to_text_amd.shard import lpt_shards, pack_batches  # noqa: E402

from bench import SR, corpus_lengths, generate_batch  # noqa: E402  (the corpus and its device-side generator)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hours", type=float, default=1000.0)
    ap.add_argument("--batch-gb", type=float, default=8.0)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # the corpus: lengths U(2, 35) s until the target is reached; the same on every rank
    lens = corpus_lengths(args.hours)
    n_utts = len(lens)
    mine = lpt_shards(1 + lens // 512, world)[rank] if world > 1 else np.arange(n_utts)
    batches = pack_batches(lens[mine], int(args.batch_gb * 1e9 / 4))

    cfg = S.MfccConfig.reference_default()
    plan = S.MfccPlan(cfg, dev)
    plan.set_profiling(True)
    fmax = max(int((1 + lens[mine[b]] // 512).sum()) for b in batches)
    umax = max(len(b) for b in batches)
    ws = plan.workspace(fmax, umax)
    out = torch.empty((fmax, cfg.n_mfcc), dtype=torch.float32, device=dev)
    recs, check = [], None
    t_all = time.perf_counter()
    for k, b in enumerate(batches):
        ids = mine[b]
        pcm, so = generate_batch(ids, lens[ids], dev)
        fo = plan.frame_offsets(so)
        d_so, d_fo = torch.from_numpy(so).to(dev), torch.from_numpy(fo).to(dev)
        F = int(fo[-1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.execute(pcm, d_so, d_fo, F, out=out[:F], workspace=ws)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        front, dct = plan.last_timings()
        hours = float(so[-1]) / SR / 3600.0
        recs.append({"batch": k, "utterances": int(len(ids)), "audio_hours": hours, "frames": F,
                     "pcm_GB": 4.0 * float(so[-1]) / 1e9, "front_kernel_ms": front, "db_dct_kernel_ms": dct,
                     "execute_wall_ms": wall * 1e3, "checksum": float(out[:F].double().sum().item())})
        if k == len(batches) // 2 and rank == 0:
            from oracle import mfcc_ref as R

            u = len(ids) // 2
            y = pcm[int(so[u]):int(so[u + 1])].cpu().numpy()
            got = out[int(fo[u]):int(fo[u + 1])].cpu().numpy()
            check = {"batch": k, "utterance_id": int(ids[u]), "samples": int(len(y)),
                     "max_abs_diff_vs_oracle_db": float(np.abs(got - R.process_audio_features(y)).max())}
        del pcm, d_so, d_fo
        if rank == 0:
            print(f"batch {k + 1}/{len(batches)}: {hours:.1f} h, front {front:.3f} ms, dB+DCT {dct:.3f} ms", flush=True)
    total_wall = time.perf_counter() - t_all
    hours = sum(r["audio_hours"] for r in recs)
    kern = sum(r["front_kernel_ms"] + r["db_dct_kernel_ms"] for r in recs) * 1e-3
    exe = sum(r["execute_wall_ms"] for r in recs) * 1e-3
    res = {"config": f"BASELINE configs[2]: {args.hours:g} h of distinct synthetic 16 kHz audio, ragged 2-35 s, "
                     f"batches of <= {args.batch_gb:g} GB PCM generated on the device, one plan, one workspace",
           "world_size": world, "rank": rank, "utterances_total": int(n_utts), "utterances_this_rank": int(len(mine)),
           "audio_hours_this_rank": hours, "batches": len(recs),
           "audio_hours_per_s_kernels": hours / kern, "audio_hours_per_s_execute_wall": hours / exe,
           "seconds_kernels": kern, "seconds_execute_wall": exe, "seconds_total_incl_generation": total_wall,
           "oracle_check": check, "per_batch": recs}
    if dist is not None:
        t = torch.tensor([exe, hours], dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        res["all_ranks"] = {"audio_hours": sum(float(p[1]) for p in parts), "max_seconds_execute_wall": max(float(p[0]) for p in parts)}
        res["all_ranks"]["audio_hours_per_s"] = res["all_ranks"]["audio_hours"] / res["all_ranks"]["max_seconds_execute_wall"]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        text = json.dumps(res, indent=1)
        print(json.dumps({k: v for k, v in res.items() if k != "per_batch"}, indent=1))
        if args.out:
            Path(args.out).write_text(text)


if __name__ == "__main__":
    main()
