"""Global cepstral mean / variance normalisation across shards -- an EXTENSION, off by default.

The reference never normalises its features (no mean/variance code anywhere under
/root/reference/src); this exists for the ASR-standard front-end configuration and is the only
place the engine needs a collective: one all-reduce (RCCL over xGMI when the process group's
backend is nccl) of 2 * n_mfcc + 1 float64 sums.
"""
from __future__ import annotations



import numpy as np
import torch

from . import _lib


def new_stats(n_mfcc: int, device) -> torch.Tensor:
    """[sum x (n_mfcc), sum x^2 (n_mfcc), frame count] as float64 zeros."""
    return torch.zeros(2 * n_mfcc + 1, dtype=torch.float64, device=device)


def accumulate(feats: torch.Tensor, stats: torch.Tensor) -> None:
    """Add this batch's per-coefficient sums to ``stats`` (HIP kernel, float64 accumulation)."""
    assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.ndim == 2
    n_rows, n_mfcc = feats.shape
    assert stats.numel() == 2 * n_mfcc + 1 and stats.dtype == torch.float64 and stats.device == feats.device
    with torch.cuda.device(feats.device):
        stream = torch.cuda.current_stream(feats.device).cuda_stream
        _lib.check(_lib.lib().mfcc_cmvn_accumulate(feats.data_ptr(), n_rows, n_mfcc, stats.data_ptr(), stream))


def allreduce(stats: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the statistics over all ranks, in place (no-op without an initialised process group).
    With backend ``nccl`` this is one RCCL all-reduce of 216 bytes for n_mfcc = 13."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def apply(feats: torch.Tensor, stats: torch.Tensor, eps: float = 1e-10) -> None:
    """In place: x = (x - mean) / sqrt(max(var, eps)), per coefficient (HIP kernel)."""
    assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous() and feats.ndim == 2
    n_rows, n_mfcc = feats.shape
    with torch.cuda.device(feats.device):
        stream = torch.cuda.current_stream(feats.device).cuda_stream
        _lib.check(_lib.lib().mfcc_cmvn_apply(feats.data_ptr(), n_rows, n_mfcc, stats.data_ptr(), float(eps), stream))


def mean_std(stats, eps: float = 1e-10):
    """Host-side view of the statistics: (mean, std) per coefficient."""
    s = stats.detach().cpu().numpy() if isinstance(stats, torch.Tensor) else np.asarray(stats, dtype=np.float64)
    n = (len(s) - 1) // 2
    cnt = s[2 * n]
    mean = s[:n] / cnt
    var = np.maximum(s[n:2 * n] / cnt - mean * mean, eps)
    return mean, np.sqrt(var)
