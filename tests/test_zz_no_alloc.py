"""mfcc_execute_noalloc (include/mfcc_hip.h): the entry point that allocates nothing, whose
workspace is a static device array inside the library.  Called through the C ABI twice in a row,
the second time with a smaller batch, and compared bit for bit with mfcc_execute on the same inputs
(which the parity tests compare with the oracle).  Collected last: it must pass alone in a fresh
process and after every other test."""
import ctypes as C

import numpy as np
import pytest

from tests.signals import speech_like

pytestmark = pytest.mark.gpu


def _batch(lib, cfg, lens, seed0, gpu):
    import torch

    ys = [speech_like(n, seed=seed0 + i) * np.float32(10.0 ** (-i / 2.0)) for i, n in enumerate(lens)]
    so = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    fo = np.zeros_like(so)
    assert lib.mfcc_batch_layout(C.byref(cfg), so.ctypes.data_as(C.POINTER(C.c_int64)), len(ys),
                                 fo.ctypes.data_as(C.POINTER(C.c_int64))) == 0
    pcm = torch.from_numpy(np.concatenate(ys)).to(gpu)
    return pcm, torch.from_numpy(so).to(gpu), torch.from_numpy(fo).to(gpu), len(ys), int(fo[-1])


def test_noalloc_entry_point_matches_mfcc_execute(gpu):
    import torch

    from speech_to_text_amd import _lib

    lib = _lib.lib()
    cfg = _lib.MfccConfigC()
    lib.mfcc_config_default(C.byref(cfg))
    cfg.n_mfcc = 13
    plan = C.c_void_p()
    _lib.check(lib.mfcc_plan_create(C.byref(cfg), C.byref(plan)))
    try:
        stream = torch.cuda.current_stream().cuda_stream
        # first call: a ragged batch with a long and several short utterances; second call: smaller
        for lens, seed0 in (([250000, 30000, 2048, 77, 91234, 600000], 900), ([41000, 5000, 123456], 950)):
            pcm, d_so, d_fo, n, F = _batch(lib, cfg, lens, seed0, gpu)
            need = lib.mfcc_workspace_bytes(plan, F, n)
            ws = torch.empty(need, dtype=torch.uint8, device=gpu)
            want = torch.full((F, 13), float("nan"), device=gpu)
            _lib.check(lib.mfcc_execute(plan, pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), None, n, F,
                                        want.data_ptr(), None, ws.data_ptr(), need, stream))
            got = torch.full((F, 13), float("nan"), device=gpu)
            _lib.check(lib.mfcc_execute_noalloc(plan, pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), None, n, F,
                                                got.data_ptr(), stream))
            torch.cuda.synchronize()
            assert torch.isfinite(want).all()
            assert torch.equal(got, want)
        # a batch beyond the static scratch is refused with a clear error, and nothing is launched
        big_frames = 1 + _lib.NOALLOC_SCRATCH_BYTES // (4 * 128)
        rc = lib.mfcc_execute_noalloc(plan, pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), None, n, big_frames,
                                      got.data_ptr(), stream)
        assert rc == _lib.MFCC_E_WORKSPACE and "static scratch" in _lib.last_error()
    finally:
        lib.mfcc_plan_destroy(plan)
