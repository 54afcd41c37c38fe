"""ctypes binding of ``libmfcc_hip.so`` (the C ABI declared in ``include/mfcc_hip.h``).

There is no CPU fallback: if the shared library is missing or cannot be loaded the import of
this module's :func:`lib` raises, and every feature function built on it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 1

MFCC_OK = 0
MFCC_E_INVALID = -1
MFCC_E_UNSUPPORTED = -2
MFCC_E_HIP = -3
MFCC_E_WORKSPACE = -4
MFCC_E_NOMEM = -5

NOALLOC_SCRATCH_BYTES = 64 * 1024 * 1024   # MFCC_NOALLOC_SCRATCH_BYTES

PAD_CONSTANT = 0
PAD_REFLECT = 1


class MfccConfigC(C.Structure):
    """``struct mfcc_config`` -- field order and types must match include/mfcc_hip.h."""

    _fields_ = [
        ("struct_size", C.c_int32),
        ("n_fft", C.c_int32),
        ("hop_length", C.c_int32),
        ("win_length", C.c_int32),
        ("center", C.c_int32),
        ("pad_mode", C.c_int32),
        ("power", C.c_float),
        ("sr", C.c_double),
        ("n_mels", C.c_int32),
        ("fmin", C.c_double),
        ("fmax", C.c_double),
        ("htk", C.c_int32),
        ("mel_norm", C.c_int32),
        ("ref", C.c_float),
        ("amin", C.c_float),
        ("top_db", C.c_float),
        ("n_mfcc", C.c_int32),
        ("dct_type", C.c_int32),
        ("dct_norm", C.c_int32),
        ("lifter", C.c_float),
        ("window", C.POINTER(C.c_double)),
        ("mel_basis", C.POINTER(C.c_float)),
    ]


class LibraryError(RuntimeError):
    """A call into libmfcc_hip failed; ``code`` is the MFCC_E_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libmfcc_hip error {code}: {message}")
        self.code = code
        self.message = message


def library_path() -> Path:
    env = os.environ.get("MFCC_HIP_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "lib" / "libmfcc_hip.so"


# name -> (restype, argtypes); every symbol include/mfcc_hip.h declares.
SYMBOLS = {
    "mfcc_config_default": (None, [C.POINTER(MfccConfigC)]),
    "mfcc_abi_version": (C.c_int, []),
    "mfcc_last_error": (C.c_char_p, []),
    "mfcc_host_window": (C.c_int, [C.POINTER(MfccConfigC), C.POINTER(C.c_double)]),
    "mfcc_host_mel_basis": (C.c_int, [C.POINTER(MfccConfigC), C.POINTER(C.c_float)]),
    "mfcc_host_dct_basis": (C.c_int, [C.POINTER(MfccConfigC), C.POINTER(C.c_float)]),
    "mfcc_num_frames": (C.c_int64, [C.POINTER(MfccConfigC), C.c_int64]),
    "mfcc_batch_layout": (C.c_int, [C.POINTER(MfccConfigC), C.POINTER(C.c_int64), C.c_int32,
                                    C.POINTER(C.c_int64)]),
    "mfcc_plan_create": (C.c_int, [C.POINTER(MfccConfigC), C.POINTER(C.c_void_p)]),
    "mfcc_plan_destroy": (None, [C.c_void_p]),
    "mfcc_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "mfcc_execute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_void_p]),
    "mfcc_execute_pcm16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "mfcc_execute_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "mfcc_workspace_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mfcc_execute_spectrogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    "mfcc_plan_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "mfcc_plan_last_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mfcc_plan_timing_history": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mfcc_plan_front_kernel": (C.c_char_p, [C.c_void_p]),
    "mfcc_to_db": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float,
                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfcc_dct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "mfcc_delta": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                             C.c_void_p, C.c_void_p]),
    "mfcc_host_pcm16_from_float": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "mfcc_host_write_feature_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                               C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                               C.POINTER(C.c_char_p)]),
    "mfcc_host_write_feature_files": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.c_char_p, C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_int64), C.c_int64, C.c_int32, C.POINTER(C.c_char_p),
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_char_p)]),
    "mfcc_host_write_feature_files_mt": (C.c_int, [C.c_int32, C.POINTER(C.c_char_p), C.c_char_p, C.c_void_p,
                                                   C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_char_p),
                                                   C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_char_p), C.c_int32]),
    "mfcc_host_pack_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int32]),
    "mfcc_host_file_error": (C.c_char_p, []),
    "mfcc_cmvn_accumulate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "mfcc_cmvn_apply": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_double,
                                  C.c_void_p]),
}

_LIB = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library with argument types set."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build it with `make -C speech_to_text_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback for this package")
    handle = C.CDLL(str(path))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(handle, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if handle.mfcc_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {handle.mfcc_abi_version()} != {ABI_VERSION}")
    _LIB = handle
    return handle


def last_error() -> str:
    return lib().mfcc_last_error().decode("utf-8", "replace")


def check(code: int) -> None:
    if code != MFCC_OK:
        raise LibraryError(code, last_error())
