"""ctypes binding of libwavenet_hip.so (declarations: include/wavenet_hip.h).

The library is the product: if it is missing or a symbol does not resolve, importing this
module raises.  There is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own copy of the HIP runtime.  Device pointers and streams handed to the
# library come from torch, so the library must bind to that same runtime instance: load
# torch first, then libwavenet_hip.so resolves libamdhip64 to the copy already mapped.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# WN_HIP_LIB names another build of the same library (used to time two builds side by side)
LIB_PATH = os.environ.get("WN_HIP_LIB") or os.path.join(_HERE, "csrc", "libwavenet_hip.so")


class WnScheme(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "sample_rate", "base_freq", "freq_step", "max_byte", "tol_check", "sep_freq", "ping_freq",
        "eof_freq", "tone_n", "sep_n", "sep_short_n", "ping_n", "eof_n", "win_n")] + \
        [("amp", C.c_double), ("gain", C.c_double)]


class WnFrameRec(C.Structure):
    _fields_ = [("crc", C.c_uint32), ("status", C.c_uint8), ("crc_ok", C.c_uint8),
                ("tipo", C.c_uint8), ("length", C.c_uint8)]


WN_OK, WN_ERR_ARG, WN_ERR_HIP, WN_ERR_UNSUPPORTED, WN_ERR_NOMEM = 0, -1, -2, -3, -4
CLS_BYTE_MASK, CLS_NONE, CLS_PING, CLS_EOF = 0x1FF, 256, 1 << 9, 1 << 10
ST_OK, ST_EMPTY, ST_SHORT_HEADER, ST_SHORT_BODY = 0, 1, 2, 3
END_EOF, END_TIMEOUT, END_OVERFLOW = 0, 1, 0x100
EMIT_LIVE, EMIT_WAV = 0, 1
FMT_F32, FMT_I16 = 0, 1
SYNTH_BESIDE = 0x100   # or-ed into the format: the small-footprint synthesis kernel
DETECT_NOALLOC_MAX_WINDOWS = 1 << 20

_vp, _i32, _i64, _u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/wavenet_hip.h declares
SIGNATURES = {
    "wn_strerror": (C.c_char_p, [C.c_int]),
    "wn_last_error": (C.c_char_p, []),
    "wn_host_class_lut": (C.c_int, [_P(WnScheme), _P(C.c_uint16), _i32]),
    "wn_host_dft_plan": (C.c_int, [_P(WnScheme), _P(_i32), _P(_i32)]),
    "wn_host_sine_tables": (C.c_int, [_P(WnScheme), _P(C.c_float), _P(C.c_int16), _i32]),
    "wn_host_frame_samples": (_i64, [_P(WnScheme), C.c_char_p, _i32, _i32]),
    "wn_scheme_init": (C.c_int, [_P(WnScheme), _P(_vp)]),
    "wn_ctx_destroy": (C.c_int, [_vp]),
    "wn_detect_set_tie_band": (C.c_int, [_vp, C.c_double]),
    "wn_detect_stats": (C.c_int, [_vp, _P(_i64), _vp]),
    "wn_crc32_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "wn_frame_build_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _vp]),
    "wn_frame_parse_validate_batch": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp]),
    "wn_synth_layout": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "wn_synth_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "wn_channel_awgn": (C.c_int, [_vp, _vp, _i64, C.c_float, _u64, _vp]),
    "wn_detect_windows": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "wn_detect_windows_noalloc": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp]),
    "wn_sync_scan": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp]),
    "wn_slice_frames": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wn_selftest_mfma": (C.c_int, [_vp, _vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C wavenet_amd/csrc`). wavenet_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class WnError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        detail = lib.wn_last_error().decode(errors="replace")
        super().__init__(f"{where}: {lib.wn_strerror(code).decode()} ({code}): {detail}")


def check(code: int, where: str) -> None:
    if code != WN_OK:
        raise WnError(code, where)
