"""
ctypes binding of libbonito_hip.so (C ABI: include/bonito_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the product path
raises. Build with ``python build.py`` (hipcc, gfx950).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BONITO_HIP_LIB") or os.path.join(_HERE, "libbonito_hip.so")     # override: A/B builds


class HipEngineError(RuntimeError):
    """An entry point of libbonito_hip.so reported an error."""


class bh_layer_t(C.Structure):
    # must mirror include/bonito_hip.h `struct bh_layer`
    _fields_ = [
        ("kind", C.c_int32), ("in_size", C.c_int32), ("out_size", C.c_int32),
        ("winlen", C.c_int32), ("stride", C.c_int32), ("padding", C.c_int32),
        ("activation", C.c_int32), ("reverse", C.c_int32),
        ("nhead", C.c_int32), ("dim_ff", C.c_int32), ("win_left", C.c_int32), ("win_right", C.c_int32),
        ("scale_factor", C.c_int32), ("groups", C.c_int32), ("add_residual", C.c_int32), ("quantize", C.c_int32), ("reserved_i", C.c_int32 * 1),
        ("scale", C.c_float), ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
        ("blank_score", C.c_float), ("alpha", C.c_float), ("eps", C.c_float), ("reserved_f", C.c_float * 2),
        ("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
        ("w2", C.c_void_p), ("w3", C.c_void_p), ("w4", C.c_void_p), ("w5", C.c_void_p),
    ]


BH_ACT = {None: 0, "none": 0, "swish": 1, "tanh": 2, "relu": 3}
BH_LAYER_CONV, BH_LAYER_LSTM, BH_LAYER_LINEAR_CRF, BH_LAYER_CLAMP = 1, 2, 3, 4
BH_LAYER_TRANSFORMER, BH_LAYER_UPSAMPLE, BH_LAYER_TCS_BLOCK, BH_LAYER_CTC_DECODER = 5, 6, 7, 8
BH_LAYER_DWCONV, BH_LAYER_RESIDUAL_PROJ, BH_LAYER_LINEAR = 9, 10, 11

_vp, _i, _f, _l, _sz = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_size_t

# name -> (restype, argtypes).  Every symbol include/bonito_hip.h declares must be listed here
# (tests/test_abi.py cross-checks the header against this table and against the built library).
SIGNATURES = {
    "bh_last_error": (C.c_char_p, []),
    "bh_abi_version": (_i, []),
    "bh_sizeof_layer": (_sz, []),
    "bh_device_count": (_i, []),
    "bh_encoder_create": (_i, [C.POINTER(bh_layer_t), _i, _i, _i, _i, C.POINTER(_vp)]),
    "bh_encoder_destroy": (None, [_vp]),
    "bh_encoder_output_shape": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "bh_encoder_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "bh_encoder_check": (_i, [_vp, _vp]),
    "bh_encoder_error_flag": (_i, [_vp]),
    "bh_encoder_last_ticket": (_l, [_vp]),
    "bh_encoder_error_flag_at": (_i, [_vp, _l]),
    "bh_encoder_ack": (_i, [_vp, _l]),
    "bh_encoder_describe": (_i, [_vp, C.c_char_p, _sz]),
    "bh_encoder_profile": (_i, [_vp, _i]),
    "bh_encoder_profile_read": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(_i)]),
    "bh_crf_viterbi_workspace": (_sz, [_i, _i, _i]),
    "bh_crf_viterbi": (_i, [_vp, _i, _i, _i, _i, _f, _l, _l, _vp, _vp, _vp, _vp, _vp]),
    "bh_crf_reverse_complement": (_i, [_vp, _vp, _i, _i, _i, _i, _l, _l, _vp]),
    "bh_crf_logz": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "bh_set_option": (_i, [C.c_char_p, _i]),
    "bh_signal_normalise": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                            _i, _vp, _vp, _vp, _vp, _vp]),
    "bh_signal_chunks": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "bh_crf_posterior_viterbi_workspace": (_sz, [_i, _i, _i]),
    "bh_crf_posterior_viterbi": (_i, [_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "bh_beam_search_workspace": (_sz, [_i, _i, _i]),
    "bh_beam_search": (_i, [_vp, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bh_linear": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _l, _l, _i, _vp]),
    "bh_conv1d_first": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _l, _l, _vp]),
    "bh_conv1d_packed_halves": (_sz, [_i, _i, _i]),
    "bh_conv1d_pack": (_i, [_vp, _i, _i, _i, _vp]),
    "bh_conv1d": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _l, _l, _vp]),
    "bh_ctc_greedy_decode": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "bh_ctc_beam_search_workspace": (_sz, [_l, _i, _i, _i]),
    "bh_ctc_beam_search": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "bh_dwconv1d": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bh_rotary_table": (_i, [_i, _i, _vp]),
    "bh_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bh_attention_prerotated": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "bh_rmsnorm_residual": (_i, [_vp, _vp, _vp, _vp, _l, _i, _f, _f, _vp]),
    "bh_lstm_pack_whh": (_i, [_vp, _i, _vp]),
    "bh_host_compact": (_l, [_vp, _l, _vp]),
    "bh_host_chunk_rows": (_l, [_vp, _l, _i, _i, _l, _l, _vp]),
    "bh_host_format_read": (_l, [_vp, _vp, _vp, _vp, _i, _l, _l, _i, _i, _i, _i, _i, _i, C.c_double, C.c_char_p, C.c_char_p, _l, _l,
                                 _vp, _l, _vp, _vp]),
    "bh_host_mean_qscore": (C.c_double, [C.c_char_p, _l]),
    "bh_host_svb16_decode": (_l, [_vp, _l, _l, _vp]),
    "bh_lstm_workspace": (_sz, [_i, _i]),
    "bh_lstm_layer": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "bh_lstm_q8_layer": (_i, [_vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "bh_encoder_debug_read": (_i, [_vp, _vp, _sz, _sz]),
    "bh_encoder_set_option": (_i, [_vp, C.c_char_p, _i]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipEngineError(
                "%s not found: the MI355X engine is not built (run `python build.py`). "
                "bonito_amd has no CPU fallback." % LIB_PATH
            )
        # Load torch (and with it the HIP runtime its wheel bundles, SONAME libamdhip64.so.7) BEFORE this library: whichever
        # libamdhip64.so.7 is mapped first serves the whole process. Loaded the other way round (this library pulling
        # /opt/rocm's runtime first, torch imported later) the two disagree and hipGetDeviceCount reports no device.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing -> loud
            fn.restype = res
            fn.argtypes = args
        if handle.bh_sizeof_layer() != C.sizeof(bh_layer_t):
            raise HipEngineError("bh_layer_t layout mismatch between include/bonito_hip.h and bonito_amd/_lib.py")
        _lib = handle
    return _lib


def last_error():
    msg = lib().bh_last_error()
    return msg.decode() if msg else ""


def check(rc, what=""):
    if rc != 0:
        raise HipEngineError("%s failed (rc=%d): %s" % (what or "libbonito_hip call", rc, last_error()))


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device` (torch is only the allocator/stream owner)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device/host address of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())
