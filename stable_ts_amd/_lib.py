"""ctypes binding of libswx.so (include/swx.h).  The product path has no CPU fallback: if the library is missing or a
call is made without a GPU, this module raises."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libswx.so")

SWX_F32 = 0
SWX_F16 = 1


class SwxError(RuntimeError):
    pass


class swx_dims(Structure):
    _fields_ = [(n, c_int32) for n in ("n_mels", "n_audio_ctx", "n_audio_state", "n_audio_head", "n_audio_layer",
                                       "n_vocab", "n_text_ctx", "n_text_state", "n_text_head", "n_text_layer")]


class swx_flac_info(Structure):
    _fields_ = [("sample_rate", c_int32), ("channels", c_int32), ("bits_per_sample", c_int32), ("min_block", c_int32),
                ("max_block", c_int32), ("total_samples", c_int64), ("md5", ctypes.c_uint8 * 16)]


class swx_decode_cfg(Structure):
    _fields_ = [
        ("n_windows", c_int32), ("n_group", c_int32), ("beam", c_int32), ("temperature", c_float),
        ("patience", c_float), ("sample_len", c_int32), ("sample_begin", c_int32), ("sot_index", c_int32),
        ("suppress_blank", c_int32), ("apply_timestamp_rules", c_int32), ("max_initial_timestamp_index", c_int32),
        ("eot", c_int32), ("sot", c_int32), ("no_timestamps", c_int32), ("timestamp_begin", c_int32),
        ("no_speech", c_int32), ("blank_token", c_int32), ("n_suppress", c_int32), ("min_tokens", c_int32),
        ("seed", c_uint64), ("window_uid", POINTER(c_int32)), ("noise", c_void_p),
    ]


# every symbol include/swx.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "swx_strerror": (c_char_p, [c_int]),
    "swx_version": (c_int, []),
    "swx_model_create": (c_int, [POINTER(swx_dims), c_int, POINTER(c_void_p)]),
    "swx_model_destroy": (None, [c_void_p]),
    "swx_weights_bytes": (c_size_t, [c_void_p]),
    "swx_bind_weights": (c_int, [c_void_p, c_void_p, c_size_t]),
    "swx_share_weights": (c_int, [c_void_p, c_void_p]),
    "swx_load_tensor": (c_int, [c_void_p, c_char_p, c_void_p, c_int64, c_void_p]),
    "swx_weights_complete": (c_int, [c_void_p]),
    "swx_missing_tensor": (c_int, [c_void_p, c_int, c_char_p, c_int]),
    "swx_weights_mark_loaded": (c_int, [c_void_p]),
    "swx_weights_finalize": (c_int, [c_void_p, c_void_p]),
    "swx_set_alignment_heads": (c_int, [c_void_p, POINTER(c_int32), c_int]),
    "swx_num_alignment_heads": (c_int, [c_void_p]),
    "swx_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "swx_bind_workspace": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int]),
    "swx_log_mel": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "swx_log_mel_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "swx_encode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "swx_cross_kv_bytes": (c_size_t, [c_void_p, c_int]),
    "swx_cross_kv": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "swx_decode": (c_int, [c_void_p, POINTER(swx_decode_cfg), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "swx_decode_gout": (c_int, [POINTER(swx_decode_cfg)]),
    "swx_score": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_int, c_int, POINTER(c_int32), c_float,
                          c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "swx_score_qk": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                             c_void_p, c_void_p, c_void_p]),
    "swx_qcap_bytes": (c_size_t, [c_void_p, c_int]),
    "swx_score_q": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "swx_heads_scratch_bytes": (c_size_t, [c_void_p, c_int]),
    "swx_heads_dynamic": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p,
                                  c_int, c_void_p, c_size_t, c_void_p]),
    "swx_heads_new": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float, c_int, c_int, c_float,
                              c_float, c_float, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "swx_weighted_sum": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "swx_forward_logits": (c_int, [c_void_p, c_void_p, POINTER(c_int32), c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "swx_align_weights_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "swx_align_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int32), c_float, c_int, c_void_p,
                                  c_void_p, c_size_t, c_void_p]),
    "swx_median_filter": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "swx_loudness_probe": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "swx_loudness_probe_scratch_bytes": (c_size_t, [c_int]),
    "swx_flac_probe": (c_int, [c_void_p, c_size_t, POINTER(swx_flac_info)]),
    "swx_flac_decode": (c_int64, [c_void_p, c_size_t, c_void_p, c_int64, POINTER(swx_flac_info)]),
    "swx_dtw_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "swx_dtw": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                        c_void_p]),
    "swx_prof_enable": (c_int, [c_int]),
    "swx_debug_flags": (c_int, [c_int]),
    "swx_prof_collect": (c_int, [POINTER(ctypes.c_double), c_int]),
    "swx_graph_stats": (c_int, [c_void_p, POINTER(c_int64)]),
    "swx_test_gemm": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                              c_int, c_int, c_int, c_void_p]),
    "swx_test_gemm_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "swx_test_dec_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "swx_test_self_attn_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "swx_test_self_attn_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "swx_test_gelu_pair": (c_int, [c_void_p, c_void_p]),
    "swx_test_lane_xor": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "swx_test_layernorm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "swx_test_attention": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                   c_int, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """Load libswx.so.  Raises SwxError when it is absent (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if build_if_missing:
            from .build import build
            build(verbose=False)
        else:
            raise SwxError(f"{LIB_PATH} not found: build it with `python -m stable_ts_amd.build` "
                           f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the header and the library ever diverge
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> int:
    if code < 0:
        msg = load().swx_strerror(code).decode()
        raise SwxError(f"{what or 'libswx'} failed: {msg} ({code})")
    return code


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise SwxError("stable_ts_amd needs a ROCm GPU (MI355X / gfx950); no CPU path exists in this package")
