"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement of the pieces of ``whisper.timing`` (openai-whisper==20250625) that the
reference imports (whisper_compatibility.py:67): ``median_filter``, ``dtw`` (= ``dtw_cpu`` +
``backtrace``; the *CPU* tie-break is the parity target) and
``merge_punctuations``.  Reference call sites: timing.py:110,138,195,468.

Pinning: ``median_filter`` and ``dtw`` are checked against the verbatim third-party ports that
ARE installed here (transformers/models/whisper/generation_whisper.py:43-112) and against the
known-answer vector of SURVEY.md §8c in tests/test_oracle_pinning.py.  ``dtw`` also has a C
restatement (oracle/dtw.c) used for full-size cases and as the CPU baseline.
"""
import ctypes
import os
from typing import List

import numpy as np
import torch
import torch.nn.functional as F


def median_filter(x: torch.Tensor, filter_width: int):
    """Median filter of width ``filter_width`` along the last dimension (reflect padding)."""
    pad_width = filter_width // 2
    if x.shape[-1] <= pad_width:
        return x  # F.pad requires the padding width to be smaller than the input dimension
    if (ndim := x.ndim) <= 2:
        x = x[None, None, :]
    assert filter_width > 0 and filter_width % 2 == 1, "`filter_width` should be an odd number"
    x = F.pad(x, (filter_width // 2, filter_width // 2, 0, 0), mode="reflect")
    result = x.unfold(-1, filter_width, 1).sort()[0][..., filter_width // 2]
    if ndim <= 2:
        result = result[0, 0]
    return result


def backtrace(trace: np.ndarray):
    i = trace.shape[0] - 1
    j = trace.shape[1] - 1
    trace[0, :] = 2
    trace[:, 0] = 1
    result = []
    while i > 0 or j > 0:
        result.append((i - 1, j - 1))
        if trace[i, j] == 0:
            i -= 1
            j -= 1
        elif trace[i, j] == 1:
            i -= 1
        elif trace[i, j] == 2:
            j -= 1
        else:
            raise ValueError("Unexpected trace[i, j]")
    result = np.array(result)
    return result[::-1, :].T


def dtw_cpu_py(x: np.ndarray):
    """Pure-Python restatement of upstream ``dtw_cpu`` (numba-jitted there).  x: float64 [N, M]."""
    N, M = x.shape
    cost = np.ones((N + 1, M + 1), dtype=np.float32) * np.inf
    trace = -np.ones((N + 1, M + 1), dtype=np.float32)
    cost[0, 0] = 0
    for j in range(1, M + 1):
        for i in range(1, N + 1):
            c0 = cost[i - 1, j - 1]
            c1 = cost[i - 1, j]
            c2 = cost[i, j - 1]
            if c0 < c1 and c0 < c2:
                c, t = c0, 0
            elif c1 < c0 and c1 < c2:
                c, t = c1, 1
            else:
                c, t = c2, 2
            cost[i, j] = x[i - 1, j - 1] + c  # f64 + f32 -> f64, stored as f32
            trace[i, j] = t
    return backtrace(trace)


_LIB = None


def _load_c():
    global _LIB
    if _LIB is None:
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(here, "_build", "liboracle_dtw.so")
        if not os.path.exists(path):
            return None
        lib = ctypes.CDLL(path)
        lib.oracle_dtw.restype = ctypes.c_int
        lib.oracle_dtw.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def dtw_cpu(x: np.ndarray):
    """x: float64 [N, M] -> (text_indices, time_indices).  Uses oracle/dtw.c when built."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib = _load_c()
    if lib is None:
        return dtw_cpu_py(x)
    N, M = x.shape
    ti = np.empty(N + M, dtype=np.int32)
    tj = np.empty(N + M, dtype=np.int32)
    n = lib.oracle_dtw(x.ctypes.data, N, M, ti.ctypes.data, tj.ctypes.data)
    if n < 0:
        raise ValueError("Unexpected trace[i, j]")
    return np.stack([ti[:n].astype(np.int64), tj[:n].astype(np.int64)])


def dtw(x: torch.Tensor) -> np.ndarray:
    # upstream tries a Triton kernel for CUDA tensors and falls back to this; the CPU rule is the target
    return dtw_cpu(x.double().cpu().numpy())


def merge_punctuations(alignment: List, prepended: str, appended: str):
    # merge prepended punctuations
    i = len(alignment) - 2
    j = len(alignment) - 1
    while i >= 0:
        previous = alignment[i]
        following = alignment[j]
        if previous.word.startswith(" ") and previous.word.strip() in prepended:
            # prepend it to the following word
            following.word = previous.word + following.word
            following.tokens = previous.tokens + following.tokens
            previous.word = ""
            previous.tokens = []
        else:
            j = i
        i -= 1

    # merge appended punctuations
    i = 0
    j = 1
    while j < len(alignment):
        previous = alignment[i]
        following = alignment[j]
        if not previous.word.endswith(" ") and following.word in appended:
            # append it to the previous word
            previous.word = previous.word + following.word
            previous.tokens = previous.tokens + following.tokens
            following.word = ""
            following.tokens = []
        else:
            i = j
        j += 1
