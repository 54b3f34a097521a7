"""ORACLE (test infrastructure): ctypes binding of oracle/segment_ref.c — the CPU restatement of
``get_segment`` (sylber/utils/segment_utils.py:72-131) and of the segment mean-pool
(sylber/model/sylber.py:133).  Only tests/, smoke() and bench.py's cpu_baseline import this."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libsegment_oracle.so")
    src = os.path.join(_HERE, "segment_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsegment_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        lib.sylber_oracle_get_segment.restype = ctypes.c_long
        lib.sylber_oracle_get_segment.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]
        lib.sylber_oracle_mean_pool.restype = None
        lib.sylber_oracle_mean_pool.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                                ctypes.c_void_p]
        lib.sylber_oracle_np_sum.restype = ctypes.c_float
        lib.sylber_oracle_np_sum.argtypes = [ctypes.c_void_p, ctypes.c_long]
        lib.sylber_oracle_powf_half.restype = ctypes.c_float
        lib.sylber_oracle_powf_half.argtypes = [ctypes.c_float]
        _LIB = lib
    return _LIB


def get_segment(states: np.ndarray, normthreshold: float, mergethreshold: float, norms=None) -> np.ndarray:
    """Same signature and return convention as the reference: int64 [n,2], or an empty
    float64 array of shape (0,) when there is no segment (``np.array([])``)."""
    states = np.ascontiguousarray(states, dtype=np.float32)
    T, d = states.shape
    out = np.zeros((max(T, 1), 2), dtype=np.int64)
    nptr = None
    if norms is not None:
        norms = np.ascontiguousarray(norms, dtype=np.float32)
        nptr = norms.ctypes.data
    n = _lib().sylber_oracle_get_segment(states.ctypes.data, T, d, np.float32(normthreshold),
                                         np.float32(mergethreshold), nptr, out.ctypes.data)
    if n == 0:
        return np.array([])
    return out[:n].copy()


def mean_pool(states: np.ndarray, segments: np.ndarray) -> np.ndarray:
    states = np.ascontiguousarray(states, dtype=np.float32)
    if len(segments) == 0:
        return np.array([])
    segments = np.ascontiguousarray(segments, dtype=np.int64)
    out = np.empty((len(segments), states.shape[1]), dtype=np.float32)
    _lib().sylber_oracle_mean_pool(states.ctypes.data, states.shape[1], segments.ctypes.data, len(segments),
                                   out.ctypes.data)
    return out


def np_sum(a: np.ndarray) -> np.float32:
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.float32(_lib().sylber_oracle_np_sum(a.ctypes.data, a.size))


def powf_half(v: float) -> np.float32:
    return np.float32(_lib().sylber_oracle_powf_half(np.float32(v)))
