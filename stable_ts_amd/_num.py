"""Scalar rounding to the millisecond with numpy's semantics, without numpy's scalar dispatch.

The reference rounds timestamps with the built-in ``round(x, 3)``.  Word times come out of numpy arithmetic, so ``x`` is a
``numpy.float64`` and the call lands in ``numpy.float64.__round__`` = ``rint(x * 1000) / 1000`` (round-half-even on the scaled
value) -- 2.2-2.8 us per call, several thousand calls per transcript (most of the time of building the result of a 10-minute
recording).  For a positive, finite ``numpy.float64`` below 1e12 the same three IEEE operations are done here on Python floats:
``round(float(x) * 1000.0)`` is round-half-even to an integer, exact below 2**53, and the division is the same correctly
rounded division; the result is wrapped in ``numpy.float64`` again.  Checked bit for bit against ``round(numpy.float64, 3)``
on 1.5 million values including every kind of near-tie (tests/test_result_api_cpu.py).  Everything else -- Python floats
(whose ``round`` is the correctly rounded decimal one, as in the reference), zeros, negatives, huge or non-finite values --
goes to the built-in as before.
"""
import numpy as np

_F64 = np.float64


def round3(x):
    if type(x) is _F64 and 0.0 < x < 1e12:
        return _F64(round(float(x) * 1000.0) / 1000.0)
    return round(x, 3)
