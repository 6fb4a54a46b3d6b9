"""numpy emulation of the two operand splits the contraction kernels use (``ihgnn_amd/csrc/split_common.hpp``) - test infrastructure.

* three bf16 terms (``split_pair``): ``hi`` = top 16 bits of ``x`` (truncation), ``mid`` = top 16 bits of ``x - hi``, ``lo`` = ``x - hi - mid``: exact, six of the nine
  partial products are accumulated (``hi hi + hi mid + mid hi + hi lo + mid mid + lo hi``);
* two fp16 terms (``split_pair_h2``): the row (or weight column) is first multiplied by the power of two that brings its largest magnitude to [2^13, 2^14)
  (``scale_up_for``), then ``hi = fp16(x)``, ``lo = fp16(x - hi)``, both round-to-nearest-even; three partial products are accumulated (``hi lo + lo hi + hi hi``).

``tests/test_host_logic.py`` holds the schemes' error bounds on these emulations (CPU); ``tests/test_gpu_parity.py`` takes its adversarial operands from
``worst_two_fp16_significand()`` / ``worst_three_bf16_significand()``.
"""
import numpy as np


def scale_up_for(m):
    """``2^(13 - floor(log2 m))`` for magnitudes ``m`` (array), the exponent clamped like the device function's."""
    m = np.asarray(m, np.float32)
    e = ((m.view(np.uint32) >> 23) & 0xff).astype(np.int64)
    e = np.clip(e, 27, 227)
    return np.ldexp(np.float64(1.0), (13 - (e - 127)).astype(np.int64))


def split_two_fp16(x_scaled):
    """``(hi, lo)`` as float64 arrays for fp32 values that already carry their row's scale."""
    x = np.asarray(x_scaled, np.float32)
    with np.errstate(over='ignore'):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float32)).astype(np.float16)          # the difference is exact in fp32
    return hi.astype(np.float64), lo.astype(np.float64)


def top16(x):
    return (np.asarray(x, np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def split_three_bf16(x):
    x = np.asarray(x, np.float32)
    hi = top16(x)
    r = x - hi
    mid = top16(r)
    lo = r - mid
    return hi.astype(np.float64), mid.astype(np.float64), lo.astype(np.float64)


def dot_two_fp16(a, b):
    """Rows of ``a`` [n, k] against columns given as rows of ``b`` [m, k] with the two-fp16 scheme: each row of ``a`` and each row of ``b`` under its own power of
    two; the partial products summed in float64 (the kernels' fp32 accumulation is left out: this isolates what the SPLIT loses).  Returns [n, m] float64."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    sa, sb = scale_up_for(np.abs(a).max(1)), scale_up_for(np.abs(b).max(1))
    ah, al = split_two_fp16((a.astype(np.float64) * sa[:, None]).astype(np.float32))
    bh, bl = split_two_fp16((b.astype(np.float64) * sb[:, None]).astype(np.float32))
    acc = ah @ bl.T + al @ bh.T + ah @ bh.T
    return acc / sa[:, None] / sb[None, :]


def dot_three_bf16(a, b):
    ah, am, al = split_three_bf16(a)
    bh, bm, bl = split_three_bf16(b)
    return ah @ bl.T + al @ bh.T + am @ bm.T + ah @ bm.T + am @ bh.T + ah @ bh.T


def all_significands():
    """Every fp32 value in [1, 2): 2^23 of them."""
    return (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3f800000)).view(np.float32)


def worst_two_fp16_significand():
    """The fp32 significand in [1, 2) whose square loses most under the two-fp16 split with every omitted term of ONE sign: with ``x = hi + lo + r`` the product of two such
    values omits ``lo lo + 2 r x`` (to first order).  Searched over all 2^23 significands (a power-of-two scale to [2^13, 2^14) does not change the pattern).
    Returns ``(x, relative_loss)`` - analytically ``x = 1 + 4093 * 2^-23``: ``hi = 1``, ``lo = 4092 * 2^-23`` (a tie rounded to even), ``r = 2^-23``, loss ~ 2^-21."""
    s = all_significands()
    hi, lo = split_two_fp16(s * np.float32(8192.0))
    x = s.astype(np.float64) * 8192.0
    r = x - hi - lo
    same_sign = (lo > 0) & (r >= 0)
    loss = np.where(same_sign, lo * lo + 2 * r * x, 0.0) / (x * x)
    k = int(np.argmax(loss))
    return float(s[k]), float(loss[k])


def worst_three_bf16_significand():
    """The same question for the truncating three-bf16 split (omitted: ``mid lo + lo mid + lo lo``): every low bit set, ``1 + (2^16 - 1) 2^-23``."""
    s = all_significands()[:: 1 << 7]                                 # mid's own low bits do not matter beyond the first few: a 2^16 subset holds the maximum
    s = np.concatenate([s, np.array([1.0 + (2.0 ** 16 - 1) * 2.0 ** -23], np.float32)])
    hi, mid, lo = split_three_bf16(s)
    loss = (2 * mid * lo + lo * lo) / (s.astype(np.float64) ** 2)
    k = int(np.argmax(loss))
    return float(s[k]), float(loss[k])
