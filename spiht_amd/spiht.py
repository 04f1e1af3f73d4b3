"""Counterpart of the reference's PyO3 extension module `spiht.spiht` (/root/reference/src/lib.rs:58-65).

Same names, argument meaning and return types:
    encode(x, ll_h, ll_w, max_bits) -> (bytes, int)                       lib.rs:24-32
    decode(data_u8, n, c, h, w, ll_h, ll_w) -> ndarray[int32, (c,h,w)]    lib.rs:35-42
Both run on the GPU through libspiht_hip.so; there is no CPU path.
`decode_with_metadata` (lib.rs:47-56) is not part of the accelerated hot path yet.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import PanicException, SpihtHipError  # noqa: F401  (re-exported)

_U64_MAX = 2 ** 64 - 1


def _as_usize(v, name):
    # PyO3 extracts `usize`: ints only, 0 <= v < 2^64, else TypeError / OverflowError
    if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
        raise TypeError("argument '%s': '%s' object cannot be interpreted as an integer" % (name, type(v).__name__))
    v = int(v)
    if v < 0:
        raise OverflowError("can't convert negative int to unsigned")
    if v > _U64_MAX:
        raise OverflowError("Python int too large to convert to C long")
    return v


def encode(x, ll_h, ll_w, max_bits):
    """Encode DWT coefficients into bytes.

    x: numpy ndarray, dtype int32, ndim 3 (c,h,w), any strides (PyReadonlyArray3<i32>, lib.rs:26).
    Returns (bytes, max_n).  The stream holds exactly min(max_bits, total) bits, packed LSB-first.
    """
    if not isinstance(x, np.ndarray):
        raise TypeError("argument 'x': '%s' object cannot be converted to 'PyArray<T, D>'" % type(x).__name__)
    if x.dtype != np.int32 or x.ndim != 3:
        raise TypeError("argument 'x': type mismatch:\n from=%s, to=int32\n dimensions: from=%d, to=3"
                        % (x.dtype, x.ndim))
    ll_h = _as_usize(ll_h, "ll_h")
    ll_w = _as_usize(ll_w, "ll_w")
    max_bits = _as_usize(max_bits, "max_bits")
    ctx = _lib.default_context()
    L = _lib.lib()
    c, h, w = x.shape
    if x.size == 0:
        raise PanicException("called `Option::unwrap()` on a `None` value")
    es = x.itemsize
    max_abs = int(np.abs(x.astype(np.int64)).max())
    bound = C.c_uint64()
    _lib.check(L.spiht_encode_bound(c, h, w, ll_h, ll_w, min(max_abs, 0xFFFFFFFF), max_bits, C.byref(bound)))
    out = np.empty(max(int(bound.value), 4), dtype=np.uint8)
    nbits = C.c_uint64()
    max_n = C.c_uint8()
    st = L.spiht_encode_i32(ctx.handle, C.c_void_p(x.ctypes.data), c, h, w, x.strides[0] // es, x.strides[1] // es,
                            x.strides[2] // es, ll_h, ll_w, max_bits, C.c_void_p(out.ctypes.data), out.size,
                            C.byref(nbits), C.byref(max_n))
    _lib.check(st)
    nbytes = (nbits.value + 7) // 8
    return out[:nbytes].tobytes(), int(max_n.value)


def _as_u8_vec(data):
    # PyO3 `Vec<u8>`: bytes, bytearray, or any sequence of ints in 0..=255 (a str is refused)
    if isinstance(data, str):
        raise TypeError("argument 'data_u8': Can't extract `str` to `Vec`")
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    if isinstance(data, np.ndarray) and data.dtype == np.uint8:
        return np.ascontiguousarray(data).reshape(-1)
    vals = list(data)
    for v in vals:
        if isinstance(v, bool) or not isinstance(v, (int, np.integer)):
            raise TypeError("argument 'data_u8': '%s' object cannot be interpreted as an integer" % type(v).__name__)
        if not 0 <= int(v) <= 255:
            raise OverflowError("out of range integral type conversion attempted")
    return np.asarray(vals, dtype=np.uint8)


def decode(data_u8, n, c, h, w, ll_h, ll_w):
    """Decode DWT coefficients from bytes.  h, w are the coefficient-array dims.  All 8*len(data) bits are data
    (lib.rs:38).  Returns a new C-contiguous int32 array (c,h,w)."""
    buf = _as_u8_vec(data_u8)
    n = _as_usize(n, "n")
    if n > 255:
        raise OverflowError("out of range integral type conversion attempted")
    c, h, w = _as_usize(c, "c"), _as_usize(h, "h"), _as_usize(w, "w")
    ll_h, ll_w = _as_usize(ll_h, "ll_h"), _as_usize(ll_w, "ll_w")
    if ll_h <= 1 or ll_w <= 1:
        raise PanicException("assertion failed: ll_h > 1")
    if c == 0 or h == 0 or w == 0:
        return np.zeros((c, h, w), dtype=np.int32)
    ctx = _lib.default_context()
    L = _lib.lib()
    out = np.empty((c, h, w), dtype=np.int32)
    st = L.spiht_decode_i32(ctx.handle, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, n, c, h, w, ll_h, ll_w,
                            C.c_void_p(out.ctypes.data))
    _lib.check(st)
    return out


def decode_with_metadata(*args, **kwargs):
    raise NotImplementedError(
        "decode_with_metadata (src/lib.rs:47-56) is outside the accelerated hot path of this build (SURVEY.md 8 f-1)")
