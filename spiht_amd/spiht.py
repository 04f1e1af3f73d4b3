"""Counterpart of the reference's PyO3 extension module `spiht.spiht` (/root/reference/src/lib.rs:58-65).

Same names, argument meaning and return types:
    encode(x, ll_h, ll_w, max_bits) -> (bytes, int)                       lib.rs:24-32
    decode(data_u8, n, c, h, w, ll_h, ll_w) -> ndarray[int32, (c,h,w)]    lib.rs:35-42
    decode_with_metadata(data_u8, n, c, h, w, ll_h, ll_w, top_slice, other_slices)
        -> (ndarray[int32, (c,h,w)], ndarray[int32, (8*len+1, 8)])        lib.rs:47-56
All three run on the GPU through libspiht_hip.so; there is no CPU path.
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
    # sizes the output buffer only (the device computes max|x| itself, pyramid.hip: k_absmax): two reductions over the
    # caller's array in place -- no widened copy
    max_abs = max(-int(x.min()), int(x.max()))
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
    out = _lib.result_array((c, h, w), np.int32)
    st = L.spiht_decode_i32(ctx.handle, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, n, c, h, w, ll_h, ll_w,
                            C.c_void_p(out.ctypes.data))
    _lib.check(st)
    return out


def decode_budgets(data_u8, n, c, h, w, ll_h, ll_w, bit_budgets):
    """Progressive decoding from ONE walk of the stream: int32 (K, c, h, w) with
    out[k] == decode(<the first bit_budgets[k] bits of data>, n, c, h, w, ll_h, ll_w)  (8 * L for the byte prefix
    data[:L]; a budget past the end decodes the whole stream).  bit_budgets must be ascending.  New in this library --
    the reference decodes one prefix per call (make_gif.py:46-61), i.e. K walks for K frames."""
    buf = _as_u8_vec(data_u8)
    n = _as_usize(n, "n")
    if n > 255:
        raise OverflowError("out of range integral type conversion attempted")
    c, h, w = _as_usize(c, "c"), _as_usize(h, "h"), _as_usize(w, "w")
    ll_h, ll_w = _as_usize(ll_h, "ll_h"), _as_usize(ll_w, "ll_w")
    bud = np.ascontiguousarray([_as_usize(b, "bit_budgets") for b in bit_budgets], dtype=np.uint64)
    if bud.size == 0:
        return np.zeros((0, c, h, w), dtype=np.int32)
    if ll_h <= 1 or ll_w <= 1:
        raise PanicException("assertion failed: ll_h > 1")
    if c == 0 or h == 0 or w == 0:
        return np.zeros((bud.size, c, h, w), dtype=np.int32)
    ctx = _lib.default_context()
    out = np.empty((bud.size, c, h, w), dtype=np.int32)
    _lib.check(_lib.lib().spiht_decode_budgets_i32(
        ctx.handle, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, n, c, h, w, ll_h, ll_w,
        C.c_void_p(bud.ctypes.data), bud.size, C.c_void_p(out.ctypes.data)))
    return out


def _as_pairs(v, name, count):
    # PyO3 `Vec<(usize, usize)>`
    if isinstance(v, str):
        raise TypeError("argument '%s': Can't extract `str` to `Vec`" % name)
    out = []
    for p in list(v):
        p = tuple(p)
        if len(p) != 2:
            raise ValueError("argument '%s': expected tuple of length 2, but got tuple of length %d" % (name, len(p)))
        out.append((_as_usize(p[0], name), _as_usize(p[1], name)))
    if count is not None and len(out) < count:
        raise PanicException("index out of bounds: the len is %d but the index is %d" % (len(out), len(out)))
    return out


def decode_with_metadata(data_u8, n, c, h, w, ll_h, ll_w, top_slice, other_slices):
    """Decode DWT coefficients from bytes and also return the intermediate metadata of the SPIHT algorithm
    (src/lib.rs:47-56 -> encoder_decoder.rs:631-841).

    top_slice: [(start_i, end_i), (start_j, end_j)] of the LL block; other_slices: per detail level, coarsest
    first, three filters (the reference wrapper passes da, ad, dd), each [(start_i, end_i), (start_j, end_j)].
    Returns (rec int32 (c,h,w), metadata int32 (8*len(data)+1, 8)); metadata row t, for the operation that
    reads stream bit t: [action 0..6, local_h, local_w, channel, filter, depth, n, current coefficient value]
    (doc comment encoder_decoder.rs:616-630)."""
    buf = _as_u8_vec(data_u8)
    n = _as_usize(n, "n")
    if n > 255:
        raise OverflowError("out of range integral type conversion attempted")
    c, h, w = _as_usize(c, "c"), _as_usize(h, "h"), _as_usize(w, "w")
    ll_h, ll_w = _as_usize(ll_h, "ll_h"), _as_usize(ll_w, "ll_w")
    top = _as_pairs(top_slice, "top_slice", 2)                      # Slices::from_vec indexes [0] and [1] (:518-523)
    levels = []
    for lv in list(other_slices):
        fl = [_as_pairs(f, "other_slices", 2) for f in list(lv)]
        if len(fl) < 3:
            raise PanicException("index out of bounds: the len is %d but the index is %d" % (len(fl), len(fl)))
        levels.append(fl)
    if ll_h <= 1 or ll_w <= 1:
        raise PanicException("assertion failed: ll_h > 1")
    rows = 8 * buf.size + 1
    if c == 0:
        return np.zeros((c, h, w), dtype=np.int32), np.zeros((rows, 8), dtype=np.int32)
    if h == 0 or w == 0:
        raise PanicException("ndarray: index out of bounds")  # the first metadata row reads rec_arr[(0,0,0)] (:683)
    topv = np.array([top[0][0], top[0][1], top[1][0], top[1][1]], dtype=np.int64)
    oth = np.array([[[f[0][0], f[0][1], f[1][0], f[1][1]] for f in lv[:3]] for lv in levels], dtype=np.int64).reshape(-1)
    oth = np.ascontiguousarray(oth if oth.size else np.zeros(1, dtype=np.int64))
    ctx = _lib.default_context()
    L = _lib.lib()
    out = np.empty((c, h, w), dtype=np.int32)
    meta = np.empty((rows, 8), dtype=np.int32)
    st = L.spiht_decode_with_metadata_i32(ctx.handle, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, n, c, h, w,
                                          ll_h, ll_w, C.c_void_p(topv.ctypes.data), C.c_void_p(oth.ctypes.data),
                                          len(levels), C.c_void_p(out.ctypes.data), C.c_void_p(meta.ctypes.data))
    _lib.check(st)
    return out, meta
