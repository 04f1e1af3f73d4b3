"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by anything under spiht_amd/.

The functions mirror the reference boundary so that tests read like the
reference's own:
    encode(x, ll_h, ll_w, max_bits) -> (bytes, max_n)      # src/lib.rs:24-32
    decode(data, n, c, h, w, ll_h, ll_w) -> int32[c,h,w]   # src/lib.rs:35-42
plus the float64 DWT front/back halves of spiht/spiht_wrapper.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

RULE_RUST = 0
RULE_PY = 1

MODES = {"reflect": 0, "symmetric": 1, "periodic": 2, "zero": 3, "constant": 4, "smooth": 5, "antisymmetric": 6, "antireflect": 7, "periodization": 8}


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("spiht_oracle.c", "dwt_oracle.c", "color_oracle.c", "wavelets_table.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        i64, u64, u8p = C.c_int64, C.c_uint64, C.POINTER(C.c_uint8)
        L.orc_encode.argtypes = [C.c_void_p, i64, i64, i64, i64, i64, i64, i64, i64, u64, C.c_int,
                                 C.POINTER(u8p), C.POINTER(u64), u8p]
        L.orc_encode.restype = C.c_int
        L.orc_encode_bits.argtypes = L.orc_encode.argtypes
        L.orc_encode_bits.restype = C.c_int
        L.orc_decode.argtypes = [C.c_void_p, u64, C.c_uint8, i64, i64, i64, i64, i64, C.c_int, C.c_void_p]
        L.orc_decode.restype = C.c_int
        L.orc_decode_bits.argtypes = L.orc_decode.argtypes
        L.orc_decode_bits.restype = C.c_int
        L.orc_decode_with_metadata.argtypes = [C.c_void_p, u64, C.c_uint8, i64, i64, i64, i64, i64, C.c_void_p,
                                               C.c_void_p, i64, C.c_void_p, C.c_void_p]
        L.orc_decode_with_metadata.restype = C.c_int
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_set_bit.argtypes = [C.c_int32, C.c_uint8, C.c_int]
        L.orc_set_bit.restype = C.c_int32
        L.orc_is_bit_set.argtypes = [C.c_int32, C.c_uint8]
        L.orc_is_element_sig.argtypes = [C.c_int32, C.c_uint8]
        L.orc_start_plane.argtypes = [C.c_int32, C.c_int]
        L.orc_start_plane.restype = C.c_uint8
        L.orc_get_offspring.argtypes = [i64] * 6 + [C.c_void_p]
        L.orc_wavelet_id.argtypes = [C.c_char_p]
        L.orc_wavelet_len.argtypes = [C.c_int]
        L.orc_wavelet_filters.argtypes = [C.c_int] + [C.c_void_p] * 4
        L.orc_dwt_max_level.argtypes = [i64, C.c_int]
        L.orc_resolve_level.argtypes = [i64, i64, C.c_int, C.c_int]
        L.orc_geometry.argtypes = [i64, i64, C.c_int, C.c_int] + [C.c_void_p] * 6
        L.orc_wavedec2_array.argtypes = [C.c_void_p, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_quantize.argtypes = [C.c_void_p, i64, i64, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_quantize.restype = None
        L.orc_dequantize.argtypes = [C.c_void_p, i64, i64, C.c_void_p, C.c_double, C.c_void_p]
        L.orc_dequantize.restype = None
        L.orc_waverec2_shape.argtypes = [i64, i64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_waverec2_shape.restype = None
        L.orc_waverec2_array.argtypes = [C.c_void_p, i64, i64, i64, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


class OraclePanic(Exception):
    """Stands in for pyo3_runtime.PanicException (assert!(ll_h > 1) etc.)."""


def _check(rc):
    if rc == 1:
        raise OraclePanic("assertion failed: ll_h > 1 && ll_w > 1")
    if rc == 2:
        raise OraclePanic("empty array")
    if rc:
        raise MemoryError("oracle rc=%d" % rc)


def encode(x, ll_h, ll_w, max_bits, rule=RULE_RUST):
    """src/lib.rs:24-32.  x: int32 ndarray [c,h,w], any strides."""
    if not isinstance(x, np.ndarray) or x.dtype != np.int32 or x.ndim != 3:
        raise TypeError("x must be a 3-D int32 ndarray")
    c, h, w = x.shape
    es = x.itemsize
    out = C.POINTER(C.c_uint8)()
    nbits = C.c_uint64()
    max_n = C.c_uint8()
    rc = lib().orc_encode(x.ctypes.data, c, h, w, x.strides[0] // es, x.strides[1] // es,
                          x.strides[2] // es, ll_h, ll_w, C.c_uint64(min(int(max_bits), 2**64 - 1)),
                          rule, C.byref(out), C.byref(nbits), C.byref(max_n))
    _check(rc)
    nbytes = (nbits.value + 7) // 8
    data = C.string_at(out, nbytes)
    lib().orc_free(out)
    return data, int(max_n.value)


def encode_nbits(x, ll_h, ll_w, max_bits, rule=RULE_RUST):
    """As encode() but also returns the exact bit count."""
    c, h, w = x.shape
    es = x.itemsize
    out = C.POINTER(C.c_uint8)()
    nbits = C.c_uint64()
    max_n = C.c_uint8()
    rc = lib().orc_encode(x.ctypes.data, c, h, w, x.strides[0] // es, x.strides[1] // es,
                          x.strides[2] // es, ll_h, ll_w, C.c_uint64(min(int(max_bits), 2**64 - 1)),
                          rule, C.byref(out), C.byref(nbits), C.byref(max_n))
    _check(rc)
    data = C.string_at(out, (nbits.value + 7) // 8)
    lib().orc_free(out)
    return data, int(max_n.value), int(nbits.value)


def decode(data, n, c, h, w, ll_h, ll_w, rule=RULE_RUST):
    """src/lib.rs:35-42.  All 8*len(data) bits are data."""
    data = bytes(bytearray(data))
    out = np.empty((c, h, w), dtype=np.int32)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    rc = lib().orc_decode(buf, len(data), n, c, h, w, ll_h, ll_w, rule, out.ctypes.data)
    _check(rc)
    return out


def decode_bits(bits, n, c, h, w, ll_h, ll_w, rule=RULE_RUST):
    """encoder_decoder.rs:307-454 on an exact bit list (no byte padding)."""
    bits = np.ascontiguousarray(np.asarray(bits, dtype=np.uint8))
    out = np.empty((c, h, w), dtype=np.int32)
    rc = lib().orc_decode_bits(bits.ctypes.data, bits.size, n, c, h, w, ll_h, ll_w, rule,
                               out.ctypes.data)
    _check(rc)
    return out


def flatten_slices(top_slice, other_slices):
    """lib.rs:49 argument structures -> (int64[4], int64[level*3*4], level) in Slices::from_vec order (:482-527)"""
    top = np.array([top_slice[0][0], top_slice[0][1], top_slice[1][0], top_slice[1][1]], dtype=np.int64)
    other = np.array([[[f[0][0], f[0][1], f[1][0], f[1][1]] for f in lv] for lv in other_slices],
                     dtype=np.int64).reshape(-1)
    return top, np.ascontiguousarray(other if other.size else np.zeros(1, np.int64)), len(other_slices)


def decode_with_metadata(data, n, c, h, w, ll_h, ll_w, top_slice, other_slices):
    """src/lib.rs:47-56 -> (rec int32[c,h,w], metadata int32[8*len(data)+1, 8]).  Metadata rows: parity unpinned."""
    data = bytes(bytearray(data))
    top, other, level = flatten_slices(top_slice, other_slices)
    out = np.empty((c, h, w), dtype=np.int32)
    meta = np.empty((8 * len(data) + 1, 8), dtype=np.int32)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    rc = lib().orc_decode_with_metadata(buf, len(data), n, c, h, w, ll_h, ll_w, top.ctypes.data, other.ctypes.data,
                                        level, out.ctypes.data, meta.ctypes.data)
    if rc == 4:
        raise OraclePanic("index out of bounds in get_local_position")
    _check(rc)
    return out, meta


def bytes_to_bits(b):
    """spiht/utils.py:6-9"""
    return np.unpackbits(np.frombuffer(b, np.uint8), bitorder="little")


def set_bit(x, n, bit):
    return lib().orc_set_bit(x, n, int(bool(bit)))


def is_bit_set(x, n):
    return bool(lib().orc_is_bit_set(x, n))


def is_element_sig(x, n):
    return bool(lib().orc_is_element_sig(x, n))


def start_plane(maxabs, rule=RULE_RUST):
    return int(lib().orc_start_plane(maxabs, rule))


def set_codes(x, ll_h, ll_w):
    """Significance of every node's D and L sets by the reference's recursion (encoder_decoder.rs:78-121, :228-237):
    -> (dcode, lcode, has_offspring) uint8 [c,h,w]; a set is significant at plane n iff its code > n."""
    x = np.ascontiguousarray(x, dtype=np.int32)
    c, h, w = x.shape
    d, l, has = (np.zeros((c, h, w), dtype=np.uint8) for _ in range(3))
    L = lib()
    L.orc_set_codes.argtypes = [C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p] * 3
    _check(L.orc_set_codes(x.ctypes.data, c, h, w, ll_h, ll_w, d.ctypes.data, l.ctypes.data, has.ctypes.data))
    return d, l, has.astype(bool)


def color3(img, A, M, p):
    """the colour model change w = M * spow(A * u, p) per pixel with the C library's pow() (color_oracle.c: independent of
    the product's power function): img float64 [3,...] -> same shape"""
    img = np.ascontiguousarray(img, dtype=np.float64)
    out = np.empty_like(img)
    A, M = np.ascontiguousarray(A, np.float64), np.ascontiguousarray(M, np.float64)
    L = lib()
    L.orc_color3.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double]
    L.orc_color3.restype = None
    L.orc_color3(img.ctypes.data, out.ctypes.data, img.size // 3, A.ctypes.data, M.ctypes.data, float(p))
    return out


def get_offspring(i, j, h, w, ll_h, ll_w):
    o = np.zeros((4, 2), dtype=np.int64)
    has = lib().orc_get_offspring(i, j, h, w, ll_h, ll_w, o.ctypes.data)
    return [tuple(int(v) for v in r) for r in o] if has else None


# ---------------- DWT front/back halves ----------------

def wavelet_id(name):
    wid = lib().orc_wavelet_id(name.encode())
    if wid < 0:
        raise ValueError("unknown wavelet %r" % name)
    return wid


def wavelet_filters(name):
    wid = wavelet_id(name)
    F = lib().orc_wavelet_len(wid)
    arrs = [np.zeros(F) for _ in range(4)]
    lib().orc_wavelet_filters(wid, *[a.ctypes.data for a in arrs])
    return arrs


def geometry(H, W, wavelet, level, mode="reflect"):
    """-> dict(level, hs, ws, ll_h, ll_w, enc_h, enc_w)   (wrapper:92-139); the mode matters when it is periodization"""
    wid = wavelet_id(wavelet)
    F = lib().orc_wavelet_len(wid)
    hs = np.zeros(64, dtype=np.int64)
    ws = np.zeros(64, dtype=np.int64)
    v = [C.c_int64() for _ in range(4)]
    lib().orc_geometry_mode.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 6
    L = lib().orc_geometry_mode(H, W, F, MODES[mode], -1 if level is None else int(level), hs.ctypes.data, ws.ctypes.data,
                                *[C.byref(t) for t in v])
    return dict(level=L, hs=hs[:L + 1].tolist(), ws=ws[:L + 1].tolist(), ll_h=v[0].value, ll_w=v[1].value,
                enc_h=v[2].value, enc_w=v[3].value)


def wavedec2_array(img, wavelet, mode, level):
    """pywt.wavedec2 + coeffs_to_array (wrapper:163-165) -> float64 [c,enc_h,enc_w]"""
    img = np.ascontiguousarray(img, dtype=np.float64)
    c, H, W = img.shape
    g = geometry(H, W, wavelet, level, mode)
    arr = np.empty((c, g["enc_h"], g["enc_w"]), dtype=np.float64)
    rc = lib().orc_wavedec2_array(img.ctypes.data, c, H, W, wavelet_id(wavelet), MODES[mode],
                                  -1 if level is None else int(level), arr.ctypes.data)
    if rc:
        raise RuntimeError("orc_wavedec2_array rc=%d" % rc)
    return arr, g


def wavedec2_array_f32(img, wavelet, mode, level):
    """the float32 transform PyWavelets runs on float32 / float16 input -> float32 [c,enc_h,enc_w]"""
    img = np.ascontiguousarray(img, dtype=np.float32)
    c, H, W = img.shape
    g = geometry(H, W, wavelet, level, mode)
    arr = np.empty((c, g["enc_h"], g["enc_w"]), dtype=np.float32)
    L = lib()
    L.orc_wavedec2_array_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = L.orc_wavedec2_array_f32(img.ctypes.data, c, H, W, wavelet_id(wavelet), MODES[mode],
                                  -1 if level is None else int(level), arr.ctypes.data)
    if rc:
        raise RuntimeError("orc_wavedec2_array_f32 rc=%d" % rc)
    return arr, g


def quantize_f32(arr, q, mults=None):
    """wrapper:167-172 on the float32 array (float32 product without channel scales, float64 with them)"""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    c = arr.shape[0]
    out = np.empty(arr.shape, dtype=np.int32)
    m = None if mults is None else np.ascontiguousarray(mults, dtype=np.float64)
    L = lib()
    L.orc_quantize_f32.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_double, C.c_void_p]
    L.orc_quantize_f32.restype = None
    L.orc_quantize_f32(arr.ctypes.data, c, arr.size // c, None if m is None else m.ctypes.data, float(q), out.ctypes.data)
    return out


def quantize(arr, q, mults=None):
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    c = arr.shape[0]
    out = np.empty(arr.shape, dtype=np.int32)
    m = None if mults is None else np.ascontiguousarray(mults, dtype=np.float64)
    lib().orc_quantize(arr.ctypes.data, c, arr.size // c, None if m is None else m.ctypes.data, float(q),
                       out.ctypes.data)
    return out


def dequantize(rec, q, mults=None):
    rec = np.ascontiguousarray(rec, dtype=np.int32)
    c = rec.shape[0]
    out = np.empty(rec.shape, dtype=np.float64)
    m = None if mults is None else np.ascontiguousarray(mults, dtype=np.float64)
    lib().orc_dequantize(rec.ctypes.data, c, rec.size // c, None if m is None else m.ctypes.data, float(q),
                         out.ctypes.data)
    return out


def waverec2_array(arr, H, W, wavelet, level, mode="reflect"):
    """pywt.array_to_coeffs + waverec2 (wrapper:275-276) -> float64 [c,H',W']; the mode matters when it is periodization"""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    c = arr.shape[0]
    wid = wavelet_id(wavelet)
    F = lib().orc_wavelet_len(wid)
    lv = -1 if level is None else int(level)
    Ho, Wo = C.c_int64(), C.c_int64()
    L = lib()
    L.orc_waverec2_shape_mode.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_waverec2_shape_mode.restype = None
    L.orc_waverec2_array_mode.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_waverec2_shape_mode(H, W, F, MODES[mode], lv, C.byref(Ho), C.byref(Wo))
    out = np.empty((c, Ho.value, Wo.value), dtype=np.float64)
    rc = L.orc_waverec2_array_mode(arr.ctypes.data, c, H, W, wid, MODES[mode], lv, out.ctypes.data)
    if rc:
        raise RuntimeError("orc_waverec2_array rc=%d" % rc)
    return out


def encode_image(image, wavelet="bior2.2", mode="reflect", level=None, q=50.0, mults=None, max_bits=None,
                 rule=RULE_RUST):
    """CPU restatement of spiht_wrapper.encode_image (wrapper:142-189), no colour conversion.
    -> (bytes, max_n, geometry)"""
    if np.asarray(image).dtype in (np.float32, np.float16):  # PyWavelets' single-precision path (_check_dtype)
        arr, g = wavedec2_array_f32(image, wavelet, mode, level)
        coeffs = quantize_f32(arr, q, mults)
    else:
        arr, g = wavedec2_array(image, wavelet, mode, level)
        coeffs = quantize(arr, q, mults)
    mb = 99999999999999999 if max_bits is None else max_bits
    data, max_n = encode(coeffs, g["ll_h"], g["ll_w"], mb, rule)
    return data, max_n, g


def decode_image(data, max_n, c, H, W, wavelet="bior2.2", level=None, q=50.0, mults=None, rule=RULE_RUST, mode="reflect"):
    """CPU restatement of spiht_wrapper.decode_image (wrapper:192-281), no colour conversion.  (The extension mode matters
    only when it is periodization.)"""
    g = geometry(H, W, wavelet, level, mode)
    rec = decode(data, max_n, c, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], rule)
    return waverec2_array(dequantize(rec, q, mults), H, W, wavelet, level, mode)
