"""Counterpart of the reference's public Python API (/root/reference/spiht/spiht_wrapper.py).

Same names, defaults, field order and exceptions:
    SpihtSettings, EncodingResult, ENCODER_DECODER_VERSION, encode_image, decode_image,
    decode_rec_array, decode_from_rec_arr, get_slices_and_h_w, quantize, dequantize
What differs is where the arithmetic runs: the multilevel DWT (pywt.wavedec2/waverec2 in the reference),
the Mallat packing, the per-channel scaling, the quantisation and the SPIHT coder all execute as HIP kernels
on the MI355X through libspiht_hip.so; the pixel array is uploaded once and only the bitstream comes back.
"""
import ctypes as C
from dataclasses import asdict, dataclass
from typing import Any, List, Optional, Tuple, Union

import numpy as np

from . import _lib, color_models
from . import spiht as spiht_rs


def quantize(arr, q_scale=10.):
    """wrapper:9-11 (host helper kept for API parity; the hot path quantises inside the DWT kernel)"""
    arr = arr * q_scale
    return arr.astype(np.int32)


def dequantize(arr, q_scale=10.):
    """wrapper:13-14"""
    return arr / q_scale


ENCODER_DECODER_VERSION = "0.0.2"


@dataclass
class SpihtSettings:
    """Parameters that are not particular to a single image (wrapper:20-63): field order is API
    (demonstrate.py:23-29 passes them positionally)."""
    wavelet: str = 'bior2.2'
    quantization_scale: float = 50.0
    mode: str = 'reflect'
    color_model: Optional[str] = None
    per_channel_quant_scales: Optional[List[float]] = None


@dataclass
class EncodingResult:
    """wrapper:65-89.  h, w, c are IMAGE dims; level may be None."""
    encoded_bytes: bytes
    h: int
    w: int
    c: int
    max_n: int
    level: Optional[int]
    _encoding_version: str = ENCODER_DECODER_VERSION

    def to_dict(self):
        return {f"encoding_result_{k}": v for k, v in asdict(self).items()}

    @staticmethod
    def from_dict(d):
        d = {k.removeprefix('encoding_result_'): v for k, v in d.items() if k.startswith('encoding_result_')}
        return EncodingResult(**d)


def _wavelet_mode_ids(spiht_settings):
    L = _lib.lib()
    wid = L.spiht_wavelet_id(str(spiht_settings.wavelet).encode())
    if wid < 0:
        # pywt.Wavelet(name) raises this ValueError for a name it does not know (spiht_wrapper.py:163 reaches it through
        # wavedec2); all 106 discrete wavelets of PyWavelets are known here (csrc/wavelets.h)
        raise ValueError("Unknown wavelet name '%s', check wavelist() for the list of available builtin wavelets." % spiht_settings.wavelet)
    mid = L.spiht_mode_id(str(spiht_settings.mode).encode())
    if mid < 0:
        # pywt.Modes.from_object raises ValueError("Unknown mode name '...'.") (reached from spiht_wrapper.py:163)
        raise ValueError("Unknown mode name '%s'." % spiht_settings.mode)
    return wid, mid


def _geometry(h, w, wid, level, mid=0):
    """sizes of the packed coefficient array; mid: the extension mode's id (periodization has its own length rule)"""
    L = _lib.lib()
    lv = C.c_int()
    v = [C.c_int64() for _ in range(6)]
    if level is not None and level < 0:
        raise ValueError("Level value of %d is too low . Minimum level is 0." % level)
    _lib.check(L.spiht_geometry_mode(int(h), int(w), wid, int(mid), -1 if level is None else int(level), C.byref(lv),
                                     *[C.byref(t) for t in v]))
    return dict(level=lv.value, ll_h=v[0].value, ll_w=v[1].value, enc_h=v[2].value, enc_w=v[3].value,
                rec_h=v[4].value, rec_w=v[5].value)


def _filter_len(wavelet):
    """pywt.Wavelet(name).dec_len, from the library's table"""
    L = _lib.lib()
    return L.spiht_wavelet_taps(L.spiht_wavelet_id(str(wavelet).encode()))


def get_slices_and_h_w(h: int, w: int, spiht_settings: SpihtSettings, level: Optional[int]):
    """wrapper:92-139: the pywt.coeffs_to_array slices of a (1,h,w) wavedec2, the height and the width of the
    packed coefficient array.  Closed form len' = (len + F - 1)//2 instead of pywt.wavedecn_shapes."""
    wid, mid = _wavelet_mode_ids(spiht_settings)
    g = _geometry(h, w, wid, level, mid)
    hs, ws = _band_sizes(h, w, spiht_settings.wavelet, g["level"], spiht_settings.mode)
    start_h, start_w = hs[-1], ws[-1]
    slices: List[Any] = [(slice(None), slice(start_h), slice(start_w))]
    for lv in range(g["level"], 0, -1):
        dh, dw = hs[lv], ws[lv]
        slices.append({
            "ad": (slice(None), slice(0, dh), slice(start_w, start_w + dw)),
            "da": (slice(None), slice(start_h, start_h + dh), slice(0, dw)),
            "dd": (slice(None), slice(start_h, start_h + dh), slice(start_w, start_w + dw)),
        })
        start_h += dh
        start_w += dw
    return slices, start_h, start_w


def _mults_arg(per_channel_quant_scales, c):
    if per_channel_quant_scales is None:
        return None, None
    m = np.ascontiguousarray(np.array(per_channel_quant_scales), dtype=np.float64)
    if m.ndim != 1 or m.shape[0] != c:
        # numpy broadcasting of channel_mults[:,None,None] * coeffs_arr fails the same way (wrapper:167-170)
        raise ValueError("operands could not be broadcast together with shapes (%d,1,1) (%d,...)" % (m.shape[0], c))
    return m, C.c_void_p(m.ctypes.data)


def encode_image(image: np.ndarray, spiht_settings: SpihtSettings = SpihtSettings(), level: Optional[int] = None,
                 max_bits: Optional[int] = None):
    """wrapper:142-189.  image: (C,H,W) floating point pixels.  Returns EncodingResult."""
    if image.ndim != 3:
        raise ValueError('image ndim must be 3: c,h,w')
    c, h, w = image.shape

    # wrapper:158-160: the colour model change happens on the GPU, inside level 1 of the transform (color_models.fused)
    color_model = spiht_settings.color_model
    if color_model is not None:
        if color_model not in color_models.SUPPORTED_MODELS:
            color_models.convert(image, 'RGB', color_model)  # raises the reference's ValueError
        if c != 3:
            raise ValueError("colour conversion needs 3 channels")
        if image.dtype in (np.float32, np.float16):
            image = image.astype(np.float64)  # colour-science computes in float64; so does the transform that follows

    wid, mid = _wavelet_mode_ids(spiht_settings)
    g = _geometry(h, w, wid, level, mid)
    mults, mults_p = _mults_arg(spiht_settings.per_channel_quant_scales, c)

    if max_bits == None:  # noqa: E711  (as the reference)
        max_bits = 99999999999999999
    max_bits = spiht_rs._as_usize(max_bits, "max_bits")

    ctx = _lib.default_context()
    L = _lib.lib()
    # PyWavelets' dtype rule (_check_dtype): float32 and float16 pixels are transformed in single precision (and then
    # quantised in single precision, wrapper:163-172), everything else in double
    f32 = image.dtype in (np.float32, np.float16)
    img = np.ascontiguousarray(image, dtype=np.float32 if f32 else np.float64)
    if f32 and g["level"] == 0:
        raise ValueError("float32 pixels with level 0 are not supported; pass float64 pixels")
    bound = C.c_uint64()
    _lib.check(L.spiht_encode_bound(c, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], 0x3FFFFFFF, max_bits,
                                    C.byref(bound)))
    out = np.empty(max(int(bound.value), 4), dtype=np.uint8)
    nbits, mn = C.c_uint64(), C.c_uint8()
    # one C call: upload, DWT + quantise + pyramid + list coder, stream back (the context keeps its device buffers)
    with color_models.fused(ctx, color_model):
        _lib.check((L.spiht_encode_image_host_f32 if f32 else L.spiht_encode_image_host_f64)(
            ctx.handle, C.c_void_p(img.ctypes.data), c, h, w, wid, mid, -1 if level is None else int(level),
            float(spiht_settings.quantization_scale), mults_p, max_bits, C.c_void_p(out.ctypes.data), out.size,
            C.byref(nbits), C.byref(mn)))
    max_n = int(mn.value)
    out = out[:(int(nbits.value) + 7) // 8]

    return EncodingResult(out.tobytes(), h, w, c, max_n, level)


def decode_image(encoding_result: EncodingResult, spiht_settings: SpihtSettings,
                 return_metadata: bool = False) -> Union[np.ndarray, Tuple[np.ndarray, np.ndarray]]:
    """wrapper:192-216"""
    if return_metadata:
        d = decode_rec_array(encoding_result, spiht_settings, return_metadata)
        spiht_metadata = d.pop("spiht_metadata", None)
        image = decode_from_rec_arr(**d, spiht_settings=spiht_settings)
        return image, spiht_metadata
    # decode_rec_array + decode_from_rec_arr (wrapper:218-281) as one C call: the coefficient array never leaves HBM
    if encoding_result._encoding_version != ENCODER_DECODER_VERSION:
        raise ValueError(encoding_result._encoding_version)
    h, w, c, level = encoding_result.h, encoding_result.w, encoding_result.c, encoding_result.level
    wid, mid = _wavelet_mode_ids(spiht_settings)
    g = _geometry(h, w, wid, level, mid)
    buf = spiht_rs._as_u8_vec(encoding_result.encoded_bytes)
    n = spiht_rs._as_usize(encoding_result.max_n, "n")
    if n > 255:
        raise OverflowError("out of range integral type conversion attempted")
    mults, mults_p = _mults_arg(spiht_settings.per_channel_quant_scales, c)
    out = _lib.result_array((c, g["rec_h"], g["rec_w"]), np.float64)  # (page-locked: the copy back is one DMA)
    ctx = _lib.default_context()
    if spiht_settings.color_model is not None and c != 3:
        raise ValueError("colour conversion needs 3 channels")
    with color_models.fused(ctx, spiht_settings.color_model):  # wrapper:278-279, inside the last level of the inverse transform
        _lib.check(_lib.lib().spiht_decode_image_host_f64(
            ctx.handle, C.c_void_p(buf.ctypes.data if buf.size else 0), buf.size, n, c, h, w, wid, mid,
            -1 if level is None else int(level), float(spiht_settings.quantization_scale), mults_p,
            C.c_void_p(out.ctypes.data)))
    return out


def _band_sizes(h, w, wavelet, levels, mode="reflect"):
    """band heights / widths per level, [0] = the image: len' = (len + F - 1) // 2 (pywt.dwt_coeff_len), under periodization
    ceil(len / 2) -- the same rule with a two-tap filter"""
    F = 2 if str(mode) == "periodization" else _filter_len(wavelet)
    hs, ws = [int(h)], [int(w)]
    for _ in range(levels):
        hs.append((hs[-1] + F - 1) // 2)
        ws.append((ws[-1] + F - 1) // 2)
    return hs, ws


def decode_rec_array(encoding_result: EncodingResult, spiht_settings: SpihtSettings, return_metadata: bool = False):
    """wrapper:218-257: stream -> int32 coefficient array (+ the coder's per-bit metadata)."""
    if encoding_result._encoding_version != ENCODER_DECODER_VERSION:
        raise ValueError(encoding_result._encoding_version)
    er = encoding_result
    wid, mid = _wavelet_mode_ids(spiht_settings)
    g = _geometry(er.h, er.w, wid, er.level, mid)
    slices, enc_h, enc_w = get_slices_and_h_w(er.h, er.w, spiht_settings, er.level)  # (returned to the caller, as the reference does)
    spiht_metadata = None
    if not return_metadata:
        rec_arr = spiht_rs.decode(er.encoded_bytes, er.max_n, er.c, enc_h, enc_w, g["ll_h"], g["ll_w"])
    else:
        # The boxes of the sub-bands inside the packed array, as (start, end) pairs per axis: the root block, then per
        # level (coarsest first) the filters in the order the reference hands them over, 'da', 'ad', 'dd' (wrapper:240).
        # (The reference reads them off pywt's slice objects, whose `start` is None on the approximation side -- which
        # PyO3 refuses, wrapper:242-245; here they come from the band sizes, so every start is a number.)
        hs, ws = _band_sizes(er.h, er.w, spiht_settings.wavelet, g["level"], spiht_settings.mode)
        top_box = [(0, g["ll_h"]), (0, g["ll_w"])]
        level_boxes = []
        row0, col0 = g["ll_h"], g["ll_w"]  # where the detail blocks of the level start
        for lv in range(g["level"], 0, -1):
            rows, cols = (row0, row0 + hs[lv]), (col0, col0 + ws[lv])
            level_boxes.append([[rows, (0, ws[lv])],     # 'da': below the approximation
                                [(0, hs[lv]), cols],     # 'ad': right of it
                                [rows, cols]])           # 'dd': diagonal
            row0, col0 = rows[1], cols[1]
        rec_arr, spiht_metadata = spiht_rs.decode_with_metadata(er.encoded_bytes, er.max_n, er.c, enc_h, enc_w, g["ll_h"],
                                                                g["ll_w"], top_box, level_boxes)
    return dict(rec_arr=rec_arr, slices=slices, spiht_metadata=spiht_metadata, h=er.h, w=er.w, level=er.level)


def decode_from_rec_arr(rec_arr: np.ndarray, h: int, w: int, level, spiht_settings: SpihtSettings, slices=None):
    """wrapper:259-281: (rec / channel_mults) / q -> inverse DWT -> colour back.  Runs on the GPU."""
    wid, mid = _wavelet_mode_ids(spiht_settings)
    g = _geometry(h, w, wid, level, mid)
    rec = np.ascontiguousarray(rec_arr, dtype=np.int32)
    if rec.ndim != 3 or rec.shape[1] != g["enc_h"] or rec.shape[2] != g["enc_w"]:
        raise ValueError("rec_arr shape %s does not match the coefficient array (c,%d,%d)"
                         % (rec.shape, g["enc_h"], g["enc_w"]))
    c = rec.shape[0]
    mults, mults_p = _mults_arg(spiht_settings.per_channel_quant_scales, c)
    ctx = _lib.default_context()
    L = _lib.lib()
    out = _lib.result_array((c, g["rec_h"], g["rec_w"]), np.float64)
    if spiht_settings.color_model is not None and c != 3:
        raise ValueError("colour conversion needs 3 channels")
    with color_models.fused(ctx, spiht_settings.color_model):
        _lib.check(L.spiht_dequant_idwt_host_f64(ctx.handle, C.c_void_p(rec.ctypes.data), c, h, w, wid, mid,
                                                 -1 if level is None else int(level),
                                                 float(spiht_settings.quantization_scale), mults_p, C.c_void_p(out.ctypes.data)))
    return out
