"""Colour conversion used by SpihtSettings.color_model (reference: spiht/color_models.py:6-13, which calls the
third-party colour-science 0.4.4 `colour.convert`).

PARITY UNPINNED: colour-science is absent from this image and the reference holds no test that sets
`color_model`, so the exact graph `colour.convert(x, 'RGB', 'IPT')` walks cannot be checked (SURVEY.md 8c,
App. F).  What is implemented is the published Ebner-Fairchild (1998) IPT transform applied to linear RGB with
sRGB primaries / D65; it is validated for self-consistency (round trip) only.  Host-side numpy: this is not on
the measured hot path.
"""
import numpy as np

SUPPORTED_MODELS = {"RGB", "IPT"}

# linear sRGB (D65) -> CIE XYZ
_RGB2XYZ = np.array([[0.4124564, 0.3575761, 0.1804375],
                     [0.2126729, 0.7151522, 0.0721750],
                     [0.0193339, 0.1191920, 0.9503041]])
_XYZ2LMS = np.array([[0.4002, 0.7075, -0.0807],
                     [-0.2280, 1.1500, 0.0612],
                     [0.0, 0.0, 0.9184]])
_LMS2IPT = np.array([[0.4000, 0.4000, 0.2000],
                     [4.4550, -4.8510, 0.3960],
                     [0.8056, 0.3572, -1.1628]])


def _rgb_to_ipt(x):
    lms = x @ (_XYZ2LMS @ _RGB2XYZ).T
    lmsp = np.sign(lms) * np.abs(lms) ** 0.43
    return lmsp @ _LMS2IPT.T


def _ipt_to_rgb(x):
    lmsp = x @ np.linalg.inv(_LMS2IPT).T
    lms = np.sign(lmsp) * np.abs(lmsp) ** (1.0 / 0.43)
    return lms @ np.linalg.inv(_XYZ2LMS @ _RGB2XYZ).T


def convert(im, src, dest):
    """im: (C,H,W) float array.  Names are matched case-sensitively as the reference does
    (color_models.py:7-10)."""
    if src not in SUPPORTED_MODELS:
        raise ValueError(f'{src} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if dest not in SUPPORTED_MODELS:
        raise ValueError(f'{dest} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if src == dest:
        return im
    im = np.moveaxis(np.asarray(im, dtype=np.float64), 0, -1)
    if im.shape[-1] != 3:
        raise ValueError("colour conversion needs 3 channels")
    out = _rgb_to_ipt(im) if (src, dest) == ("RGB", "IPT") else _ipt_to_rgb(im)
    return np.moveaxis(out, -1, 0)



def device_convert(ctx, d_ptr, B, npix, src, dest):
    """The same conversion on the GPU, in place, for B images [B,3,npix] (float64) at device pointer d_ptr: one
    elementwise kernel on the context's stream (spiht_color3_batch_f64).  Same matrices as above; the power function is
    the device library's, so results agree with convert() to a few ulp, not bit for bit."""
    import ctypes as C
    from . import _lib
    if src not in SUPPORTED_MODELS:
        raise ValueError(f'{src} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if dest not in SUPPORTED_MODELS:
        raise ValueError(f'{dest} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if src == dest:
        return
    if (src, dest) == ("RGB", "IPT"):
        A, M, p = _XYZ2LMS @ _RGB2XYZ, _LMS2IPT, 0.43
    else:
        A, M, p = np.linalg.inv(_LMS2IPT), np.linalg.inv(_XYZ2LMS @ _RGB2XYZ), 1.0 / 0.43
    A, M = np.ascontiguousarray(A, np.float64), np.ascontiguousarray(M, np.float64)
    _lib.check(_lib.lib().spiht_color3_batch_f64(ctx.handle, C.c_void_p(d_ptr), C.c_void_p(d_ptr), int(B), int(npix),
                                                 C.c_void_p(A.ctypes.data), C.c_void_p(M.ctypes.data), float(p)))
