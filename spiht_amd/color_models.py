"""Colour conversion used by SpihtSettings.color_model (reference: spiht/color_models.py:6-13, which calls the
third-party colour-science 0.4.4 `colour.convert`).

PARITY UNPINNED: colour-science is absent from this image and the reference holds no test that sets
`color_model`, so the exact graph `colour.convert(x, 'RGB', 'IPT')` walks cannot be checked (SURVEY.md 8c,
App. F).  What is implemented is the published Ebner-Fairchild (1998) IPT transform applied to linear RGB with
sRGB primaries / D65; it is validated for self-consistency (round trip) only.  `convert` is host-side numpy (API
parity, checker); the codec applies the change on the GPU inside level 1 of the transforms (`fused`).
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



def _params(src, dest):
    """(A, M, p) of w = M * spow(A * u, p) for one direction"""
    if (src, dest) == ("RGB", "IPT"):
        A, M, p = _XYZ2LMS @ _RGB2XYZ, _LMS2IPT, 0.43
    else:
        A, M, p = np.linalg.inv(_LMS2IPT), np.linalg.inv(_XYZ2LMS @ _RGB2XYZ), 1.0 / 0.43
    return np.ascontiguousarray(A, np.float64), np.ascontiguousarray(M, np.float64), float(p)


class fused:
    """Context manager: while active, the image-level calls on `ctx` take / return RGB pixels and code them in
    `color_model` -- the change is fused into level 1 of the transforms (spiht_ctx_set_color3; reference:
    spiht_wrapper.py:158-160, :278-279).  None or 'RGB': nothing to do."""

    def __init__(self, ctx, color_model):
        if color_model is not None and color_model not in SUPPORTED_MODELS:
            raise ValueError(f'{color_model} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
        self.ctx, self.on = ctx, color_model not in (None, "RGB")
        self.model = color_model

    def __enter__(self):
        if self.on:
            import ctypes as C
            from . import _lib
            Af, Mf, pf = _params("RGB", self.model)
            Ai, Mi, pi = _params(self.model, "RGB")
            vp = C.c_void_p
            _lib.check(_lib.lib().spiht_ctx_set_color3(self.ctx.handle, vp(Af.ctypes.data), vp(Mf.ctypes.data), pf,
                                                       vp(Ai.ctypes.data), vp(Mi.ctypes.data), pi))
        return self

    def __exit__(self, *exc):
        if self.on:
            from . import _lib
            _lib.check(_lib.lib().spiht_ctx_set_color3(self.ctx.handle, None, None, 0.0, None, None, 0.0))
        return False


def device_convert(ctx, d_ptr, B, npix, src, dest):
    """The same conversion on the GPU as a pass of its own, in place, for B images [B,3,npix] (float64) at device
    pointer d_ptr: one elementwise kernel on the context's stream (spiht_color3_batch_f64).  The codec itself does not
    use it (the change is fused into the transform, `fused`): it is the checker of the fused kernels -- same function,
    same bits -- and a utility.  Same matrices as above; the power function is the device library's, so results agree
    with convert() to a few ulp, not bit for bit."""
    import ctypes as C
    from . import _lib
    if src not in SUPPORTED_MODELS:
        raise ValueError(f'{src} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if dest not in SUPPORTED_MODELS:
        raise ValueError(f'{dest} is not a supported color model. Supported models are {SUPPORTED_MODELS}')
    if src == dest:
        return
    A, M, p = _params(src, dest)
    _lib.check(_lib.lib().spiht_color3_batch_f64(ctx.handle, C.c_void_p(d_ptr), C.c_void_p(d_ptr), int(B), int(npix),
                                                 C.c_void_p(A.ctypes.data), C.c_void_p(M.ctypes.data), float(p)))
