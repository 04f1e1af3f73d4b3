"""Golden-vector capture (run ONLY in the build container, never on the GPU box).

Imports the reference in place from /root/reference (nothing is copied) with
in-memory stand-ins for the two modules that cannot exist here (the un-built
Rust extension `spiht.spiht` and the absent `colour` package), runs it on small
seeded inputs and stores inputs + outputs as .npz fixtures next to this file.

  part "loops"   (any interpreter)             spiht/spiht_py.py encode/decode list logic on
                                               int32 arrays (pywt stubbed as pass-through)
                                               -> spiht_py_loops.npz
  part "wrapper" (/opt/conda/bin/python3.9,    spiht/spiht_wrapper.py front half (pixels -> int32
                  real PyWavelets 1.1.1)       coefficients handed to the Rust boundary), back half
                                               (int32 rec array -> pixels), geometry, filter banks
                                               -> wrapper_pywt.npz

  part "wavelets" / "modes" (python3.9, PyWavelets 1.1.1; PyWavelets alone, the reference is not imported)
                                               every wavelet with at most 20 taps / the extension modes that are not index
                                               maps and periodization -> wavelets_pywt.npz, modes_pywt.npz

  part "blocky"  (python3.9, PyWavelets 1.1.1)  float64 coefficient arrays of piecewise-constant images, bit for bit
                                               -> blocky_pywt.npz (pins the summation order at the right / bottom edge)

usage:
  PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py loops
  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -W ignore tests/golden/make_golden.py wrapper
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def synth_image(seed, c, H, W):
    """SURVEY.md 8(d) pixel-domain generator (1/f^2-like, uint8-rounded, /255)."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((c, H, W))
    b = np.cumsum(np.cumsum(g, axis=1), axis=2)
    mn = b.min(axis=(1, 2), keepdims=True)
    mx = b.max(axis=(1, 2), keepdims=True)
    b = (b - mn) / (mx - mn)
    b = b + 0.02 * rng.standard_normal((c, H, W))
    return np.round(np.clip(b, 0, 1) * 255).astype(np.uint8) / 255


def synth_coeffs(seed, c, h, w, ll_h, ll_w, scale=3000.0):
    """SURVEY.md 8(d) coefficient-domain generator."""
    rng = np.random.default_rng(seed)
    i = np.arange(h)[:, None]
    j = np.arange(w)[None, :]
    t = np.maximum(0, np.ceil(np.log2(np.maximum((i + 1) / ll_h, (j + 1) / ll_w))))
    sc = scale * 2.0 ** (-1.3 * t)
    return np.trunc(rng.laplace(0, 1, (c, h, w)) * sc[None]).astype(np.int32)


def _stub_common():
    sys.path.insert(0, REF)
    col = types.ModuleType("colour")
    col.COLOURSPACE_MODELS = ("RGB", "IPT")
    sys.modules["colour"] = col


def part_loops():
    _stub_common()
    ext = types.ModuleType("spiht.spiht")
    ext.encode = ext.decode = None
    sys.modules["spiht.spiht"] = ext
    sys.modules["pywt"] = types.ModuleType("pywt")  # pass-through; replaced below
    import spiht.spiht_py as sp
    from collections import namedtuple

    sp.EncodingResult = namedtuple(
        "ER", "encoded_bytes h w c max_n ll_h ll_w wavelet quantization_scale slices mode")

    class _LL:
        def __init__(s, shape):
            s.shape = shape

    st = {}
    sp.pywt.wavedec2 = lambda image, wavelet=None, level=None, mode=None: [
        _LL((image.shape[0], st["ll_h"], st["ll_w"])), image]
    sp.pywt.coeffs_to_array = lambda coeffs, axes=None: (
        coeffs[1].astype(np.float64), [(slice(None), slice(0, st["ll_h"]), slice(0, st["ll_w"]))])
    sp.pywt.array_to_coeffs = lambda arr, slices, output_format=None: arr
    sp.pywt.waverec2 = lambda coeffs, mode=None, wavelet=None: coeffs

    cases = []
    rng = np.random.default_rng(7)
    # (c,h,w,ll_h,ll_w,max_bits) -- dyadic, odd sizes (Q3), odd LL (Q4), shallow trees (Q5)
    shapes = [(1, 8, 8, 2, 2, 4000), (1, 16, 16, 2, 2, 10000), (2, 16, 16, 4, 4, 3000),
              (3, 13, 17, 3, 5, 20000), (1, 21, 19, 5, 3, 20000), (3, 24, 40, 3, 5, 2500),
              (1, 32, 32, 2, 2, 777), (2, 11, 23, 4, 6, 20000), (1, 4, 4, 2, 2, 500),
              (3, 33, 29, 6, 5, 6001), (1, 8, 8, 4, 4, 3000), (1, 6, 6, 3, 3, 3000)]
    for idx, (c, h, w, lh, lw, mb) in enumerate(shapes):
        if idx % 2 == 0:
            arr = synth_coeffs(100 + idx, c, h, w, lh, lw, scale=300.0)
        else:
            arr = rng.normal(0, 16, (c, h, w)).astype(np.int32)
        if idx == 1:
            arr = 32 * np.ones((c, h, w), np.int32)  # simple_test_encode input (encoder_decoder.rs:865-875)
        if int(np.abs(arr).max()) == 0:
            arr[0, 0, 0] = 5
        st.update(ll_h=lh, ll_w=lw)
        with contextlib.redirect_stdout(io.StringIO()):
            r = sp.encode_image_py(arr, level=None, max_bits=mb, quantization_scale=1)
            rec = sp.decode_image_py(r)
        bits = np.array([int(b) for b in r.encoded_bytes], dtype=np.uint8)  # zero-filled to max_bits
        cases.append(dict(arr=arr, ll_h=lh, ll_w=lw, max_bits=mb, bits=bits, max_n=int(r.max_n),
                          rec=np.asarray(rec).astype(np.int32)))
    # helper known answers straight from the reference functions
    helpers = dict(
        set_bit=np.array([[x, n, b, sp.set_bit(x, n, b)] for x, n, b in
                          [(-96, 5, 0), (-96, 5, 1), (-64, 5, 1), (96, 5, 1), (96, 5, 0), (7, 0, 0), (-7, 3, 1)]],
                         dtype=np.int64),
        offspring=np.array([[i, j, h, w, lh, lw] + [v for o in (sp.get_offspring(i, j, h, w, lh, lw) or [(-1, -1)] * 4)
                                                    for v in o]
                            for (i, j, h, w, lh, lw) in
                            [(0, 0, 16, 16, 2, 2), (0, 1, 16, 16, 2, 2), (1, 0, 16, 16, 2, 2), (1, 1, 16, 16, 2, 2),
                             (2, 4, 13, 17, 3, 5), (2, 3, 13, 17, 3, 5), (5, 7, 13, 17, 3, 5), (6, 8, 13, 17, 3, 5),
                             (3, 3, 16, 16, 2, 2), (7, 7, 16, 16, 2, 2), (8, 1, 16, 16, 2, 2), (12, 18, 1111, 1949, 13, 19)]],
                           dtype=np.int64))
    out = {}
    for n, cs in enumerate(cases):
        for k, v in cs.items():
            out["case%02d_%s" % (n, k)] = np.asarray(v)
    out["ncases"] = np.array(len(cases))
    out.update({"helper_" + k: v for k, v in helpers.items()})
    np.savez_compressed(os.path.join(HERE, "spiht_py_loops.npz"), **out)
    print("wrote spiht_py_loops.npz:", len(cases), "cases")


def part_wrapper():
    _stub_common()
    cap = {}
    ext = types.ModuleType("spiht.spiht")

    def _enc(arr, ll_h, ll_w, mb):
        cap.update(arr=np.array(arr, copy=True), ll_h=ll_h, ll_w=ll_w, max_bits=mb)
        return b"", 0

    ext.encode = _enc
    ext.decode = None
    sys.modules["spiht.spiht"] = ext
    import pywt
    from spiht.spiht_wrapper import SpihtSettings, encode_image, decode_from_rec_arr, get_slices_and_h_w

    out = {"pywt_version": np.array(pywt.__version__)}
    for name in ["bior2.2", "bior4.4", "bior6.8", "haar"]:
        fb = pywt.Wavelet(name).filter_bank
        out["fb_" + name] = np.array(fb, dtype=np.float64)

    # geometry: the five BASELINE configs + a sweep of odd/even shapes
    geo = []
    geo_cases = [(512, 512, "bior2.2", 5), (1080, 1920, "bior2.2", 7), (1024, 1024, "bior2.2", None),
                 (4096, 4096, "bior6.8", 9), (256, 342, "bior2.2", None), (256, 511, "bior2.2", None),
                 (289, 206, "bior4.4", None), (33, 47, "bior2.2", 2), (64, 64, "bior4.4", 3),
                 (100, 37, "bior6.8", 1), (75, 75, "haar", 3), (1024, 768, "bior2.2", 6)]
    for (H, W, wv, lv) in geo_cases:
        s = SpihtSettings(wavelet=wv)
        slices, eh, ew = get_slices_and_h_w(H, W, s, lv)
        nlev = len(slices) - 1
        geo.append([H, W, ["bior2.2", "bior4.4", "bior6.8", "haar"].index(wv), -1 if lv is None else lv,
                    slices[0][1].stop, slices[0][2].stop, eh, ew, nlev])
    out["geometry"] = np.array(geo, dtype=np.int64)

    # front half + back half
    cases = [
        dict(c=1, H=32, W=32, wavelet="bior2.2", mode="reflect", level=2, q=50.0, mults=None),
        dict(c=3, H=48, W=64, wavelet="bior2.2", mode="reflect", level=None, q=50.0, mults=None),
        dict(c=3, H=37, W=53, wavelet="bior2.2", mode="reflect", level=2, q=50.0, mults=None),
        dict(c=3, H=64, W=96, wavelet="bior2.2", mode="reflect", level=3, q=1.0, mults=[100.0, 20.0, 20.0]),
        dict(c=1, H=96, W=128, wavelet="bior4.4", mode="symmetric", level=None, q=50.0, mults=None),
        dict(c=2, H=45, W=70, wavelet="bior4.4", mode="symmetric", level=2, q=255.0, mults=[1.0, 0.2]),
        dict(c=1, H=160, W=144, wavelet="bior6.8", mode="reflect", level=None, q=50.0, mults=None),
        dict(c=1, H=80, W=80, wavelet="bior6.8", mode="reflect", level=4, q=50.0, mults=None),  # level > max (Q13)
        dict(c=1, H=40, W=56, wavelet="haar", mode="reflect", level=3, q=50.0, mults=None),
        dict(c=1, H=50, W=41, wavelet="bior2.2", mode="periodic", level=2, q=50.0, mults=None),
        dict(c=1, H=50, W=41, wavelet="bior2.2", mode="zero", level=2, q=50.0, mults=None),
        dict(c=1, H=50, W=41, wavelet="bior2.2", mode="constant", level=2, q=50.0, mults=None),
        dict(c=3, H=120, W=200, wavelet="bior2.2", mode="reflect", level=4, q=50.0, mults=None),
    ]
    rng = np.random.default_rng(11)
    for n, cs in enumerate(cases):
        img = synth_image(1000 + n, cs["c"], cs["H"], cs["W"])
        s = SpihtSettings(wavelet=cs["wavelet"], quantization_scale=cs["q"], mode=cs["mode"],
                          per_channel_quant_scales=cs["mults"])
        encode_image(img, s, level=cs["level"], max_bits=None)
        coeffs = cap["arr"]
        # float (unquantised) packed array for tolerance checks of the DWT alone
        co = pywt.wavedec2(img, wavelet=cs["wavelet"], level=cs["level"], mode=cs["mode"])
        farr, _ = pywt.coeffs_to_array(co, axes=(-2, -1))
        # back half on a perturbed rec array (as a truncated decode would give)
        rec = (coeffs - (coeffs % 4) * (rng.random(coeffs.shape) < 0.5)).astype(np.int32)
        rec_img = decode_from_rec_arr(rec, cs["H"], cs["W"], cs["level"], s)
        p = "case%02d_" % n
        out[p + "img"] = img
        out[p + "coeffs"] = coeffs
        out[p + "farr"] = farr
        out[p + "ll"] = np.array([cap["ll_h"], cap["ll_w"]])
        out[p + "max_bits"] = np.array(cap["max_bits"], dtype=np.uint64)
        out[p + "rec"] = rec
        out[p + "rec_img"] = rec_img
        out[p + "meta"] = np.array([cs["c"], cs["H"], cs["W"], -1 if cs["level"] is None else cs["level"]])
        out[p + "wavelet"] = np.array(cs["wavelet"])
        out[p + "mode"] = np.array(cs["mode"])
        out[p + "q"] = np.array(cs["q"])
        out[p + "mults"] = np.array(cs["mults"] if cs["mults"] else [], dtype=np.float64)
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "wrapper_pywt.npz"), **out)
    print("wrote wrapper_pywt.npz:", len(cases), "cases; pywt", pywt.__version__)


def part_bench():
    """Digests of what the reference wrapper hands to its Rust core (spiht_wrapper.py:163-172, PyWavelets) for the
    eight 1080p images of bench.py and a few odd-sized uint8-derived ones: on such pixels the products coefficient x q
    often land exactly on integers, so the quantised array is sensitive to the last bit of the transform."""
    import hashlib
    import pywt
    cases = [(1000 + i, 3, 1080, 1920, "bior2.2", 7, 50.0) for i in range(8)]
    cases += [(5, 3, 511, 733, "bior2.2", None, 50.0), (6, 1, 512, 512, "bior4.4", 5, 50.0), (7, 3, 300, 301, "bior6.8", 3, 50.0),
              (8, 3, 257, 255, "bior2.2", 4, 10.0)]
    out = {"cases": np.array([[s, c, h, w, -1 if lv is None else lv] for s, c, h, w, _, lv, _ in cases]),
           "wavelets": np.array([wv for _, _, _, _, wv, _, _ in cases]), "q": np.array([q for *_, q in cases])}
    dig = []
    for seed, c, h, w, wv, lv, q in cases:
        img = synth_image(seed, c, h, w)
        arr, _ = pywt.coeffs_to_array(pywt.wavedec2(img, wv, level=lv, mode="reflect"), axes=(-2, -1))
        qa = np.ascontiguousarray((arr * q).astype(np.int32))
        dig.append(hashlib.sha1(qa.tobytes()).hexdigest() + ":%dx%dx%d" % qa.shape)
    out["sha1"] = np.array(dig)
    np.savez_compressed(os.path.join(HERE, "bench_pywt_digests.npz"), **out)
    print("wrote bench_pywt_digests.npz:", len(cases), "cases; pywt", pywt.__version__)


def part_wrapper32():
    """float32 pixels: PyWavelets transforms them in single precision and the wrapper quantises in single precision
    (double once per-channel scales are applied).  Full float32 coefficient arrays for two small images, digests of
    the int32 arrays handed to the Rust core for more and larger ones."""
    import hashlib
    import pywt
    cases = [(21, 3, 64, 80, "bior2.2", 3, 50.0, None), (22, 1, 57, 43, "bior4.4", 2, 10.0, None),
             (23, 3, 96, 128, "bior2.2", None, 50.0, [50.0, 15.0, 15.0]), (24, 3, 511, 733, "bior2.2", None, 50.0, None),
             (1000, 3, 1080, 1920, "bior2.2", 7, 50.0, None), (25, 1, 300, 301, "bior6.8", 3, 33.3, None),
             (26, 3, 257, 255, "bior2.2", 4, 0.1, None),
             (27, 2, 61, 77, "bior2.2", 2, 50.0, None, "symmetric"), (28, 1, 64, 48, "bior4.4", None, 50.0, None, "periodic"),
             (29, 3, 45, 52, "bior2.2", 1, 50.0, None, "zero"), (30, 2, 61, 77, "bior6.8", 2, 50.0, None, "constant"),
             (31, 1, 50, 41, "haar", 3, 20.0, None, "symmetric")]
    out = {"ncases": np.array(len(cases))}
    for i, cs in enumerate(cases):
        seed, c, h, w, wv, lv, q, mults = cs[:8]
        mode = cs[8] if len(cs) > 8 else "reflect"
        img = synth_image(seed, c, h, w).astype(np.float32 if i != 1 else np.float16)
        arr, _ = pywt.coeffs_to_array(pywt.wavedec2(img, wavelet=wv, level=lv, mode=mode), axes=(-2, -1))
        assert arr.dtype == np.float32
        a2 = arr
        if mults is not None:
            a2 = np.array(mults)[:, None, None] * a2
        qa = np.ascontiguousarray((a2 * q).astype(np.int32))
        p = "c%d_" % i
        out[p + "meta"] = np.array([seed, c, h, w, -1 if lv is None else lv])
        out[p + "wavelet"] = np.array(wv)
        out[p + "mode"] = np.array(mode)
        out[p + "q"] = np.array(q)
        out[p + "mults"] = np.array(mults if mults else [], dtype=np.float64)
        out[p + "f16"] = np.array(i == 1)
        out[p + "sha1"] = np.array(hashlib.sha1(qa.tobytes()).hexdigest() + ":%dx%dx%d" % qa.shape)
        if h * w < 20000:
            out[p + "arr"] = arr
            out[p + "quant"] = qa
    np.savez_compressed(os.path.join(HERE, "wrapper32_pywt.npz"), **out)
    print("wrote wrapper32_pywt.npz:", len(cases), "cases; pywt", pywt.__version__, "numpy", np.__version__)


def blocky_image(seed, c, H, W):
    """piecewise-constant picture with a handful of grey levels (uint8 / 255): on such pixels many products
    coefficient x q are integers up to the last bits of the sum, so the truncating quantiser sees the ORDER of the
    additions (PyWavelets adds the taps that read the signal extension first on the right / bottom overhang)"""
    rng = np.random.default_rng(seed)
    levels = rng.choice(256, size=int(rng.integers(2, 6)), replace=False)
    bh, bw = int(rng.integers(2, 9)), int(rng.integers(2, 9))
    cells = rng.choice(levels, size=(c, (H + bh - 1) // bh, (W + bw - 1) // bw))
    img = np.kron(cells, np.ones((bh, bw), dtype=np.int64))[:, :H, :W]
    return img.astype(np.uint8) / 255


def part_blocky():
    """float64 packed coefficient arrays (every bit) and the int32 arrays handed to the Rust core, for blocky images of odd
    and even sizes: pins the summation order of the forward transform at the right / bottom edge."""
    import pywt
    cases = []
    rng = np.random.default_rng(99)
    wvs = ["bior2.2", "bior2.2", "bior2.2", "bior4.4", "bior6.8"]
    modes = ["reflect", "reflect", "symmetric", "reflect", "reflect", "periodic", "zero", "constant"]
    for i in range(40):
        H, W = int(rng.integers(17, 72)), int(rng.integers(17, 72))
        wv = wvs[i % len(wvs)]
        F = {"bior2.2": 6, "bior4.4": 10, "bior6.8": 18}[wv]
        lv = 1 + i % 3
        while lv > 1 and min(H, W) < F * 2 ** (lv - 1):  # every level's input at least as long as the filter
            lv -= 1
        cases.append((300 + i, 1 + i % 3, H, W, wv, lv, [50.0, 50.0, 10.0, 255.0][i % 4], modes[i % len(modes)]))
    out = {"ncases": np.array(len(cases))}
    for i, (seed, c, H, W, wv, lv, q, mode) in enumerate(cases):
        img = blocky_image(seed, c, H, W)
        arr, _ = pywt.coeffs_to_array(pywt.wavedec2(img, wavelet=wv, level=lv, mode=mode), axes=(-2, -1))
        assert arr.dtype == np.float64
        p = "c%d_" % i
        out[p + "meta"] = np.array([seed, c, H, W, lv])
        out[p + "wavelet"] = np.array(wv)
        out[p + "mode"] = np.array(mode)
        out[p + "q"] = np.array(q)
        out[p + "arr"] = arr
        out[p + "quant"] = np.ascontiguousarray((arr * q).astype(np.int32))
    np.savez_compressed(os.path.join(HERE, "blocky_pywt.npz"), **out)
    print("wrote blocky_pywt.npz:", len(cases), "cases; pywt", pywt.__version__)


def _transform_cases(out, cases, pywt):
    """what the reference's wrapper does with SpihtSettings.wavelet / .mode around its coder (spiht_wrapper.py:163-172:
    wavedec2 -> coeffs_to_array -> * q -> int32; :259-276: / q -> array_to_coeffs -> waverec2), with PyWavelets alone:
    the float64 packed array (every bit), the int32 array, and the picture waverec2 gives back from a thinned-out copy of it"""
    rng = np.random.default_rng(5)
    for i, (seed, c, H, W, wv, lv, q, mode, blocky) in enumerate(cases):
        img = blocky_image(seed, c, H, W) if blocky else synth_image(seed, c, H, W)
        co = pywt.wavedec2(img, wavelet=wv, level=lv, mode=mode)
        arr, slices = pywt.coeffs_to_array(co, axes=(-2, -1))
        assert arr.dtype == np.float64
        qa = np.ascontiguousarray((arr * q).astype(np.int32))
        rec = (qa - (qa % 4) * (rng.random(qa.shape) < 0.5)).astype(np.int32)
        back = pywt.waverec2(pywt.array_to_coeffs(rec / q, slices, output_format="wavedec2"), mode=mode, wavelet=wv)
        p = "c%d_" % i
        out[p + "meta"] = np.array([seed, c, H, W, lv, int(blocky)])
        out[p + "wavelet"] = np.array(wv)
        out[p + "mode"] = np.array(mode)
        out[p + "q"] = np.array(q)
        out[p + "arr"] = arr
        out[p + "quant"] = qa
        out[p + "rec"] = rec
        out[p + "rec_img"] = back
    out["ncases"] = np.array(len(cases))


def part_wavelets():
    """every discrete wavelet of PyWavelets with at most 20 taps (the reference hands SpihtSettings.wavelet straight to
    pywt: spiht_wrapper.py:163, :276), small odd-sized pictures, two or three levels, the five index-mapping extension
    modes in turn -> wavelets_pywt.npz"""
    import pywt
    names = [n for n in pywt.wavelist(kind="discrete") if pywt.Wavelet(n).dec_len <= 20]
    modes = ["reflect", "symmetric", "periodic", "zero", "constant"]
    rng = np.random.default_rng(123)
    cases = []
    for i, n in enumerate(names):
        F = pywt.Wavelet(n).dec_len
        H, W = int(rng.integers(2 * F + 3, 2 * F + 40)), int(rng.integers(2 * F + 3, 2 * F + 40))
        lv = 1 + i % 3
        while lv > 1 and min(H, W) < F * 2 ** (lv - 1):  # every level's input at least as long as the filter
            lv -= 1
        cases.append((400 + i, 1 + i % 2, H, W, n, lv, [50.0, 255.0, 10.0][i % 3], modes[i % len(modes)], i % 2 == 1))
    out = {"pywt_version": np.array(pywt.__version__), "names": np.array(names),
           "dec_len": np.array([pywt.Wavelet(n).dec_len for n in names])}
    for n in names:
        out["fb_" + n] = np.array(pywt.Wavelet(n).filter_bank, dtype=np.float64)
    _transform_cases(out, cases, pywt)
    np.savez_compressed(os.path.join(HERE, "wavelets_pywt.npz"), **out)
    print("wrote wavelets_pywt.npz:", len(cases), "cases; pywt", pywt.__version__)


def part_modes():
    """the four extension modes of PyWavelets that are not index maps -- smooth, antisymmetric, antireflect -- and
    periodization (another length rule: ceil(N / 2) per level), over several filter lengths, odd and even sizes, inputs
    longer and (for the first three) shorter than the filter -> modes_pywt.npz"""
    import pywt
    rng = np.random.default_rng(321)
    cases = []
    wvs = ["bior2.2", "haar", "bior4.4", "db2", "sym4", "bior6.8", "coif1", "db7"]
    k = 0
    for mode in ["smooth", "antisymmetric", "antireflect", "periodization"]:
        for j in range(10):
            wv = wvs[(j + k) % len(wvs)]
            F = pywt.Wavelet(wv).dec_len
            H, W = int(rng.integers(F + 1, 4 * F + 30)), int(rng.integers(F + 1, 4 * F + 30))
            lv = 1 + j % 3
            if j < 8:
                while lv > 1 and min(H, W) < F * 2 ** (lv - 1):
                    lv -= 1
            cases.append((600 + k, 1 + j % 2, H, W, wv, lv, [50.0, 255.0, 10.0][j % 3], mode, j % 2 == 0))
            k += 1
    out = {"pywt_version": np.array(pywt.__version__)}
    _transform_cases(out, cases, pywt)
    np.savez_compressed(os.path.join(HERE, "modes_pywt.npz"), **out)
    print("wrote modes_pywt.npz:", len(cases), "cases; pywt", pywt.__version__)


def part_modes32():
    """the same four modes on float32 pixels (PyWavelets then transforms in single precision): coefficient arrays, every
    bit, and the int32 array the wrapper's float32 quantisation makes of them -> modes32_pywt.npz"""
    import pywt
    rng = np.random.default_rng(77)
    out = {"pywt_version": np.array(pywt.__version__)}
    i = 0
    # (reflect with the coiflets: PyWavelets builds their single-precision filters from a float table -- not the doubles rounded)
    for mode in ["smooth", "antisymmetric", "antireflect", "periodization", "reflect"]:
        for j, wv in enumerate(["bior2.2", "bior4.4", "db2", "sym4", "haar", "coif1"] if mode != "reflect" else ["coif1", "coif2", "coif3"]):
            F = pywt.Wavelet(wv).dec_len
            H, W = int(rng.integers(max(3, F - 2), 3 * F + 20)), int(rng.integers(max(3, F - 2), 3 * F + 20))
            lv = 1 + j % 3
            img = synth_image(800 + i, 1 + j % 2, H, W).astype(np.float32)
            arr, _ = pywt.coeffs_to_array(pywt.wavedec2(img, wavelet=wv, level=lv, mode=mode), axes=(-2, -1))
            assert arr.dtype == np.float32
            p = "c%d_" % i
            out[p + "img"] = img
            out[p + "arr"] = arr
            out[p + "quant"] = np.ascontiguousarray((arr * 50.0).astype(np.int32))
            out[p + "wavelet"] = np.array(wv)
            out[p + "mode"] = np.array(mode)
            out[p + "level"] = np.array(lv)
            i += 1
    out["ncases"] = np.array(i)
    np.savez_compressed(os.path.join(HERE, "modes32_pywt.npz"), **out)
    print("wrote modes32_pywt.npz:", i, "cases; pywt", pywt.__version__)


def part_short():
    """inputs SHORTER than the filter (levels above pywt.dwt_max_level: PyWavelets warns and transforms them all the same,
    and so does the reference, spiht_wrapper.py:163): one and two levels, all eight extension modes in float64, the five
    index-map modes in float32 as well -> short_pywt.npz (coefficient arrays, every bit)"""
    import pywt
    rng = np.random.default_rng(1)
    out = {"pywt_version": np.array(pywt.__version__)}
    i = 0
    for wv in ["bior2.2", "bior6.8", "db4"]:
        F = pywt.Wavelet(wv).dec_len
        for mode in ["reflect", "symmetric", "periodic", "zero", "constant", "smooth", "antisymmetric", "antireflect"]:
            for k, (H, W) in enumerate([(F - 1, F + 3), (F // 2, 2 * F), (3, F - 2), (2, 5), (F + 5, F - 3)]):
                if H < 2 or W < 2:
                    continue
                img = rng.random((1 + k % 2, H, W))
                for dt in (np.float64, np.float32):
                    if dt == np.float32 and mode in ("smooth", "antisymmetric", "antireflect"):
                        continue
                    lv = 1 + k % 2
                    arr, _ = pywt.coeffs_to_array(pywt.wavedec2(img.astype(dt), wv, mode=mode, level=lv), axes=(-2, -1))
                    assert arr.dtype == dt
                    p = "c%d_" % i
                    out[p + "img"] = img.astype(dt)
                    out[p + "arr"] = arr
                    out[p + "wavelet"] = np.array(wv)
                    out[p + "mode"] = np.array(mode)
                    out[p + "level"] = np.array(lv)
                    i += 1
    out["ncases"] = np.array(i)
    np.savez_compressed(os.path.join(HERE, "short_pywt.npz"), **out)
    print("wrote short_pywt.npz:", i, "cases; pywt", pywt.__version__)


def thin_out(qa, seed):
    """a decoder's partial picture of the int32 array: about half the coefficients lose their two low bits"""
    rng = np.random.default_rng(seed)
    return (qa - (qa % 4) * (rng.random(qa.shape) < 0.5)).astype(np.int32)


def part_long():
    """the 53 discrete wavelets of PyWavelets with MORE than 20 taps (db11-38, sym11-20, coif4-17, dmey: up to 102 taps),
    which the reference takes like any other name (spiht_wrapper.py:163, :276): pictures longer and shorter than the filter,
    one or two levels, all nine extension modes in turn, float64 -- and float32 pixels for the coiflets (whose
    single-precision filters PyWavelets builds from a float table) and a few others.  SHA-256 of the packed array, of the
    int32 array and of the picture waverec2 gives back from thin_out(int32 array), for every case; the arrays themselves for
    every fifth -> long_pywt.npz"""
    import hashlib
    import pywt
    names = [n for n in pywt.wavelist(kind="discrete") if pywt.Wavelet(n).dec_len > 20]
    modes = ["reflect", "symmetric", "periodic", "zero", "constant", "smooth", "antisymmetric", "antireflect", "periodization"]
    rng = np.random.default_rng(2024)
    sha = lambda a: np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
    out = {"pywt_version": np.array(pywt.__version__), "names": np.array(names),
           "dec_len": np.array([pywt.Wavelet(n).dec_len for n in names])}
    for n in names:
        out["fb_" + n] = np.array(pywt.Wavelet(n).filter_bank, dtype=np.float64)
    i = 0
    for k, n in enumerate(names):
        F = pywt.Wavelet(n).dec_len
        for dt in (np.float64, np.float32):
            if dt == np.float32 and not (n.startswith("coif") or n in ("db11", "db38", "sym20", "dmey")):
                continue
            kind = (k + (dt == np.float32)) % 3  # both sides longer than the filter / one shorter / both shorter
            H = int(rng.integers(F + 1, F + 24)) if kind == 0 else int(rng.integers(3, F))
            W = int(rng.integers(F + 1, F + 40)) if kind <= 1 else int(rng.integers(3, F))
            if k % 2:
                H, W = W, H
            lv, c, q = 1 + k % 2, 1 + (k % 4 == 0), [50.0, 255.0, 10.0][k % 3]
            mode = modes[(k + 4 * (dt == np.float32)) % len(modes)]
            img = (blocky_image(900 + i, c, H, W) if k % 2 else synth_image(900 + i, c, H, W)).astype(dt)
            co = pywt.wavedec2(img, wavelet=n, level=lv, mode=mode)
            arr, slices = pywt.coeffs_to_array(co, axes=(-2, -1))
            assert arr.dtype == dt
            qa = np.ascontiguousarray((arr * dt(q)).astype(np.int32))
            p = "c%d_" % i
            out[p + "meta"] = np.array([900 + i, c, H, W, lv, k % 2, int(dt == np.float32)])
            out[p + "wavelet"], out[p + "mode"], out[p + "q"] = np.array(n), np.array(mode), np.array(q)
            out[p + "shape"] = np.array(arr.shape)
            out[p + "sha_arr"], out[p + "sha_quant"] = sha(arr), sha(qa)
            if dt == np.float64:
                rec = thin_out(qa, 900 + i)
                back = pywt.waverec2(pywt.array_to_coeffs(rec / q, slices, output_format="wavedec2"), mode=mode, wavelet=n)
                out[p + "back_shape"], out[p + "sha_rec_img"] = np.array(back.shape), sha(back)
            if i % 5 == 0:
                out[p + "arr"] = arr
                if dt == np.float64:
                    out[p + "rec_img"] = back
            i += 1
    out["ncases"] = np.array(i)
    np.savez_compressed(os.path.join(HERE, "long_pywt.npz"), **out)
    print("wrote long_pywt.npz:", i, "cases; pywt", pywt.__version__)


if __name__ == "__main__":
    {"loops": part_loops, "wrapper": part_wrapper, "bench": part_bench, "wrapper32": part_wrapper32,
     "blocky": part_blocky, "wavelets": part_wavelets, "modes": part_modes, "modes32": part_modes32, "short": part_short, "long": part_long}[sys.argv[1]]()
