"""CPU tests: the C-ABI library loads and exports every symbol include/spiht_hip.h declares; the Python
boundary mirrors the reference's names and argument checking.  No compute calls (no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "spiht_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(spiht_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from spiht_amd import _lib
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for s in declared:
        assert hasattr(L, s), "libspiht_hip.so does not export %s" % s
    assert sorted(_lib.SYMBOLS) == declared
    assert L.spiht_abi_version() == 2
    # the shipped library is not a diagnostic build (-DSPIHT_DIAG -DDEC_PROF: timers inside the list decoder's loops,
    # tools/prof_decode.py): its entry points are not there
    for s in ("spiht_debug_words", "spiht_debug_words_ext"):
        assert not hasattr(L, s), "libspiht_hip.so is a diagnostic build (%s)" % s


def test_package_surface_matches_reference_init():
    import spiht_amd
    # /root/reference/spiht/__init__.py:1-2
    for name in ["encode_image", "decode_image", "EncodingResult", "SpihtSettings", "ENCODER_DECODER_VERSION", "encode",
                 "decode"]:
        assert hasattr(spiht_amd, name)
    s = spiht_amd.SpihtSettings()
    assert (s.wavelet, s.quantization_scale, s.mode, s.color_model, s.per_channel_quant_scales) == \
        ("bior2.2", 50.0, "reflect", None, None)
    # positional order is API (demonstrate.py:23-29)
    s = spiht_amd.SpihtSettings("bior4.4", 1.0, "symmetric", "IPT", [100., 20., 20.])
    assert s.mode == "symmetric" and s.per_channel_quant_scales == [100., 20., 20.]
    assert spiht_amd.ENCODER_DECODER_VERSION == "0.0.2"
    r = spiht_amd.EncodingResult(b"ab", 4, 5, 3, 7, None)
    d = r.to_dict()
    assert d["encoding_result_h"] == 4 and d["encoding_result__encoding_version"] == "0.0.2"
    assert spiht_amd.EncodingResult.from_dict(d) == r


def test_geometry_and_bound_without_gpu():
    """host-only entry points of the C ABI (no device needed)"""
    import ctypes as C
    from spiht_amd import _lib
    from spiht_amd.spiht_wrapper import SpihtSettings, get_slices_and_h_w
    L = _lib.lib()
    # SURVEY.md App. A
    for (H, W, wv, lv, ll, enc) in [(512, 512, "bior2.2", 5, (20, 20), (533, 533)),
                                    (1080, 1920, "bior2.2", 7, (13, 19), (1111, 1949)),
                                    (1024, 1024, "bior2.2", None, (12, 12), (1053, 1053)),
                                    (4096, 4096, "bior6.8", 9, (24, 24), (4241, 4241))]:
        slices, eh, ew = get_slices_and_h_w(H, W, SpihtSettings(wavelet=wv), lv)
        assert (slices[0][1].stop, slices[0][2].stop) == ll and (eh, ew) == enc
    b = C.c_uint64()
    assert L.spiht_encode_bound(3, 1111, 1949, 13, 19, 6485, 1036800, C.byref(b)) == 0
    assert b.value == 129600
    assert L.spiht_encode_bound(3, 1111, 1949, 13, 19, 6485, 0, C.byref(b)) == 0
    assert b.value > 3 * 1111 * 1949 // 8
    assert L.spiht_encode_bound(1, 8, 8, 1, 2, 5, 0, C.byref(b)) == _lib.ERR_LL
    assert L.spiht_encode_bound(1, 6, 8, 4, 2, 5, 0, C.byref(b)) == _lib.ERR_SHAPE
    assert L.spiht_wavelet_id(b"bior2.2") >= 0 and L.spiht_wavelet_id(b"nope") < 0
    assert L.spiht_mode_id(b"reflect") == 0 and L.spiht_mode_id(b"smooth") == 5 and L.spiht_mode_id(b"periodization") == 8 and L.spiht_mode_id(b"nope") < 0


def test_geometry_matches_golden_pywt_shapes():
    from spiht_amd.spiht_wrapper import SpihtSettings, get_slices_and_h_w
    w = np.load(os.path.join(ROOT, "tests", "golden", "wrapper_pywt.npz"))
    names = ["bior2.2", "bior4.4", "bior6.8", "haar"]
    for row in w["geometry"]:
        H, W, wi, lv, llh, llw, eh, ew, nlev = [int(v) for v in row]
        slices, gh, gw = get_slices_and_h_w(H, W, SpihtSettings(wavelet=names[wi]), None if lv < 0 else lv)
        assert (slices[0][1].stop, slices[0][2].stop, gh, gw, len(slices) - 1) == (llh, llw, eh, ew, nlev)


def test_argument_checking_mirrors_pyo3():
    import spiht_amd
    from spiht_amd.spiht import PanicException
    with pytest.raises(TypeError):
        spiht_amd.encode([[1, 2], [3, 4]], 2, 2, 10)
    with pytest.raises(TypeError):
        spiht_amd.encode(np.zeros((1, 8, 8), np.int64), 2, 2, 10)
    with pytest.raises(TypeError):
        spiht_amd.encode(np.zeros((8, 8), np.int32), 2, 2, 10)
    with pytest.raises(OverflowError):
        spiht_amd.encode(np.zeros((1, 8, 8), np.int32), 2, 2, -1)
    with pytest.raises(TypeError):
        spiht_amd.decode("abc", 3, 1, 8, 8, 2, 2)
    with pytest.raises(PanicException):
        spiht_amd.decode(b"\x00", 3, 1, 8, 8, 1, 2)
    with pytest.raises(ValueError):
        spiht_amd.encode_image(np.zeros((8, 8)))
    with pytest.raises(ValueError):
        spiht_amd.decode_image(spiht_amd.EncodingResult(b"", 8, 8, 1, 0, None, "0.0.1"), spiht_amd.SpihtSettings())


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must raise, not compute."""
    import spiht_amd
    from spiht_amd import _lib
    try:
        _lib.Context(0).close()
        pytest.skip("a GPU is present")
    except _lib.SpihtHipError:
        pass
    with pytest.raises(_lib.SpihtHipError):
        spiht_amd.encode(np.ones((1, 8, 8), np.int32), 2, 2, 100)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "spiht_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
