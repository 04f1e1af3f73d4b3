"""CPU tests of the pure-Python/numpy path (counterpart of the reference's legacy spiht/spiht_py.py, BASELINE
config 1): it implements the Rust semantics, so it must agree with the oracle bit for bit."""
import os

import numpy as np

from conftest import synth_image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_list_coder_matches_oracle_on_fixture_inputs(oracle):
    from spiht_amd.spiht_py import decode_py, encode_py
    g = np.load(os.path.join(GOLD, "spiht_py_loops.npz"))
    for k in range(int(g["ncases"])):
        p = "case%02d_" % k
        arr, lh, lw, mb = g[p + "arr"], int(g[p + "ll_h"]), int(g[p + "ll_w"]), int(g[p + "max_bits"])
        data, mn = encode_py(arr, lh, lw, mb)
        ref, ref_n = oracle.encode(arr, lh, lw, mb)
        assert (data, mn) == (ref, ref_n), k
        c, h, w = arr.shape
        assert np.array_equal(decode_py(data, mn, c, h, w, lh, lw), oracle.decode(ref, ref_n, c, h, w, lh, lw))
        full, n2 = encode_py(arr, lh, lw, 10**12)
        assert (full, n2) == oracle.encode(arr, lh, lw, 10**12)


def test_appendix_c_vector(oracle):
    from spiht_amd.spiht_py import decode_py, encode_py
    x = np.array([[[26, 6, 13, 10], [-7, 7, 6, 4], [4, -4, 4, -3], [2, -2, -2, 0]]], np.int32)
    data, mn = encode_py(x, 2, 2, 10**12)
    assert (data.hex(), mn) == ("03f8f0fec7a12b7d200302", 4)
    assert np.array_equal(decode_py(data, mn, 1, 4, 4, 2, 2), x)


def test_config1_512_gray_python_path(oracle):
    """BASELINE config 1: single 512x512 grayscale, bior2.2 level 5, 0.5 bpp, pure-Python path"""
    from spiht_amd.spiht_py import decode_image_py, encode_image_py
    img = synth_image(1000, 1, 512, 512)
    mb = int(512 * 512 * 0.5)
    enc = encode_image_py(img, 'bior2.2', 5, mb, 50, 'reflect')
    ref_bytes, ref_n, g = oracle.encode_image(img, "bior2.2", "reflect", 5, 50.0, None, mb)
    assert (g["enc_h"], g["ll_h"]) == (533, 20)
    assert enc.max_n == ref_n and enc.encoded_bytes == ref_bytes and len(enc.encoded_bytes) == 16384
    dec = decode_image_py(enc)
    ref = oracle.decode_image(ref_bytes, ref_n, 1, 512, 512, "bior2.2", 5, 50.0, None)
    assert dec.shape == ref.shape and np.abs(dec - ref).max() < 1e-12
    assert np.abs(dec - img).mean() < 0.05


def test_odd_sizes_and_level_none(oracle):
    from spiht_amd.spiht_py import decode_image_py, encode_image_py
    img = synth_image(5, 3, 45, 61)
    enc = encode_image_py(img, 'bior2.2', None, 4000, 50, 'reflect')
    ref_bytes, ref_n, _ = oracle.encode_image(img, "bior2.2", "reflect", None, 50.0, None, 4000)
    assert (enc.encoded_bytes, enc.max_n) == (ref_bytes, ref_n)
    dec = decode_image_py(enc)
    ref = oracle.decode_image(ref_bytes, ref_n, 3, 45, 61, "bior2.2", None, 50.0, None)
    assert dec.shape == ref.shape and np.abs(dec - ref).max() < 1e-12
