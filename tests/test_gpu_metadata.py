"""GPU parity tests of decode_with_metadata (src/lib.rs:47-56 -> encoder_decoder.rs:631-841) through the C ABI
against the CPU oracle's restatement: decoded array and every metadata row bit-exact.

The reference holds no expected metadata rows (its tests only compare the decoded array), so the rows are
pinned to the oracle's restatement of the source only -- "parity unpinned" in DESIGN.md's sense."""
import numpy as np
import pytest

from conftest import synth_coeffs

pytestmark = pytest.mark.gpu
UNLIMITED = 99999999999999999


def tree_generations(h, w, lh, lw):
    best = 1
    for r, c in ((0, lw), (lh, 0)):
        t = 1
        while 2 * (r << (t - 1)) + 1 < h and 2 * (c << (t - 1)) + 1 < w:
            t += 1
        best = max(best, t)
    return best


def nominal_slices(lh, lw, levels):
    """Mallat-like slices for an arbitrary (h, w, ll) tree: level t band = [ll*2^(t-1), ll*2^t)"""
    top = [(0, lh), (0, lw)]
    other = []
    for t in range(1, levels + 1):
        a_h, a_w = lh << (t - 1), lw << (t - 1)
        other.append([
            [(a_h, 2 * a_h), (0, a_w)],        # da
            [(0, a_h), (a_w, 2 * a_w)],        # ad
            [(a_h, 2 * a_h), (a_w, 2 * a_w)],  # dd
        ])
    return top, other


def _check(O, d, n, shape, lh, lw, top, other, check_rec=True):
    import spiht_amd
    c, h, w = shape
    r_ref, m_ref = O.decode_with_metadata(d, n, c, h, w, lh, lw, top, other)
    r, m = spiht_amd.spiht.decode_with_metadata(d, n, c, h, w, lh, lw, top, other)
    assert m.dtype == np.int32 and m.shape == (8 * len(d) + 1, 8) and m.flags.c_contiguous
    assert r.dtype == np.int32 and r.shape == (c, h, w)
    if not np.array_equal(m, m_ref):
        bad = np.argwhere((m != m_ref).any(axis=1))[:, 0]
        q = int(bad[0])
        raise AssertionError("metadata differs in %d rows of %d, first row %d: got %s want %s (stream %d bytes)"
                             % (len(bad), len(m), q, m[q].tolist(), m_ref[q].tolist(), len(d)))
    if check_rec:
        assert np.array_equal(r, r_ref)
    return r, m


def test_like_rust_tests(oracle):
    """encoder_decoder.rs:929-966: Slices::new_basic(4, 32, 32) / (2, 8, 8); the decoded array is lossless"""
    import spiht_amd
    rng = np.random.default_rng(42)
    for (c, h, w), lv in (((4, 32, 32), 4), ((1, 8, 8), 2)):
        top = [(0, h // 2), (0, w // 2)]  # new_basic: every level gets the same three half-size slices (:529-578)
        other = [[[(0, h // 2), (w // 2, w)], [(h // 2, h), (0, w // 2)], [(h // 2, h), (w // 2, w)]]] * lv
        for _ in range(4):
            x = rng.normal(0, 16, (c, h, w)).astype(np.int32)
            d, n = spiht_amd.encode(x, 2, 2, 10000000)
            r, m = _check(oracle, d, n, (c, h, w), 2, 2, top, other)
            assert np.array_equal(r, x)
            assert set(np.unique(m[:, 0]).tolist()) <= set(range(7))


SHAPES = [  # c, h, w, ll_h, ll_w  (odd LL sizes -> duplicated tree nodes under two different filters)
    (1, 8, 8, 2, 2), (2, 16, 16, 4, 4), (3, 13, 17, 3, 5), (1, 21, 19, 5, 3), (3, 24, 40, 3, 5), (2, 11, 23, 4, 6),
    (1, 4, 4, 2, 2), (3, 33, 29, 6, 5), (1, 6, 6, 3, 3), (3, 70, 100, 5, 7), (1, 129, 65, 9, 5), (1, 300, 200, 3, 2),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_metadata_parity(oracle, shape):
    import spiht_amd
    c, h, w, lh, lw = shape
    G = tree_generations(h, w, lh, lw)
    top, other = nominal_slices(lh, lw, G)
    x = synth_coeffs(h * 1000 + w, c, h, w, lh, lw, scale=60.0)
    d, n = spiht_amd.encode(x, lh, lw, UNLIMITED)
    _check(oracle, d, n, (c, h, w), lh, lw, top, other)
    # every way a stream can end: inside each kind of operation, on byte boundaries (pad bits are data, Q9)
    for nb in sorted(set(list(range(0, min(len(d), 40))) + [len(d) // 3, len(d) // 2, len(d) - 1])):
        if 0 <= nb <= len(d):
            _check(oracle, d[:nb], n, (c, h, w), lh, lw, top, other)
    # more bytes than the coder needs: the rows of bits that are never read stay zero
    r, m = _check(oracle, d + b"\xa5" * 9, n, (c, h, w), lh, lw, top, other)
    assert not m[-8:].any()
    # a deeper `level` than the tree is fine (depth counts down from level)
    top2, other2 = nominal_slices(lh, lw, G + 2)
    _check(oracle, d[: len(d) // 2], n, (c, h, w), lh, lw, top2, other2)


@pytest.mark.parametrize("shape", [(1, 16, 16, 2, 2), (2, 32, 48, 4, 6), (3, 24, 24, 6, 6)])
def test_metadata_arbitrary_bytes(oracle, shape):
    """not encoder output: the running value column replays writes in stream order whatever the bits are"""
    c, h, w, lh, lw = shape
    G = tree_generations(h, w, lh, lw)
    top, other = nominal_slices(lh, lw, G)
    rng = np.random.default_rng(7)
    for nb in (1, 7, 64, 500):
        for n in (0, 3, 9):
            d = rng.integers(0, 256, nb, dtype=np.uint8).tobytes()
            _check(oracle, d, n, (c, h, w), lh, lw, top, other)


def test_metadata_errors(oracle):
    import spiht_amd
    x = synth_coeffs(3, 1, 32, 32, 2, 2, scale=40.0)
    d, n = spiht_amd.encode(x, 2, 2, UNLIMITED)
    top, other = nominal_slices(2, 2, 3)  # the tree has 4 generations
    with pytest.raises(spiht_amd.spiht.PanicException):
        spiht_amd.spiht.decode_with_metadata(d, n, 1, 32, 32, 2, 2, top, other)
    with pytest.raises(oracle.OraclePanic):
        oracle.decode_with_metadata(d, n, 1, 32, 32, 2, 2, top, other)
    top, other = nominal_slices(2, 2, 4)
    with pytest.raises(spiht_amd.spiht.PanicException):
        spiht_amd.spiht.decode_with_metadata(d, n, 1, 32, 32, 1, 2, top, other)
    with pytest.raises(TypeError):
        spiht_amd.spiht.decode_with_metadata(d, n, 1, 32, 32, 2, 2, [(None, 2), (0, 2)], other)
    r, m = spiht_amd.spiht.decode_with_metadata(b"", n, 1, 32, 32, 2, 2, top, other)
    assert m.shape == (1, 8) and m[0].tolist() == [0, -100000, -100000, 0, 0, 4, n, 0] and not r.any()


def test_wrapper_return_metadata(oracle):
    """spiht_wrapper.py:192-216, 232-250 and tests/test_spiht.py:19-28"""
    import spiht_amd
    from spiht_amd import SpihtSettings, encode_image, decode_image
    rng = np.random.default_rng(5)
    img = rng.random((3, 72, 100))
    for settings, level in ((SpihtSettings(), None), (SpihtSettings(wavelet="bior4.4", quantization_scale=20.0), 2)):
        enc = encode_image(img, settings, level=level, max_bits=20000)
        im, meta = decode_image(enc, settings, return_metadata=True)
        im2 = decode_image(enc, settings, return_metadata=False)
        assert np.allclose(im, im2)
        # the same call against the oracle
        from spiht_amd.spiht_wrapper import get_slices_and_h_w
        slices, enc_h, enc_w = get_slices_and_h_w(72, 100, settings, level)
        ll_h, ll_w = slices[0][1].stop, slices[0][2].stop
        top = [(0, ll_h), (0, ll_w)]
        other = [[[(s[k][1].start or 0, s[k][1].stop), (s[k][2].start or 0, s[k][2].stop)] for k in ("da", "ad", "dd")]
                 for s in slices[1:]]
        r_ref, m_ref = oracle.decode_with_metadata(enc.encoded_bytes, enc.max_n, 3, enc_h, enc_w, ll_h, ll_w, top, other)
        assert meta.shape == (8 * len(enc.encoded_bytes) + 1, 8)
        assert np.array_equal(meta, m_ref)


def test_metadata_random_sweep(oracle):
    """seeded random geometries, budgets and truncation points: every metadata row against the oracle"""
    import os
    import spiht_amd
    rng = np.random.default_rng(777)
    for case in range(int(os.environ.get("SPIHT_SWEEP_N", "40"))):
        c = int(rng.integers(1, 4))
        lh, lw = int(rng.integers(2, 8)), int(rng.integers(2, 8))
        need_h = 2 * lh if lh % 2 == 0 else 2 * lh - 1
        need_w = 2 * lw if lw % 2 == 0 else 2 * lw - 1
        h, w = int(rng.integers(need_h, need_h + 50)), int(rng.integers(need_w, need_w + 50))
        G = tree_generations(h, w, lh, lw)
        top, other = nominal_slices(lh, lw, G + int(rng.integers(0, 2)))
        x = synth_coeffs(int(rng.integers(1 << 30)), c, h, w, lh, lw, scale=float(10 ** rng.uniform(0.5, 3.5)))
        mb = [int(rng.integers(1, 300)), int(rng.integers(300, 9000)), UNLIMITED][case % 3]
        d, n = spiht_amd.encode(x, lh, lw, mb)
        if len(d) > 1 and case % 2:
            d = d[: int(rng.integers(0, len(d)))]
        try:
            _check(oracle, d, n, (c, h, w), lh, lw, top, other)
        except AssertionError as e:
            raise AssertionError("case %d: c=%d h=%d w=%d ll=%dx%d max_bits=%d bytes=%d: %s" % (case, c, h, w, lh, lw, mb, len(d), e))
