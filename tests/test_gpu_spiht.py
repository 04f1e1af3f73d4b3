"""GPU parity tests of the SPIHT list coder (through the C ABI) against the CPU oracle: bit-exact streams,
max_n and decoded int32 arrays."""
import os

import numpy as np
import pytest

from conftest import synth_coeffs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
UNLIMITED = 99999999999999999


def _check_encode(O, x, lh, lw, mb):
    import spiht_amd
    d_ref, n_ref, nb_ref = O.encode_nbits(x, lh, lw, mb)
    d, n = spiht_amd.encode(x, lh, lw, mb)
    assert n == n_ref
    assert len(d) == len(d_ref) == (nb_ref + 7) // 8
    if d != d_ref:
        a, b = np.frombuffer(d, np.uint8), np.frombuffer(d_ref, np.uint8)
        first = int(np.nonzero(a != b)[0][0])
        raise AssertionError("stream differs at byte %d of %d (bit budget %d)" % (first, len(d), mb))
    return d, n


def _check_decode(O, d, n, shape, lh, lw):
    import spiht_amd
    c, h, w = shape
    r_ref = O.decode(d, n, c, h, w, lh, lw)
    r = spiht_amd.decode(d, n, c, h, w, lh, lw)
    assert r.dtype == np.int32 and r.shape == (c, h, w) and r.flags.c_contiguous
    if not np.array_equal(r, r_ref):
        bad = np.argwhere(r != r_ref)
        raise AssertionError("decoded array differs at %d cells, first %s: got %d want %d (stream %d bytes)"
                             % (len(bad), bad[0], r[tuple(bad[0])], r_ref[tuple(bad[0])], len(d)))
    return r


def test_known_answer_vectors(oracle):
    import spiht_amd
    x = 32 * np.ones((1, 16, 16), np.int32)
    d, n = spiht_amd.encode(x, 2, 2, 10000)
    assert n == 5 and d == b"\xff" * 73 + b"\x3f" + b"\x00" * 160  # encoder_decoder.rs:865-875 / App. C
    assert np.array_equal(spiht_amd.decode(d, n, 1, 16, 16, 2, 2), x)
    x = np.array([[[26, 6, 13, 10], [-7, 7, 6, 4], [4, -4, 4, -3], [2, -2, -2, 0]]], np.int32)
    d, n = spiht_amd.encode(x, 2, 2, UNLIMITED)
    assert (d.hex(), n) == ("03f8f0fec7a12b7d200302", 4)
    assert np.array_equal(spiht_amd.decode(d, n, 1, 4, 4, 2, 2), x)
    for mb, row0 in [(8, [24, 0, 0, 0]), (16, [24, 0, 12, 12]), (24, [24, 6, 12, 12])]:
        d, n = spiht_amd.encode(x, 2, 2, mb)
        assert spiht_amd.decode(d, n, 1, 4, 4, 2, 2)[0, 0].tolist() == row0


def test_roundtrip_like_rust_tests(oracle):
    """encoder_decoder.rs:878-985: lossless on dyadic shapes, ll = 2x2"""
    import spiht_amd
    x = 32 * np.ones((1, 16, 16), np.int32)
    x[:, 1::2, :] *= -1
    d, n = spiht_amd.encode(x, 2, 2, 10000)
    assert np.array_equal(spiht_amd.decode(d, n, 1, 16, 16, 2, 2), x)
    rng = np.random.default_rng(42)
    for shape in [(1, 8, 8), (4, 32, 32)]:
        for _ in range(6):
            a = rng.normal(0, 16, shape).astype(np.int32)
            d, n = _check_encode(oracle, a, 2, 2, 10000000)
            assert np.array_equal(spiht_amd.decode(d, n, *shape, 2, 2), a)


SHAPES = [  # c, h, w, ll_h, ll_w
    (1, 8, 8, 2, 2), (1, 16, 16, 2, 2), (2, 16, 16, 4, 4), (3, 13, 17, 3, 5), (1, 21, 19, 5, 3), (3, 24, 40, 3, 5),
    (1, 32, 32, 2, 2), (2, 11, 23, 4, 6), (1, 4, 4, 2, 2), (3, 33, 29, 6, 5), (1, 8, 8, 4, 4), (1, 6, 6, 3, 3),
    (1, 64, 64, 2, 2), (3, 70, 100, 5, 7), (1, 129, 65, 9, 5), (4, 48, 48, 6, 6), (1, 300, 200, 3, 2),
]


@pytest.mark.parametrize("shape", SHAPES)
def test_encode_decode_parity_small(oracle, shape):
    c, h, w, lh, lw = shape
    rng = np.random.default_rng(hash(shape) % 2**32)
    xs = [synth_coeffs(5, c, h, w, lh, lw, scale=400.0), rng.normal(0, 16, (c, h, w)).astype(np.int32)]
    z = np.zeros((c, h, w), np.int32)
    z[0, h - 1, w // 2] = -3
    xs.append(z)
    for x in xs:
        _, _, total = oracle.encode_nbits(x, lh, lw, UNLIMITED)
        budgets = [UNLIMITED, 0, 1, 2, 7, 8, 9, total, total - 1, total + 5] + \
            [int(v) for v in rng.integers(1, max(2, total), 6)]
        for mb in budgets:
            if mb < 0:
                continue
            d, n = _check_encode(oracle, x, lh, lw, mb)
            _check_decode(oracle, d, n, (c, h, w), lh, lw)


def test_all_zero_and_tiny(oracle):
    import spiht_amd
    x = np.zeros((2, 8, 8), np.int32)
    d, n = _check_encode(oracle, x, 2, 2, UNLIMITED)
    _check_decode(oracle, d, n, (2, 8, 8), 2, 2)
    assert np.array_equal(spiht_amd.decode(b"", 5, 1, 8, 8, 2, 2), np.zeros((1, 8, 8), np.int32))


def test_strided_views(oracle):
    rng = np.random.default_rng(1)
    big = rng.normal(0, 40, (3, 40, 64)).astype(np.int32)
    for view in [big[:, ::2, :], big[:, :, ::2], big[::2], big.transpose(0, 2, 1), big[:, ::-1, :]]:
        c, h, w = view.shape
        _check_encode(oracle, view, 2, 2, 5000)


def test_prefix_truncation_and_pad_bits(oracle):
    """make_gif.py:46-55 decodes byte prefixes; lib.rs:38 makes the pad bits data.
    13x17 / ll 3x5 has duplicated tree nodes (Q4)."""
    for (c, h, w, lh, lw) in [(3, 13, 17, 3, 5), (1, 32, 32, 2, 2), (2, 26, 38, 13, 19)]:
        x = synth_coeffs(9, c, h, w, lh, lw, scale=500.0)
        full, n, total = oracle.encode_nbits(x, lh, lw, UNLIMITED)
        for nb in list(range(0, min(len(full), 40))) + list(range(max(0, len(full) - 12), len(full) + 1)):
            _check_decode(oracle, full[:nb], n, (c, h, w), lh, lw)
        # every bit budget in a window: the cut lands on sig bits, sign bits, refinement bits
        for mb in list(range(1, 130)) + list(range(max(1, total - 70), total + 2)):
            d, n2 = _check_encode(oracle, x, lh, lw, mb)
            _check_decode(oracle, d, n2, (c, h, w), lh, lw)


def test_decode_arbitrary_bytes(oracle):
    """Any byte string is a valid input (lib.rs:38 hands every bit to the decoder).  On trees with duplicated nodes
    (odd ll_h / ll_w) two or three list entries write the same cell, and for bytes no encoder produced
    their operations differ: the cell must end as the reference's sequential writes leave it
    (encoder_decoder.rs:352-451)."""
    rng = np.random.default_rng(5)
    for (c, h, w, lh, lw) in [(1, 16, 16, 2, 2), (2, 24, 40, 4, 6), (3, 13, 17, 3, 5), (2, 26, 38, 13, 19),
                              (1, 23, 31, 5, 6), (2, 40, 37, 6, 9), (3, 64, 96, 3, 3)]:
        for ln in [1, 3, 17, 200, 1500]:
            for n in (9, 3, 0):
                d = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
                _check_decode(oracle, d, n, (c, h, w), lh, lw)
        # streams dense in ones (many significance hits: duplicated cells get written by all their list entries)
        for ln in [40, 600]:
            d = (rng.integers(0, 256, ln, dtype=np.uint8) | rng.integers(0, 256, ln, dtype=np.uint8)).astype(np.uint8).tobytes()
            _check_decode(oracle, d, 6, (c, h, w), lh, lw)


def test_decode_long_stretches_of_one_entry_type(oracle):
    """The LIS walk looks a 64-bit window up instead of hopping through it when the 64 queue entries it can meet are all
    type A (decode.hip: DecShared::tabfm, the helper's walks from the nine entry points), and takes runs of such windows in
    a loop of its own.  Medium and large arrays whose queues have long stretches of one type (profiles/
    r04_lis_type_runs.txt), under byte strings no encoder made, with few, some, half and mostly ones -- every entry point,
    the merges of the helper's walks, windows with no fired entry and windows full of them, runs that end at the end of the
    queue, of a chunk block (4096 entries) and of the stream -- and under encoder streams cut at many lengths; both widths
    of the decoder."""
    from spiht_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(31)
    geoms = [(3, 300, 420, 5, 7), (1, 512, 512, 4, 4), (3, 345, 287, 11, 9), (2, 640, 520, 10, 9)]
    try:
        for waves in (12, 8):
            ctx.set_decoder_waves(waves)
            for gi, (c, h, w, lh, lw) in enumerate(geoms):
                for p1 in (0.02, 0.08, 0.25, 0.5, 0.8):
                    ln = int(rng.integers(3000, 60000))
                    bits = (rng.random(8 * ln) < p1).astype(np.uint8)
                    d = np.packbits(bits, bitorder="little").tobytes()
                    for n in (11, 5):
                        _check_decode(oracle, d, n, (c, h, w), lh, lw)
                x = synth_coeffs(50 + gi, c, h, w, lh, lw, scale=float(10 ** rng.uniform(2.5, 4.5)))
                d, n = oracle.encode(x, lh, lw, min(8 * c * h * w, 600000))
                for cut in sorted({len(d), len(d) // 2, len(d) // 3 + 1, len(d) // 7, int(rng.integers(1, len(d)))}):
                    _check_decode(oracle, d[:cut], n, (c, h, w), lh, lw)
    finally:
        ctx.set_decoder_waves(12)


def test_decoder_with_eight_wavefronts(oracle):
    """The 8-wavefront build of the decoder (spiht_ctx_set_decoder_waves: the list-coding contexts of the pipelined
    schedule use it) decodes what the 12-wavefront one does: encoder streams, their prefixes, arbitrary bytes on trees
    with duplicated nodes, and the metadata rows."""
    import spiht_amd
    from spiht_amd import _lib
    ctx = _lib.default_context()
    with pytest.raises(ValueError):
        ctx.set_decoder_waves(10)
    ctx.set_decoder_waves(8)
    try:
        rng = np.random.default_rng(23)
        for (c, h, w, lh, lw) in [(1, 16, 16, 2, 2), (3, 13, 17, 3, 5), (2, 26, 38, 13, 19), (3, 293, 501, 13, 19), (1, 533, 533, 20, 20)]:
            x = synth_coeffs(7, c, h, w, lh, lw)
            d, n = oracle.encode(x, lh, lw, min(200000, 8 * c * h * w))
            for cut in (len(d), len(d) // 2, 7, 1):
                _check_decode(oracle, d[:cut], n, (c, h, w), lh, lw)
            for ln in (3, 200, 1500):
                _check_decode(oracle, rng.integers(0, 256, ln, dtype=np.uint8).tobytes(), 6, (c, h, w), lh, lw)
        c, h, w, lh, lw = 3, 40, 56, 5, 7
        x = synth_coeffs(9, c, h, w, lh, lw)
        d, n = oracle.encode(x, lh, lw, 6000)
        top = [(0, lh), (0, lw)]
        other = [[[(lh, 2 * lh), (0, lw)], [(0, lh), (lw, 2 * lw)], [(lh, 2 * lh), (lw, 2 * lw)]],
                 [[(2 * lh, 4 * lh), (0, 2 * lw)], [(0, 2 * lh), (2 * lw, 4 * lw)], [(2 * lh, 4 * lh), (2 * lw, 4 * lw)]],
                 [[(4 * lh, 8 * lh), (0, 4 * lw)], [(0, 4 * lh), (4 * lw, 8 * lw)], [(4 * lh, 8 * lh), (4 * lw, 8 * lw)]]]
        rec, meta = spiht_amd.spiht.decode_with_metadata(d, n, c, h, w, lh, lw, top, other)
        rec_o, meta_o = oracle.decode_with_metadata(d, n, c, h, w, lh, lw, top, other)
        assert np.array_equal(rec, rec_o) and np.array_equal(meta, meta_o)
    finally:
        ctx.set_decoder_waves(12)


def test_decode_budgets_one_walk(oracle):
    """spiht_decode_budgets_i32: K bit budgets of one stream from one walk equal K separate decodes of the bit prefixes
    (oracle: orc_decode_bits on bits[:b]) -- encoder streams and arbitrary bytes, trees with duplicated nodes included,
    budgets at byte boundaries, inside a token (between a significance bit and its sign bit) and past the end."""
    import spiht_amd
    from spiht_amd.spiht import decode_budgets
    rng = np.random.default_rng(31)
    for (c, h, w, lh, lw) in [(1, 16, 16, 2, 2), (3, 13, 17, 3, 5), (2, 26, 38, 13, 19), (3, 96, 136, 6, 9), (3, 293, 501, 13, 19)]:
        x = synth_coeffs(3, c, h, w, lh, lw)
        d, n = oracle.encode(x, lh, lw, min(60000, 8 * c * h * w))
        streams = [(d, n), (rng.integers(0, 256, 700, dtype=np.uint8).tobytes(), 6)]
        for (data, nn) in streams:
            bits = oracle.bytes_to_bits(data)
            nb = len(bits)
            bud = sorted(set([0, 1, 2, 3, 5, 8, 9, 17, 64, nb // 3, nb // 2, nb // 2 + 1, nb - 1, nb, nb + 40]
                             + [int(v) for v in rng.integers(0, nb + 1, 12)]))
            got = decode_budgets(data, nn, c, h, w, lh, lw, bud)
            assert got.shape == (len(bud), c, h, w) and got.dtype == np.int32
            for k, b in enumerate(bud):
                ref = oracle.decode_bits(bits[:b], nn, c, h, w, lh, lw)
                if not np.array_equal(got[k], ref):
                    bad = np.argwhere(got[k] != ref)
                    raise AssertionError("budget %d of %d bits, geometry %s: %d cells differ, first %s: got %d want %d"
                                         % (b, nb, (c, h, w, lh, lw), len(bad), bad[0], got[k][tuple(bad[0])], ref[tuple(bad[0])]))
            # the last budget is the plain decode
            assert np.array_equal(got[-1], spiht_amd.decode(data, nn, c, h, w, lh, lw))
    with pytest.raises(ValueError):
        decode_budgets(d, n, c, h, w, lh, lw, [100, 50])  # not ascending
    assert decode_budgets(d, n, c, h, w, lh, lw, []).shape == (0, c, h, w)


def test_decode_bit_flipped_stream_odd_ll(oracle):
    """A damaged encoder stream on the geometry class of BASELINE config 2 (ll 13x19, both odd): after the first flipped
    bit the decoder walks a different path than the encoder did and the duplicated cells' list entries diverge."""
    c, h, w, lh, lw = 3, 293, 501, 13, 19
    x = synth_coeffs(42, c, h, w, lh, lw)
    d, n = oracle.encode(x, lh, lw, 120000)
    rng = np.random.default_rng(11)
    for trial in range(6):
        b = bytearray(d)
        for pos in rng.integers(0, len(b) * 8, 1 + 3 * trial):
            b[int(pos) >> 3] ^= 1 << (int(pos) & 7)
        _check_decode(oracle, bytes(b), n, (c, h, w), lh, lw)
        _check_decode(oracle, bytes(b[:len(b) // (trial + 1)]), n, (c, h, w), lh, lw)


PYR_SHAPES = [(1, 16, 16, 2, 2), (3, 13, 17, 3, 5), (2, 26, 38, 13, 19), (1, 23, 31, 5, 6), (2, 40, 37, 6, 9),
              (1, 64, 64, 2, 2), (3, 293, 501, 13, 19), (1, 533, 533, 20, 20)]


@pytest.mark.parametrize("shape", PYR_SHAPES)
def test_pyramid_matches_oracle(oracle, shape):
    """The D / L significance pyramid (csrc/pyramid.hip) against the recursion it replaces, element by element:
    is_set_sig / is_l_sig (encoder_decoder.rs:78-121) evaluated per node and per plane by the oracle.  Only nodes with
    offspring carry a code (the others are never looked up)."""
    import ctypes as C
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    c, h, w, lh, lw = shape
    B = 2
    xs = np.stack([synth_coeffs(77 + b, c, h, w, lh, lw) for b in range(B)])
    xs[1, :, h // 2:, :] = 0            # empty sets: code 0
    xs[1, 0, h - 1, w - 1] = -(1 << 20)  # a lone large value in the last row / column (unreachable when h or w is odd)
    xs[1, 0, (h - 1) // 2, (w - 1) // 2] = 1 << 17
    L, ctx, vp = _lib.lib(), _lib.default_context(), C.c_void_p
    n = c * h * w
    d_x = DeviceArray(ctx, (B, n), np.int32)
    d_dm, d_lm, d_ma = DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B,), np.uint32)
    d_x.upload(xs.reshape(B, n))
    ctx.memset(d_dm.ptr, 0xEE, d_dm.nbytes)
    ctx.memset(d_lm.ptr, 0xEE, d_lm.nbytes)
    _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(d_x.ptr), B, c, h, w, lh, lw, vp(d_dm.ptr), vp(d_lm.ptr), vp(d_ma.ptr)))
    ctx.synchronize()
    dm, lm, ma = d_dm.download().reshape(B, c, h, w), d_lm.download().reshape(B, c, h, w), d_ma.download()
    for b in range(B):
        d_ref, l_ref, has = oracle.set_codes(xs[b], lh, lw)
        assert int(ma[b]) == int(np.abs(xs[b].astype(np.int64)).max())
        # L is looked up only for type-B entries: nodes with grand-offspring by the reference's raw-coordinate rule
        # (has_descendents_past_offspring, encoder_decoder.rs:7-12, :258) -- which it applies to root-block nodes too
        I, J = np.arange(h)[:, None], np.arange(w)[None, :]
        b_entry = (((4 * I + 3 < h) & (4 * J + 3 < w)))[None]
        for name, got, ref, where in (("D", dm[b], d_ref, has), ("L", lm[b], l_ref, has & b_entry)):
            bad = np.argwhere((got != ref) & where)
            assert len(bad) == 0, "%s code differs at %d nodes, first %s: got %d want %d" % (
                name, len(bad), bad[0], got[tuple(bad[0])], ref[tuple(bad[0])])


def test_golden_python_twin_inputs_rust_rule(oracle):
    """The fixture inputs captured from spiht_py.py, coded with the Rust rule on GPU == oracle."""
    g = np.load(os.path.join(GOLD, "spiht_py_loops.npz"))
    for k in range(int(g["ncases"])):
        p = "case%02d_" % k
        arr, lh, lw, mb = g[p + "arr"], int(g[p + "ll_h"]), int(g[p + "ll_w"]), int(g[p + "max_bits"])
        d, n = _check_encode(oracle, arr, lh, lw, mb)
        _check_decode(oracle, d, n, arr.shape, lh, lw)


def test_panics_and_errors():
    import spiht_amd
    from spiht_amd.spiht import PanicException
    x = np.ones((1, 8, 8), np.int32)
    with pytest.raises(PanicException):
        spiht_amd.encode(x, 1, 2, 100)
    with pytest.raises(PanicException):
        spiht_amd.encode(x, 2, 1, 100)
    with pytest.raises(PanicException):
        spiht_amd.encode(np.ones((1, 6, 8), np.int32), 4, 2, 100)  # offspring outside the array
    with pytest.raises(PanicException):
        spiht_amd.encode(np.ones((0, 8, 8), np.int32), 2, 2, 100)
    with pytest.raises(ValueError):
        spiht_amd.encode(np.full((1, 8, 8), 2**30, np.int32), 2, 2, 100)


def test_medium_realistic(oracle):
    """A 3x293x501 array with ll 13x19 (the odd/odd geometry of the 1080p config, Q3+Q4)."""
    x = synth_coeffs(42, 3, 293, 501, 13, 19)
    for mb in [12345, 100000, UNLIMITED]:
        d, n = _check_encode(oracle, x, 13, 19, mb)
        _check_decode(oracle, d, n, x.shape, 13, 19)
    full, n, total = oracle.encode_nbits(x, 13, 19, UNLIMITED)
    for nb in [1000, 12345 // 8, len(full) // 3, len(full) - 1]:
        _check_decode(oracle, full[:nb], n, x.shape, 13, 19)


@pytest.mark.parametrize("wide", [False, True])
def test_random_geometries_and_budgets(oracle, wide):
    """seeded random sweep: channels, odd / even sizes, LL blocks of every parity, magnitudes from 0 to 2^29, budgets
    from a few bits to unlimited -- stream, max_n and decoded array against the oracle, decode of random prefixes too.
    wide: the same cases through the several-CUs-per-image encoder (forced on for these small arrays, 2 to 9 workgroups)"""
    import spiht_amd
    from spiht_amd import _lib
    rng = np.random.default_rng(20261004)
    done = 0
    n_cases = int(os.environ.get("SPIHT_SWEEP_N", "48"))  # larger sweeps on demand
    ctx = _lib.default_context()
    ctx.set_option("wide_encode", 2 if wide else 1)
    try:
        _random_cases(oracle, rng, n_cases, ctx if wide else None)
    finally:
        ctx.set_option("wide_encode", 1)
        ctx.set_option("wide_groups", 0)
        ctx.set_option("wide_solo", 24576)


def _random_cases(oracle, rng, n_cases, wide_ctx):
    done = 0
    while done < n_cases:
        if wide_ctx is not None:
            wide_ctx.set_option("wide_groups", 2 + done % 8)
            wide_ctx.set_option("wide_solo", [0, 300, 4000][done % 3])  # (the group takes over from the first plane / early / late)
        c = int(rng.integers(1, 5))
        lh, lw = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        need_h = 2 * lh if lh % 2 == 0 else 2 * lh - 1
        need_w = 2 * lw if lw % 2 == 0 else 2 * lw - 1
        h, w = int(rng.integers(need_h, need_h + 70)), int(rng.integers(need_w, need_w + 70))
        kind = done % 4
        if kind == 0:
            x = synth_coeffs(int(rng.integers(1 << 30)), c, h, w, lh, lw, scale=float(10 ** rng.uniform(0.5, 4.5)))
        elif kind == 1:
            x = rng.integers(-3, 4, (c, h, w)).astype(np.int32)                      # tiny values, many zeros
        elif kind == 2:
            x = np.zeros((c, h, w), np.int32)
            idx = rng.integers(0, x.size, 12)
            x.reshape(-1)[idx] = rng.integers(-(1 << 29), 1 << 29, 12)               # a few huge coefficients
        else:
            x = (rng.laplace(0, 1, (c, h, w)) * 200).astype(np.int32)
            x[:, h // 2:, :] = 0                                                     # an empty half
        mb = [int(rng.integers(1, 400)), int(rng.integers(400, 20000)), UNLIMITED][done % 3]
        try:
            d, n = _check_encode(oracle, x, lh, lw, mb)
            _check_decode(oracle, d, n, (c, h, w), lh, lw)
            if len(d) > 2:
                _check_decode(oracle, d[: int(rng.integers(1, len(d)))], n, (c, h, w), lh, lw)
        except Exception as e:
            raise AssertionError("case %d: c=%d h=%d w=%d ll=%dx%d kind=%d max_bits=%d max|x|=%d: %r"
                                 % (done, c, h, w, lh, lw, kind, mb, int(np.abs(x).max()), e))
        done += 1


def test_unscatter_lists_puts_the_zeros_back(oracle):
    """spiht_unscatter_lists_batch_i32 (include/spiht_hip.h, "two halves"): after it the decoder's output array is
    all zero again -- through the decoder's lists when they still exist, by zero-fill when another list-coding call
    used the context in between -- and the next decode into the same array is exact."""
    import ctypes as C
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    L = _lib.lib()
    ctx = _lib.Context(0)
    vp = C.c_void_p
    B, c, h, w, lh, lw, mb = 3, 2, 40, 56, 5, 7, 6000
    xs = np.stack([synth_coeffs(900 + b, c, h, w, lh, lw) for b in range(B)])
    n = c * h * w
    d_x, d_rec = DeviceArray(ctx, (B, n), np.int32), DeviceArray(ctx, (B, n), np.int32)
    d_dm, d_lm, d_ma = DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B,), np.uint32)
    slot = ((mb + 7) // 8 + 3) & ~3
    d_out, d_nb, d_mn, d_ny = (DeviceArray(ctx, (B, slot), np.uint8), DeviceArray(ctx, (B,), np.uint64),
                               DeviceArray(ctx, (B,), np.uint8), DeviceArray(ctx, (B,), np.uint64))
    d_x.upload(xs.reshape(B, n))

    def encode():
        _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(d_x.ptr), B, c, h, w, lh, lw, vp(d_dm.ptr), vp(d_lm.ptr), vp(d_ma.ptr)))
        _lib.check(L.spiht_encode_lists_batch_i32(ctx.handle, vp(d_x.ptr), vp(d_dm.ptr), vp(d_lm.ptr), vp(d_ma.ptr), B, c, h, w,
                                                  lh, lw, mb, vp(d_out.ptr), slot, vp(d_nb.ptr), vp(d_mn.ptr)))
        _lib.check(L.spiht_nbits_to_nbytes(ctx.handle, vp(d_nb.ptr), B, vp(d_ny.ptr)))

    def decode():
        _lib.check(L.spiht_decode_lists_batch_i32(ctx.handle, vp(d_out.ptr), slot, vp(d_ny.ptr), vp(d_mn.ptr), B, c, h, w, lh, lw,
                                                  vp(d_rec.ptr)))

    def unscatter():
        _lib.check(L.spiht_unscatter_lists_batch_i32(ctx.handle, vp(d_rec.ptr), B, c, h, w))

    ctx.memset(d_rec.ptr, 0, d_rec.nbytes)
    encode()
    ref = []
    for b in range(B):
        d, mn, nbits = oracle.encode_nbits(xs[b], lh, lw, mb)
        ref.append(oracle.decode(d, mn, c, h, w, lh, lw).reshape(-1))
    ref = np.stack(ref)
    for between in (False, True, False):
        decode()
        ctx.synchronize()
        assert np.array_equal(d_rec.download(), ref)
        if between:
            encode()  # the lists now belong to the encoder: the call has to fall back to a zero-fill
        unscatter()
        ctx.synchronize()
        assert not d_rec.download().any()


def test_wide_encoder_matches_oracle(oracle):
    """One image on several CUs (encode_wide.hip: chunks of a pass on a group of workgroups, decoupled look-back scan,
    group barriers between the passes; the planes with short lists by workgroup 0 alone): forced on for every size
    (option "wide_encode" = 2), with 2 to 40 workgroups per image -- streams equal to the oracle's, budgets that end inside
    every kind of pass, unlimited budgets, images smaller than one chunk and images of many chunks per pass, duplicated
    tree nodes (odd root blocks), and a small batch (several groups in one launch)."""
    import spiht_amd
    from spiht_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(77)
    try:
        ctx.set_option("wide_encode", 2)
        cases = [(3, 300, 420, 5, 7, 3000.0), (1, 512, 512, 4, 4, 20000.0), (3, 131, 203, 3, 5, 800.0), (2, 40, 56, 5, 7, 300.0),
                 (3, 345, 287, 11, 9, 60000.0), (4, 260, 260, 2, 2, 5000.0)]
        for k, (c, h, w, lh, lw, scale) in enumerate(cases):
            x = synth_coeffs(100 + k, c, h, w, lh, lw, scale=scale)
            n = x.size
            for G in (0, 2, 5, 16, 40):
                ctx.set_option("wide_groups", G)
                budgets = [UNLIMITED, n // 3, n // 11 + 13, int(rng.integers(1, n)), int(rng.integers(1, 2000))]
                for mb in budgets[: 5 if G in (0, 5) else 2]:
                    try:
                        _check_encode(oracle, x, lh, lw, mb)
                    except Exception as e:
                        raise AssertionError("case %d (c=%d %dx%d ll %dx%d) G=%d max_bits=%d: %r" % (k, c, h, w, lh, lw, G, mb, e))
        # a batch of four through the batched entry point: four groups side by side
        ctx.set_option("wide_groups", 8)
        xs = np.stack([synth_coeffs(200 + b, 3, 200, 280, 7, 9, scale=4000.0) for b in range(4)])
        import ctypes as C
        from spiht_amd.batch import DeviceArray
        L, vp = _lib.lib(), C.c_void_p
        mb, slot = 150000, (150000 // 8 + 8) // 4 * 4
        d_x, d_out = DeviceArray(ctx, xs.shape, np.int32), DeviceArray(ctx, (4, slot), np.uint8)
        d_nbits, d_maxn = DeviceArray(ctx, (4,), np.uint64), DeviceArray(ctx, (4,), np.uint8)
        d_x.upload(xs)
        _lib.check(L.spiht_encode_batch_i32(ctx.handle, vp(d_x.ptr), 4, 3, 200, 280, 7, 9, mb, vp(d_out.ptr), slot, vp(d_nbits.ptr),
                                            vp(d_maxn.ptr)))
        out, nbits, maxn = d_out.download(), d_nbits.download(), d_maxn.download()
        for b in range(4):
            d_ref, n_ref, nb_ref = oracle.encode_nbits(xs[b], 7, 9, mb)
            assert int(nbits[b]) == nb_ref and int(maxn[b]) == n_ref and out[b, :len(d_ref)].tobytes() == d_ref, b
        for a_ in (d_x, d_out, d_nbits, d_maxn):
            a_.free()
    finally:
        ctx.set_option("wide_groups", 0)
        ctx.set_option("wide_encode", 1)


def test_wide_encoder_falls_back_when_every_group_gives_up(oracle):
    """A group of workgroups that finds itself "given up" (WideCtl::bad & 2: a wait ran out because the group was not all
    resident) leaves at once, and the k_encode<redo> launch queued behind codes those images with one workgroup each, from
    scratch: option "wide_encode" = 3 marks every group so before the launch.  Streams, bit counts and start planes equal the
    oracle's; the statistics say every image went that way."""
    import ctypes as C
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    ctx = _lib.Context(0)
    try:
        ctx.set_option("wide_encode", 3)
        L, vp = _lib.lib(), C.c_void_p
        for k, (c, h, w, lh, lw, scale, mb) in enumerate([(3, 300, 420, 5, 7, 3000.0, 123457), (1, 512, 512, 4, 4, 20000.0, UNLIMITED),
                                                          (3, 131, 203, 3, 5, 800.0, 999)]):
            x = synth_coeffs(500 + k, c, h, w, lh, lw, scale=scale)
            d_ref, n_ref, nb_ref = oracle.encode_nbits(x, lh, lw, mb)
            out = np.full(len(d_ref) + 64, 0xEE, np.uint8)
            nbits, max_n = C.c_uint64(), C.c_uint8()
            _lib.check(L.spiht_encode_i32(ctx.handle, vp(x.ctypes.data), c, h, w, h * w, w, 1, lh, lw, mb, vp(out.ctypes.data), out.size,
                                          C.byref(nbits), C.byref(max_n)))
            assert (int(nbits.value), int(max_n.value)) == (nb_ref, n_ref) and out[:len(d_ref)].tobytes() == d_ref, k
            assert ctx.wide_stats() == (1, 1)
        # a batch: some slots hold what the several-CUs kernel would have left (here: nothing), all come out right
        xs = np.stack([synth_coeffs(520 + b, 3, 200, 280, 7, 9, scale=4000.0) for b in range(3)])
        mb, slot = 150000, (150000 // 8 + 8) // 4 * 4
        d_x, d_out = DeviceArray(ctx, xs.shape, np.int32), DeviceArray(ctx, (3, slot), np.uint8)
        d_nbits, d_maxn = DeviceArray(ctx, (3,), np.uint64), DeviceArray(ctx, (3,), np.uint8)
        d_x.upload(xs)
        _lib.check(L.spiht_encode_batch_i32(ctx.handle, vp(d_x.ptr), 3, 3, 200, 280, 7, 9, mb, vp(d_out.ptr), slot, vp(d_nbits.ptr),
                                            vp(d_maxn.ptr)))
        ctx.synchronize()
        assert ctx.wide_stats() == (3, 3)
        out, nbits, maxn = d_out.download(), d_nbits.download(), d_maxn.download()
        for b in range(3):
            d_ref, n_ref, nb_ref = oracle.encode_nbits(xs[b], 7, 9, mb)
            assert int(nbits[b]) == nb_ref and int(maxn[b]) == n_ref and out[b, :len(d_ref)].tobytes() == d_ref, b
    finally:
        ctx.close()


def test_wide_encoder_beside_a_kernel_that_holds_the_cus(oracle, tmp_path):
    """The reference's encode never fails on valid input (src/lib.rs:24-32).  The several-CUs-per-image encoder needs its
    workgroups resident together, and a kernel of somebody else's can hold the CUs: tests/native/filler.hip (built here with
    hipcc, launched on a stream the library knows nothing of) holds all CUs but 40 for 0.4 s -- one workgroup of 1024 threads
    and 96 KB of LDS per CU, beside which a 1024-thread encoder workgroup finds no registers -- then a 1080p-sized array is
    encoded with 64 workgroups asked for: 40 get a CU, 24 do not.  The group gives up after its bounded wait and the
    single-workgroup kernel behind it codes the image: the stream is the oracle's and the call returns long before the
    filler ends."""
    import ctypes as C
    import subprocess
    import time
    import spiht_amd
    from spiht_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libfiller.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC",
                           os.path.join(root, "tests", "native", "filler.hip"), "-o", so])
    F = C.CDLL(so)
    F.filler_launch.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint64]
    c, h, w, lh, lw = 3, 1111, 1949, 13, 19
    x = synth_coeffs(42, c, h, w, lh, lw)
    mb = 1036800
    d_ref, n_ref, _ = oracle.encode_nbits(x, lh, lw, mb)
    ctx = _lib.default_context()
    ctx.set_option("wide_groups", 64)
    try:
        d, n = spiht_amd.encode(x, lh, lw, mb)  # (sizes the context's buffers; alone: every group resident)
        assert (d, n) == (d_ref, n_ref) and ctx.wide_stats() == (1, 0)
        ncu = F.filler_num_cu()
        assert ncu > 64
        hold_s = 0.4
        rc = F.filler_launch(ncu - 40, 1024, 96 * 1024, int(hold_s * 2.4e9))
        if rc == -2:
            pytest.skip("this device does not give a workgroup 96 KB of LDS: no way to hold a CU with one workgroup")
        assert rc == 0
        time.sleep(0.02)  # (the filler's workgroups are on the CUs)
        t0 = time.perf_counter()
        d, n = spiht_amd.encode(x, lh, lw, mb)
        dt = time.perf_counter() - t0
        groups, gave_up = ctx.wide_stats()
        assert F.filler_wait() == 0
        assert (d, n) == (d_ref, n_ref), "stream differs (groups %d, gave up %d, %.3f s)" % (groups, gave_up, dt)
        assert dt < 0.6 * hold_s, "the encode call waited for the filler: %.3f s (gave up: %d)" % (dt, gave_up)
        print("encode beside the filler: %.1f ms, groups that gave up: %d of %d" % (dt * 1e3, gave_up, groups))
    finally:
        F.filler_wait()
        ctx.set_option("wide_groups", 0)


def test_wide_encoder_from_two_contexts_at_once(oracle):
    """Two threads, a context each, 160 workgroups per image: the two grids together exceed the CUs, and the workgroups of
    a grid wait for one another -- half-resident twins would wait for ever.  The library chains such launches one at a
    time per device (api.cpp: encode_lists_device); every stream must come out right, no spin limit hit."""
    import ctypes as C
    import threading
    from spiht_amd import _lib
    x = [synth_coeffs(300 + t, 3, 400, 520, 7, 9, scale=5000.0) for t in range(2)]
    ref = [oracle.encode_nbits(x[t], 7, 9, 400000)[:2] for t in range(2)]
    errs = []

    def work(t):
        try:
            ctx = _lib.Context(0)
            ctx.set_option("wide_groups", 160)
            L = _lib.lib()
            c, h, w = x[t].shape
            out = np.empty(400000 // 8 + 8, np.uint8)
            nbits, max_n = C.c_uint64(), C.c_uint8()
            for _ in range(12):
                _lib.check(L.spiht_encode_i32(ctx.handle, C.c_void_p(x[t].ctypes.data), c, h, w, h * w, w, 1, 7, 9, 400000,
                                              C.c_void_p(out.ctypes.data), out.size, C.byref(nbits), C.byref(max_n)))
                d, n = out[: (nbits.value + 7) // 8].tobytes(), int(max_n.value)
                if (d, n) != ref[t]:
                    errs.append("thread %d: stream differs" % t)
                    break
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append("thread %d: %r" % (t, e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errs and not any(t_.is_alive() for t_ in th), errs
