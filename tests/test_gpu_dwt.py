"""GPU parity tests of the DWT / inverse DWT kernels (through the C ABI) against the float64 oracle and the
PyWavelets-captured golden arrays."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gpu_dwt(img, wavelet, mode, level, q, mults):
    from spiht_amd import _lib
    ctx, L = _lib.default_context(), _lib.lib()
    img = np.ascontiguousarray(img, np.float64)
    B, c, H, W = img.shape
    wid, mid = L.spiht_wavelet_id(wavelet.encode()), L.spiht_mode_id(mode.encode())
    v = [C.c_int64() for _ in range(6)]
    lv = C.c_int()
    _lib.check(L.spiht_geometry_mode(H, W, wid, mid, -1 if level is None else level, C.byref(lv), *[C.byref(t) for t in v]))
    eh, ew = v[2].value, v[3].value
    out = np.empty((B, c, eh, ew), np.int32)
    d_in, d_out = ctx.alloc(img.nbytes), ctx.alloc(out.nbytes)
    m = None if mults is None else np.ascontiguousarray(mults, np.float64)
    try:
        ctx.upload(d_in, img)
        ctx.memset(d_out, 0xFF, out.nbytes)  # prove padding cells are written
        _lib.check(L.spiht_dwt_quant_batch_f64(ctx.handle, C.c_void_p(d_in), B, c, H, W, wid, mid,
                                               -1 if level is None else level, float(q),
                                               None if m is None else C.c_void_p(m.ctypes.data), C.c_void_p(d_out)))
        ctx.download(out, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    return out


def _gpu_idwt(rec, H, W, wavelet, mode, level, q, mults):
    from spiht_amd import _lib
    ctx, L = _lib.default_context(), _lib.lib()
    rec = np.ascontiguousarray(rec, np.int32)
    B, c = rec.shape[:2]
    wid, mid = L.spiht_wavelet_id(wavelet.encode()), L.spiht_mode_id(mode.encode())
    v = [C.c_int64() for _ in range(6)]
    lv = C.c_int()
    _lib.check(L.spiht_geometry_mode(H, W, wid, mid, -1 if level is None else level, C.byref(lv), *[C.byref(t) for t in v]))
    out = np.empty((B, c, v[4].value, v[5].value), np.float64)
    d_in, d_out = ctx.alloc(rec.nbytes), ctx.alloc(out.nbytes)
    m = None if mults is None else np.ascontiguousarray(mults, np.float64)
    try:
        ctx.upload(d_in, rec)
        _lib.check(L.spiht_dequant_idwt_batch_f64(ctx.handle, C.c_void_p(d_in), B, c, H, W, wid, mid,
                                                  -1 if level is None else level, float(q),
                                                  None if m is None else C.c_void_p(m.ctypes.data), C.c_void_p(d_out)))
        ctx.download(out, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    return out


def _cases():
    w = np.load(os.path.join(GOLD, "wrapper_pywt.npz"))
    for k in range(int(w["ncases"])):
        p = "case%02d_" % k
        c, H, W, lv = [int(v) for v in w[p + "meta"]]
        m = w[p + "mults"]
        yield dict(img=w[p + "img"], coeffs=w[p + "coeffs"], rec=w[p + "rec"], rec_img=w[p + "rec_img"], c=c, H=H, W=W,
                   level=None if lv < 0 else lv, wavelet=str(w[p + "wavelet"]), mode=str(w[p + "mode"]),
                   q=float(w[p + "q"]), mults=None if m.size == 0 else m)


def test_forward_matches_pywt_goldens_and_oracle(oracle):
    for cs in _cases():
        got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"], cs["mults"])[0]
        assert np.array_equal(got, cs["coeffs"]), (cs["wavelet"], cs["mode"], cs["level"])
        arr, _ = oracle.wavedec2_array(cs["img"], cs["wavelet"], cs["mode"], cs["level"])
        assert np.array_equal(got, oracle.quantize(arr, cs["q"], cs["mults"]))


def test_forward_matches_pywt_on_blocky_images(oracle):
    """The quantised coefficient arrays PyWavelets + the wrapper's arithmetic produce for piecewise-constant 8-bit
    pictures (tests/golden/blocky_pywt.npz): the bottom / right overhang is summed in pywt's order (k_dwt_edge)."""
    from test_oracle import blocky_cases
    n = 0
    for cs in blocky_cases():
        got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        bad = np.argwhere(got != cs["quant"])
        assert len(bad) == 0, (cs["wavelet"], cs["mode"], cs["img"].shape, len(bad), bad[:4])
        n += 1
    assert n == 40


def test_inverse_matches_pywt_goldens_and_oracle(oracle):
    for cs in _cases():
        got = _gpu_idwt(cs["rec"][None], cs["H"], cs["W"], cs["wavelet"], cs["mode"], cs["level"], cs["q"], cs["mults"])[0]
        assert got.shape == cs["rec_img"].shape
        # tolerance vs pywt: summation order differs
        assert np.array_equal(got, cs["rec_img"])  # bit-identical to pywt.waverec2 (same order of additions)
        ref = oracle.waverec2_array(oracle.dequantize(cs["rec"], cs["q"], cs["mults"]), cs["H"], cs["W"], cs["wavelet"],
                                    cs["level"])
        # same summation order as the oracle: bit for bit
        assert np.array_equal(got, ref), (cs["wavelet"], cs["level"], float(np.abs(got - ref).max()))


@pytest.mark.parametrize("cfg", [
    (2, 3, 120, 200, "bior2.2", "reflect", None, 50.0, None),
    (1, 1, 512, 512, "bior2.2", "reflect", 5, 50.0, None),          # BASELINE config 1 geometry
    (1, 3, 270, 480, "bior2.2", "reflect", 5, 50.0, None),
    (3, 1, 131, 77, "bior4.4", "symmetric", 3, 255.0, None),
    (1, 3, 256, 256, "bior6.8", "reflect", 5, 1.0, [50.0, 15.0, 15.0]),  # level > pywt max (Q13)
    (1, 1, 65, 33, "haar", "periodic", 3, 50.0, None),
    (1, 2, 40, 40, "bior2.2", "zero", 0, 50.0, [2.0, 0.5]),          # level 0: quantise only
    (1, 1, 9, 300, "bior2.2", "constant", 1, 50.0, None),
    # many planes (batches): odd sizes with the trim rule, long filter, per-channel scales
    (50, 3, 72, 100, "bior2.2", "reflect", 4, 50.0, None),
    (43, 3, 1080 // 2, 1920 // 2, "bior2.2", "reflect", 6, 50.0, None),
    (129, 1, 61, 47, "bior4.4", "symmetric", 3, 255.0, None),
    (44, 3, 96, 96, "bior6.8", "reflect", 3, 20.0, [3.0, 1.0, 0.5]),
])
def test_forward_inverse_vs_oracle(oracle, cfg):
    B, c, H, W, wavelet, mode, level, q, mults = cfg
    imgs = np.stack([synth_image(1000 + b, c, H, W) for b in range(B)])
    got = _gpu_dwt(imgs, wavelet, mode, level, q, mults)
    check = range(B) if B <= 8 else sorted({0, 1, B // 2, B - 2, B - 1})  # many planes: the oracle checks a few images
    for b in check:
        arr, g = oracle.wavedec2_array(imgs[b], wavelet, mode, level)
        ref = oracle.quantize(arr, q, mults)
        assert np.array_equal(got[b], ref), (cfg, int((got[b] != ref).sum()))
    back = _gpu_idwt(got, H, W, wavelet, mode, level, q, mults)
    for b in check:
        ref = oracle.waverec2_array(oracle.dequantize(got[b], q, mults), H, W, wavelet, level)
        assert back[b].shape == ref.shape
        assert np.array_equal(back[b], ref), (cfg, float(np.abs(back[b] - ref).max()))
        # quantisation error only: the reconstruction is close to the input
        assert np.abs(back[b][:, :H, :W] - imgs[b]).max() < 0.2


def test_quantised_transform_of_bench_images_matches_pywt_digests():
    """the same digests as tests/test_oracle.py, from the GPU transform (through the C ABI): what bench.py codes is what
    the reference's PyWavelets front end would hand to its Rust core"""
    import ctypes as C
    import hashlib
    import os
    from conftest import synth_image
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_pywt_digests.npz"))
    ctx, L = _lib.default_context(), _lib.lib()
    for row, wv, q, dig in zip(z["cases"], z["wavelets"], z["q"], z["sha1"]):
        seed, c, h, w, lv = [int(v) for v in row]
        wid = L.spiht_wavelet_id(str(wv).encode())
        lvl, v = C.c_int(), [C.c_int64() for _ in range(6)]
        _lib.check(L.spiht_geometry(h, w, wid, lv, C.byref(lvl), *[C.byref(t) for t in v]))
        enc_h, enc_w = v[2].value, v[3].value
        d_img = DeviceArray(ctx, (c, h, w), np.float64)
        d_co = DeviceArray(ctx, (c, enc_h, enc_w), np.int32)
        d_img.upload(synth_image(seed, c, h, w))
        _lib.check(L.spiht_dwt_quant_batch_f64(ctx.handle, C.c_void_p(d_img.ptr), 1, c, h, w, wid, 0, lv, float(q), None,
                                               C.c_void_p(d_co.ptr)))
        ctx.synchronize()
        qa = d_co.download()
        assert hashlib.sha1(qa.tobytes()).hexdigest() + ":%dx%dx%d" % qa.shape == str(dig), (seed, h, w, str(wv))
        d_img.free(); d_co.free()


def _w32_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wrapper32_pywt.npz"))
    for i in range(int(z["ncases"])):
        p = "c%d_" % i
        seed, c, h, w, lv = [int(v) for v in z[p + "meta"]]
        mults = z[p + "mults"]
        yield dict(seed=seed, c=c, h=h, w=w, level=None if lv < 0 else lv, wavelet=str(z[p + "wavelet"]), q=float(z[p + "q"]),
                   mults=None if mults.size == 0 else mults.tolist(), f16=bool(z[p + "f16"]), sha1=str(z[p + "sha1"]),
                   mode=str(z[p + "mode"]),
                   quant=z[p + "quant"] if p + "quant" in z.files else None)


def test_float32_pixels_follow_pywt_single_precision(oracle):
    """float32 / float16 pixels: PyWavelets transforms and the wrapper quantises in single precision.  GPU == pywt 1.1.1
    goldens (tests/golden/wrapper32_pywt.npz) on the int32 arrays handed to the SPIHT core, == oracle, and the streams
    of encode_image follow"""
    import ctypes as C
    import hashlib
    import spiht_amd
    from conftest import synth_image
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    ctx, L = _lib.default_context(), _lib.lib()
    for cs in _w32_cases():
        c, h, w = cs["c"], cs["h"], cs["w"]
        img = synth_image(cs["seed"], c, h, w).astype(np.float16 if cs["f16"] else np.float32)
        wid = L.spiht_wavelet_id(cs["wavelet"].encode())
        lv = -1 if cs["level"] is None else cs["level"]
        lvl, v = C.c_int(), [C.c_int64() for _ in range(6)]
        _lib.check(L.spiht_geometry(h, w, wid, lv, C.byref(lvl), *[C.byref(t) for t in v]))
        enc_h, enc_w = v[2].value, v[3].value
        d_img = DeviceArray(ctx, (c, h, w), np.float32)
        d_co = DeviceArray(ctx, (c, enc_h, enc_w), np.int32)
        d_img.upload(img.astype(np.float32))
        m = None if cs["mults"] is None else np.ascontiguousarray(cs["mults"], dtype=np.float64)
        _lib.check(L.spiht_dwt_quant_batch_f32(ctx.handle, C.c_void_p(d_img.ptr), 1, c, h, w, wid, _lib.MODES[cs["mode"]], lv, cs["q"],
                                               None if m is None else C.c_void_p(m.ctypes.data), C.c_void_p(d_co.ptr)))
        ctx.synchronize()
        qa = d_co.download()
        d_img.free(); d_co.free()
        if cs["quant"] is not None:
            assert np.array_equal(qa, cs["quant"]), (cs["seed"], "vs pywt array")
        assert hashlib.sha1(qa.tobytes()).hexdigest() + ":%dx%dx%d" % qa.shape == cs["sha1"], (cs["seed"], h, w)
        # the wrapper picks the single-precision path from the dtype, and the stream equals the oracle's
        s = spiht_amd.SpihtSettings(wavelet=cs["wavelet"], quantization_scale=cs["q"], per_channel_quant_scales=cs["mults"],
                                    mode=cs["mode"])
        mb = 20000
        enc = spiht_amd.encode_image(img, s, level=cs["level"], max_bits=mb)
        ref_bytes, ref_n, _ = oracle.encode_image(img, cs["wavelet"], cs["mode"], cs["level"], cs["q"], cs["mults"], mb)
        assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n
        enc64 = spiht_amd.encode_image(img.astype(np.float64), s, level=cs["level"], max_bits=mb)
        assert enc64.encoded_bytes == oracle.encode_image(img.astype(np.float64), cs["wavelet"], cs["mode"], cs["level"], cs["q"],
                                                          cs["mults"], mb)[0]


# (c, H, W, wavelet, mode, level): odd and even band offsets, bands narrower / shorter than a tile, every filter length,
# the five extension modes, one level (nothing written ahead), cfg2's geometry
D1_CASES = [(3, 96, 160, "bior2.2", "reflect", 3), (1, 97, 163, "bior2.2", "symmetric", 4), (2, 130, 75, "bior4.4", "periodic", 2),
            (1, 61, 47, "haar", "zero", 3), (1, 64, 64, "haar", "reflect", 5), (2, 200, 264, "bior6.8", "constant", 3),
            (1, 70, 90, "bior2.2", "reflect", 1), (1, 333, 517, "bior4.4", "reflect", None),
            (3, 1080, 1920, "bior2.2", "reflect", 7)]


@pytest.mark.parametrize("case", D1_CASES)
def test_pyramid_behind_the_forward_transform(oracle, case):
    """spiht_dwt_pyramid_batch_f64 (the front half of the encoder as the pipelined schedule queues it): the significance
    pyramid it leaves must be the recursion's (encoder_decoder.rs:78-121, evaluated per node by the oracle) at every node
    with offspring -- and the same as the pyramid pass on its own gives over the same coefficients."""
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    c, H, W, wavelet, mode, level = case
    L, ctx, vp = _lib.lib(), _lib.Context(0), C.c_void_p
    wid, mid = L.spiht_wavelet_id(wavelet.encode()), L.spiht_mode_id(mode.encode())
    lv = -1 if level is None else level
    v = [C.c_int64() for _ in range(6)]
    lu = C.c_int()
    _lib.check(L.spiht_geometry(H, W, wid, lv, C.byref(lu), *[C.byref(t) for t in v]))
    lh, lw, eh, ew = (t.value for t in v[:4])
    B = 2
    n = c * eh * ew
    imgs = np.stack([synth_image(500 + b, c, H, W) for b in range(B)])
    imgs[1, :, : H // 2] = 0.25  # a flat half: empty sets (code 0) next to non-empty ones
    d_img = DeviceArray(ctx, imgs.shape, np.float64)
    d_img.upload(imgs)
    co, dm, lm, ma = (DeviceArray(ctx, (B, n), np.int32), DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B, n), np.uint8),
                      DeviceArray(ctx, (B,), np.uint32))
    ctx.memset(dm.ptr, 0xEE, dm.nbytes)
    ctx.memset(lm.ptr, 0xEE, lm.nbytes)
    mults = np.array([1.0, 0.5, 2.0][:c])
    _lib.check(L.spiht_dwt_pyramid_batch_f64(ctx.handle, vp(d_img.ptr), B, c, H, W, wid, mid, lv, 50.0, vp(mults.ctypes.data),
                                             vp(co.ptr), vp(dm.ptr), vp(lm.ptr), vp(ma.ptr)))
    ctx.synchronize()
    x = co.download().reshape(B, c, eh, ew)
    d_got, l_got = dm.download().reshape(B, c, eh, ew), lm.download().reshape(B, c, eh, ew)
    # the pyramid pass on its own over the same coefficients
    ctx.memset(dm.ptr, 0xEE, dm.nbytes)
    ctx.memset(lm.ptr, 0xEE, lm.nbytes)
    _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(co.ptr), B, c, eh, ew, lh, lw, vp(dm.ptr), vp(lm.ptr), vp(None)))
    ctx.synchronize()
    d_alone, l_alone = dm.download().reshape(B, c, eh, ew), lm.download().reshape(B, c, eh, ew)
    I, J = np.arange(eh)[:, None], np.arange(ew)[None, :]
    b_entry = ((4 * I + 3 < eh) & (4 * J + 3 < ew))[None]
    for b in range(B):
        if c * eh * ew <= 400000:
            d_ref, l_ref, has = oracle.set_codes(x[b], lh, lw)
        else:  # (the recursion takes minutes at 1080p: there the pass on its own, itself checked against it, is the reference)
            has = ((2 * I + 1 < eh) & (2 * J + 1 < ew) & ~((I < lh) & (J < lw) & (I % 2 == 0) & (J % 2 == 0)))[None].repeat(c, 0)
            d_ref, l_ref = d_alone[b], l_alone[b]
        for name, got, alone, ref, where in (("D", d_got[b], d_alone[b], d_ref, has), ("L", l_got[b], l_alone[b], l_ref, has & b_entry)):
            bad = np.argwhere((got != ref) & where)
            assert len(bad) == 0, "%s code differs at %d nodes, first %s: got %d want %d" % (
                name, len(bad), bad[0], got[tuple(bad[0])], ref[tuple(bad[0])])
            assert np.array_equal(got[where], alone[where])
    for a in (d_img, co, dm, lm, ma):
        a.free()


def test_pads_persist_tells_layouts_of_one_array_apart(oracle):
    """Option "pads_persist": the zero padding of coeffs_to_array is written once per array and layout.  Periodization packs
    the bands by another length rule than the other modes -- other strips in the same array at the same picture size, filter
    and level count -- so the record of an array must hold the mode's layout too: alternating reflect and periodization into
    ONE buffer, every result must equal the oracle's array (a stale record would leave band values of the other layout in
    padding cells, and the list coder would code them)."""
    from spiht_amd import _lib
    ctx, L = _lib.Context(0), _lib.lib()
    c, H, W, wavelet, level, q = 2, 75, 118, "bior2.2", 3, 50.0
    img = np.ascontiguousarray(synth_image(77, c, H, W)[None], np.float64)
    wid = L.spiht_wavelet_id(wavelet.encode())
    ref, shape = {}, {}
    for mode in ("reflect", "periodization"):
        arr, g = oracle.wavedec2_array(img[0], wavelet, mode, level)
        ref[mode], shape[mode] = oracle.quantize(arr, q, None), (c, g["enc_h"], g["enc_w"])
    assert shape["reflect"] != shape["periodization"]
    nmax = max(int(np.prod(v)) for v in shape.values())
    d_in, d_out = ctx.alloc(img.nbytes), ctx.alloc(nmax * 4)
    try:
        ctx.set_option("pads_persist", 1)
        ctx.upload(d_in, img)
        ctx.memset(d_out, 0xFF, nmax * 4)
        for it, mode in enumerate(["reflect", "periodization", "reflect", "reflect", "periodization", "periodization"]):
            mid = L.spiht_mode_id(mode.encode())
            _lib.check(L.spiht_dwt_quant_batch_f64(ctx.handle, C.c_void_p(d_in), 1, c, H, W, wid, mid, level, q, None, C.c_void_p(d_out)))
            got = np.empty(shape[mode], np.int32)
            ctx.download(got, d_out)
            assert np.array_equal(got, ref[mode]), (it, mode, int((got != ref[mode]).sum()))
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
        ctx.close()


def test_maxabs_small_batch_after_large_batch(oracle):
    """max|coefficient| of an image is raised by one atomic per workgroup, left out when a cached look at the word shows
    it holds as much already (dwt.hip: block_raise_max, MAXLOOK 1).  That look must never see a value from BEFORE the
    word was last zeroed: a large-magnitude batch and then a small-magnitude one go through the same context and the
    same buffer, several times over; start plane and stream of every image must be the oracle's."""
    import spiht_amd
    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec
    c, H, W, B = 1, 192, 256, 6
    ctx = _lib.default_context()
    big = np.stack([synth_image(40 + b, c, H, W) * 4000.0 for b in range(B)])
    small = np.stack([synth_image(60 + b, c, H, W) * 0.02 for b in range(B)])
    s = spiht_amd.SpihtSettings()
    codec = BatchCodec(c, H, W, s, 4, 6000, ctx=ctx)
    want = {}
    for name, imgs in (("big", big), ("small", small)):
        want[name] = [oracle.encode_image(imgs[b], "bior2.2", "reflect", 4, 50.0, None, 6000)[:2] for b in range(B)]
    assert min(w[1] for w in want["big"]) > max(w[1] for w in want["small"]) + 8  # the start planes are far apart
    for _ in range(4):
        for name, imgs in (("big", big), ("small", small)):
            res = codec.encode(imgs)
            for b in range(B):
                assert res[b].max_n == want[name][b][1], (name, b)
                assert res[b].encoded_bytes == want[name][b][0], (name, b)


# (c, H, W, wavelet, level, bits per pixel): halos of 0, 2, 4 and 8 band positions, odd band sizes, bands smaller than a tile
L1F_CASES = [(3, 270, 480, "bior2.2", 5, 0.5), (1, 333, 517, "bior4.4", 4, 1.0), (2, 200, 264, "bior6.8", 3, 2.0),
             (1, 129, 257, "haar", 2, 0.25), (3, 1080, 1920, "bior2.2", 7, 0.5), (1, 70, 90, "bior2.2", 1, 1.0)]


@pytest.mark.parametrize("case", L1F_CASES)
def test_level1_tile_occupancy_words(oracle, case):
    """The list decoder leaves one word per (plane, tile) of the inverse transform's level 1 (common.h: L1Flags) and the
    inverse level-1 kernels do not read the detail bands of a tile whose word is zero.  Checked here: (i) wherever a word
    is zero, every detail-band cell the tile stages -- its own 12 x 64 band positions and the halo of F/2 - 1 behind them --
    is zero in the decoded array; (ii) the picture is the same, bit for bit, with and without the words, and equals the
    oracle's (waverec2 of the dense array); (iii) at the lower rates most words are zero (the point of it)."""
    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray
    from spiht_amd.spiht_wrapper import SpihtSettings
    c, H, W, wavelet, level, bpp = case
    F = {"haar": 2, "bior2.2": 6, "bior4.4": 10, "bior6.8": 18}[wavelet]
    B = 2
    L, ctx, vp = _lib.lib(), _lib.default_context(), C.c_void_p
    s = SpihtSettings(wavelet=wavelet)
    mb = int(H * W * bpp)
    cd = BatchCodec(c, H, W, s, level, mb, ctx=ctx)
    g = cd.geom
    imgs = np.stack([synth_image(700 + b, c, H, W) for b in range(B)])
    res = cd.encode(imgs)
    stride = max(4, (max(len(r.encoded_bytes) for r in res) + 3) & ~3)
    data = np.zeros((B, stride), np.uint8)
    for b, r in enumerate(res):
        data[b, :len(r.encoded_bytes)] = np.frombuffer(r.encoded_bytes, np.uint8)
    d_data, d_nb, d_mn = DeviceArray(ctx, data.shape, np.uint8), DeviceArray(ctx, (B,), np.uint64), DeviceArray(ctx, (B,), np.uint8)
    d_data.upload(data)
    d_nb.upload(np.array([len(r.encoded_bytes) for r in res], np.uint64))
    d_mn.upload(np.array([r.max_n for r in res], np.uint8))
    n = c * g["enc_h"] * g["enc_w"]
    d_rec = DeviceArray(ctx, (B, n), np.int32)
    d_rec.zero()
    nw = C.c_uint64()
    _lib.check(L.spiht_l1_flags_words(c, H, W, cd.wid, cd.mid, cd._lv, C.byref(nw)))
    assert (nw.value > 0) == (g["level"] >= 2)
    d_fl = DeviceArray(ctx, (B, max(nw.value, 1)), np.uint32)
    ctx.memset(d_fl.ptr, 0xEE, d_fl.nbytes)  # (the call zero-fills them itself)
    _lib.check(L.spiht_decode_lists_flags_batch_i32(ctx.handle, vp(d_data.ptr), stride, vp(d_nb.ptr), vp(d_mn.ptr), B, c, H, W,
                                                    cd.wid, cd.mid, cd._lv, vp(d_rec.ptr), vp(d_fl.ptr if nw.value else None)))
    outs = []
    for fl in (d_fl.ptr if nw.value else None, None):
        d_img = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)
        _lib.check(L.spiht_dequant_idwt_flags_batch_f64(ctx.handle, vp(d_rec.ptr), vp(fl), B, c, H, W, cd.wid, cd.mid, cd._lv,
                                                        float(s.quantization_scale), None, vp(d_img.ptr)))
        ctx.synchronize()
        outs.append(d_img.download())
        d_img.free()
    assert np.array_equal(outs[0], outs[1])
    rec = d_rec.download().reshape(B, c, g["enc_h"], g["enc_w"])
    for b in range(B):
        ref = oracle.decode_image(res[b].encoded_bytes, res[b].max_n, c, H, W, wavelet, level, float(s.quantization_scale), None)
        assert np.array_equal(outs[0][b], ref)
    if nw.value:
        hs, ws = (H + F - 1) // 2, (W + F - 1) // 2          # level-1 band size
        oh, ow = g["enc_h"] - hs, g["enc_w"] - ws              # its offsets in the packed array
        gy, gx = (2 * hs - F + 2 + 23) // 24, (2 * ws - F + 2 + 127) // 128
        assert nw.value == c * gy * gx
        fl = d_fl.download().reshape(B, c, gy, gx)
        assert set(np.unique(fl)) <= {0, 1}
        bands = np.stack([rec[:, :, :hs, ow:ow + ws], rec[:, :, oh:oh + hs, :ws], rec[:, :, oh:oh + hs, ow:ow + ws]], axis=2) != 0
        occ = bands.any(axis=2)  # [B, c, hs, ws]: some detail band holds a value at this band position
        hf1 = F // 2 - 1
        for ty in range(gy):
            for tx in range(gx):
                staged = occ[:, :, 12 * ty:12 * ty + 12 + hf1, 64 * tx:64 * tx + 64 + hf1].any(axis=(2, 3))
                assert not (staged & (fl[:, :, ty, tx] == 0)).any(), (ty, tx)
        if bpp <= 0.5:
            assert fl.mean() < 0.5, float(fl.mean())
        # ... and for streams no encoder made: random bytes put values anywhere in the tree, padding cells included
        rng = np.random.default_rng(17)
        for trial in range(3):
            junk = rng.integers(0, 256, (B, stride), dtype=np.uint8)
            d_data.upload(junk)
            d_nb.upload(np.array([stride - 3 * trial, stride // 2], np.uint64))
            d_mn.upload(np.array([9 + trial, 6], np.uint8))
            d_rec.zero()
            _lib.check(L.spiht_decode_lists_flags_batch_i32(ctx.handle, vp(d_data.ptr), stride, vp(d_nb.ptr), vp(d_mn.ptr), B, c, H, W,
                                                            cd.wid, cd.mid, cd._lv, vp(d_rec.ptr), vp(d_fl.ptr)))
            two = []
            for f in (d_fl.ptr, None):
                d_img = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)
                _lib.check(L.spiht_dequant_idwt_flags_batch_f64(ctx.handle, vp(d_rec.ptr), vp(f), B, c, H, W, cd.wid, cd.mid, cd._lv,
                                                                float(s.quantization_scale), None, vp(d_img.ptr)))
                ctx.synchronize()
                two.append(d_img.download())
                d_img.free()
            assert np.array_equal(two[0], two[1]), trial
    for a in (d_data, d_nb, d_mn, d_rec, d_fl):
        a.free()


def test_every_wavelet_up_to_20_taps(oracle):
    """SpihtSettings.wavelet goes to PyWavelets as it is in the reference (spiht_wrapper.py:163, :276): each of the 53
    discrete wavelets with at most 20 taps on the GPU against PyWavelets 1.1.1 (tests/golden/wavelets_pywt.npz) -- the
    int32 array handed to the coder, and the picture back from a thinned-out copy of it, bit for bit -- and against the
    oracle on a multi-tile picture per filter length."""
    from test_oracle import transform_cases
    seen = {}
    for cs in transform_cases("wavelets_pywt.npz"):
        got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        bad = np.argwhere(got != cs["quant"])
        assert len(bad) == 0, (cs["wavelet"], cs["mode"], cs["img"].shape, len(bad), bad[:4])
        back = _gpu_idwt(cs["rec"][None], cs["H"], cs["W"], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        assert back.shape == cs["rec_img"].shape and np.array_equal(back, cs["rec_img"]), (cs["wavelet"], cs["level"])
        seen.setdefault(len(oracle.wavelet_filters(cs["wavelet"])[0]), cs["wavelet"])
    assert len(seen) == 10  # filter lengths 2, 4, ..., 20
    # larger than a tile in both directions, several levels, a batch: one wavelet per filter length
    for F, wv in sorted(seen.items()):
        imgs = np.stack([synth_image(70 + b, 2, 150, 301) for b in range(2)])
        got = _gpu_dwt(imgs, wv, "reflect", 3, 50.0, [1.0, 0.5])
        back = _gpu_idwt(got, 150, 301, wv, "reflect", 3, 50.0, [1.0, 0.5])
        for b in range(2):
            arr, _ = oracle.wavedec2_array(imgs[b], wv, "reflect", 3)
            assert np.array_equal(got[b], oracle.quantize(arr, 50.0, [1.0, 0.5])), (wv, F)
            ref = oracle.waverec2_array(oracle.dequantize(got[b], 50.0, [1.0, 0.5]), 150, 301, wv, 3)
            assert np.array_equal(back[b], ref), (wv, F)
    import spiht_amd
    with pytest.raises(ValueError):
        spiht_amd.encode_image(np.zeros((1, 64, 64)), spiht_amd.SpihtSettings(wavelet="db39"))   # (PyWavelets stops at db38)
    with pytest.raises(ValueError):
        spiht_amd.encode_image(np.zeros((1, 64, 64)), spiht_amd.SpihtSettings(wavelet="nonsense"))


def test_computed_extension_modes(oracle):
    """smooth / antisymmetric / antireflect / periodization (the reference passes SpihtSettings.mode to PyWavelets as it is):
    the two-pass forward level (dwt.hip: k_dwt_axis_ext) and, for periodization -- another length rule, ceil(n / 2) -- the
    two-pass inverse level (k_idwt_axis_per) against PyWavelets 1.1.1 (tests/golden/modes_pywt.npz) and, on a batch of
    pictures larger than a tile with channel scales and a colour model, against the oracle; the coder behind it included."""
    import spiht_amd
    from test_oracle import transform_cases
    n = 0
    for cs in transform_cases("modes_pywt.npz"):
        got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        bad = np.argwhere(got != cs["quant"])
        assert len(bad) == 0, (cs["wavelet"], cs["mode"], cs["img"].shape, len(bad), bad[:4])
        back = _gpu_idwt(cs["rec"][None], cs["H"], cs["W"], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        assert np.array_equal(back, cs["rec_img"])
        n += 1
    assert n == 40
    for mode in ("smooth", "antisymmetric", "antireflect", "periodization"):
        imgs = np.stack([synth_image(90 + b, 3, 131, 203) for b in range(3)])
        got = _gpu_dwt(imgs, "bior4.4", mode, 3, 50.0, [1.0, 0.5, 2.0])
        for b in range(3):
            arr, _ = oracle.wavedec2_array(imgs[b], "bior4.4", mode, 3)
            assert np.array_equal(got[b], oracle.quantize(arr, 50.0, [1.0, 0.5, 2.0])), mode
        s = spiht_amd.SpihtSettings(mode=mode)
        enc = spiht_amd.encode_image(imgs[0], s, 3, 20000)
        ref_bytes, ref_n, _ = oracle.encode_image(imgs[0], "bior2.2", mode, 3, 50.0, None, 20000)
        assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n, mode
        dec = spiht_amd.decode_image(enc, s)
        assert np.array_equal(dec, oracle.decode_image(ref_bytes, ref_n, 3, 131, 203, "bior2.2", 3, 50.0, None, mode=mode))
        # with a colour model: the change runs as a pass of its own in front of the two-pass level
        sc = spiht_amd.SpihtSettings(mode=mode, quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0])
        e2 = spiht_amd.encode_image(imgs[1], sc, 3, 20000)
        from spiht_amd import _lib, color_models
        from spiht_amd.batch import DeviceArray
        ctx = _lib.default_context()
        d = DeviceArray(ctx, imgs[1].shape, np.float64)
        d.upload(imgs[1])
        color_models.device_convert(ctx, d.ptr, 1, 131 * 203, "RGB", "IPT")   # the same kernel, as a call of its own
        ctx.synchronize()
        s_plain = spiht_amd.SpihtSettings(mode=mode, quantization_scale=1.0, per_channel_quant_scales=[50.0, 15.0, 15.0])
        e3 = spiht_amd.encode_image(d.download(), s_plain, 3, 20000)
        d.free()
        assert e2.encoded_bytes == e3.encoded_bytes and e2.max_n == e3.max_n, mode
        d2 = spiht_amd.decode_image(e2, sc)
        # (a loose bound: at this small budget the extrapolated borders of "smooth" cost most of the bits -- start plane 11
        # instead of 8 -- and the CPU oracle gives the same 0.19 mean error for it, 0.05 with the default mode)
        assert np.abs(d2[:, :131, :203] - imgs[1]).mean() < 0.25
    with pytest.raises(ValueError):
        spiht_amd.encode_image(imgs[0], spiht_amd.SpihtSettings(mode="nonsense"), 2)


def _gpu_dwt_f32(img, wavelet, mode, level, q):
    from spiht_amd import _lib
    ctx, L = _lib.default_context(), _lib.lib()
    img = np.ascontiguousarray(img, np.float32)
    B, c, H, W = img.shape
    wid, mid = L.spiht_wavelet_id(wavelet.encode()), L.spiht_mode_id(mode.encode())
    v = [C.c_int64() for _ in range(6)]
    lv = C.c_int()
    _lib.check(L.spiht_geometry_mode(H, W, wid, mid, level, C.byref(lv), *[C.byref(t) for t in v]))
    out = np.empty((B, c, v[2].value, v[3].value), np.int32)
    d_in, d_out = ctx.alloc(img.nbytes), ctx.alloc(out.nbytes)
    try:
        ctx.upload(d_in, img)
        _lib.check(L.spiht_dwt_quant_batch_f32(ctx.handle, C.c_void_p(d_in), B, c, H, W, wid, mid, level, float(q), None, C.c_void_p(d_out)))
        ctx.download(out, d_out)
    finally:
        ctx.free(d_in)
        ctx.free(d_out)
    return out


def test_inputs_shorter_than_the_filter(oracle):
    """Levels above pywt.dwt_max_level (the reference only warns, spiht_wrapper.py:163): the quantised arrays the GPU makes
    of inputs shorter than the filter against PyWavelets 1.1.1 (tests/golden/short_pywt.npz; q = 1000 so that the last
    bits of the coefficients decide) in float64 and float32, and a whole float32 image through encode_image at a level
    three above the maximum."""
    import spiht_amd
    from test_oracle import short_cases
    n = 0
    for cs in short_cases():
        q = 1000.0
        if cs["img"].dtype == np.float32:
            got = _gpu_dwt_f32(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], q)[0]
            want = (cs["arr"] * np.float32(q)).astype(np.int32)
        else:
            got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], q, None)[0]
            want = (cs["arr"] * q).astype(np.int32)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, (cs["wavelet"], cs["mode"], cs["img"].shape, str(cs["img"].dtype), len(bad), bad[:3])
        n += 1
    assert n == 195
    img = synth_image(5, 3, 40, 56).astype(np.float32)
    s = spiht_amd.SpihtSettings(wavelet="bior6.8")
    enc = spiht_amd.encode_image(img, s, level=4, max_bits=30000)   # pywt.dwt_max_level(40, 18) = 1
    arr, _ = oracle.wavedec2_array_f32(img, "bior6.8", "reflect", 4)
    g = oracle.geometry(40, 56, "bior6.8", 4)
    ref_bytes, ref_n = oracle.encode(oracle.quantize_f32(arr, 50.0), g["ll_h"], g["ll_w"], 30000)[:2]
    assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n


def test_single_precision_computed_modes_and_coiflets(oracle):
    """float32 pixels x the modes that compute their extension, periodization, and the coiflets' own single-precision filters:
    the int32 array the GPU hands to the coder against PyWavelets 1.1.1 (tests/golden/modes32_pywt.npz), and a whole
    encode_image against the oracle."""
    import spiht_amd
    from test_oracle import modes32_cases
    n = 0
    for cs in modes32_cases():
        got = _gpu_dwt_f32(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], 50.0)[0]
        bad = np.argwhere(got != cs["quant"])
        assert len(bad) == 0, (cs["wavelet"], cs["mode"], cs["img"].shape, len(bad), bad[:3])
        n += 1
    assert n == 27
    img = synth_image(9, 3, 70, 93).astype(np.float32)
    for wv, mode in (("coif2", "smooth"), ("db3", "periodization"), ("coif1", "reflect")):
        enc = spiht_amd.encode_image(img, spiht_amd.SpihtSettings(wavelet=wv, mode=mode), level=2, max_bits=40000)
        ref_bytes, ref_n, _ = oracle.encode_image(img, wv, mode, 2, 50.0, None, 40000)
        assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n, (wv, mode)


def test_wavelets_above_20_taps(oracle):
    """db11-38, sym11-20, coif4-17, dmey (22 to 102 taps; the reference takes any PyWavelets name, spiht_wrapper.py:163, :276)
    go through the two-pass levels with the filters read from device memory (dwt.hip: k_dwt_axis_ext / k_idwt_axis_per):
    the int32 array and the picture back against PyWavelets 1.1.1 (tests/golden/long_pywt.npz), all nine modes, float64 and
    float32; then pictures larger than a tile with channel scales and a colour model against the oracle, the coder included."""
    import spiht_amd
    from golden.make_golden import thin_out
    from test_oracle import long_cases, sha256_of
    n = 0
    for cs in long_cases():
        tag = (cs["wavelet"], cs["mode"], cs["level"], cs["img"].shape, cs["img"].dtype)
        if cs["f32"]:
            got = _gpu_dwt_f32(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"])[0]
        else:
            got = _gpu_dwt(cs["img"][None], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
        assert got.shape == cs["shape"], tag
        assert np.array_equal(sha256_of(got), cs["sha_quant"]), tag
        if not cs["f32"]:
            back = _gpu_idwt(thin_out(got, cs["seed"])[None], cs["H"], cs["W"], cs["wavelet"], cs["mode"], cs["level"], cs["q"], None)[0]
            assert back.shape == tuple(cs["back_shape"]) and np.array_equal(sha256_of(back), cs["sha_rec_img"]), tag
        n += 1
    assert n == 71
    imgs = np.stack([synth_image(70 + b, 3, 211, 318) for b in range(3)])
    for wv, mode, lv in (("db20", "reflect", 3), ("coif8", "periodization", 2), ("dmey", "symmetric", None), ("sym13", "antireflect", 2),
                         ("db38", "periodic", 1), ("coif17", "zero", 2)):
        got = _gpu_dwt(imgs, wv, mode, lv, 50.0, [1.0, 0.5, 2.0])
        for b in range(3):
            arr, _ = oracle.wavedec2_array(imgs[b], wv, mode, lv)
            assert np.array_equal(got[b], oracle.quantize(arr, 50.0, [1.0, 0.5, 2.0])), (wv, mode)
        rec = np.stack([thin_out(got[b], b) for b in range(3)])
        back = _gpu_idwt(rec, 211, 318, wv, mode, lv, 50.0, [1.0, 0.5, 2.0])
        for b in range(3):
            ref = oracle.waverec2_array(oracle.dequantize(rec[b], 50.0, [1.0, 0.5, 2.0]), 211, 318, wv, lv, mode)
            assert np.array_equal(back[b].view(np.uint64), ref.view(np.uint64)), (wv, mode)
        s = spiht_amd.SpihtSettings(wavelet=wv, mode=mode)
        enc = spiht_amd.encode_image(imgs[0], s, lv, 30000)
        ref_bytes, ref_n, _ = oracle.encode_image(imgs[0], wv, mode, lv, 50.0, None, 30000)
        assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n, (wv, mode)
        dec = spiht_amd.decode_image(enc, s)
        assert np.array_equal(dec, oracle.decode_image(ref_bytes, ref_n, 3, 211, 318, wv, enc.level, 50.0, None, mode=mode)), (wv, mode)
    # a colour model around a long filter: the change runs as a pass of its own on either side of the two-pass levels
    sc = spiht_amd.SpihtSettings(wavelet="sym16", color_model="IPT", quantization_scale=1.0, per_channel_quant_scales=[50.0, 15.0, 15.0])
    e2 = spiht_amd.encode_image(imgs[1], sc, 3, 200000)
    from spiht_amd import _lib, color_models
    from spiht_amd.batch import DeviceArray
    ctx = _lib.default_context()
    d = DeviceArray(ctx, imgs[1].shape, np.float64)
    d.upload(imgs[1])
    color_models.device_convert(ctx, d.ptr, 1, 211 * 318, "RGB", "IPT")   # the same kernel, as a call of its own
    ctx.synchronize()
    ref_bytes, ref_n, _ = oracle.encode_image(d.download(), "sym16", "reflect", 3, 1.0, [50.0, 15.0, 15.0], 200000)
    d.free()
    assert e2.encoded_bytes == ref_bytes and e2.max_n == ref_n
    d2 = spiht_amd.decode_image(e2, sc)
    assert np.abs(d2[:, :211, :318] - imgs[1]).mean() < 0.05
