"""GPU tests of the image-level API (spiht_wrapper counterpart, batched codec) against the CPU oracle, including
BASELINE.json's configurations at full size."""
import os

import numpy as np
import pytest

from conftest import synth_image

pytestmark = pytest.mark.gpu


def _oracle_roundtrip(O, img, settings, level, max_bits):
    data, max_n, g = O.encode_image(img, settings.wavelet, settings.mode, level, settings.quantization_scale,
                                    settings.per_channel_quant_scales, max_bits)
    c, H, W = img.shape
    dec = O.decode_image(data, max_n, c, H, W, settings.wavelet, level, settings.quantization_scale,
                         settings.per_channel_quant_scales)
    return data, max_n, dec


@pytest.mark.parametrize("cfg", [
    dict(c=1, H=32, W=32, level=2, max_bits=None),
    dict(c=3, H=48, W=64, level=None, max_bits=3000),
    dict(c=3, H=37, W=53, level=2, max_bits=12345),
    dict(c=3, H=64, W=96, level=3, max_bits=4000, q=1.0, mults=[100.0, 20.0, 20.0]),
    dict(c=1, H=96, W=128, level=None, max_bits=9999, wavelet="bior4.4", mode="symmetric"),
    dict(c=2, H=45, W=70, level=2, max_bits=None, wavelet="bior4.4", mode="symmetric", q=255.0, mults=[1.0, 0.2]),
    dict(c=1, H=160, W=144, level=None, max_bits=20001, wavelet="bior6.8"),
    dict(c=1, H=40, W=56, level=3, max_bits=777, wavelet="haar"),
])
def test_encode_image_decode_image_vs_oracle(oracle, cfg):
    import spiht_amd
    img = synth_image(2000 + cfg["H"], cfg["c"], cfg["H"], cfg["W"])
    s = spiht_amd.SpihtSettings(wavelet=cfg.get("wavelet", "bior2.2"), quantization_scale=cfg.get("q", 50.0),
                                mode=cfg.get("mode", "reflect"), per_channel_quant_scales=cfg.get("mults"))
    enc = spiht_amd.encode_image(img, s, level=cfg["level"], max_bits=cfg["max_bits"])
    ref_bytes, ref_n, ref_dec = _oracle_roundtrip(oracle, img, s, cfg["level"], cfg["max_bits"])
    assert isinstance(enc, spiht_amd.EncodingResult)
    assert (enc.h, enc.w, enc.c, enc.level, enc._encoding_version) == (cfg["H"], cfg["W"], cfg["c"], cfg["level"], "0.0.2")
    assert enc.max_n == ref_n
    assert enc.encoded_bytes == ref_bytes
    dec = spiht_amd.decode_image(enc, s)
    assert dec.dtype == np.float64 and dec.shape == ref_dec.shape
    assert np.array_equal(dec, ref_dec)
    # through the dict form (wrapper:83-89)
    dec2 = spiht_amd.decode_image(spiht_amd.EncodingResult.from_dict(enc.to_dict()), s)
    assert np.array_equal(dec2, dec)


def test_decode_rec_array_and_from_rec_arr(oracle):
    from spiht_amd.spiht_wrapper import SpihtSettings, decode_from_rec_arr, decode_rec_array, encode_image
    img = synth_image(7, 3, 50, 41)
    s = SpihtSettings()
    enc = encode_image(img, s, level=2, max_bits=5000)
    d = decode_rec_array(enc, s)
    g = oracle.geometry(50, 41, "bior2.2", 2)
    ref = oracle.decode(enc.encoded_bytes, enc.max_n, 3, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"])
    assert np.array_equal(d["rec_arr"], ref)
    assert d["spiht_metadata"] is None and (d["h"], d["w"], d["level"]) == (50, 41, 2)
    im = decode_from_rec_arr(d["rec_arr"], 50, 41, 2, s)
    assert np.array_equal(im, oracle.waverec2_array(oracle.dequantize(ref, 50.0), 50, 41, "bior2.2", 2))
    dm = decode_rec_array(enc, s, return_metadata=True)
    assert np.array_equal(dm["rec_arr"], ref) and dm["spiht_metadata"].shape == (8 * len(enc.encoded_bytes) + 1, 8)


def test_colour_model_ipt_self_consistency():
    """parity unpinned (colour-science absent): round trip only"""
    import spiht_amd
    img = synth_image(3, 3, 64, 64)
    s = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[100.0, 20.0, 20.0])
    enc = spiht_amd.encode_image(img, s, level=3, max_bits=None)
    dec = spiht_amd.decode_image(enc, s)
    assert np.abs(dec - img).mean() < 0.05  # coarse quantisation of the chroma channels (scale 20)
    with pytest.raises(ValueError):
        spiht_amd.encode_image(img, spiht_amd.SpihtSettings(color_model="ipt"))  # case-sensitive like the reference


def test_batch_codec_matches_single_image_path(oracle):
    import spiht_amd
    from spiht_amd.batch import BatchCodec
    B, c, H, W, level, max_bits = 5, 3, 72, 88, 3, 6000
    imgs = np.stack([synth_image(1000 + b, c, H, W) for b in range(B)])
    imgs[3] = 0.0  # an all-zero image in the batch
    s = spiht_amd.SpihtSettings()
    codec = BatchCodec(c, H, W, s, level, max_bits)
    results = codec.encode(imgs)
    assert len(results) == B
    for b in range(B):
        ref_bytes, ref_n, _ = _oracle_roundtrip(oracle, imgs[b], s, level, max_bits)
        assert results[b].encoded_bytes == ref_bytes and results[b].max_n == ref_n
        one = spiht_amd.encode_image(imgs[b], s, level=level, max_bits=max_bits)
        assert one.encoded_bytes == results[b].encoded_bytes
    dec = codec.decode(results)
    for b in range(B):
        assert np.array_equal(dec[b], spiht_amd.decode_image(results[b], s))
    # ragged prefixes in one decode batch (make_gif.py:46-55)
    cut = [spiht_amd.EncodingResult(r.encoded_bytes[:k], H, W, c, r.max_n, level) for r, k in zip(results, [0, 1, 333, 750, 17])]
    dec = codec.decode(cut)
    for b in range(B):
        assert np.array_equal(dec[b], spiht_amd.decode_image(cut[b], s))


def test_config1_512_gray(oracle):
    """BASELINE config 1 geometry: 512x512 gray, bior2.2 level 5, 0.5 bpp"""
    import spiht_amd
    img = synth_image(1000, 1, 512, 512)
    s = spiht_amd.SpihtSettings()
    mb = int(512 * 512 * 0.5)
    enc = spiht_amd.encode_image(img, s, level=5, max_bits=mb)
    ref_bytes, ref_n, ref_dec = _oracle_roundtrip(oracle, img, s, 5, mb)
    assert len(enc.encoded_bytes) == 16384 and enc.encoded_bytes == ref_bytes and enc.max_n == ref_n
    assert np.array_equal(spiht_amd.decode_image(enc, s), ref_dec)


def test_config2_1080p_rgb(oracle):
    """BASELINE config 2: 1920x1080 RGB, bior2.2 reflect level 7, 0.5 bpp (coefficient array 3x1111x1949, LL 13x19)"""
    import spiht_amd
    img = synth_image(1000, 3, 1080, 1920)
    s = spiht_amd.SpihtSettings()
    mb = int(1080 * 1920 * 0.5)
    enc = spiht_amd.encode_image(img, s, level=7, max_bits=mb)
    ref_bytes, ref_n, ref_dec = _oracle_roundtrip(oracle, img, s, 7, mb)
    assert len(enc.encoded_bytes) == 129600
    assert enc.max_n == ref_n and enc.encoded_bytes == ref_bytes
    dec = spiht_amd.decode_image(enc, s)
    assert np.array_equal(dec, ref_dec)
    # byte prefixes of the same stream (progressive decoding)
    for k in [1, 1000, 64800, 129599]:
        e2 = spiht_amd.EncodingResult(enc.encoded_bytes[:k], 1080, 1920, 3, enc.max_n, 7)
        g = oracle.geometry(1080, 1920, "bior2.2", 7)
        r = spiht_amd.spiht_wrapper.decode_rec_array(e2, s)["rec_arr"]
        assert np.array_equal(r, oracle.decode(e2.encoded_bytes, enc.max_n, 3, g["enc_h"], g["enc_w"], 13, 19))


def test_config3_1024_scaled_channels_batch(oracle):
    """BASELINE config 3 geometry: 1024x1024 RGB, level None (7), q=1 with per-channel scales [50,15,15], 0.1 bpp =
    104857 bits (7 pad bits).  Colour conversion is a host-side pre-step and is left out here."""
    import spiht_amd
    from spiht_amd.batch import BatchCodec
    s = spiht_amd.SpihtSettings(quantization_scale=1.0, per_channel_quant_scales=[50.0, 15.0, 15.0])
    mb = int(1024 * 1024 * 0.1)
    assert mb == 104857
    imgs = np.stack([synth_image(1000 + b, 3, 1024, 1024) for b in range(3)])
    codec = BatchCodec(3, 1024, 1024, s, None, mb)
    assert (codec.geom["enc_h"], codec.geom["enc_w"], codec.geom["ll_h"], codec.geom["level"]) == (1053, 1053, 12, 7)
    res = codec.encode(imgs)
    dec = codec.decode(res)
    for b in range(3):
        assert len(res[b].encoded_bytes) == 13108
        if b != 1:
            continue  # one full oracle comparison keeps the test short
        ref_bytes, ref_n, ref_dec = _oracle_roundtrip(oracle, imgs[b], s, None, mb)
        assert res[b].encoded_bytes == ref_bytes and res[b].max_n == ref_n
        assert np.array_equal(dec[b], ref_dec)


def test_config5_4096_bior68_bpp_sweep(oracle):
    """BASELINE config 5: 4096x4096 RGB, bior6.8 reflect, level 9 (above pywt's max level 7), bpp sweep.
    The code is embedded: every budget's stream is a bit prefix of the largest one, so one oracle encode pins all
    four; the decoder is checked against the oracle on each stream."""
    import spiht_amd
    img = synth_image(1000, 3, 4096, 4096)
    s = spiht_amd.SpihtSettings(wavelet="bior6.8")
    g = oracle.geometry(4096, 4096, "bior6.8", 9)
    assert (g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"]) == (4241, 4241, 24, 24)
    budgets = [int(4096 * 4096 * bpp) for bpp in (0.075, 0.1, 0.5, 1.0)]
    assert budgets == [1258291, 1677721, 8388608, 16777216]
    ref_bytes, ref_n, _ = oracle.encode_image(img, "bior6.8", "reflect", 9, 50.0, None, budgets[-1])
    ref_bits = oracle.bytes_to_bits(ref_bytes)
    for mb in budgets:
        enc = spiht_amd.encode_image(img, s, level=9, max_bits=mb)
        assert enc.max_n == ref_n and len(enc.encoded_bytes) == (mb + 7) // 8
        bits = oracle.bytes_to_bits(enc.encoded_bytes)
        assert np.array_equal(bits[:mb], ref_bits[:mb]) and not bits[mb:].any()
        rec = spiht_amd.spiht_wrapper.decode_rec_array(enc, s)["rec_arr"]
        assert np.array_equal(rec, oracle.decode(enc.encoded_bytes, enc.max_n, 3, 4241, 4241, 24, 24))
        if mb == budgets[1]:
            dec = spiht_amd.decode_image(enc, s)
            ref = oracle.waverec2_array(oracle.dequantize(rec, 50.0), 4096, 4096, "bior6.8", 9)
            assert np.array_equal(dec, ref)
            assert np.abs(dec - img).mean() < 0.1


def test_more_images_than_slots(oracle):
    """B larger than the number of list-scratch slots: every workgroup codes several images in turn (the
    encoder has <= 256 slots, the decoder <= 2048)."""
    import spiht_amd
    from spiht_amd.batch import BatchCodec
    B, c, H, W, level, max_bits = 2300, 1, 16, 24, 1, 700
    rng = np.random.default_rng(1)
    imgs = rng.integers(0, 256, (B, c, H, W)).astype(np.float64) / 255
    imgs[7] = 0.0
    s = spiht_amd.SpihtSettings()
    codec = BatchCodec(c, H, W, s, level, max_bits)
    res = codec.encode(imgs)
    dec = codec.decode(res)
    for b in list(range(0, B, 97)) + [7, 255, 256, 257, 2047, 2048, 2049, B - 1]:
        ref_bytes, ref_n, ref_dec = _oracle_roundtrip(oracle, imgs[b], s, level, max_bits)
        assert res[b].encoded_bytes == ref_bytes and res[b].max_n == ref_n, b
        assert np.array_equal(dec[b], ref_dec), b


def test_raw_batch_entry_points(oracle):
    """spiht_encode_batch_i32 / spiht_decode_batch_i32 / spiht_pyramid_batch_i32 on device buffers"""
    import ctypes as C
    from conftest import synth_coeffs
    from spiht_amd import _lib
    from spiht_amd.batch import DeviceArray
    ctx, L = _lib.default_context(), _lib.lib()
    B, c, h, w, lh, lw, mb = 6, 3, 37, 53, 5, 7, 3001
    xs = np.stack([synth_coeffs(50 + b, c, h, w, lh, lw, scale=800.0) for b in range(B)])
    slot = ((mb + 7) // 8 + 3) & ~3
    d_x = DeviceArray(ctx, xs.shape, np.int32)
    d_out = DeviceArray(ctx, (B, slot), np.uint8)
    d_nbits = DeviceArray(ctx, (B,), np.uint64)
    d_nbytes = DeviceArray(ctx, (B,), np.uint64)
    d_maxn = DeviceArray(ctx, (B,), np.uint8)
    d_rec = DeviceArray(ctx, xs.shape, np.int32)
    d_dm = DeviceArray(ctx, xs.shape, np.uint8)
    d_lm = DeviceArray(ctx, xs.shape, np.uint8)
    d_mx = DeviceArray(ctx, (B,), np.uint32)
    vp = C.c_void_p
    d_x.upload(xs)
    _lib.check(L.spiht_encode_batch_i32(ctx.handle, vp(d_x.ptr), B, c, h, w, lh, lw, mb, vp(d_out.ptr), slot, vp(d_nbits.ptr),
                                        vp(d_maxn.ptr)))
    _lib.check(L.spiht_nbits_to_nbytes(ctx.handle, vp(d_nbits.ptr), B, vp(d_nbytes.ptr)))
    _lib.check(L.spiht_decode_batch_i32(ctx.handle, vp(d_out.ptr), slot, vp(d_nbytes.ptr), vp(d_maxn.ptr), B, c, h, w, lh, lw,
                                        vp(d_rec.ptr)))
    _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(d_x.ptr), B, c, h, w, lh, lw, vp(d_dm.ptr), vp(d_lm.ptr), vp(d_mx.ptr)))
    ctx.synchronize()
    out, nbits, maxn, rec, mx = d_out.download(), d_nbits.download(), d_maxn.download(), d_rec.download(), d_mx.download()
    for b in range(B):
        ref, ref_n, ref_nb = oracle.encode_nbits(xs[b], lh, lw, mb)
        assert int(nbits[b]) == ref_nb and int(maxn[b]) == ref_n
        assert out[b, :(ref_nb + 7) // 8].tobytes() == ref and not out[b, (ref_nb + 7) // 8:].any()
        assert np.array_equal(rec[b], oracle.decode(ref, ref_n, c, h, w, lh, lw))
        assert int(mx[b]) == int(np.abs(xs[b]).max())
    # a too-small slot is reported by synchronize(), not silently truncated
    _lib.check(L.spiht_encode_batch_i32(ctx.handle, vp(d_x.ptr), B, c, h, w, lh, lw, 10 ** 9, vp(d_out.ptr), slot, vp(d_nbits.ptr),
                                        vp(d_maxn.ptr)))
    with pytest.raises(ValueError):
        ctx.synchronize()
    ctx.synchronize()  # the error word was cleared


def test_progressive_prefixes_in_one_batch(oracle):
    """make_gif.py:46-61: byte prefixes of one stream, decoded together; quality never gets worse"""
    import spiht_amd
    from spiht_amd.batch import BatchCodec
    c, H, W = 3, 120, 168
    img = synth_image(77, c, H, W)
    s = spiht_amd.SpihtSettings()
    codec = BatchCodec(c, H, W, s, None, None)
    res = codec.encode(img[None])[0]
    ks = [0, 1, 50, 400, 3000, len(res.encoded_bytes) // 2, len(res.encoded_bytes)]
    for one_walk in (True, False):  # one walk of the stream with per-node replay / K prefixes as K streams of a batch
        ims = codec.decode_prefixes(res, ks, one_walk=one_walk)
        errs = []
        for k, im in zip(ks, ims):
            one = spiht_amd.decode_image(spiht_amd.EncodingResult(res.encoded_bytes[:k], H, W, c, res.max_n, None), s)
            assert np.array_equal(im[:, :H, :W], one), (one_walk, k)
            errs.append(float(np.abs(one - img).mean()))
        assert errs[-1] < errs[3] < errs[0] and errs[-1] < 0.01
    # order of the lengths is the caller's; lengths past the end mean the whole stream
    ks2 = [3000, 7, len(res.encoded_bytes) + 100, 400]
    ims2 = codec.decode_prefixes(res, ks2)
    for k, im in zip(ks2, ims2):
        one = spiht_amd.decode_image(spiht_amd.EncodingResult(res.encoded_bytes[:k], H, W, c, res.max_n, None), s)
        assert np.array_equal(im[:, :H, :W], one), k


def test_random_images_settings_and_budgets(oracle):
    """seeded random sweep over image sizes, wavelets, extension modes, levels, quantisation and per-channel scales,
    pixel dtype (float64 / float32) and bit budgets: streams, max_n and decoded images against the oracle"""
    import spiht_amd
    rng = np.random.default_rng(4102026)
    wavelets = ["bior2.2", "bior2.2", "bior4.4", "bior6.8", "haar", "db4", "sym5", "coif2", "rbio3.3", "db10", "bior3.1"]
    for case in range(int(__import__("os").environ.get("SPIHT_SWEEP_N", "40"))):
        c = int(rng.integers(1, 4))
        wv = wavelets[int(rng.integers(len(wavelets)))]
        F = len(oracle.wavelet_filters(wv)[0])
        H, W = int(rng.integers(2 * F + 8, 150)), int(rng.integers(2 * F + 8, 150))
        f32 = bool(case % 3 == 2)
        mode = ["reflect", "symmetric", "periodic", "zero", "constant", "smooth", "antisymmetric", "antireflect",
                "periodization"][int(rng.integers(9))]
        maxlv = int(np.floor(np.log2(min(H, W) / (F - 1)))) if F > 2 else int(np.floor(np.log2(min(H, W))))
        level = [None, 1, 2, max(1, maxlv), maxlv + 2][int(rng.integers(5))]   # (the last: inputs shorter than the filter)
        q = float([50.0, 10.0, 255.0, 3.3][int(rng.integers(4))])
        mults = None if rng.integers(2) else [float(v) for v in rng.uniform(0.2, 3.0, c).round(2)]
        mb = [None, int(rng.integers(8, 600)), int(rng.integers(600, 40000))][case % 3 if not f32 else int(rng.integers(3))]
        img = synth_image(int(rng.integers(1 << 30)), c, H, W)
        if f32:
            img = img.astype(np.float32)
        s = spiht_amd.SpihtSettings(wavelet=wv, quantization_scale=q, mode=mode, per_channel_quant_scales=mults)
        tag = "case %d: c=%d %dx%d %s %s level=%s q=%g mults=%s max_bits=%s %s" % (case, c, H, W, wv, mode, level, q, mults, mb,
                                                                                    img.dtype)
        try:
            enc = spiht_amd.encode_image(img, s, level=level, max_bits=mb)
        except spiht_amd.spiht.PanicException:
            # an LL block of a single row / column (e.g. haar at its maximal level): the Rust core asserts ll > 1
            with pytest.raises(oracle.OraclePanic):
                oracle.encode_image(img, wv, mode, level, q, mults, mb)
            continue
        ref_bytes, ref_n, g = oracle.encode_image(img, wv, mode, level, q, mults, mb)
        assert enc.max_n == ref_n and enc.encoded_bytes == ref_bytes, tag
        dec = spiht_amd.decode_image(enc, s)
        ref = oracle.decode_image(ref_bytes, ref_n, c, H, W, wv, level, q, mults, mode=mode)
        assert np.array_equal(dec, ref), tag


@pytest.mark.gpu
def test_cli_encode_save_load_decode(tmp_path, oracle):
    """`python -m spiht_amd.encode_decode` (reference: encode_decode.py:17-90): file in, file out, and the saved
    container decodes to the same picture; the stream equals the oracle's for the same settings."""
    pytest.importorskip("PIL")
    import spiht_amd
    from spiht_amd import utils
    from spiht_amd.encode_decode import build_parser, main
    img = np.round(synth_image(77, 3, 120, 200) * 255) / 255
    utils.imsave(tmp_path / "in.png", img)
    args = build_parser().parse_args([str(tmp_path / "in.png"), "--bpp", "0.5", "--color_model", "RGB",
                                      "--per_channel_quant_scales", "1., 1., 1.", "--out", str(tmp_path / "out.png"),
                                      "--save", str(tmp_path / "e.spiht")])
    enc, dec = main(args)
    assert (enc.h, enc.w, enc.c, enc.level) == (120, 200, 3, 3) and len(enc.encoded_bytes) == (round(0.5 * 120 * 200) + 7) // 8
    assert dec.shape == (3, 120, 200) and float(((img - dec) ** 2).mean()) < 1e-2
    s = spiht_amd.SpihtSettings(quantization_scale=255.0, color_model="RGB", per_channel_quant_scales=[1.0, 1.0, 1.0])
    ref_bytes, ref_n, _ = _oracle_roundtrip(oracle, utils.imload(tmp_path / "in.png"), s, 3, round(0.5 * 120 * 200))
    assert enc.encoded_bytes == ref_bytes and enc.max_n == ref_n
    out = utils.imload(tmp_path / "out.png")
    assert out.shape == (3, 120, 200)
    args2 = build_parser().parse_args([str(tmp_path / "in.png"), "--color_model", "RGB", "--per_channel_quant_scales",
                                       "1., 1., 1.", "--out", str(tmp_path / "out2.png"), "--load", str(tmp_path / "e.spiht")])
    enc2, dec2 = main(args2)
    assert enc2 == enc and np.array_equal(dec2, dec)
    assert np.array_equal(utils.imload(tmp_path / "out2.png"), out)


@pytest.mark.gpu
def test_transform_and_pyramid_queued_separately():
    """include/spiht_hip.h "two halves": spiht_dwt_pyramid_batch_f64 without the pyramid + spiht_pyramid_batch_i32 without
    the max pass give what the fused call gives"""
    import ctypes as C
    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray
    from spiht_amd.spiht_wrapper import SpihtSettings
    L, vp = _lib.lib(), C.c_void_p
    ctx = _lib.Context(0)
    B, c, H, W = 2, 3, 70, 90
    cd = BatchCodec(c, H, W, SpihtSettings(), None, 5000, ctx=ctx)
    g = cd.geom
    n = c * g["enc_h"] * g["enc_w"]
    d_img = DeviceArray(ctx, (B, c, H, W), np.float64)
    d_img.upload(np.stack([synth_image(31 + b, c, H, W) for b in range(B)]))
    outs = []
    for split in (False, True):
        co, dm, lm, ma = (DeviceArray(ctx, (B, n), np.int32), DeviceArray(ctx, (B, n), np.uint8),
                          DeviceArray(ctx, (B, n), np.uint8), DeviceArray(ctx, (B,), np.uint32))
        for a in (co, dm, lm, ma):
            ctx.memset(a.ptr, 0, a.nbytes)
        q = float(cd.settings.quantization_scale)
        _lib.check(L.spiht_dwt_pyramid_batch_f64(ctx.handle, vp(d_img.ptr), B, c, H, W, cd.wid, cd.mid, cd._lv, q, cd._mults_p,
                                                 vp(co.ptr), vp(None if split else dm.ptr), vp(None if split else lm.ptr), vp(ma.ptr)))
        if split:
            _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(co.ptr), B, c, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"],
                                                 vp(dm.ptr), vp(lm.ptr), vp(None)))
        ctx.synchronize()
        outs.append([a.download() for a in (co, dm, lm, ma)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][3].all() and outs[0][1].any()


@pytest.mark.gpu
def test_device_colour_conversion_and_config3_batch():
    """SURVEY.md 8 f-2 / BASELINE config 3 (IPT, per-channel scales [50,15,15], 0.1 bpp): the colour model is changed on
    the device.  Colour parity is unpinned (colour-science is not available), so the device kernel is held to the host
    implementation of the published transform: a few ulp, and the same coded picture up to rare quantiser flips."""
    import spiht_amd
    from spiht_amd import _lib, color_models
    from spiht_amd.batch import BatchCodec, DeviceArray
    ctx = _lib.default_context()
    rng = np.random.default_rng(5)
    x = rng.random((2, 3, 33, 47))
    x[0, :, 0, :5] = 0.0
    from oracle import oracle as O
    for src, dst in (("RGB", "IPT"), ("IPT", "RGB")):
        inp = x if src == "RGB" else np.stack([color_models.convert(im, "RGB", "IPT") for im in x])
        d = DeviceArray(ctx, inp.shape, np.float64)
        d.upload(inp)
        color_models.device_convert(ctx, d.ptr, 2, 33 * 47, src, dst)
        ctx.synchronize()
        got = d.download()
        ref = np.stack([color_models.convert(im, src, dst) for im in inp])
        assert np.abs(got - ref).max() < 1e-13, float(np.abs(got - ref).max())
        # the oracle (oracle/color_oracle.c: the same transform with the C library's pow -- its own arithmetic, nothing of
        # the product): the device's power function is within 4 units in the last place (tests/test_oracle.py), the
        # values are O(1) and the output matrix has entries up to 4.9 -> a few 1e-15 in the result
        twin = np.stack([O.color3(im, *color_models._params(src, dst)) for im in inp])
        assert np.abs(got - twin).max() < 2e-14, float(np.abs(got - twin).max())
    # the published known answer of the XYZ -> IPT half (colour-science's XYZ_to_IPT example), on the device kernel
    import ctypes as C
    xyz = np.tile(np.array([0.20654008, 0.12197225, 0.05136952]).reshape(1, 3, 1), (1, 1, 64))
    d = DeviceArray(ctx, xyz.shape, np.float64)
    d.upload(xyz)
    A, M = np.ascontiguousarray(color_models._XYZ2LMS), np.ascontiguousarray(color_models._LMS2IPT)
    _lib.check(_lib.lib().spiht_color3_batch_f64(ctx.handle, C.c_void_p(d.ptr), C.c_void_p(d.ptr), 1, 64, C.c_void_p(A.ctypes.data),
                                                 C.c_void_p(M.ctypes.data), color_models.IPT_EXPONENT))
    ctx.synchronize()
    assert np.abs(d.download()[0, :, 0] - np.array([0.38426191, 0.38487306, 0.18886838])).max() < 5e-9
    s = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0])
    H = W = 256
    imgs = np.stack([synth_image(40 + b, 3, H, W) for b in range(2)])
    mb = int(H * W * 0.5)
    codec = BatchCodec(3, H, W, s, None, mb)
    res = codec.encode(imgs)
    dec = codec.decode(res)
    for b in range(2):
        one = spiht_amd.encode_image(imgs[b], s, None, mb)          # host colour conversion, same coder
        assert len(res[b].encoded_bytes) == len(one.encoded_bytes) == (mb + 7) // 8 and res[b].max_n == one.max_n
        back = spiht_amd.decode_image(one, s)
        assert np.abs(dec[b] - back).max() < 0.05 and np.abs(dec[b][:, :H, :W] - imgs[b]).mean() < 0.06
    with pytest.raises(ValueError):
        BatchCodec(1, 32, 32, spiht_amd.SpihtSettings(color_model="IPT"), None, 500).encode(np.zeros((1, 1, 32, 32)))


@pytest.mark.gpu
def test_config4_shard_256_distinct_1080p():
    """BASELINE config 4's per-GPU shard, as bench.py times it: 256 DISTINCT 1920x1080 RGB images (image i: seed
    1000 + i), bior2.2 reflect level 7, 0.5 bpp, through the pipelined schedule bench.py times -- the library's
    own (spiht_pipeline_submit, csrc/pipeline.cpp; spiht_amd.batch.Pipeline) --, three consecutive steps so that both buffer
    sets and both list-coding contexts are used and the second set's output passes through the split inverse transform
    beside the next step's decoder.  EVERY image's stream, max_n and decoded picture is compared with the CPU oracle (one
    oracle round trip per image, spread over the host cores)."""
    import spiht_amd
    from parity_workers import digest, oracle_roundtrip_job, pool, synth_u8_job
    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray, Pipeline
    B = int(os.environ.get("SPIHT_TEST_SHARD", "256"))
    c, H, W, level = 3, 1080, 1920, 7
    mb = int(H * W * 0.5)
    s = spiht_amd.SpihtSettings()
    ctx = _lib.default_context()
    codec = BatchCodec(c, H, W, s, level, mb, ctx=ctx)
    g, slot = codec.geom, codec.slot_stride
    d_img = DeviceArray(ctx, (B, c, H, W), np.float64)
    per = c * H * W * 8
    with pool() as ex:
        for b, u8 in enumerate(ex.map(synth_u8_job, [(1000 + b, c, H, W) for b in range(B)])):
            d_img.upload(u8 / 255, offset_bytes=b * per)
        d_out, d_nbits, d_maxn = DeviceArray(ctx, (B, slot), np.uint8), DeviceArray(ctx, (B,), np.uint64), DeviceArray(ctx, (B,), np.uint8)
        d_rec = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)
        pipe = Pipeline(codec, B)
        for _ in range(3):  # steps 1 and 2 leave through different buffer sets; the last one is what is compared
            pipe.submit(d_img.ptr, d_out.ptr, d_nbits.ptr, d_maxn.ptr, d_rec.ptr)
        pipe.synchronize()
        nbits, maxn, streams = d_nbits.download(), d_maxn.download(), d_out.download()
        assert (nbits == mb).all()
        jobs = [((1000 + b, c, H, W), "bior2.2", "reflect", level, 50.0, None, mb) for b in range(B)]
        rec1 = np.empty((c, g["rec_h"], g["rec_w"]), np.float64)
        bad = []
        for b, (data, mn, dg) in enumerate(ex.map(oracle_roundtrip_job, jobs)):
            ctx.download(rec1, d_rec.ptr + b * c * g["rec_h"] * g["rec_w"] * 8)
            if streams[b, :mb // 8].tobytes() != data or int(maxn[b]) != mn or digest(rec1) != dg:
                bad.append(b)
    pipe.close()
    assert not bad, "images that differ from the oracle: %s" % bad[:16]


@pytest.mark.gpu
def test_config3_full_ipt_256_batch():
    """BASELINE config 3 in full: 256 x 1024x1024 RGB, IPT colour model, quantization_scale 1 with per-channel scales
    [50,15,15], level None (7), 0.1 bpp = 104 857 bits (7 pad bits), one batch.  Colour parity itself is unpinned
    (colour-science is not available anywhere, the reference has no colour test): the IPT pixels the GPU codes are
    downloaded (the stand-alone colour kernel) and THOSE go to the oracle, so the transform, the quantiser and the coder
    are checked on exactly what they coded -- streams and decoded IPT pictures bit for bit on a seeded sample of 32
    images, bit counts and sizes on all 256; the way back to RGB is held to the host implementation of the same
    published transform."""
    import spiht_amd
    from parity_workers import digest, oracle_roundtrip_job, pool, synth_u8_job
    from spiht_amd import _lib, color_models
    from spiht_amd.batch import BatchCodec, DeviceArray
    B = int(os.environ.get("SPIHT_TEST_CFG3_BATCH", "256"))
    c, H, W = 3, 1024, 1024
    mb = int(H * W * 0.1)
    assert mb == 104857
    mults = [50.0, 15.0, 15.0]
    s = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=mults)
    s_raw = spiht_amd.SpihtSettings(quantization_scale=1.0, per_channel_quant_scales=mults)  # same coder, no colour step
    ctx = _lib.default_context()
    codec, codec_raw = BatchCodec(c, H, W, s, None, mb, ctx=ctx), BatchCodec(c, H, W, s_raw, None, mb, ctx=ctx)
    assert (codec.geom["enc_h"], codec.geom["enc_w"], codec.geom["ll_h"], codec.geom["level"]) == (1053, 1053, 12, 7)
    rng = np.random.default_rng(3)
    sample = sorted(int(v) for v in rng.choice(B, size=min(32, B), replace=False))
    with pool() as ex:
        imgs = np.empty((B, c, H, W), np.float64)
        for b, u8 in enumerate(ex.map(synth_u8_job, [(1000 + b, c, H, W) for b in range(B)])):
            imgs[b] = u8 / 255
        res = codec.encode(imgs)                   # RGB in: colour model changed on the device, then coded
        assert all(len(r.encoded_bytes) == 13108 and (r.h, r.w, r.c) == (H, W, c) for r in res)
        # the IPT pixels of the sample, from the stand-alone colour kernel
        d = DeviceArray(ctx, (len(sample), c, H, W), np.float64)
        d.upload(imgs[sample])
        color_models.device_convert(ctx, d.ptr, len(sample), H * W, "RGB", "IPT")
        ctx.synchronize()
        ipt = d.download()
        d.free()
        del imgs
        dec_ipt = codec_raw.decode([res[b] for b in sample])  # decoded pictures before the way back to RGB
        dec_rgb = codec.decode([res[b] for b in sample[:4]])
        jobs = [(ipt[k], "bior2.2", "reflect", None, 1.0, mults, mb) for k in range(len(sample))]
        for k, (data, mn, dg) in enumerate(ex.map(oracle_roundtrip_job, jobs)):
            b = sample[k]
            assert res[b].encoded_bytes == data and res[b].max_n == mn, "stream of image %d differs from the oracle's" % b
            assert digest(dec_ipt[k]) == dg, "decoded image %d differs from the oracle's" % b
    for k in range(len(dec_rgb)):
        back = color_models.convert(dec_ipt[k], "IPT", "RGB")
        assert np.abs(dec_rgb[k] - back).max() < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(H=96, W=160, wavelet="bior2.2", mode="reflect", level=3, mults=[50.0, 15.0, 15.0], q=1.0, mb=9000),
    dict(H=293, W=501, wavelet="bior2.2", mode="reflect", level=None, mults=None, q=50.0, mb=None),      # odd sizes, several strips
    dict(H=67, W=131, wavelet="bior4.4", mode="symmetric", level=2, mults=[1.0, 0.2, 0.2], q=255.0, mb=20000),
    dict(H=150, W=145, wavelet="bior6.8", mode="reflect", level=2, mults=None, q=50.0, mb=30000),
    dict(H=64, W=70, wavelet="bior2.2", mode="zero", level=1, mults=None, q=50.0, mb=5000),               # one level: LL from rec
    dict(H=40, W=56, wavelet="haar", mode="periodic", level=2, mults=None, q=50.0, mb=4000),
    dict(H=33, W=47, wavelet="bior2.2", mode="constant", level=1, mults=None, q=50.0, mb=3000),           # smaller than a strip
    dict(H=300, W=260, wavelet="bior2.2", mode="reflect", level=4, mults=[100.0, 20.0, 20.0], q=1.0, mb=40000),
])
def test_fused_colour_equals_separate_colour_pass(cfg):
    """the colour model change inside level 1 of the transforms (k_dwt1_color / k_idwt1_color) against the
    same change as a pass of its own (k_color3) around the plain transform: the same function of the same pixels, so
    streams, start planes and decoded pictures are equal bit for bit."""
    import spiht_amd
    from spiht_amd import _lib, color_models
    from spiht_amd.batch import BatchCodec, DeviceArray
    H, W, B = cfg["H"], cfg["W"], 3
    kw = dict(wavelet=cfg["wavelet"], mode=cfg["mode"], quantization_scale=cfg["q"], per_channel_quant_scales=cfg["mults"])
    s_fused = spiht_amd.SpihtSettings(color_model="IPT", **kw)
    s_plain = spiht_amd.SpihtSettings(**kw)
    ctx = _lib.default_context()
    imgs = np.stack([synth_image(600 + b, 3, H, W) for b in range(B)])
    imgs[1, :, :5, :7] = 0.0      # zero and saturated patches (the signed power at 0)
    imgs[2, 0] = 1.0
    fused, plain = BatchCodec(3, H, W, s_fused, cfg["level"], cfg["mb"], ctx=ctx), BatchCodec(3, H, W, s_plain, cfg["level"], cfg["mb"], ctx=ctx)
    d = DeviceArray(ctx, imgs.shape, np.float64)
    d.upload(imgs)
    color_models.device_convert(ctx, d.ptr, B, H * W, "RGB", "IPT")
    ctx.synchronize()
    ipt = d.download()
    d.free()
    res_f, res_p = fused.encode(imgs), plain.encode(ipt)
    for b in range(B):
        assert res_f[b].max_n == res_p[b].max_n and res_f[b].encoded_bytes == res_p[b].encoded_bytes, (cfg, b)
    dec_p = plain.decode(res_p)                    # pictures in the coded colour model ...
    d = DeviceArray(ctx, dec_p.shape, np.float64)
    d.upload(dec_p)
    color_models.device_convert(ctx, d.ptr, B, dec_p.shape[2] * dec_p.shape[3], "IPT", "RGB")
    ctx.synchronize()
    back = d.download()                            # ... and back to RGB by the pass of its own
    d.free()
    dec_f = fused.decode(res_f)
    assert np.array_equal(dec_f, back), (cfg, float(np.abs(dec_f - back).max()))
    # the single-image drop-in calls take the same path
    one = spiht_amd.encode_image(imgs[0], s_fused, cfg["level"], cfg["mb"])
    assert one.encoded_bytes == res_f[0].encoded_bytes and one.max_n == res_f[0].max_n
    assert np.array_equal(spiht_amd.decode_image(one, s_fused), dec_f[0])
    # and the published transform on the host agrees to rounding
    host = color_models.convert(imgs[0], "RGB", "IPT")
    assert np.abs(host - ipt[0]).max() < 1e-12


@pytest.mark.gpu
def test_colour_model_is_per_caller_across_threads(oracle):
    """The colour model is state of the (shared, per-device default) context, set and cleared around each image call
    (color_models.fused).  Threads that code with and without a colour model at the same time must each get exactly what
    they get alone: the context's mutex is held from the set to the clear (spiht_ctx_lock / spiht_ctx_unlock)."""
    import threading
    import spiht_amd
    c, H, W, level, mb = 3, 72, 104, 3, 9000
    imgs = [synth_image(900 + k, c, H, W) for k in range(4)]
    s_rgb = spiht_amd.SpihtSettings()
    s_ipt = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0])
    alone = {}
    for name, s in (("rgb", s_rgb), ("ipt", s_ipt)):
        for k, im in enumerate(imgs):
            e = spiht_amd.encode_image(im, s, level, mb)
            alone[name, k] = (e.encoded_bytes, e.max_n, spiht_amd.decode_image(e, s))
    assert alone["rgb", 0][0] != alone["ipt", 0][0]
    errors = []

    def work(name, s, reps):
        try:
            for it in range(reps):
                k = it % len(imgs)
                e = spiht_amd.encode_image(imgs[k], s, level, mb)
                d = spiht_amd.decode_image(e, s)
                if (e.encoded_bytes, e.max_n) != alone[name, k][:2] or not np.array_equal(d, alone[name, k][2]):
                    errors.append((name, it))
        except Exception as ex:  # noqa: BLE001
            errors.append((name, repr(ex)))

    threads = [threading.Thread(target=work, args=(n, s, 60)) for n, s in (("rgb", s_rgb), ("ipt", s_ipt), ("rgb", s_rgb), ("ipt", s_ipt))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(3, 96, 136, 5, 9000, 7, None, False), (1, 61, 47, 2, 2500, 4, 1, False), (3, 130, 200, 3, None, 4, 4, False),
                                 (3, 96, 136, 3, 9000, 4, 3, True), (2, 200, 264, 9, 40000, 6, None, False), (3, 130, 75, 3, None, 4, None, False),
                                 (1, 61, 47, 1, 2500, 5, None, False)])
def test_c_pipeline_matches_fused(cfg):
    """include/spiht_hip.h spiht_pipeline_*: the pipelined schedule queued by the library itself (csrc/pipeline.cpp) codes the
    same streams and pictures as the fused calls -- several steps with different contents through both buffer sets, one
    level (no occupancy words), unlimited budget, a colour model."""
    import spiht_amd
    from spiht_amd import _lib
    from spiht_amd.batch import BatchCodec, DeviceArray, Pipeline
    c, H, W, B, mb, steps, level, colour = cfg
    s = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0]) if colour \
        else spiht_amd.SpihtSettings()
    ctx = _lib.default_context()
    codec = BatchCodec(c, H, W, s, level, mb, ctx=ctx)
    g = codec.geom
    pl = Pipeline(codec, B, own_context=(level == 4))  # (one case with the HBM-bound passes on a context of the pipeline's own)
    imgs = [np.stack([synth_image(300 * st + b, c, H, W) for b in range(B)]) for st in range(steps)]
    d_imgs = [DeviceArray(ctx, (B, c, H, W), np.float64) for _ in range(steps)]
    d_recs = [DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64) for _ in range(steps)]
    d_outs = [DeviceArray(ctx, (B, pl.slot_stride), np.uint8) for _ in range(steps)]
    d_nbits = [DeviceArray(ctx, (B,), np.uint64) for _ in range(steps)]
    d_maxn = [DeviceArray(ctx, (B,), np.uint8) for _ in range(steps)]
    for st in range(steps):
        d_imgs[st].upload(imgs[st])
    ctx.synchronize()
    for st in range(steps):
        pl.submit(d_imgs[st].ptr, d_outs[st].ptr, d_nbits[st].ptr, d_maxn[st].ptr, d_recs[st].ptr)
    pl.synchronize()
    for st in range(steps):
        res = codec.encode(imgs[st])
        nb, out, mn = d_nbits[st].download(), d_outs[st].download(), d_maxn[st].download()
        for b in range(B):
            assert (int(nb[b]) + 7) // 8 == len(res[b].encoded_bytes) and int(mn[b]) == res[b].max_n
            assert out[b, :len(res[b].encoded_bytes)].tobytes() == res[b].encoded_bytes
        assert np.array_equal(d_recs[st].download(), np.stack(codec.decode(res)))
    pl.close()
    for d in d_imgs + d_recs + d_outs + d_nbits + d_maxn:
        d.free()


@pytest.mark.gpu
def test_returned_arrays_are_the_callers(oracle):
    """decode_image / decode return NEW arrays (spiht_wrapper.py:192-216, lib.rs:35-42).  Here they are page-locked memory from
    the library's pool (spiht_host_alloc): an array a caller still holds is never handed out again, one that was dropped is
    (that is the point of the pool), and the arrays behave like any numpy array (writable, sliceable, survive arithmetic)."""
    import gc
    import spiht_amd
    from spiht_amd import _lib
    c, H, W, level, mb = 3, 360, 512, 4, 80000  # 4.4 MB per decoded picture: above result_array's 1 MB threshold
    s = spiht_amd.SpihtSettings()
    imgs = [synth_image(40 + k, c, H, W) for k in range(3)]
    encs = [spiht_amd.encode_image(im, s, level, mb) for im in imgs]
    refs = [oracle.decode_image(e.encoded_bytes, e.max_n, c, H, W, "bior2.2", level, 50.0, None) for e in encs]
    held = [spiht_amd.decode_image(e, s) for e in encs]          # three arrays alive at once
    ptrs = [a.ctypes.data for a in held]
    assert len(set(ptrs)) == 3
    for a, r in zip(held, refs):
        assert a.dtype == np.float64 and a.flags.writeable and np.array_equal(a, r)
    more = spiht_amd.decode_image(encs[0], s)                       # a fourth one while the three are held
    assert more.ctypes.data not in ptrs and all(np.array_equal(a, r) for a, r in zip(held, refs))
    view = held[1][:, 10:20]                                        # a view keeps its buffer alive
    p1 = held[1].ctypes.data
    held[1] = None
    gc.collect()
    again = spiht_amd.decode_image(encs[2], s)
    assert again.ctypes.data != p1 and np.array_equal(view, refs[1][:, 10:20])
    del view
    gc.collect()
    back = spiht_amd.decode_image(encs[2], s)                       # now the dropped buffer may come back (the pool's purpose)
    assert np.array_equal(back, refs[2])
    back += 1.0                                                     # in-place arithmetic on the caller's array
    assert np.array_equal(back, refs[2] + 1.0) and np.array_equal(again, refs[2])
    raw = spiht_amd.decode(encs[0].encoded_bytes, encs[0].max_n, c, *[oracle.geometry(H, W, "bior2.2", level)[k] for k in ("enc_h", "enc_w", "ll_h", "ll_w")])
    assert raw.dtype == np.int32 and raw.flags.c_contiguous and raw.flags.writeable
    big = _lib.result_array((2 << 20,), np.uint8)                   # the helper itself; small requests are ordinary arrays
    small = _lib.result_array((100,), np.uint8)
    assert big.shape == (2 << 20,) and small.shape == (100,)


@pytest.mark.gpu
def test_pipeline_leaves_nothing_on_a_borrowed_context(oracle):
    """spiht_pipeline_create_on runs the HBM-bound passes on the CALLER's context.  Its colour model and options are put on
    that context only while one of its calls queues work: after an IPT pipeline has run there, a plain 3-channel encode /
    decode on the same context codes RGB (stream and picture equal the oracle's); options the caller had set are what they
    were; and a colour block of the caller's between two steps (set and clear on the shared context) does not take the
    colour model away from the pipeline's later steps."""
    import spiht_amd
    from spiht_amd import _lib, color_models
    from spiht_amd.batch import BatchCodec, DeviceArray, Pipeline
    c, H, W, B, mb, level, steps = 3, 96, 136, 3, 9000, 3, 4
    ctx = _lib.Context(0)
    s_ipt = spiht_amd.SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0])
    s_rgb = spiht_amd.SpihtSettings()
    codec_ipt, codec_rgb = BatchCodec(c, H, W, s_ipt, level, mb, ctx=ctx), BatchCodec(c, H, W, s_rgb, level, mb, ctx=ctx)
    g = codec_ipt.geom
    ctx.set_option("idwt_groups", 2)
    pl = Pipeline(codec_ipt, B)
    imgs = [np.stack([synth_image(800 + 10 * st + b, c, H, W) for b in range(B)]) for st in range(steps)]
    d_imgs = [DeviceArray(ctx, (B, c, H, W), np.float64) for _ in range(steps)]
    d_recs = [DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64) for _ in range(steps)]
    d_outs = [DeviceArray(ctx, (B, pl.slot_stride), np.uint8) for _ in range(steps)]
    d_nbits, d_maxn = [DeviceArray(ctx, (B,), np.uint64) for _ in range(steps)], [DeviceArray(ctx, (B,), np.uint8) for _ in range(steps)]
    for st in range(steps):
        d_imgs[st].upload(imgs[st])
    ctx.synchronize()
    plain = []
    for st in range(steps):
        pl.submit(d_imgs[st].ptr, d_outs[st].ptr, d_nbits[st].ptr, d_maxn[st].ptr, d_recs[st].ptr)
        assert (ctx.get_option("idwt_groups"), ctx.get_option("pads_persist")) == (2, 0)
        # between the steps the caller codes RGB on the same context, and uses a colour block of its own
        plain.append(codec_rgb.encode(imgs[st][:1])[0])
        with color_models.fused(ctx, "IPT"):
            pass
    pl.synchronize()
    for st in range(steps):
        ref_bytes, ref_n, _g = oracle.encode_image(imgs[st][0], "bior2.2", "reflect", level, 50.0, None, mb)
        assert plain[st].encoded_bytes == ref_bytes and plain[st].max_n == ref_n, st
        res = codec_ipt.encode(imgs[st])  # the fused calls with the colour model: what the pipeline must have produced
        nb, out, mn = d_nbits[st].download(), d_outs[st].download(), d_maxn[st].download()
        for b in range(B):
            assert int(mn[b]) == res[b].max_n and out[b, :(int(nb[b]) + 7) // 8].tobytes() == res[b].encoded_bytes, (st, b)
        assert np.array_equal(d_recs[st].download(), np.stack(codec_ipt.decode(res))), st
    dec = codec_rgb.decode([plain[0]])[0]
    ref = oracle.decode_image(plain[0].encoded_bytes, plain[0].max_n, c, H, W, "bior2.2", level, 50.0, None)
    assert np.array_equal(dec, ref)
    pl.close()
    assert (ctx.get_option("idwt_groups"), ctx.get_option("pads_persist")) == (2, 0)
    ctx.close()


@pytest.mark.gpu
def test_c_caller_of_the_pipeline(tmp_path):
    """INTEGRATION.md's C example for real: tests/native/pipeline_smoke.c, compiled with gcc against include/spiht_hip.h and
    linked with libspiht_hip.so, runs the pipelined round trip with no Python in the process; its per-step checksums of
    streams and decoded pictures equal those of the fused Python calls on the same pixels."""
    import subprocess
    import spiht_amd
    from spiht_amd.batch import BatchCodec
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pipeline_smoke")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "native", "pipeline_smoke.c"),
                           "-o", exe, "-L", os.path.join(root, "spiht_amd"), "-l:libspiht_hip.so", "-Wl,-rpath," + os.path.join(root, "spiht_amd")])
    B, c, H, W, level, mb, steps = 3, 3, 96, 136, 3, 9000, 4
    p = subprocess.run([exe] + [str(v) for v in (B, c, H, W, level, mb, steps)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln.split() for ln in p.stdout.splitlines() if ln.startswith("step")]
    assert len(lines) == steps

    def fnv(b, h=1469598103934665603):
        for v in bytes(b):
            h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    codec = BatchCodec(c, H, W, spiht_amd.SpihtSettings(), level, mb)
    for s in range(steps):
        t = np.arange(B * c * H * W, dtype=np.uint64)
        x, y, k = t % W, (t // W) % H, t // (W * H)
        hsh = ((t + 1) * np.uint64(2654435761) + np.uint64(s * 40503)) & np.uint64(0xFFFFFFFF)
        v = (x * 3 + y * 5 + k * 17 + np.uint64(s * 29)) % 200 + (hsh >> np.uint64(28))
        imgs = (v.astype(np.float64) / 255.0).reshape(B, c, H, W)
        res = codec.encode(imgs)
        hs = 1469598103934665603
        for r in res:
            hs = ((hs ^ fnv(r.encoded_bytes)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        dec = np.stack(codec.decode(res))
        ln = lines[s]
        assert int(ln[3], 16) == hs, (s, ln)
        assert int(ln[5]) == min(mb, 8 * len(res[0].encoded_bytes)) or (int(ln[5]) + 7) // 8 == len(res[0].encoded_bytes)
        assert int(ln[7]) == res[0].max_n
        assert int(ln[9], 16) == fnv(np.ascontiguousarray(dec).tobytes()), s
