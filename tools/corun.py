#!/usr/bin/env python3
"""Diagnostic: how much the list decoder and one HBM-bound pass slow each other down when they run at the same time
on two contexts.  python tools/corun.py [batch] -> one line per co-runner."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings
from bench import synth_image, H, W, C_IMG, LEVEL, BPP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _lib.default_context(0)
ctx2 = _lib.Context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
n = C_IMG * g["enc_h"] * g["enc_w"]
img = synth_image(1000, C_IMG, H, W)
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
for b in range(B):
    d_img.upload(img, offset_bytes=b * C_IMG * H * W * 8)
d_out = DeviceArray(ctx, (B, codec.slot_stride), np.uint8)
d_nbits = DeviceArray(ctx, (B,), np.uint64)
d_nbytes = DeviceArray(ctx, (B,), np.uint64)
d_maxn = DeviceArray(ctx, (B,), np.uint8)
d_rec = DeviceArray(ctx, (B, n), np.int32)
d_rec2 = DeviceArray(ctx, (B, n), np.int32)
d_coef = DeviceArray(ctx, (B, n), np.int32)
d_dm = DeviceArray(ctx, (B, n), np.uint8)
d_lm = DeviceArray(ctx, (B, n), np.uint8)
d_ma = DeviceArray(ctx, (B,), np.uint32)
d_img2 = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
vp = C.c_void_p
codec.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
codec.nbits_to_nbytes(d_nbits.ptr, B, d_nbytes.ptr)
codec.decode_device(d_out.ptr, d_nbytes.ptr, d_maxn.ptr, B, d_img2.ptr, d_rec=d_rec2.ptr)
ctx.synchronize()
wid, mid = codec.wid, codec.mid


def decode():
    ctx.memset(d_rec.ptr, 0, d_rec.nbytes)
    _lib.check(L.spiht_decode_lists_batch_i32(ctx.handle, vp(d_out.ptr), codec.slot_stride, vp(d_nbytes.ptr), vp(d_maxn.ptr), B,
                                              C_IMG, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], vp(d_rec.ptr)))


def encode_lists():
    _lib.check(L.spiht_encode_lists_batch_i32(ctx.handle, vp(d_coef.ptr), vp(d_dm.ptr), vp(d_lm.ptr), vp(d_ma.ptr), B, C_IMG,
                                              g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], codec.max_bits, vp(d_out.ptr),
                                              codec.slot_stride, vp(d_nbits.ptr), vp(d_maxn.ptr)))


HOGS = {
    "dwt": lambda: _lib.check(L.spiht_dwt_quant_batch_f64(ctx2.handle, vp(d_img.ptr), B, C_IMG, H, W, wid, mid, LEVEL, 50.0, None,
                                                        vp(d_coef.ptr))),
    "dwt+pyr": lambda: _lib.check(L.spiht_dwt_pyramid_batch_f64(ctx2.handle, vp(d_img.ptr), B, C_IMG, H, W, wid, mid, LEVEL, 50.0,
                                                               None, vp(d_coef.ptr), vp(d_dm.ptr), vp(d_lm.ptr), vp(d_ma.ptr))),
    "idwt": lambda: _lib.check(L.spiht_dequant_idwt_batch_f64(ctx2.handle, vp(d_rec2.ptr), B, C_IMG, H, W, wid, mid, LEVEL, 50.0,
                                                             None, vp(d_img2.ptr))),
    "memset": lambda: ctx2.memset(d_img2.ptr, 0, d_img2.nbytes),
}


def timed(fn_main, hog, reps):
    for cx in (ctx, ctx2):
        cx.synchronize()
        cx.reset_timing()
        cx.set_timing(True)
    if hog:
        for _ in range(reps):
            HOGS[hog]()
    fn_main()
    ctx.synchronize()
    ctx2.synchronize()
    out = {}
    for cx in (ctx, ctx2):
        cx.set_timing(False)
        for k, (ms, cnt) in cx.timing().items():
            if cnt:
                out[k] = out.get(k, 0.0) + ms
    return out


HOGS["dwt+pyr"]()  # valid pyramid for encode_lists
ctx2.synchronize()
for name, fn in (("decode", decode), ("encode_lists", encode_lists)):
    key = "decode_lists" if name == "decode" else "encode_lists"
    base = timed(fn, None, 0)[key]
    print("%-12s alone: %.2f ms" % (name, base))
    for hog in HOGS:
        alone = timed(lambda: None, hog, 1)
        t_alone = sum(v for k, v in alone.items() if k not in ("decode_lists", "encode_lists"))
        reps = max(1, int(1.3 * base / max(t_alone, 0.1)) + 1)
        r = timed(fn, hog, reps)
        t_h = sum(v for k, v in r.items() if k not in ("decode_lists", "encode_lists")) / reps
        print("  with %-8s x%d: %s %.2f ms (x%.2f);  %s %.2f -> %.2f ms per call (x%.2f)"
              % (hog, reps, name, r[key], r[key] / base, hog, t_alone, t_h, t_h / max(t_alone, 1e-9)))
