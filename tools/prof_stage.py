#!/usr/bin/env python3
"""Run one stage of the hot path in a loop (for rocprofv3 --pmc / --kernel-trace runs).
usage: prof_stage.py {dwt|idwt|idwtf|pyramid|encode|decode|all} [batch] [iters]
idwtf: the inverse transform reading the decoder's level-1 occupancy words (the default of the image-level calls); PROF_BPP
in the environment: bits per pixel of the streams (default: the bench's)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings
from bench import synth_image, H, W, C_IMG, LEVEL, BPP
BPP = float(os.environ.get("PROF_BPP", BPP))

stage = sys.argv[1] if len(sys.argv) > 1 else "all"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = _lib.default_context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
base = [synth_image(1000 + i, C_IMG, H, W) for i in range(4)]
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
for b in range(B):
    d_img.upload(base[b % 4], offset_bytes=b * C_IMG * H * W * 8)
d_co = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.int32)
d_rec = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.int32)
d_out = DeviceArray(ctx, (B, codec.slot_stride), np.uint8)
d_nbits = DeviceArray(ctx, (B,), np.uint64)
d_nbytes = DeviceArray(ctx, (B,), np.uint64)
d_maxn = DeviceArray(ctx, (B,), np.uint8)
d_img2 = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
d_dm = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.uint8)
d_lm = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.uint8)
d_mx = DeviceArray(ctx, (B,), np.uint32)
vp = C.c_void_p
# prepare inputs of every stage once
codec.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr, d_co.ptr)
codec.nbits_to_nbytes(d_nbits.ptr, B, d_nbytes.ptr)
_lib.check(L.spiht_decode_batch_i32(ctx.handle, vp(d_out.ptr), codec.slot_stride, vp(d_nbytes.ptr), vp(d_maxn.ptr), B, C_IMG,
                                    g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], vp(d_rec.ptr)))
ctx.synchronize()
d_fl = None
if stage == "idwtf":
    nw = C.c_uint64()
    _lib.check(L.spiht_l1_flags_words(C_IMG, H, W, codec.wid, codec.mid, LEVEL, C.byref(nw)))
    d_fl = DeviceArray(ctx, (B, nw.value), np.uint32)
    ctx.memset(d_rec.ptr, 0, d_rec.nbytes)
    _lib.check(L.spiht_decode_lists_flags_batch_i32(ctx.handle, vp(d_out.ptr), codec.slot_stride, vp(d_nbytes.ptr), vp(d_maxn.ptr), B,
                                                    C_IMG, H, W, codec.wid, codec.mid, LEVEL, vp(d_rec.ptr), vp(d_fl.ptr)))
    ctx.synchronize()
    print("occupied_tiles_fraction %.5f" % float(d_fl.download().mean()))
for _ in range(iters):
    if stage in ("dwt", "all"):
        _lib.check(L.spiht_dwt_quant_batch_f64(ctx.handle, vp(d_img.ptr), B, C_IMG, H, W, codec.wid, codec.mid, LEVEL, 50.0, None, vp(d_co.ptr)))
    if stage in ("pyramid", "all"):
        _lib.check(L.spiht_pyramid_batch_i32(ctx.handle, vp(d_co.ptr), B, C_IMG, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], vp(d_dm.ptr), vp(d_lm.ptr), vp(d_mx.ptr)))
    if stage in ("encode", "all"):
        _lib.check(L.spiht_encode_batch_i32(ctx.handle, vp(d_co.ptr), B, C_IMG, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], int(H * W * BPP), vp(d_out.ptr), codec.slot_stride, vp(d_nbits.ptr), vp(d_maxn.ptr)))
    if stage in ("decode", "all"):
        _lib.check(L.spiht_decode_batch_i32(ctx.handle, vp(d_out.ptr), codec.slot_stride, vp(d_nbytes.ptr), vp(d_maxn.ptr), B, C_IMG, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], vp(d_rec.ptr)))
    if stage == "idwtf":
        _lib.check(L.spiht_dequant_idwt_flags_batch_f64(ctx.handle, vp(d_rec.ptr), vp(d_fl.ptr), B, C_IMG, H, W, codec.wid, codec.mid, LEVEL, 50.0, None, vp(d_img2.ptr)))
    if stage in ("idwt", "all"):
        _lib.check(L.spiht_dequant_idwt_batch_f64(ctx.handle, vp(d_rec.ptr), B, C_IMG, H, W, codec.wid, codec.mid, LEVEL, 50.0, None, vp(d_img2.ptr)))
ctx.synchronize()
print("done", stage, B, iters)
