#!/usr/bin/env python3
"""Diagnostic (needs a diagnostic build: tools/build_variant.sh diag -DSPIHT_DIAG, then
SPIHT_HIP_LIB=build/var_diag/spiht_amd/libspiht_hip.so): WHICH doing of a list-coding workgroup costs the transform
kernels beside it?  256 synthetic neighbours (one per CU, 512 threads, the 8-wavefront decoder's 17 KB of LDS), each kind
doing one thing (pyramid.hip: k_spin_kind), beside the forward level 1, the coarse forward levels, and the inverse
transform's two parts.  python tools/corun_kinds.py [batch]"""
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
L.spiht_debug_spin_kind.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int]
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
n = C_IMG * g["enc_h"] * g["enc_w"]
img = synth_image(1000, C_IMG, H, W)
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
for b in range(B):
    d_img.upload(img, offset_bytes=b * C_IMG * H * W * 8)
d_coef = DeviceArray(ctx, (B, n), np.int32)
d_img2 = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
vp = C.c_void_p


def dwt():
    _lib.check(L.spiht_dwt_quant_batch_f64(ctx2.handle, vp(d_img.ptr), B, C_IMG, H, W, codec.wid, codec.mid, LEVEL, 50.0, None,
                                           vp(d_coef.ptr)))


def idwt():
    _lib.check(L.spiht_dequant_idwt_batch_f64(ctx2.handle, vp(d_coef.ptr), B, C_IMG, H, W, codec.wid, codec.mid, LEVEL, 50.0, None,
                                              vp(d_img2.ptr)))


def once(fn, mode, threads, lds):
    ctx.synchronize(); ctx2.synchronize()
    ctx2.reset_timing(); ctx2.set_timing(True)
    if mode is not None:
        _lib.check(L.spiht_debug_spin_kind(ctx.handle, 256, threads, 60_000_000, lds, mode))  # 25 ms of s_memtime ticks (shader clock): outlasts it
    fn()
    ctx2.synchronize()
    ctx2.set_timing(False)
    t = {k: ms for k, (ms, c) in ctx2.timing().items() if c}
    ctx.synchronize()
    return t


def best(fn, mode, threads=512, lds=16832, reps=5):
    rs = [once(fn, mode, threads, lds) for _ in range(reps)]
    return {k: min(r[k] for r in rs) for k in rs[0]}


dwt(); idwt(); ctx2.synchronize()
KINDS = ((None, "alone"), (0, "idle (sleeping)"), (1, "wavefront 0: dependent scalar chain"), (2, "all wavefronts poll LDS"),
         (3, "wavefront 0: dependent vector chain"), (4, "scalar chain + vector/LDS wavefront"), (5, "all wavefronts scalar work"),
         (16, "idle, 96 VGPRs per thread"), (17, "scalar chain, 96 VGPRs"), (20, "scalar chain + vector/LDS, 96 VGPRs"))
for mode, name in KINDS:
    f, i = best(dwt, mode), best(idwt, mode)
    print("%-40s forward level 1 %6.2f  rest %5.2f   inverse rest %5.2f  level 1 %6.2f ms"
          % (name, f.get("dwt_level1", 0), f.get("dwt_rest", 0), i.get("idwt_rest", 0), i.get("idwt_level1", 0)), flush=True)
