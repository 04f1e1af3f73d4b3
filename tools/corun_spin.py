#!/usr/bin/env python3
"""Diagnostic (needs a diagnostic build: tools/build_variant.sh diag -DSPIHT_DIAG, then SPIHT_HIP_LIB=build/var_diag/spiht_amd/libspiht_hip.so): does the forward / inverse DWT slow down merely because another kernel's workgroups are RESIDENT on the
CUs (no memory traffic at all)?  python tools/corun_spin.py [batch]"""
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
L.spiht_debug_spin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_uint32]
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


def timed_once(fn, spin):
    ctx.synchronize(); ctx2.synchronize()
    ctx2.reset_timing(); ctx2.set_timing(True)
    if spin:
        blocks, threads, lds = spin
        _lib.check(L.spiht_debug_spin(ctx.handle, blocks, threads, 40_000_000, lds))  # outlasts the transform
    fn()
    ctx2.synchronize()
    ctx2.set_timing(False)
    t = sum(ms for k, (ms, c) in ctx2.timing().items() if c)
    ctx.synchronize()
    return t


def timed(fn, spin, reps=7):
    ts = sorted(timed_once(fn, spin) for _ in range(reps))
    return ts[0], ts[len(ts) // 2]


dwt(); idwt(); ctx2.synchronize()
CONFIGS = ((256, 512, 0), (256, 512, 8192), (256, 512, 16832), (256, 512, 24784), (256, 512, 40000), (256, 64, 0), (256, 64, 16832),
           (512, 512, 16832))
for name, fn in (("dwt", dwt), ("idwt", idwt)):
    bmin, bmed = timed(fn, None)
    print("%-5s alone: min %.2f  median %.2f ms" % (name, bmin, bmed))
    for spin in CONFIGS:
        tmin, tmed = timed(fn, spin)
        print("   with %4d idle workgroups of %4d threads, %5d B LDS: min %.2f (x%.2f)  median %.2f (x%.2f)"
              % (spin[0], spin[1], spin[2], tmin, tmin / bmin, tmed, tmed / bmed))
