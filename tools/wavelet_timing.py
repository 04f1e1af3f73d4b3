#!/usr/bin/env python3
"""What a wavelet / extension mode costs: 16 1080p RGB pictures, pixels in HBM, wall time of a synchronised
encode (transform + list coding at 0.5 bpp) and decode per picture -- the tiled level kernels (filters up to 20 taps, the
five index-map modes) against the plain two-pass levels (the computed modes, periodization, filters of 22 to 102 taps)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_image
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings

B, C_IMG, H, W = 16, 3, 1080, 1920
ctx = _lib.default_context(0)
base = [synth_image(1000 + i, C_IMG, H, W) for i in range(4)]
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
for b in range(B):
    d_img.upload(base[b % 4], offset_bytes=b * C_IMG * H * W * 8)
for wv, mode in (("bior2.2", "reflect"), ("bior4.4", "symmetric"), ("db10", "reflect"), ("bior2.2", "smooth"), ("bior4.4", "periodization"),
                 ("db11", "reflect"), ("sym20", "reflect"), ("dmey", "symmetric"), ("coif17", "reflect")):
    s = SpihtSettings(wavelet=wv, mode=mode)
    cd = BatchCodec(C_IMG, H, W, s, None, int(H * W * 0.5), ctx=ctx)
    g = cd.geom
    d_out = DeviceArray(ctx, (B, cd.slot_stride), np.uint8)
    d_nbits, d_maxn, d_ny = DeviceArray(ctx, (B,), np.uint64), DeviceArray(ctx, (B,), np.uint8), DeviceArray(ctx, (B,), np.uint64)
    d_rec = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    te, td = [], []
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        cd.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
        ctx.synchronize()
        t1 = time.perf_counter()
        cd.nbits_to_nbytes(d_nbits.ptr, B, d_ny.ptr)
        cd.decode_device(d_out.ptr, d_ny.ptr, d_maxn.ptr, B, d_rec.ptr)
        ctx.synchronize()
        t2 = time.perf_counter()
        te.append((t1 - t0) * 1e3 / B)
        td.append((t2 - t1) * 1e3 / B)
    taps = _lib.lib().spiht_wavelet_taps(_lib.lib().spiht_wavelet_id(wv.encode()))
    print("%-8s %3d taps  %-13s level %d   encode %7.3f ms  decode %7.3f ms per picture" % (wv, taps, mode, g["level"], min(te), min(td)), flush=True)
    for a in (d_out, d_nbits, d_maxn, d_ny, d_rec):
        a.free()
