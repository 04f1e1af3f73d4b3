#!/usr/bin/env python3
"""Other batch shapes than the bench's (a sanity check that nothing in the path has a per-image or per-level cost that only
shows with many small or few large pictures): pixels in HBM, wall time of a synchronised encode and decode call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import synth_image
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings
ctx = _lib.default_context(0)
for (B, c, H, W, bpp) in ((4096, 3, 64, 64, 1.0), (1024, 3, 256, 256, 0.5), (64, 1, 2048, 2048, 0.25)):
    cd = BatchCodec(c, H, W, SpihtSettings(), None, int(H * W * bpp), ctx=ctx)
    g = cd.geom
    base = [synth_image(5 + i, c, H, W) for i in range(4)]
    d_img = DeviceArray(ctx, (B, c, H, W), np.float64)
    for b in range(B):
        d_img.upload(base[b % 4], offset_bytes=b * c * H * W * 8)
    d_out = DeviceArray(ctx, (B, cd.slot_stride), np.uint8)
    d_nbits, d_maxn, d_ny = DeviceArray(ctx, (B,), np.uint64), DeviceArray(ctx, (B,), np.uint8), DeviceArray(ctx, (B,), np.uint64)
    d_rec = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)
    te, td = [], []
    for _ in range(3):
        ctx.synchronize(); t0 = time.perf_counter()
        cd.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
        ctx.synchronize(); t1 = time.perf_counter()
        cd.nbits_to_nbytes(d_nbits.ptr, B, d_ny.ptr)
        cd.decode_device(d_out.ptr, d_ny.ptr, d_maxn.ptr, B, d_rec.ptr)
        ctx.synchronize(); t2 = time.perf_counter()
        te.append((t1 - t0) * 1e3); td.append((t2 - t1) * 1e3)
    print("%5d x %dx%dx%d at %.2f bpp: encode %8.3f ms  decode %8.3f ms  -> %8.1f Mpixels/s, %.0f images/s" % (B, c, H, W, bpp, min(te), min(td), B * H * W / ((min(te) + min(td)) * 1e-3) / 1e6, B / ((min(te) + min(td)) * 1e-3)), flush=True)
    for a in (d_img, d_out, d_nbits, d_maxn, d_ny, d_rec):
        a.free()
