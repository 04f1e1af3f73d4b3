#!/usr/bin/env python3
"""Where the time of BASELINE config 3's colour model change goes: the change as a pass of its own (k_color3), the plain
transforms, and the transforms with the change fused into level 1 (stage timers of the library, HIP events)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_image
from spiht_amd import _lib, color_models
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings

B, c, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 3, 1024, 1024
ctx = _lib.default_context(0)
L = _lib.lib()
vp = C.c_void_p
base = [synth_image(1000 + i, c, H, W) for i in range(4)]
d_img = DeviceArray(ctx, (B, c, H, W), np.float64)
for b in range(B):
    d_img.upload(base[b % 4], offset_bytes=b * c * H * W * 8)
d_tmp = DeviceArray(ctx, (B, c, H, W), np.float64)
mults = [50.0, 15.0, 15.0]


def timed(fn, reps=3):
    fn()
    ctx.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        t.append((time.perf_counter() - t0) * 1e3)
    return sorted(t)[len(t) // 2]


A, M, p = color_models._params("RGB", "IPT")
Ai, Mi, pi = color_models._params("IPT", "RGB")
print("k_color3 RGB->IPT  %.2f ms" % timed(lambda: _lib.check(L.spiht_color3_batch_f64(
    ctx.handle, vp(d_img.ptr), vp(d_tmp.ptr), B, H * W, vp(A.ctypes.data), vp(M.ctypes.data), p))))
print("k_color3 IPT->RGB  %.2f ms" % timed(lambda: _lib.check(L.spiht_color3_batch_f64(
    ctx.handle, vp(d_tmp.ptr), vp(d_tmp.ptr), B, H * W, vp(Ai.ctypes.data), vp(Mi.ctypes.data), pi))))
for cm in (None, "IPT"):
    s = SpihtSettings(quantization_scale=1.0, color_model=cm, per_channel_quant_scales=mults)
    cd = BatchCodec(c, H, W, s, None, int(H * W * 0.1), ctx=ctx)
    g = cd.geom
    d_out, d_nb, d_mn, d_ny = (DeviceArray(ctx, (B, cd.slot_stride), np.uint8), DeviceArray(ctx, (B,), np.uint64),
                               DeviceArray(ctx, (B,), np.uint8), DeviceArray(ctx, (B,), np.uint64))
    d_rec = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)

    def rt():
        cd.encode_device(d_img.ptr, B, d_out.ptr, d_nb.ptr, d_mn.ptr)
        cd.nbits_to_nbytes(d_nb.ptr, B, d_ny.ptr)
        cd.decode_device(d_out.ptr, d_ny.ptr, d_mn.ptr, B, d_rec.ptr)
    rt()
    ctx.synchronize()
    ctx.reset_timing()
    ctx.set_timing(True)
    for _ in range(3):
        rt()
    ctx.synchronize()
    ctx.set_timing(False)
    print("colour model %s:" % cm, {k: round(v[0] / 3, 3) for k, v in ctx.timing().items() if v[1]})
    for a in (d_out, d_nb, d_mn, d_ny, d_rec):
        a.free()
