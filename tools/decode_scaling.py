#!/usr/bin/env python3
"""Diagnostic: list-decoder and list-encoder kernel time against the number of images in the launch (one workgroup per image)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings
from bench import synth_image, H, W, C_IMG, LEVEL, BPP

ctx = _lib.default_context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
BMAX = 1024
base = [synth_image(1000 + i, C_IMG, H, W) for i in range(8)]
d_img = DeviceArray(ctx, (BMAX, C_IMG, H, W), np.float64)
for b in range(BMAX):
    d_img.upload(base[b % 8], offset_bytes=b * C_IMG * H * W * 8)
d_out = DeviceArray(ctx, (BMAX, codec.slot_stride), np.uint8)
d_nbits = DeviceArray(ctx, (BMAX,), np.uint64)
d_nbytes = DeviceArray(ctx, (BMAX,), np.uint64)
d_maxn = DeviceArray(ctx, (BMAX,), np.uint8)
d_rec = DeviceArray(ctx, (BMAX, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
for B in (1, 32, 64, 128, 256, 512, 1024):
    for rep in range(2):
        ctx.synchronize(); ctx.reset_timing(); ctx.set_timing(True)
        codec.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
        codec.nbits_to_nbytes(d_nbits.ptr, B, d_nbytes.ptr)
        codec.decode_device(d_out.ptr, d_nbytes.ptr, d_maxn.ptr, B, d_rec.ptr)
        ctx.synchronize(); ctx.set_timing(False)
        t = ctx.timing()
    print("B=%5d  decode_lists %7.2f ms (%6.1f us/image)   encode_lists %6.2f ms (%6.1f us/image)"
          % (B, t["decode_lists"][0], 1e3 * t["decode_lists"][0] / B, t["encode_lists"][0], 1e3 * t["encode_lists"][0] / B))
