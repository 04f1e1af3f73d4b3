#!/usr/bin/env python3
"""Progressive decoding of one 1080p stream to K byte prefixes (make_gif.py:46-61's pattern): coefficient arrays on the
device by (a) one walk + per-node replay (spiht_decode_budgets_dev_i32) and (b) K prefixes as K streams of one batch
(spiht_decode_batch_i32: K walks on K CUs); wall time around a synchronised call, inputs on the host as make_gif has them."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_image, H, W, C_IMG, LEVEL, BPP
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings

ctx = _lib.default_context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
res = codec.encode(synth_image(1000, C_IMG, H, W)[None])[0]
n = len(res.encoded_bytes)
vp = C.c_void_p
print("one %dx%d RGB stream of %d bytes (0.5 bpp); K prefixes of equal spacing; ms per call (best of 5)" % (W, H, n))
for K in (1, 4, 16, 64, 128):
    lens = [max(1, n * (k + 1) // K) for k in range(K)]
    d_rec = DeviceArray(ctx, (K, C_IMG, g["enc_h"], g["enc_w"]), np.int32)
    bud = np.ascontiguousarray([8 * k for k in lens], dtype=np.uint64)
    data = np.frombuffer(res.encoded_bytes, np.uint8)
    stride = (n + 3) & ~3
    batch = np.zeros((K, stride), np.uint8)
    for k, ln in enumerate(lens):
        batch[k, :ln] = data[:ln]
    d_data = DeviceArray(ctx, batch.shape, np.uint8)
    d_nb = DeviceArray(ctx, (K,), np.uint64)
    d_mn = DeviceArray(ctx, (K,), np.uint8)
    ta, tb = [], []
    for it in range(6):
        t0 = time.perf_counter()
        _lib.check(L.spiht_decode_budgets_dev_i32(ctx.handle, vp(data.ctypes.data), n, int(res.max_n), C_IMG, g["enc_h"], g["enc_w"],
                                                  g["ll_h"], g["ll_w"], vp(bud.ctypes.data), K, vp(d_rec.ptr)))
        ctx.synchronize()
        ta.append(time.perf_counter() - t0)
        a = d_rec.download() if K <= 16 else None
        t0 = time.perf_counter()
        d_data.upload(batch)
        d_nb.upload(np.asarray(lens, np.uint64))
        d_mn.upload(np.full(K, res.max_n, np.uint8))
        ctx.memset(d_rec.ptr, 0, d_rec.nbytes)
        _lib.check(L.spiht_decode_batch_i32(ctx.handle, vp(d_data.ptr), stride, vp(d_nb.ptr), vp(d_mn.ptr), K, C_IMG, g["enc_h"],
                                            g["enc_w"], g["ll_h"], g["ll_w"], vp(d_rec.ptr)))
        ctx.synchronize()
        tb.append(time.perf_counter() - t0)
        if a is not None and it == 0:
            assert np.array_equal(a, d_rec.download()), "the two ways differ"
    print("K = %3d   one walk + replay %7.2f ms     K streams in one batch %7.2f ms" % (K, 1e3 * min(ta[1:]), 1e3 * min(tb[1:])))
    for d in (d_rec, d_data, d_nb, d_mn):
        d.free()
