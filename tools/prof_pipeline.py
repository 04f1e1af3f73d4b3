#!/usr/bin/env python3
"""Diagnostic (needs the diagnostic build: tools/build_variant.sh prof -DSPIHT_DIAG -DDEC_PROF, then
SPIHT_HIP_LIB=build/var_prof/spiht_amd/libspiht_hip.so python tools/prof_pipeline.py [steps]): the list decoder INSIDE the pipelined
schedule bench.py times -- where the sequencer wavefront of image 0 spends its time there, and when and how long every one of the
256 decoder workgroups of the last step ran (tools/prof_decode.py does the same for the decoder alone or beside one other kernel)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray, Pipeline
from spiht_amd.spiht_wrapper import SpihtSettings
from bench import synth_image, H, W, C_IMG, LEVEL, BPP

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, DISTINCT = 256, 16
ctx = _lib.default_context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
imgs = [synth_image(1000 + k, C_IMG, H, W) for k in range(DISTINCT)]
for b in range(B):
    d_img.upload(imgs[b % DISTINCT], offset_bytes=b * C_IMG * H * W * 8)
d_out = DeviceArray(ctx, (B, codec.slot_stride), np.uint8)
d_nbits = DeviceArray(ctx, (B,), np.uint64)
d_maxn = DeviceArray(ctx, (B,), np.uint8)
d_rec = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
pipe = Pipeline(codec, B)
for _ in range(STEPS):
    pipe.submit(d_img.ptr, d_out.ptr, d_nbits.ptr, d_maxn.ptr, d_rec.ptr)
pipe.flush()
pipe.synchronize()
vp = C.c_void_p
names = ["lip_seq", "lis_blocks", "lis_seq", "lis_wait", "refine", "table_windows", "items_lis", "other", "generations", "scatter", "blocks", "lis_publish",
         "lis_window", "lis_hops", "run_loop", "run_breaks_table_late", "ring_wait", "run_windows", "table_tried", "table_late", "table_past_type_A",
         "first_entry_not_A", "window_not_whole", "-"]
L.spiht_debug_words_ext.argtypes = [vp, vp]
for which, c in zip(("L0", "L1"), pipe.contexts()[1:]):
    ext = (C.c_uint32 * 2048)()
    _lib.check(L.spiht_debug_words_ext(c.handle, ext))
    w = list(ext)[16:40]
    tot = sum(w[k] for k in (0, 1, 2, 3, 4, 7, 9))
    print("== list-coding context %s, its last decode; sequencer of image 0: %.2f Mcycles" % (which, tot * 1024 / 1e6))
    print("   " + ", ".join("%s %.2f" % (names[k], w[k] * 1024 / 1e6) for k in (0, 1, 2, 3, 4, 9, 11, 12, 13, 14, 16)))
    ww = list(ext)[40:45]
    print("   worker 0: %d windows; waiting for a window %.2f Mcycles, for memory %.2f, for the chain %.2f, in windows in all %.2f (%.0f cycles a window)"
          % (ww[4], ww[0] * 1024 / 1e6, ww[1] * 1024 / 1e6, ww[2] * 1024 / 1e6, ww[3] * 1024 / 1e6, ww[3] * 1024 / max(1, ww[4])))
    r = np.array(list(ext)[64:64 + 4 * B], dtype=np.int64).reshape(-1, 4)
    cu = (r[:, 1] & 15) * 64 + ((r[:, 0] >> 13) & 7) * 16 + ((r[:, 0] >> 12) & 1) * 8 + ((r[:, 0] >> 8) & 15)
    t0 = (r[:, 2] - r[:, 2].min()) % (1 << 32)
    dur = (r[:, 3] - r[:, 2]) % (1 << 32)
    ids, cnt = np.unique(cu, return_counts=True)
    print("   workgroups: %d on %d different CUs (%d CUs with 2, %d with 3 and more)" % (len(cu), len(ids), int((cnt == 2).sum()), int((cnt >= 3).sum())))
    print("   start after the first [us]: median %.1f, 90 %% %.1f, max %.1f" % (np.median(t0) / 100, np.percentile(t0, 90) / 100, t0.max() / 100))
    print("   duration [ms]: min %.2f, median %.2f, 90 %% %.2f, max %.2f; end of the last after the first start %.2f" %
          (dur.min() / 1e5, np.median(dur) / 1e5, np.percentile(dur, 90) / 1e5, dur.max() / 1e5, (t0 + dur).max() / 1e5))
    shared = np.isin(cu, ids[cnt >= 2])
    if shared.any() and (~shared).any():
        print("   duration [ms], median: alone on its CU %.2f, sharing it %.2f" % (np.median(dur[~shared]) / 1e5, np.median(dur[shared]) / 1e5))
    # the slowest workgroups: which images (seed 1000 + b % 16)
    order = np.argsort(dur)[::-1][:8]
    print("   slowest: " + ", ".join("b=%d (image %d) %.2f ms" % (b, b % DISTINCT, dur[b] / 1e5) for b in order))
pipe.close()
