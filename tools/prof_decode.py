#!/usr/bin/env python3
"""Diagnostic (needs a diagnostic build: tools/build_variant.sh prof -DSPIHT_DIAG -DDEC_PROF, then SPIHT_HIP_LIB=build/var_prof/spiht_amd/libspiht_hip.so): where the decoder's sequencer wavefront spends its time (needs the -DDEC_PROF build:
SPIHT_HIP_LIB=spiht_amd/libspiht_hip_prof.so python tools/prof_decode.py [batch])."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiht_amd import _lib
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings
from bench import synth_image, H, W, C_IMG, LEVEL, BPP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
# another picture than the bench's: PROF_H, PROF_W, PROF_LEVEL, PROF_WAVELET, PROF_BPP in the environment
H, W, LEVEL = int(os.environ.get("PROF_H", H)), int(os.environ.get("PROF_W", W)), int(os.environ.get("PROF_LEVEL", LEVEL))
BPP = float(os.environ.get("PROF_BPP", BPP))
ctx = _lib.default_context(0)
L = _lib.lib()
codec = BatchCodec(C_IMG, H, W, SpihtSettings(wavelet=os.environ.get("PROF_WAVELET", "bior2.2")), LEVEL, int(H * W * BPP), ctx=ctx)
g = codec.geom
img = synth_image(1000, C_IMG, H, W)
d_img = DeviceArray(ctx, (B, C_IMG, H, W), np.float64)
for b in range(B):
    d_img.upload(img, offset_bytes=b * C_IMG * H * W * 8)
d_out = DeviceArray(ctx, (B, codec.slot_stride), np.uint8)
d_nbits = DeviceArray(ctx, (B,), np.uint64)
d_nbytes = DeviceArray(ctx, (B,), np.uint64)
d_maxn = DeviceArray(ctx, (B,), np.uint8)
d_rec = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.int32)
codec.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)
codec.nbits_to_nbytes(d_nbits.ptr, B, d_nbytes.ptr)
vp = C.c_void_p
# run the decoder while a second stream saturates HBM: "hog" with forward transforms, "hogi" with inverse transforms
HOG = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] in ("hog", "hogi") else ""
if HOG:
    ctx2 = _lib.Context(0)
    d_coef = DeviceArray(ctx, (B, C_IMG, g["enc_h"], g["enc_w"]), np.int32)
    d_pix = DeviceArray(ctx, (B, C_IMG, g["rec_h"], g["rec_w"]), np.float64)
    wid, mid = L.spiht_wavelet_id(os.environ.get("PROF_WAVELET", "bior2.2").encode()), L.spiht_mode_id(b"reflect")
    ctx2.memset(d_coef.ptr, 0, d_coef.nbytes)
    ctx2.synchronize()
ctx.set_timing(True)
for it in range(2):
    if it == 1:
        ctx.reset_timing()
    if HOG and it == 1:
        for _ in range(4):
            if HOG == "hog":
                _lib.check(L.spiht_dwt_quant_batch_f64(ctx2.handle, vp(d_img.ptr), B, C_IMG, H, W, wid, mid, LEVEL, 50.0, None,
                                                       vp(d_coef.ptr)))
            else:
                _lib.check(L.spiht_dequant_idwt_batch_f64(ctx2.handle, vp(d_coef.ptr), B, C_IMG, H, W, wid, mid, LEVEL, 50.0, None,
                                                          vp(d_pix.ptr)))
    _lib.check(L.spiht_decode_batch_i32(ctx.handle, vp(d_out.ptr), codec.slot_stride, vp(d_nbytes.ptr), vp(d_maxn.ptr), B,
                                        C_IMG, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"], vp(d_rec.ptr)))
ctx.synchronize()
if HOG:
    ctx2.synchronize()
ctx.set_timing(False)
print("decoder kernel (HIP events): %.2f ms for %d images" % (ctx.timing()["decode_lists"][0], B))
words = (C.c_uint32 * 64)()
L.spiht_debug_words.argtypes = [vp, vp]
_lib.check(L.spiht_debug_words(ctx.handle, words))
w = list(words)[16:40]
names = ["lip_seq", "lis_blocks", "lis_seq", "lis_wait", "refine", "table_windows", "items_lis", "other", "generations", "scatter", "blocks", "lis_publish", "lis_window", "lis_hops",
         "run_loop", "run_breaks_table_late", "ring_wait", "run_windows", "table_tried", "table_late", "table_past_type_A", "first_entry_not_A", "window_not_whole", "-"]
tot = sum(w[k] for k in (0, 1, 2, 3, 4, 7, 9))
for k, nm in enumerate(names):
    if k in (5, 6, 8, 10, 15) or k >= 17:
        print("%-12s %d" % (nm, w[k]))
    else:
        print("%-12s %8.3f Mcycles (%.1f%%)" % (nm, w[k] * 1024 / 1e6, 100.0 * w[k] / max(tot, 1)))
ww = list(words)[40:45]
print("worker 0: %d windows; waiting for a window %.2f Mcycles, for memory (its own loads and stores in flight) %.2f, for the chain of list lengths %.2f, in windows in all %.2f (%.0f cycles a window)"
      % (ww[4], ww[0] * 1024 / 1e6, ww[1] * 1024 / 1e6, ww[2] * 1024 / 1e6, ww[3] * 1024 / 1e6, ww[3] * 1024 / max(1, ww[4])))
print("total %.3f Mcycles of the sequencer wavefront (s_memtime: shader clocks, 2.1-2.4 GHz => about %.2f ms)" % (tot * 1024 / 1e6, tot * 1024 / 2.4e9 * 1e3))

if B > 1:
    ext = (C.c_uint32 * 2048)()
    L.spiht_debug_words_ext.argtypes = [vp, vp]
    _lib.check(L.spiht_debug_words_ext(ctx.handle, ext))
    r = np.array(list(ext)[64:64 + 4 * min(B, 448)], dtype=np.int64).reshape(-1, 4)
    cu = (r[:, 1] & 15) * 64 + ((r[:, 0] >> 13) & 7) * 16 + ((r[:, 0] >> 12) & 1) * 8 + ((r[:, 0] >> 8) & 15)   # (xcc, se, sh, cu): an id, not a count
    t0 = (r[:, 2] - r[:, 2].min()) % (1 << 32)
    dur = (r[:, 3] - r[:, 2]) % (1 << 32)
    ids, cnt = np.unique(cu, return_counts=True)
    print("workgroups: %d on %d different CUs (%d CUs with 2, %d with 3 and more)" % (len(cu), len(ids), int((cnt == 2).sum()), int((cnt >= 3).sum())))
    print("start after the first [us]: median %.1f, 90 %% %.1f, max %.1f" % (np.median(t0) / 100, np.percentile(t0, 90) / 100, t0.max() / 100))
    print("duration [ms]: min %.2f, median %.2f, 90 %% %.2f, max %.2f; end of the last after the first start %.2f" %
          (dur.min() / 1e5, np.median(dur) / 1e5, np.percentile(dur, 90) / 1e5, dur.max() / 1e5, (t0 + dur).max() / 1e5))
    shared = np.isin(cu, ids[cnt >= 2])
    if shared.any() and (~shared).any():
        print("duration [ms], median: alone on its CU %.2f, sharing it %.2f" % (np.median(dur[~shared]) / 1e5, np.median(dur[shared]) / 1e5))
