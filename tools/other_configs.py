#!/usr/bin/env python3
"""Timings of the BASELINE configurations that are not the bench line: cfg3 (256 x 1024x1024 RGB, IPT,
per-channel scales, 0.1 bpp) and cfg5 (one 4096x4096 RGB image, bior6.8 level 9, bpp sweep), pixels resident in HBM,
wall time around a synchronised round trip; beside each, the CPU oracle on the same inputs (one core)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_image
from spiht_amd import _lib, color_models
from spiht_amd.batch import BatchCodec, DeviceArray
from spiht_amd.spiht_wrapper import SpihtSettings


def run(name, B, c, H, W, settings, level, max_bits, distinct, reps=5, cpu_images=0):
    ctx = _lib.default_context(0)
    cd = BatchCodec(c, H, W, settings, level, max_bits, ctx=ctx)
    g = cd.geom
    d_img = DeviceArray(ctx, (B, c, H, W), np.float64)
    base = [synth_image(1000 + i, c, H, W) for i in range(distinct)]
    for b in range(B):
        d_img.upload(base[b % distinct], offset_bytes=b * c * H * W * 8)
    d_out = DeviceArray(ctx, (B, cd.slot_stride), np.uint8)
    d_nbits, d_maxn, d_ny = DeviceArray(ctx, (B,), np.uint64), DeviceArray(ctx, (B,), np.uint8), DeviceArray(ctx, (B,), np.uint64)
    d_rec = DeviceArray(ctx, (B, c, g["rec_h"], g["rec_w"]), np.float64)
    te, td = [], []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        cd.encode_device(d_img.ptr, B, d_out.ptr, d_nbits.ptr, d_maxn.ptr)  # colour model change fused into level 1
        ctx.synchronize()
        t1 = time.perf_counter()
        cd.nbits_to_nbytes(d_nbits.ptr, B, d_ny.ptr)
        cd.decode_device(d_out.ptr, d_ny.ptr, d_maxn.ptr, B, d_rec.ptr)
        ctx.synchronize()
        t2 = time.perf_counter()
        te.append((t1 - t0) * 1e3)
        td.append((t2 - t1) * 1e3)
    e, d = sorted(te)[len(te) // 2], sorted(td)[len(td) // 2]
    nb = d_nbits.download()
    print("%-44s encode %8.2f ms  decode %8.2f ms  -> %8.1f Mpixels/s   (bits %d, max_n %d)"
          % (name, e, d, B * H * W / ((e + d) * 1e-3) / 1e6, int(nb[0]), int(d_maxn.download()[0])))
    for a in (d_img, d_out, d_nbits, d_maxn, d_ny, d_rec):
        a.free()
    if cpu_images:  # the CPU oracle on the same inputs, one core (the colour model change by the host implementation)
        from oracle import oracle as O
        m = settings.per_channel_quant_scales
        tce = tcd = 0.0
        for i in range(cpu_images):
            img = base[i % distinct]
            t0 = time.perf_counter()
            if settings.color_model not in (None, "RGB"):
                img = color_models.convert(img, "RGB", settings.color_model)
            data, mn, _ = O.encode_image(img, settings.wavelet, settings.mode, level, settings.quantization_scale, m, max_bits)
            t1 = time.perf_counter()
            r = O.decode_image(data, mn, c, H, W, settings.wavelet, level, settings.quantization_scale, m)
            if settings.color_model not in (None, "RGB"):
                r = color_models.convert(r, settings.color_model, "RGB")
            t2 = time.perf_counter()
            tce += t1 - t0
            tcd += t2 - t1
        print("%-44s CPU oracle, 1 core, %d image(s): encode %.2f s  decode %.2f s per image -> %.2f Mpixels/s"
              % ("", cpu_images, tce / cpu_images, tcd / cpu_images, cpu_images * H * W / (tce + tcd) / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "single":  # the two single-image configurations alone (for a kernel trace)
        run("cfg2  1 x 1920x1080 RGB, bior2.2 L7, 0.5 bpp", 1, 3, 1080, 1920, SpihtSettings(), 7, int(1080 * 1920 * 0.5), 1)
        run("cfg5  1 x 4096x4096 RGB, bior6.8 L9, 1.000 bpp", 1, 3, 4096, 4096, SpihtSettings(wavelet="bior6.8"), 9, 4096 * 4096, 1, reps=3)
        sys.exit(0)
    s3 = SpihtSettings(quantization_scale=1.0, color_model="IPT", per_channel_quant_scales=[50.0, 15.0, 15.0])
    run("cfg3  256 x 1024x1024 RGB, IPT, 0.1 bpp", 256, 3, 1024, 1024, s3, None, int(1024 * 1024 * 0.1), 8, cpu_images=4)
    s5 = SpihtSettings(wavelet="bior6.8")
    for bpp in (0.075, 0.1, 0.5, 1.0):
        run("cfg5  1 x 4096x4096 RGB, bior6.8 L9, %.3f bpp" % bpp, 1, 3, 4096, 4096, s5, 9, int(4096 * 4096 * bpp), 1, reps=3,
            cpu_images=1)
    run("cfg2  1 x 1920x1080 RGB, bior2.2 L7, 0.5 bpp", 1, 3, 1080, 1920, SpihtSettings(), 7, int(1080 * 1920 * 0.5), 1, cpu_images=2)
