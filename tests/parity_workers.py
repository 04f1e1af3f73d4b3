"""Worker-process helpers of the full-size parity tests (spawned with multiprocessing, so they live in an importable
module): image synthesis and CPU-oracle round trips, many images at a time across the host's cores."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def digest(a):
    from bench import digest as d
    return d(a)


def synth_u8_job(job):
    """(seed, c, H, W) -> uint8 [c,H,W]; the pixels are this / 255 (conftest.synth_image)"""
    from bench import synth_u8
    return synth_u8(*job)


def oracle_roundtrip_job(job):
    """(pixels float64 [c,H,W] or (seed,c,H,W), wavelet, mode, level, q, mults, max_bits)
    -> (stream bytes, max_n, digest of the decoded image)"""
    from oracle import oracle as O
    img, wavelet, mode, level, q, mults, mb = job
    if isinstance(img, tuple):
        from bench import synth_u8
        img = synth_u8(*img) / 255
    c, H, W = img.shape
    data, mn, _ = O.encode_image(img, wavelet, mode, level, q, mults, mb)
    rec = O.decode_image(data, mn, c, H, W, wavelet, level, q, mults)
    return data, mn, digest(rec)


def pool(n=None):
    import concurrent.futures as cf
    import multiprocessing as mp
    n = n or max(1, min(16, os.cpu_count() or 1))
    return cf.ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
