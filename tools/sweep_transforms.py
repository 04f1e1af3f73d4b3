"""Random sweep of the transforms on the GPU against the CPU oracle (test infrastructure, like tests/): wavelet (all 106),
extension mode (all nine), picture size on either side of the filter length, level, float64 / float32, channel scales.
For every case the int32 array of spiht_dwt_quant_batch_* and the picture of spiht_dequant_idwt_batch_f64 (from a
thinned-out copy of that array) must equal the oracle's in every bit.

    python tools/sweep_transforms.py [cases=1000] [seed=0] [long|short|all]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
from spiht_amd import _lib  # noqa: E402
from test_gpu_dwt import _gpu_dwt, _gpu_dwt_f32, _gpu_idwt  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    L = _lib.lib()
    names = []
    i = 0
    while L.spiht_wavelet_taps(i) > 0:
        i += 1
    for nm in oracle.WAVELET_NAMES if hasattr(oracle, "WAVELET_NAMES") else []:
        names.append(nm)
    if not names:
        import re
        txt = open(os.path.join(ROOT, "spiht_amd", "csrc", "wavelets.h")).read()
        names = re.findall(r'^    \{"([^"]+)", \d+,', txt, re.M)
    assert len(names) == i, (len(names), i)
    taps = {nm: L.spiht_wavelet_taps(L.spiht_wavelet_id(nm.encode())) for nm in names}
    if which == "long":
        names = [nm for nm in names if taps[nm] > 20]
    elif which == "short":
        names = [nm for nm in names if taps[nm] <= 20]
    modes = list(oracle.MODES)
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n):
        wv = names[int(rng.integers(len(names)))]
        F = taps[wv]
        mode = modes[int(rng.integers(len(modes)))]
        hi = max(8, 3 * F)
        H, W = int(rng.integers(2, hi)), int(rng.integers(2, hi))
        if rng.random() < 0.15:
            H, W = int(rng.integers(100, 400)), int(rng.integers(100, 500))
        c, B = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        lv = int(rng.integers(0, 4))
        f32 = rng.random() < 0.3
        if f32:
            lv = max(lv, 1)  # (no transform, float32 pixels: refused -- the wrapper quantises those in float64 itself)
        mults = None if (f32 or rng.random() < 0.5) else list(rng.choice([1.0, 0.5, 2.0, 3.0], size=c))
        q = float(rng.choice([50.0, 255.0, 10.0, 1.0]))
        img = rng.random((B, c, H, W))
        tag = (k, wv, F, mode, (B, c, H, W), lv, "f32" if f32 else "f64", mults, q)
        try:
            if f32:
                img = img.astype(np.float32)
                got = _gpu_dwt_f32(img, wv, mode, lv, q)
                ref = [oracle.quantize_f32(oracle.wavedec2_array_f32(img[b], wv, mode, lv)[0], q) for b in range(B)]
            else:
                got = _gpu_dwt(img, wv, mode, lv, q, mults)
                ref = [oracle.quantize(oracle.wavedec2_array(img[b], wv, mode, lv)[0], q, mults) for b in range(B)]
            ok = all(np.array_equal(got[b], ref[b]) for b in range(B))
            if ok and not f32:
                rec = (got - (got % 4) * (rng.random(got.shape) < 0.5)).astype(np.int32)
                back = _gpu_idwt(rec, H, W, wv, mode, lv, q, mults)
                for b in range(B):
                    r = oracle.waverec2_array(oracle.dequantize(rec[b], q, mults), H, W, wv, lv, mode)
                    ok = ok and back[b].shape == r.shape and np.array_equal(back[b].view(np.uint64), r.view(np.uint64))
        except Exception as e:  # a refusal on one side only is a finding too
            ok = False
            tag = tag + (repr(e),)
        if not ok:
            bad += 1
            print("MISMATCH", tag, flush=True)
        if (k + 1) % 200 == 0:
            print("%d cases, %d mismatches" % (k + 1, bad), flush=True)
    print("done: %d cases, %d mismatches" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
