"""Pure-Python / numpy CPU path: counterpart of the reference's legacy twin `spiht/spiht_py.py`
(encode_image_py :102-246, decode_image_py :248-371), which BASELINE.json's configuration 1 runs
("plumbing, no GPU").

This is NOT a fallback of the GPU path: nothing in spiht_amd routes here, and `encode_image`/`decode_image` never
call it.  It exists because the reference ships such a module.  Two deliberate differences from the reference's
twin, which is bit-rotted (constructs EncodingResult with 11 positional fields) and emits a different stream from
the Rust core: this one implements the RUST semantics (encoder_decoder.rs), so its streams are
interchangeable with `spiht_amd.encode()/decode()`, and it returns packed bytes in the current 7-field
EncodingResult.  The DWT is a numpy restatement of the PyWavelets calls the reference makes.

The significance tests use a precomputed max-descendant pyramid instead of the reference's recursion
(same values, see csrc/pyramid.hip); the list loops are plain Python.
"""
import math

import numpy as np

from .spiht_wrapper import EncodingResult

_FILTERS = {
    "bior2.2": (
        [0.0, -0.1767766952966369, 0.3535533905932738, 1.0606601717798212, 0.3535533905932738, -0.1767766952966369],
        [0.0, 0.3535533905932738, -0.7071067811865476, 0.3535533905932738, 0.0, 0.0],
        [0.0, 0.3535533905932738, 0.7071067811865476, 0.3535533905932738, 0.0, 0.0],
        [0.0, 0.1767766952966369, 0.3535533905932738, -1.0606601717798212, 0.3535533905932738, 0.1767766952966369]),
    "haar": ([0.7071067811865476, 0.7071067811865476], [-0.7071067811865476, 0.7071067811865476],
             [0.7071067811865476, 0.7071067811865476], [0.7071067811865476, -0.7071067811865476]),
}


def _ext(idx, n, mode):
    idx = np.asarray(idx)
    if mode == "reflect":
        if n == 1:
            return np.zeros_like(idx)
        p = 2 * (n - 1)
        m = np.mod(idx, p)
        return np.where(m < n, m, p - m)
    if mode == "symmetric":
        p = 2 * n
        m = np.mod(idx, p)
        return np.where(m < n, m, p - 1 - m)
    if mode == "periodic":
        return np.mod(idx, n)
    raise ValueError("mode %r is not supported by the Python path" % mode)


def _dwt_axis(x, lo, hi, mode, axis):
    x = np.moveaxis(x, axis, -1)
    n, F = x.shape[-1], len(lo)
    L = (n + F - 1) // 2
    o = np.arange(L)
    a = np.zeros(x.shape[:-1] + (L,))
    d = np.zeros_like(a)
    for j in range(F):  # ascending tap index, as the oracle and the HIP kernel
        v = x[..., _ext(2 * o + 1 - j, n, mode)]
        a = a + lo[j] * v
        d = d + hi[j] * v
    return np.moveaxis(a, -1, axis), np.moveaxis(d, -1, axis)


def _idwt_axis(ca, cd, lo, hi, axis):
    ca, cd = np.moveaxis(ca, axis, -1), np.moveaxis(cd, axis, -1)
    L, F = ca.shape[-1], len(lo)
    n = 2 * L - F + 2
    out = np.zeros(ca.shape[:-1] + (n,))
    pos = np.arange(n)
    for s in range(F // 2):  # band index k = n//2 + s, ascending
        k = pos // 2 + s
        t = pos + F - 2 - 2 * k
        ok = k < L
        kk = np.minimum(k, L - 1)
        term = ca[..., kk] * np.asarray(lo)[t] + cd[..., kk] * np.asarray(hi)[t]
        out = out + np.where(ok, term, 0.0)
    return np.moveaxis(out, -1, axis)


def _max_level(n, F):
    if F <= 1 or n < F - 1:
        return 0
    return int(math.floor(math.log2(n // (F - 1)))) if n // (F - 1) >= 1 else 0


def _wavedec2_array(image, wavelet, level, mode):
    lo, hi, _, _ = _FILTERS[wavelet]
    F = len(lo)
    c, H, W = image.shape
    if level is None:
        level = min(_max_level(H, F), _max_level(W, F))
    a = np.asarray(image, dtype=np.float64)
    details = []
    for _ in range(level):
        al, ah = _dwt_axis(a, lo, hi, mode, 1)      # axis -2 first (pywt dwtn)
        aa, ad = _dwt_axis(al, lo, hi, mode, 2)
        da, dd = _dwt_axis(ah, lo, hi, mode, 2)
        details.append((ad, da, dd))
        a = aa
    ll_h, ll_w = a.shape[1:]
    eh = ll_h + sum(d[0].shape[1] for d in details)
    ew = ll_w + sum(d[0].shape[2] for d in details)
    arr = np.zeros((c, eh, ew))
    arr[:, :ll_h, :ll_w] = a
    oh, ow = ll_h, ll_w
    for ad, da, dd in reversed(details):  # pywt.coeffs_to_array: coarsest first, zero padded
        h2, w2 = ad.shape[1:]
        arr[:, :h2, ow:ow + w2] = ad
        arr[:, oh:oh + h2, :w2] = da
        arr[:, oh:oh + h2, ow:ow + w2] = dd
        oh += h2
        ow += w2
    shapes = [d[0].shape[1:] for d in reversed(details)]
    return arr, ll_h, ll_w, shapes


def _waverec2_array(arr, ll_h, ll_w, shapes, wavelet):
    _, _, lo, hi = _FILTERS[wavelet]
    a = arr[:, :ll_h, :ll_w]
    oh, ow = ll_h, ll_w
    for (h2, w2) in shapes:
        if a.shape[1] == h2 + 1:
            a = a[:, :-1]
        if a.shape[2] == w2 + 1:
            a = a[:, :, :-1]
        ad = arr[:, :h2, ow:ow + w2]
        da = arr[:, oh:oh + h2, :w2]
        dd = arr[:, oh:oh + h2, ow:ow + w2]
        tl = _idwt_axis(a, ad, lo, hi, 2)   # axis -1 first (pywt idwtn)
        th = _idwt_axis(da, dd, lo, hi, 2)
        a = _idwt_axis(tl, th, lo, hi, 1)
        oh += h2
        ow += w2
    return a


# ---- tree helpers (encoder_decoder.rs:7-12, 43-75) -----------------------------------------------------------

def _offspring(i, j, h, w, ll_h, ll_w):
    if i < ll_h and j < ll_w:
        if i % 2 == 0 and j % 2 == 0:
            return None
        r, cc = (i % 2) * ll_h + i // 2 * 2, (j % 2) * ll_w + j // 2 * 2
    else:
        if 2 * i + 1 >= h or 2 * j + 1 >= w:
            return None
        r, cc = 2 * i, 2 * j
    return ((r, cc), (r, cc + 1), (r + 1, cc), (r + 1, cc + 1))


def _l_exists(i, j, h, w):
    return not ((i * 2 + 1) * 2 + 1 >= h or (j * 2 + 1) * 2 + 1 >= w)


def _pyramid(mag, ll_h, ll_w):
    """D[k,i,j] = max magnitude over all descendants, L = over descendants past the offspring (0 if none)."""
    c, h, w = mag.shape
    D = np.zeros_like(mag)
    L = np.zeros_like(mag)
    d = 1
    rounds = 0
    while ((ll_w << (rounds + 1)) + 1 < w) or ((ll_h << (rounds + 1)) + 1 < h):
        rounds += 1
    for d in range(1, rounds + 1):
        ii, jj = np.arange((h - 1 + (1 << d) - 1) >> d), np.arange((w - 1 + (1 << d) - 1) >> d)
        I, J = np.meshgrid(ii, jj, indexing="ij")
        sel = (I * (1 << d) + 1 < h) & (J * (1 << d) + 1 < w) & ~((I * (2 << d) + 1 < h) & (J * (2 << d) + 1 < w)) & \
            ~((I < ll_h) & (J < ll_w))
        I, J = I[sel], J[sel]
        dv = np.zeros((c, I.size), mag.dtype)
        lv = np.zeros((c, I.size), mag.dtype)
        for a in (0, 1):
            for b in (0, 1):
                oi, oj = 2 * I + a, 2 * J + b
                s = mag[:, oi, oj]
                has = (2 * oi + 1 < h) & (2 * oj + 1 < w)
                dc = np.where(has[None], D[:, oi, oj], 0)
                dv = np.maximum(dv, np.maximum(s, dc))
                lv = np.maximum(lv, dc)
        D[:, I, J] = dv
        L[:, I, J] = lv
    for i in range(ll_h):
        for j in range(ll_w):
            off = _offspring(i, j, h, w, ll_h, ll_w)
            if off is None:
                continue
            dv = np.zeros(c, mag.dtype)
            lv = np.zeros(c, mag.dtype)
            for (oi, oj) in off:
                dc = D[:, oi, oj] if (2 * oi + 1 < h and 2 * oj + 1 < w) else 0
                dv = np.maximum(dv, np.maximum(mag[:, oi, oj], dc))
                lv = np.maximum(lv, dc)
            D[:, i, j] = dv
            L[:, i, j] = lv
    return D, L


class _End(Exception):
    pass


def encode_py(arr, ll_h, ll_w, max_bits):
    """encoder_decoder.rs:155-303 on an int32 array -> (bytes, max_n)"""
    arr = np.asarray(arr)
    c, h, w = arr.shape
    if not (ll_h > 1 and ll_w > 1):
        raise AssertionError("assertion failed: ll_h > 1 && ll_w > 1")
    mag = np.abs(arr.astype(np.int64))
    mx = int(mag.max())
    n = 0 if mx == 0 else int(np.uint8(max(0.0, min(255.0, float(np.log2(np.float32(mx)))))))
    max_n = n
    D, L = _pyramid(mag, ll_h, ll_w)
    out = []

    def push(b):
        out.append(1 if b else 0)
        if len(out) == max_bits:
            raise _End()

    lip = [(k, i, j) for i in range(ll_h) for j in range(ll_w) for k in range(c)]
    lis = [(True, k, i, j) for i in range(ll_h) for j in range(ll_w) if not (i % 2 == 0 and j % 2 == 0) for k in range(c)]
    lsp = []
    try:
        while True:
            T = 1 << n
            lsp_len = len(lsp)
            keep = []
            for (k, i, j) in lip:
                sig = mag[k, i, j] >= T
                push(sig)
                if sig:
                    lsp.append((k, i, j))
                    push(arr[k, i, j] >= 0)
                else:
                    keep.append((k, i, j))
            lip = keep
            retain = []
            q = 0
            while q < len(lis):  # entries appended during the pass are processed in the same pass
                t, k, i, j = lis[q]
                q += 1
                off = _offspring(i, j, h, w, ll_h, ll_w)
                if t:
                    dsig = off is not None and D[k, i, j] >= T
                    push(dsig)
                    if dsig:
                        for (l, m) in off:
                            s = mag[k, l, m] >= T
                            push(s)
                            if s:
                                lsp.append((k, l, m))
                                push(arr[k, l, m] >= 0)
                            else:
                                lip.append((k, l, m))
                        if _l_exists(i, j, h, w):
                            lis.append((False, k, i, j))
                    else:
                        retain.append((t, k, i, j))
                else:
                    lsig = L[k, i, j] >= T
                    push(lsig)
                    if lsig:
                        if off is not None:
                            lis.extend((True, k, l, m) for (l, m) in off)
                    else:
                        retain.append((t, k, i, j))
            lis = retain
            for q in range(lsp_len):
                k, i, j = lsp[q]
                push((int(mag[k, i, j]) >> n) & 1)
            if n == 0:
                break
            n -= 1
    except _End:
        pass
    bits = np.array(out, dtype=np.uint8)
    return np.packbits(bits, bitorder="little").tobytes(), max_n


def decode_py(data, n, c, h, w, ll_h, ll_w):
    """encoder_decoder.rs:307-454; all 8*len(data) bits are data (lib.rs:38)"""
    if not (ll_h > 1 and ll_w > 1):
        raise AssertionError("assertion failed: ll_h > 1 && ll_w > 1")
    bits = np.unpackbits(np.frombuffer(bytes(data), np.uint8), bitorder="little").tolist()
    rec = np.zeros((c, h, w), dtype=np.int32)
    cur = [0]

    def pop():
        if cur[0] >= len(bits):
            raise _End()
        v = bits[cur[0]]
        cur[0] += 1
        return v

    def set_bit(x, nn, b):
        if x >= 0:
            return (x | (1 << nn)) if b else (x & ~(1 << nn))
        return -((-x) | (1 << nn)) if b else -((-x) & ~(1 << nn))

    lip = [(k, i, j) for i in range(ll_h) for j in range(ll_w) for k in range(c)]
    lis = [(True, k, i, j) for i in range(ll_h) for j in range(ll_w) if not (i % 2 == 0 and j % 2 == 0) for k in range(c)]
    lsp = []
    try:
        while True:
            base = 1 if n == 0 else (1 << (n - 1)) + (1 << n)
            lsp_len = len(lsp)
            keep = []
            for (k, i, j) in lip:
                if pop():
                    lsp.append((k, i, j))
                    rec[k, i, j] = base * (pop() * 2 - 1)
                else:
                    keep.append((k, i, j))
            lip = keep
            retain = []
            q = 0
            while q < len(lis):
                t, k, i, j = lis[q]
                q += 1
                off = _offspring(i, j, h, w, ll_h, ll_w)
                if t:
                    if pop():
                        if off is not None:
                            for (l, m) in off:
                                if pop():
                                    lsp.append((k, l, m))
                                    rec[k, l, m] = (pop() * 2 - 1) * base
                                else:
                                    lip.append((k, l, m))
                        if _l_exists(i, j, h, w):
                            lis.append((False, k, i, j))
                    else:
                        retain.append((t, k, i, j))
                else:
                    if pop():
                        if off is not None:
                            lis.extend((True, k, l, m) for (l, m) in off)
                    else:
                        retain.append((t, k, i, j))
            lis = retain
            for q in range(lsp_len):
                k, i, j = lsp[q]
                rec[k, i, j] = set_bit(int(rec[k, i, j]), n, pop())
            if n == 0:
                break
            n -= 1
    except _End:
        pass
    return rec


def encode_image_py(image, wavelet='bior2.2', level=6, max_bits=None, quantization_scale=50, mode='reflect'):
    """spiht_py.py:102-246 (argument order and defaults kept).  Returns the current EncodingResult."""
    if image.ndim != 3:
        raise ValueError('image ndim must be 3: c,h,w')
    c, H, W = image.shape
    arr, ll_h, ll_w, _ = _wavedec2_array(image, wavelet, level, mode)
    q = (arr * quantization_scale).astype(np.int32)
    if max_bits is None:
        max_bits = 99999999999999999
    data, max_n = encode_py(q, ll_h, ll_w, max_bits)
    return EncodingResult(data, H, W, c, max_n, level)


def decode_image_py(encoding_result, wavelet='bior2.2', quantization_scale=50, mode='reflect'):
    """spiht_py.py:248-371.  The legacy result object carried wavelet/scale/mode; the current one does not, so
    they are arguments here."""
    r = encoding_result
    probe = np.zeros((r.c, r.h, r.w))
    arr0, ll_h, ll_w, shapes = _wavedec2_array(probe, wavelet, r.level, mode)
    rec = decode_py(r.encoded_bytes, r.max_n, r.c, arr0.shape[1], arr0.shape[2], ll_h, ll_w)
    return _waverec2_array(rec / quantization_scale, ll_h, ll_w, shapes, wavelet)
