"""The bit-plane form of the list decoder's token lengths (spiht_amd/csrc/decode.hip: helper_phase), restated in Python integers and
held against the sequential parse it replaces -- CPU only, a guard on the arithmetic (the kernel itself is held to the oracle by the
`-m gpu` tests).  A fired type-A entry at stream position q is followed by four offspring tokens, each '0' or '1' + sign bit
(encoder_decoder.rs:389-410); its length is 5 + the number of significant offspring.  For all 64 positions of a window at once:
offspring k's bit is the window shifted right by k ... 2k-1, selected by the running count of significant ones."""
import random

M = (1 << 64) - 1


def planes(lo, hi):
    """three 64-bit planes e0, e1, e2 of (length - 5) for the 64 positions of the window `lo` (next window `hi`)"""
    def X(k):
        return ((lo >> k) | (hi << (64 - k))) & M
    s1 = X(1)
    s2 = (s1 & X(3)) | (~s1 & X(2))
    a0, a1 = s1 ^ s2, s1 & s2
    s3 = (a1 & X(5)) | (~a1 & ((a0 & X(4)) | (~a0 & X(3))))
    b0, b1 = a0 ^ s3, a1 | (a0 & s3)
    s4 = (b1 & ((b0 & X(7)) | (~b0 & X(6)))) | (~b1 & ((b0 & X(5)) | (~b0 & X(4))))
    cy = b0 & s4
    return (b0 ^ s4) & M, (b1 ^ cy) & M, (b1 & cy) & M


def parse(bits128, q):
    """the sequential form: tokens from position q + 1 on"""
    p, ns = q + 1, 0
    for _ in range(4):
        s = (bits128 >> p) & 1
        p += 1 + s
        ns += s
    return 5 + ns


def check(lo, hi):
    e0, e1, e2 = planes(lo, hi)
    both = lo | (hi << 64)
    for q in range(64):
        got = 5 + ((e0 >> q) & 1) + 2 * ((e1 >> q) & 1) + 4 * ((e2 >> q) & 1)
        assert got == parse(both, q), (hex(lo), hex(hi), q)


def test_every_context_of_eight_bits_at_every_position():
    # the length at q depends on bits q+1 .. q+8 only: all 256 contexts, at the window's start, middle and across its end
    for ctx in range(256):
        for q in (0, 1, 31, 32, 55, 56, 57, 60, 62, 63):
            both = ctx << (q + 1)
            check(both & M, (both >> 64) & M)


def test_random_windows():
    rnd = random.Random(4)
    for density in (0.02, 0.2, 0.5, 0.8, 0.98):
        for _ in range(400):
            lo = sum((rnd.random() < density) << b for b in range(64))
            hi = sum((rnd.random() < density) << b for b in range(64))
            check(lo, hi)


def test_all_ones_and_all_zeros():
    check(0, 0)
    check(M, M)
    check(M, 0)
    check(0, M)


# ---- the all-type-A table of a window (decode.hip: helper_phase, second half) ----------------------------------------------
def tok_len(e, q):
    return 5 + ((e[0] >> q) & 1) + 2 * ((e[1] >> q) & 1) + 4 * ((e[2] >> q) & 1)


def ctz(x):
    return (x & -x).bit_length() - 1


def table_by_merging(lo, hi):
    """per entry point 0..8: (fired-entry positions, entries that start in the window, overhang), as the helper works them out:
    the walk from entry point 0 in full, the others only until they fall in step with it"""
    e = planes(lo, hi)
    S0 = fm0 = 0
    p = 0
    while p < 64:
        rem = lo >> p
        if rem == 0:
            S0 |= (M << p) & M
            p = 64
            break
        q = p + ctz(rem)
        S0 |= (((2 << (q - p)) - 1) << p) & M
        fm0 |= 1 << q
        p = q + tok_len(e, q)
    exit0 = p - 64
    rows = [(fm0, bin(S0).count("1"), exit0)]
    for o in range(1, 9):
        pp, cnt, ex, fmk = o, 0, 0, 0
        while True:
            if pp >= 64:
                ex = pp - 64
                break
            z = ctz(((lo | S0) >> pp) | (1 << (63 - pp)))
            cnt += z
            pp += z
            if (S0 >> pp) & 1:
                cnt += bin(S0 >> pp).count("1")
                fmk |= fm0 & ((M << pp) & M)
                ex = exit0
                break
            if not (lo >> pp) & 1:
                cnt += 1
                ex = 0
                pp = 64
                break
            fmk |= 1 << pp
            cnt += 1
            pp += tok_len(e, pp)
        rows.append((fmk, cnt, ex))
    return rows


def table_by_walking(lo, hi, o):
    """the same by the plain walk: every entry is type A, '0' takes one bit, '1' takes 5 + significant offspring"""
    both = lo | (hi << 64)
    p, cnt, fm = o, 0, 0
    while p < 64:
        cnt += 1
        if (both >> p) & 1:
            fm |= 1 << p
            p += parse(both, p)
        else:
            p += 1
    return fm, cnt, p - 64


def test_table_rows_equal_the_plain_walk():
    rnd = random.Random(5)
    for density in (0.0, 0.02, 0.1, 0.3, 0.5, 0.8, 1.0):
        for _ in range(300):
            lo = sum((rnd.random() < density) << b for b in range(64))
            hi = sum((rnd.random() < density) << b for b in range(64))
            rows = table_by_merging(lo, hi)
            for o in range(9):
                assert rows[o] == table_by_walking(lo, hi, o), (hex(lo), hex(hi), o)
    # fired entries right at the window's end, from every entry point
    for tail in range(1, 10):
        lo = (1 << 63) | (1 << (63 - tail))
        for hi in (0, M, 0x155, 0x0F):
            rows = table_by_merging(lo, hi)
            for o in range(9):
                assert rows[o] == table_by_walking(lo, hi, o), (hex(lo), hex(hi), o)
