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
