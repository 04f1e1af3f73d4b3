#!/usr/bin/env python3
"""How much of a stream's LIS passes lies in runs of ONE entry type?  (CPU only; round 4)

The list decoder's sequencer (csrc/decode.hip) walks the LIS pass serially because the bit a list entry reads decides how
many bits it takes -- 1, or 5..9 for a type-A entry that fires -- and whether an entry CAN fire depends on its type, i.e. on
its index in the queue, while the bits are addressed by stream position.  Where the queue is of one type over a long
stretch the walk does not depend on that alignment: all-B stretches take one bit per entry, all-A stretches are a pure
function of the stream position (as the LIP pass, which is decoded by a scan without any walker).  This tool counts, on
the oracle's per-bit trace (decode_with_metadata: action 2 = type-A entry's bit, 5 = type-B entry's bit), how the LIS
entries of a stream are distributed over runs of equal type, and how many 64-bit windows of the LIS passes would see a
uniform queue.  Usage: tools/lis_type_runs.py [cfg2|cfg5]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def nominal_slices(lh, lw, levels):
    top = [(0, lh), (0, lw)]
    other = []
    for t in range(1, levels + 1):
        a_h, a_w = lh << (t - 1), lw << (t - 1)
        other.append([[(a_h, 2 * a_h), (0, a_w)], [(0, a_h), (a_w, 2 * a_w)], [(a_h, 2 * a_h), (a_w, 2 * a_w)]])
    return top, other


def generations(h, w, lh, lw):
    best = 1
    for r, c in ((0, lw), (lh, 0)):
        t = 1
        while 2 * (r << (t - 1)) + 1 < h and 2 * (c << (t - 1)) + 1 < w:
            t += 1
        best = max(best, t)
    return best


def analyse(name, data, n, c, h, w, lh, lw):
    from oracle import oracle as O
    top, other = nominal_slices(lh, lw, generations(h, w, lh, lw))
    _rec, meta = O.decode_with_metadata(data, n, c, h, w, lh, lw, top, other)
    act = meta[:-1, 0].astype(np.int8)       # one row per stream bit
    plane = meta[:-1, 6]
    nb = act.size
    is_ent = (act == 2) | (act == 5)         # rows where a LIS entry starts
    lis_bit = (act >= 2) & (act <= 5)
    # fired type-A entries: an action-2 row followed by an action-3 row
    nxt = np.append(act[1:], -1)
    fired_a = (act == 2) & (nxt == 3)
    ent_pos = np.nonzero(is_ent)[0]
    ent_type = act[ent_pos]                  # 2 = A, 5 = B, in processing order
    ent_plane = plane[ent_pos]
    # runs of equal type inside a plane's LIS pass (generation boundaries are not in the trace: they can only split runs
    # further, so these run lengths are upper bounds)
    brk = np.nonzero((ent_type[1:] != ent_type[:-1]) | (ent_plane[1:] != ent_plane[:-1]))[0] + 1
    starts = np.concatenate([[0], brk])
    lens = np.diff(np.concatenate([starts, [ent_type.size]]))
    rtype = ent_type[starts]
    tot = ent_type.size
    print("== %s: %d stream bits, %d of them in LIS passes (%.1f %%); %d LIS entry visits (%d type A, %d type B), %d fired type-A entries"
          % (name, nb, int(lis_bit.sum()), 100.0 * lis_bit.sum() / nb, tot, int((ent_type == 2).sum()), int((ent_type == 5).sum()),
             int(fired_a.sum())))
    print("   runs of one type: %d, mean length %.2f entries, median %d, 90th percentile %d, longest %d"
          % (lens.size, lens.mean(), int(np.median(lens)), int(np.percentile(lens, 90)), int(lens.max())))
    for thr in (8, 16, 32, 64, 128):
        m = lens >= thr
        print("   entries in runs of >= %3d: %5.1f %%  (type A %5.1f %%, type B %5.1f %%)"
              % (thr, 100.0 * lens[m].sum() / tot, 100.0 * lens[m & (rtype == 2)].sum() / tot, 100.0 * lens[m & (rtype == 5)].sum() / tot))
    # 64-bit windows of the LIS passes: uniform if every entry that starts in the window -- and the 64 queue entries from
    # the window's first entry on, which is what the sequencer's type mask covers -- are of one type
    run_id = np.repeat(np.arange(lens.size), lens)           # run of every entry
    run_end = np.repeat(starts + lens, lens)                  # index one past the entry's run
    widx = ent_pos >> 6
    first_in_win = np.concatenate([[True], widx[1:] != widx[:-1]])
    fi = np.nonzero(first_in_win)[0]                          # first entry of every window that has one
    uniform = run_end[fi] - fi >= 64
    hops_in_uniform = 0
    fa_ent = fired_a[ent_pos]
    csum = np.concatenate([[0], np.cumsum(fa_ent)])
    nxt_fi = np.append(fi[1:], ent_type.size)
    hops_per_win = csum[nxt_fi] - csum[fi]
    hops_in_uniform = int(hops_per_win[uniform].sum())
    print("   64-bit windows with LIS entries: %d; with a queue of ONE type over the 64 entries the window can reach: %d (%.1f %%), "
          "holding %.1f %% of the fired type-A entries (the sequencer's hops: %d in all, %.2f per window)"
          % (fi.size, int(uniform.sum()), 100.0 * uniform.sum() / fi.size, 100.0 * hops_in_uniform / max(1, int(fa_ent.sum())),
             int(fa_ent.sum()), fa_ent.sum() / fi.size))
    # the same windows by the types of the entries that actually START in them (a window with fired entries reaches far fewer
    # than 64 entries): kinds A, B, AB (type A then type B), BA, and more changes than one
    chg = (ent_type[1:] != ent_type[:-1]).astype(np.int64)
    cchg = np.concatenate([[0], np.cumsum(chg)])               # changes before entry i (between i-1 and i counted at i)
    last = nxt_fi - 1
    nchg = cchg[last] - cchg[fi]
    first_t = ent_type[fi]
    kinds = {"all A": (nchg == 0) & (first_t == 2), "all B": (nchg == 0) & (first_t == 5), "A then B": (nchg == 1) & (first_t == 2),
             "B then A": (nchg == 1) & (first_t == 5), "two changes": nchg == 2, "three and more": nchg >= 3}
    for k, m in kinds.items():
        print("   windows whose own entries are %-15s %6d (%5.1f %%), %5.1f %% of the hops, %5.1f entries a window"
              % (k + ":", int(m.sum()), 100.0 * m.sum() / fi.size, 100.0 * hops_per_win[m].sum() / max(1, int(fa_ent.sum())),
                 (nxt_fi - fi)[m].mean() if m.any() else 0.0))
    return dict(bits=nb, entries=tot, hops=int(fa_ent.sum()), windows=int(fi.size), uniform_windows=int(uniform.sum()))


def main():
    from oracle import oracle as O
    from conftest import synth_coeffs, synth_image
    which = sys.argv[1:] or ["cfg2"]
    if "cfg2" in which:
        img = synth_image(1000, 3, 1080, 1920)
        data, n, g = O.encode_image(img, "bior2.2", "reflect", 7, 50.0, None, int(1080 * 1920 * 0.5))
        analyse("cfg2: bench image seed 1000, 1080p RGB, bior2.2 level 7, 0.5 bpp", data, n, 3, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"])
    if "cfg5" in which:
        g = O.geometry(4096, 4096, "bior6.8", 9)
        x = synth_coeffs(42, 3, g["enc_h"], g["enc_w"], g["ll_h"], g["ll_w"])
        for bpp in (0.1, 1.0):
            data, n = O.encode(x, g["ll_h"], g["ll_w"], int(4096 * 4096 * bpp))
            analyse("cfg5: 4096x4096 RGB coefficient array (SURVEY 8d generator), bior6.8 level 9, %g bpp" % bpp, data, n, 3, g["enc_h"], g["enc_w"],
                    g["ll_h"], g["ll_w"])


if __name__ == "__main__":
    main()
