#!/usr/bin/env python3
"""HBM traffic of the HBM-bound kernels from rocprofv3 PMC counters -> profiles/dwt_l1_traffic.json (the forward level-1
kernel, read by bench.py) and profiles/hbm_traffic_other.json (inverse level 1, significance pyramid: all its kernels).

Run on the GPU box (separate --pmc passes per counter, as MI355X_MICROARCH.md prescribes):
    python tools/collect_traffic.py [batch]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so the read
side is doubled before comparing with byte counts."""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256  # the bench's launch size
ITERS = 2


OCC = {}  # stage run -> fraction of occupied level-1 tiles printed by prof_stage.py idwtf


def counters(stage, kernel, bpp=None):
    """per counter: {(kernel name, grid size): mean counter value per launch} of the launches of the kernels whose name
    contains `kernel` in a run of `stage`"""
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        if bpp is not None:
            env["PROF_BPP"] = str(bpp)
        txt = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                              "python3", os.path.join(ROOT, "tools", "prof_stage.py"), stage, str(B), str(ITERS)],
                             cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
        for ln in txt.splitlines():
            if ln.startswith("occupied_tiles_fraction"):
                OCC[(stage, bpp)] = float(ln.split()[1])
        by = {}
        for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(p)):
                if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    by.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
        # persistent kernels have one grid size for every level they run: the ITERS largest values are level 1's
        res[counter] = {g: sum(sorted(v)[-ITERS:]) / min(len(v), ITERS) for g, v in by.items()}
    return res


def bytes_of(c, keys):
    rd = sum(c["FETCH_SIZE"][g] for g in keys) * 1024 * 2  # gfx950: FETCH_SIZE counts 64 B per 128 B request
    wr = sum(c["WRITE_SIZE"][g] for g in keys) * 1024
    return rd, wr


def level1(c):
    """the launches of level 1: per distinct kernel name (the interior-tile and the edge-tile instantiation) the one
    with the largest grid"""
    best = {}
    for (name, grid) in c["FETCH_SIZE"]:
        if name not in best or grid > best[name]:
            best[name] = grid
    return [(name, grid) for name, grid in best.items()]


os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
ONLY = sys.argv[2] if len(sys.argv) > 2 else None  # "idwtf": only the inverse level 1 with occupancy words, into the committed file; "dwt": only the forward level 1
if ONLY == "idwtf":
    other = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_other.json")))
    assert other["images_per_launch"] == B
    other["idwt_level1_with_occupancy_words"] = {}
    for bpp in (0.1, 0.5, 1.0):
        c = counters("idwtf", "false, true>", bpp)
        rd, wr = bytes_of(c, level1(c))
        other["idwt_level1_with_occupancy_words"]["%g bpp" % bpp] = {
            "occupied_tiles_fraction": OCC.get(("idwtf", bpp)), "read_bytes_per_image": rd / B, "write_bytes_per_image": wr / B,
            "hbm_bytes_per_image": (rd + wr) / B,
            "algorithmic_bytes_per_image_all_tiles_empty": 3 * (1080 * 1920 * 8 + 542 * 962 * 8)}
    json.dump(other, open(os.path.join(ROOT, "gpurun_out", "hbm_traffic_other.json"), "w"), indent=1)
    print(json.dumps(other))
    sys.exit(0)
# forward level 1: the largest grid of k_dwt_level and of the overhang fix-up kernel behind it (k_dwt_edge)
c = counters("dwt", "k_dwt_")
k1 = level1(c)
rd, wr = bytes_of(c, k1)
out = {"images_per_launch": B, "kernels": ["%s grid %d" % k for k in k1],
       "FETCH_SIZE_KiB_raw": sum(c["FETCH_SIZE"][k] for k in k1), "WRITE_SIZE_KiB_raw": sum(c["WRITE_SIZE"][k] for k in k1),
       "read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_image": (rd + wr) / B,
       "algorithmic_bytes_per_image": 3 * (1080 * 1920 * 8 + 542 * 962 * 20)}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "dwt_l1_traffic.json"), "w"), indent=1)
print(json.dumps(out))
if ONLY == "dwt":  # the forward level 1 alone (its kernel changed: the other files stand)
    sys.exit(0)
# inverse level 1 (largest grid of k_idwt_level) and the pyramid (all rounds of k_pyr_round + k_pyr_ll are small)
other = {"images_per_launch": B}
c = counters("idwt", "k_idwt_level_pf")  # the persistent kernel takes the large levels (one grid size: see counters())
if not c["FETCH_SIZE"]:
    c = counters("idwt", "k_idwt_level")
rd, wr = bytes_of(c, level1(c))
other["idwt_level1"] = {"read_bytes_per_image": rd / B, "write_bytes_per_image": wr / B, "hbm_bytes_per_image": (rd + wr) / B,
                        "algorithmic_bytes_per_image": 3 * (1080 * 1920 * 8 + 542 * 962 * 20)}
# ... and with the decoder's occupancy words (the default of the image-level calls): the detail bands of empty tiles are not
# read, so the read side depends on the bit rate
other["idwt_level1_with_occupancy_words"] = {}
for bpp in (0.1, 0.5, 1.0):
    c = counters("idwtf", "false, true>", bpp)  # the instantiation of the persistent kernel that reads the words: level 1 only
    rd, wr = bytes_of(c, level1(c))
    other["idwt_level1_with_occupancy_words"]["%g bpp" % bpp] = {
        "occupied_tiles_fraction": OCC.get(("idwtf", bpp)), "read_bytes_per_image": rd / B, "write_bytes_per_image": wr / B,
        "hbm_bytes_per_image": (rd + wr) / B,
        "algorithmic_bytes_per_image_all_tiles_empty": 3 * (1080 * 1920 * 8 + 542 * 962 * 8)}
c = counters("pyramid", "k_pyr_")  # k_pyr_12 (depths 1 and 2), k_pyr_round (deeper rounds), k_pyr_ll (root block)
rd, wr = bytes_of(c, list(c["FETCH_SIZE"]))
n_coef, n_par = 3 * 1111 * 1949, 3 * (1111 // 2) * (1949 // 2)
other["pyramid_rounds"] = {"read_bytes_per_image": rd / B, "write_bytes_per_image": wr / B, "hbm_bytes_per_image": (rd + wr) / B,
                           "algorithmic_bytes_per_image": 4 * n_coef + n_par + n_par // 4}
json.dump(other, open(os.path.join(ROOT, "gpurun_out", "hbm_traffic_other.json"), "w"), indent=1)
print(json.dumps(other))
