#!/usr/bin/env python3
"""HBM traffic of the forward-DWT level-1 kernel from rocprofv3 PMC counters -> profiles/dwt_l1_traffic.json.

Run on the GPU box (two separate --pmc passes, as MI355X_MICROARCH.md prescribes):
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out = {"images_per_launch": B}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                           "python3", os.path.join(ROOT, "tools", "prof_stage.py"), "dwt", str(B), "2"],
                          cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    best = {}
    for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "k_dwt_level" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                gs = int(r["Grid_Size"])
                best.setdefault(gs, []).append(float(r["Counter_Value"]))
    gs = max(best)  # the level-1 launch has the largest grid
    out[counter + "_KiB_raw"] = sum(best[gs]) / len(best[gs])
out["read_bytes"] = out["FETCH_SIZE_KiB_raw"] * 1024 * 2  # gfx950: FETCH_SIZE counts 64 B per 128 B request
out["write_bytes"] = out["WRITE_SIZE_KiB_raw"] * 1024
out["hbm_bytes_per_launch"] = out["read_bytes"] + out["write_bytes"]
out["hbm_bytes_per_image"] = out["hbm_bytes_per_launch"] / B
out["algorithmic_bytes_per_image"] = 3 * (1080 * 1920 * 8 + 542 * 962 * 20)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for dst in (os.path.join(ROOT, "gpurun_out", "dwt_l1_traffic.json"),):
    json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
