import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    d[(r['Kernel_Name'], g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
rows = sorted(d.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))
print("rocprofv3 --kernel-trace of `python3 bench.py` (256 images per launch; the single-image latency probe adds the small-grid launches, the kernel-alone probe a few serial ones): per kernel and grid size")
for (k, g), v in rows[:16]:
    print(f"{k[:40]:40s}   grid {g:10d}  launches {len(v):3d}  avg {sum(v)/len(v):8.3f} ms  min {min(v):8.3f}  max {max(v):8.3f}")
