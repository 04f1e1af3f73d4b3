"""Timeline of the pipelined step out of a rocprofv3 --kernel-trace CSV: per HIP queue (= library context) the kernels
of one steady-state step in start order with their duration and the idle gap in front of each, so that what a stage's
event-bracketed time is made of (kernels vs the gaps between dependent launches) can be read off.
    python3 tools/timeline.py <rocprof output dir> [step index from the end, default 3]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for r in csv.DictReader(open(f)):
    g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name'], g))
rows.sort()
# a step starts with the level-1 forward kernel (the largest k_dwt_level grid)
big = max(g for _, _, _, k, g in rows if k.startswith('void k_dwt_level<'))
starts = [s for s, _, _, k, g in rows if k.startswith('void k_dwt_level<') and g == big]
if len(starts) < back + 2:
    sys.exit("not enough steps in the trace")
t0, t1 = starts[-back - 1], starts[-back]
print("step of %.3f ms (between two level-1 forward launches, %d launches from the end)" % ((t1 - t0) / 1e6, back))
byq = collections.defaultdict(list)
for s, e, q, k, g in rows:
    if t0 <= s < t1:
        byq[q].append((s, e, k, g))
for q, v in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in v) / 1e6
    print("\nqueue %s: %d kernels, busy %.3f ms" % (q, len(v), busy))
    prev = None
    for s, e, k, g in v:
        gap = (s - prev) / 1e3 if prev is not None else 0.0
        print("  +%8.3f ms  %-44s grid %10d  %8.3f ms   gap before %7.1f us" % ((s - t0) / 1e6, k[:44], g, (e - s) / 1e6, gap))
        prev = e
