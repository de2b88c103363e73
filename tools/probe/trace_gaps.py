"""usage: trace_gaps.py <kernel_trace.csv> [n_last]  -- the last n_last dispatches (default 400) as a timeline summary:
per kernel name mean duration, count, and the mean idle gap BEFORE it (start - previous end)."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n:]
stat = collections.OrderedDict()
prev_end = None
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
    d = stat.setdefault(name, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) / 1e3
    if prev_end is not None:
        d[2] += max(0, s - prev_end) / 1e3
    busy += e - s
    prev_end = e
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} dispatches over {span / 1e3:.1f} us: busy {busy / 1e3:.1f} us ({100 * busy / span:.1f} %)")
for k, (c, d, g) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:5d} x {d / c:8.2f} us  gap before {g / c:6.2f} us  total {d:9.1f} us  {k}")
