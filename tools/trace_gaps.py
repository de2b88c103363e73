"""Kernel durations and the gaps between consecutive dispatches from a rocprofv3 --kernel-trace CSV
(usage: trace_gaps.py <kernel_trace.csv> [last N dispatches = 2000]): per (previous kernel -> kernel) pair the median gap
from the previous kernel's end to this one's start and this kernel's median duration, in microseconds."""
import csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
short = lambda s: re.sub(r"<.*", "", s.replace("void ", "").replace("lpc::", ""))[:40]
pairs = {}
for a, b in zip(rows, rows[1:]):
    k = (short(a["Kernel_Name"]), short(b["Kernel_Name"]))
    pairs.setdefault(k, []).append(((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3,
                                    (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3))
print("| previous kernel -> kernel | dispatches | gap (end -> start), us | duration, us |\n|---|---|---|---|")
tot = 0.0
for k, v in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
    if len(v) < max(3, len(rows) // 200): continue
    g, d = statistics.median(x[0] for x in v), statistics.median(x[1] for x in v)
    print("| %s -> %s | %d | %.2f | %.2f |" % (k[0], k[1], len(v), g, d))
