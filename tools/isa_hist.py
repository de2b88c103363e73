"""Instruction histogram of one kernel in a `hipcc -S` listing: python tools/isa_hist.py file.s <mangled-name substring> [top]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for m in re.finditer(r"\n(_Z\w+):", txt):
    if sys.argv[2] not in m.group(1):
        continue
    body = txt[m.end():].split("s_endpgm")[0]
    c = collections.Counter()
    for ln in body.splitlines():
        mm = re.match(r"\s+((?:v|s|ds|global|buffer|flat|scratch)_\w+)", ln)
        if mm:
            c[mm.group(1)] += 1
    print(m.group(1)[:100], "VALU", sum(v for k, v in c.items() if k.startswith("v_")))
    print("  " + ", ".join(f"{k} {v}" for k, v in c.most_common(top)))
