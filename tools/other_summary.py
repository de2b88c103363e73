#!/usr/bin/env python3
"""prints the other_configs block of a bench.py JSON line, one line per config + one per kernel"""
import json, sys
r = json.load(open(sys.argv[1]))
for c in r.get("other_configs", []):
    print(f"{c['config'][:40]:40s} {c['value']:10.1f} {c['unit']:20s} {c['ms_per_call']:8.3f} ms  frac {c['whole_call_frac_of_peak']:.3f}  " +
          " ".join(f"{k}={v['ms']:.3f}/{(v['GBps'] or 0) / 1000:.2f}" for k, v in c["kernels"].items()))
