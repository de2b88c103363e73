#!/usr/bin/env python3
"""one line per kernel from a bench.py JSON line: ms, algorithmic GB/s"""
import json, sys
r = json.load(open(sys.argv[1]))
print(f"{r['value']:.1f} {r['unit']}  ({r['ms_per_step'] / r['config'].get('n_iter', 1):.3f} ms/it)  " +
      "  ".join(f"{k}={v['ms']:.3f}ms/{v['GBps'] / 1000:.2f}TB/s" for k, v in r["kernels"].items()))
