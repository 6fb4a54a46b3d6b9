#!/usr/bin/env python3
"""Per-kernel share of a step from a bench.py JSON line: python tools/bench_table.py <file>; several files: one summary line each."""
import json
import os
import sys


def load(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


if len(sys.argv) > 2:
    for path in sys.argv[1:]:
        try:
            d = load(path)
            r = d.get('roofline') or {}
            print(f"{os.path.basename(os.path.dirname(path))}/{os.path.basename(path):34s} {d['config'].get('workload', '')[:28]:28s} ms/step {d['ms_per_step']:9.3f} value {d['value']:.4g} "
                  f"recorded {d.get('recorded_step_ms_per_step')} fwd {d.get('fwd_only_ms')} | roofline {str(r.get('kernel', ''))[:24]} {r.get('avg_us')} frac {r.get('frac')} traffic {r.get('traffic')}")
        except Exception as e:                                       # a run that failed leaves an empty file: say so and go on
            print(path, 'unreadable:', e)
    sys.exit(0)
d = load(sys.argv[1])
print('ms/step', d['ms_per_step'], 'batch-rows last layer', d.get('batch_rows_last_layer_ms_per_step'), 'forward only', d.get('fwd_only_ms'))
total = 0.0
for name, v in sorted(d.get('kernels_us', {}).items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches_per_step']):
    ms = v['avg_us'] * v['launches_per_step'] / 1e3
    total += ms
    print(f"{name:32s} {v['avg_us'] / 1e3:9.3f} ms x {v['launches_per_step']:4.1f} = {ms:9.2f} ms")
print('sum of bracketed launches', round(total, 2), 'ms')
