#!/usr/bin/env python3
"""Per-kernel share of a step from a bench.py JSON line: python tools/bench_table.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('ms/step', d['ms_per_step'], 'batch-rows last layer', d.get('batch_rows_last_layer_ms_per_step'), 'forward only', d.get('fwd_only_ms'))
total = 0.0
for name, v in sorted(d.get('kernels_us', {}).items(), key=lambda kv: -kv[1]['avg_us'] * kv[1]['launches_per_step']):
    ms = v['avg_us'] * v['launches_per_step'] / 1e3
    total += ms
    print(f"{name:32s} {v['avg_us'] / 1e3:9.3f} ms x {v['launches_per_step']:4.1f} = {ms:9.2f} ms")
print('sum of bracketed launches', round(total, 2), 'ms')
