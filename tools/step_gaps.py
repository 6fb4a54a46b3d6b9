#!/usr/bin/env python3
"""Gaps between the kernels of one training step, from a rocprofv3 kernel trace of bench.py:

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events
    python3 tools/step_gaps.py DIR [MIN_GAP_US] [--all | --diff STEP_A STEP_B]

The step = the launches between the last two Adam launches.  Prints its span, the sum of its kernels, and every gap of at least MIN_GAP_US (default 50) with the kernels on
either side - what the host (allocator, Python) adds to a step that is otherwise back-to-back kernels."""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith('--') else 50.0
    path = max(glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    if len(adam) < 2:
        sys.exit('fewer than two Adam launches in the trace')
    # the last launch of a step's Adam group: Adam launches that follow each other belong to one step
    ends = [i for n, i in enumerate(adam) if n + 1 == len(adam) or adam[n + 1] != i + 1]
    if '--diff' in sys.argv:                                               # two steps side by side in launch order: --diff A B (step numbers of --all)
        k = sys.argv.index('--diff')
        sa, sb = int(sys.argv[k + 1]), int(sys.argv[k + 2])
        la, lb = rows[ends[sa - 1] + 1:ends[sa] + 1], rows[ends[sb - 1] + 1:ends[sb] + 1]
        for n in range(max(len(la), len(lb))):
            ra, rb = (la[n] if n < len(la) else None), (lb[n] if n < len(lb) else None)
            da = (int(ra['End_Timestamp']) - int(ra['Start_Timestamp'])) / 1e3 if ra else 0.0
            db = (int(rb['End_Timestamp']) - int(rb['Start_Timestamp'])) / 1e3 if rb else 0.0
            flag = '  <--' if abs(da - db) > 0.05 * max(da, db) and abs(da - db) > 200 else ''
            print(f'{n:3d} {da:10.1f} {db:10.1f} us  {(ra or rb)["Kernel_Name"][28:120]:92s} | {(rb or ra)["Kernel_Name"][28:70]}{flag}')
        return
    if '--all' in sys.argv:                                                # every step of the trace: span and the kernels that differ most from the last step's
        import collections
        def table(i, j):
            t = collections.defaultdict(float)
            for r in rows[i + 1:j + 1]:
                t[r['Kernel_Name'][:90]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            return t
        ref = table(ends[-2], ends[-1])
        for n in range(1, len(ends)):
            i, j = ends[n - 1], ends[n]
            span = (int(rows[j]['End_Timestamp']) - int(rows[i]['End_Timestamp'])) / 1e6
            t = table(i, j)
            diff = sorted(((t.get(k, 0.0) - ref.get(k, 0.0), k) for k in set(t) | set(ref)), key=lambda x: -abs(x[0]))[:4]
            print(f'step {n:3d}: {j - i:4d} launches, span {span:9.3f} ms, kernels {sum(t.values()):9.3f} ms | vs last step: ' + '; '.join(f'{d:+.2f} {k[:48]}' for d, k in diff if abs(d) > 0.3))
        return
    a, b = ends[-2], ends[-1]
    step = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]['End_Timestamp']), int(step[-1]['End_Timestamp'])
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
    print(f'{len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, gaps {(t1 - t0 - busy) / 1e6:.3f} ms')
    prev_end, prev_name = t0, 'adam_kernel (previous step)'
    for r in step:
        gap = (int(r['Start_Timestamp']) - prev_end) / 1e3
        if gap >= min_gap:
            print(f'  gap {gap:10.1f} us  after {prev_name[:70]:70s} before {r["Kernel_Name"][:70]}')
        prev_end, prev_name = int(r['End_Timestamp']), r['Kernel_Name']


if __name__ == '__main__':
    main()
