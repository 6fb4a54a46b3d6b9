"""tools/step_kernels.py DIR: the kernels of ONE steady-state training step in launch order, from a `rocprofv3 --kernel-trace --output-format csv`
run of bench.py (DIR = its -d directory): name, duration, gap to the previous kernel's end.  The step is cut at the Adam launches."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = adam[len(adam) // 2], adam[len(adam) // 2 + 1]
prev_end, total, busy = int(rows[a]['End_Timestamp']), 0, 0
for r in rows[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').replace('at::native::', '')[:70]
    print(f'{(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {name}')
    busy += e - s
    prev_end = e
print(f'step {(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms, {b - a} launches')
