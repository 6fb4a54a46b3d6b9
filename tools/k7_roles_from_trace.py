#!/usr/bin/env python3
"""tools/k7_roles_from_trace.py DIR: K7's launches by ROLE from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py (DIR = its -d
directory).  All of K7's launches share one kernel name (`node_segment_sum_kernel<4, 32>`) and, at one workload, one grid, so rocprofv3's
--stats line averages unlike jobs; this cuts the trace into training steps at the Adam launches and averages the large K7 launches of the
split-arithmetic steps by their POSITION in the step, which is fixed (bench.py --config C3, ihgnn, order 3):
forward: hyperedge features -> nodes, two-hop (layer 1), two-hop (layer 2); backward: two-hop (layer 2, masked), two-hop (layer 1), member
gradients -> nodes, first-order cotangent -> nodes.  Prints one JSON object."""
import csv
import glob
import json
import sys

# (node-level form of the interactive layer - the default at d = 64 / 128 / 256: no hyperedge -> node launch, six long K7 launches per step; IHG_NODE_LEVEL_FORWARD=0: seven)
ROLES = ['k7.edges_to_nodes', 'k7.two_hop (layer 1)', 'k7.two_hop (layer 2)', 'k7.two_hop_bwd_masked (layer 2: only the batch rows of its cotangent are non-zero)', 'k7.two_hop_bwd (layer 1)',
         'k7.member_gradients_rows', 'k7.first_order_gradient']


def main():
    f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
    per_role = [[] for _ in ROLES]
    steps = 0
    all_steps = []
    for a, b in zip(adam, adam[1:]):
        step = rows[a + 1:b + 1]
        if not any('interact_bwd_members_split' in r['Kernel_Name'] for r in step):
            continue                                                     # a step of the fp32-MFMA comparison pass, or not a training step
        k7 = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in step
              if 'node_segment_sum_kernel' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 100_000]
        if len(k7) == len(ROLES) - 1:
            k7 = [None] + k7                                             # node-level form: the first role does not exist
        if len(k7) != len(ROLES) or k7[2] < 0.8 * k7[1]:                 # (the steps with the last layer restricted to the batch rows have a short second two-hop)
            continue
        all_steps.append(k7)
    # the headline steps pull only the batch rows in the last layer's backward (a short fourth launch); the trace also holds the bench's
    # comparison steps with the dense pull
    headline = [k7 for k7 in all_steps if k7[3] < 0.7 * k7[4]] or all_steps
    for k7 in headline:
        steps += 1
        for lst, us in zip(per_role, k7):
            if us is not None:
                lst.append(us)
    out = dict(command='rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline',
               kernel='node_segment_sum_kernel<4, 32>', steps=steps,
               note='launches longer than 100 us, by position in the training step, steps with every layer over all rows (the small launches of the same kernel are the bag means and the batch tail)',
               roles={name: dict(avg_us=round(sum(v) / len(v), 1), min_us=round(min(v), 1), max_us=round(max(v), 1), launches=len(v)) for name, v in zip(ROLES, per_role) if v})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
