#!/usr/bin/env python3
"""Turn what tools/profile_round.sh left under gpurun_out/<tag>/ into the committed summaries under profiles/<round>/.

    python tools/collect_profiles.py gpurun_out/<tag> profiles/r2 <prefix> [--config C3]
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys


def head_commit():
    """Short hash of the commit the profiled tree belongs to (the collection runs in the repository, right after the GPU call)."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        rev = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=here, capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(['git', 'status', '--porcelain', '--', 'ihgnn_amd', 'bench.py', 'tools/kbench.py'], cwd=here, capture_output=True, text=True).stdout.strip()
        return rev + ('+uncommitted' if dirty else '')
    except Exception:
        return 'unrecorded'


def main():
    src, dst, prefix = sys.argv[1:4]
    config = sys.argv[sys.argv.index('--config') + 1] if '--config' in sys.argv else 'C3'
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
    if stats:
        shutil.copy(max(stats, key=os.path.getmtime), os.path.join(dst, f'{prefix}_kernel_stats.csv'))
    for name in ('bench_under_rocprof.json', 'bench_default.json', 'k7_by_role.json'):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f'{prefix}_{name}'))
    fetch, write = os.path.join(src, 'pmc_FETCH_SIZE'), os.path.join(src, 'pmc_WRITE_SIZE')
    if os.path.isdir(fetch) and os.path.isdir(write):
        here = os.path.dirname(os.path.abspath(__file__))
        table = json.loads(subprocess.run([sys.executable, os.path.join(here, 'pmc_summary.py'), fetch, write, '--skip', '2'], capture_output=True, text=True, check=True).stdout)
        bench = json.load(open(os.path.join(src, 'pmc_FETCH_SIZE.json')))
        steps = '--steps 2 --warmup 1' if config == 'C5' else '--steps 3 --warmup 2'       # what tools/profile_round.sh runs for this config
        out = dict(command=f'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --config {config} {steps} --no-cpu-baseline '
                           '--no-extras --no-kernel-events   (one pass per counter, in situ: every kernel of the training step)',
                   notes='KiB counters; FETCH_SIZE doubled (gfx950 counts 128-B requests of wide reads as 64 B); both counters sit at the L2 <-> fabric '
                         'boundary, so Infinity-Cache hits are included: hbm_bytes_per_launch is L2-miss traffic, an upper bound of the HBM bytes',
                   workload=config, dim=bench['config']['dim'], edges=bench['config']['edges'], rows=bench['config'].get('distinct_hyperedges', bench['config']['edges']),
                   node_rows=bench['config'].get('nodes_in_hyperedges', bench['config']['nodes']), commit=head_commit(), kernels=table)
        for key, v in table.items():
            if key.startswith('edge_gather_sum_kernel'):
                out['edge_gather_sum'] = dict(hbm_bytes_per_launch=v['hbm_bytes_per_launch'], avg_us_under_pmc=v['avg_us_under_pmc'])
            if key.startswith('node_pair_sums_kernel'):                   # one launch per step under its own name: in situ
                # bench.py brackets the OP (the pair-sum kernel and the finish kernel of its split rows, which follows it): add that launch's time
                def finish_after_pairs(directory):
                    path = max(glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
                    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
                    seen, extra = set(), []
                    for i, r in enumerate(rows):
                        if 'node_pair_sums_kernel' in r['Kernel_Name'] and r['Start_Timestamp'] not in seen:
                            seen.add(r['Start_Timestamp'])
                            nxt = next((x for x in rows[i + 1:i + 8] if 'heavy_finish_kernel' in x['Kernel_Name']), None)
                            if nxt is not None:
                                extra.append((int(nxt['End_Timestamp']) - int(nxt['Start_Timestamp'])) / 1e3)
                    return sum(extra) / len(extra) if extra else 0.0
                finish_us = finish_after_pairs(fetch)
                out['node_pair_sums'] = dict(hbm_bytes_per_launch=v['hbm_bytes_per_launch'], avg_us_under_pmc=round(v['avg_us_under_pmc'] + finish_us, 1),
                                             kernel_us=v['avg_us_under_pmc'], finish_kernel_us=round(finish_us, 1),
                                             source='in situ: the pair-sum launch of each training step of the counter passes over bench.py (FETCH_SIZE doubled + WRITE_SIZE); '
                                                    'time = the pair-sum kernel + the finish kernel of its split rows (what bench.py brackets)')
        # K5 and K7's hyperedge -> node launch on their own (kbench): the step may not launch K5 at all, and K7's launch roles share a name
        f57, w57 = os.path.join(src, 'pmc_k57_FETCH_SIZE'), os.path.join(src, 'pmc_k57_WRITE_SIZE')
        if os.path.isdir(f57) and os.path.isdir(w57):
            alone = json.loads(subprocess.run([sys.executable, os.path.join(here, 'pmc_summary.py'), f57, w57, '--skip', '2'], capture_output=True, text=True, check=True).stdout)
            source = f'tools/kbench.py --config {config} --rounds 4 --ops k5,k7 (the kernel launched on its own, same shapes)'
            for key, v in alone.items():
                entry = dict(hbm_bytes_per_launch=v['hbm_bytes_per_launch'], avg_us_under_pmc=v['avg_us_under_pmc'], source=source)
                if key.startswith('edge_gather_sum_kernel') and 'edge_gather_sum' not in out:
                    out['edge_gather_sum'] = entry
                if key.startswith('node_segment_sum_kernel'):
                    out['k7.edges_to_nodes'] = entry
            out['kernels_alone'] = alone
        # K7's launches IN SITU by role: the seven long launches of a training step come in a fixed order (tools/k7_roles_from_trace.py); the
        # counter passes are cut into steps at the Adam launches and the first long K7 launch of a step is the hyperedge -> node pass of the
        # interactive layer - the kernel bench.py's `roofline` brackets inside the timed region
        def k7_in_situ(directory, counter):
            path = max(glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
            rows = sorted((r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter), key=lambda r: int(r['Start_Timestamp']))
            adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
            picked = []
            for a, b in zip(adam, adam[1:]):
                k7 = [r for r in rows[a + 1:b + 1] if 'node_segment_sum_kernel' in r['Kernel_Name'] and int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 100_000]
                if len(k7) == 7:                                         # (the hyperedge form of the interactive layer: IHG_NODE_LEVEL_FORWARD=0)
                    picked.append((float(k7[0]['Counter_Value']), (int(k7[0]['End_Timestamp']) - int(k7[0]['Start_Timestamp'])) / 1e3))
            return picked
        f_rows, w_rows = k7_in_situ(fetch, 'FETCH_SIZE'), k7_in_situ(write, 'WRITE_SIZE')
        if f_rows and w_rows:
            fk, wk = sum(v for v, _ in f_rows) / len(f_rows), sum(v for v, _ in w_rows) / len(w_rows)
            us = sum(t for _, t in f_rows + w_rows) / len(f_rows + w_rows)
            out['k7.edges_to_nodes'] = dict(hbm_bytes_per_launch=int(2 * fk * 1024 + wk * 1024), avg_us_under_pmc=round(us, 1), launches=len(f_rows),
                                            source='in situ: the first long K7 launch of each training step of the counter passes over bench.py (FETCH_SIZE doubled + WRITE_SIZE)')
        # the whole training step: the counter passes cut into steps at the Adam launches, every kernel between two of them summed
        def step_total(directory, counter):
            path = max(glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
            rows = sorted((r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter), key=lambda r: int(r['Start_Timestamp']))
            adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
            sums = [sum(float(r['Counter_Value']) for r in rows[a + 1:b + 1]) for a, b in zip(adam, adam[1:])]
            return (sum(sums[1:]) / len(sums[1:]) if len(sums) > 1 else (sums[0] if sums else None)), len(sums)     # (the first full step may still be a warm-up one)
        f_step, n_steps = step_total(fetch, 'FETCH_SIZE')
        w_step, _ = step_total(write, 'WRITE_SIZE')
        if f_step is not None and w_step is not None:
            out['step_l2_miss_bytes'] = int(2 * f_step * 1024 + w_step * 1024)
            out['step_l2_miss_note'] = f'FETCH_SIZE (doubled) + WRITE_SIZE summed over every kernel between two Adam launches, averaged over the last {max(n_steps - 1, 1)} steps of the pass'
        json.dump(out, open(os.path.join(dst, f'pmc_traffic_{config}.json'), 'w'), indent=1)
        for d, c in ((fetch, 'fetch_size'), (write, 'write_size')):
            f = max(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
            shutil.copy(f, os.path.join(dst, f'{prefix}_pmc_{c}.csv'))
    mf = glob.glob(os.path.join(src, 'pmc_mfma', '**', '*counter_collection.csv'), recursive=True)
    if mf:
        mf = [max(mf, key=os.path.getmtime)]      # (gpurun merges into gpurun_out/: an earlier run's files may still be there)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(mf[0])):
            name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
            name = (name[5:] if name.startswith('void ') else name).split('(')[0]
            if 'interact' in name:
                agg[name][r['Counter_Name']].append((float(r['Counter_Value']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
        program = f'tools/kbench.py --config {config} --rounds 3 --ops layer' if config == 'C3' else f'bench.py --config {config} --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-events'
        out = dict(command=f'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 {program}', notes='GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs', commit=head_commit(), kernels={})
        for name, c in agg.items():
            busy = sum(v for v, _ in c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024
            gui = sum(v for v, _ in c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE']) / 8
            us = sum(t for _, t in c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'])
            out['kernels'][name] = dict(mfma_busy_cycles_per_simd=round(busy), gpu_cycles_per_xcd=round(gui), mfma_busy_fraction=round(busy / gui, 4),
                                        avg_us_under_pmc=round(us, 1), clock_ghz=round(gui / us / 1e3, 3))
        json.dump(out, open(os.path.join(dst, f'{prefix}_pmc_mfma.json'), 'w'), indent=1)
        shutil.copy(mf[0], os.path.join(dst, f'{prefix}_pmc_mfma.csv'))
    print(sorted(os.listdir(dst)))


if __name__ == '__main__':
    main()
