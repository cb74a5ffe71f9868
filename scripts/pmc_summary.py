"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc_*_<tag>) into profiles/r01_<tag>_pmc_summary.txt
and profiles/latest_traffic.json.  usage: python scripts/pmc_summary.py v6 "<headline>" """
import csv, glob, json, sys, collections

tag, head = sys.argv[1], sys.argv[2]
ENVS, STEPS = 4096, 100
out = [head,
       'commands: rocprofv3 --pmc <counters> --output-format csv -- python bench.py --steps 300 --warmup 300 --no-cpu-baseline',
       f'full launches = {ENVS} envs x {STEPS} steps; algorithmic bytes/launch = 4204 B x {ENVS*STEPS} = {4204*ENVS*STEPS:.3e}', '']
acc = collections.defaultdict(list)
for nm in ('fetch', 'write', 'sq'):
    for f in glob.glob(f'gpurun_out/pmc_{nm}_{tag}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'fmj_step' not in r['Kernel_Name'] or '<true' not in r['Kernel_Name']:
                continue
            v = float(r['Counter_Value'])
            acc[r['Counter_Name']].append(v)
            if nm != 'sq':
                out.append(f"{r['Counter_Name']},{r['Kernel_Name'][:44]},grid={r['Grid_Size']},scratch={r.get('Scratch_Size', r.get('Private_Segment_Size','?'))},"
                           f"vgpr={r.get('VGPR_Count','?')},value_KiB={v:.3f}")
mean = {k: sum(v) / len(v) for k, v in acc.items()}
fb, wb = mean['FETCH_SIZE'] * 1024, mean['WRITE_SIZE'] * 1024
out += ['', f'per full launch: FETCH_SIZE {fb/1e9:.3f} GB, WRITE_SIZE {wb/1e9:.3f} GB, sum {(fb+wb)/1e9:.3f} GB',
        '(FETCH_SIZE can read 1/2 of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md); these reads are 4-16 B/lane '
        'table and state reads, uncalibrated, reported as counted)', '']
waves = ENVS // 2 if 'dual' in head else ENVS
for k in sorted(mean):
    if k.startswith('SQ_'):
        out.append(f'{k:<22s} per launch {mean[k]:>14.0f}   per wave-step ({waves} waves x {STEPS} steps) {mean[k]/(waves*STEPS):>9.1f}')
open(f'profiles/r01_{tag}_pmc_summary.txt', 'w').write('\n'.join(out) + '\n')
json.dump({'workload': 'swim', 'steps_per_launch': STEPS, 'envs': ENVS, 'fetch_bytes': fb, 'write_bytes': wb,
           'source': f'profiles/r01_{tag}_pmc_summary.txt'}, open('profiles/latest_traffic.json', 'w'))
print('\n'.join(out))
