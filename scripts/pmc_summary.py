"""Summarise the passes of scripts/profile.sh (gpurun_out/prof_<tag>/) into profiles/<tag>_pmc_summary.txt,
profiles/<tag>_kernel_stats.csv, profiles/<tag>_bench.json and an entry of profiles/latest_traffic.json (HBM bytes and issue
shares per env-step, which bench.py scales to its own launch size).
usage: python scripts/pmc_summary.py <tag> <workload> "<headline>" """
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, workload, head = sys.argv[1], sys.argv[2], sys.argv[3]
src = f'gpurun_out/prof_{tag}'
bench = json.loads(open(f'{src}/bench.json').read().strip().splitlines()[-1])
envs, steps = bench['config']['envs_per_gpu'], bench['config']['steps_per_launch']
b_alg = bench['roofline']['algorithmic_bytes_per_env_step']
dual = 'two envs per wave' in bench['roofline']['kernel']
waves = (envs + 1)//2 if dual else envs
out = [head, f'source: scripts/profile.sh {tag} (rocprofv3 --pmc passes around: python bench.py --no-cpu-baseline --no-extras --steps 1000 --warmup 1000 ...)',
       f'bench line of the same build: {bench["value"]/1e6:.1f} M env-steps/s, launch {bench["launch_ms"]["median"]:.3f} ms (median of {bench["launch_ms"]["n"]})',
       f'full launches = {envs} envs x {steps} steps; algorithmic bytes/launch = {b_alg} B x {envs*steps} = {b_alg*envs*steps:.3e}', '']
acc = collections.defaultdict(list)
meta = {}
for nm in ('fetch', 'write', 'sq', 'sq2'):
    for f in glob.glob(f'{src}/{nm}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'fmj_step' not in k or '<true' not in k:
                continue
            if int(r['Grid_Size']) != waves*64:          # full-size launches of the step kernel only
                continue
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = dict(kernel=k[:60], grid=r['Grid_Size'], vgpr=r.get('VGPR_Count', '?'), sgpr=r.get('SGPR_Count', '?'),
                        scratch=r.get('Scratch_Size', r.get('Private_Segment_Size', '?')), lds=r.get('LDS_Block_Size', '?'))
out.append(f'kernel {meta}')
mean = {k: sum(v)/len(v) for k, v in acc.items()}
n = {k: len(v) for k, v in acc.items()}
fb, wb = mean['FETCH_SIZE']*1024, mean['WRITE_SIZE']*1024
out += ['', f'per full launch (mean of {n["FETCH_SIZE"]} / {n["WRITE_SIZE"]} launches): FETCH_SIZE {fb/1e9:.4f} GB, WRITE_SIZE {wb/1e9:.4f} GB, sum {(fb+wb)/1e9:.4f} GB'
        f' = {(fb+wb)/(envs*steps):.1f} B per env-step against {b_alg} B algorithmic',
        '(FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM); the reads here are 4-16 B/lane table and state '
        'reads that stay in L2, reported as counted; WRITE_SIZE is exact for the 8/16-B row stores)', '']
for k in sorted(mean):
    if k.startswith('SQ_'):
        out.append(f'{k:<24s} per launch {mean[k]:>16.0f}   per wave-step ({waves} waves x {steps} steps) {mean[k]/(waves*steps):>10.1f}')
wc = mean.get('SQ_WAVE_CYCLES')
binding = None
if wc:
    binding = dict(valu_active_share=mean['SQ_ACTIVE_INST_VALU']/wc, wait_any_share=mean['SQ_WAIT_ANY']/wc,
                   note='SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES and SQ_WAIT_ANY / SQ_WAVE_CYCLES (quad-cycles of wave lifetime)')
    if 'SQ_WAIT_INST_ANY' in mean:
        binding['wait_inst_share'] = mean['SQ_WAIT_INST_ANY']/wc
    out += ['', f'shares of the wave lifetime: VALU active {binding["valu_active_share"]:.3f}, waiting at s_waitcnt {binding["wait_any_share"]:.3f}'
            + (f', issue stalls {binding["wait_inst_share"]:.3f}' if 'wait_inst_share' in binding else '')]
os.makedirs('profiles', exist_ok=True)
open(f'profiles/{tag}_pmc_summary.txt', 'w').write('\n'.join(out) + '\n')
for f in glob.glob(f'{src}/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, f'profiles/{tag}_kernel_stats.csv')
json.dump(bench, open(f'profiles/{tag}_bench.json', 'w'))
path = 'profiles/latest_traffic.json'
cur = [e for e in (json.load(open(path)) if os.path.exists(path) else []) if not (e['workload'] == workload and e['envs'] == envs)]
cur.append(dict(workload=workload, envs=envs, fetch_bytes_per_env_step=fb/(envs*steps), write_bytes_per_env_step=wb/(envs*steps),
                binding=binding, source=f'profiles/{tag}_pmc_summary.txt', build_id=bench.get('build_id')))
json.dump(cur, open(path, 'w'), indent=1)
print('\n'.join(out))
