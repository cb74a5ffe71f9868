"""Register / spill / scratch figures of every step-kernel instantiation of one register row length (default 20), from the
compiler's resource remarks.  usage: python scripts/kres.py [MAXD] [extra hipcc flags]"""
import os
import re
import subprocess
import sys

maxd = sys.argv[1] if len(sys.argv) > 1 else '20'
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'farms_mujoco_amd', 'csrc', 'fmj_hip.hip')
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fno-slp-vectorize', '-mllvm', '-pragma-unroll-threshold=131072', '-fPIC',
       f'-DFMJ_TU_MAXD={maxd}'] + sys.argv[2:] + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', f'/tmp/fmj_kres_{maxd}.o']
text = subprocess.run(cmd, capture_output=True, text=True).stderr
keep = {'VGPRs': 'VGPR', 'TotalSGPRs': 'SGPR', 'VGPRs Spill': 'vspill', 'SGPRs Spill': 'sspill', 'ScratchSize [bytes/lane]': 'scratch',
        'Occupancy [waves/SIMD]': 'occ', 'LDS Size [bytes/block]': 'lds'}
cur = None
for line in text.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        if cur:
            print(cur)
        name = re.sub(r'^_Z\d+', '', t.split(':', 1)[1].strip()).replace('8DevModel8StepArgs', '')
        cur = f'{name:44s}'
    else:
        k, v = t.split(':', 1)
        if k.strip() in keep:
            cur += f' {keep[k.strip()]}={v.strip()}'
if cur:
    print(cur)
