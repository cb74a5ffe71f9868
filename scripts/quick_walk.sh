#!/bin/bash
# quick GPU check for the constraint path: contact / morphology parity tests, then the walk and mixed bench lines
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_contacts.py tests/test_gpu_morphologies.py tests/test_gpu_random_trees.py tests/test_gpu_api_paths.py -m gpu -q -p no:cacheprovider -x > gpurun_out/quickw.log 2>&1; echo rc=$?; tail -3 gpurun_out/quickw.log
for w in ${WORKLOADS:-walk mixed}; do timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 1000 --warmup 1000 > gpurun_out/bw.json 2> gpurun_out/bw.err; python -c "import sys,json; d=json.loads(open('gpurun_out/bw.json').read().strip().splitlines()[-1]); print('$w', 'M/s %.2f' % (d['value']/1e6), d['launch_ms'])" || tail -3 gpurun_out/bw.err; done
