#!/bin/bash
# quick GPU check used while tuning kernels: the parity tests that exercise the two-env kernel, then bench lines
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_golden.py tests/test_gpu_step_parity.py tests/test_gpu_fused_parity.py tests/test_gpu_api_paths.py tests/test_gpu_random_trees.py tests/test_gpu_morphologies.py tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -x > gpurun_out/quick.log 2>&1; echo rc=$?; tail -3 gpurun_out/quick.log
for e in ${ENVS:-4096 8192}; do timeout -k 10 200 python bench.py --envs-per-gpu $e --no-cpu-baseline --no-extras --steps 1000 --warmup 1000 > gpurun_out/b.json 2> gpurun_out/b.err; python -c "import sys,json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1]); print('envs', d['config']['envs_per_gpu'], 'M/s %.1f' % (d['value']/1e6), d['launch_ms'])" || tail -3 gpurun_out/b.err; done
