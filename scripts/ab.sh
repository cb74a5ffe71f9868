#!/bin/bash
# A/B on one box: builds of the library against each other, alternating, REPS times.
# default pair: csrc/libfmj_hip_base.so (scripts/build_base.sh <rev>) and csrc/libfmj_hip.so; SOS="a.so b.so c.so" names others
w=${1:-walk}; shift
for i in $(seq ${REPS:-2}); do for so in ${SOS:-libfmj_hip_base.so libfmj_hip.so}; do
  FMJ_SO=$PWD/farms_mujoco_amd/csrc/$so timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 1000 --warmup 1000 "$@" 2> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so', '$w', round(d['value']/1e6,2), round(d['launch_ms']['median'],3))" || exit 1
done; done
