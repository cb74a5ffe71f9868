#!/bin/bash
# instruction-cache counters of one bench workload (own PMC pass, no tracing).  usage: scripts/icache.sh <workload>
set -e
w=${1:-walk}
out=$PWD/gpurun_out/icache_$w
mkdir -p $out
export TMPDIR=/tmp
args="--workload $w --no-cpu-baseline --no-extras --steps 300 --warmup 300"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $out -- python3 bench.py $args > /dev/null
find $out -name "*_agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k, {c: round(v/cnt[(k, c)]) for c, v in d.items()})
PY
