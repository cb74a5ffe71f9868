#!/bin/bash
# Profile one bench configuration on the GPU box: kernel-trace stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate
# passes, MI355X_MICROARCH.md "rocprofv3 PMC slots") and SQ issue counters.  usage: scripts/profile.sh <tag> <bench args...>
# Outputs go to gpurun_out/prof_<tag>/ (scratch); scripts/pmc_summary.py turns them into profiles/<tag>_*.
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
args="--no-cpu-baseline --no-extras --steps ${STEPS:-1000} --warmup ${WARMUP:-1000} $@"
export TMPDIR=/tmp
python3 bench.py $args > $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py $args > /dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py $args > /dev/null
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py $args > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAIT_ANY --output-format csv -d $out/sq -- python3 bench.py $args > /dev/null
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $out/sq2 -- python3 bench.py $args > /dev/null
# keep only the small CSVs
find $out -name "*_agent_info.csv" -delete
ls -R $out | head -40
