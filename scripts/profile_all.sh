#!/bin/bash
# Round profile set on the GPU box: kernel-trace stats + PMC passes for every bench workload.
# usage: scripts/profile_all.sh <round tag, e.g. r04_v31> [a|b]   (two halves: one gpurun call each fits the 20-minute limit)
set -e
t=$1; half=${2:-ab}
if [[ $half == *a* ]]; then
scripts/profile.sh ${t}_4096 --min-seconds 1
scripts/profile.sh ${t}_8192 --min-seconds 1 --envs-per-gpu 8192
WARMUP=1000 scripts/profile.sh ${t}_walk --workload walk --min-seconds 0
WARMUP=1000 scripts/profile.sh ${t}_walk_newton --workload walk_newton --min-seconds 0
scripts/profile.sh ${t}_mixed --workload mixed --min-seconds 1
fi
if [[ $half == *b* ]]; then
WARMUP=1000 scripts/profile.sh ${t}_walk_pairs --workload walk_pairs --min-seconds 0
WARMUP=1000 scripts/profile.sh ${t}_walk_hfield --workload walk_hfield --min-seconds 0
WARMUP=1000 scripts/profile.sh ${t}_walk_mesh --workload walk_mesh --min-seconds 0
WARMUP=1000 scripts/profile.sh ${t}_walk_elliptic --workload walk_elliptic --min-seconds 0
WARMUP=1000 scripts/profile.sh ${t}_walk_pairs_newton --workload walk_pairs_newton --min-seconds 0
fi
