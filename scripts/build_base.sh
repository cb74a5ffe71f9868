#!/bin/bash
# A/B helper: build csrc of a git revision (default HEAD) as csrc/libfmj_hip_base.so; run with FMJ_SO=<that path>
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/fmj_base_XXXX)
git -C "$root" archive "$rev" farms_mujoco_amd/csrc include farms_mujoco_amd/_lib.py farms_mujoco_amd/__init__.py | tar -x -C "$tmp"
# the base is loaded by TODAY's host code: give it today's ABI number (fields are only ever appended to the structs, so an older
# library simply does not read the new tail)
abi=$(grep -o 'define FMJ_ABI_VERSION [0-9]*' "$root/include/fmj.h" | awk '{print $3}')
sed -i "s/#define FMJ_ABI_VERSION .*/#define FMJ_ABI_VERSION $abi/" "$tmp/include/fmj.h"
(cd "$tmp" && python - <<PY
import importlib.util, sys, os
spec = importlib.util.spec_from_file_location('_lib', 'farms_mujoco_amd/_lib.py'); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
m.build(force=True, out='libfmj_hip_base.so')
PY
)
cp "$tmp/farms_mujoco_amd/csrc/libfmj_hip_base.so" "$root/farms_mujoco_amd/csrc/"
rm -rf "$tmp"
ls -la "$root/farms_mujoco_amd/csrc/libfmj_hip_base.so"
