#!/bin/bash
# Developer tool: ISA of one kernel instantiation without debug directives -> /tmp/<tag>.s, plus static instruction totals.
#   bash tools/isa_extract.sh <tag> <mangled-name-prefix, e.g. _Z6k_stepILi16ELb0ELb1E> [extra hipcc flags]
TAG=$1; K=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 --cuda-device-only -S -I$ROOT/include "$@" \
    $ROOT/atc-reinforcement-learning_amd/csrc/atc_step.hip -o /tmp/$TAG.full.s 2>/dev/null
awk -v k="$K" 'index($0,k)==1 && /:/ {on=1} on {print} on && /s_endpgm/ {exit}' /tmp/$TAG.full.s | grep -v "^\s*;" | grep -v "^\s*\.\(p2align\|loc\)" > /tmp/$TAG.s
echo "$K: $(grep -c '^\s*v_' /tmp/$TAG.s) VALU, $(grep -c '^\s*s_' /tmp/$TAG.s) SALU/SMEM, $(grep -c 's_load' /tmp/$TAG.s) s_load, $(grep -c 's_waitcnt' /tmp/$TAG.s) s_waitcnt, $(grep -c 's_cbranch' /tmp/$TAG.s) branches, $(grep -c 'scratch_' /tmp/$TAG.s) scratch ops"
grep -A30 "^\s*\.amdhsa_kernel $K" /tmp/$TAG.full.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed" 
