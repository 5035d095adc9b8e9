#!/bin/bash
# Developer tool: an A/B build of the library with extra -D flags -> build_variants/libatcstep_<name>.so (git-ignored, travels
# with gpurun; selected with bench.py --lib).     bash tools/build_variant.sh <name> [-DATC_...=...]
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -fPIC -shared -Wno-unused-function -I$ROOT/include "$@" \
    $ROOT/atc-reinforcement-learning_amd/csrc/atc_step.hip -o $ROOT/build_variants/libatcstep_$NAME.so && echo built $NAME
