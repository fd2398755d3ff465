#!/bin/bash
# usage: tools/experiments/build_variant.sh NAME [hipcc flags...]  -> climate_model_amd/csrc/libx_NAME.so (an experiment build of the same ABI, git-ignored)
set -e
cd "$(dirname "$0")/../../climate_model_amd/csrc"
n=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-inline-asm --offload-arch=gfx950 -DDYCORE_AB_SWITCHES "$@" -shared -o libx_$n.so dycore_hip.hip -ldl
