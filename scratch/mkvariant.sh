#!/bin/bash
# usage: scratch/mkvariant.sh name [-DFLAG ...]   -> scratch/libs/name.so
set -e
name=$1; shift
cd "$(dirname "$0")/../mono_lidar_depth_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function -shared "$@" -o ../../scratch/libs/$name.so mld_api.hip mld_params.cpp
