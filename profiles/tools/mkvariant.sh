#!/bin/bash
# usage: profiles/tools/mkvariant.sh name [-DFLAG ...]   -> profiles/tools/libs/name.so
set -e
name=$1; shift
cd "$(dirname "$0")/../../mono_lidar_depth_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -pthread -Wall -Wno-unused-function -shared "$@" -o ../../profiles/tools/libs/$name.so mld_api.hip mld_params.cpp mld_host.cpp
