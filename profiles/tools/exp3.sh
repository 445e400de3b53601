#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp3
MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/stamps.so timeout 300 python profiles/tools/stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp3/stamps.log
bash profiles/tools/ab2.sh 2 2>&1 | tee gpurun_out/exp3/ab.log
