#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp2
O=gpurun_out/exp2
timeout 200 profiles/tools/libs/randgather 4096 > $O/randgather.log 2>&1
cat $O/randgather.log
MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/stamps.so timeout 300 python profiles/tools/stamps.py > $O/stamps.log 2>&1
cat $O/stamps.log | grep -v amdgpu.ids
bash profiles/tools/pmc_feature.sh r2a 2>&1 | tail -30
rm -f profiles/tools/libs/stamps.so.bak
mv profiles/tools/libs/stamps.so /tmp/stamps.so; mv profiles/tools/libs/randgather /tmp/randgather
cp mono_lidar_depth_amd/lib/libmld_hip.so profiles/tools/libs/base.so
bash profiles/tools/ab.sh 2>&1 | grep -v amdgpu.ids
