#!/bin/bash
cd $GRAFT_REPO_ROOT
MLD_HIP_LIBRARY=$PWD/profiles/tools/libs/stamps.so timeout 300 python profiles/tools/stamps.py 2>&1 | grep -v amdgpu.ids
bash profiles/tools/pmc_sq.sh fused
bash profiles/tools/pmc_sq.sh legacy MLD_LEGACY_SPLIT=1
