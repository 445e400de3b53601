#!/bin/bash
cd $GRAFT_REPO_ROOT
bash profiles/tools/ab2.sh 2
