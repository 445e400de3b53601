#!/bin/bash
# kernel timeline of the default bench (two contexts alternating)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0 --no-estimated --no-kernel-timing --verify-slots 0 > $R/gpurun_out/tl.log 2>&1
cd $R
python3 profiles/tools/timeline.py gpurun_out/tl ${1:-40} | head -${2:-26}
