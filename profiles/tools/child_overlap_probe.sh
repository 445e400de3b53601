#!/bin/bash
# Does a second process with an initialised (idle) GPU context change how two streams of THIS process overlap?
# the dense two-context leg (S = 16) alone, then as the child of a python that has initialised the GPU
show() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['configs']['5']['batched']['16']; t=d['two_contexts']['classify']
print('one', round(d['ms_per_step'],4), 'two', round(t['ms_per_step'],4), {k:round(v*1e3) for k,v in t['kernels_ms_per_launch'].items()}, d['verified'])"; }
echo -n "alone:            "; python bench_support/run_legs.py --legs c5b16t 2>/dev/null | show
echo -n "under GPU parent: "; python - <<'P' | show
import subprocess, sys, torch
torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
r = subprocess.run([sys.executable, "bench_support/run_legs.py", "--legs", "c5b16t"], capture_output=True, text=True)
print(r.stdout)
P
echo -n "alone again:      "; python bench_support/run_legs.py --legs c5b16t 2>/dev/null | show
