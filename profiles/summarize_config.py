#!/usr/bin/env python3
"""profiles/<tag>_config<N>_summary.md from gpurun_out/prof_<tag>/trace_c<N> (rocprofv3 --kernel-trace --stats of
`python3 bench_support/run_legs.py --legs cN`) and that run's JSON object."""
import glob
import json
import os
import sys

import pandas as pd

tag, cfg = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
fs = sorted(glob.glob(f"{src}/trace_c{cfg}/*/*kernel_stats.csv"), key=os.path.getmtime)
ks = pd.read_csv(fs[-1])
ks = ks[ks.Name.str.contains("mld::")]
b = json.loads(open(f"{src}/bench_c{cfg}.json").read().strip().splitlines()[-1])
b = b.get("configs", {}).get(cfg, b)  # (bench_support/run_legs.py prints {"configs": {cfg: leg}, ...})
lines = [f"# rocprofv3 summary - {tag}, BASELINE config {cfg}", "",
         f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench_support/run_legs.py --legs c{cfg}`", "",
         b["workload"], "",
         "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for _, r in ks.iterrows():
    lines.append(f"| `{r.Name.split('(')[0]}` | {r.Calls} | {r.AverageNs / 1e3:.1f} | {r.MinNs / 1e3:.1f} | "
                 f"{r.MaxNs / 1e3:.1f} | {r.Percentage:.2f} |")
lines += ["", "bench.py object of the same run:", "", "```json", json.dumps(b, indent=1), "```", ""]
open(f"profiles/{tag}_config{cfg}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:20]))
