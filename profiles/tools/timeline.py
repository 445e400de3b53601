#!/usr/bin/env python3
"""Start / end of the last kernels in a rocprofv3 --kernel-trace CSV (who runs beside whom)."""
import glob
import os
import sys

import pandas as pd

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = sorted(glob.glob(f"{d}/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
df = pd.read_csv(f)
df = df[df.Kernel_Name.str.contains("mld::")].sort_values("Start_Timestamp").tail(n)
t0 = df.Start_Timestamp.min()
for _, r in df.iterrows():
    name = r.Kernel_Name.split("(")[0].replace("mld::", "").replace("void ", "")
    print(f"{name:28s} q{r.Queue_Id:<3} start {(r.Start_Timestamp - t0) / 1e3:9.1f} us  dur {(r.End_Timestamp - r.Start_Timestamp) / 1e3:8.1f} us  "
          f"end {(r.End_Timestamp - t0) / 1e3:9.1f}")
