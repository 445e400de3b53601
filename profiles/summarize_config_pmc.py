#!/usr/bin/env python3
"""HBM traffic per launch of one BASELINE-config leg from its rocprofv3 passes (profiles/run_profile.sh):
    gpurun_out/prof_<tag>/leg_<key>_trace   --kernel-trace --stats  (launch durations, kernels alone: one context)
    gpurun_out/prof_<tag>/leg_<key>_fetch   --pmc FETCH_SIZE        (KiB; doubled for k_project_scatter's wide loads,
    gpurun_out/prof_<tag>/leg_<key>_write   --pmc WRITE_SIZE         MI355X_MICROARCH.md, gfx950 correction)
-> profiles/traffic.json["configs"][key] (what bench.py's config_roofline prices) and profiles/<tag>_leg_<key>.md.

usage: summarize_config_pmc.py TAG KEY FRAMES_PER_LAUNCH "command line that was profiled" """
import glob
import json
import os
import sys

import pandas as pd

tag, key, frames, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
src = f"gpurun_out/prof_{tag}"
KERNELS = ("k_project_scatter", "k_classify", "k_feature_fused", "k_feature_wave")


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


ks = pd.read_csv(newest(f"{src}/leg_{key}_trace/*/*kernel_stats.csv"))
ks = ks[ks.Name.str.contains("mld::")]
rows = {}
for d in ("fetch", "write"):
    f_ = newest(f"{src}/leg_{key}_{d}/*/*_counter_collection.csv")
    df = pd.read_csv(f_)
    df = df[df.Kernel_Name.str.contains("|".join(KERNELS))]
    df["k"] = df.Kernel_Name.str.extract(r"(k_\w+)")
    for (k, c), v in df.groupby(["k", "Counter_Name"]).Counter_Value.mean().items():
        rows[(k, c)] = v
lines = [f"# rocprofv3 counters - {tag}, leg `{key}`", "", f"Command (three passes: `--kernel-trace --stats`, `--kernel-trace --pmc FETCH_SIZE`, "
         f"`--kernel-trace --pmc WRITE_SIZE`): `{cmd}`", "",
         "| kernel | calls | avg us (trace pass) | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM bytes per launch | TB/s | of 8 TB/s |",
         "|---|---|---|---|---|---|---|---|"]
entry = {"frames_per_launch": frames,
         "source": f"profiles/{tag}_leg_{key}.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; launch_s from the "
                   "--kernel-trace pass of the same command)"}
for k in KERNELS:
    sel = ks[ks.Name.str.contains(k + r"\b|" + k + "<")]
    if sel.empty or (k, "FETCH_SIZE") not in rows:
        continue
    # several instantiations of a kernel (k_project_scatter<true/false>) fold into one row, weighted by calls
    calls = int(sel.Calls.sum())
    avg = float((sel.AverageNs * sel.Calls).sum() / max(1, calls)) * 1e-9
    f, w = rows[(k, "FETCH_SIZE")] * 1024, rows.get((k, "WRITE_SIZE"), 0.0) * 1024
    fc = 2 * f if k == "k_project_scatter" else f
    entry[k] = {"fetch_bytes_corrected": fc, "write_bytes": w, "hbm_bytes_per_launch": fc + w, "launch_s": avg, "calls": calls}
    lines.append(f"| `{k}` | {calls} | {avg * 1e6:.1f} | {rows[(k, 'FETCH_SIZE')]:,.0f} | {rows.get((k, 'WRITE_SIZE'), 0.0):,.0f} | "
                 f"{(fc + w) / 1e6:,.1f} MB | {(fc + w) / avg / 1e12:.2f} | {(fc + w) / avg / 8e12:.3f} |")
lines += ["", "FETCH_SIZE is doubled for `k_project_scatter` (gfx950 reports half the bytes of its 16 B / lane coalesced loads, "
          "MI355X_MICROARCH.md); the feature kernels' 4-byte / 16-byte gathers are left as reported.", ""]
try:
    b = json.loads(open(f"{src}/leg_{key}_trace.json").read().strip().splitlines()[-1])
    lines += ["bench.py object of the trace pass:", "", "```json", json.dumps(b, indent=1)[:6000], "```", ""]
except Exception as e:  # noqa: BLE001
    lines += [f"(bench object of the trace pass not parsed: {e})", ""]
tj = json.load(open("profiles/traffic.json"))
tj.setdefault("configs", {})[key] = entry
json.dump(tj, open("profiles/traffic.json", "w"), indent=1)
open(f"profiles/{tag}_leg_{key}.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:14]))
