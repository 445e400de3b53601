#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by profiles/run_profile.sh) into profiles/<tag>_*.{csv,md}."""
import glob
import json
import shutil
import sys

import pandas as pd

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = f"gpurun_out/prof_{tag}"
import os


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


stats = newest(f"{src}/trace/*/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
lines = [f"# rocprofv3 summary — {tag}", "",
         "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 "
         "--cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0` (see profiles/run_profile.sh); PMC "
         "passes are separate runs of the same command with `--pmc` (4 steps).", ""]
ks = pd.read_csv(stats)
ks = ks[ks.Name.str.contains("mld::")]
lines += ["## kernel-trace --stats", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for _, r in ks.iterrows():
    lines.append(f"| `{r.Name.split('(')[0]}` | {r.Calls} | {r.AverageNs / 1e3:.1f} | {r.MinNs / 1e3:.1f} | "
                 f"{r.MaxNs / 1e3:.1f} | {r.Percentage:.2f} |")
try:
    b = json.loads(open(f"{src}/bench_trace.json").read().strip().splitlines()[-1])
    rk = b["roofline"]["kernels"]
    lines += ["", "bench.py hipEvent averages in the same run: " +
              ", ".join(f"`{k}` {v.get('avg_ms', 0) * 1e3:.1f} us" for k, v in rk.items()), ""]
    lines += [f"bench.py line of that run: value {b['value'] / 1e9:.3f} G associations/s, ms_per_step {b['ms_per_step']:.4f}, "
              f"verified {b['verified']}, roofline.frac {b['roofline']['frac']:.3f} ({b['roofline']['kernel']})", ""]
except Exception as e:  # noqa: BLE001
    lines += ["", f"(bench json not parsed: {e})", ""]
lines += ["## PMC (per launch, mean over launches)", "",
          "FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) "
          "coalesced read (MI355X_MICROARCH.md §HBM): `hbm_read_corrected` doubles it for k_project_scatter "
          "(float4 cloud loads); the feature kernel's 4-B map reads / 16-B gathers are uncalibrated and left as is.", "",
          "| kernel | counter | value |", "|---|---|---|"]
rows = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f_ = newest(f"{src}/{d}/*/*_counter_collection.csv")
    if not f_:
        continue
    df = pd.read_csv(f_)
    df = df[df.Kernel_Name.str.contains("k_project_scatter|k_classify|k_feature_fused|k_feature_wave|k_rs_batch")]
    df["k"] = df.Kernel_Name.str.extract(r"(k_\w+)")
    g = df.groupby(["k", "Counter_Name"]).Counter_Value.mean()
    for (k, c), v in g.items():
        rows[(k, c)] = v
        lines.append(f"| `{k}` | {c} | {v:,.1f} |")
lines += ["", "## derived", ""]
traffic = {}
for k in ("k_project_scatter", "k_classify", "k_feature_fused"):
    if (k, "FETCH_SIZE") in rows and (k, "WRITE_SIZE") in rows:
        f, w = rows[(k, "FETCH_SIZE")] * 1024, rows[(k, "WRITE_SIZE")] * 1024
        fc = 2 * f if k == "k_project_scatter" else f
        avg = float(ks[ks.Name.str.contains(k)].AverageNs.iloc[0]) * 1e-9
        traffic[k] = {"fetch_bytes_corrected": fc, "write_bytes": w, "hbm_bytes_per_launch": fc + w, "launch_s": avg}
        lines.append(f"* `{k}`: HBM traffic per launch = {fc / 1e6:,.1f} MB read (corrected) + {w / 1e6:,.1f} MB "
                     f"written = {(fc + w) / 1e6:,.1f} MB -> {(fc + w) / avg / 1e12:.2f} TB/s over the {avg * 1e6:.1f} us launch")
try:
    traffic["frames_per_launch"] = b["config"].get("frame_slots_per_launch", b["config"]["frames_per_step"])
    traffic["source"] = f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print("traffic.json not written:", e)
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
