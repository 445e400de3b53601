#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by profiles/run_profile.sh) into profiles/<tag>_*.{csv,md}."""
import glob
import json
import shutil
import sys

import pandas as pd

tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
src = f"gpurun_out/prof_{tag}"
import os


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


stats = newest(f"{src}/trace/*/*kernel_stats.csv")
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
stats_x = newest(f"{src}/trace_x/*/*kernel_stats.csv")
if stats_x:
    shutil.copy(stats_x, f"profiles/{tag}_kernel_stats_one_context.csv")
lines = [f"# rocprofv3 summary — {tag}", "",
         "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 5 --repeats 1 "
         "--cpu-seconds 0 --latency-frames 0 --streaming-batches 0 --config-frames 0` (see profiles/run_profile.sh): the "
         "bench default (with `--no-estimated --no-exclusive`, so that only launches of the timed schedule are averaged), "
         "two contexts of 1024 frame slots, consecutive steps alternating between them, so the projection "
         "of one step runs BESIDE the previous step's feature kernels and the launch durations below are durations of "
         "kernels that share the GPU.  The second table "
         "is the same command with `--contexts 1` (one stream, 1024 frame slots per launch, every kernel alone).  PMC "
         "passes are separate runs with `--contexts 1 --pmc ...` (4 steps; rocprofv3 serialises the kernels there).", ""]


def table(path, title, bjson):
    out = []
    k = pd.read_csv(path)
    k = k[k.Name.str.contains("mld::")]
    out += [f"## kernel-trace --stats: {title}", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for _, r in k.iterrows():
        out.append(f"| `{r.Name.split('(')[0]}` | {r.Calls} | {r.AverageNs / 1e3:.1f} | {r.MinNs / 1e3:.1f} | "
                   f"{r.MaxNs / 1e3:.1f} | {r.Percentage:.2f} |")
    bb = None
    try:
        bb = json.loads(open(bjson).read().strip().splitlines()[-1])
        rk = bb["roofline"]["kernels"]
        out += ["", "bench.py hipEvent averages in the same run: " +
                ", ".join(f"`{kk}` {v.get('avg_ms', 0) * 1e3:.1f} us" for kk, v in rk.items()), ""]
        rf = bb["roofline"]
        out += [f"bench.py line of that run: value {bb['value'] / 1e9:.3f} G associations/s, ms_per_step {bb['ms_per_step']:.4f}, "
                f"verified {bb['verified']}, roofline.frac {rf['frac']:.3f} ({rf['kernel']}, {rf['kernel_ms'] * 1e3:.1f} us, "
                f"{bb['config']['frame_slots_per_launch']} frames per launch)" +
                (f", roofline.second.frac {rf['second']['frac']:.3f} ({rf['second']['kernel']}, {rf['second']['kernel_ms'] * 1e3:.1f} us)"
                 if rf.get("second") else "") +
                (f", roofline.frac_alone {rf['frac_alone']:.3f} ({rf['alone_ms'] * 1e3:.1f} us alone)" if rf.get("alone_ms") else ""), ""]
    except Exception as e:  # noqa: BLE001
        out += ["", f"(bench json not parsed: {e})", ""]
    return k, bb, out


ks2, b2, t2 = table(stats, "two contexts, steps alternating (bench default)", f"{src}/bench_trace.json")
lines += t2
if stats_x:
    ks, b, t1 = table(stats_x, "one context (`--contexts 1`)", f"{src}/bench_trace_x.json")
    lines += t1
else:
    ks, b = ks2, b2
stats_e = newest(f"{src}/trace_e/*/*kernel_stats.csv")
if stats_e:
    ke = pd.read_csv(stats_e)
    ke = ke[ke.Name.str.contains("k_rs_batch")]
    try:
        be = json.loads(open(f"{src}/bench_trace_e.json").read().strip().splitlines()[-1])["plane_estimated"]
        lines += ["## plane-estimated leg (same command with 2 steps; `k_rs_batch` only)", ""]
        for _, r in ke.iterrows():
            lines.append(f"`{r.Name.split('(')[0]}`: {r.Calls} calls, avg {r.AverageNs / 1e3:.1f} us "
                         f"({be['frame_slots_per_launch']} frames per launch); bench.py: {be['ms_per_step']:.3f} ms per step, "
                         f"{be['ransac_us_per_frame']:.2f} us of RANSAC per frame, verified {be['verified']}")
        lines.append("")
    except Exception as e:  # noqa: BLE001
        lines += [f"(plane-estimated leg not parsed: {e})", ""]
stats_l = newest(f"{src}/trace_l/*/*kernel_stats.csv")
if stats_l:
    kl = pd.read_csv(stats_l)
    kl = kl[kl.Name.str.contains("mld::")]
    lines += ["## one frame per call (latency legs: supplied plane, RANSAC, semantic; `--latency-frames 100`)", "",
              "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
    for _, r in kl.iterrows():
        lines.append(f"| `{r.Name.split('(')[0]}` | {r.Calls} | {r.AverageNs / 1e3:.1f} | {r.MinNs / 1e3:.1f} | {r.MaxNs / 1e3:.1f} |")
    try:
        bl = json.loads(open(f"{src}/bench_trace_l.json").read().strip().splitlines()[-1])["latency"]
        for name, leg in (("supplied", bl), ("estimated.ransac", bl["estimated"]["ransac"]), ("estimated.semantic", bl["estimated"]["semantic"])):
            lines.append("")
            lines.append(f"bench.py `latency.{name}` (under the tracer): median {leg['ms_per_frame_median'] * 1e3:.1f} us, p99 "
                         f"{leg['ms_per_frame_p99'] * 1e3:.1f} us; breakdown (instrumented pass) " +
                         ", ".join(f"{k} {v:.1f}" for k, v in (leg.get('breakdown_us_median') or {}).items()))
    except Exception as e:  # noqa: BLE001
        lines += ["", f"(latency legs not parsed: {e})"]
    lines.append("")
    try:
        bfull = json.loads(open(f"{src}/bench_latency.json").read().strip().splitlines()[-1])
        bl = bfull["latency"]
        st, cb = bfull.get("streaming"), None
        try:  # (the cpu_baseline of the un-traced default run of the same session)
            cb = json.loads(open(f"{src}/bench_default_detail.json").read())["cpu_baseline"]
        except Exception:  # noqa: BLE001
            pass
        if st:
            lines += [f"PCIe-inclusive throughput (`streaming`: {st['frames_per_batch']}-frame batches from pinned host memory, double-buffered, "
                      f"results copied back): {st['frames_per_s'] / 1e3:.1f} k frames/s = {st['associations_per_s'] / 1e6:.1f} M associations/s, "
                      f"{st['h2d_GBps']:.1f} GB/s host to device.", ""]
        if cb:
            lines += [f"`cpu_baseline` (the restated reference path on this host, kind `{cb['kind']}`, {cb['cores']} threads): "
                      f"{cb['value'] / 1e6:.2f} M associations/s, {cb['ms_per_frame']:.2f} ms per frame (stage A {cb['stage_a_ms']:.2f} ms serial).", ""]
        lines += ["Without the tracer (same box, same session; `bench.py --latency-frames 200`, 200 calls per leg) - the numbers of "
                  "record for the one-frame calls:", "",
                  "| leg | median us | p99 us | host phases of the un-instrumented call (median us) | GPU phases of the instrumented pass (median us) |",
                  "|---|---|---|---|---|"]
        legs = [("supplied plane (`mld_calculate_depth_frame`)", bl), ("supplied plane, cloud in pinned host memory", bl.get("pinned_source") or {}),
                ("plane estimated in the call, RANSAC (`mld_calculate_depth_frame_estimate`)", bl["estimated"]["ransac"]),
                ("plane estimated in the call, semantic label image", bl["estimated"]["semantic"])]
        for name, leg in legs:
            if "ms_per_frame_median" not in leg:
                continue
            hp = ", ".join(f"{k[:-3]} {v:.1f}" for k, v in (leg.get("host_us_median") or {}).items())
            gp = ", ".join(f"{k[:-3]} {v:.1f}" for k, v in (leg.get("breakdown_us_median") or {}).items()
                           if k in ("h2d_us", "plane_us", "kernels_us", "d2h_us", "gpu_us"))
            lines.append(f"| {name} | {leg['ms_per_frame_median'] * 1e3:.1f} | {leg['ms_per_frame_p99'] * 1e3:.1f} | {hp} | {gp} |")
        lines.append("")
        pr = bl.get("process")
        if pr:
            lines += ["`latency.process` - `TrackletDepthModule.process` per frame (host cloud, fresh un-segmented SemanticPlane, "
                      f"{pr.get('tracks', 2000)} tracks), time inside the C-ABI per frame: one call (`mld_tracklets_frame`) median "
                      f"{pr['one_call']['ms_per_frame_median'] * 1e3:.1f} us / p99 {pr['one_call']['ms_per_frame_p99'] * 1e3:.1f} us; "
                      f"three calls (set cloud, estimate plane, tracklet depths) median {pr['two_calls']['ms_per_frame_median'] * 1e3:.1f} us / "
                      f"p99 {pr['two_calls']['ms_per_frame_p99'] * 1e3:.1f} us.", ""]
    except Exception as e:  # noqa: BLE001
        lines += [f"(un-traced latency legs not parsed: {e})", ""]
lines += ["## PMC (per launch, mean over launches)", "",
          "FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide (16 B/lane) "
          "coalesced read (MI355X_MICROARCH.md §HBM): `hbm_read_corrected` doubles it for k_project_scatter "
          "(float4 cloud loads); the feature kernel's 4-B map reads / 16-B gathers are uncalibrated and left as is.", "",
          "| kernel | counter | value |", "|---|---|---|"]
rows = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcp", "pmc_tcc"):
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
# gather roof of the lane-per-feature kernel: the cache-line requests of its divergent loads (TCP = per-CU L1, TCC = L2)
# against the line rates random gathers reach on this part (profiles/tools/randgather.hip, run in the same session)
fk = "k_feature_fused"
if fk in traffic and (fk, "TCP_TCC_READ_REQ_sum") in rows:
    traffic[fk]["tcp_total_cache_accesses"] = rows.get((fk, "TCP_TOTAL_CACHE_ACCESSES_sum"), 0.0)
    traffic[fk]["tcp_tcc_read_req"] = rows[(fk, "TCP_TCC_READ_REQ_sum")]
    traffic[fk]["tcc_ea_rdreq"] = rows.get((fk, "TCC_EA0_RDREQ_sum"), 0.0)
    traffic[fk]["tcc_hit"] = rows.get((fk, "TCC_HIT_sum"), 0.0)
    traffic[fk]["tcc_miss"] = rows.get((fk, "TCC_MISS_sum"), 0.0)
rg = f"{src}/randgather.txt"
if os.path.exists(rg):
    import re
    shutil.copy(rg, f"profiles/{tag}_randgather.txt")
    rates = {}
    for ln in open(rg):
        m = re.match(r"buffer\s+(\d+) KiB: 4B x4 ([\d.]+) G/s .*\| 4B x8 ([\d.]+) G/s \| 16B x4 ([\d.]+) G/s", ln)
        if m:
            rates[int(m.group(1))] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    if rates:
        pick = lambda kb: max(rates[kb][0], rates[kb][1])  # noqa: E731  best 4-byte gather rate at that footprint
        traffic["gather_ceilings"] = {"l1_Glines_s": pick(16), "l2_Glines_s": pick(2048), "mall_Glines_s": pick(65536),
                                      "hbm_Glines_s": pick(4194304),
                                      "source": f"profiles/{tag}_randgather.txt (random 64-B-line gathers, every CU issuing; "
                                                "buffer 16 KiB / 2 MiB / 64 MiB / 4 GiB)"}
        lines += ["", "## gather roof of `k_feature_fused`", "",
                  f"random-gather line rates (profiles/{tag}_randgather.txt): L1 {pick(16):.1f}, L2 {pick(2048):.1f}, "
                  f"Infinity Cache {pick(65536):.1f}, HBM {pick(4194304):.1f} G lines/s"]
        if "tcp_tcc_read_req" in traffic.get(fk, {}):
            t = traffic[fk]
            l2r, ear = t["tcp_tcc_read_req"], t["tcc_ea_rdreq"]
            floor = max(l2r - ear, 0) / (pick(2048) * 1e9) + ear / (pick(4194304) * 1e9)
            lines.append(f"* `{fk}` per launch: {t['tcp_total_cache_accesses'] / 1e6:.1f} M L1 line accesses, {l2r / 1e6:.1f} M "
                         f"requests to L2, {ear / 1e6:.1f} M fetched from memory -> service time at those rates "
                         f"{floor * 1e6:.1f} us = {floor / t['launch_s']:.2f} of the {t['launch_s'] * 1e6:.1f} us launch "
                         f"({l2r / t['launch_s'] / 1e9:.1f} G lines/s achieved)")
# ---- the roofline of record, reproducible from this file: RULE = the kernel with the longest average launch in the
# bench-default schedule (first table); bytes = the counters above; frac = bytes / that average / 8 TB/s
try:
    long2 = ks2[ks2.Name.str.contains("k_project_scatter|k_classify|k_feature_fused|k_feature_wave")].sort_values("AverageNs", ascending=False)
    lines += ["", "## roofline of record (rule: longest average launch of the bench-default schedule, first table)", "",
              "| rank | kernel | avg us (two contexts) | counter bytes per launch | TB/s | of 8 TB/s |", "|---|---|---|---|---|---|"]
    for i, (_, r) in enumerate(long2.iterrows()):
        kn = next((k for k in traffic if k in r.Name), None)
        if kn is None or i > 1:
            continue
        nb, t = traffic[kn]["hbm_bytes_per_launch"], r.AverageNs * 1e-9
        lines.append(f"| {'roofline.kernel' if i == 0 else 'roofline.second'} | `{kn}` | {t * 1e6:.1f} | {nb / 1e6:,.1f} MB | "
                     f"{nb / t / 1e12:.3f} | **{nb / t / 8e12:.3f}** |")
    lines += ["", "(`bench.py` prices the same counter bytes - `profiles/traffic.json` - on the hipEvent duration of ITS run: "
              "`roofline.frac = roofline.traffic / roofline.kernel_ms / 8 TB/s`.)"]
except Exception as e:  # noqa: BLE001
    lines += ["", f"(roofline of record not derived: {e})"]
try:
    traffic["frames_per_launch"] = b["config"].get("frame_slots_per_launch", b["config"]["frames_per_step"])
    traffic["source"] = (f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCP_* / TCC_*, separate passes, one "
                         "context; launch_s = that kernel alone, from the --contexts 1 trace)")
    try:  # the per-leg entries (summarize_config_pmc.py) live in the same file: keep them
        traffic["configs"] = json.load(open("profiles/traffic.json")).get("configs", {})
    except Exception:  # noqa: BLE001
        pass
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
except Exception as e:  # noqa: BLE001
    print("traffic.json not written:", e)
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
