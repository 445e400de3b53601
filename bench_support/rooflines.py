"""Physical rooflines: HBM bytes the PMC counters saw (committed profile, profiles/traffic.json) over hipEvent durations of the run."""
from __future__ import annotations

import json
import os
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # same table: what a float4 copy reaches on this part (79 % of the spec)


def design_bytes_project(cloud, cam, T, inl):
    """Bytes k_project_scatter has to move for one frame in THIS design: the cloud once (16 B/point), one 4-byte map
    entry per point that lands in the image in front of the camera, the occupancy words those points set, and the
    inlier-mask words read for them.  (No map clear, no camera-frame copy: DESIGN.md §2.)"""
    xyz = cloud[:, :3].astype(np.float64)
    p = xyz @ T[:, :3].T + T[:, 3]
    with np.errstate(invalid="ignore", divide="ignore"):
        u = (cam.focal_length * p[:, 0] + cam.principal_point_x * p[:, 2]) / p[:, 2]
        v = (cam.focal_length * p[:, 1] + cam.principal_point_y * p[:, 2]) / p[:, 2]
        vis = (p[:, 2] > 0) & (u > 0) & (u < cam.width) & (v > 0) & (v < cam.height)
    idx = np.nonzero(vis)[0]
    words = np.unique((u[idx].astype(np.int64) >> 5) * 100000 + v[idx].astype(np.int64)).size
    mwords = np.unique(idx >> 5).size
    return {"bytes": 16 * cloud.shape[0] + 4 * idx.size + 4 * words + 4 * mwords, "n_front_in_image": int(idx.size),
            "bitmap_words": int(words)}


def pmc_traffic(kernel, frames_per_launch):
    """HBM bytes per launch of `kernel` and the launch time they were measured with (the kernel alone on the GPU), from
    the committed rocprofv3 PMC passes (profiles/traffic.json, written by profiles/summarize.py: FETCH_SIZE/WRITE_SIZE
    in separate --pmc runs, gfx950 correction applied), scaled from the profile's frames per launch to this run's."""
    try:
        t = json.loads((ROOT / "profiles" / "traffic.json").read_text())
        scale = float(frames_per_launch) / float(t["frames_per_launch"])  # traffic is proportional to the frames
        return float(t[kernel]["hbm_bytes_per_launch"]) * scale, float(t[kernel]["launch_s"]) * scale, t.get("source", "")
    except Exception:  # noqa: BLE001
        return None


def config_roofline(key, kt, frames_per_launch):
    """Physical roofline of one BASELINE-config leg: its dominant kernel (longest average launch, hipEvents of THIS run)
    priced on the HBM bytes the PMC counters saw for that kernel in the committed profile of the same leg
    (profiles/traffic.json["configs"][key]: FETCH_SIZE, gfx950-corrected for the projection's wide loads, + WRITE_SIZE,
    separate --pmc passes), scaled to this run's frames per launch.  None where no counter profile is committed."""
    tj = traffic_profile_json() or {}
    prof = (tj.get("configs") or {}).get(key)
    if prof is None and key.startswith("5b"):  # config 5 at another batch size: the S = 256 profile, scaled by the slots
        key = "5b256"
        prof = (tj.get("configs") or {}).get(key)
    times = {k: v["avg_ms"] for k, v in kt.items() if v.get("avg_ms", 0.0) > 0}
    if not prof or not times:
        return None
    scale = float(frames_per_launch) / float(prof["frames_per_launch"])
    per = {}
    for k, t_ms in times.items():
        if k in prof:
            nb = float(prof[k]["hbm_bytes_per_launch"]) * scale
            per[k] = {"kernel_ms": t_ms, "traffic": nb, "achieved": nb / (t_ms * 1e-3) / 1e9,
                      "frac": nb / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "profile_launch_ms": float(prof[k]["launch_s"]) * 1e3 * scale}
    dominant = max(times, key=times.get)
    if dominant not in per:
        return None
    total = sum(v["traffic"] for v in per.values())
    t_all = sum(times.values())
    return {"bound": "hbm", "kernel": dominant, "kernel_ms": times[dominant], "achieved": per[dominant]["achieved"],
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per[dominant]["frac"], "traffic": per[dominant]["traffic"],
            "all_kernels_frac": total / (t_all * 1e-3) / 1e9 / HBM_PEAK_GBS,  # counter bytes of the leg's kernels / their summed launch times
            "kernels": per, "bytes_source": f"profiles/traffic.json configs.{key} ({prof.get('source', '')})"}


def traffic_profile_json():
    try:
        return json.loads((ROOT / "profiles" / "traffic.json").read_text())
    except Exception:  # noqa: BLE001
        return None



def headline_roofline(kt, kt_x, S, B, N, steps, elapsed, n_contexts, design_project, formula_project, formula_feature):
    """(compact, detail) roofline objects of the headline step.

    RULE (fixed): `kernel` = the kernel with the LONGEST AVERAGE LAUNCH in the timed schedule (hipEvents of
    this run, the same quantity `rocprofv3 --kernel-trace --stats` reports as the kernel's average duration in
    profiles/r6_kernel_stats.csv) - no byte-weighted tie-break; the other long kernel rides along as `second`.
    `traffic` = HBM bytes the PMC counters saw per launch of that kernel (FETCH_SIZE with the gfx950 correction +
    WRITE_SIZE, separate --pmc passes; committed in profiles/traffic.json, scaled to this run's frames per launch),
    `achieved` = traffic / kernel_ms, `frac` = achieved / 8 TB/s.  SURVEY 8(d)'s formula bytes are kept in the detail only."""
    tj = traffic_profile_json()

    def ms(k):
        return kt.get(k, {}).get("avg_ms", 0.0)

    def x_ms(k):
        return kt_x.get(k, {}).get("avg_ms", 0.0)

    def gbps(nbytes, t_ms):
        return (nbytes / (t_ms * 1e-3)) / 1e9 if (nbytes and t_ms > 0) else None

    def frac_of(nbytes, t_ms):
        g = gbps(nbytes, t_ms)
        return g / HBM_PEAK_GBS if g is not None else None

    def pmc_bytes(name):
        t = pmc_traffic(name, S)
        return t[0] if t else None

    def entry(name):
        cb = pmc_bytes(name)
        e = {"kernel": name, "kernel_ms": ms(name), "traffic": cb, "achieved": gbps(cb, ms(name)), "frac": frac_of(cb, ms(name))}
        if x_ms(name) > 0:
            e["alone_ms"] = x_ms(name)
            e["frac_alone"] = frac_of(cb, x_ms(name))
        return e

    timed = [k for k in kt if ms(k) > 0 and k != "k_rs_batch"]
    order = sorted(timed, key=ms, reverse=True)
    dominant = order[0] if order else "k_project_scatter"
    second = order[1] if len(order) > 1 else None
    dom = entry(dominant)
    source = "PMC FETCH_SIZE (gfx950-corrected) + WRITE_SIZE per launch, profiles/traffic.json, scaled to this launch size"
    if dom["traffic"] is None and dominant == "k_project_scatter":  # no committed counter profile: the design bytes
        dom.update({"achieved": gbps(design_project, ms(dominant)), "frac": frac_of(design_project, ms(dominant))})
        source = "design bytes (no profiles/traffic.json)"
    step_s = elapsed / steps
    sets_per_step = B // S
    counter_bytes = None
    if tj:
        scale = float(S) / float(tj["frames_per_launch"])
        counter_bytes = sum(float(tj[k]["hbm_bytes_per_launch"]) for k in
                            ("k_project_scatter", "k_classify", "k_feature_fused", "k_feature_wave") if k in tj) * scale * sets_per_step
    compulsory = 16.0 * N * B
    compact = {
        "bound": "hbm", "kernel": dominant, "kernel_ms": dom["kernel_ms"], "traffic": dom["traffic"],
        "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
        "rule": "kernel = longest average launch in the timed schedule (hipEvents on the kernels' streams, this run)",
        "bytes_source": source,
        "frac_alone": dom.get("frac_alone"), "alone_ms": dom.get("alone_ms"),
        "second": entry(second) if second else None,
        "kernels_ms": {k: round(ms(k), 5) for k in kt},                                  # average launch, timed schedule
        "kernels_alone_ms": ({k: round(x_ms(k), 5) for k in kt_x} or None),              # the same launches alone
        "whole_step": {"counter_bytes": counter_bytes, "step_ms": 1e3 * step_s,
                       "frac": (counter_bytes / step_s / 1e9 / HBM_PEAK_GBS) if counter_bytes else None,
                       "compulsory_frac": compulsory / step_s / 1e9 / HBM_PEAK_GBS},
        "concurrent": (f"{n_contexts} contexts: one's projection beside the other's feature kernels" if n_contexts > 1 else None),
    }
    # ---- detail: gather roof of the feature kernel, HBM-busy model, design / formula figures, every kernel
    gather = hbm_busy = None
    fj = (tj or {}).get("k_feature_fused", {})
    ceil = (tj or {}).get("gather_ceilings")
    if ceil and fj.get("tcp_tcc_read_req") and ms("k_feature_fused") > 0:
        scale = float(S) / float(tj["frames_per_launch"])
        l2_req = fj["tcp_tcc_read_req"] * scale           # L1 misses: lines requested from L2
        hbm_req = fj.get("tcc_ea_rdreq", 0.0) * scale      # of those, lines L2 had to fetch from memory (64 B each)
        t_s = ms("k_feature_fused") * 1e-3
        t_floor = max(l2_req - hbm_req, 0.0) / (ceil["l2_Glines_s"] * 1e9) + hbm_req / (ceil["hbm_Glines_s"] * 1e9)
        gather = {"requests_per_launch": l2_req, "memory_fetches_per_launch": hbm_req,
                  "achieved_Glines_s": l2_req / t_s / 1e9, "frac": t_floor / t_s, "ceilings": ceil,
                  "model": "lines requested from L2 (TCP_TCC_READ_REQ) priced at the random-gather rate of L2-resident lines, "
                           "the share fetched from memory (TCC_EA0_RDREQ) at the rate of memory-resident lines"}
    try:
        if tj:
            scale = float(S) / float(tj["frames_per_launch"]) * sets_per_step
            stream_rate = float(tj["k_project_scatter"]["hbm_bytes_per_launch"]) / float(tj["k_project_scatter"]["launch_s"])
            streamed = (float(tj["k_project_scatter"]["hbm_bytes_per_launch"]) + float(tj["k_classify"]["hbm_bytes_per_launch"])) * scale
            lines = float(tj["k_feature_fused"]["tcc_ea_rdreq"]) * scale
            line_rate = float(tj["gather_ceilings"]["hbm_Glines_s"]) * 1e9
            busy_s = streamed / stream_rate + lines / line_rate
            hbm_busy = {"streamed_bytes": streamed, "stream_rate_GBps": stream_rate / 1e9, "random_lines": lines,
                        "random_line_rate_Glines_s": line_rate / 1e9, "busy_ms": 1e3 * busy_s, "frac_of_step": busy_s / step_s}
    except (KeyError, TypeError, ZeroDivisionError):
        hbm_busy = None
    kernels = {k: {**kt.get(k, {}), **entry(k)} for k in kt}
    if "k_project_scatter" in kernels:
        kernels["k_project_scatter"].update({
            "design_bytes_per_launch": design_project, "design_frac": frac_of(design_project, ms("k_project_scatter")),
            "design_model": "16 B/point + 4 B per map entry + occupancy and inlier-mask words touched",
            "formula_bytes_per_launch": formula_project, "formula_frac": frac_of(formula_project, ms("k_project_scatter"))})
    if "k_feature_fused" in kernels:
        kernels["k_feature_fused"].update({"formula_bytes_per_launch": formula_feature, "gather": gather,
                                           "formula_frac": frac_of(formula_feature, ms("k_feature_fused"))})
    detail = {
        **compact, "kernels": kernels,
        "exclusive_kernels_ms": {k: v.get("avg_ms", 0.0) for k, v in kt_x.items()} or None,
        "whole_step": {**compact["whole_step"], "hbm_busy": hbm_busy, "compulsory_bytes": compulsory,
                       "frac_of_copy_rate": (counter_bytes / step_s / 1e9 / HBM_COPY_GBS) if counter_bytes else None,
                       "formula_frac": (formula_project + formula_feature) * sets_per_step / step_s / 1e9 / HBM_PEAK_GBS,
                       "counter_source": (tj or {}).get("source")},
        "formula_note": ("SURVEY 8(d)'s per-unit formula charges a per-frame clear of the 1.86 MB pixel map and a 28 B camera-frame "
                         "copy per visible point that this design never performs, and 4 B per window cell where the kernels scan a "
                         "63 KB occupancy bitmap: formula bytes over measured time may exceed the HBM peak and are no bound here"),
    }
    return compact, detail
