#!/usr/bin/env python3
"""bench.py -- ORB kfeatures/sec, extract + match (BASELINE.json metric), on 1..N MI355X of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload euroc|kitti|tumvi]

A "step" = one pass of the hot path over one batch of synthetic frames of ONE camera sequence per GPU:
  euroc (default, the metric's configuration): 256 frames 752x480, nFeatures=1000 -- extract (pyramid, FAST, quad-tree,
        orientation, blur, rBRIEF) + frame-to-frame SearchByProjection (th=15, rotation check) between consecutive frames
        + D2H of keypoints, descriptors, counts and match indices into pinned host memory;
  kitti (BASELINE config 3): 128 rectified stereo pairs 1241x376, nFeatures=2000 -- left + right extraction +
        Frame::ComputeStereoMatches (row-band Hamming, SAD sub-pixel, median rejection) on the device + D2H;
  tumvi (BASELINE config 4): 128 frames 1024x1024, nFeatures=1500 -- extract + SearchByProjection(Frame, MapPoints) against
        10,000 map-point descriptors per frame (Tracking.cc:3390-3413) + D2H.
`value` follows the bench contract: inputs are resident in HBM when the timed region starts.  The same run also measures the
host-input rate (`pcie_inclusive`: frames start in pinned host memory, orbx_extract_batch_host uploads batch i+1 while batch
i computes) -- SURVEY.md 8d's "wall clock covering H2D ... D2H".

Processes: with WORLD_SIZE unset this file is a LAUNCHER: it starts one rank process per GPU itself (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment) and relays rank 0's JSON line; under
torch.distributed.run it is a rank directly (and --gpus must equal WORLD_SIZE).  Ranks shard independent camera sequences
(seed = 10 + rank), no data-path collective; torch.distributed over GLOO (host tensors) carries only the timing
barrier and the gather of every rank's (seconds, features) -- no RCCL communicator is created on the GPUs (north_star: "RCCL not
required").  After the timed loop rank 0 checks EVERY frame and EVERY match vector of the last timed step (keypoints, descriptors, match
indices) against the CPU oracle: "parity_checked" in the JSON line; a mismatch is a failed run.  The K timed steps are then repeated
(`repeats`: median / min / max over the regions), the drop-in call is timed one frame at a time (`latency`), and at --gpus 1 the other two
BASELINE configurations (KITTI stereo, TUM-VI map-point search) run as child processes of the same file (`other_workloads`).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

NLEVELS = 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (width, height, nfeatures, default frames per step, lapping area)
    "euroc": (752, 480, 1000, 256, (0, 1000)),
    "kitti": (1241, 376, 2000, 128, (0, 0)),      # 128 pairs
    "tumvi": (1024, 1024, 1500, 128, (0, 1000)),  # 128 frames
}
N_MAPPOINTS = 10000


# ---------------------------------------------------------------------------------------------------------
# algorithmic bytes (SURVEY.md 8d / DESIGN.md 4): every stage reads its input once and writes its output once
# ---------------------------------------------------------------------------------------------------------
def algorithmic_bytes(sizes, n_frames, feats, cands):
    """Per-kernel algorithmic bytes of ONE launch over n_frames frames.  P = sum of level pixels, N = keypoints out,
    C = FAST candidates (both totals over the launch)."""
    px = [a * b for a, b in sizes]
    P = sum(px)
    N, Cn = feats, cands
    per = {
        "k_pyr_base": 2 * px[0] * n_frames,                       # image read + level-0 write
        "k_pyr_resize": ((P - px[-1]) + (P - px[0])) * n_frames,  # the whole chain; divided by its launches per step in roofline_from_profile
        "k_fast_strip": P * n_frames + 12 * Cn,                     # FAST stage (k_fast_strip + its list pass k_fast_wave_list): pyramid read + candidates at SURVEY.md
                                                                    # 8(d)'s 12 bytes (x, y, score) -- the kernels pack a candidate into 4 bytes; through round 3 the line used 4 * C
        "k_blur": 2 * P * n_frames,
        "k_octree": 8 * Cn + 4 * N,                               # candidates read + gathered, keypoints out
        "k_finalize": 16 * N,
        "k_describe": N * (749 + 512 + 32 + 28),
        "k_window_best2": (32 + 28) * 2 * N + 16 * N,
        "k_greedy_resolve": 16 * N + 4 * N,
    }
    extract_total = (5 * P - px[0] - px[-1]) * n_frames + 12 * Cn + 1321 * N  # B_extract of SURVEY.md 8d
    return per, extract_total


def as_executed_bytes(sizes, n_frames, feats, cands, ran):
    """Algorithmic bytes of the extraction AS THE DEFAULT PATH EXECUTES IT (round 5), for the kernels in `ran`: SURVEY.md 8(d)'s B_extract charges
    a blurred copy of the pyramid (2 P) and a resize chain that re-reads every level; k_describe_fused neither reads nor writes a blurred pyramid
    (it reads the 43 x 43 raw window of a keypoint), k_pyr_stream reads the frame once and writes levels 1 .. 7 once, and with level 0 in place
    there is no level-0 copy.  Candidates stay at SURVEY's 12 bytes."""
    px = [a * b for a, b in sizes]
    P = sum(px)
    b = 0
    if "k_pyr_base" in ran:
        b += 2 * px[0] * n_frames
    b += (px[0] + (P - px[0])) * n_frames if "k_pyr_resize" in ran else 0      # frame in, levels 1 .. 7 out
    b += P * n_frames + 12 * cands                                              # FAST
    b += 8 * cands + 4 * feats + 16 * feats                                     # quad-tree + finalize
    b += (2 * P * n_frames + feats * 1321) if "k_blur" in ran else feats * (43 * 43 + 32 + 28)
    return b


def roofline_from_profile(ex, prof, passes, sizes, n_frames, n_feat, n_cand, launches_scale=1.0):
    """roofline object for the kernel with the largest share of GPU time; prof = {kernel: (avg_ms, launches)}."""
    per, extract_total = algorithmic_bytes(sizes, n_frames, n_feat, n_cand)
    kernels, tot_ms = {}, 0.0
    for name, (ms, cnt) in prof.items():
        if cnt == 0 or name not in per:
            continue
        lps = cnt / float(passes)
        if name == "k_pyr_resize":
            per[name] = per[name] / max(lps, 1.0)   # algorithmic bytes of an average launch of the chain
        kernels[name] = {"avg_ms": round(ms, 4), "launches_per_step": lps, "alg_GBs": round(per[name] / (ms * 1e-3) / 1e9, 1)}
        tot_ms += ms * lps
    dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
    ach = per[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
    ext_ms = sum(v["avg_ms"] * v["launches_per_step"] for k, v in kernels.items() if not k.startswith(("k_window", "k_greedy")))
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": int(per[dom]), "avg_launch_ms": kernels[dom]["avg_ms"],
                "timing": "HIP events on the extractor's own stream, serialized passes after the timed loop",
                "kernel_time_share": round(kernels[dom]["avg_ms"] * kernels[dom]["launches_per_step"] / tot_ms, 3),
                "extract_all_kernels_GBs": round(extract_total / (ext_ms * 1e-3) / 1e9, 1),
                "extract_all_kernels_frac": round(extract_total / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    exec_bytes = as_executed_bytes(sizes, n_frames, n_feat, n_cand, set(kernels))
    roofline["extract_all_kernels_note"] = ("extract_all_kernels_* divide SURVEY.md 8(d)'s B_extract (which charges a blurred pyramid copy and a "
                                            "level-by-level resize chain) by the serialized kernel time; extract_as_executed_* divide the bytes the "
                                            "kernels that actually ran are asked to move (as_executed_bytes in bench.py)")
    roofline["extract_as_executed_bytes"] = int(exec_bytes)
    roofline["extract_as_executed_GBs"] = round(exec_bytes / (ext_ms * 1e-3) / 1e9, 1)
    roofline["extract_as_executed_frac"] = round(exec_bytes / (ext_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return roofline, kernels


# ---------------------------------------------------------------------------------------------------------
# PMC traffic of the dominant kernel, measured IN THIS RUN (default at --gpus 1; --no-pmc skips it): two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE:
# they do not fit one pass) over a short serialized child run of this same file, parsed from rocprofv3's database
# ---------------------------------------------------------------------------------------------------------
# a profile slot may cover more than one kernel: the FAST stage = first pass for every cell + the list pass over the cells it left
STAGE_KERNELS = {"k_fast_strip": ("k_fast_strip", "k_fast_wave_list", "k_fast_cells"), "k_octree": ("k_octree_par_t", "k_octree_rest", "k_octree"),
                 "k_window_best2": ("k_grid_build", "k_window_best2")}


def pmc_traffic(kernel, workload, batch):
    """HBM-side bytes per launch of a stage, measured in this run: two rocprofv3 --pmc child passes of this same file.
    Reads: 32 * TCC_EA0_RDREQ_32B + 64 * TCC_EA0_RDREQ_64B + 128 * TCC_EA0_RDREQ_128B (the request-size classes; on gfx950 practically
    every read request is a 128-byte one, which the derived FETCH_SIZE counter tallies at 64 bytes -- it under-reports every kernel of this
    path by 2x).  Writes: WRITE_SIZE (32 / 64-byte write requests; agrees with known store volumes)."""
    import re
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    group = STAGE_KERNELS.get(kernel, (kernel,))
    out = {}
    passes = {"read": ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"], "write": ["WRITE_SIZE"]}
    weight = {"TCC_EA0_RDREQ_32B_sum": 32.0, "TCC_EA0_RDREQ_64B_sum": 64.0, "TCC_EA0_RDREQ_128B_sum": 128.0, "WRITE_SIZE": 1024.0}
    for name, counters in passes.items():
        td = tempfile.mkdtemp(prefix="orbx_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *counters, "-d", td, "-o", "p", "--", sys.executable, str(ROOT / "bench.py"), "--pmc-child", "--workload", workload,
               "--batch", str(batch), "--steps", "2", "--warmup", "1"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=240)
        except subprocess.TimeoutExpired:
            shutil.rmtree(td, ignore_errors=True)
            return None, f"rocprofv3 --pmc {' '.join(counters)} timed out"
        dbs = list(Path(td).rglob("*.db"))
        if r.returncode != 0 or not dbs:
            shutil.rmtree(td, ignore_errors=True)
            return None, f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode})"
        c = sqlite3.connect(str(dbs[0]))
        tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
        view = "counters_collection" if "counters_collection" in tabs else next((t for t in tabs if t.startswith("counters_collection")), None)
        if view is None:
            return None, "no counters_collection view in the rocprofv3 database"
        per = {k: [0.0, 0] for k in group}   # bytes summed over counters, dispatches
        seen = {k: set() for k in group}
        for kname, cname, val, did in c.execute(f"select kernel_name, counter_name, value, dispatch_id from {view}"):
            k = re.sub(r"<.*>", "", kname.split("(")[0].replace("void ", "").replace("orbx::", ""))
            if k in per and cname in weight:
                per[k][0] += val * weight[cname]
                seen[k].add(did)
        c.close()
        shutil.rmtree(td, ignore_errors=True)
        if not any(seen.values()):
            return None, f"kernel {kernel} not in the {name} pass"
        out[name] = sum(per[k][0] / len(seen[k]) for k in group if seen[k])   # per launch, summed over the stage's kernels
    return out, None


SQ_CLOCK_GHZ = 2.4        # MI355X peak engine clock (MI355X_MICROARCH.md); the measured clock of the pass is reported beside it
N_SIMDS = 1024            # 256 CUs x 4 SIMDs
N_XCD = 8
VALU_ISSUE_CYCLES = 4     # a wave64 VALU instruction of these integer / packed kernels occupies its SIMD's vector issue for one quad-cycle:
                          # SQ_ACTIVE_INST_VALU (quad-cycles) == SQ_INSTS_VALU to 1 % on every kernel of the path (profiles/r05_g_pmc_sq_counters.csv)


def pmc_sq(workload, batch):
    """VALU-issue side of the roofline, measured in this run (round 6): ONE rocprofv3 --pmc child pass with SQ counters (+ --kernel-trace for the
    dispatch durations of the same pass, + GRBM_GUI_ACTIVE for the clock the pass ran at, when the counter exists).  Per kernel, averaged per launch:
    SQ_INSTS_VALU (wave-instructions), SQ_WAVES, SQ_ACTIVE_INST_VALU, the launch duration under the counters, and the issue time
    SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / clock -- the floor of a kernel whose every SIMD issues a vector instruction whenever it can."""
    import re
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    base = ["SQ_INSTS_VALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY"]
    err = None
    for counters in (base + ["GRBM_GUI_ACTIVE"], base):
        td = tempfile.mkdtemp(prefix="orbx_sq_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", *counters, "-d", td, "-o", "p", "--", sys.executable, str(ROOT / "bench.py"), "--pmc-child", "--workload", workload,
               "--batch", str(batch), "--steps", "2", "--warmup", "1"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=240)
        except subprocess.TimeoutExpired:
            shutil.rmtree(td, ignore_errors=True)
            err = "rocprofv3 SQ pass timed out"
            continue
        dbs = list(Path(td).rglob("*.db"))
        if r.returncode != 0 or not dbs:
            shutil.rmtree(td, ignore_errors=True)
            err = f"rocprofv3 SQ pass failed (rc {r.returncode})"
            continue
        try:
            c = sqlite3.connect(str(dbs[0]))
            tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')")]
            view = "counters_collection" if "counters_collection" in tabs else next((t for t in tabs if t.startswith("counters_collection")), None)
            if view is None:
                raise RuntimeError("no counters_collection view in the rocprofv3 database")

            def short(kname):
                return re.sub(r"<.*>", "", kname.split("(")[0].replace("void ", "").replace("orbx::", ""))
            acc = {}   # kernel -> counter -> [sum, set(dispatches)]
            for kname, cname, val, did in c.execute(f"select kernel_name, counter_name, value, dispatch_id from {view}"):
                k = short(kname)
                if not k.startswith("k_"):
                    continue
                a_ = acc.setdefault(k, {}).setdefault(cname, [0.0, set()])
                a_[0] += val
                a_[1].add(did)
            dur = {}
            kview = "kernels" if "kernels" in tabs else next((t for t in tabs if t.startswith("kernels")), None)
            if kview:
                for kname, st, en in c.execute(f"select name, start, end from {kview}"):
                    d_ = dur.setdefault(short(kname), [0.0, 0])
                    d_[0] += (en - st) * 1e-3
                    d_[1] += 1
            c.close()
        except Exception as e:  # noqa: BLE001
            shutil.rmtree(td, ignore_errors=True)
            err = f"rocprofv3 SQ pass: {e}"
            continue
        shutil.rmtree(td, ignore_errors=True)
        out = {}
        for k, cs in acc.items():
            avg = {cn: v[0] / max(len(v[1]), 1) for cn, v in cs.items()}
            if "SQ_INSTS_VALU" not in avg:
                continue
            e = {"valu_insts": int(avg["SQ_INSTS_VALU"]), "waves": int(avg.get("SQ_WAVES", 0)), "launches_in_pass": len(cs["SQ_INSTS_VALU"][1]),
                 "valu_per_wave": round(avg["SQ_INSTS_VALU"] / max(avg.get("SQ_WAVES", 1), 1), 1),
                 "salu_per_wave": round(avg.get("SQ_INSTS_SALU", 0) / max(avg.get("SQ_WAVES", 1), 1), 1),
                 "lds_per_wave": round(avg.get("SQ_INSTS_LDS", 0) / max(avg.get("SQ_WAVES", 1), 1), 1),
                 "active_inst_valu_quadcycles": int(avg.get("SQ_ACTIVE_INST_VALU", 0)),
                 "issue_us_at_peak_clock": round(avg["SQ_INSTS_VALU"] * VALU_ISSUE_CYCLES / N_SIMDS / (SQ_CLOCK_GHZ * 1e3), 2)}
            if k in dur and dur[k][1]:
                e["launch_us_under_counters"] = round(dur[k][0] / dur[k][1], 2)
                if "GRBM_GUI_ACTIVE" in avg and e["launch_us_under_counters"] > 0:   # the counter is the sum over the part's 8 XCDs (r06_c: 17.8 "GHz" raw)
                    e["clock_ghz_under_counters"] = round(avg["GRBM_GUI_ACTIVE"] / N_XCD / e["launch_us_under_counters"] / 1e3, 3)
            out[k] = e
        if out:
            return out, None
        err = "no SQ counters in the pass"
    return None, err


def valu_issue_block(sq, roofline, kernels, ms_per_step):
    """roofline.valu_issue: the ceiling that binds this path.  For the dominant kernel and for the step: vector issue time against
    the kernel / step time.  frac = issue time at the PEAK clock / measured time (a lower bound of the true fraction: under a VALU-dense kernel the
    part clocks below its peak -- clock_ghz_under_counters is what the counter pass saw)."""
    dom = roofline["kernel"]
    group = STAGE_KERNELS.get(dom, (dom,))
    rows = {k: v for k, v in sq.items()}
    dom_issue = sum(rows[k]["issue_us_at_peak_clock"] for k in group if k in rows)
    dom_us = roofline["avg_launch_ms"] * 1e3
    blk = {"model": f"SQ_INSTS_VALU x {VALU_ISSUE_CYCLES} cycles / {N_SIMDS} SIMDs / {SQ_CLOCK_GHZ} GHz per launch (in-run rocprofv3 --pmc pass); frac = issue time / measured time",
           "kernel": dom, "issue_us": round(dom_issue, 1), "launch_us": round(dom_us, 1), "frac": round(dom_issue / dom_us, 3) if dom_us > 0 else None,
           "valu_per_wave": rows.get(dom, {}).get("valu_per_wave"), "waves": rows.get(dom, {}).get("waves")}
    clk = rows.get(dom, {}).get("clock_ghz_under_counters")
    if clk:
        blk["clock_ghz_under_counters"] = clk
        blk["frac_at_that_clock"] = round(dom_issue * SQ_CLOCK_GHZ / clk / (rows[dom].get("launch_us_under_counters") or dom_us), 3)
    # the step: every kernel of the pass x its launches per step (the child pass runs the same step as the timed loop)
    lps = {}
    for name, v in kernels.items():
        for k in STAGE_KERNELS.get(name, (name,)):
            lps[k] = v["launches_per_step"]
    step_issue = 0.0
    for k, v in rows.items():
        n = next((l for kk, l in lps.items() if k.startswith(kk) or kk.startswith(k)), 1.0)
        step_issue += v["issue_us_at_peak_clock"] * n
    blk["step_issue_us"] = round(step_issue, 1)
    blk["step_frac"] = round(step_issue / (ms_per_step * 1e3), 3)
    blk["per_kernel"] = rows
    return blk


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (CPU restatement of the reference path) timed on this box's host cores, bounded sample
# ---------------------------------------------------------------------------------------------------------
def _oracle_euroc_sample(frames, idx, nfeat, w, h):
    from oracle import oracle_binding as ob
    oex = ob.OracleExtractor(nfeat, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
    sf = oex.tables()["scale"]
    feats, prev = 0, None
    for t in idx:
        _, k, d = oex.extract(frames[t % len(frames)], lap=(0, 1000))
        feats += len(k)
        if prev is not None:
            k0, d0 = prev
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"],
                     angle=k0["angle"], desc=d0, has_obs=np.ones(len(k0), np.uint8))
            grid = ob.OracleGrid(k, 0.0, float(w), 0.0, float(h))
            ob.search_by_projection_frame(grid, d, sf, q, 15.0, 0, True)
        prev = (k, d)
    return feats


def reference_build_child(td, n_ref):
    """Child process of cpu_baseline: the compiled reference (oracle/_ref) on the frames saved in td; prints one JSON object."""
    from oracle import oracle_binding as ob
    from oracle import ref_binding as rb
    w, h, nf = WORKLOADS["euroc"][:3]
    frames = np.load(os.path.join(td, "frames.npy"))
    sf = ob.OracleExtractor(nf, 1.2, NLEVELS, 20, 7).tables()["scale"]
    rex = rb.RefExtractor(nf, 1.2, NLEVELS, 20, 7)
    t0 = time.perf_counter()
    feats, prev = 0, None
    for t in range(n_ref):
        _, k, d = rex.extract(frames[t % len(frames)], (0, 1000))
        feats += len(k)
        if prev is not None:
            k0, d0 = prev
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, z=np.ones(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
                     desc=d0, has_obs=np.ones(len(k0), np.uint8))
            rb.ref_search_by_projection_frame(rb.RefFrame(k, d, 0.0, float(w), 0.0, float(h), sf), q, 15.0, 0, True)
        prev = (k, d)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1,
                      "sample": f"{n_ref} frames, {dt:.1f} s, reference ORBextractor.cc + ORBmatcher.cc over oracle/ocv_shim"}))


def cpu_baseline_euroc(frames, n_sample, w, h, nfeat):
    t0 = time.perf_counter()
    feats = _oracle_euroc_sample(frames, range(n_sample), nfeat, w, h)
    dt = time.perf_counter() - t0
    out = {"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port",
           "sample": f"{n_sample} frames of the same workload (extract + frame-to-frame match), {dt:.1f} s, "
                     f"oracle/ C++ restatement -O3 x86-64-v3, 1 thread; host has {os.cpu_count()} logical cores"}
    # frame-parallel on all host cores: the same sample split over one thread per core (ctypes releases the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        nc = max(1, min(os.cpu_count() or 1, 64))
        if nc > 1:
            per = 12   # 12 frames (about 0.4 s) per thread
            chunks = [range(c * per, (c + 1) * per) for c in range(nc)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(nc) as pool:
                fs = list(pool.map(lambda r: _oracle_euroc_sample(frames, r, nfeat, w, h), chunks))
            dt = time.perf_counter() - t0
            out["all_cores"] = {"value": round(sum(fs) / dt / 1e3, 3), "unit": "kfeatures/s", "cores": nc,
                                "sample": f"{nc} threads x {per} frames, {dt:.1f} s"}
            out["value_all_cores"] = out["all_cores"]["value"]   # flat copies: a reader that keeps only cpu_baseline's scalars sees both figures
            out["cores_all"] = nc
    except Exception as e:   # the 1-thread baseline stands on its own
        out["all_cores"] = {"error": str(e)[:200]}
    # beside it, when oracle/_ref travelled here: the reference's OWN ORBextractor.cc + ORBmatcher.cc (compiled where they lie in the
    # build container against the stand-in OpenCV / SLAM types, whose image primitives are the oracle's scalar ones) on a quarter of
    # the sample -- shows the port is not slower than the code it restates; not a substitute for an OpenCV-backed build.  Runs in
    # a child process: nothing that library does can take the bench line down with it.
    try:
        from oracle import ref_binding as rb
        if rb.available() and rb.matcher_available():
            import tempfile
            n_ref = max(2, n_sample // 4)
            with tempfile.TemporaryDirectory() as td:
                np.save(os.path.join(td, "frames.npy"), np.ascontiguousarray(frames[:min(n_ref, len(frames))]))
                r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--reference-build-child", td, str(n_ref)],
                                   capture_output=True, text=True, timeout=300)
            out["reference_build"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": f"child exit {r.returncode}"}
    except Exception as e:
        out["reference_build"] = {"error": str(e)[:200]}
    return out


def cpu_baseline_kitti(pairs, n_sample, w, h, nfeat, bf, b):
    from oracle import oracle_binding as ob
    oel = ob.OracleExtractor(nfeat, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
    oer = ob.OracleExtractor(nfeat, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
    tb = oel.tables()
    t0 = time.perf_counter()
    feats = 0
    for t in range(n_sample):
        L, R = pairs[t % len(pairs)]
        _, kl, dl = oel.extract(L, lap=(0, 0))
        _, kr, dr = oer.extract(R, lap=(0, 0))
        pl = [np.ascontiguousarray(oel.level_padded(l)[19:-19, 19:-19]) for l in range(NLEVELS)]
        pr = [np.ascontiguousarray(oer.level_padded(l)[19:-19, 19:-19]) for l in range(NLEVELS)]
        ob.compute_stereo_matches(kl, dl, kr, dr, tb["scale"], tb["inv_scale"], pl, pr, bf, b)
        feats += len(kl) + len(kr)
    dt = time.perf_counter() - t0
    return {"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port",
            "sample": f"{n_sample} stereo pairs of the same workload (2 x extract + ComputeStereoMatches), {dt:.1f} s, oracle/ C++ "
                      f"restatement, 1 thread; host has {os.cpu_count()} logical cores"}


def cpu_baseline_tumvi(frames, mp_sets, n_sample, w, h, nfeat):
    from oracle import oracle_binding as ob
    oex = ob.OracleExtractor(nfeat, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
    sf = oex.tables()["scale"]
    t0 = time.perf_counter()
    feats = 0
    for t in range(n_sample):
        f = t % len(frames)
        _, k, d = oex.extract(frames[f], lap=(0, 1000))
        grid = ob.OracleGrid(k, 0.0, float(w), 0.0, float(h))
        ob.search_by_projection_mappoints(grid, d, sf, mp_sets(f), 1.0, 0.8)
        feats += len(k)
    dt = time.perf_counter() - t0
    return {"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port",
            "sample": f"{n_sample} frames of the same workload (extract + SearchByProjection vs {N_MAPPOINTS} map points), {dt:.1f} s, "
                      f"oracle/ C++ restatement, 1 thread; host has {os.cpu_count()} logical cores"}


# ---------------------------------------------------------------------------------------------------------
# rank process
# ---------------------------------------------------------------------------------------------------------
def parse_cpulist(text):
    ids = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        ids.update(range(int(a), int(b or a) + 1))
    return ids


def rank_core_set(index, bdfs, sysfs="/sys/bus/pci/devices"):
    """CPU cores for the rank that drives GPU `index` of the node's GPUs `bdfs` (PCI addresses in device order): the cores local to the GPU
    (sysfs local_cpulist), and -- where several GPUs hang off the same cores, e.g. four per socket on an 8-GPU node -- an equal, DISJOINT share of
    them per GPU, so that eight ranks do not all run (and first-touch their pinned rings) on the first cores of a socket.  Returns (core set, numa node,
    cpulist text); raises when sysfs has no entry.  Pure function of the sysfs tree: tests/test_sharding_gloo.py runs it on a fake 8-GPU topology."""
    lists = []
    for b in bdfs:
        lists.append(open(f"{sysfs}/{b}/local_cpulist").read().strip())
    mine = lists[index]
    node = open(f"{sysfs}/{bdfs[index]}/numa_node").read().strip()
    cores = sorted(parse_cpulist(mine))
    peers = [i for i, l in enumerate(lists) if l == mine]
    k, n = peers.index(index), len(peers)
    share = cores[k * len(cores) // n:(k + 1) * len(cores) // n] if len(cores) >= n else cores
    return set(share), node, mine


def bind_to_gpu_numa_node(torch, index):
    """Run this rank (and first-touch its pinned buffers) on the CPU cores local to its GPU: the host-input leg moves 92 MB per step
    over PCIe, and pinned memory on the other socket costs a third of the bandwidth.  GPUs that share a set of local cores split it
    (rank_core_set).  Best effort; returns a short description."""
    try:
        bdfs = []
        for d in range(torch.cuda.device_count()):
            p = torch.cuda.get_device_properties(d)
            bdfs.append(f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0")
        ids, node, cpus = rank_core_set(index, bdfs)
        if ids:
            os.sched_setaffinity(0, ids)
            return f"gpu {bdfs[index]} numa {node} cpus {cpus} -> {len(ids)} cores from {min(ids)}"
    except Exception as e:   # no sysfs entry / no permission: keep the inherited affinity
        return f"unbound ({type(e).__name__})"
    return "unbound"


class Rank:
    def __init__(self, args):
        self.args = args
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dry = bool(os.environ.get("ORBX_BENCH_DRY"))   # launcher / sharding test on a box without GPUs: no device work
        self.device = self.local_rank
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if not self.dry:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs a GPU (liborbx has no CPU path)")
            # --share-gpus: ranks beyond the visible GPUs wrap around (rank r on GPU r mod count) -- the real multi-rank path (gloo group, one
            # extractor + streams per rank, barrier, gather, parity on every rank) exercised on a single-GPU box; not a scaling number
            self.device = self.local_rank % torch.cuda.device_count() if args.share_gpus else self.local_rank
            if torch.cuda.device_count() <= self.device:
                raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
            torch.cuda.set_device(self.device)
            self.numa = bind_to_gpu_numa_node(torch, self.device)
        self.group = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            from orb_slam3_amd import sharding
            sharding.init_host_group(self.rank, self.world)   # gloo: the barrier and the gather are host-side, nothing but HIP on the GPUs
            self.group = "gloo"

    def barrier(self, extractors=()):
        if not self.dry:
            self.torch.cuda.synchronize()
            for e in extractors:
                e.sync()
        if self.world > 1:
            self.dist.barrier()
        if not self.dry:
            self.torch.cuda.synchronize()

    # The timed region runs with Python's cyclic garbage collector off (and the heap collected just before): a generation-2 collection
    # of a process that has imported torch takes 40-80 ms -- several times a whole 20-step timed region -- and landed inside it in
    # about one run in twenty (seen as a single 82 ms stall: 5.5 instead of 1.35 ms per step).  Host-harness noise, not the path's.
    # `warm` = the W untimed warm-up steps, run AFTER the collection: the timed region then follows them directly.  (Until round 4 the collection sat
    # between warm-up and region -- tens of milliseconds of an idle device, long enough for it to leave its working clocks: every 20-step region
    # started cold.)
    def timed_begin(self, extractors=(), warm=None):
        import gc
        gc.collect()
        gc.disable()
        if warm is not None:
            warm()
        self.barrier(extractors)
        return time.perf_counter()

    def timed_end(self, t0, extractors=()):
        import gc
        self.barrier(extractors)
        dt = time.perf_counter() - t0
        gc.enable()
        return dt

    def reduce(self, dt, units):
        from orb_slam3_amd import sharding
        dt_max, units_all, per = sharding.gather_throughput(dt, units)
        self.per_rank = [{"rank": r, "seconds": round(t, 6), "kfeatures_per_s": round(u / t / 1e3, 2) if t > 0 else None} for r, (t, u) in enumerate(per)]
        return dt_max, units_all

    def gather_objects(self, obj):
        """every rank's small python object (a parity verdict) on every rank, over the host group"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def finish(self, out):
        if self.rank == 0:
            print(json.dumps(out), flush=True)
        if self.world > 1:
            self.dist.destroy_process_group()


def base_line(R, metric, value, dt_max, extra_cfg):
    a = R.args
    return {"metric": metric, "value": round(value, 2), "unit": "kfeatures/s", "n_gpus": R.world, "steps": a.steps, "warmup": a.warmup,
            "settle_steps": a.settle,
            "ms_per_step": round(dt_max / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "config": extra_cfg, "process_group": R.group, "per_rank": getattr(R, "per_rank", None)}


STAMPS = []          # host time at which every step's results had arrived (appended by the run loops, cut per region by region_gaps)


def region_gaps():
    """Largest and median gap between consecutive step completions since the last call, in ms (then clears the list): a region that is slow because
    of ONE long stall (host thread descheduled, a driver housekeeping pause) looks different from one whose every step is slow (clocks, contention)."""
    g = np.diff(np.array(STAMPS)) * 1e3 if len(STAMPS) > 2 else np.zeros(1)
    STAMPS.clear()
    return {"max_gap_ms": round(float(g.max()), 3), "median_gap_ms": round(float(np.median(g)), 3), "argmax": int(g.argmax())}


def repeats_block(regions, steps, gaps=None):
    order = [round(t / steps * 1e3, 3) for t, _ in regions]
    ms = sorted(t / steps * 1e3 for t, _ in regions)
    vals = sorted(f / t / 1e3 for t, f in regions)
    return {"regions": len(regions), "steps_per_region": steps, "ms_per_step_in_order": order, "step_gaps": gaps,
            "ms_per_step": {"median": round(float(np.median(ms)), 3), "min": round(ms[0], 3), "max": round(ms[-1], 3)},
            "value": {"median": round(float(np.median(vals)), 2), "min": round(vals[0], 2), "max": round(vals[-1], 2)},
            "note": "`value` / `ms_per_step` of the line are the FIRST region (exactly K steps after W warm-up steps); the others follow back to back"}


def settle(run, a, done=0):
    """Untimed steps before the W warm-up steps: at least a.settle of them AND at least a.settle_seconds of device work -- a fresh process
    starts with the device in a low power state and the runtime's pools cold; with 13 steps of 0.8 ms before it the first timed region of the
    KITTI workload came out at 1.13 instead of 0.78 ms per step in one run of three (the other four regions of the same run: 0.78).
    Sets a.settle to the number of steps actually run (reported as settle_steps)."""
    t0 = time.perf_counter()
    while done < a.settle or (time.perf_counter() - t0 < a.settle_seconds and done < 4000):
        n = a.settle - done if done < a.settle else 8
        run(n)
        done += n
    a.settle = done


def pinned(torch, shape, dtype):
    return torch.empty(shape, dtype=dtype).pin_memory()


def bench_threads(args):
    """`--threads`: SURVEY.md 8(e) as ONE process -- N host worker threads, thread s on GPU s mod G, each with its own extractor (its own HIP streams),
    its own resident frames and its own pinned result ring; no process group, no collective.  The same extract + match + download loop as the
    euroc workload's `value`; value = features of all threads / the slowest thread's time between a common start and its own last result.
    Parity of every thread's last step against the oracle.  (tests/cpp/multi_gpu_demo.cpp is the same shape in C++.)"""
    import threading
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H, NF, _, LAP = WORKLOADS["euroc"]
    B = args.batch or WORKLOADS["euroc"][3]
    N = max(1, args.gpus)
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py --threads: no GPU")
    start = threading.Barrier(N + 1)
    res = [None] * N

    class HostSet:
        def __init__(self, cap):
            self.kps = pinned(torch, (B, cap, 28), torch.uint8)
            self.desc = pinned(torch, (B, cap, 32), torch.uint8)
            self.cnt = pinned(torch, (B,), torch.int32).zero_()
            self.mono = pinned(torch, (B,), torch.int32).zero_()
            self.match = pinned(torch, (B, cap), torch.int32)
            self.nm = pinned(torch, (B,), torch.int32).zero_()

    def work(i):
        dev = i % ndev
        canvas = synth.make_canvas(10 + i)
        frames = np.stack([synth.frame_from_canvas(canvas, t, W, H, 1000 * (10 + i) + t) for t in range(B)])
        d_frames = torch.from_numpy(frames).to(f"cuda:{dev}")
        torch.cuda.synchronize(dev)
        ex = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=dev)
        host = [HostSet(ex.output_capacity(W, H)) for _ in range(2)]

        def run(nsteps):
            feats = 0
            for k in range(nsteps + 1):
                if k < nsteps:
                    hs = host[k % 2]
                    ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, LAP)
                    ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
                    ex.download_async(hs.kps.data_ptr(), hs.desc.data_ptr(), hs.cnt.data_ptr(), hs.mono.data_ptr(), hs.match.data_ptr(), hs.nm.data_ptr())
                if k >= 1:
                    ex.download_wait()
                    feats += int(host[(k - 1) % 2].cnt.sum())
            return feats
        run(max(args.warmup, 1) + min(args.settle, 8))
        ex.sync()
        start.wait()
        t0 = time.perf_counter()
        feats = run(args.steps)
        ex.sync()
        dt = time.perf_counter() - t0
        parity = verify_euroc(frames, host[(args.steps - 1) % 2], args.verify, W, H, NF, ex) if args.verify > 0 else None
        res[i] = {"thread": i, "device": dev, "features": feats, "seconds": dt, "ms_per_step": round(dt / args.steps * 1e3, 3), "parity_checked": parity}

    threads = [threading.Thread(target=work, args=(i,)) for i in range(N)]
    for t in threads:
        t.start()
    start.wait()
    for t in threads:
        t.join()
    if any(r is None for r in res):
        raise SystemExit("bench.py --threads: a worker thread died")
    dt_max = max(r["seconds"] for r in res)
    feats = sum(r["features"] for r in res)
    out = {"metric": "ORB kfeatures/sec extract+match, EuRoC 752x480 nFeatures=1000", "value": round(feats / dt_max / 1e3, 2), "unit": "kfeatures/s",
           "n_gpus": min(N, ndev), "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "EuRoC-shaped 752x480 mono, nFeatures=1000: extract + frame-to-frame SearchByProjection(th=15) + D2H; inputs resident in HBM",
                      "frames_per_step_per_thread": B, "sequences": N,
                      "parallelism": f"ONE process, {N} host worker threads, thread s on GPU s mod {ndev}: one extractor + its HIP streams + one pinned result ring per thread; no collective"},
           "host_mode": "threads", "threads": N, "visible_gpus": ndev, "per_thread": res,
           "parity_checked": all(r["parity_checked"] is not None for r in res) if args.verify > 0 else None}
    print(json.dumps(out), flush=True)
    return 0


def bench_dry(R):
    """No-GPU path of the launcher test: every rank 'processes' its own sequence and the reduction runs over gloo."""
    a = R.args
    fail = os.environ.get("ORBX_BENCH_DRY_FAIL")   # test hook of the launcher's retry path: rank 1 dies always / once
    if fail and R.rank == R.world - 1:
        if fail == "always" or not os.path.exists(fail):
            if fail != "always":
                open(fail, "w").close()
            os._exit(134)
    t0 = R.timed_begin()
    units = 1000.0 * a.steps * (R.rank + 1)
    time.sleep(0.01)
    dt_max, units_all = R.reduce(R.timed_end(t0), units)
    for pr, pc in zip(R.per_rank, R.gather_objects({"checked_by_rank": R.rank})):   # the path the ranks' parity verdicts take to rank 0
        pr["parity_checked"] = pc
    out = base_line(R, "dry run (ORBX_BENCH_DRY): launcher + sharding only", units_all / dt_max / 1e3, dt_max,
                    {"workload": "none", "sequences": R.world})
    out["roofline"] = None
    out["cpu_baseline"] = None
    R.finish(out)


def bench_euroc(R):
    a, torch = R.args, R.torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H, NF, _, LAP = WORKLOADS["euroc"]
    B = a.batch or WORKLOADS["euroc"][3]
    seed = 10 + R.rank
    from orb_slam3_amd import dataset
    data = "synthetic"
    # NS input sets of B frames each, used in rotation by the timed loop: one set of 256 frames is 92 MB, less than the 256 MB Infinity Cache, so a
    # loop over ONE resident set could be served its frame reads from the cache; three sets (277 MB) cannot (`single_input_set` reports the
    # difference).  Set k = frames k*B .. (k+1)*B - 1 of the rank's sequence.
    NS = max(1, a.input_sets)
    if dataset.dataset_dir("euroc"):   # a real EuRoC sequence: rank r takes frames r*NS*B .. (r+1)*NS*B - 1
        allf = dataset.load_mono("euroc", NS * B, W, H, start=R.rank * NS * B)
        data = f"dataset: {dataset.dataset_dir('euroc')} (first {NS * B} frames per rank)"
    else:
        canvas = synth.make_scene_canvas(a.scene, seed)
        if a.scene != "quads":
            data = f"synthetic, STRESS scene {a.scene} (synth.make_texture_canvas: 1/f noise + dense high-contrast texture; blend:<a> mixes it into the quad scene); not the metric's scene"
        allf = np.stack([synth.frame_from_canvas(canvas, t, W, H, 1000 * seed + t) for t in range(NS * B)])
    sets = [np.ascontiguousarray(allf[k * B:(k + 1) * B]) for k in range(NS)]
    frames = sets[0]
    h_sets = [torch.from_numpy(f).pin_memory() for f in sets]   # the camera thread's buffers (pcie_inclusive leg)
    d_sets = [torch.from_numpy(f).cuda() for f in sets]         # resident inputs of the contract's `value`
    torch.cuda.synchronize()
    rotate = [True]
    step_no = [0]       # steps enqueued so far: step i reads set i mod NS
    last_set = [0]

    ex = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=R.device)
    cap = ex.output_capacity(W, H)

    class HostSet:   # pinned host destinations (the Tracking thread's buffers)
        def __init__(self):
            self.kps = pinned(torch, (B, cap, 28), torch.uint8)
            self.desc = pinned(torch, (B, cap, 32), torch.uint8)
            self.cnt = pinned(torch, (B,), torch.int32).zero_()
            self.mono = pinned(torch, (B,), torch.int32).zero_()
            self.match = pinned(torch, (B, cap), torch.int32)
            self.nm = pinned(torch, (B,), torch.int32).zero_()
    host = [HostSet(), HostSet()]   # double-buffered: the D2H of step i overlaps the kernels of step i+1
    ablate = set(filter(None, (a.ablate or "").split(",")))   # DIAGNOSTIC (--ablate): parts of the step left out; the line says so and is not a result
    if ablate:
        a.verify = 0

    def enqueue(i, from_host):
        hs = host[i % 2]
        k = step_no[0] % NS if rotate[0] else 0
        step_no[0] += 1
        last_set[0] = k
        if from_host:
            ex.extract_batch_host(h_sets[k].data_ptr(), B, W, H, W, W * H, LAP)
        else:
            ex.extract_batch_device(d_sets[k].data_ptr(), B, W, H, W, W * H, LAP)
        if "nomatch" not in ablate:
            ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        if "nodl" in ablate:     # DIAGNOSTIC: only the counts travel (4 B per frame) -- what the D2H of keypoints / descriptors costs the step
            ex.download_async(0, 0, hs.cnt.data_ptr(), hs.mono.data_ptr(), 0, 0)
        else:
            ex.download_async(hs.kps.data_ptr(), hs.desc.data_ptr(), hs.cnt.data_ptr(), hs.mono.data_ptr(),
                              hs.match.data_ptr() if "nomatch" not in ablate else 0, hs.nm.data_ptr() if "nomatch" not in ablate else 0)

    def run(nsteps, from_host=False):
        """nsteps pipelined steps, two batches in flight; returns the number of features delivered to the host."""
        feats = 0
        for i in range(nsteps + 1):
            if i < nsteps:
                t = time.perf_counter()
                enqueue(i, from_host)
                host_enqueue[0] += time.perf_counter() - t
            if i >= 1:
                ex.download_wait()
                STAMPS.append(time.perf_counter())
                feats += int(host[(i - 1) % 2].cnt.sum())
        return feats

    host_enqueue = [0.0]   # seconds the host thread spent issuing work (HIP API calls through the C ABI), per timed region

    def timed(from_host):
        t0 = R.timed_begin([ex], warm=lambda: run(max(a.warmup, 1), from_host))
        STAMPS.clear()
        host_enqueue[0] = 0.0
        feats = run(a.steps, from_host)
        return R.timed_end(t0, [ex]), feats

    t_settle = time.perf_counter()
    run(min(a.settle, 8), False)
    R.barrier([ex])
    settle_ms = (time.perf_counter() - t_settle) / max(min(a.settle, 8), 1) * 1e3   # the fresh process's first steps: start-up transient included
    # the FAST stage's candidate queues follow the imagery (orbx_tune_fast_queues: results never depend on them): on the quad scene 1 % of the cells take the
    # list pass and nothing changes, on --scene texture two thirds do and the queues grow in up to three steps -- all before the warm-up
    fast_queues = []
    for _ in range(3):
        fast_queues.append(ex.tune_fast_queues(1))
        if not fast_queues[-1]["changed"]:
            break
        run(2, False)
        R.barrier([ex])
    settle(lambda n: run(n, False), a, done=min(a.settle, 8))
    STAMPS.clear()
    dt, feats = timed(False)
    gaps = [region_gaps()]
    enqueue_ms = host_enqueue[0] / a.steps * 1e3
    last = host[(a.steps - 1) % 2]
    nmatch = int(last.nm[1:].sum())
    dt_max, feats_all = R.reduce(dt, feats)
    per_rank = R.per_rank

    # ---- parity of the delivered results (last timed step) against the CPU oracle: EVERY rank checks its own sequence ----
    parity = None
    if a.verify > 0:
        parity = verify_euroc(sets[last_set[0]], last, a.verify, W, H, NF, ex)   # the set the last timed step read
    parity_all = R.gather_objects(parity)
    for pr, pc in zip(per_rank, parity_all):
        pr["parity_checked"] = pc

    # ---- the K timed steps again, `repeat` regions in all (each between its own barriers): spread of the number above ----
    regions = [(dt_max, feats_all)]
    for _ in range(max(a.repeat, 1) - 1):
        t0 = R.timed_begin([ex], warm=lambda: run(max(a.warmup, 1), False))
        STAMPS.clear()
        f_r = run(a.steps, False)
        regions.append(R.reduce(R.timed_end(t0, [ex]), f_r))
        gaps.append(region_gaps())
    R.per_rank = per_rank
    repeats = repeats_block(regions, a.steps, gaps)
    # ---- one more region over ONE resident set (what rounds 1-4 timed): does the Infinity Cache serve the frame reads then? ----
    single = None
    if NS > 1:
        rotate[0] = False
        t0 = R.timed_begin([ex], warm=lambda: run(max(a.warmup, 1), False))
        STAMPS.clear()
        f_r = run(a.steps, False)
        dt_s, f_s = R.reduce(R.timed_end(t0, [ex]), f_r)
        rotate[0] = True
        single = {"ms_per_step": round(dt_s / a.steps * 1e3, 3), "value": round(f_s / dt_s / 1e3, 2),
                  "note": f"the same loop re-reading ONE resident set of {B} frames ({B * W * H / 1e6:.0f} MB, inside the 256 MB Infinity Cache); `value` rotates {NS} sets"}
    R.per_rank = per_rank

    # ---- the same loop with the frames starting in pinned host memory (upload inside the timed region) ----
    dt_h, feats_h = timed(True)
    dt_h_max, feats_h_all = R.reduce(dt_h, feats_h)
    per_rank_host = R.per_rank
    R.per_rank = per_rank     # `per_rank` of the line belongs to `value` (resident inputs); the host-input leg carries its own list
    pcie_parity = None
    if a.verify > 0:
        pcie_parity = verify_euroc(sets[last_set[0]], host[(a.steps - 1) % 2], a.verify, W, H, NF, ex)   # the host-input path delivers the same results
    for pr, pc in zip(per_rank_host, R.gather_objects(pcie_parity)):
        pr["parity_checked"] = pc
    pcie = {"value": round(feats_h_all / dt_h_max / 1e3, 2), "unit": "kfeatures/s", "ms_per_step": round(dt_h_max / a.steps * 1e3, 3),
            "h2d_bytes_per_step": int(B * W * H), "h2d_GBs": round(B * W * H / (dt_h_max / a.steps) / 1e9, 1), "host_affinity": R.numa,
            "parity_checked": pcie_parity, "per_rank": per_rank_host,
            "note": "frames in pinned host memory; orbx_extract_batch_host uploads batch i+1 on its own stream while batch i computes"}

    # ---- per-kernel timing with HIP events on the extractor's stream (separate, untimed, serialized passes) ----
    roofline, kernels = None, {}
    if R.rank == 0 and not a.no_profile:
        passes = 3
        ex.profile_enable(True)
        for _ in range(passes):
            run(1)
        prof = ex.profile_read()
        ex.profile_enable(False)
        n_feat = int(last.cnt.sum())
        samp = list(range(0, B, max(1, B // 8)))   # candidate count sampled on a few frames
        n_cand = int(sum(len(ex.debug_candidates(l, f)) for f in samp for l in range(NLEVELS)) * B / len(samp))
        sizes = [ex.level_size(l, (W, H)) for l in range(NLEVELS)]
        roofline, kernels = roofline_from_profile(ex, prof, passes, sizes, B, n_feat, n_cand)
        if a.pmc and R.world == 1:
            tr, err = pmc_traffic(roofline["kernel"], "euroc", B)
            if tr:
                roofline["traffic"] = int(tr["read"] + tr["write"])
                roofline["traffic_detail"] = {"read_bytes": int(tr["read"]), "write_bytes": int(tr["write"]),
                                              "traffic_over_algorithmic": round((tr["read"] + tr["write"]) / roofline["alg_bytes_per_launch"], 2),
                                              "note": "rocprofv3 --pmc child passes of this run; reads = 32/64/128-byte request classes "
                                                      "(FETCH_SIZE tallies gfx950's 128-byte requests at 64 bytes), writes = WRITE_SIZE"}
            else:
                roofline["traffic_error"] = err
            sq, err = pmc_sq("euroc", B)
            if sq:
                roofline["valu_issue"] = valu_issue_block(sq, roofline, kernels, dt_max / a.steps * 1e3)
            else:
                roofline["valu_issue_error"] = err

    cpu = None
    if R.rank == 0 and R.world == 1 and a.cpu_frames > 0:
        cpu = cpu_baseline_euroc(frames, a.cpu_frames, W, H, NF)

    # ---- single-frame latency of the drop-in call (Tracking calls operator() once per frame: Frame.cc:418-425, ORBextractor.cc:1086) ----
    latency = None
    if R.rank == 0 and a.latency > 0:
        latency = latency_euroc(osa, frames, a.latency, W, H, NF, LAP, R.device, cpu)
        latency["calls_per_entry_point"] = latency_calls(osa, R.device, max(a.latency // 4, 10))

    # ---- BASELINE configs 3 and 4 in the same line (1 GPU): child processes of this file, own parity checks, no CPU / PMC legs ----
    others = None
    if R.rank == 0 and R.world == 1 and a.other_workloads:
        others = {wl: other_workload_child(wl, a) for wl in ("kitti", "tumvi")}

    out = base_line(R, "ORB kfeatures/sec extract+match, EuRoC 752x480 nFeatures=1000", feats_all / dt_max / 1e3, dt_max,
                    {"workload": "EuRoC-shaped 752x480 mono, nFeatures=1000, 8 levels, scale 1.2, FAST 20/7: extract + frame-to-frame "
                                 "SearchByProjection(th=15) + D2H of results; inputs resident in HBM",
                     "frames_per_step_per_gpu": B, "sequences": R.world, "features_per_frame": round(feats / a.steps / B, 1),
                     "matches_per_frame": round(nmatch / max(B - 1, 1), 1), "parallelism": f"{R.world} independent sequences, one per GPU"})
    out["data"] = data
    out["stage_stats_last_step"] = ex.stage_stats()   # cells on the FAST list pass, quad-tree tiers, candidates: which paths the frames exercise
    out["fast_queues"] = {"tuning_after_settle": fast_queues, "in_force": ex.tune_fast_queues(0),
                          "note": "orbx_tune_fast_queues(mode 1) after the first settle steps; `changed` false = the default queues (512 groups / 816 pixels per wave)"}
    if ablate:
        out["ablation"] = sorted(ablate)
        out["metric"] = "DIAGNOSTIC, NOT A RESULT (parts of the step left out: " + ",".join(sorted(ablate)) + "): " + out["metric"]
    out.update({"roofline": roofline, "cpu_baseline": cpu, "pcie_inclusive": pcie, "parity_checked": parity, "kernels": kernels,
                "host_enqueue_ms_per_step": round(enqueue_ms, 3), "settle_ms_per_step": round(settle_ms, 3),
                "repeats": repeats, "latency": latency, "other_workloads": others, "input_sets": NS, "single_input_set": single,
                "value_device_resident": round(feats_all / dt_max / 1e3, 2), "value_pcie_inclusive": pcie["value"],
                "metric_definition_note": "TWO clocks, both in this line.  `value` (= value_device_resident) = features delivered to pinned host memory "
                                          "per second with the input frames already resident in HBM when the timed region starts: the bench contract "
                                          "of this build ('inputs already resident in HBM ... the PCIe-inclusive rate is never value').  SURVEY.md "
                                          "8(d)'s wall clock covering H2D of the images, all kernels, D2H of the results is `value_pcie_inclusive` "
                                          "(= pcie_inclusive.value: frames start in pinned host memory, the upload of batch i+1 overlaps batch i); it "
                                          "is bound by the host link, not by the device (pcie_inclusive.h2d_GBs against ~55 GB/s of PCIe 5 x16), and "
                                          "has its own per_rank list and parity check"})
    R.finish(out)


def latency_euroc(osa, frames, n_calls, W, H, NF, LAP, device, cpu):
    """orbx_extract one frame at a time on a fresh extractor: host image in, keypoints + descriptors out (synchronous: H2D, every kernel
    of the extraction at batch size 1, D2H), the way Frame::ExtractORB calls the reference."""
    ex1 = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=device)
    for i in range(20):
        ex1(frames[i % len(frames)], None, LAP)
    ts = []
    for i in range(n_calls):
        t0 = time.perf_counter()
        ex1(frames[i % len(frames)], None, LAP)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    out = {"call": "ORBextractor::operator() (orbx_extract), one 752x480 frame per call, host image in, host results out",
           "calls": n_calls, "median_ms": round(float(np.median(ts)), 3), "p90_ms": round(float(np.percentile(ts, 90)), 3),
           "min_ms": round(float(ts.min()), 3), "frames_per_s": round(1e3 / float(np.median(ts)), 1)}
    if cpu and cpu.get("value"):
        out["cpu_oracle_ms_per_frame"] = round(1000.0 / cpu["value"], 3)   # kfeatures/s at ~1000 features per frame (extract + match, 1 thread)
    return out


def latency_calls(osa, device, n_calls, n_cpu=5, small=False):
    """latency.calls: one matcher call at a time through the C ABI, host arrays in, host arrays out -- the way SLAM consumes the path
    (Tracking.cc:3390-3413 calls SearchByProjection once per frame, LocalMapping.cc:412 SearchForTriangulation once per key-frame pair,
    Frame.cc:811 ComputeStereoMatches once per stereo frame).  Each call beside the CPU oracle's time for the IDENTICAL call (1 thread, a few
    repetitions), parity asserted on the timed inputs.  Untimed set-up: extraction of the frames the calls work on."""
    from orb_slam3_amd import synth
    from oracle import oracle_binding as ob
    rng = np.random.default_rng(77)

    def timed(fn, n):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        return r, {"median_ms": round(float(np.median(ts)), 4), "p90_ms": round(float(np.percentile(ts, 90)), 4), "min_ms": round(float(ts.min()), 4)}

    def entry(name, ref, gpu_fn, cpu_fn, same, extra=None):
        g, tg = timed(gpu_fn, n_calls)
        c, tc = timed(cpu_fn, n_cpu)
        ok = bool(same(g, c))
        if not ok:
            print(f"PARITY FAILURE in latency.calls: {name}", file=sys.stderr)
        e = {"call": name, "reference": ref, **tg, "calls": n_calls, "cpu_oracle_median_ms": tc["median_ms"], "cpu_calls": n_cpu,
             "speedup_vs_cpu_oracle": round(tc["median_ms"] / tg["median_ms"], 2), "parity_checked": ok}
        if extra:
            e.update(extra)
        return e

    def noisy(d, p):
        flip = np.packbits(rng.random((len(d), 256)) < p, axis=1, bitorder="little")
        return np.ascontiguousarray(d ^ flip)

    out = []
    # ---- M1: SearchByProjection(Frame, MapPoints): 1024 x 1024 frame, nFeatures 1500, 10 000 map points (BASELINE config 4) ----
    canvas = synth.make_canvas(4)
    ex = osa.ORBextractor(1500, 1.2, NLEVELS, 20, 7, device=device)
    kk, dd = [], []
    for t in range(1, 9):
        _, k, d = ex(synth.frame_from_canvas(canvas, t, 1024, 1024, 5000 + t), None, (0, 1000))
        k = k.copy(); k["x"] += 2.0 * t; k["y"] += 1.0 * t
        kk.append(k); dd.append(d)
    _, kf, df = ex(synth.frame_from_canvas(canvas, 0, 1024, 1024, 5000), None, (0, 1000))
    sf = ex.GetScaleFactors()
    src_k, src_d = np.concatenate(kk)[:N_MAPPOINTS], np.concatenate(dd)[:N_MAPPOINTS]
    n_mp = len(src_k)
    mp = dict(proj_x=src_k["x"] + rng.normal(0, 2, n_mp).astype(np.float32), proj_y=src_k["y"] + rng.normal(0, 2, n_mp).astype(np.float32),
              proj_xr=np.zeros(n_mp, np.float32), level=src_k["octave"], view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32),
              desc=noisy(src_d, 0.04), in_view=(rng.random(n_mp) < 0.95).astype(np.uint8), has_obs=(rng.random(n_mp) < 0.97).astype(np.uint8))
    occ = (rng.random(len(kf)) < 0.1).astype(np.uint8)
    m = osa.ORBmatcher(0.8, True, device=device)
    F = osa.FrameView(kf, df, 0.0, 1024.0, 0.0, 1024.0, sf)
    grid = ob.OracleGrid(kf, 0.0, 1024.0, 0.0, 1024.0)
    out.append(entry("SearchByProjection(Frame, MapPoints) [orbx_search_by_projection_mappoints]: 10000 map points, 1024x1024 frame of %d keypoints, th=1" % len(kf),
                     "ORBmatcher.cc:43-213, Tracking.cc:3390-3413",
                     lambda: m.SearchByProjection(F, mp, 1.0, occ), lambda: ob.search_by_projection_mappoints(grid, df, sf, mp, 1.0, 0.8, None, occ),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1]), {"note": "the oracle's time excludes its grid (Frame::AssignFeaturesToGrid runs at frame "
                                                                                "construction in the reference); the device call builds its grid inside"}))
    # ---- M2: SearchByProjection(CurrentFrame, LastFrame): consecutive 752 x 480 frames, th=15 ----
    canvas1 = synth.make_canvas(1)
    ex2 = osa.ORBextractor(1000, 1.2, NLEVELS, 20, 7, device=device)
    _, k0, d0 = ex2(synth.frame_from_canvas(canvas1, 0, 752, 480, 1000), None, (0, 1000))
    _, k1, d1 = ex2(synth.frame_from_canvas(canvas1, 1, 752, 480, 1001), None, (0, 1000))
    sf2 = ex2.GetScaleFactors()
    q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"], desc=d0,
             has_obs=(rng.random(len(k0)) < 0.9).astype(np.uint8))
    occ1 = (rng.random(len(k1)) < 0.05).astype(np.uint8)
    m2 = osa.ORBmatcher(0.9, True, device=device)
    F1 = osa.FrameView(k1, d1, 0.0, 752.0, 0.0, 480.0, sf2)
    g1 = ob.OracleGrid(k1, 0.0, 752.0, 0.0, 480.0)
    out.append(entry("SearchByProjection(CurrentFrame, LastFrame) [orbx_search_by_projection_frame]: %d points of the last frame, th=15" % len(k0),
                     "ORBmatcher.cc:1676-1887, Tracking.cc:2886",
                     lambda: m2.SearchByProjectionFrame(F1, q, 15.0, 0, occ1), lambda: ob.search_by_projection_frame(g1, d1, sf2, q, 15.0, 0, True, None, occ1),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    # ---- M8: Frame::ComputeStereoMatches, 1241 x 376 pair, nFeatures 2000, Hamming row band + SAD refinement + median rejection ----
    canvas3 = synth.make_canvas(3)
    left, right = synth.make_stereo_pair(3, 0, 1241, 376, canvas3)
    exl, exr = osa.ORBextractor(2000, 1.2, NLEVELS, 20, 7, device=device), osa.ORBextractor(2000, 1.2, NLEVELS, 20, 7, device=device)
    _, kl, dl = exl(left, None, (0, 0))
    _, kr, dr = exr(right, None, (0, 0))
    sf3, isf3 = exl.GetScaleFactors(), exl.GetInverseScaleFactors()
    pyl = [np.ascontiguousarray(pp) for pp in exl.mvImagePyramid]
    pyr = [np.ascontiguousarray(pp) for pp in exr.mvImagePyramid]
    bf, b = 0.53716 * 718.856, 0.53716
    m3 = osa.ORBmatcher(device=device)
    out.append(entry("Frame::ComputeStereoMatches [orbx_compute_stereo_matches]: 1241x376 pair, %d / %d keypoints, host pyramids in (as mvImagePyramid)" % (len(kl), len(kr)),
                     "Frame.cc:811-981",
                     lambda: m3.ComputeStereoMatches(kl, dl, kr, dr, sf3, isf3, pyl, pyr, bf, b),
                     lambda: ob.compute_stereo_matches(kl, dl, kr, dr, sf3, isf3, pyl, pyr, bf, b)[:3],
                     lambda g, c: g[0] == c[0] and g[1].tobytes() == c[1].tobytes() and g[2].tobytes() == c[2].tobytes(),
                     {"note": "the call uploads both 8-level pyramids (the reference's mvImagePyramid is host memory at this boundary): %.1f MB per call"
                              % (sum(pp.nbytes for pp in pyl + pyr) / 1e6)}))
    # ---- M5 / M7 / Fuse on the 752 x 480 pair ----
    def nodes(k, n_nodes):
        return (np.floor(k["x"] / 60).astype(np.int64) * 7 + np.floor(k["y"] / 60).astype(np.int64) * 13 + k["octave"] * 31) % n_nodes
    na, nb = nodes(k0, 100), nodes(k1, 100)
    flip = rng.random(len(nb)) < 0.15
    nb[flip] = rng.integers(0, 100, flip.sum())
    fva, fvb = osa.FeatureVector.from_node_of_feature(na), osa.FeatureVector.from_node_of_feature(nb)
    valid0 = (rng.random(len(k0)) < 0.7).astype(np.uint8)
    m5 = osa.ORBmatcher(0.7, True, device=device)
    out.append(entry("SearchByBoW(KeyFrame, Frame) [orbx_search_by_bow_frame]: %d x %d features, 100 vocabulary nodes" % (len(k0), len(k1)),
                     "ORBmatcher.cc:223-425, Tracking.cc:2770",
                     lambda: m5.SearchByBoWFrame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb),
                     lambda: ob.search_by_bow_frame(d0, k0["angle"], valid0, fva, d1, k1["angle"], fvb, 0.7, True),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    sg = (sf2 * sf2).astype(np.float32)
    na6, nb6 = nodes(k0, 60), nodes(k1, 60)
    fva6, fvb6 = osa.FeatureVector.from_node_of_feature(na6), osa.FeatureVector.from_node_of_feature(nb6)
    s0 = (rng.random(len(k0)) < 0.3).astype(np.uint8)
    s1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    K = np.array([[458.654, 0, 367.215], [0, 457.296, 248.375], [0, 0, 1]])
    tt = np.array([0.11, 0.004, 0.01])
    a_ = 0.01
    Rm = np.array([[np.cos(a_), 0, np.sin(a_)], [0, 1, 0], [-np.sin(a_), 0, np.cos(a_)]])
    tx = np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
    Fm = (np.linalg.inv(K).T @ tx @ Rm @ np.linalg.inv(K)).astype(np.float32)
    ep = (410.0, 236.0)
    m7 = osa.ORBmatcher(0.6, True, device=device)
    out.append(entry("SearchForTriangulation(KF1, KF2), pinhole gates on the device [orbx_search_for_triangulation_pinhole]: %d x %d features" % (len(k0), len(k1)),
                     "ORBmatcher.cc:907-1146, LocalMapping.cc:412",
                     lambda: m7.SearchForTriangulationPinhole(k0, d0, s0, fva6, k1, d1, s1, fvb6, sf2, sg, Fm, ep, None, None, False, False),
                     lambda: ob.search_for_triangulation_pinhole(k0, d0, s0, None, fva6, k1, d1, s1, None, fvb6, sf2, sg, Fm, ep, False, True, fma=True),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    # the fisheye rig's form: KannalaBrandt8::epipolarConstrain (unproject, parallax, JacobiSVD triangulation, two reprojection tests) per surviving candidate
    fk1, fnl1, fd1, fid1, fk2, fnl2, fd2, fid2, fR12, ft12, fcams = synth.make_fisheye_keyframes(np.random.default_rng(77), 1000)
    ffv1, ffv2 = osa.FeatureVector.from_node_of_feature(fid1 % 60), osa.FeatureVector.from_node_of_feature(fid2 % 60)
    fs1 = (np.random.default_rng(78).random(len(fk1)) < 0.3).astype(np.uint8)
    fs2 = (np.random.default_rng(79).random(len(fk2)) < 0.3).astype(np.uint8)
    m7k = osa.ORBmatcher(0.6, True, device=device)
    out.append(entry("SearchForTriangulation(KF1, KF2), fisheye rig, KannalaBrandt8::epipolarConstrain on the device [orbx_search_for_triangulation_kb8]: %d x %d features"
                     % (len(fk1), len(fk2)), "ORBmatcher.cc:907-1146 (:1036-1072), KannalaBrandt8.cpp:216-221, 305-400, LocalMapping.cc:412",
                     lambda: m7k.SearchForTriangulationKB8(fk1, fnl1, fd1, fs1, ffv1, fk2, fnl2, fd2, fs2, ffv2, sg, sg, fcams, fcams, fR12, ft12, False),
                     lambda: ob.search_for_triangulation_kb8(fk1, fnl1, fd1, fs1, ffv1, fk2, fnl2, fd2, fs2, ffv2, sg, sg, fcams, fcams, fR12, ft12, False, True),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    isg = ex2.GetInverseScaleSigmaSquares()
    lvl = k0["octave"]
    qf = dict(u=k0["x"] - 2.0 + rng.normal(0, 1.2, len(k0)).astype(np.float32), v=k0["y"] - 1.0 + rng.normal(0, 1.2, len(k0)).astype(np.float32),
              ur=(k0["x"] - 20.0).astype(np.float32), r=(np.float32(3.0) * sf2[lvl]).astype(np.float32), level=lvl, desc=d0)
    Ff = osa.FrameView(k1, d1, 0.0, 752.0, 0.0, 480.0, sf2, None)
    mf = osa.ORBmatcher(device=device)
    out.append(entry("Fuse(KeyFrame, MapPoints), candidate search [orbx_fuse_search]: %d map points into a key frame of %d features" % (len(k0), len(k1)),
                     "ORBmatcher.cc:1148-1455, LocalMapping.cc:676",
                     lambda: mf.FuseSearch(Ff, qf, isg, False), lambda: ob.fuse_search(g1, d1, None, isg, qf, fma=True),
                     lambda g, c: np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1])))
    # ---- round 6: the six entry points that had parity but no clock (VERDICT r5 'missing' 6) ----
    # M9: cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) -- the brute-force "__popcll + wave-reduce" kernel of north_star (Frame.cc:1144)
    mk = osa.ORBmatcher(device=device)
    for nq_, nt_ in (((200, 150),) if small else ((1000, 1000), (5000, 5000))):
        qd_ = rng.integers(0, 256, (nq_, 32), dtype=np.uint8)
        td_ = noisy(np.ascontiguousarray(qd_[rng.permutation(nq_)[:nt_]]), 0.08)
        td_[7] = td_[3]   # an exact distance tie: the lower train index must win
        out.append(entry("BFMatcher::knnMatch(k=2), brute force [orbx_knn2]: %d x %d descriptors" % (nq_, nt_), "Frame.cc:1126-1166 (:1144)",
                         lambda qd_=qd_, td_=td_: mk.knn2(qd_, td_), lambda qd_=qd_, td_=td_: ob.knn2(qd_, td_),
                         lambda g, c: np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1]),
                         {"algorithmic_bytes": 32 * (nq_ + nt_) + 16 * nq_, "hamming_distances": nq_ * nt_}))
    # M3 / M4: SearchByProjection window forms (relocalisation: levels [l-1, l+1], ORBdist 100, rotation check; Sim3: [l-1, l], TH_LOW * ratio)
    lv0 = k0["octave"]
    base_q = dict(x=k0["x"] - 2.0 + rng.normal(0, 1.0, len(k0)).astype(np.float32), y=k0["y"] - 1.0, angle=k0["angle"], desc=d0)
    occw = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    q3 = dict(base_q, r=(np.float32(10.0) * sf2[lv0]).astype(np.float32), min_level=lv0 - 1, max_level=lv0 + 1)
    mw = osa.ORBmatcher(0.9, True, device=device)
    out.append(entry("SearchByProjection(Frame, KeyFrame, sAlreadyFound, th=10, ORBdist=100) [orbx_search_by_projection_window]: %d queries into %d features" % (len(k0), len(k1)),
                     "ORBmatcher.cc:1889-2010, Tracking.cc:3726",
                     lambda: mw.SearchByProjectionWindow(F1, q3, 100.0, True, occw), lambda: ob.search_by_projection_window(g1, d1, q3, 100.0, True, False, occw),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    q4 = dict(base_q, r=(np.float32(8.0) * sf2[lv0]).astype(np.float32), min_level=lv0 - 1, max_level=lv0)
    out.append(entry("SearchByProjection(KeyFrame, Sim3, MapPoints, th=8, ratioHamming=1.5) [orbx_search_by_projection_window]: %d queries into %d features" % (len(k0), len(k1)),
                     "ORBmatcher.cc:427-646, LoopClosing.cc:755",
                     lambda: mw.SearchByProjectionWindow(F1, q4, 75.0, False, occw), lambda: ob.search_by_projection_window(g1, d1, q4, 75.0, False, True, occw),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    # M6: SearchForInitialization with the 5 x nFeatures extractor of the monocular initialisation (Tracking.cc:601, 2491): level-0 keypoints only
    exi = osa.ORBextractor(1500 if small else 5000, 1.2, NLEVELS, 20, 7, device=device)
    _, ki0, di0 = exi(synth.frame_from_canvas(canvas1, 0, 752, 480, 1000), None, (0, 1000))
    _, ki1, di1 = exi(synth.frame_from_canvas(canvas1, 4, 752, 480, 1004), None, (0, 1000))
    Fi = osa.FrameView(ki1, di1, 0.0, 752.0, 0.0, 480.0, sf2)
    gi = ob.OracleGrid(ki1, 0.0, 752.0, 0.0, 480.0)
    prev0 = np.ascontiguousarray(np.stack([ki0["x"], ki0["y"]], axis=1).astype(np.float32))
    mi = osa.ORBmatcher(0.9, True, device=device)
    out.append(entry("SearchForInitialization(F1, F2, windowSize=100) [orbx_search_for_initialization]: %d x %d keypoints of the 5 x nFeatures extractor (%d on level 0)"
                     % (len(ki0), len(ki1), int((ki0["octave"] == 0).sum())), "ORBmatcher.cc:648-763, Tracking.cc:2491",
                     lambda: mi.SearchForInitialization(ki0, di0, Fi, prev0.copy(), 100), lambda: ob.search_for_initialization(ki0, di0, gi, di1, prev0.copy(), 100, 0.9, True),
                     lambda g, c: g[0] == c[0] and np.array_equal(g[1], c[1])))
    # (f)2: DBoW2 transform of one frame's descriptors through a k = 10, L = 5 vocabulary (ORBvoc is k = 10, L = 6; levelsup = 4 as Frame::ComputeBoW)
    kv, Lv = 10, (3 if small else 5)
    n_nodes = (kv ** (Lv + 1) - 1) // (kv - 1)
    ids = np.arange(n_nodes, dtype=np.int64)
    first_child = ids * kv + 1
    is_leaf = first_child >= n_nodes
    cp = np.concatenate([[0], np.cumsum(np.where(is_leaf, 0, kv))]).astype(np.int32)
    ci = np.arange(1, n_nodes, dtype=np.int32)
    nd_desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    wi = np.where(is_leaf, np.cumsum(is_leaf) - 1, -1).astype(np.int32)
    voc = osa.ORBVocabulary(Lv, cp, ci, nd_desc, wi, device=device)
    mb = osa.ORBmatcher(device=device)
    out.append(entry("Frame::ComputeBoW = TemplatedVocabulary::transform(levelsup=4) [orbx_bow_transform]: %d descriptors, k=%d L=%d vocabulary (%d nodes resident)"
                     % (len(d0), kv, Lv, n_nodes), "Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1262, Frame.cc:738-745",
                     lambda: mb.BowTransform(voc, d0, 4), lambda: ob.bow_transform(cp, ci, nd_desc, wi, Lv, 4, d0),
                     lambda g, c: np.array_equal(g[0], c[0]) and np.array_equal(g[1], c[1])))
    # (f)4: MapPoint::ComputeDistinctiveDescriptors for the map points a new key frame touches (LocalMapping.cc:290: one call per map point in the reference)
    sizes = rng.integers(2, 25, 500)
    set_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    based = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    dset = noisy(np.repeat(based, sizes, axis=0), 0.08)
    out.append(entry("MapPoint::ComputeDistinctiveDescriptors x %d map points [orbx_distinctive_descriptors]: %d observations" % (len(sizes), len(dset)),
                     "MapPoint.cc:329-403, LocalMapping.cc:290",
                     lambda: mb.DistinctiveDescriptors(dset, set_ptr), lambda: ob.distinctive_descriptors(dset, set_ptr), lambda g, c: np.array_equal(g, c)))
    # (f)3: Frame::isInFrustum for the local map (Tracking::SearchLocalPoints, Tracking.cc:3360-3380): 10 000 map points against one pose
    a_, b_, c_ = 0.11, -0.07, 0.05
    Rx = np.array([[1, 0, 0], [0, np.cos(a_), -np.sin(a_)], [0, np.sin(a_), np.cos(a_)]])
    Ry = np.array([[np.cos(b_), 0, np.sin(b_)], [0, 1, 0], [-np.sin(b_), 0, np.cos(b_)]])
    Rz = np.array([[np.cos(c_), -np.sin(c_), 0], [np.sin(c_), np.cos(c_), 0], [0, 0, 1]])
    Rcw = (Rz @ Ry @ Rx).astype(np.float32)
    tcw = np.array([0.3, -0.2, 0.1], np.float32)
    Ow = (-(Rcw.astype(np.float64).T @ tcw.astype(np.float64))).astype(np.float32)
    camf = (458.654, 457.296, 367.215, 248.375, 47.9)
    bnd = np.array([-10.5, 760.25, -8.0, 488.5], np.float32)
    nfp = 1000 if small else N_MAPPOINTS
    posf = rng.uniform(-6, 6, (nfp, 3)).astype(np.float32)
    posf[:, 2] = rng.uniform(-2, 12, nfp)
    towards = Ow[None, :] - posf
    towards /= np.linalg.norm(towards, axis=1, keepdims=True)
    nrm = (towards + rng.normal(0, 0.4, (nfp, 3))).astype(np.float32)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    dist = np.linalg.norm(posf - Ow, axis=1)
    mxd = (dist * rng.uniform(0.7, 4.0, nfp)).astype(np.float32)
    mnd = (mxd / rng.uniform(1.5, 4.3, nfp)).astype(np.float32)
    lsf = np.float32(np.log(1.2))
    keysf = ("in_view", "proj_x", "proj_y", "proj_xr", "depth", "level", "view_cos")

    def same_frustum(g, c):
        iv = c["in_view"].astype(bool)
        ok = np.array_equal(g["in_view"], c["in_view"]) and g["proj_x"].tobytes() == c["proj_x"].tobytes() and g["proj_y"].tobytes() == c["proj_y"].tobytes()
        for k in ("proj_xr", "depth", "view_cos"):
            ok = ok and g[k][iv].tobytes() == c[k][iv].tobytes()
        lv_d = np.nonzero(g["level"][iv] != c["level"][iv])[0]   # PredictScale goes through logf: may differ only at an integer boundary
        return bool(ok and len(lv_d) <= 2)
    out.append(entry("Frame::isInFrustum x %d map points [orbx_is_in_frustum]: one pose, pinhole" % nfp, "Frame.cc:512-586, Tracking.cc:3360-3380",
                     lambda: mb.isInFrustum(camf[:4] + (0, 0, 0, 0, 0, camf[4]), (Rcw, tcw, Ow), bnd, lsf, 8, 0.5, posf, nrm, mnd, mxd),
                     lambda: ob.is_in_frustum(Rcw, tcw, Ow, camf, bnd, lsf, 8, 0.5, posf, nrm, mnd, mxd), same_frustum))
    # the fisheye rig's form (TUM-VI): both cameras, KannalaBrandt8::project
    kbl = np.array((190.978477, 190.973307, 254.931706, 256.897442, 0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367), np.float32)
    kbr = np.array((190.442369, 190.434438, 252.598164, 254.917230, 0.0034003171, 0.0017669271, -0.0026631290, 0.0003299517), np.float32)
    Rrl = np.eye(3, dtype=np.float32)
    trl = np.array([-0.101, 0.002, 0.001], np.float32)
    views = [(Rcw, tcw, Ow, kbl), ((Rrl @ Rcw).astype(np.float32), (Rrl @ tcw + trl).astype(np.float32), (Rcw.T @ (-trl) + Ow).astype(np.float32), kbr)]
    bndf = np.array([0.0, 512.0, 0.0, 512.0], np.float32)

    def same_checks(g, c):
        ok = True
        for r in (0, 1):
            both = (g["in_view"][r] & c[r]["in_view"]).astype(bool)
            ok = ok and int((g["in_view"][r] != c[r]["in_view"]).sum()) <= 2
            for k in ("proj_x", "proj_y"):   # the azimuth's cos / sin are double: one float ulp allowed (orbx.h)
                ok = ok and int(np.abs(g[k][r][both].view(np.int32).astype(np.int64) - c[r][k][both].view(np.int32).astype(np.int64)).max(initial=0)) <= 1
            ok = ok and g["depth"][r][both].tobytes() == c[r]["depth"][both].tobytes() and g["view_cos"][r][both].tobytes() == c[r]["view_cos"][both].tobytes()
        return bool(ok)
    out.append(entry("Frame::isInFrustumChecks x %d map points x 2 cameras [orbx_is_in_frustum_checks]: fisheye rig, KannalaBrandt8::project" % nfp,
                     "Frame.cc:577-590, 1168-1240, KannalaBrandt8.cpp:67-85",
                     lambda: mb.isInFrustumChecks(views, bndf, lsf, 8, 0.5, posf, nrm, mnd, mxd),
                     lambda: [ob.is_in_frustum_checks(v, bndf, lsf, 8, 0.5, posf, nrm, mnd, mxd) for v in views], same_checks))
    return out


def other_workload_child(wl, a):
    """`python bench.py --workload wl` (same K / W, parity check on, no CPU baseline, no PMC pass) as a child process; the fields a reader of
    the euroc line needs."""
    cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", wl, "--steps", str(a.steps), "--warmup", str(a.warmup), "--cpu-frames", "0",
           "--no-pmc", "--gpus", "1"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ORBX_BENCH_RANK_PROCESS"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=420)
        line = next((l for l in reversed(r.stdout.strip().splitlines()) if l.startswith("{")), None)
        if r.returncode != 0 or not line:
            return {"error": f"child exit {r.returncode}: {r.stderr[-300:]}"}
        d = json.loads(line)
        rf = d.get("roofline") or {}
        return {"metric": d["metric"], "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                "config": d["config"], "parity_checked": d.get("parity_checked"), "repeats": d.get("repeats"),
                "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "frac", "extract_all_kernels_frac")}, "data": d.get("data")}
    except Exception as e:   # the euroc line stands on its own
        return {"error": str(e)[:300]}


def verify_euroc(frames, hs, n_check, W, H, NF, ex):
    """Compare the delivered keypoints / descriptors of frames and the match vectors of consecutive pairs with the CPU oracle (bit-exact).
    n_check >= 4 (the default): EVERY frame and EVERY pair of the step, the oracle frame-parallel on the host cores (ctypes releases the
    GIL); smaller values sample.  Raises on any difference."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_binding as ob
    B = len(frames)
    full = n_check >= 4
    fset = list(range(B)) if full else sorted(set(f for s0 in np.linspace(0, B - 2, n_check) for f in (int(s0), int(s0) + 1)))
    pairs = list(range(B - 1)) if full else sorted(set(int(s0) for s0 in np.linspace(0, B - 2, n_check)))
    sf = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA).tables()["scale"]
    nthreads = max(1, min(os.cpu_count() or 1, 64, len(fset)))

    def extract_chunk(fs):
        oex = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
        return [(f,) + tuple(oex.extract(frames[f], lap=(0, 1000))) for f in fs]

    with ThreadPoolExecutor(nthreads) as pool:
        res = {f: (mono, k, d) for chunk in pool.map(extract_chunk, [fset[i::nthreads] for i in range(nthreads)]) for f, mono, k, d in chunk}
        for f in fset:
            mono, k, d = res[f]
            n = int(hs.cnt[f])
            if n != len(k) or int(hs.mono[f]) != mono:
                raise SystemExit(f"PARITY FAILURE: frame {f}: {n} keypoints / mono {int(hs.mono[f])}, oracle {len(k)} / {mono}")
            if hs.kps[f, :n].numpy().tobytes() != k.tobytes():
                raise SystemExit(f"PARITY FAILURE: frame {f}: keypoints differ from the oracle")
            if not np.array_equal(hs.desc[f, :n].numpy(), d):
                raise SystemExit(f"PARITY FAILURE: frame {f}: descriptors differ from the oracle")

        def match_pair(s0):
            (_, k0, d0), (_, k1, d1) = res[s0], res[s0 + 1]
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
                     desc=d0, has_obs=np.ones(len(k0), np.uint8))
            grid = ob.OracleGrid(k1, 0.0, float(W), 0.0, float(H))
            on, ocm = ob.search_by_projection_frame(grid, d1, sf, q, 15.0, 0, True, None, None)
            return s0, on, ocm, len(k1)

        for s0, on, ocm, n1 in pool.map(match_pair, pairs):
            if int(hs.nm[s0 + 1]) != on or not np.array_equal(hs.match[s0 + 1, :n1].numpy(), ocm):
                raise SystemExit(f"PARITY FAILURE: match vector of frame pair ({s0}, {s0 + 1}) differs from the oracle")
    return {"frames": len(fset), "match_pairs": len(pairs), "of_frames": B, "oracle_threads": nthreads}


def bench_kitti(R):
    """BASELINE config 3: KITTI-shaped 1241x376 rectified stereo, nFeatures=2000: left + right extraction and
    Frame::ComputeStereoMatches (Hamming row band + SAD + median rejection) all on the device, results to the host."""
    a, torch = R.args, R.torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H, NF, _, LAP = WORKLOADS["kitti"]
    B = a.batch or WORKLOADS["kitti"][3]
    from orb_slam3_amd import dataset
    data = "synthetic"
    if dataset.dataset_dir("kitti"):
        pairs = dataset.load_stereo("kitti", B, W, H)
        data = f"dataset: {dataset.dataset_dir('kitti')} (first {B} pairs)"
    else:
        canvas = synth.make_canvas(30 + R.rank, size=2600, n_shapes=4000)
        pairs = [synth.make_stereo_pair(30 + R.rank, t, W, H, canvas) for t in range(B)]
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    torch.cuda.synchronize()
    exl = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=R.device)
    exr = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=R.device)
    cap = exl.output_capacity(W, H)
    exr.output_capacity(W, H)
    bf, b = 0.53716 * 718.856, 0.53716
    class HostSet:   # pinned host destinations of one step: both cameras' features + the stereo result
        def __init__(self):
            self.cam = {k: [pinned(torch, (B, cap, 28), torch.uint8), pinned(torch, (B, cap, 32), torch.uint8),
                            pinned(torch, (B,), torch.int32).zero_(), pinned(torch, (B,), torch.int32).zero_()] for k in "lr"}
            self.ur, self.depth = pinned(torch, (B, cap), torch.float32), pinned(torch, (B, cap), torch.float32)
            self.nm = pinned(torch, (B,), torch.int32).zero_()
    sets = [HostSet(), HostSet()]   # double-buffered: the results of step i travel while step i + 1 is extracted

    def enqueue(i):
        hs = sets[i % 2]
        exl.extract_batch_device(dl.data_ptr(), B, W, H, W, W * H, LAP)
        exr.extract_batch_device(dr.data_ptr(), B, W, H, W, W * H, LAP)
        exl.stereo_batch_device(exr, bf, b)
        for e, k in ((exl, "l"), (exr, "r")):
            e.download_async(*[t.data_ptr() for t in hs.cam[k]])
        exl.stereo_download_async(hs.ur.data_ptr(), hs.depth.data_ptr(), hs.nm.data_ptr())

    def run(nsteps):
        """nsteps pipelined steps, two in flight (as the EuRoC loop); returns the number of features delivered to the host."""
        f = 0
        for i in range(nsteps + 1):
            if i < nsteps:
                enqueue(i)
            if i >= 1:
                exl.download_wait()
                exr.download_wait()
                exl.stereo_download_wait()
                STAMPS.append(time.perf_counter())
                hs = sets[(i - 1) % 2]
                f += int(hs.cam["l"][2].sum()) + int(hs.cam["r"][2].sum())
        return f

    settle(run, a)

    gaps = []

    def region():
        t0 = R.timed_begin([exl, exr], warm=lambda: run(max(a.warmup, 1)))
        STAMPS.clear()
        f_r = run(a.steps)
        out = R.reduce(R.timed_end(t0, [exl, exr]), f_r), f_r
        gaps.append(region_gaps())
        return out

    (dt_max, feats_all), feats = region()
    per_rank = R.per_rank
    regions = [(dt_max, feats_all)] + [region()[0] for _ in range(max(a.repeat, 1) - 1)]
    last = sets[(a.steps - 1) % 2]
    host, h_ur, h_depth, h_nm = last.cam, last.ur, last.depth, last.nm   # the last step's results: what the parity check reads
    R.per_rank = per_rank

    parity = None
    if R.rank == 0 and a.verify > 0:
        from oracle import oracle_binding as ob
        from concurrent.futures import ThreadPoolExecutor
        tb = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA).tables()
        # a.verify >= 4 (the default): EVERY pair of the delivered step, pair-parallel on the host cores (ctypes releases the GIL)
        fset = list(range(B)) if a.verify >= 4 else sorted(set(int(x) for x in np.linspace(0, B - 1, a.verify)))
        nthreads = max(1, min(os.cpu_count() or 1, 64, len(fset)))

        def check_chunk(fs):
            oel = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
            oer = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
            for f in fs:
                res = []
                for oe, img, k in ((oel, pairs[f][0], "l"), (oer, pairs[f][1], "r")):
                    _, kk, dd = oe.extract(img, lap=LAP)
                    n = int(host[k][2][f])
                    if n != len(kk) or host[k][0][f, :n].numpy().tobytes() != kk.tobytes() or not np.array_equal(host[k][1][f, :n].numpy(), dd):
                        return f"stereo pair {f} ({k}): extraction differs from the oracle"
                    res.append((kk, dd))
                pl = [np.ascontiguousarray(oel.level_padded(l)[19:-19, 19:-19]) for l in range(NLEVELS)]
                pr = [np.ascontiguousarray(oer.level_padded(l)[19:-19, 19:-19]) for l in range(NLEVELS)]
                on, our, odepth, _, _ = ob.compute_stereo_matches(res[0][0], res[0][1], res[1][0], res[1][1], tb["scale"], tb["inv_scale"], pl, pr, bf, b)
                n = len(res[0][0])
                if int(h_nm[f]) != on or h_ur[f, :n].numpy().tobytes() != our.tobytes() or h_depth[f, :n].numpy().tobytes() != odepth.tobytes():
                    return f"stereo pair {f}: mvuRight / mvDepth differ from the oracle"
            return None

        with ThreadPoolExecutor(nthreads) as pool:
            for err in pool.map(check_chunk, [fset[i::nthreads] for i in range(nthreads)]):
                if err:
                    raise SystemExit("PARITY FAILURE: " + err)
        parity = {"pairs": len(fset), "of_pairs": B, "oracle_threads": nthreads}

    roofline, kernels = None, {}
    if R.rank == 0 and not a.no_profile:
        passes = 3
        exl.profile_enable(True)
        for _ in range(passes):
            exl.extract_batch_device(dl.data_ptr(), B, W, H, W, W * H, LAP)
            exl.sync()
        prof = exl.profile_read()
        exl.profile_enable(False)
        n_feat = int(host["l"][2].sum())
        samp = list(range(0, B, max(1, B // 8)))
        n_cand = int(sum(len(exl.debug_candidates(l, f)) for f in samp for l in range(NLEVELS)) * B / len(samp))
        sizes = [exl.level_size(l, (W, H)) for l in range(NLEVELS)]
        roofline, kernels = roofline_from_profile(exl, prof, passes, sizes, B, n_feat, n_cand)
        roofline["note"] = "left extractor's launch (the right one is identical in shape); stereo kernels are not in this table"

    cpu = None
    if R.rank == 0 and R.world == 1 and a.cpu_frames > 0:
        cpu = cpu_baseline_kitti(pairs, max(2, a.cpu_frames // 6), W, H, NF, bf, b)

    out = base_line(R, "ORB kfeatures/sec extract+match, KITTI 1241x376 stereo nFeatures=2000", feats_all / dt_max / 1e3, dt_max,
                    {"workload": "KITTI-shaped 1241x376 rectified stereo, nFeatures=2000: left+right extract + ComputeStereoMatches "
                                 "(row-band Hamming, SAD sub-pixel, median rejection) on device + D2H; inputs resident in HBM",
                     "pairs_per_step_per_gpu": B, "sequences": R.world, "features_per_pair": round(feats / a.steps / B, 1),
                     "stereo_matches_per_pair": round(float(h_nm.sum()) / B, 1)})
    out["data"] = data
    out.update({"roofline": roofline, "cpu_baseline": cpu, "parity_checked": parity, "kernels": kernels, "repeats": repeats_block(regions, a.steps, gaps)})
    R.finish(out)


def make_mappoints(rng, kps_list, desc_list, t, n_mp):
    """map points = features of the up to 8 previous frames with each descriptor bit flipped w.p. 0.04,
    projected at the source keypoint position + N(0, 2 px), predicted level = source octave, viewCos uniform [0.9, 1]."""
    src = [s for s in range(max(0, t - 8), t)] or [t]
    k = np.concatenate([kps_list[s] for s in src])
    d = np.concatenate([desc_list[s] for s in src])
    idx = rng.integers(0, len(k), n_mp)
    k, d = k[idx], d[idx].copy()
    flips = rng.random((n_mp, 256)) < 0.04
    d ^= np.packbits(flips, axis=1, bitorder="little")
    return dict(proj_x=(k["x"] + rng.normal(0, 2, n_mp)).astype(np.float32), proj_y=(k["y"] + rng.normal(0, 2, n_mp)).astype(np.float32),
                proj_xr=np.zeros(n_mp, np.float32), level=k["octave"].astype(np.int32),
                view_cos=rng.uniform(0.9, 1.0, n_mp).astype(np.float32), desc=d,
                in_view=np.ones(n_mp, np.uint8), has_obs=np.ones(n_mp, np.uint8))


def bench_tumvi(R):
    """BASELINE config 4: TUM-VI-shaped 1024x1024, nFeatures=1500: extract + SearchByProjection(Frame, MapPoints) (M1,
    ORBmatcher.cc:39-141 as called from Tracking.cc:3390-3413) against 10,000 map points per frame, all on the device."""
    a, torch = R.args, R.torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H, NF, _, LAP = WORKLOADS["tumvi"]
    B = a.batch or WORKLOADS["tumvi"][3]
    seed = 40 + R.rank
    canvas = synth.make_canvas(seed, size=2600, n_shapes=4000)
    yy, xx = np.mgrid[0:H, 0:W]
    vign = (1.0 - 0.45 * (((xx - W / 2) ** 2 + (yy - H / 2) ** 2) / (W * W / 2.0))).astype(np.float32)   # radial vignetting
    from orb_slam3_amd import dataset
    data = "synthetic"
    if dataset.dataset_dir("tumvi"):
        frames = dataset.load_mono("tumvi", B, W, H, start=R.rank * B)
        data = f"dataset: {dataset.dataset_dir('tumvi')} (first {B} frames per rank)"
    else:
        frames = np.stack([np.clip(np.rint(synth.frame_from_canvas(canvas, t, W, H, 1000 * seed + t).astype(np.float32) * vign), 0, 255).astype(np.uint8)
                           for t in range(B)])
    d_frames = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ex = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7, device=R.device)
    cap = ex.output_capacity(W, H)
    # map points from the frames' own features (one untimed extraction), resident on the device like the local map's descriptors
    ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, LAP)
    feats0 = [ex.download(f) for f in range(B)]
    rng = np.random.default_rng(5000 + seed)
    mps = [make_mappoints(rng, [x[1] for x in feats0], [x[2] for x in feats0], f, N_MAPPOINTS) for f in range(B)]
    dev = {k: torch.from_numpy(np.stack([m[k] for m in mps])).cuda() for k in ("proj_x", "proj_y", "level", "view_cos", "desc", "in_view")}
    torch.cuda.synchronize()
    hs = [dict(kps=pinned(torch, (B, cap, 28), torch.uint8), desc=pinned(torch, (B, cap, 32), torch.uint8), cnt=pinned(torch, (B,), torch.int32).zero_(),
               mono=pinned(torch, (B,), torch.int32).zero_(), match=pinned(torch, (B, cap), torch.int32), nm=pinned(torch, (B,), torch.int32).zero_())
          for _ in range(2)]

    def enqueue(i):
        h = hs[i % 2]
        ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, LAP)
        ex.search_mappoints_batch_device(N_MAPPOINTS, dev["proj_x"].data_ptr(), dev["proj_y"].data_ptr(), dev["level"].data_ptr(),
                                         dev["view_cos"].data_ptr(), dev["in_view"].data_ptr(), dev["desc"].data_ptr(), th=1.0, nnratio=0.8)
        ex.download_async(h["kps"].data_ptr(), h["desc"].data_ptr(), h["cnt"].data_ptr(), h["mono"].data_ptr(), h["match"].data_ptr(), h["nm"].data_ptr())

    def run(nsteps):
        feats = 0
        for i in range(nsteps + 1):
            if i < nsteps:
                t = time.perf_counter()
                enqueue(i)
                host_enqueue[0] += time.perf_counter() - t
            if i >= 1:
                ex.download_wait()
                STAMPS.append(time.perf_counter())
                feats += int(hs[(i - 1) % 2]["cnt"].sum())
        return feats

    host_enqueue = [0.0]
    settle(run, a)

    gaps = []

    def region():
        t0 = R.timed_begin([ex], warm=lambda: run(max(a.warmup, 1)))
        host_enqueue[0] = 0.0
        STAMPS.clear()
        f_r = run(a.steps)
        out = R.reduce(R.timed_end(t0, [ex]), f_r), f_r
        gaps.append(dict(region_gaps(), host_enqueue_ms_per_step=round(host_enqueue[0] / a.steps * 1e3, 3)))
        return out

    (dt_max, feats_all), feats = region()
    per_rank = R.per_rank
    enq = host_enqueue[0]
    last = hs[(a.steps - 1) % 2]
    # Every region processes the same frames, so the last region's final step is what the parity check reads.  (Rounds 2-3 cloned the first region's
    # delivered step here -- 15 MB of pinned result buffers into fresh pageable memory -- and THAT was TUM-VI's "slow mode": in two of three processes
    # the device stalled once for 20 - 60 ms about 25 ms later, inside the second region; without the clone, never: profiles/r04_m_*.)
    regions = [(dt_max, feats_all)] + [region()[0] for _ in range(max(a.repeat, 1) - 1)]
    R.per_rank = per_rank
    host_enqueue[0] = enq
    last = hs[(a.steps - 1) % 2]

    parity = None
    if R.rank == 0 and a.verify > 0:
        from oracle import oracle_binding as ob
        from concurrent.futures import ThreadPoolExecutor
        sf = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA).tables()["scale"]
        # a.verify >= 4 (the default): EVERY frame of the delivered step, frame-parallel on the host cores
        fset = list(range(B)) if a.verify >= 4 else sorted(set(int(x) for x in np.linspace(0, B - 1, a.verify)))
        nthreads = max(1, min(os.cpu_count() or 1, 64, len(fset)))

        def check_chunk(fs):
            oex = ob.OracleExtractor(NF, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
            for f in fs:
                mono, k, d = oex.extract(frames[f], lap=LAP)
                n = int(last["cnt"][f])
                if n != len(k) or last["kps"][f, :n].numpy().tobytes() != k.tobytes() or not np.array_equal(last["desc"][f, :n].numpy(), d):
                    return f"frame {f}: extraction differs from the oracle"
                grid = ob.OracleGrid(k, 0.0, float(W), 0.0, float(H))
                on, ofm = ob.search_by_projection_mappoints(grid, d, sf, mps[f], 1.0, 0.8)
                if int(last["nm"][f]) != on or not np.array_equal(last["match"][f, :n].numpy(), ofm):
                    return f"frame {f}: SearchByProjection(map points) differs from the oracle"
            return None

        with ThreadPoolExecutor(nthreads) as pool:
            for err in pool.map(check_chunk, [fset[i::nthreads] for i in range(nthreads)]):
                if err:
                    raise SystemExit("PARITY FAILURE: " + err)
        parity = {"frames": len(fset), "of_frames": B, "oracle_threads": nthreads}

    roofline, kernels = None, {}
    if R.rank == 0 and not a.no_profile:
        passes = 3
        ex.profile_enable(True)
        for _ in range(passes):
            ex.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, LAP)
            ex.sync()
        prof = ex.profile_read()
        ex.profile_enable(False)
        n_feat = int(last["cnt"].sum())
        samp = list(range(0, B, max(1, B // 8)))
        n_cand = int(sum(len(ex.debug_candidates(l, f)) for f in samp for l in range(NLEVELS)) * B / len(samp))
        sizes = [ex.level_size(l, (W, H)) for l in range(NLEVELS)]
        roofline, kernels = roofline_from_profile(ex, prof, passes, sizes, B, n_feat, n_cand)

    cpu = None
    if R.rank == 0 and R.world == 1 and a.cpu_frames > 0:
        cpu = cpu_baseline_tumvi(frames, lambda f: mps[f], max(2, a.cpu_frames // 6), W, H, NF)

    out = base_line(R, "ORB kfeatures/sec extract+match, TUM-VI 1024x1024 nFeatures=1500", feats_all / dt_max / 1e3, dt_max,
                    {"workload": "TUM-VI-shaped 1024x1024 mono, nFeatures=1500: extract + SearchByProjection(Frame, MapPoints) against "
                                 f"{N_MAPPOINTS} map points per frame (th=1, nnratio=0.8) + D2H; inputs resident in HBM",
                     "frames_per_step_per_gpu": B, "sequences": R.world, "features_per_frame": round(feats / a.steps / B, 1),
                     "map_point_matches_per_frame": round(float(last["nm"].sum()) / B, 1)})
    out["data"] = data
    out.update({"roofline": roofline, "cpu_baseline": cpu, "parity_checked": parity, "kernels": kernels,
                "host_enqueue_ms_per_step": round(host_enqueue[0] / a.steps * 1e3, 3), "repeats": repeats_block(regions, a.steps, gaps)})
    R.finish(out)


def pmc_child(args):
    """A few serialized steps of the workload's extraction + match for a rocprofv3 --pmc pass (no timing, no output)."""
    import torch
    import orb_slam3_amd as osa
    from orb_slam3_amd import synth
    W, H, NF, Bd, LAP = WORKLOADS[args.workload]
    B = args.batch or Bd
    canvas = synth.make_scene_canvas(args.scene, 10)
    frames = np.stack([synth.frame_from_canvas(canvas, t, W, H, 10000 + t) for t in range(B)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    ex = osa.ORBextractor(NF, 1.2, NLEVELS, 20, 7)
    for _ in range(3):   # the candidate queues as the bench's loop ends up with them (orbx_tune_fast_queues)
        ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, LAP)
        if not ex.tune_fast_queues(1)["changed"]:
            break
    for _ in range(args.warmup + args.steps):
        ex.extract_batch_device(d.data_ptr(), B, W, H, W, W * H, LAP)
        if args.workload == "euroc":
            ex.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        ex.sync()


# ---------------------------------------------------------------------------------------------------------
# launcher: python bench.py --gpus N starts the N rank processes itself
# ---------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_once(argv, n):
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ORBX_BENCH_RANK_PROCESS="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py")] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    return rcs, out0 or ""


def launcher(args, argv):
    n = args.gpus
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    attempts, history = 0, []
    while True:
        attempts += 1
        rcs, out0 = launch_once(argv, n)
        line = next((l for l in reversed(out0.strip().splitlines()) if l.startswith("{")), None)
        if all(rc == 0 for rc in rcs) and line:
            out = json.loads(line)
            out["attempts"] = attempts
            if history:
                out["failed_attempts"] = history   # reported, never hidden: an attempt that died is a defect to chase
            print(json.dumps(out), flush=True)
            return 0
        history.append({"rank_exit_codes": rcs})
        sys.stderr.write(f"bench.py: attempt {attempts} failed (rank exit codes {rcs})\n")
        if attempts >= 1 + args.retries:
            return 1


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--reference-build-child":
        return reference_build_child(sys.argv[2], int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--settle", type=int, default=8,
                    help="untimed pipeline steps run once before the W warm-up steps (reported as settle_steps): in a fresh process the HIP "
                         "runtime's first ~10 batches enqueue 5x slower (0.5 instead of 0.1 ms of host time per step, its signal and "
                         "command pools still growing), and with a short warm-up that start-up transient landed in the timed region "
                         "(TUM-VI workload, warm-up 3: 1.29-1.31 ms per step; warm-up 10: 1.07)")
    ap.add_argument("--settle-seconds", dest="settle_seconds", type=float, default=0.25, help="... and at least this long (see settle())")
    ap.add_argument("--batch", type=int, default=0, help="frames (stereo pairs) per step per GPU; 0 = the workload's default (256 / 128 / 128, SURVEY.md 8d)")
    ap.add_argument("--cpu-frames", type=int, default=384, help="frames in the CPU baseline sample (0 = skip); 384 = about 13 s of one core for euroc")
    ap.add_argument("--verify", type=int, default=4, help="parity of the last timed step against the CPU oracle on rank 0: >= 4 = every frame and every "
                                                          "match vector of the step (euroc), 1-3 = that many sampled pairs, 0 = skip")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--repeat", type=int, default=5, help="timed regions of K steps each (the line's value is the first; `repeats` reports median / min / max)")
    ap.add_argument("--input-sets", dest="input_sets", type=int, default=3,
                    help="euroc workload: resident input sets of `batch` frames used in rotation by the timed loop (3 x 92 MB > the 256 MB Infinity Cache)")
    ap.add_argument("--latency", type=int, default=200, help="single-frame orbx_extract calls timed for the `latency` block (0 = skip)")
    ap.add_argument("--other-workloads", dest="other_workloads", action="store_true", default=None,
                    help="also run the kitti and tumvi workloads as child processes and embed their results (default for the euroc workload at --gpus 1)")
    ap.add_argument("--no-other-workloads", dest="other_workloads", action="store_false")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=True,
                    help="measure roofline.traffic in this run: two extra rocprofv3 --pmc child passes of a few steps, about 40 s (default at --gpus 1)")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="skip the rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--retries", type=int, default=1, help="launcher: re-run the whole job this many times if a rank process dies (reported in the JSON)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="euroc",
                    help="euroc = BASELINE metric config; kitti = config 3 (stereo); tumvi = config 4 (map-point projection search)")
    ap.add_argument("--share-gpus", dest="share_gpus", action="store_true",
                    help="ranks beyond the visible GPUs share them (rank r on GPU r mod count): exercises the N-rank path on a box with fewer GPUs")
    ap.add_argument("--ablate", default="", help="DIAGNOSTIC for the euroc workload: comma list of nodl (no D2H of keypoints / descriptors / matches), nomatch "
                                                 "(no frame-to-frame matcher); the line is marked and no parity check runs")
    ap.add_argument("--scene", default="quads",
                    help="euroc workload, synthetic frames: quads = SURVEY.md 8(d)'s scene (the metric); texture = the stress scene with natural-image "
                         "statistics and 13 x the FAST candidates; blend:<a> (0..1) = (1 - a) quads + a texture, densities in between "
                         "(whole-step parity as usual; stage_stats_last_step says which paths it took; tools/density_sweep.sh)")
    ap.add_argument("--threads", action="store_true",
                    help="SURVEY.md 8(e) as ONE process: --gpus N host worker threads (thread s on GPU s mod the visible GPUs), each with its own extractor "
                         "and pinned ring, instead of N rank processes; euroc workload, device-resident clock, parity of every thread's last step")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.other_workloads is None:
        args.other_workloads = args.workload == "euroc" and args.cpu_frames > 0 and not args.no_profile   # the full default run only
    if args.pmc_child:
        return pmc_child(args)
    if args.threads:
        return bench_threads(args)

    if "WORLD_SIZE" not in os.environ:
        return launcher(args, sys.argv[1:])
    if int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts the rank processes itself)")
    R = Rank(args)
    if R.dry:
        return bench_dry(R)
    return {"euroc": bench_euroc, "kitti": bench_kitti, "tumvi": bench_tumvi}[args.workload](R)


if __name__ == "__main__":
    sys.exit(main() or 0)
