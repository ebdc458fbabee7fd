#!/usr/bin/env python3
"""bench.py -- ORB kfeatures/sec, extract + match, EuRoC 752x480 nFeatures=1000 (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic EuRoC-shaped frames of ONE camera sequence:
  extract (pyramid, FAST, quad-tree, orientation, blur, rBRIEF) of B frames resident in HBM
  + frame-to-frame SearchByProjection (th=15, rotation check) between consecutive frames of the batch
  + D2H of keypoints, descriptors, counts and match indices into pinned host memory.
Frames are already resident in HBM when the timed region starts.  One process per GPU; ranks shard
independent sequences (seed = 10 + rank), no data-path collective; torch.distributed (RCCL)
is used only for the timing barrier and the max-over-ranks reduction.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W, H, NFEATURES, NLEVELS = 752, 480, 1000, 8
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def level_sizes(ex, w, h):
    return [ex.level_size(l, (w, h)) for l in range(NLEVELS)]


def algorithmic_bytes(sizes, n_frames, feats, cands):
    """Per-kernel algorithmic bytes of one launch (SURVEY.md 8d / DESIGN.md): every stage reads its input once
    and writes its output once.  P = sum of level pixels, N = keypoints out, C = FAST candidates."""
    px = [a * b for a, b in sizes]
    P = sum(px)
    N, Cn = feats, cands
    per = {
        "k_pyr_base": 2 * px[0] * n_frames,                       # image read + level-0 write
        "k_pyr_resize": ((P - px[-1]) + (P - px[0])) * n_frames / (NLEVELS - 1),  # per launch (7 launches)
        "k_fast_wave": P * n_frames + 4 * Cn,                     # pyramid read + packed candidates
        "k_blur": 2 * P * n_frames,
        "k_octree": 8 * Cn + 4 * N,                               # candidates read + gathered, keypoints out
        "k_finalize": 16 * N,
        "k_describe": N * (749 + 512 + 32 + 28),
        "k_window_best2": (32 + 28) * 2 * N + 16 * N,
        "k_greedy_resolve": 16 * N + 4 * N,
    }
    extract_total = (5 * P - px[0] - px[-1]) * n_frames + 12 * Cn + 1321 * N  # B_extract of SURVEY.md 8d
    return per, extract_total


def reference_build_child(td, n_ref):
    """Child process of cpu_baseline: the compiled reference (oracle/_ref) on the frames saved in td; prints one JSON object."""
    from oracle import oracle_binding as ob
    from oracle import ref_binding as rb
    frames = np.load(os.path.join(td, "frames.npy"))
    sf = ob.OracleExtractor(NFEATURES, 1.2, NLEVELS, 20, 7).tables()["scale"]
    rex = rb.RefExtractor(NFEATURES, 1.2, NLEVELS, 20, 7)
    t0 = time.perf_counter()
    feats, prev = 0, None
    for t in range(n_ref):
        _, k, d = rex.extract(frames[t % len(frames)], (0, 1000))
        feats += len(k)
        if prev is not None:
            k0, d0 = prev
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, z=np.ones(len(k0), np.float32), octave=k0["octave"], angle=k0["angle"],
                     desc=d0, has_obs=np.ones(len(k0), np.uint8))
            rb.ref_search_by_projection_frame(rb.RefFrame(k, d, 0.0, float(W), 0.0, float(H), sf), q, 15.0, 0, True)
        prev = (k, d)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1,
                      "sample": f"{n_ref} frames, {dt:.1f} s, reference ORBextractor.cc + ORBmatcher.cc over oracle/ocv_shim"}))


def cpu_baseline(frames, n_sample):
    """The oracle (CPU restatement of the reference path) timed on this box's host cores, 1 thread."""
    from oracle import oracle_binding as ob
    oex = ob.OracleExtractor(NFEATURES, 1.2, NLEVELS, 20, 7, flags=ob.FLAG_DESC_FMA)
    sf = oex.tables()["scale"]
    t0 = time.perf_counter()
    feats = 0
    prev = None
    for t in range(n_sample):
        _, k, d = oex.extract(frames[t % len(frames)], lap=(0, 1000))
        feats += len(k)
        if prev is not None:
            k0, d0 = prev
            q = dict(u=k0["x"] - 2.0, v=k0["y"] - 1.0, ur=np.zeros(len(k0), np.float32), octave=k0["octave"],
                     angle=k0["angle"], desc=d0, has_obs=np.ones(len(k0), np.uint8))
            grid = ob.OracleGrid(k, 0.0, float(W), 0.0, float(H))
            ob.search_by_projection_frame(grid, d, sf, q, 15.0, 0, True)
        prev = (k, d)
    dt = time.perf_counter() - t0
    out = {"value": round(feats / dt / 1e3, 3), "unit": "kfeatures/s", "cores": 1, "kind": "port",
           "sample": f"{n_sample} frames of the same workload (extract + frame-to-frame match), {dt:.1f} s, "
                     f"oracle/ C++ restatement -O3 x86-64-v3, host CPU {os.cpu_count()} logical cores available"}
    # beside it, when oracle/_ref travelled here: the reference's OWN ORBextractor.cc + ORBmatcher.cc (compiled where they lie in the
    # build container against the stand-in OpenCV / SLAM types, whose image primitives are the oracle's scalar ones) on a quarter of
    # the sample -- shows the port is not slower than the code it restates; not a substitute for an OpenCV-backed build.  Runs in
    # a child process: nothing that library does can take the bench line down with it.
    try:
        from oracle import ref_binding as rb
        if rb.available() and rb.matcher_available():
            import subprocess
            import tempfile
            n_ref = max(2, n_sample // 4)
            with tempfile.TemporaryDirectory() as td:
                np.save(os.path.join(td, "frames.npy"), np.ascontiguousarray(frames[:min(n_ref, len(frames))]))
                r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--reference-build-child", td, str(n_ref)],
                                   capture_output=True, text=True, timeout=300)
            out["reference_build"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": f"child exit {r.returncode}"}
    except Exception as e:   # the baseline above stands on its own
        out["reference_build"] = {"error": str(e)[:200]}
    return out


def bench_kitti(args, rank, local_rank, world, torch, dist, osa, synth):
    """BASELINE config 3: KITTI-shaped 1241x376 rectified stereo, nFeatures=2000: left + right extraction and
    Frame::ComputeStereoMatches (Hamming row band + SAD + median rejection) all on the device, results to the host."""
    from orb_slam3_amd import sharding
    w, h, nf, B = 1241, 376, 2000, min(args.batch, 64)
    canvas = synth.make_canvas(30 + rank, size=2600, n_shapes=4000)
    pairs = [synth.make_stereo_pair(30 + rank, t, w, h, canvas) for t in range(B)]
    dl = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dr = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    exl = osa.ORBextractor(nf, 1.2, NLEVELS, 20, 7, device=local_rank)
    exr = osa.ORBextractor(nf, 1.2, NLEVELS, 20, 7, device=local_rank)
    cap = exl.output_capacity(w, h)
    bf, b = 0.53716 * 718.856, 0.53716
    host = {k: [torch.empty((B, cap, 28), dtype=torch.uint8).pin_memory(), torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory(),
                torch.zeros(B, dtype=torch.int32).pin_memory(), torch.zeros(B, dtype=torch.int32).pin_memory()] for k in "lr"}

    def step():
        exl.extract_batch_device(dl.data_ptr(), B, w, h, w, w * h, (0, 0))
        exr.extract_batch_device(dr.data_ptr(), B, w, h, w, w * h, (0, 0))
        exl.stereo_batch_device(exr, bf, b)
        for e, k in ((exl, "l"), (exr, "r")):
            e.download_async(*[t.data_ptr() for t in host[k]])
        exl.download_wait()
        exr.download_wait()
        nm = sum(exl.stereo_download(f)[0] for f in (0, B - 1))   # includes the stream sync for the stereo results
        return int(host["l"][2].sum()) + int(host["r"][2].sum()), nm

    def barrier():
        torch.cuda.synchronize(); exl.sync(); exr.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    feats = 0
    for _ in range(args.steps):
        f, nm = step()
        feats += f
    barrier()
    dt = time.perf_counter() - t0
    dt_max, feats_all = sharding.reduce_throughput(dt, feats, device="cuda")
    if rank == 0:
        print(json.dumps({
            "metric": "ORB kfeatures/sec extract+match, KITTI 1241x376 stereo nFeatures=2000", "value": round(feats_all / dt_max / 1e3, 2),
            "unit": "kfeatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "KITTI-shaped 1241x376 rectified stereo, nFeatures=2000: left+right extract + ComputeStereoMatches "
                                   "(row-band Hamming, SAD sub-pixel, median rejection) on device + D2H", "pairs_per_step_per_gpu": B,
                       "features_per_pair": round(feats / args.steps / B, 1), "stereo_matches_first_last_frame": nm},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--reference-build-child":
        return reference_build_child(sys.argv[2], int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per step (per GPU)")
    ap.add_argument("--cpu-frames", type=int, default=384, help="frames in the CPU baseline sample, cycling over the batch (0 = skip); 384 = about 13 s of one core")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--workload", choices=["euroc", "kitti"], default="euroc",
                    help="euroc = BASELINE metric config (mono extract + frame-to-frame match); kitti = config 3 (stereo extract + ComputeStereoMatches)")
    ap.add_argument("--lanes", type=int, default=1, help="extractor instances alternated over consecutive batches (each has its own streams and workspace)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (liborbx has no CPU path)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import orb_slam3_amd as osa
    from orb_slam3_amd import synth

    if args.workload == "kitti":
        return bench_kitti(args, rank, local_rank, world, torch, dist, osa, synth)

    B = args.batch
    canvas = synth.make_canvas(10 + rank)
    frames = np.stack([synth.frame_from_canvas(canvas, t, W, H, 1000 * (10 + rank) + t) for t in range(B)])
    d_frames = torch.from_numpy(frames).cuda()

    exs = [osa.ORBextractor(NFEATURES, 1.2, NLEVELS, 20, 7, device=local_rank) for _ in range(max(1, args.lanes))]
    ex = exs[0]
    NL = len(exs)
    cap = ex.output_capacity(W, H)
    # pinned host destinations (the Tracking thread's buffers), double-buffered: the D2H of step i overlaps the
    # kernels of step i+1 (copy stream inside liborbx); a step is complete when its results are on the host
    class HostSet:
        def __init__(self):
            self.kps = torch.empty((B, cap, 28), dtype=torch.uint8).pin_memory()
            self.desc = torch.empty((B, cap, 32), dtype=torch.uint8).pin_memory()
            self.cnt = torch.zeros(B, dtype=torch.int32).pin_memory()
            self.mono = torch.zeros(B, dtype=torch.int32).pin_memory()
            self.match = torch.empty((B, cap), dtype=torch.int32).pin_memory()
            self.nm = torch.zeros(B, dtype=torch.int32).pin_memory()
    host = [HostSet() for _ in range(2 * NL)]

    def enqueue(i):
        e, hs = exs[i % NL], host[i % (2 * NL)]
        e.extract_batch_device(d_frames.data_ptr(), B, W, H, W, W * H, (0, 1000))
        if not os.environ.get("ORBX_BENCH_SKIP_MATCH"):   # diagnostic switches, never set for a reported number
            e.match_consecutive_device(th=15.0, du=-2.0, dv=-1.0, check_orientation=True)
        e.download_async(hs.kps.data_ptr(), hs.desc.data_ptr(), hs.cnt.data_ptr(), hs.mono.data_ptr(),
                         hs.match.data_ptr(), hs.nm.data_ptr())

    def run(nsteps):
        """nsteps pipelined steps (two batches in flight per extractor lane, lanes alternate over consecutive batches);
        returns the number of features delivered to the host."""
        feats = 0
        depth = 2 * NL
        for i in range(nsteps + depth - 1):
            if i < nsteps:
                enqueue(i)
            j = i - (depth - 1)           # oldest batch still in flight
            if j >= 0:
                exs[j % NL].download_wait()
                feats += int(host[j % (2 * NL)].cnt.sum())
        return feats

    def step():   # un-pipelined single step (profiling passes)
        return run(1)

    def barrier():
        torch.cuda.synchronize()
        for e in exs:
            e.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(max(args.warmup, 1))
    barrier()
    t0 = time.perf_counter()
    feats = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    nmatch = int(host[(args.steps - 1) % (2 * NL)].nm[1:].sum())
    h_cnt = host[(args.steps - 1) % (2 * NL)].cnt

    from orb_slam3_amd import sharding
    dt_max, feats_all = sharding.reduce_throughput(dt, feats, device="cuda")

    # ---- per-kernel timing with HIP events on the extractor's stream (separate, untimed passes) ----
    roofline, kernels = None, {}
    if rank == 0 and not args.no_profile:
        ex.profile_enable(True)
        for _ in range(3):
            step()
        prof = ex.profile_read()
        ex.profile_enable(False)
        n_feat = int(h_cnt.sum())
        n_cand = 0
        for f in range(0, B, max(1, B // 8)):   # sample the candidate count on a few frames
            n_cand += sum(len(ex.debug_candidates(l, f)) for l in range(NLEVELS))
        n_cand = int(n_cand * B / len(range(0, B, max(1, B // 8))))
        per, extract_total = algorithmic_bytes(level_sizes(ex, W, H), B, n_feat, n_cand)
        tot_ms = 0.0
        for name, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            launches_per_step = cnt / 3.0
            kernels[name] = {"avg_ms": round(ms, 4), "launches_per_step": launches_per_step,
                             "alg_GBs": round(per[name] / (ms * 1e-3) / 1e9, 1)}
            tot_ms += ms * launches_per_step
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"])
        ach = per[dom] / (kernels[dom]["avg_ms"] * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": int(per[dom]), "avg_launch_ms": kernels[dom]["avg_ms"],
                    "kernel_time_share": round(kernels[dom]["avg_ms"] * kernels[dom]["launches_per_step"] / tot_ms, 3),
                    "extract_all_kernels_GBs": round(extract_total / (sum(v["avg_ms"] * v["launches_per_step"] for k, v in kernels.items() if not k.startswith(("k_window", "k_greedy"))) * 1e-3) / 1e9, 1)}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        cpu = cpu_baseline(frames, args.cpu_frames)

    if rank == 0:
        out = {
            "metric": "ORB kfeatures/sec extract+match, EuRoC 752x480 nFeatures=1000",
            "value": round(feats_all / dt_max / 1e3, 2), "unit": "kfeatures/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "EuRoC-shaped 752x480 mono, nFeatures=1000, 8 levels, scale 1.2, FAST 20/7: extract + "
                                   "frame-to-frame SearchByProjection(th=15) + D2H of results",
                       "frames_per_step_per_gpu": B, "sequences": world, "extractor_lanes_per_gpu": NL, "features_per_frame": round(feats / args.steps / B, 1),
                       "matches_per_frame": round(nmatch / max(B - 1, 1), 1), "parallelism": f"{world} independent sequences"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
