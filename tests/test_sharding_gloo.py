"""CPU test of the N>1 host path: world_size 2 over gloo (no GPU): sequence sharding + throughput reduction."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from orb_slam3_amd import sharding
    sharding.init_host_group(rank, world)   # what bench.py's Rank uses for the GPU run as well: gloo, no communicator on the devices
    assert dist.get_backend() == "gloo"
    seqs = sharding.sequences_for_rank(8, world, rank)
    # pretend each sequence takes (1 + rank) seconds and yields 1000 features per frame over 10 frames
    local_t = float(len(seqs)) * (1 + rank)
    local_u = float(len(seqs)) * 10 * 1000
    dist.barrier()
    t, u = sharding.reduce_throughput(local_t, local_u)
    t2, u2, per = sharding.gather_throughput(local_t, local_u)
    assert (t2, u2) == (t, u) and per == [(4.0, 40000.0), (8.0, 40000.0)]   # every rank's (seconds, units): a straggler is visible
    q.put((rank, seqs, t, u))
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    for _, _, t, u in res:
        assert t == 8.0            # max over ranks: rank 1 needs 4 * 2 s
        assert u == 80000.0        # sum over ranks


def test_single_rank_is_identity():
    from orb_slam3_amd import sharding
    assert sharding.sequences_for_rank(3, 1, 0) == [0, 1, 2]
    assert sharding.reduce_throughput(1.5, 42.0) == (1.5, 42.0)


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None) if k not in env_extra else None
    return subprocess.run([sys.executable, str(root / "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus 2` (no torchrun) starts 2 rank processes itself; dry mode: gloo, no device work."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"ORBX_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["config"]["sequences"] == 2 and out["attempts"] == 1
    assert out["process_group"] == "gloo" and [p["rank"] for p in out["per_rank"]] == [0, 1]
    # every rank's parity verdict (here a stand-in object) reaches rank 0 over the host group, in rank order
    assert [p["parity_checked"] for p in out["per_rank"]] == [{"checked_by_rank": 0}, {"checked_by_rank": 1}]
    # rank r reports 1000 * steps * (r + 1) units: the sum over both ranks arrived on rank 0
    assert abs(out["value"] * out["ms_per_step"] * 3 / 1e3 * 1e3 - 9000.0) < 9000.0 * 0.02


def test_bench_share_gpus_flag_runs_more_ranks_than_gpus():
    """--share-gpus (rank r on GPU r mod count: the N-rank path on a box with fewer GPUs, used for profiles/r04_two_ranks_one_gpu.json): accepted by
    launcher and ranks; dry mode has no device, the mapping itself is a GPU-box matter."""
    import json
    r = _run_bench(["--gpus", "3", "--share-gpus", "--steps", "2", "--warmup", "1"], {"ORBX_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 3 and len(out["per_rank"]) == 3


def test_bench_refuses_a_mismatched_world_size():
    """Under an external launcher --gpus must equal WORLD_SIZE: no silent single-GPU run."""
    r = _run_bench(["--gpus", "8"], {"ORBX_BENCH_DRY": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_reports_failed_attempts(tmp_path):
    """A rank that dies makes the launcher retry once and say so; two failures fail the run."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--retries", "0"], {"ORBX_BENCH_DRY": "1", "ORBX_BENCH_DRY_FAIL": "always"})
    assert r.returncode != 0
    flag = tmp_path / "failed_once"
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"ORBX_BENCH_DRY": "1", "ORBX_BENCH_DRY_FAIL": str(flag)})
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["attempts"] == 2 and len(out["failed_attempts"]) == 1


def test_gpu_ranks_use_the_host_process_group_only():
    """The non-dry rank path creates its process group through sharding.init_host_group (gloo) and nothing else: no "nccl" backend, no
    device tensors in the barrier / reduction -- `bench.py --gpus 8` needs nothing but HIP on the GPUs."""
    import re
    from pathlib import Path
    src = (Path(__file__).resolve().parent.parent / "bench.py").read_text()
    rank_cls = src[src.index("class Rank:"):src.index("def base_line(")]
    assert "init_host_group" in rank_cls and "init_process_group" not in rank_cls
    assert not re.search(r"[\"']nccl[\"']", src)
    sh = (Path(__file__).resolve().parent.parent / "orb_slam3_amd" / "sharding.py").read_text()
    assert 'init_process_group("gloo"' in sh and "cuda" not in sh.split("def gather_throughput")[1].split("def reduce_throughput")[0]


def test_rank_core_sets_on_a_fake_eight_gpu_node(tmp_path):
    """bench.py binds a rank to the cores local to its GPU before it allocates pinned memory (never run on an 8-GPU node by the
    builder).  rank_core_set is a pure function of the sysfs tree: on a fake node with two sockets x four GPUs every local rank must get a non-empty
    core set inside its GPU's local_cpulist, the eight sets must be pairwise disjoint, and together they must cover every core; a GPU alone on its
    cores gets all of them; a missing sysfs entry raises (bind_to_gpu_numa_node then reports `unbound`)."""
    import sys
    from pathlib import Path
    import pytest
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    sysfs = tmp_path / "devices"
    lists = ["0-63,128-191"] * 4 + ["64-127,192-255"] * 4
    bdfs = [f"0000:{0x10 + 0x10 * i:02x}:00.0" for i in range(8)]
    for b, l, node in zip(bdfs, lists, [0] * 4 + [1] * 4):
        (sysfs / b).mkdir(parents=True)
        (sysfs / b / "local_cpulist").write_text(l + "\n")
        (sysfs / b / "numa_node").write_text(f"{node}\n")
    sets = [bench.rank_core_set(i, bdfs, str(sysfs)) for i in range(8)]
    for i, (cores, node, text) in enumerate(sets):
        assert len(cores) == 32 and cores <= bench.parse_cpulist(lists[i]) and node == str(i // 4) and text == lists[i]
    for i in range(8):
        for j in range(i + 1, 8):
            assert not (sets[i][0] & sets[j][0]), (i, j)
    assert set().union(*[c for c, _, _ in sets]) == set(range(256))
    alone = bench.rank_core_set(0, bdfs[:1], str(sysfs))[0]
    assert alone == bench.parse_cpulist(lists[0])
    with pytest.raises(OSError):
        bench.rank_core_set(0, ["0000:ff:00.0"], str(sysfs))
