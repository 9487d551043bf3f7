"""world_size-2 gloo test of the multi-sequence harness (the N>1 path of bench.py uses the same functions over RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dpvo_amd import multiseq


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seqs = multiseq.partition([f"seq{i}" for i in range(5)], rank, world)
    clock = multiseq.Clock(dist=dist, device="cpu")
    clock.start()
    frames = 0
    for s in seqs:                       # pretend to track: rank 0 has 3 sequences, rank 1 has 2
        frames += 10
    import time
    time.sleep(0.05 * (rank + 1))
    secs = clock.stop()
    res = multiseq.gather_results(frames, secs, extra=rank, dist=dist)
    q.put((rank, seqs, res))
    dist.destroy_process_group()


def test_two_rank_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    got.sort()
    (r0, s0, res0), (r1, s1, res1) = got
    assert s0 == ["seq0", "seq2", "seq4"] and s1 == ["seq1", "seq3"]
    assert res0["frames"] == 50.0 and res1["frames"] == 50.0
    assert abs(res0["seconds"] - res1["seconds"]) < 1e-9          # both ranks agree on the max
    assert res0["seconds"] >= 0.1 - 1e-3                          # the slower rank (0.1 s) defines the job time
    assert [r[2] for r in res0["per_rank"]] == [0.0, 1.0]


def test_single_process_path():
    res = multiseq.gather_results(7, 0.5)
    assert res["fps"] == 14.0 and multiseq.partition(range(4), 0, 1) == [0, 1, 2, 3]


import pytest


@pytest.mark.gpu
def test_bench_self_spawns_two_real_trackers():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run and every
    rank runs the REAL tracker on its own sequence (on a 1-GPU box the two ranks share the device over gloo); rank 0 prints one
    JSON line with n_gpus = 2, the max-over-ranks time and one record per rank (seconds, host CPU time per frame)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "45",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 12 and len(rec["per_rank"]) == 2, rec
    assert rec["state"]["finite"] and "error" not in rec, rec
    secs = [r["seconds"] for r in rec["per_rank"]]
    assert abs(rec["value"] - 2 * 12 / max(secs)) / rec["value"] < 1e-2, rec     # whole-job frames/sec = N*K / max over ranks
    assert all(r["host_cpu_us_per_frame"] > 0 for r in rec["per_rank"]), rec
    # what DESIGN.md section 6 says an accepted multi-GPU line carries per rank: device, frames/sec, the pinned core slice (disjoint)
    import torch
    n_dev = torch.cuda.device_count()
    assert all(0 <= r["device"] < n_dev and r["frames_per_sec"] > 0 for r in rec["per_rank"]), rec
    assert rec["placement"]["backend"] == ("nccl" if n_dev >= 2 else "gloo"), rec
    pins = [r["pinned_to"] for r in rec["per_rank"]]
    if all(pins):
        spans = sorted(tuple(int(x) for x in p_.split("-")) for p_ in pins)
        assert spans[0][1] < spans[1][0], ("the ranks' core slices overlap", pins)


def test_rank_placement_logic():
    """bench.py's device / backend / host-core selection (multiseq.place_rank) for the shapes of node it can meet, with a faked device
    count: 8 ranks on 8 devices over RCCL with disjoint core slices; more ranks than devices -> shared devices over gloo; a small host
    (fewer cores than ranks) still gives every rank a core; nonsense raises."""
    import pytest
    got = [multiseq.place_rank(r, 8, 8, n_cpu=64) for r in range(8)]
    assert [g[0] for g in got] == ["nccl"] * 8 and [g[1] for g in got] == list(range(8))
    cores = [set(g[2]) for g in got]
    assert all(len(c) == 8 for c in cores) and len(set().union(*cores)) == 64            # disjoint slices that cover the host
    assert multiseq.place_rank(3, 8, 8, allowed=[2, 3, 5, 7, 11, 13, 17, 19])[2] == [7]   # an affinity mask handed down by a launcher
    got = [multiseq.place_rank(r, 4, 1, n_cpu=8) for r in range(4)]
    assert [g[0] for g in got] == ["gloo"] * 4 and [g[1] for g in got] == [0] * 4
    got = [multiseq.place_rank(r, 8, 2, n_cpu=4) for r in range(8)]
    assert [g[1] for g in got] == [0, 1] * 4 and all(len(g[2]) == 1 for g in got)         # 8 ranks on 4 cores: one each, wrapping
    assert multiseq.place_rank(0, 1, 8)[0] == "nccl" and multiseq.place_rank(0, 2, 2)[1] == 0
    with pytest.raises(RuntimeError):
        multiseq.place_rank(0, 2, 0)
    with pytest.raises(ValueError):
        multiseq.place_rank(2, 2, 2)


def _nccl_must_fail_loudly(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        multiseq.init_distributed("nccl", torch.device("cpu"))          # no GPU here: RCCL cannot come up
        q.put("initialised")
    except RuntimeError as e:
        q.put(str(e))


def test_nccl_init_failure_is_loud_not_a_gloo_fallback():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_must_fail_loudly, args=(0, 1, _free_port(), q))
    p.start()
    msg = q.get(timeout=120)
    p.join(60)
    assert "could not be initialised" in msg and "Not falling back to gloo" in msg, msg


def test_bench_launcher_and_placement_for_eight_gpus():
    """VERDICT r4 #6 (no 8-GPU node to run on): what `python bench.py --gpus 8` WOULD execute, built without a GPU -- the
    torch.distributed.run command line the driver's contract names, and the placement every one of the 8 ranks derives from its
    LOCAL_RANK on a node that shows 8 devices and 256 host cores: backend "nccl" (= RCCL), 8 distinct devices,
    8 disjoint contiguous slices of 32 cores."""
    import sys
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd, env = bench.launcher_command(8, argv, port=29517, environ={"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    i = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[i + 1:] == argv, "the ranks must see the launcher's own arguments"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_ADDR"] == "127.0.0.1" and env["PATH"] == "/usr/bin"
    assert "HIP_VISIBLE_DEVICES" not in env and "ROCR_VISIBLE_DEVICES" not in env, "every rank keeps all 8 devices visible (xGMI peers)"
    placed = [multiseq.place_rank(r, 8, 8, allowed=range(256)) for r in range(8)]
    assert [b for b, _, _ in placed] == ["nccl"] * 8
    assert sorted(d for _, d, _ in placed) == list(range(8))
    cores = [c for _, _, c in placed]
    assert all(len(c) == 32 and c == list(range(c[0], c[0] + 32)) for c in cores)
    assert len(set().union(*map(set, cores))) == 256
    # ... and on the 1-GPU boxes of this pool the same command degrades loudly-labelled: gloo, shared device (smoke mode)
    assert [multiseq.place_rank(r, 8, 1, allowed=range(16))[:2] for r in range(8)] == [("gloo", 0)] * 8
