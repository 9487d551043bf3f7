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
