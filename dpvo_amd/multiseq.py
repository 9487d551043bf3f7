"""Multi-sequence harness: one process per GPU, one sequence per process (BASELINE config 4).

The hot path does not shard inside a sequence (every BA iteration reduces all edges into one small system), so
multi-GPU = replicas: sequences are dealt round-robin to ranks and NO data-path collective exists.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the start / stop barriers
and for gathering one small record per rank.
"""
import os
import time

import torch


def partition(sequences, rank, world):
    """sequence s -> rank s mod world"""
    return list(sequences)[rank::world]


class Clock:
    """barrier + device sync on both sides of the timed region; elapsed() is this rank's wall time."""

    def __init__(self, dist=None, device=None):
        self.dist, self.device = dist, device
        self.t0 = self.t1 = None

    def _sync(self):
        if self.device is not None and torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()

    def start(self):
        self._sync()
        self.t0 = time.perf_counter()

    def stop(self):
        if self.device is not None and torch.device(self.device).type == "cuda":
            torch.cuda.synchronize(self.device)
        self.t1 = time.perf_counter()
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
        return self.t1 - self.t0


def gather_results(frames, seconds, extra=0.0, dist=None, device="cpu"):
    """all_gather of (frames, seconds, extra...) per rank -> whole-job record (`extra`: one number or a short sequence of numbers that
    travels with the rank's record, e.g. host CPU time per frame, device index, first / last pinned core).
    fps = sum(frames) / max(seconds): the job is done when the slowest rank is done."""
    extras = [float(extra)] if not isinstance(extra, (list, tuple)) else [float(x) for x in extra]
    rec = torch.tensor([float(frames), float(seconds)] + extras, dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(rec) for _ in range(dist.get_world_size())]
        dist.all_gather(out, rec)
        recs = torch.stack(out).cpu()
    else:
        recs = rec[None].cpu()
    total_frames = float(recs[:, 0].sum())
    max_seconds = float(recs[:, 1].max())
    return {"frames": total_frames, "seconds": max_seconds, "fps": total_frames / max_seconds,
            "per_rank": recs.tolist()}


def place_rank(local_rank, world, n_dev, n_cpu=None, allowed=None):
    """Where rank `local_rank` of `world` ranks on this node runs: (backend, device index, cpu set).
      * n_dev >= world: one device per rank, backend "nccl" (= RCCL over xGMI): the measurement configuration;
      * n_dev <  world: the ranks share the devices round-robin over "gloo" -- a smoke mode for the launch path only;
      * n_dev == 0: no placement possible.
    The host cores are dealt out in contiguous, disjoint slices of the allowed set (a tracker keeps one host thread busy pacing its
    stream and one HIP runtime thread beside it: 8 ranks on a small host must not migrate over each other); with fewer cores than
    ranks the slices wrap and the JSON line says so."""
    if n_dev <= 0:
        raise RuntimeError("no HIP device visible: bench.py measures on a GPU only")
    if not 0 <= local_rank < world:
        raise ValueError(f"LOCAL_RANK={local_rank} outside [0, {world})")
    backend = "nccl" if n_dev >= world else "gloo"
    cpus = sorted(allowed) if allowed is not None else list(range(n_cpu if n_cpu else (os.cpu_count() or 1)))
    per = max(1, len(cpus) // world)
    lo = (local_rank * per) % len(cpus)
    mine = [cpus[(lo + i) % len(cpus)] for i in range(per)]
    return backend, local_rank % n_dev, sorted(set(mine))


def pin_rank(cpus):
    """os.sched_setaffinity for this process (and the threads it starts afterwards); returns the set actually in force"""
    try:
        os.sched_setaffinity(0, set(cpus))
        return sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return None


def init_distributed(backend, device):
    """torch.distributed for the barriers and the result gather.  With backend "nccl" a failure is FATAL and says why: a silent
    fall-back to gloo would turn a broken RCCL setup into a plausible-looking multi-GPU number."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what this driver stack supports
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", device_id=device)
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)                                    # the first collective creates the communicator: fail HERE
            torch.cuda.synchronize(device)
            if int(probe.item()) != dist.get_world_size():
                raise RuntimeError(f"all_reduce over RCCL returned {probe.item()} for {dist.get_world_size()} ranks")
        except Exception as e:
            raise RuntimeError(f"RCCL (torch.distributed backend 'nccl') could not be initialised on {device} with "
                               f"WORLD_SIZE={os.environ.get('WORLD_SIZE')}: {e!r}.  Not falling back to gloo: fix the node "
                               "(HSA_ENABLE_IPC_MODE_LEGACY=0, visible devices, MASTER_ADDR=127.0.0.1) or run fewer ranks.") from e
    else:
        dist.init_process_group("gloo")
    return dist
