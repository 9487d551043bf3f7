"""Multi-sequence harness: one process per GPU, one sequence per process (BASELINE config 4).

The hot path does not shard inside a sequence (every BA iteration reduces all edges into one small system), so
multi-GPU = replicas: sequences are dealt round-robin to ranks and NO data-path collective exists.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only for the start / stop barriers
and for gathering one small record per rank.
"""
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
    """all_gather of (frames, seconds, extra) per rank -> whole-job record.
    fps = sum(frames) / max(seconds): the job is done when the slowest rank is done."""
    rec = torch.tensor([float(frames), float(seconds), float(extra)], dtype=torch.float64, device=device)
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
