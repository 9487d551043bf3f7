#!/usr/bin/env python
"""bench.py -- frames/sec of the DPVO per-frame path on MI355X (BASELINE.json metric, config 2).

    python bench.py --gpus N --steps K --warmup W

One *step* = one `slam(t, image, intrinsics)` call on a synthetic 480x640 stream at the steady state of
config/default.yaml (96 patches/frame, E = 45 312 active edges): patchify (HIP MFMA encoders + HIP patch
gathers) -> edge bookkeeping -> reproject -> two-level correlation -> update operator -> 2 BA iterations ->
keyframe test + edge removal.  Random-init weights (no dpvo.pth on disk), synthetic images; the two data-dependent
gates that cannot behave sensibly with random weights are pinned so that the graph reaches and keeps the default
steady state: the initialisation motion probe is accepted (dpvo.py:441-444) and no keyframe is dropped
(KEYFRAME_THRESH = -1; the flow test itself, with its host read-backs, still runs every frame).

N > 1: replicas only -- every rank tracks its own sequence (one process per GPU), RCCL is used for the start/stop
barriers and the gather of one small record per rank.  `value` is the whole-job frames/sec =
N*K / max-over-ranks seconds; scaling is "weak".  Launched either by torch.distributed.run (RANK / WORLD_SIZE in the
environment) or plainly as `python bench.py --gpus N`, which re-executes itself under torch.distributed.run with N ranks
on 127.0.0.1.  If the box shows fewer than N devices the ranks share them round-robin over gloo (a smoke mode for the
launch path and the host-side cost, flagged in `config.parallelism`; not a scaling measurement).
`per_rank` carries, for every rank, its seconds, frames/sec, the host CPU time it spent per frame (time.process_time), its device index
and the host cores it pinned itself to.

The JSON line also carries
  roofline     -- the correlation kernel (corr_pyramid_kernel), mean launch duration from HIP events on the launch stream
                  inside the timed region (around every 4th launch, `launches` of them: PROFILE_EVERY), against the
                  8 TB/s HBM3E peak, three ways:
                    frac = frac_streaming: SURVEY.md 8d's ALGORITHMIC bytes (E x 52 884 B, overlapping windows counted per edge)
                      / duration / peak, as the measurement contract defines `achieved` -- exceeds 1 because the per-XCD L2s serve
                      the overlap: a rate of work, NOT a utilisation;
                    frac_l2_fabric: bytes per launch seen by the memory-side counters (`traffic`, committed PMC pass
                      profiles/rNN_corr_pmc.json, corrected as MI355X_MICROARCH.md prescribes) / duration / peak -- what crossed
                      L2 <-> fabric (Infinity-Cache hits included: not HBM alone);
                    frac_compulsory: every byte touched once (live pyramid + templates + coords + indices + output);
  roofline_update -- the update operator (net.py:74-92): reference FLOPs (5.40 MFLOP per edge) / mean duration
                  of the whole operator (HIP events around it on the launch stream) against the 2.5 PFLOP/s dense f16 MFMA peak;
  cpu_baseline -- the CPU oracle ("port": oracle/liboracle.so + oracle/update_ref.py) timed on rank 0 at N = 1 on
                  two full hot-path steps (reproject, corr, update, 2 BA iterations at E = 45 312), ~10 s;
  ref_baseline -- the REFERENCE's own tracker (its Python + its CUDA kernels compiled for gfx950 by oracle/build_ref.py, with torch
                  stand-ins for torch_scatter / lietorch_backends: oracle/ref_pipeline.py) on the same box, same stream, same weights,
                  same steady state (E = 45 312): frames/sec over 20 frames, rank 0 at N = 1 only; null when oracle/_ref is absent.
                  A reported baseline (the only same-node comparator north_star's ">= reference frames/sec" has), never `value`;
  box          -- the shader clock this box sustains under a full-chip / quarter-chip MFMA load (tools/probes/clock_probe.hip);
  frame_period_ms -- device-side frame period (correlation start to correlation start, averaged over windows of `window_frames`
                  frames: the HIP events sit on every 4th frame) over the timed region: median, p90, max;
  state        -- whether the tracker state after the run is sane (finite poses, fraction of edges that project in
                  bounds): with random weights nothing guarantees that, and a diverged state would make the correlation
                  kernel skip its work; such a run carries an "error" field.
"""
import argparse
import json
import os
import sys
import time

import torch

PROFILE_EVERY = 4       # HIP events around the correlation kernel / the update operator on every 4th timed frame (see main())

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_EDGE = 52884          # algorithmic bytes per edge, both pyramid levels, f16 features
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = 2500.0                                       # dense f16 MFMA peak (MI355X_MICROARCH.md)
UPDATE_FLOP_EDGE = 2 * (882 * 384 + 16 * 384 * 384 + 2 * 384 * 2)    # Update.forward per edge: 5.40 MFLOP


def corr_source_sha256():
    """fingerprint of the correlation kernel's source: corr.hip + corr_dev.h (the kernel's device code lives in the header since round 5)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("corr.hip", "corr_dev.h"):
        h.update(open(os.path.join(ROOT, "dpvo_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def pmc_traffic(config="default"):
    """HBM-side bytes per corr_pyramid_kernel launch from the newest committed PMC pass (tools/pmc_corr.sh, two separate
    rocprofv3 --pmc runs of this same command; corrected as MI355X_MICROARCH.md prescribes) -- but only if that pass measured
    THIS kernel: the file carries the SHA-256 of dpvo_amd/csrc/corr.hip + corr_dev.h it was taken with, and a file whose fingerprint does not
    match the source in the tree (or has none) is refused.  Returns (bytes or None, reason)."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_corr_pmc.json" if config == "default" else f"r*_corr_pmc_{config}.json")))
    if not files:
        return None, "no PMC pass committed"
    try:
        rec = json.load(open(files[-1]))
        now = corr_source_sha256()
        if rec.get("corr_hip_sha256") != now:
            return None, f"{os.path.basename(files[-1])} was measured on a different corr.hip (stale): refused"
        return float(rec["traffic_bytes_per_launch"]), os.path.basename(files[-1])
    except Exception as e:
        return None, f"unreadable PMC record: {e!r}"


def make_stream(n_frames, ht, wd, device, seed=1234):
    """Pre-staged synthetic frames: a fixed low-pass random texture, translated a few pixels per frame."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    tex = torch.rand(3, ht + 256, wd + 256, generator=g)
    tex = torch.nn.functional.avg_pool2d(tex[None], 5, 1, 2)[0]
    tex = (255 * (tex - tex.min()) / (tex.max() - tex.min())).to(torch.uint8)
    frames = []
    for t in range(n_frames):
        dx, dy = (3 * t) % 256, (2 * t) % 256
        frames.append(tex[:, dy:dy + ht, dx:dx + wd])
    return torch.stack(frames).to(device)


def make_plane_stream(n_frames, ht, wd, device, seed=1234, depth_m=4.0, step_m=0.02, yaw_deg=0.2, intr=(320.0, 320.0, 320.0, 240.0)):
    """SURVEY.md 8(d), the stream-level input of the metric: `n_frames` uint8 frames rendered from ONE fixed low-pass-filtered random
    texture (seed 1234) on a fronto-parallel plane `depth_m` in front of a camera that translates `step_m` per frame along x while its
    yaw swings by +-`yaw_deg` per frame (a triangle wave of period 40 frames: |yaw| <= 2 degrees), intrinsics calib/tartan.txt.  Rendered
    on the CPU (device independent: the same frames on every box), then staged in HBM.  The image motion is 320 * 0.02 / 4 = 1.6 px per
    frame from the translation alone."""
    import math
    g = torch.Generator(device="cpu").manual_seed(seed)
    T = 2048
    # two octaves of low-pass noise, 1 texel = 1 cm on the plane (the 480 x 640 view covers 6 m x 8 m of it)
    lo = torch.nn.functional.interpolate(torch.rand(1, 3, T // 16, T // 16, generator=g), size=(T, T), mode="bicubic", align_corners=False)
    hi = torch.nn.functional.interpolate(torch.rand(1, 3, T // 4, T // 4, generator=g), size=(T, T), mode="bicubic", align_corners=False)
    tex = 0.6 * lo + 0.4 * hi
    tex = (tex - tex.amin()) / (tex.amax() - tex.amin())
    fx, fy, cx, cy = intr
    v, u = torch.meshgrid(torch.arange(ht, dtype=torch.float64), torch.arange(wd, dtype=torch.float64), indexing="ij")
    rays = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones_like(u)], -1)             # camera frame
    frames = []
    for t in range(n_frames):
        ph = t % 40
        yaw = math.radians(yaw_deg) * (ph if ph < 10 else 20 - ph if ph < 30 else ph - 40)
        c, s_ = math.cos(yaw), math.sin(yaw)
        R = torch.tensor([[c, 0.0, s_], [0.0, 1.0, 0.0], [-s_, 0.0, c]], dtype=torch.float64)      # camera -> world
        d = rays @ R.T
        lam = depth_m / d[..., 2]
        X = d[..., 0] * lam + (step_m * t - 0.5 * step_m * n_frames)                            # world x of the hit point (metres)
        Y = d[..., 1] * lam
        grid = torch.stack([X * 100.0 / (T / 2), Y * 100.0 / (T / 2)], -1).float()[None]      # texels -> [-1, 1]
        img = torch.nn.functional.grid_sample(tex, grid, mode="bilinear", padding_mode="reflection", align_corners=False)[0]
        frames.append((255.0 * img).clamp_(0, 255).to(torch.uint8))
    return torch.stack(frames).to(device)


def unforced_probe_leg(cfg, net, ht, wd, device, frames, intr, max_frames=48):
    """The same tracker on the same stream WITHOUT the pinned initialisation probe (dpvo.py:441-444: frames are only accepted while the
    median flow of the probe is >= 2 px): does a random-weight tracker initialise by itself on the 8(d) stream?  Reports the probe
    values it saw (the flow head of random weights emits noise whatever the image motion is, so this says something about the
    weights, not about the stream) and whether n reached 8."""
    from dpvo_amd.dpvo import DPVO
    try:
        slam = DPVO(cfg, net, ht=ht, wd=wd, device=device)
        probes = []
        real = slam.motion_probe

        def probe():
            v = float(real())
            probes.append(round(v, 3))
            return v
        slam.motion_probe = probe
        with torch.no_grad():
            for t in range(min(max_frames, frames.shape[0])):
                slam(float(t), frames[t], intr, image_ready=False)
                if slam.is_initialized:
                    break
            slam.flush()
        torch.cuda.synchronize(device)
        return {"initialised_by_itself": bool(slam.is_initialized), "frames_offered": t + 1, "keyframes": int(slam.n),
                "probe_px_first": probes[:6], "probe_px_min_max": [min(probes), max(probes)] if probes else None, "threshold_px": 2.0}
    except Exception as e:          # noqa: BLE001  (a side leg must never take the headline measurement down)
        return {"initialised_by_itself": None, "error": repr(e)[:300]}


def box_clock():
    """Shader clock this box sustains under a full-chip MFMA load (tools/probes/clock_probe.hip, built by
    __graft_entry__.build(); ~1 s after the timed region): the same build measures +-3 % frames/sec from box to box, and this is
    the box-side number next to it.  None when the probe binary is missing."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "probes", "clock_probe.bin")
    if not os.path.isfile(exe):
        return None
    try:
        torch.cuda.synchronize()
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=60)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:          # noqa: BLE001  (telemetry only)
        return {"error": str(e)[:200]}


def cpu_baseline():
    """One hot-path step on the host cores with the oracle (the checker, used here as the CPU 'port')."""
    import numpy as np
    import oracle
    from oracle import update_ref
    from dpvo_amd import synthetic as S
    from dpvo_amd.net import Update
    oracle.build()
    cores = min(os.cpu_count() or 1, 16)     # the oracle's OpenMP loop and torch-CPU both capped at 16 threads
    torch.set_num_threads(cores)
    try:
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    ii, jj, kk = S.replay_graph(40)
    E = ii.numel()
    gmap, f0, f1, imap = S.make_features()
    poses, patches, intr = S.make_scene(40)
    torch.manual_seed(1234)
    sd = Update(3).state_dict()
    g32, a, b = gmap.float().numpy(), f0.float().numpy(), f1.float().numpy()
    gen = torch.Generator().manual_seed(0)
    net = torch.randn(E, 384, generator=gen)
    nstep = 2                                # ~10-12 s of CPU work on the GPU box's host cores
    t0 = time.perf_counter()
    for _ in range(nstep):
        coords = oracle.reproject(poses.numpy(), patches.numpy(), intr.numpy(), ii.numpy(), jj.numpy(), kk.numpy(), dtype=np.float32)
        corr = oracle.corr_pyramid(g32, [a, b], coords, (kk % 3456).numpy(), (jj % 36).numpy(), dtype=np.float32)
        inp = imap[kk % 3456]
        _, delta, weight = update_ref.update_forward(sd, net, inp, torch.from_numpy(corr), ii, jj, kk)
        target = coords[:, :, 1, 1] + delta.numpy().astype(np.float32)
        oracle.ba(poses.numpy(), patches.numpy(), intr.numpy(), target, weight.numpy(), 1e-4, ii.numpy(), jj.numpy(), kk.numpy(),
                  30, 40, iterations=2, dtype=np.float32)
    dt = (time.perf_counter() - t0) / nstep
    return {"value": 1.0 / dt, "unit": "frames/sec", "cores": cores, "kind": "port",
            "sample": f"{nstep} hot-path steps (reproject+corr+update+2 BA iters) at E={E} on the CPU oracle "
                      f"(C/OpenMP f32 for corr/reproject/BA, torch-CPU for the update operator); encoders excluded; "
                      f"{dt:.1f} s per step"}


def ref_baseline(device, ht, wd, cfg, frames, intr, seed, warm=53, timed=20):
    """frames/sec of the reference's own tracker on this box (test infrastructure used as a comparator, like cpu_baseline)"""
    try:
        from oracle import ref_pipeline as RP
        if not RP.available():
            return {"frames_per_sec": None, "reason": "oracle/_ref not built (needs /root/reference at build time)"}
        from dpvo_amd.net import VONet
        torch.manual_seed(seed)
        sd = {k: v.detach().clone() for k, v in VONet().state_dict().items()}
        slam = RP.make_tracker(RP.make_cfg(cfg), sd, ht, wd, accept_probe=True)
        fps, dt = RP.throughput(slam, frames, intr, warm, timed, seed=seed)
        E = int(slam.pg.ii.numel())
        finite = bool(torch.isfinite(slam.pg.poses_[:slam.n]).all().item())
        return {"frames_per_sec": round(fps, 2), "ms_per_frame": round(1e3 * dt / timed, 2), "frames": timed, "edges": E, "finite": finite,
                "kind": "reference Python (dpvo/dpvo.py, net.py, ...) + reference kernels (cuda_corr, cuda_ba) hipified for gfx950, torch "
                        "stand-ins for torch_scatter / lietorch_backends, torch-MIOpen encoders under autocast; same box, same stream, "
                        "same weights, same steady state"}
    except Exception as e:          # noqa: BLE001  (a comparator must never take the measurement down)
        return {"frames_per_sec": None, "error": repr(e)[:300]}


def loop_closure_leg(cfg, ht, wd, device, frames, intr, n_img, seed, warm=70, timed=45, backend_thresh=None):
    """BASELINE config 5 (LOOP_CLOSURE=True) on the bench stream: see the call site.  backend_thresh = 0.0: no candidate ever passes
    the flow test, i.e. the configuration as it runs on a sequence WITHOUT revisits -- edges_loop evaluated on every frame, nothing found,
    every frame on the one-call path."""
    from dpvo_amd.dpvo import DPVO
    from dpvo_amd.net import VONet
    try:
        c = cfg.clone()
        c.LOOP_CLOSURE = True
        if backend_thresh is not None:
            c.BACKEND_THRESH = backend_thresh
        c.BUFFER_SIZE = max(c.BUFFER_SIZE, warm + timed + 80)
        torch.manual_seed(seed)
        slam = DPVO(c, VONet(), ht=ht, wd=wd, device=device)
        slam.motion_probe = lambda: 1.0e9
        with torch.no_grad():
            for t in range(warm):
                slam(float(t), frames[t % n_img], intr, image_ready=False)
            slam.flush(); torch.cuda.synchronize(device)
            gb0, fast = int(slam.ran_global_ba.sum()), 0
            t0 = time.perf_counter()
            for t in range(warm, warm + timed):
                pend = slam._fu_pending
                slam(float(t), frames[t % n_img], intr, image_ready=False)
                fast += int(slam._fu_pending is not None and slam._fu_pending is not pend)
            slam.flush(); torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
        gb = int(slam.ran_global_ba.sum()) - gb0
        return {"frames": timed, "frames_per_sec": round(timed / dt, 1), "ms_per_frame": round(1e3 * dt / timed, 3),
                "frames_on_the_one_call_path": fast, "global_ba_runs": gb, "keyframes": int(slam.n),
                "active_edges": int(slam.pg.ii.numel()), "inactive_edges": int(slam.pg.ii_inac.numel()),
                "finite": bool(torch.isfinite(slam.pg.poses_[:slam.n]).all().item())}
    except Exception as e:          # noqa: BLE001  (a side leg must never take the headline measurement down)
        return {"frames_per_sec": None, "error": repr(e)[:300]}


def host_image_leg(cfg, ht, wd, device, frames, intr, n_img, seed, warm=50, timed=40):
    """The PCIe-inclusive rates (never `value`: the metric is quoted with the inputs resident in HBM).  The frames start in PAGEABLE host
    memory as HWC uint8 arrays, the way a reader process delivers them, and reach the tracker in two ways:
      reference_loop  -- demo.py:38-44 as written: `torch.from_numpy(image).permute(2, 0, 1).cuda()` on the caller's stream, then
                         slam(t, image, intrinsics); the upload from pageable memory synchronises that stream, i.e. drains the pipeline
      host_handoff    -- slam(t, torch.from_numpy(image).permute(2, 0, 1), intrinsics): the CPU tensor itself; the tracker uploads it
                         through its pinned ring on the encoder stream (DPVO._upload_image)."""
    from dpvo_amd.dpvo import DPVO
    from dpvo_amd.net import VONet
    try:
        host = [f.permute(1, 2, 0).contiguous().cpu().numpy() for f in frames[:min(n_img, 64)]]
        torch.manual_seed(seed)
        slam = DPVO(cfg, VONet(), ht=ht, wd=wd, device=device)
        slam.motion_probe = lambda: 1.0e9
        out, t = {}, 0
        with torch.no_grad():
            for name, n_warm in (("reference_loop", warm), ("host_handoff", 15)):
                def step(t):
                    img = torch.from_numpy(host[t % len(host)]).permute(2, 0, 1)
                    slam(float(t), img.cuda() if name == "reference_loop" else img, intr)
                for _ in range(n_warm):
                    step(t); t += 1
                slam.flush(); torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(timed):
                    step(t); t += 1
                slam.flush(); torch.cuda.synchronize(device)
                dt = time.perf_counter() - t0
                out[name] = {"frames": timed, "frames_per_sec": round(timed / dt, 1), "ms_per_frame": round(1e3 * dt / timed, 4)}
        out["finite"] = bool(torch.isfinite(slam.pg.poses_[:slam.n]).all().item())
        out["edges"] = int(slam.pg.ii.numel())
        return out
    except Exception as e:          # noqa: BLE001  (a side leg must never take the headline measurement down)
        return {"error": repr(e)[:300]}


def launcher_command(gpus, argv, port=None, environ=None):
    """(command, environment) with which `python bench.py --gpus N` re-executes itself: N ranks of ONE node under torch.distributed.run,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve), dmabuf IPC for RCCL.  Every rank then places itself from
    LOCAL_RANK alone (multiseq.place_rank: device LOCAL_RANK over "nccl" when the node shows >= N devices, a disjoint slice of the host
    cores) -- no HIP_VISIBLE_DEVICES games: all devices stay visible to every rank, which is what RCCL's xGMI peer access wants.
    A function of its arguments (tests/test_multiseq.py builds the 8-GPU command without a GPU)."""
    if port is None:
        import socket
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = dict(os.environ if environ is None else environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return cmd, env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=45)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-baseline", action="store_true")
    ap.add_argument("--config", default="default", choices=["default", "fast"])
    ap.add_argument("--update-tiling", type=int, default=None,
                    help="measurements: dpvo_update_fused_params_t.tiling of the update operator (include/dpvo_hip.h; default: the library's)")
    ap.add_argument("--seed-offset", type=int, default=None, help="sequence / weight seed offset (default: the rank)")
    ap.add_argument("--drop-every", type=int, default=0,
                    help="k > 0: the keyframe test drops keyframe n - KEYFRAME_INDEX on every k-th frame (scripted decision: the "
                         "remove-frame + renumber + ring-shift branch of dpvo.py:266-310 runs INSIDE the timed region); 0: never "
                         "(the steady state the metric is quoted on); at N = 1 the default run appends a short second leg with k = 3")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU) on this node
        import subprocess
        cmd, env = launcher_command(args.gpus, sys.argv[1:])
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    n_dev = torch.cuda.device_count()
    from dpvo_amd import multiseq
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    # one device per rank over RCCL when the node has them, shared devices over gloo otherwise (smoke mode, see docstring); every rank
    # pinned to its own slice of the host cores
    backend, dev_index, cpus = multiseq.place_rank(local_rank, world, n_dev, allowed=allowed)
    pinned = multiseq.pin_rank(cpus) if world > 1 and not os.environ.get("DPVO_BENCH_NO_PIN") else None
    dist = None
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    if world > 1:
        dist = multiseq.init_distributed(backend, device)          # (raises if RCCL cannot be brought up: never a silent gloo run)

    import dpvo_amd.dpvo as dpvo_mod
    dpvo_mod._PROFILE_POOL = True       # the tracker creates its pool of timing events up front (warm-up), not in the timed region
    # HIP events around the correlation kernel / the update operator on every 4th timed frame: each event record is a marker the
    # stream stalls on, three of them per frame cost ~10 us of every frame (946-955 -> 957-960 frames/sec on one box, 963 without any)
    dpvo_mod._PROFILE_EVERY = PROFILE_EVERY
    from dpvo_amd import altcorr
    from dpvo_amd.altcorr import correlation as corr_mod
    from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML, FAST_YAML
    from dpvo_amd.dpvo import DPVO
    from dpvo_amd.net import VONet
    from dpvo_amd import net as net_mod

    if os.environ.get("DPVO_BENCH_MAIN_PRIO"):       # experiment: the tracker's main stream with a HIP stream priority
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=int(os.environ["DPVO_BENCH_MAIN_PRIO"])))
    cfg = base_cfg.clone()
    cfg.merge_from_dict(DEFAULT_YAML if args.config == "default" else FAST_YAML)
    cfg.KEYFRAME_THRESH = -1.0                       # keep every keyframe (see module docstring)
    ht, wd = 480, 640
    # the metric is quoted at the steady state of default.yaml (E = 45 312 edges), reached ~36 frames after initialisation:
    # frames needed beyond the requested warm-up are run as an untimed pre-roll BEFORE the W warm-up steps
    preroll = max(0, 45 - args.warmup)
    total = preroll + args.warmup + args.steps
    cfg.BUFFER_SIZE = max(cfg.BUFFER_SIZE, total + 80)      # every frame stays a keyframe in this workload (+ the second leg)
    seed_off = rank if args.seed_offset is None else args.seed_offset
    torch.manual_seed(1234 + seed_off)
    net = VONet()
    shared = world > 1 and backend != "nccl"        # smoke mode: ranks share a device (see docstring)
    # The tracker as a drop-in caller constructs it (demo.py:46: DPVO(cfg, network, ht, wd, viz)): since round 6 the defaults ARE the
    # pipeline (decision of frame t resolved under frame t + 1's encoders, which run on a second HIP stream).  The one exception: several
    # ranks sharing one device (smoke mode) -- two processes time-slicing a GPU with a multi-stream tracker each hit a memory access
    # fault on this ROCm stack (profiles/README.md) -- run without the second stream.
    slam = DPVO(cfg, net, ht=ht, wd=wd, device=device, **({"overlap_encoders": False} if shared else {}))
    if args.update_tiling is not None:
        slam.network.update.tiling = args.update_tiling
    slam.motion_probe = lambda: 1.0e9                # accept the initialisation probe (random weights)
    if args.drop_every > 0:
        slam.keyframe_override = lambda counter: counter % args.drop_every == 0
    # the perspective plane render; as many frames as the headline leg tracks (no wrap-around inside it), at least 128
    n_img = min(300, max(128, total))
    frames = make_plane_stream(n_img, ht, wd, device, seed=1234 + seed_off)
    torch.cuda.synchronize(device)                   # the stream is resident before the first frame is tracked
    intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=device)   # calib/tartan.txt

    # counter passes only (tools/pmc_corr.sh): rocprofv3 --pmc serialises every dispatch, and with more than a few dozen
    # dispatches outstanding (the 12 back-to-back updates of the initialisation frame) it faults; per-kernel counters do not
    # care about a host sync per frame.  Never set for a timed run.
    _SYNC_EVERY_FRAME = bool(int(os.environ.get("DPVO_BENCH_SYNC_EVERY_FRAME", "0")))

    # DPVO_BENCH_HOST_TRACE=1 (diagnosis of a slow frame, never for a quoted run): host wall time of every step and every garbage
    # collection with its generation and duration, to stderr
    _HOST_TRACE = [] if os.environ.get("DPVO_BENCH_HOST_TRACE") else None
    if _HOST_TRACE is not None:
        import gc
        _gc_t = [0.0]

        def _gc_cb(phase, info):
            if phase == "start":
                _gc_t[0] = time.perf_counter()
            else:
                _HOST_TRACE.append(("gc", info["generation"], round(1e3 * (time.perf_counter() - _gc_t[0]), 3), info["collected"]))
        gc.callbacks.append(_gc_cb)

    def step(t):
        # image_ready=False: the frames were staged in HBM and synchronised before the timed region (the metric is quoted with
        # resident inputs), so the encoder stream need not wait for the compute stream
        if _HOST_TRACE is not None:
            t0_ = time.perf_counter()
        slam(float(t), frames[t % n_img], intr, image_ready=False)
        if _HOST_TRACE is not None:
            _HOST_TRACE.append(("step", t, round(1e3 * (time.perf_counter() - t0_), 3)))
        if _SYNC_EVERY_FRAME:
            torch.cuda.synchronize(device)

    clock = multiseq.Clock(dist=dist, device=device)       # barrier + device sync on both sides of the timed region
    with torch.no_grad():
        for t in range(preroll + args.warmup):
            step(t)
        corr_mod.PROFILE = []
        net_mod.PROFILE = []
        clock.start()                                 # barrier + torch.cuda.synchronize()
        cpu0 = time.process_time()
        for t in range(preroll + args.warmup, total):
            step(t)
        slam.flush()                                  # the last frame's deferred keyframe decision belongs to the timed region
        cpu1 = time.process_time()
        local = clock.stop()                          # torch.cuda.synchronize() + barrier
    if _HOST_TRACE is not None:
        first = preroll + args.warmup
        print("host trace (ms): " + " ".join(f"{e[1] - first}:{e[2]}" if e[0] == "step" else f"[gc{e[1]} {e[2]} ms, {e[3]} collected]"
                                             for e in _HOST_TRACE if e[0] == "gc" or e[1] >= first), file=sys.stderr)
    prof = corr_mod.PROFILE
    uprof = net_mod.PROFILE
    corr_mod.PROFILE = None
    net_mod.PROFILE = None
    E_now = int(slam.pg.ii.numel())
    # second leg (N = 1, default run only): the same tracker goes on with a keyframe dropped every 3rd frame -- the branch a
    # real sequence takes on most frames (dpvo.py:266-310: edges of the dropped frame removed, ids renumbered, ring buffers
    # shifted down).  Reported beside the headline, never part of `value`.
    drop_leg = None
    if world == 1 and args.drop_every == 0 and not os.environ.get("DPVO_BENCH_NO_DROP_LEG"):
        k_drop, n_warm, n_timed = 3, 12, 36
        slam.keyframe_override = lambda counter: counter % k_drop == 0
        with torch.no_grad():
            for t in range(total, total + n_warm):
                step(t)
            slam.flush(); torch.cuda.synchronize(device)
            n0, t0 = slam.n, time.perf_counter()
            for t in range(total + n_warm, total + n_warm + n_timed):
                step(t)
            slam.flush(); torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
        drop_leg = {"drop_every": k_drop, "frames": n_timed, "frames_per_sec": round(n_timed / dt, 1),
                    "ms_per_frame": round(1e3 * dt / n_timed, 4), "keyframes_dropped": n_timed - (slam.n - n0),
                    "edges_after": int(slam.pg.ii.numel())}
        slam.keyframe_override = None
    # third leg (N = 1, default run only): BASELINE config 5 -- a second tracker with LOOP_CLOSURE=True on the same stream from its first
    # frame: PatchGraph.edges_loop is evaluated whenever it is due, loop edges go in, update() runs the global BA (active + inactive
    # edges, device Cholesky) while long-range edges are active, and every other frame takes the one-call path.  Reported beside the
    # headline: frames/sec over the timed frames, how many of them ran a global BA / the one-call path, and the size of the last one.
    lc_leg = None
    if world == 1 and args.drop_every == 0 and args.config == "default" and not os.environ.get("DPVO_BENCH_NO_LC_LEG"):
        lc_leg = loop_closure_leg(cfg, ht, wd, device, frames, intr, n_img, seed=1234 + seed_off)
        # ... and the same configuration when no loop is ever found (the common case on real sequences): what LOOP_CLOSURE=True costs a
        # frame that has nothing to close -- one dpvo_loop_flow launch + one read-back per frame
        nl = loop_closure_leg(cfg, ht, wd, device, frames, intr, n_img, seed=1234 + seed_off, warm=60, timed=60, backend_thresh=0.0)
        lc_leg["no_loop_found"] = {k: nl.get(k) for k in ("frames", "frames_per_sec", "ms_per_frame", "frames_on_the_one_call_path",
                                                           "global_ba_runs", "error") if k in nl}
    # fourth leg (N = 1, default run only; SURVEY 8(d) / VERDICT r5 #8): the initialisation probe NOT pinned -- does this stream + these
    # (random) weights initialise a tracker by themselves?
    probe_leg = None
    if world == 1 and args.drop_every == 0 and args.config == "default" and not os.environ.get("DPVO_BENCH_NO_PROBE_LEG"):
        torch.manual_seed(1234 + seed_off)
        probe_leg = unforced_probe_leg(cfg, VONet(), ht, wd, device, frames, intr)
    # fifth leg (N = 1, default run only): the frames handed over in host memory (tier rule 4: the PCIe-inclusive rate beside the metric)
    host_leg = None
    if world == 1 and args.drop_every == 0 and args.config == "default" and not os.environ.get("DPVO_BENCH_NO_HOST_LEG"):
        host_leg = host_image_leg(cfg, ht, wd, device, frames, intr, n_img, seed=1234 + seed_off)
    res = multiseq.gather_results(args.steps, local, extra=[1e6 * (cpu1 - cpu0) / args.steps, dev_index,
                                                           pinned[0] if pinned else -1, pinned[-1] if pinned else -1], dist=dist,
                                  device=device if backend == "nccl" else "cpu")
    elapsed = res["seconds"]                          # max over ranks

    corr_ms = [s.elapsed_time(e) for s, e, _ in prof]
    corr_edges = [n for _, _, n in prof]
    # frame-to-frame period on the device (start of one frame's correlation kernel to the next one's): with 20-60 timed frames one
    # hiccup moves `value` by several per cent -- the median says what the steady state is, the max what the hiccup was
    period = None
    stride = PROFILE_EVERY              # (events on every stride-th frame: a window is stride frames)
    if len(prof) > 2:
        raw = [prof[i][0].elapsed_time(prof[i + 1][0]) / stride for i in range(len(prof) - 1)]
        per = sorted(raw)
        period = {"median": round(per[len(per) // 2], 4), "p90": round(per[int(0.9 * (len(per) - 1))], 4), "max": round(per[-1], 4),
                  "max_at_frame": stride * (raw.index(per[-1]) + 1), "frames_per_sec_at_median": round(1e3 / per[len(per) // 2], 1),
                  "window_frames": stride}
    roof = None
    if corr_ms:
        avg_ms = sum(corr_ms) / len(corr_ms)
        avg_E = sum(corr_edges) / len(corr_edges)
        streaming = avg_E * B_EDGE / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(args.config)
        # every byte touched once: live part of the pyramid (34 of 36 frames, both levels) + templates + coords + indices + output
        compulsory_bytes = (34.0 / 36.0) * 36 * 128 * 2 * (120 * 160 + 30 * 40) + 22 * 96 * 9 * 128 * 2 + avg_E * (144 + 32 + 1792)
        counter = traffic / (avg_ms * 1e-3) / 1e9 if traffic else None
        # `achieved` / `frac` as the measurement contract defines them: SURVEY 8(d)'s ALGORITHMIC bytes per launch / the measured launch
        # duration.  The 8(d) model counts every edge's windows in full, and windows of neighbouring edges overlap: the per-XCD L2s
        # serve the overlap, so the figure exceeds the HBM peak -- it is a rate of work, not a utilisation.  `traffic` is what the
        # memory-side counters saw (L2 <-> fabric, Infinity-Cache hits included: NOT HBM alone); `frac_l2_fabric` = traffic / duration /
        # peak, `frac_compulsory` = every byte once.  (Rounds 1-5 reported the counter figure as `frac`: VERDICT r5 #6.)
        roof = {"bound": "hbm", "kernel": "corr_pyramid_kernel",
                "achieved": round(streaming, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(streaming / HBM_PEAK_GBS, 4),
                "frac_kind": "streaming: SURVEY 8d algorithmic bytes (overlapping windows counted per edge; the L2s serve the overlap, hence > 1)",
                "frac_l2_fabric": round(counter / HBM_PEAK_GBS, 4) if counter is not None else None,
                "frac_streaming": round(streaming / HBM_PEAK_GBS, 4),
                "frac_compulsory": round(compulsory_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": avg_E * B_EDGE, "compulsory_bytes": round(compulsory_bytes),
                "avg_launch_ms": round(avg_ms, 4), "edges_per_launch": round(avg_E, 1), "bytes_per_edge": B_EDGE,
                "launches": len(corr_ms)}
    roof_u = None
    if uprof:
        ums = [s.elapsed_time(e) for s, e, _ in uprof]
        uE = [n for _, _, n in uprof]
        avg_ms = sum(ums) / len(ums)
        avg_E = sum(uE) / len(uE)
        flops = avg_E * UPDATE_FLOP_EDGE
        tf = flops / (avg_ms * 1e-3) / 1e12
        path = "7 launches, dpvo_amd/csrc/update_fused.hip" if net_mod.FUSED_DEFAULT else "23 launches, dpvo_amd/csrc/update.hip"
        roof_u = {"bound": "mfma", "kernel": f"update operator (Update.forward: {path})",
                  "flops": flops, "avg_ms": round(avg_ms, 4), "achieved_tflops": round(tf, 1), "peak": MFMA_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "edges_per_call": round(avg_E, 1),
                  "flop_per_edge": UPDATE_FLOP_EDGE, "calls": len(ums),
                  "note": "reference FLOPs; the kernels execute 2 more E-row GEMMs (h applied per edge instead of per group)"}

    if rank == 0:
        out = {
            "metric": f"frames/sec (480x640, {cfg.PATCHES_PER_FRAME} patches/frame)", "value": round(world * args.steps / elapsed, 3),
            "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 features / f32 accumulate, f32 BA", "data": "synthetic",
            "config": {"workload": f"synthetic 480x640 stream (SURVEY 8d: textured plane at 4 m, 2 cm per frame, +-0.2 deg per frame yaw, calib/tartan.txt, "
                                   f"CPU-rendered with seed 1234, resident in HBM), {cfg.PATCHES_PER_FRAME} patches/frame, {args.config}.yaml, "
                                   f"steady state E={E_now} edges, random-init weights (initialisation probe pinned, no keyframe dropped: "
                                   f"`unforced_probe` says what happens without the pin), one sequence per GPU",
                       "patches_per_frame": cfg.PATCHES_PER_FRAME, "edges": E_now, "drop_every": args.drop_every, "parallelism": f"replicas x{world}" + ("" if backend == "nccl" or world == 1 else
                                                                f" sharing {n_dev} device(s) over gloo (launch-path smoke mode)")},
            "frame_period_ms": period,
            "roofline": roof, "roofline_update": roof_u, "with_keyframe_drops": drop_leg, "with_loop_closure": lc_leg,
            "unforced_probe": probe_leg, "images_from_host_memory": host_leg,
            "per_rank": [{"rank": i, "frames": r[0], "seconds": round(r[1], 6), "frames_per_sec": round(r[0] / r[1], 1),
                          "host_cpu_us_per_frame": round(r[2], 1), "device": int(r[3]),
                          "pinned_to": (f"{int(r[4])}-{int(r[5])}" if r[4] >= 0 else None)}
                         for i, r in enumerate(res["per_rank"])],
            "placement": {"backend": backend if world > 1 else None, "devices_visible": n_dev, "host_cores_allowed": len(allowed) if allowed else None,
                          "rank0_pinned_to": (f"{pinned[0]}-{pinned[-1]} ({len(pinned)} cores)" if pinned else None), "cores_per_rank": len(cpus)},
        }
        # the tracker state after the run: with random weights nothing guarantees that it stays sane, and a diverged state
        # (NaN poses, every edge projecting out of bounds) would make the correlation kernel skip its work
        co = slam.reproject()[0, :, :, 1, 1]
        inb = ((co[:, 0] > 0) & (co[:, 0] < wd / 4) & (co[:, 1] > 0) & (co[:, 1] < ht / 4)).float().mean().item()
        out["state"] = {"finite": bool(torch.isfinite(slam.pg.poses_[:slam.n]).all().item()), "edges_in_bounds": round(inb, 4)}
        if not out["state"]["finite"] or inb < 0.5:
            out["error"] = "tracker state diverged during the run: the timing above is not a valid measurement"
        if os.environ.get("DPVO_BENCH_THREADS"):           # CPU seconds per thread of this process (who burns the host time?)
            th = {}
            for t_ in os.listdir("/proc/self/task"):
                try:
                    f_ = open(f"/proc/self/task/{t_}/stat").read().rsplit(")", 1)[1].split()
                    nm_ = open(f"/proc/self/task/{t_}/comm").read().strip()
                    th[f"{t_}:{nm_}"] = round((int(f_[11]) + int(f_[12])) / os.sysconf("SC_CLK_TCK"), 2)
                except OSError:
                    pass
            out["threads_cpu_s"] = th
        if os.environ.get("DPVO_BENCH_DIAG"):              # state fingerprint + buffer addresses (run-to-run comparisons)
            out["diag"] = {"pose_sum": float(slam.pg.poses_[:slam.n].double().abs().sum().item()),
                           "depth_sum": float(slam.pg.patches_[:slam.n, :, 2].double().abs().sum().item()),
                           "net_sum": float(slam.pg.net.double().abs().sum().item()), "in_bounds": round(inb, 4),
                           "ptr": {k: hex(getattr(slam, k).data_ptr()) for k in ("_fmap1_cl", "_fmap2_cl", "_gmap_cl", "imap_")},
                           "corr_ms_minmax": [round(min(corr_ms), 4), round(max(corr_ms), 4)]}
        if world == 1 and not os.environ.get("DPVO_BENCH_NO_BOX"):     # (skipped under rocprofv3: the probe is a child process)
            out["box"] = box_clock()
        # The slow kind of box (one call in ~10 on this pool; profiles/r05_b_*_slow_box.txt, r04: "650-667 on a throttled one"): the
        # kernels that run ONE wave per SIMD or one workgroup in all (K1 / K7 of the update operator, the BA solve, the Cholesky
        # panels) take 1.6-2.1x as long while the 12-waves-per-CU correlation kernel is unaffected, and the sustained-MFMA clock probe
        # above does not see it.  What does: the ratio of the two stages' times, measured in THIS run (1.8-1.9 on the usual boxes).
        if roof is not None and roof_u is not None and roof.get("avg_launch_ms"):
            ratio = round(roof_u["avg_ms"] / roof["avg_launch_ms"], 2)
            if not isinstance(out.get("box"), dict):
                out["box"] = {}
            out["box"]["update_over_corr"] = ratio
            if ratio > 2.3 and args.config == "default":
                out["box"]["slow_box"] = ("update operator / correlation = %.2f (1.8-1.9 on the pool's usual boxes): this box runs the "
                                          "one-wave-per-SIMD kernels 1.6-2x slower, frames/sec here is ~0.7x of the usual" % ratio)
        if world == 1 and not args.no_ref_baseline and args.config == "default":
            out["ref_baseline"] = ref_baseline(device, ht, wd, cfg, frames, intr, 1234 + seed_off)
            if out["ref_baseline"].get("frames_per_sec"):
                out["ref_baseline"]["speedup"] = round(out["value"] / out["ref_baseline"]["frames_per_sec"], 1)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
