#!/usr/bin/env python
"""The config-5 leg of bench.py (LOOP_CLOSURE=True, 45 timed frames, 18 of them with a global BA) under the measurement hooks that are left
(module attributes since round 6 -- the package reads no measurement switch from the environment any more; the round-5 variants that lost
their A/B are gone from the product, their numbers are in profiles/r05_*_lc_ab.txt), each variant in its own process, `reps` times each,
interleaved so that a box's drift hits all variants alike:
    python tools/lc_ab.py [reps=2]
Prints frames/sec per run and the per-variant mean.  Dev tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (label, {"module.attribute": value} assigned after import, {"update tiling": int} via Update.tiling)
VARIANTS = [
    ("product", {}),
    ("radix plans (dpvo_amd.graph._PLAN_WIDE = False)", {"dpvo_amd.graph._PLAN_WIDE": False}),
    ("PatchGraph.normalize as torch operations (dpvo_amd.patchgraph._NORMALIZE_FUSED = False)", {"dpvo_amd.patchgraph._NORMALIZE_FUSED": False}),
    ("update operator: every kernel on the 4-wave geometry (tiling 1)", {"tiling": 1}),
    ("update operator: every kernel on the 12-wave geometry (tiling 29)", {"tiling": 29}),
]
if os.environ.get("LC_AB_ONLY"):       # comma-separated substrings of the variant names to keep
    keep = [k.strip() for k in os.environ["LC_AB_ONLY"].split(",")]
    VARIANTS = [v for v in VARIANTS if any(k in v[0] for k in keep)]
CHILD = r"""
import importlib, json, os, sys, torch
sys.path.insert(0, %r)
import bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
sets = json.loads(os.environ.get("LC_AB_SET", "{}"))
for k, v in sets.items():
    if k == "tiling":
        import dpvo_amd.net as N
        _init = N.Update.__init__
        def init(self, *a, _v=v, **kw):
            _init(self, *a, **kw); self.tiling = _v
        N.Update.__init__ = init
    else:
        mod, attr = k.rsplit(".", 1)
        setattr(importlib.import_module(mod), attr, v)
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
print("LC_AB " + json.dumps(bench.loop_closure_leg(cfg, 480, 640, dev, frames, intr, 64, seed=1234)))
""" % ROOT


def run(sets):
    env = dict(os.environ, LC_AB_SET=json.dumps(sets))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    for line in out.stdout.splitlines():
        if line.startswith("LC_AB "):
            return json.loads(line[6:])
    return {"frames_per_sec": None, "error": (out.stderr or out.stdout)[-400:]}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    res = {name: [] for name, _ in VARIANTS}
    for r in range(reps):
        for name, env in VARIANTS:
            d = run(env)
            res[name].append(d.get("frames_per_sec"))
            print(f"run {r}  {name:86s} {d.get('frames_per_sec')} frames/sec  global BAs {d.get('global_ba_runs')}  finite {d.get('finite')}"
                  + (f"  ERROR {d.get('error')}" if d.get("error") else ""), flush=True)
    print()
    for name, _ in VARIANTS:
        v = [x for x in res[name] if x]
        print(f"{name:86s} mean {sum(v) / max(len(v), 1):7.1f} frames/sec over {len(v)} runs  {v}")


if __name__ == "__main__":
    main()
