#!/usr/bin/env python
"""The config-5 leg of bench.py (LOOP_CLOSURE=True, 45 timed frames, 18 of them with a global BA) under the measurement switches of
round 5, each variant in its own process (the switches are read at import), `reps` times each, interleaved so that a box's drift hits
all variants alike:
    python tools/lc_ab.py [reps=2]
Prints frames/sec per run and the per-variant mean; also the mean wall time of the global-BA frames with a device sync per frame
(LC_SYNC=1 of tools/lc_profile.py) for the first and the last variant.  Dev tool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [
    ("product", {}),
    ("radix plans (DPVO_PLAN_WIDE=0)", {"DPVO_PLAN_WIDE": "0"}),
    ("PatchGraph.normalize as torch operations (DPVO_NORMALIZE_FUSED=0)", {"DPVO_NORMALIZE_FUSED": "0"}),
    ("frame state entry by entry (DPVO_COMPOSITE_LR=0)", {"DPVO_COMPOSITE_LR": "0"}),
    ("plan in front of reproject / corr (DPVO_PLAN_FIRST=1)", {"DPVO_PLAN_FIRST": "1"}),
    ("all switches back", {"DPVO_PLAN_WIDE": "0", "DPVO_GBA_CAT": "1", "DPVO_CHOL_BACK_STEPS": "1", "DPVO_COMPOSITE_LR": "0",
                           "DPVO_NORMALIZE_FUSED": "0", "DPVO_PLAN_FIRST": "1"}),
]
if os.environ.get("LC_AB_ALL"):        # the two switches that measured within the noise (profiles/r05_f_lc_ab.txt)
    VARIANTS[3:3] = [("five torch.cat (DPVO_GBA_CAT=1)", {"DPVO_GBA_CAT": "1"}),
                     ("back substitution per column (DPVO_CHOL_BACK_STEPS=1)", {"DPVO_CHOL_BACK_STEPS": "1"})]
_BV0 = os.path.join(ROOT, "dpvo_amd", "libdpvo_hip_bv0.so")          # tools/gba_bv_ab.sh build: the row kernel of rounds 4-5
if os.path.exists(_BV0):
    _LBL = os.environ.get("LC_AB_LIB_LABEL", "row kernel of rounds 4-5")
    VARIANTS.insert(len(VARIANTS) - 1, (_LBL + " (libdpvo_hip_bv0.so)", {"DPVO_HIP_LIB": _BV0}))
    VARIANTS[-1] = (VARIANTS[-1][0] + " + " + _LBL, dict(VARIANTS[-1][1], DPVO_HIP_LIB=_BV0))
if os.environ.get("LC_AB_ONLY"):       # comma-separated substrings of the variant names to keep
    keep = [k.strip() for k in os.environ["LC_AB_ONLY"].split(",")]
    VARIANTS = [v for v in VARIANTS if any(k in v[0] for k in keep)]
CHILD = r"""
import json, sys, torch
sys.path.insert(0, %r)
import bench
from dpvo_amd.config import cfg as base_cfg, DEFAULT_YAML
dev = torch.device("cuda:0")
cfg = base_cfg.clone(); cfg.merge_from_dict(DEFAULT_YAML); cfg.KEYFRAME_THRESH = -1.0
frames = bench.make_stream(64, 480, 640, dev)
intr = torch.tensor([320.0, 320.0, 320.0, 240.0], device=dev)
print("LC_AB " + json.dumps(bench.loop_closure_leg(cfg, 480, 640, dev, frames, intr, 64, seed=1234)))
""" % ROOT


def run(env_extra):
    env = dict(os.environ, **env_extra)
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
    for name, env in ((VARIANTS[0], VARIANTS[-1]) if not os.environ.get("LC_AB_ONLY") else ()):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lc_profile.py")], env=dict(os.environ, LC_SYNC="1", **env),
                             capture_output=True, text=True, timeout=600)
        for line in out.stdout.splitlines():
            if line.startswith("global-BA frames"):
                print(f"{name}: {line}")


if __name__ == "__main__":
    main()
