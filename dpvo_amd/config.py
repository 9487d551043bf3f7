"""Configuration node compatible with the reference's yacs `cfg` (dpvo/config.py:1-38) -- yacs itself is not a
dependency: this is a ~40-line attribute dict with `merge_from_file` / `merge_from_list` / `clone`."""
import copy

import yaml

_DEFAULTS = dict(
    BUFFER_SIZE=4096,
    CENTROID_SEL_STRAT='RANDOM',
    PATCHES_PER_FRAME=80,
    REMOVAL_WINDOW=20,
    OPTIMIZATION_WINDOW=12,
    PATCH_LIFETIME=12,
    KEYFRAME_INDEX=4,
    KEYFRAME_THRESH=12.5,
    MOTION_MODEL='DAMPED_LINEAR',
    MOTION_DAMPING=0.5,
    MIXED_PRECISION=True,
    LOOP_CLOSURE=False,
    BACKEND_THRESH=64.0,
    MAX_EDGE_AGE=1000,
    GLOBAL_OPT_FREQ=15,
    CLASSIC_LOOP_CLOSURE=False,
    LOOP_CLOSE_WINDOW_SIZE=3,
    LOOP_RETR_THRESH=0.04,
)

# config/default.yaml:4-17 and config/fast.yaml:4-17 of the reference
DEFAULT_YAML = dict(PATCHES_PER_FRAME=96, REMOVAL_WINDOW=22, OPTIMIZATION_WINDOW=10, PATCH_LIFETIME=13,
                    KEYFRAME_THRESH=15.0, MOTION_MODEL='DAMPED_LINEAR', MOTION_DAMPING=0.5, MIXED_PRECISION=True,
                    CENTROID_SEL_STRAT='RANDOM')
FAST_YAML = dict(PATCHES_PER_FRAME=48, REMOVAL_WINDOW=16, OPTIMIZATION_WINDOW=7, PATCH_LIFETIME=11,
                 KEYFRAME_THRESH=15.0, MOTION_MODEL='DAMPED_LINEAR', MOTION_DAMPING=0.5, MIXED_PRECISION=True,
                 CENTROID_SEL_STRAT='RANDOM')


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return CfgNode(copy.deepcopy(dict(self)))

    def merge_from_dict(self, d):
        for k, v in d.items():
            if k not in self:
                raise KeyError(f"Non-existent config key: {k}")
            self[k] = type(self[k])(v) if not isinstance(self[k], bool) else (v if isinstance(v, bool) else str(v) == 'True')

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        assert len(opts) % 2 == 0
        self.merge_from_dict(dict(zip(opts[0::2], opts[1::2])))


cfg = CfgNode(_DEFAULTS)


def default_config():
    c = cfg.clone()
    c.merge_from_dict(DEFAULT_YAML)
    return c
