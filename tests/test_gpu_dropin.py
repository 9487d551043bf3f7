"""The reference's own `demo.py`, UNMODIFIED (staged by oracle/build_ref.py under oracle/_ref/pyref/scripts/), on the MI355X: after
`dpvo_amd.compat.install()` its `from dpvo.dpvo import DPVO`, `from dpvo.config import cfg`, `from dpvo.stream import image_stream ...`
resolve to this package, and its `run()` (demo.py:25-56) -- reader process, `DPVO(cfg, network, ht=H, wd=W, viz=viz)`, `slam(t, image,
intrinsics)` per frame, `slam.pg.points_ / colors_ / slam.m`, `slam.terminate()` -- drives the tracker exactly as a user switching over
would.  What is asserted: the tracker built by THAT constructor call is the pipeline bench.py times -- deferred keyframe
record, encoders on a second stream, one C-ABI call per frame (`slam._fu is not None`) -- and the state demo.py reads between and after
the calls is current."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = os.path.join(ROOT, "oracle", "_ref", "pyref", "scripts")


def test_reference_demo_script_runs_the_one_call_pipeline(tmp_path, monkeypatch):
    if not os.path.isfile(os.path.join(SCRIPTS, "demo.py")):
        pytest.skip("oracle/_ref/pyref/scripts not staged (oracle/build_ref.py needs /root/reference)")
    from PIL import Image
    import dpvo_amd.compat as compat
    import dpvo_amd.dpvo as ours
    import dpvo_amd.stream          # noqa: F401  (binds its optional cv2 -- absent here: Pillow -- before the empty stand-in below exists)
    from dpvo_amd.net import VONet
    for name in ("cv2", "evo", "evo.main_ape", "evo.core", "evo.core.sync", "evo.core.metrics", "evo.core.trajectory", "evo.tools",
                 "evo.tools.file_interface", "evo.tools.plot", "plyfile"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    sys.modules["evo.core.metrics"].PoseRelation = object
    sys.modules["evo.core.trajectory"].PoseTrajectory3D = object
    sys.modules["evo.core"].sync = sys.modules["evo.core.sync"]
    sys.modules["evo.tools"].file_interface = sys.modules["evo.tools.file_interface"]
    sys.modules["evo.tools"].plot = sys.modules["evo.tools.plot"]
    sys.modules["evo"].main_ape = sys.modules["evo.main_ape"]
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    compat.install(force=True)
    monkeypatch.syspath_prepend(SCRIPTS)
    sys.modules.pop("demo", None)
    import demo
    assert demo.DPVO is ours.DPVO and demo.image_stream.__module__ == "dpvo_amd.stream"

    # 40 frames of a textured plane sliding under the camera; random-init weights saved the way the reference's checkpoints are
    rng = np.random.default_rng(0)
    tex = rng.integers(0, 255, (260, 400, 3), dtype=np.uint8)
    n = 40
    for i in range(n):
        Image.fromarray(tex[i:i + 96, 2 * i:2 * i + 128]).save(tmp_path / f"{i:06d}.png")
    (tmp_path / "calib.txt").write_text("100.0 100.0 64.0 48.0")
    torch.manual_seed(7)
    torch.save(VONet().state_dict(), tmp_path / "random.pth")
    cfg = demo.cfg.clone() if hasattr(demo.cfg, "clone") else demo.cfg
    cfg.merge_from_list(["PATCHES_PER_FRAME", "16", "BUFFER_SIZE", "128", "KEYFRAME_THRESH", "-1.0"])

    seen = {}
    real_init, real_term = ours.DPVO.__init__, ours.DPVO.terminate

    def init(self, *a, **k):
        seen["ctor"] = (a[2:], dict(k))                     # what demo.py:46 passes besides cfg and network
        real_init(self, *a, **k)
        self.motion_probe = lambda: 1.0e9                   # random weights: the initialisation probe means nothing (bench.py does the same)

    def term(self):
        seen["slam"] = self
        seen["fu"] = self._fu is not None
        seen["pending_at_terminate"] = self._fu_pending is not None or self._kf_pending is not None
        return real_term(self)
    monkeypatch.setattr(ours.DPVO, "__init__", init)
    monkeypatch.setattr(ours.DPVO, "terminate", term)
    (poses, tstamps), (points, colors, calib) = demo.run(cfg, str(tmp_path / "random.pth"), str(tmp_path), str(tmp_path / "calib.txt"), stride=1)
    slam = seen["slam"]
    assert seen["ctor"][0] == () and set(seen["ctor"][1]) == {"ht", "wd", "viz"}, seen["ctor"]       # demo.py:46, nothing of ours
    assert slam.defer_keyframe and slam.overlap_encoders, "the drop-in constructor must build the pipelined tracker"
    assert seen["fu"], "demo.run's call sequence must reach the one-call frame path (dpvo_frame_update)"
    # demo.py:52-53 read slam.pg.points_ / colors_ / slam.m BEFORE terminate(): the accessor has resolved the last frame's record
    assert not seen["pending_at_terminate"]
    assert slam.n == n and slam.m == n * 16 and points.shape == (n * 16, 3) and colors.shape == (n * 16, 3)
    assert poses.shape == (n, 7) and np.isfinite(poses).all() and len(tstamps) == n and calib[-2:] == (96, 128)
