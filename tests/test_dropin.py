"""Drop-in surface: the reference's scripts resolve their `dpvo.*` imports against dpvo_amd (compat.install()), and the
frame readers / result writers they use exist with the same call signatures."""
import inspect
import os
import sys
import types
from multiprocessing import Queue

import numpy as np
import pytest

REF = "/root/reference"


def test_stream_reads_a_directory_like_the_reference(tmp_path):
    from PIL import Image
    from dpvo_amd import stream
    rng = np.random.default_rng(0)
    for i in range(5):
        Image.fromarray(rng.integers(0, 255, (50, 70, 3), dtype=np.uint8)).save(tmp_path / f"{i:04d}.png")
    (tmp_path / "calib.txt").write_text("320.0 320.0 35.0 25.0")
    q = Queue()
    stream.image_stream(q, str(tmp_path), str(tmp_path / "calib.txt"), 2, 0)
    got = []
    while True:
        t, img, intr = q.get(timeout=10)
        if t < 0:
            break
        got.append((t, img, intr))
    assert [g[0] for g in got] == [0, 1, 2]                      # frames 0, 2, 4 at stride 2
    assert got[0][1].shape == (48, 64, 3) and got[0][1].dtype == np.uint8      # cropped to multiples of 16 (stream.py:36-37)
    assert np.allclose(got[0][2], [320, 320, 35, 25])
    ref = np.asarray(Image.open(tmp_path / "0000.png"))[:48, :64, ::-1]        # BGR like cv2.imread
    assert np.array_equal(got[0][1], ref)


def test_plot_utils_writers(tmp_path):
    from dpvo_amd import plot_utils as PU
    ts = np.arange(20, dtype=np.float64)
    poses = np.zeros((20, 7)); poses[:, 0] = ts * 0.1; poses[:, 6] = 1
    PU.plot_trajectory((ts, poses), (ts, poses * 1.0), "t", str(tmp_path / "p.pdf"))
    assert (tmp_path / "p.pdf").stat().st_size > 1000
    PU.save_ply(str(tmp_path / "cloud"), np.random.rand(10, 3), np.random.randint(0, 255, (10, 3)))
    head = (tmp_path / "cloud.ply").read_bytes()
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 10\n") and len(head) > 10 * 15
    PU.save_output_for_COLMAP(str(tmp_path / "colmap"), (ts, poses), np.random.rand(10, 3), np.zeros((10, 3), np.uint8), 1, 1, 0, 0)
    assert len((tmp_path / "colmap" / "images.txt").read_text().splitlines()) == 40


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (this container only)")
def test_reference_scripts_import_and_call_us_unchanged(tmp_path, monkeypatch):
    """demo.py and evaluate_euroc.py of the reference, imported UNMODIFIED after compat.install(): every `dpvo.*` import they make
    resolves to dpvo_amd (stream and plot_utils included), and their own `run()` drives our classes with arguments our signatures
    accept -- checked by running evaluate_euroc.run on 6 synthetic PNG frames with the tracker replaced by a signature-checking
    recorder (no GPU here; the real tracker behind the same calls is exercised by tests/test_gpu_evaluate.py)."""
    from PIL import Image
    import torch
    import dpvo_amd.compat as compat
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
    import dpvo_amd.dpvo as ours
    real_sig = inspect.signature(ours.DPVO.__init__)
    call_sig = inspect.signature(ours.DPVO.__call__)
    calls = []

    class Recorder:
        def __init__(self, *a, **k):
            real_sig.bind(self, *a, **k)                         # raises TypeError if the reference passes something we do not take
            calls.append(("init", k))
            self.m = 0
            self.pg = types.SimpleNamespace(points_=torch.zeros(4, 3), colors_=torch.zeros(1, 4, 3, dtype=torch.uint8))

        def __call__(self, *a, **k):
            call_sig.bind(self, *a, **k)
            t, image, intrinsics = a
            assert image.dtype == torch.uint8 and image.shape[0] == 3 and intrinsics.shape == (4,)
            calls.append(("frame", t))

        def terminate(self):
            return np.zeros((len([c for c in calls if c[0] == "frame"]), 7)), np.arange(3.0)
    monkeypatch.setattr(ours, "DPVO", Recorder)
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.syspath_prepend(REF)
    for mod in ("evaluate_euroc", "demo"):
        sys.modules.pop(mod, None)
    import evaluate_euroc
    import demo
    assert evaluate_euroc.DPVO is Recorder and demo.image_stream.__module__ == "dpvo_amd.stream"
    assert evaluate_euroc.plot_trajectory.__module__ == "dpvo_amd.plot_utils" and demo.save_ply.__module__ == "dpvo_amd.plot_utils"
    rng = np.random.default_rng(0)
    for i in range(6):
        Image.fromarray(rng.integers(0, 255, (64, 96, 3), dtype=np.uint8)).save(tmp_path / f"{i:04d}.png")
    (tmp_path / "calib.txt").write_text("100.0 100.0 48.0 32.0")
    poses, ts = evaluate_euroc.run(evaluate_euroc.cfg, "dpvo.pth", str(tmp_path), str(tmp_path / "calib.txt"), stride=2)
    assert [c for c in calls if c[0] == "frame"] == [("frame", 0), ("frame", 1), ("frame", 2)] and poses.shape == (3, 7)
    (p2, t2), (pts, cols, calib) = demo.run(demo.cfg, "dpvo.pth", str(tmp_path), str(tmp_path / "calib.txt"), stride=1)
    assert p2.shape[0] == 9 and calib[-2:] == (64, 96)
