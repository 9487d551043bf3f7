"""tools/evaluate.py end to end on the GPU: PNG frames on disk -> reader process -> DPVO -> terminate() -> TUM file + ATE."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_evaluate_runner_on_synthetic_frames(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    tex = rng.integers(0, 255, (200, 300, 3), dtype=np.uint8)
    n = 30
    for i in range(n):                                             # a textured plane sliding under the camera
        Image.fromarray(tex[i:i + 96, 2 * i:2 * i + 128]).save(tmp_path / f"{1000 + i:06d}.png")
    (tmp_path / "calib.txt").write_text("100.0 100.0 64.0 48.0")
    gt = tmp_path / "gt.txt"
    with open(gt, "w") as f:
        for i in range(n):
            f.write(f"{1000 + i} {0.02 * i} {0.01 * i} 0 0 0 0 1\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "evaluate.py"), "--random-weights", "--imagedir", str(tmp_path),
                          "--calib", str(tmp_path / "calib.txt"), "--groundtruth", str(gt), "--timestamps-from-names",
                          "--save-trajectory", str(tmp_path / "est.txt"), "--opts", "PATCHES_PER_FRAME", "16", "BUFFER_SIZE", "128"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["frames"] == n and rec["finite"] and rec["matched"] == n and np.isfinite(rec["ate_rmse_m"])
    rows = np.loadtxt(tmp_path / "est.txt")
    assert rows.shape == (n, 8) and rows[0, 0] == 1000.0
