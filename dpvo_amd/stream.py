"""Frame readers with the interface of the reference's dpvo/stream.py (`image_stream`, `video_stream`: a producer that puts
`(t, image HxWx3 uint8 BGR, intrinsics [fx fy cx cy])` on a queue and ends with `(-1, ...)`), for demo.py / evaluate_*.py and for
tools/evaluate.py.  Host-side I/O around the hot path (out of the measured path).

Decoding: OpenCV when it is installed (then also lens undistortion and video files, as in the reference); otherwise Pillow for
the image formats; `.npy` frames always work.  Images are cropped to multiples of 16 exactly like stream.py:36-37."""
import os
from itertools import chain
from pathlib import Path

import numpy as np

try:                                  # optional, exactly what the reference uses
    import cv2
except ImportError:                   # pragma: no cover
    cv2 = None


def load_calib(calib):
    """calibration text file: fx fy cx cy [k1 k2 p1 p2 ...]  ->  (intrinsics [4], K [3,3], distortion or None)"""
    c = np.loadtxt(calib, delimiter=" ").reshape(-1)
    fx, fy, cx, cy = c[:4]
    K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
    return np.array([fx, fy, cx, cy]), K, (c[4:] if c.size > 4 else None)


def read_image(path):
    """-> HxWx3 uint8 in BGR channel order (what cv2.imread returns and the tracker's colour gather expects)"""
    path = str(path)
    if path.endswith(".npy"):
        img = np.load(path)
    elif cv2 is not None:
        img = cv2.imread(path)
    else:
        from PIL import Image
        img = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
    if img is None:
        raise IOError(f"cannot read {path}")
    if img.ndim == 2:
        img = np.repeat(img[:, :, None], 3, axis=2)
    return np.ascontiguousarray(img[:, :, :3], dtype=np.uint8)


def _undistort(img, K, dist):
    if dist is None or not np.any(dist):
        return img
    if cv2 is None:
        raise RuntimeError("the calibration file has distortion coefficients: lens undistortion needs OpenCV (cv2)")
    return cv2.undistort(img, K, dist)


def _crop16(img):
    h, w = img.shape[:2]
    return img[:h - h % 16, :w - w % 16]


def list_images(imagedir, stride=1, skip=0):
    exts = ("*.png", "*.jpeg", "*.jpg", "*.npy")
    return sorted(chain.from_iterable(Path(imagedir).glob(e) for e in exts))[skip::stride]


def image_stream(queue, imagedir, calib, stride, skip=0):
    """directory of frames -> queue (stream.py:8-42)"""
    intr, K, dist = load_calib(calib)
    assert os.path.exists(imagedir), imagedir
    image = None
    for t, f in enumerate(list_images(imagedir, stride, skip)):
        image = _crop16(_undistort(read_image(f), K, dist))
        queue.put((t, image, intr.copy()))
    queue.put((-1, image, intr.copy()))


def video_stream(queue, imagedir, calib, stride, skip=0):
    """video file -> queue, frames halved like the reference (stream.py:45-95); needs OpenCV"""
    if cv2 is None:
        raise RuntimeError("video input needs OpenCV (cv2)")
    intr, K, dist = load_calib(calib)
    assert os.path.exists(imagedir), imagedir
    cap = cv2.VideoCapture(imagedir)
    for _ in range(skip):
        cap.read()
    t, image = 0, None
    while True:
        ok = False
        for _ in range(stride):
            ok, frame = cap.read()
            if not ok:
                break
        if not ok:
            break
        frame = _undistort(frame, K, dist)
        frame = cv2.resize(frame, None, fx=0.5, fy=0.5, interpolation=cv2.INTER_AREA)
        image = _crop16(frame)
        queue.put((t, image, intr * 0.5))
        t += 1
    queue.put((-1, image, intr * 0.5))
    cap.release()
