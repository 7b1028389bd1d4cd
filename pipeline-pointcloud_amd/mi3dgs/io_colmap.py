"""COLMAP sparse-model binary I/O (cameras.bin / images.bin / points3D.bin).

This is the input side of `Train-Stage1`: the reference hands the trainer
`D/colmap/sparse/0/{cameras,images,points3D}.bin` plus `D/images[_k]/`
(source/container/src/main.py:2094-2095, 2098-2122) and itself only peeks at the first 8
bytes of points3D.bin (`read_colmap_points3d_count`, main.py:406-417).  Parsing was done by
nerfstudio's ColmapDataParser / gsplat's Parser upstream; this is
an independent reader of the published COLMAP binary layout, plus a writer used to make
synthetic datasets for the tests.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

# model id -> (name, number of parameters)
CAMERA_MODELS = {
    0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
    5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
    9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12),
}
MODEL_IDS = {v[0]: k for k, v in CAMERA_MODELS.items()}


@dataclass
class Camera:
    id: int
    model: str
    width: int
    height: int
    params: np.ndarray

    def pinhole(self) -> Tuple[float, float, float, float]:
        """(fx, fy, cx, cy); models with a single focal length repeat it."""
        p = self.params
        if self.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE", "RADIAL_FISHEYE"):
            return float(p[0]), float(p[0]), float(p[1]), float(p[2])
        return float(p[0]), float(p[1]), float(p[2]), float(p[3])

    def distortion(self) -> np.ndarray:
        n_lin = 3 if self.model in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL", "RADIAL", "SIMPLE_RADIAL_FISHEYE",
                                     "RADIAL_FISHEYE") else 4
        return np.asarray(self.params[n_lin:], dtype=np.float64)


@dataclass
class Image:
    id: int
    qvec: np.ndarray      # (qw, qx, qy, qz), world -> camera
    tvec: np.ndarray
    camera_id: int
    name: str

    def world_to_camera(self) -> np.ndarray:
        w, x, y, z = self.qvec
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=np.float64)
        M = np.eye(4)
        M[:3, :3] = R
        M[:3, 3] = self.tvec
        return M


def read_points3d_count(path: str) -> int:
    """Number of sparse points = first 8 bytes, little-endian u64 (reference main.py:406-417)."""
    if not os.path.isfile(path):
        return 0
    with open(path, "rb") as f:
        h = f.read(8)
    return int(struct.unpack("<Q", h)[0]) if len(h) == 8 else 0


def read_cameras(path: str) -> Dict[int, Camera]:
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cam_id, model_id, w, h = struct.unpack("<iiQQ", f.read(24))
            if model_id not in CAMERA_MODELS:
                raise ValueError(f"{path}: unknown COLMAP camera model id {model_id}")
            name, npar = CAMERA_MODELS[model_id]
            params = np.frombuffer(f.read(8 * npar), dtype="<f8").copy()
            out[cam_id] = Camera(cam_id, name, int(w), int(h), params)
    return out


def read_images(path: str) -> Dict[int, Image]:
    out = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            (img_id,) = struct.unpack("<i", f.read(4))
            q = np.frombuffer(f.read(32), dtype="<f8").copy()
            t = np.frombuffer(f.read(24), dtype="<f8").copy()
            (cam_id,) = struct.unpack("<i", f.read(4))
            name = bytearray()
            while True:
                c = f.read(1)
                if c in (b"\x00", b""):
                    break
                name += c
            (n2d,) = struct.unpack("<Q", f.read(8))
            f.seek(24 * n2d, os.SEEK_CUR)          # (x, y, point3D_id) triples are not needed
            out[img_id] = Image(img_id, q, t, cam_id, name.decode("utf-8"))
    return out


def read_points3d(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> xyz[P,3] float64, rgb[P,3] uint8, error[P] float64."""
    with open(path, "rb") as f:
        buf = f.read()
    (n,) = struct.unpack_from("<Q", buf, 0)
    xyz = np.empty((n, 3), np.float64)
    rgb = np.empty((n, 3), np.uint8)
    err = np.empty(n, np.float64)
    o = 8
    for i in range(n):
        # id u64 | xyz 3 f64 | rgb 3 u8 | error f64 | track_len u64 | track (u32,u32)*len
        xyz[i] = struct.unpack_from("<3d", buf, o + 8)
        rgb[i] = struct.unpack_from("<3B", buf, o + 32)
        err[i], tl = struct.unpack_from("<dQ", buf, o + 35)
        o += 51 + 8 * tl
    return xyz, rgb, err


# ------------------------------------------------------------------------------ writer
def write_model(sparse_dir: str, cameras: List[Camera], images: List[Image], xyz: np.ndarray, rgb: np.ndarray) -> None:
    """Writes a minimal but valid binary model (no 2-D observations, empty tracks)."""
    os.makedirs(sparse_dir, exist_ok=True)
    with open(os.path.join(sparse_dir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras:
            f.write(struct.pack("<iiQQ", c.id, MODEL_IDS[c.model], c.width, c.height))
            f.write(np.asarray(c.params, dtype="<f8").tobytes())
    with open(os.path.join(sparse_dir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for im in images:
            f.write(struct.pack("<i", im.id))
            f.write(np.asarray(im.qvec, dtype="<f8").tobytes())
            f.write(np.asarray(im.tvec, dtype="<f8").tobytes())
            f.write(struct.pack("<i", im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            f.write(struct.pack("<Q", 0))
    with open(os.path.join(sparse_dir, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        for i in range(len(xyz)):
            f.write(struct.pack("<Q3d3BdQ", i + 1, *[float(v) for v in xyz[i]], *[int(v) for v in rgb[i]], 0.5, 0))


def rotmat_to_qvec(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> (qw, qx, qy, qz), COLMAP convention."""
    m = R
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = 0.5 / np.sqrt(tr + 1.0)
        q = [0.25 / s, (m[2, 1] - m[1, 2]) * s, (m[0, 2] - m[2, 0]) * s, (m[1, 0] - m[0, 1]) * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = 2.0 * np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2])
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = 2.0 * np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2])
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = 2.0 * np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1])
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def find_sparse_dir(data_dir: str) -> str:
    """The layouts the reference produces: D/colmap/sparse/0 after the move at main.py:2094-2095
    (ns-train colmap dataparser), D/sparse/0 before it (gsplat Parser)."""
    for rel in ("colmap/sparse/0", "sparse/0", "colmap/sparse", "sparse"):
        d = os.path.join(data_dir, rel)
        if os.path.isfile(os.path.join(d, "cameras.bin")):
            return d
    raise FileNotFoundError(f"no COLMAP binary model (cameras.bin) under {data_dir}/colmap/sparse/0 or {data_dir}/sparse/0")
