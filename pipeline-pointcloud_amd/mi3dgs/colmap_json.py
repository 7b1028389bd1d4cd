"""COLMAP model -> nerfstudio `transforms.json` (+ ASCII point cloud): what the reference's
`training/colmap_to_nerfstudio_cam.py` (source/container/src/pipeline/training/, run as the
"Colmap-to-Nerfstudio" component, main.py:1220-1226) obtains from nerfstudio's `colmap_to_json`.
[UPSTREAM-UNVERIFIED]: nerfstudio is not in this image; the published behaviour is restated:
camera-to-world = inverse of COLMAP's world-to-camera, OpenCV -> OpenGL camera axes (y, z
negated), world axes permuted (x, z, -y) with the same `applied_transform` recorded, one shared
camera, `file_path` = images/<name>, points written as `ply_file_path`.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Dict

import numpy as np

from . import io_colmap

APPLIED = np.array([[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, -1.0, 0.0, 0.0]])     # rows (0, 2, 1), third negated


def camera_params(cam: io_colmap.Camera) -> Dict:
    """nerfstudio parse_colmap_camera_params: intrinsics + distortion in its key names."""
    p = [float(v) for v in cam.params]
    out = {"w": int(cam.width), "h": int(cam.height)}
    model = "OPENCV"
    k = dict(k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    if cam.model == "SIMPLE_PINHOLE":
        fl_x = fl_y = p[0]; cx, cy = p[1], p[2]
    elif cam.model == "PINHOLE":
        fl_x, fl_y, cx, cy = p[0], p[1], p[2], p[3]
    elif cam.model == "SIMPLE_RADIAL":
        fl_x = fl_y = p[0]; cx, cy = p[1], p[2]; k["k1"] = p[3]
    elif cam.model == "RADIAL":
        fl_x = fl_y = p[0]; cx, cy = p[1], p[2]; k["k1"], k["k2"] = p[3], p[4]
    elif cam.model == "OPENCV":
        fl_x, fl_y, cx, cy = p[0], p[1], p[2], p[3]
        k["k1"], k["k2"], k["p1"], k["p2"] = p[4], p[5], p[6], p[7]
    elif cam.model == "OPENCV_FISHEYE":
        fl_x, fl_y, cx, cy = p[0], p[1], p[2], p[3]
        k = dict(k1=p[4], k2=p[5], k3=p[6], k4=p[7])
        model = "OPENCV_FISHEYE"
    elif cam.model == "SIMPLE_RADIAL_FISHEYE":
        fl_x = fl_y = p[0]; cx, cy = p[1], p[2]
        k = dict(k1=p[3], k2=0.0, k3=0.0, k4=0.0)
        model = "OPENCV_FISHEYE"
    elif cam.model == "RADIAL_FISHEYE":
        fl_x = fl_y = p[0]; cx, cy = p[1], p[2]
        k = dict(k1=p[3], k2=p[4], k3=0.0, k4=0.0)
        model = "OPENCV_FISHEYE"
    else:
        raise NotImplementedError(f"{cam.model} camera model is not supported yet!")
    out.update(fl_x=fl_x, fl_y=fl_y, cx=cx, cy=cy, **k)
    out["camera_model"] = model
    return out


def colmap_to_json(recon_dir, output_dir, ply_filename: str = "sparse_pc.ply", keep_original_world_coordinate: bool = False) -> int:
    """Writes `output_dir/transforms.json` and the point cloud; returns the number of frames."""
    recon_dir, output_dir = Path(recon_dir), Path(output_dir)
    cams = io_colmap.read_cameras(str(recon_dir / "cameras.bin"))
    imgs = io_colmap.read_images(str(recon_dir / "images.bin"))
    if set(cams.keys()) != {1}:
        raise RuntimeError("Only single camera shared for all images is supported.")
    frames = []
    for im_id in sorted(imgs):
        im = imgs[im_id]
        c2w = np.linalg.inv(im.world_to_camera())
        c2w[0:3, 1:3] *= -1.0                                   # OpenCV camera axes -> OpenGL
        if not keep_original_world_coordinate:
            c2w = c2w[np.array([0, 2, 1, 3]), :]
            c2w[2, :] *= -1.0
        frames.append({"file_path": (Path("./images") / im.name).as_posix(), "transform_matrix": c2w.tolist(),
                       "colmap_im_id": int(im_id)})
    out = camera_params(cams[1])
    out["frames"] = frames
    applied = None
    if not keep_original_world_coordinate:
        applied = APPLIED
        out["applied_transform"] = applied.tolist()
    assert str(ply_filename).endswith(".ply"), f"ply_filename: {ply_filename} does not end with '.ply'"
    write_points_ply(recon_dir, output_dir / ply_filename, applied)
    out["ply_file_path"] = str(ply_filename)
    with open(output_dir / "transforms.json", "w", encoding="utf-8") as f:
        json.dump(out, f, indent=4)
    return len(frames)


def write_points_ply(recon_dir, path, applied=None) -> int:
    """nerfstudio create_ply_from_colmap: ASCII PLY, float xyz (in the transformed world) + uint8 rgb."""
    xyz, rgb, _ = io_colmap.read_points3d(str(Path(recon_dir) / "points3D.bin"))
    pts = np.asarray(xyz, dtype=np.float32)
    if applied is not None:
        pts = (np.asarray(applied[:, :3], dtype=np.float32) @ pts.T).T + np.asarray(applied[:, 3], dtype=np.float32)
    os.makedirs(os.path.dirname(os.path.abspath(str(path))), exist_ok=True)
    with open(path, "w", encoding="utf-8") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {len(pts)}\n")
        f.write("property float x\nproperty float y\nproperty float z\n")
        f.write("property uint8 red\nproperty uint8 green\nproperty uint8 blue\nend_header\n")
        for (x, y, z), (r, g, b) in zip(pts, np.asarray(rgb, dtype=np.uint8)):
            f.write(f"{x:8f} {y:8f} {z:8f} {int(r)} {int(g)} {int(b)}\n")
    return len(pts)


def main(argv=None) -> int:
    """`colmap_to_nerfstudio_cam.py -d DATA_DIR`: same arguments, messages and failure text as the reference script."""
    import argparse
    ap = argparse.ArgumentParser(prog="create-transform", description="Create the NeRF Studio transform for COLMAP input data")
    ap.add_argument("-d", "--data_dir", required=True, default=None, action="store",
                    help="Target data directory for the COLMAP project root directory")
    a = ap.parse_args(argv)
    path = str(a.data_dir)
    sparse_path = f"{path}/sparse/0"
    ply_path = f"{sparse_path}/sparse.ply"
    if os.path.isdir(path):
        if os.path.isdir(sparse_path):
            print("Input path exists...creating transforms.json file")
            try:
                print(f"Sparse Path: {sparse_path}")
                print(f"PLY Filename: {ply_path}")
                colmap_to_json(recon_dir=Path(sparse_path), output_dir=Path(path), ply_filename=ply_path)
            except Exception as e:      # noqa: BLE001 - the reference converts everything into RuntimeError
                raise RuntimeError(f"Script failed to complete successfully: {e}") from e
        else:
            print(f"Sparse path does not currently exist: {sparse_path}")
    else:
        print(f"Input path: {path} doesn't exist...exiting")
    return 0
