"""Rotate / mirror a trained splat: the reference's `rotate_splat.py` and `mirror_splat.py`
(source/container/src/pipeline/post_processing/, invoked at main.py:1481-1523, 1556-1592) as
batched tensor operations on whatever device the splats live on.

The reference loops over Gaussians in Python through scipy and transforms the SH coefficients
only approximately: it multiplies `f_rest[:, 0:9]` three at a time by R^T, which (f_rest being
channel-major) touches the red channel's band 1 and part of its band 2 only, and leaves bands
2-3 of every channel unrotated.  Two modes:
  sh_mode="reference": that arithmetic exactly, so the output PLY equals the reference's;
  sh_mode="exact"    : the real-SH rotation of bands 1-3 (view-dependent colour then really is
                       the rotated scene's: rendering the rotated splats from the rotated camera
                       reproduces the original image; tests/test_gpu_transform.py).
Quaternions are wxyz, composed as q_R (x) q (rotation) / M R_g with a column flip (mirror), and
returned normalised, as scipy's `Rotation` does.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)


def create_rotation_matrix(axis: str, angle_degrees: float) -> torch.Tensor:
    a = math.radians(angle_degrees)
    c, s = math.cos(a), math.sin(a)
    if axis == "x":
        m = [[1, 0, 0], [0, c, -s], [0, s, c]]
    elif axis == "y":
        m = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
    else:
        m = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    return torch.tensor(m, dtype=torch.float64)


def parse_rotation_spec(spec: str) -> List[Tuple[str, float]]:
    """"x:270,y:180,z:0" -> [("x", 270.0), ...]; malformed parts are skipped, as the reference does."""
    out = []
    for part in (spec or "").split(","):
        if ":" in part:
            axis, angle = part.split(":")
            axis = axis.strip().lower()
            if axis in ("x", "y", "z"):
                try:
                    out.append((axis, float(angle.strip())))
                except ValueError:
                    pass
    return out


def _basis(d: torch.Tensor) -> torch.Tensor:
    """Real SH basis, bands 0-3, the engine's convention (csrc/project.hip sh_basis)."""
    d = d / d.norm(dim=-1, keepdim=True)
    x, y, z = d.unbind(-1)
    xx, yy, zz = x * x, y * y, z * z
    return torch.stack([
        torch.full_like(x, SH_C0), -SH_C1 * y, SH_C1 * z, -SH_C1 * x,
        SH_C2[0] * x * y, SH_C2[1] * y * z, SH_C2[2] * (2 * zz - xx - yy), SH_C2[3] * x * z, SH_C2[4] * (xx - yy),
        SH_C3[0] * y * (3 * xx - yy), SH_C3[1] * x * y * z, SH_C3[2] * y * (4 * zz - xx - yy),
        SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy), SH_C3[4] * x * (4 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
        SH_C3[6] * x * (xx - 3 * yy)], dim=-1)


def sh_rotation_matrices(M: torch.Tensor) -> List[torch.Tensor]:
    """Per-band matrices D_l (l = 1..3) with c'_l = D_l c_l for the orthogonal map x -> M x
    (rotation or reflection): f'(d) = f(M^T d).  Solved by least squares on fixed directions."""
    g = torch.Generator().manual_seed(1234)
    d = torch.randn(256, 3, generator=g, dtype=torch.float64)
    Y = _basis(d)                       # f' sampled at d ...
    Ysrc = _basis(d @ M.double())       # ... equals f sampled at M^T d   (row vectors: d M)
    out = []
    for lo, hi in ((1, 4), (4, 9), (9, 16)):
        D = torch.linalg.lstsq(Y[:, lo:hi], Ysrc[:, lo:hi]).solution     # Y D = Ysrc  ->  c' = D c
        out.append(D)
    return out


def _quat_mul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def _mat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """[...,3,3] proper rotations -> wxyz (branch on the largest diagonal term, like scipy)."""
    m = R
    t = torch.stack([m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2], m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]], -1)
    k = t.argmax(-1)
    w = torch.empty(R.shape[:-2] + (4,), dtype=R.dtype, device=R.device)
    c0 = torch.stack([1 + m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2], m[..., 2, 1] - m[..., 1, 2],
                      m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]], -1)
    c1 = torch.stack([m[..., 2, 1] - m[..., 1, 2], 1 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2],
                      m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0]], -1)
    c2 = torch.stack([m[..., 0, 2] - m[..., 2, 0], m[..., 0, 1] + m[..., 1, 0],
                      1 - m[..., 0, 0] + m[..., 1, 1] - m[..., 2, 2], m[..., 1, 2] + m[..., 2, 1]], -1)
    c3 = torch.stack([m[..., 1, 0] - m[..., 0, 1], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1],
                      1 - m[..., 0, 0] - m[..., 1, 1] + m[..., 2, 2]], -1)
    for i, c in enumerate((c0, c1, c2, c3)):
        sel = k == i
        w[sel] = c[sel]
    return w / w.norm(dim=-1, keepdim=True)


def _quat_to_mat(q: torch.Tensor) -> torch.Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))


def _apply_sh(splats: Dict[str, torch.Tensor], M: torch.Tensor, sh_mode: str, reference_right_multiplier: torch.Tensor):
    shN = splats["shN"]
    n = shN.shape[0]
    if sh_mode == "reference":
        # f_rest is channel-major [15 R | 15 G | 15 B]; the reference multiplies columns 0:9, three at a time
        rest = shN.reshape(n, 15, 3).transpose(1, 2).reshape(n, 45).clone()
        W = reference_right_multiplier.to(rest.dtype).to(rest.device)
        src = rest.clone()
        for i in range(0, 9, 3):
            rest[:, i:i + 3] = src[:, i:i + 3] @ W
        return rest.reshape(n, 3, 15).transpose(1, 2).contiguous()
    if sh_mode != "exact":
        raise ValueError("sh_mode must be 'exact' or 'reference'")
    out = shN.reshape(n, 15, 3).clone()
    for (lo, hi), D in zip(((0, 3), (3, 8), (8, 15)), sh_rotation_matrices(M)):
        out[:, lo:hi, :] = torch.einsum("kj,njc->nkc", D.to(out.dtype).to(out.device), shN.reshape(n, 15, 3)[:, lo:hi, :])
    return out


def rotate_splats(splats: Dict[str, torch.Tensor], R: torch.Tensor, sh_mode: str = "exact") -> Dict[str, torch.Tensor]:
    """x -> R x for one rotation matrix (reference rotate_gaussians)."""
    dt, dev = splats["means"].dtype, splats["means"].device
    Rd = R.to(dt).to(dev)
    out = dict(splats)
    out["means"] = splats["means"] @ Rd.T
    qR = _mat_to_quat(R.double()).to(dt).to(dev)
    q = splats["quats"] / splats["quats"].norm(dim=-1, keepdim=True)
    out["quats"] = _quat_mul(qR.expand_as(q), q)
    out["shN"] = _apply_sh(splats, R, sh_mode, R.double().T)
    return out


def mirror_splats(splats: Dict[str, torch.Tensor], axis: str, sh_mode: str = "exact") -> Dict[str, torch.Tensor]:
    """Reflection of one coordinate (reference mirror_ply)."""
    dt, dev = splats["means"].dtype, splats["means"].device
    M = torch.eye(3, dtype=torch.float64)
    M["xyz".index(axis), "xyz".index(axis)] = -1.0
    Md = M.to(dt).to(dev)
    out = dict(splats)
    out["means"] = splats["means"] @ Md
    Rm = Md @ _quat_to_mat(splats["quats"])
    Rm = Rm.clone()
    Rm[..., :, 0] = -Rm[..., :, 0]                     # det(M R) = -1 always: flip one column (covariance unchanged)
    out["quats"] = _mat_to_quat(Rm)
    out["shN"] = _apply_sh(splats, M, sh_mode, M)
    return out


def rotate_ply(path_in: str, path_out: str, rotations: str, sh_mode: str = "exact", device=None) -> int:
    from . import io_ply
    S = io_ply.read_ply(path_in)
    if device is not None:
        S = {k: v.to(device) for k, v in S.items()}
    for axis, angle in parse_rotation_spec(rotations):
        S = rotate_splats(S, create_rotation_matrix(axis, angle), sh_mode)
    return io_ply.write_ply(path_out or path_in, S, drop_nonfinite=False)


def mirror_ply(path_in: str, path_out: str, axis: str = "x", sh_mode: str = "exact", device=None) -> int:
    from . import io_ply
    S = io_ply.read_ply(path_in)
    if device is not None:
        S = {k: v.to(device) for k, v in S.items()}
    return io_ply.write_ply(path_out or path_in, mirror_splats(S, axis, sh_mode), drop_nonfinite=False)


def main_rotate(argv=None) -> int:
    """Same flags as the reference script: [-i IN] [-o OUT] --rotations x:270,y:180,z:0 | --axis A --angle DEG."""
    import argparse
    ap = argparse.ArgumentParser(description="Rotate a Gaussian Splatting PLY (mi3dgs)")
    ap.add_argument("input_ply", nargs="?", default=None)
    ap.add_argument("output_ply", nargs="?", default=None)
    ap.add_argument("--input", "-i", default=None)
    ap.add_argument("--output", "-o", default=None)
    ap.add_argument("--angle", type=float, default=None)
    ap.add_argument("--axis", choices=["x", "y", "z"], default=None)
    ap.add_argument("--rotations", default=None)
    ap.add_argument("--sh-mode", choices=["exact", "reference"], default="reference",
                    help="reference: the upstream script's SH arithmetic (identical file); exact: true rotation of bands 1-3")
    a = ap.parse_args(argv)
    src = a.input or a.input_ply
    if src is None:
        ap.error("Input path is required. Use positional argument or --input/-i flag.")
    spec = a.rotations or (f"{a.axis}:{a.angle}" if a.axis and a.angle is not None else "")
    if not parse_rotation_spec(spec):
        print("No rotations specified, exiting")
        return 0
    dev = "cuda" if torch.cuda.is_available() else None
    n = rotate_ply(src, a.output or a.output_ply or src, spec, a.sh_mode, dev)
    print(f"[mi3dgs] rotated {n} Gaussians ({spec}, SH {a.sh_mode})")
    return 0


def main_mirror(argv=None) -> int:
    import argparse
    ap = argparse.ArgumentParser(description="Mirror a Gaussian Splatting PLY (mi3dgs)")
    ap.add_argument("--input", "-i", required=True)
    ap.add_argument("--output", "-o", default=None)
    ap.add_argument("--axis", "-a", choices=["x", "y", "z"], default="x")
    ap.add_argument("--sh-mode", choices=["exact", "reference"], default="reference",
                    help="reference: the upstream script's SH arithmetic (identical file); exact: true rotation of bands 1-3")
    a = ap.parse_args(argv)
    dev = "cuda" if torch.cuda.is_available() else None
    n = mirror_ply(a.input, a.output, a.axis, a.sh_mode, dev)
    print(f"[mi3dgs] mirrored {n} Gaussians along {a.axis} (SH {a.sh_mode})")
    return 0
