"""Export surface of `Train-Stage1`: the gsplat `.pt` checkpoint, the 3DGS `.ply` and `.splat`.

PLY schema = what the reference's own consumers require:
  post_processing/gsplat_pt_to_ply.py:53-75  field order x y z nx ny nz f_dc_0..2 f_rest_0..44
      opacity scale_0..2 rot_0..3, f_rest channel-major (15 R, 15 G, 15 B);
  post_processing/spz/src/cc/load-spz.cc:572-750  strict parser: second line exactly
      `format binary_little_endian 1.0`, every property `property float`, rot_0 = w,
      opacity pre-sigmoid, scales log-space;
  post_processing/rotate_splat.py:54-74  reads the same names back.
`.splat` (32 bytes per Gaussian, position / scale / rgba8 / quat8) is an addition: the
reference has no writer for it.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch

SH_C0 = 0.28209479177387814
PLY_FIELDS = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] +
              ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def splats_to_rows(splats: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[N,62] float32 in PLY field order (on the tensors' device)."""
    n = splats["means"].shape[0]
    f_dc = splats["sh0"].reshape(n, 1, 3).transpose(1, 2).reshape(n, 3)
    f_rest = splats["shN"].reshape(n, 15, 3).transpose(1, 2).reshape(n, 45)        # channel-major
    return torch.cat([splats["means"].reshape(n, 3), torch.zeros(n, 3, device=f_dc.device, dtype=f_dc.dtype), f_dc,
                      f_rest, splats["opacities"].reshape(n, 1), splats["scales"].reshape(n, 3),
                      splats["quats"].reshape(n, 4)], dim=1).float().contiguous()


def write_ply(path: str, splats: Dict[str, torch.Tensor], drop_nonfinite: bool = True) -> int:
    """drop_nonfinite: ns-export removes Gaussians with any NaN/inf value; pass False to keep
    e.g. the `opacity = +inf` rows the reference's SPZ decoder emits for alpha 255/255."""
    rows = splats_to_rows(splats)
    if drop_nonfinite:
        rows = rows[torch.isfinite(rows).all(dim=1)]
    rows = rows.cpu().numpy().astype("<f4")
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % rows.shape[0]
    header += "".join(f"property float {name}\n" for name in PLY_FIELDS) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        rows.tofile(f)
    return int(rows.shape[0])


def read_ply(path: str) -> Dict[str, torch.Tensor]:
    """Reads a PLY written by write_ply / ns-export / the reference's SPZ converter."""
    with open(path, "rb") as f:
        names, n = [], 0
        line = f.readline().strip()
        if line != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = f.readline().strip()
        if fmt != b"format binary_little_endian 1.0":
            raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.split()
            if tok[:2] == [b"element", b"vertex"]:
                n = int(tok[2])
            elif tok[:1] == [b"property"]:
                if tok[1] != b"float":
                    raise ValueError(f"{path}: property {tok[2]!r} is not float")
                names.append(tok[2].decode())
            elif tok[:1] == [b"end_header"]:
                break
        data = np.fromfile(f, dtype="<f4", count=n * len(names)).reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    t = torch.from_numpy(data.copy())

    def cols(keys):
        return t[:, [col[k] for k in keys]]

    n_rest = sum(1 for k in names if k.startswith("f_rest_"))
    shN = torch.zeros(n, 15, 3)
    if n_rest:
        per = n_rest // 3
        r = cols([f"f_rest_{i}" for i in range(n_rest)]).reshape(n, 3, per).transpose(1, 2)
        shN[:, :per] = r
    return dict(means=cols(["x", "y", "z"]).contiguous(), sh0=cols(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(n, 1, 3).contiguous(),
                shN=shN, opacities=t[:, col["opacity"]].contiguous(), scales=cols(["scale_0", "scale_1", "scale_2"]).contiguous(),
                quats=cols(["rot_0", "rot_1", "rot_2", "rot_3"]).contiguous())


def save_checkpoint(path: str, splats: Dict[str, torch.Tensor], step: int) -> None:
    """gsplat simple_trainer checkpoint: {"step", "splats": state_dict} -- what
    post_processing/gsplat_pt_to_ply.py:45-50 loads with weights_only=True."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"step": int(step), "splats": {k: v.detach().cpu().contiguous() for k, v in splats.items()}}, path)


def load_checkpoint(path: str) -> Dict:
    return torch.load(path, map_location="cpu", weights_only=True)


def write_splat(path: str, splats: Dict[str, torch.Tensor]) -> int:
    """antimatter15 `.splat`: per Gaussian 3 f32 position, 3 f32 scale (linear), 4 u8 rgba,
    4 u8 quaternion (wxyz, 128 + 128 q); sorted by -size * opacity so viewers can truncate."""
    n = splats["means"].shape[0]
    means = splats["means"].reshape(n, 3).float().cpu()
    scales = torch.exp(splats["scales"].reshape(n, 3).float().cpu())
    alpha = torch.sigmoid(splats["opacities"].reshape(n).float().cpu())
    rgb = (0.5 + SH_C0 * splats["sh0"].reshape(n, 3).float().cpu()).clamp(0, 1)
    q = splats["quats"].reshape(n, 4).float().cpu()
    q = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    order = torch.argsort(-(scales.prod(1) * alpha))
    rec = np.zeros(n, dtype=[("pos", "<f4", 3), ("scale", "<f4", 3), ("rgba", "u1", 4), ("rot", "u1", 4)])
    rec["pos"] = means[order].numpy()
    rec["scale"] = scales[order].numpy()
    rgba = torch.cat([rgb, alpha[:, None]], 1)[order]
    rec["rgba"] = (rgba * 255.0).round().clamp(0, 255).to(torch.uint8).numpy()
    rec["rot"] = (q[order] * 128.0 + 128.0).round().clamp(0, 255).to(torch.uint8).numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    rec.tofile(path)
    return n
