"""Rotate / mirror post-processing: `reference` mode against the numpy/scipy
restatement of the reference scripts, `exact` mode against SH theory."""
import os

import numpy as np
import pytest
import torch

from mi3dgs import io_ply, transform
from oracle import gs_oracle as O
from oracle import post_oracle as PO

WOLF = os.path.join(os.path.dirname(__file__), "golden", "wolf_1k.ply")


def _splats(n=200, seed=0):
    g = torch.Generator().manual_seed(seed)
    return dict(means=torch.randn(n, 3, generator=g, dtype=torch.float64),
                quats=torch.randn(n, 4, generator=g, dtype=torch.float64), scales=torch.randn(n, 3, generator=g, dtype=torch.float64),
                opacities=torch.randn(n, generator=g, dtype=torch.float64), sh0=torch.randn(n, 1, 3, generator=g, dtype=torch.float64),
                shN=torch.randn(n, 15, 3, generator=g, dtype=torch.float64))


def _f_rest(S):
    n = S["shN"].shape[0]
    return S["shN"].transpose(1, 2).reshape(n, 45).numpy()


def _same_rotation(q_a, q_b, tol=1e-9):
    d = np.abs((q_a * q_b).sum(1))          # q and -q are the same rotation
    return np.all(np.abs(d - 1) < tol)


@pytest.mark.parametrize("spec", ["x:270,y:180,z:0", "x:180,y:180", "y:-90", "z:37.5,x:12"])
def test_reference_mode_equals_the_reference_arithmetic(spec):
    S = _splats()
    pos, quat, dc, rest = S["means"].numpy(), S["quats"].numpy(), S["sh0"][:, 0].numpy(), _f_rest(S)
    for axis, angle in PO.parse_rotation_spec(spec):
        pos, quat, dc, rest = PO.rotate_gaussians(pos, quat, dc, rest, PO.create_rotation_matrix(axis, angle))
    T = S
    assert transform.parse_rotation_spec(spec) == PO.parse_rotation_spec(spec)
    for axis, angle in transform.parse_rotation_spec(spec):
        T = transform.rotate_splats(T, transform.create_rotation_matrix(axis, angle), "reference")
    assert np.allclose(T["means"].numpy(), pos, atol=1e-12)
    assert _same_rotation(T["quats"].numpy(), quat)
    assert np.allclose(_f_rest(T), rest, atol=1e-12) and torch.equal(T["sh0"], S["sh0"])
    assert torch.equal(T["scales"], S["scales"]) and torch.equal(T["opacities"], S["opacities"])


@pytest.mark.parametrize("axis", ["x", "y", "z"])
def test_mirror_reference_mode_equals_the_reference_arithmetic(axis):
    S = _splats(seed=1)
    pos, quat, rest = PO.mirror(S["means"].numpy(), S["quats"].numpy(), _f_rest(S), axis)
    T = transform.mirror_splats(S, axis, "reference")
    assert np.allclose(T["means"].numpy(), pos, atol=1e-12) and _same_rotation(T["quats"].numpy(), quat)
    assert np.allclose(_f_rest(T), rest, atol=1e-12)
    # the covariance is reflected: Sigma' = M Sigma M
    M = np.eye(3); M["xyz".index(axis), "xyz".index(axis)] = -1
    cov = O.quat_scale_to_covar(S["quats"], S["scales"].exp()).numpy()
    cov2 = O.quat_scale_to_covar(T["quats"], T["scales"].exp()).numpy()
    assert np.allclose(cov2, M @ cov @ M, atol=1e-9)


def test_exact_sh_rotation_is_the_rotated_function():
    """f'(d) = f(R^T d) for every direction, all bands, every channel; band energy is preserved."""
    S = _splats(50, seed=2)
    R = transform.create_rotation_matrix("x", 33.0) @ transform.create_rotation_matrix("z", -71.0)
    T = transform.rotate_splats(S, R, "exact")
    g = torch.Generator().manual_seed(3)
    d = torch.randn(40, 3, generator=g, dtype=torch.float64)
    c0 = torch.cat([S["sh0"], S["shN"]], 1)
    c1 = torch.cat([T["sh0"], T["shN"]], 1)
    f_src = O.spherical_harmonics(3, (d @ R)[None].expand(50, -1, -1), c0[:, None].expand(-1, 40, -1, -1))
    f_rot = O.spherical_harmonics(3, d[None].expand(50, -1, -1), c1[:, None].expand(-1, 40, -1, -1))
    assert torch.allclose(f_rot, f_src, atol=1e-9)
    for lo, hi in ((0, 3), (3, 8), (8, 15)):
        assert torch.allclose(T["shN"][:, lo:hi].pow(2).sum(1), S["shN"][:, lo:hi].pow(2).sum(1), atol=1e-9)
    # the reference's approximation is NOT that function once bands 2-3 are non-zero
    Tr = transform.rotate_splats(S, R, "reference")
    f_ref = O.spherical_harmonics(3, d[None].expand(50, -1, -1), torch.cat([Tr["sh0"], Tr["shN"]], 1)[:, None].expand(-1, 40, -1, -1))
    assert float((f_ref - f_src).abs().max()) > 0.1


def test_exact_mirror_is_the_reflected_function():
    S = _splats(30, seed=4)
    T = transform.mirror_splats(S, "y", "exact")
    M = torch.diag(torch.tensor([1.0, -1.0, 1.0], dtype=torch.float64))
    d = torch.randn(25, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    f_src = O.spherical_harmonics(3, (d @ M)[None].expand(30, -1, -1), torch.cat([S["sh0"], S["shN"]], 1)[:, None].expand(-1, 25, -1, -1))
    f_mir = O.spherical_harmonics(3, d[None].expand(30, -1, -1), torch.cat([T["sh0"], T["shN"]], 1)[:, None].expand(-1, 25, -1, -1))
    assert torch.allclose(f_mir, f_src, atol=1e-9)


def test_four_quarter_turns_are_the_identity_and_files_roundtrip(tmp_path):
    src = str(tmp_path / "w.ply")
    S = io_ply.read_ply(WOLF)
    io_ply.write_ply(src, S, drop_nonfinite=False)
    for mode in ("exact", "reference"):
        out = str(tmp_path / f"r_{mode}.ply")
        transform.rotate_ply(src, out, "z:90,z:90,z:90,z:90", mode)
        T = io_ply.read_ply(out)
        assert torch.allclose(T["means"], S["means"], atol=1e-5) and torch.allclose(T["shN"], S["shN"], atol=1e-5)
        assert torch.equal(T["opacities"], S["opacities"]) and torch.equal(T["scales"], S["scales"])
    assert transform.main_mirror(["-i", src, "-o", str(tmp_path / "m.ply"), "--axis", "x"]) == 0
    Mx = io_ply.read_ply(str(tmp_path / "m.ply"))
    assert torch.allclose(Mx["means"][:, 0], -S["means"][:, 0]) and torch.equal(Mx["means"][:, 1:], S["means"][:, 1:])
    assert transform.main_rotate(["-i", src, "--rotations", "x:270,y:180,z:0"]) == 0       # in place, as main.py:1481 does
