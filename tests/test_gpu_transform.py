"""Rotate / mirror with exact SH rotation: rendering the transformed splats from the transformed
camera reproduces the original image (what the reference's band-1-only approximation cannot)."""
import pytest
import torch

from helpers import small_scene
from mi3dgs import ops, transform

pytestmark = pytest.mark.gpu


def _render(P, V, K, w, h):
    img, alpha, _ = ops.rasterization(P["means"], P["quats"], P["scales"].exp(), torch.sigmoid(P["opacities"]),
                                      torch.cat([P["sh0"], P["shN"]], 1), V[None], K[None], w, h, sh_degree=3)
    return img[0], alpha[0]


def _moved_camera(V, M):
    """World x -> M x (M orthogonal): the camera that sees the moved world as before is V [M^T 0; 0 1]."""
    T = torch.eye(4, device=V.device)
    T[:3, :3] = M.to(V.device).float().T
    return V @ T


@pytest.mark.parametrize("spec", ["x:270,y:180,z:0", "z:37.5,x:12"])
def test_rotated_splats_render_the_same_from_the_rotated_camera(spec):
    sc = small_scene(n=3000, seed=11, width=160, height=128, n_views=2, fx=150.0).to(torch.device("cuda:0"))
    P, R = dict(sc.params), torch.eye(3, dtype=torch.float64)
    for axis, angle in transform.parse_rotation_spec(spec):
        Ri = transform.create_rotation_matrix(axis, angle)
        P, R = transform.rotate_splats(P, Ri, "exact"), Ri @ R
    for v in range(2):
        a, aa = _render(sc.params, sc.viewmats[v], sc.Ks[v], 160, 128)
        b, ba = _render(P, _moved_camera(sc.viewmats[v], R), sc.Ks[v], 160, 128)
        assert float((a - b).abs().max()) < 2e-3 and float((aa - ba).abs().max()) < 2e-3
    # the reference's SH handling does not have this property on a scene with view-dependent colour
    Q = dict(sc.params)
    for axis, angle in transform.parse_rotation_spec(spec):
        Q = transform.rotate_splats(Q, transform.create_rotation_matrix(axis, angle), "reference")
    c, ca = _render(Q, _moved_camera(sc.viewmats[0], R), sc.Ks[0], 160, 128)
    a, aa = _render(sc.params, sc.viewmats[0], sc.Ks[0], 160, 128)
    assert float((aa - ca).abs().max()) < 2e-3 and float((a - c).abs().max()) > 2e-2


@pytest.mark.parametrize("axis", ["x", "y", "z"])
def test_mirrored_splats_render_the_same_from_the_mirrored_camera(axis):
    sc = small_scene(n=3000, seed=12, width=160, height=128, n_views=1, fx=150.0).to(torch.device("cuda:0"))
    M = torch.eye(3, dtype=torch.float64)
    M["xyz".index(axis), "xyz".index(axis)] = -1
    P = transform.mirror_splats(dict(sc.params), axis, "exact")
    a, aa = _render(sc.params, sc.viewmats[0], sc.Ks[0], 160, 128)
    b, ba = _render(P, _moved_camera(sc.viewmats[0], M), sc.Ks[0], 160, 128)
    assert float((a - b).abs().max()) < 2e-3 and float((aa - ba).abs().max()) < 2e-3


# ---- the device path against the oracle: what `rotate_splat.py` / `mirror_splat.py` run on the job's GPU
def _f_rest(S):
    n = S["shN"].shape[0]
    return S["shN"].transpose(1, 2).reshape(n, 45).double().cpu().numpy()


def _same_rotation(qa, qb, tol):
    import numpy as np
    qa = qa / np.linalg.norm(qa, axis=1, keepdims=True)
    qb = qb / np.linalg.norm(qb, axis=1, keepdims=True)
    return bool(np.all(np.abs(np.abs((qa * qb).sum(1)) - 1) < tol))


@pytest.mark.parametrize("spec", ["x:270,y:180,z:0", "x:180,y:180", "z:37.5,x:12"])
def test_rotate_on_the_device_matches_the_reference_arithmetic(spec):
    """float32 tensors on cuda:0 through mi3dgs.transform (sh_mode='reference', the shim's default) against oracle/post_oracle.py,
    the numpy / scipy restatement of the reference's rotate_splat.py:90-178, in float64."""
    import numpy as np
    from oracle import post_oracle as PO
    dev = torch.device("cuda:0")
    sc = small_scene(n=5000, seed=21)
    S = {k: v.float() for k, v in sc.params.items()}
    pos, quat = S["means"].double().numpy(), S["quats"].double().numpy()
    dc, rest = S["sh0"][:, 0].double().numpy(), _f_rest(S)
    for axis, angle in PO.parse_rotation_spec(spec):
        pos, quat, dc, rest = PO.rotate_gaussians(pos, quat, dc, rest, PO.create_rotation_matrix(axis, angle))
    T = {k: v.to(dev) for k, v in S.items()}
    for axis, angle in transform.parse_rotation_spec(spec):
        T = transform.rotate_splats(T, transform.create_rotation_matrix(axis, angle), "reference")
    assert all(v.is_cuda and v.dtype == torch.float32 for v in T.values())
    assert np.allclose(T["means"].double().cpu().numpy(), pos, atol=2e-6)
    assert _same_rotation(T["quats"].double().cpu().numpy(), quat, 2e-6)
    assert np.allclose(_f_rest(T), rest, atol=2e-6) and torch.equal(T["sh0"].cpu(), S["sh0"])
    assert torch.equal(T["scales"].cpu(), S["scales"]) and torch.equal(T["opacities"].cpu(), S["opacities"])


@pytest.mark.parametrize("axis", ["x", "y", "z"])
def test_mirror_on_the_device_matches_the_reference_arithmetic(axis):
    import numpy as np
    from oracle import post_oracle as PO
    dev = torch.device("cuda:0")
    sc = small_scene(n=5000, seed=22)
    S = {k: v.float() for k, v in sc.params.items()}
    pos, quat, rest = PO.mirror(S["means"].double().numpy(), S["quats"].double().numpy(), _f_rest(S), axis)
    T = transform.mirror_splats({k: v.to(dev) for k, v in S.items()}, axis, "reference")
    assert all(v.is_cuda for v in T.values())
    assert np.allclose(T["means"].double().cpu().numpy(), pos, atol=1e-6) and _same_rotation(T["quats"].double().cpu().numpy(), quat, 2e-6)
    assert np.allclose(_f_rest(T), rest, atol=1e-6)
