"""Small host-side helpers of the renderer (mirror of the reference sdfrenderer/renderer/utils_rasterer.py:6-24,59-83)."""
import numpy as np
import torch


def qrot(q, v):
    """Rotate vector(s) v (*,3) by quaternion(s) q (*,4): v + 2 (q0 (qv x v) + qv x (qv x v))."""
    assert q.shape[-1] == 4 and v.shape[-1] == 3 and q.shape[:-1] == v.shape[:-1]
    shape = v.shape
    q = q.reshape(-1, 4)
    v = v.reshape(-1, 3)
    qv = q[:, 1:]
    uv = torch.cross(qv, v, dim=1)
    uuv = torch.cross(qv, uv, dim=1)
    return (v + 2 * (q[:, :1] * uv + uuv)).view(shape)


def qrot_matrix(q):
    """3x3 matrix M with qrot(q, v) == M v (linear in v for any q, unit or not); differentiable in q."""
    w, x, y, z = q[0], q[1], q[2], q[3]
    zero = torch.zeros_like(w)
    S = torch.stack([torch.stack([zero, -z, y]), torch.stack([z, zero, -x]), torch.stack([-y, x, zero])])
    eye = torch.eye(3, dtype=q.dtype, device=q.device)
    return eye + 2 * (w * S + S @ S)


def calibration_matrix(resolution_px, diagonal_mm, focal_len_mm, skew=0.):
    """Pinhole K from sensor diagonal and focal length (both mm); pixel aspect follows the resolution."""
    rx, ry = resolution_px
    diag_px = np.sqrt(rx ** 2 + ry ** 2)
    mx = rx / (rx / diag_px * diagonal_mm)
    my = ry / (ry / diag_px * diagonal_mm)
    return np.array([[focal_len_mm * mx, skew, rx / 2], [0, focal_len_mm * my, ry / 2], [0, 0, 1]])
