"""Parameter algebra of the camera model at camera-count scale (a few dozen 3x3 matrices): mirrors
the helper names of /root/reference model/camera_utils.py that the model's public methods need
(`ortho2rotation` :78, `rotation2orth` :136, `get_44_rotation_matrix_from_33_rotation_matrix` :184,
`intrinsic_param_to_K` :191).  These serve get_intrinsic()/get_extrinsic() (logging, evaluation, the
PRD loss); the per-ray hot path does this arithmetic inside the HIP ray-generator kernel."""
import torch


def rotation2orth(rot: torch.Tensor) -> torch.Tensor:
    """[C,3,3] -> [C,6]: the first two *columns* of each rotation, concatenated."""
    return torch.cat([rot[:, :, 0], rot[:, :, 1]], dim=-1)


def _unit(v):
    mag = torch.sqrt((v ** 2).sum(1, keepdim=True))
    return v / (torch.clamp(mag, min=1e-8) + 1e-10)


def ortho2rotation(poses: torch.Tensor) -> torch.Tensor:
    """[C,6] -> [C,3,3] by Gram-Schmidt; columns x, y, x cross y."""
    a1, a2 = poses[:, 0:3], poses[:, 3:6]
    x = _unit(a1)
    f = (x * a2).sum(1, keepdim=True) / (torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8) + 1e-10)
    y = _unit(a2 - f * x)
    z = torch.stack([x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1], x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2],
                     x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]], dim=1)
    return torch.stack([x, y, z], dim=2)


def get_44_rotation_matrix_from_33_rotation_matrix(m: torch.Tensor) -> torch.Tensor:
    out = torch.zeros((m.shape[0], 4, 4), device=m.device, dtype=m.dtype)
    out[:, :3, :3] = m
    out[:, 3, 3] = 1
    return out


def intrinsic_param_to_K(intrinsics: torch.Tensor) -> torch.Tensor:
    K = torch.eye(4, 4, device=intrinsics.device, dtype=intrinsics.dtype)
    K[[0, 1, 0, 1], [0, 1, 2, 2]] = intrinsics
    return K
