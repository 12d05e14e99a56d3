"""Parameter algebra of the camera model at camera-count scale (a few dozen 3x3 matrices): mirrors
the helper names of /root/reference model/camera_utils.py that the model's public methods need
(`ortho2rotation` :78, `rotation2orth` :136, `get_44_rotation_matrix_from_33_rotation_matrix` :184,
`intrinsic_param_to_K` :191).  These serve get_intrinsic()/get_extrinsic() (logging, evaluation, the
PRD loss); the per-ray hot path does this arithmetic inside the HIP ray-generator kernel.

The remaining public names of the reference module (`make_rand_axis` :11, `R_axis_angle` :17, `to_pil` :60,
`to_pil_normalize` :68, `rot_from_angle` :140, `angle_from_rot` :177) are host-side helpers that
`run_nerf.py` and the data loaders reach through `from model.camera_model import *`; they are provided so
the star-import surface of the mirror equals the reference's."""
import numpy as np
import torch
from PIL import Image

__all__ = ["make_rand_axis", "R_axis_angle", "to_pil", "to_pil_normalize", "ortho2rotation", "rotation2orth",
           "rot_from_angle", "angle_from_rot", "get_44_rotation_matrix_from_33_rotation_matrix",
           "intrinsic_param_to_K", "np", "torch", "Image"]


def make_rand_axis(batch_size):
    """[B,3] random unit vectors (numpy's global generator, as the reference's noise injection uses it)."""
    v = np.random.rand(batch_size, 3) - 0.5
    return v / np.linalg.norm(v, 2, 1, keepdims=True)


def R_axis_angle(axis, angle):
    """Rodrigues rotation matrices [B,3,3] (float64 numpy) of `angle` [B,1] radians about unit `axis` [B,3]:
    R = cos a * I + sin a * [axis]_x + (1 - cos a) * axis axis^T."""
    axis = np.asarray(axis, dtype=np.float64)
    ca, sa = np.cos(angle)[:, :, None], np.sin(angle)[:, :, None]
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    zero = np.zeros_like(x)
    cross = np.stack([np.stack([zero, -z, y], -1), np.stack([z, zero, -x], -1), np.stack([-y, x, zero], -1)], 1)
    outer = axis[:, :, None] * axis[:, None, :]
    return ca * np.eye(3)[None] + sa * cross + (1.0 - ca) * outer


def _as_image_array(array):
    if isinstance(array, torch.Tensor):
        array = array.detach().cpu()
        if array.dim() > 3 and array.shape[2] != 3:
            array = array.permute(1, 2, 0)
        array = array.numpy()
    return array


def to_pil(array):
    """[H,W(,3)] values in [0,1] -> 8-bit PIL image."""
    return Image.fromarray(np.uint8(_as_image_array(array) * 255))


def to_pil_normalize(array):
    """As to_pil after an affine stretch of the value range to [0,1] (a constant image divides 0/0 as in
    the reference: NaN -> 0 after the uint8 cast, with numpy's warning)."""
    a = _as_image_array(array)
    lo, hi = a.min(), a.max()
    with np.errstate(invalid="ignore", divide="ignore"):
        return Image.fromarray(np.uint8((a - lo) / (hi - lo) * 255))


def rot_from_angle(euler: torch.Tensor) -> torch.Tensor:
    """[B,3] angles -> [B,3,3] = RZ RY RX where each factor is the *transpose* of the textbook elementary
    rotation (the reference stacks rows along the last axis, :146-175)."""
    c, s = torch.cos(euler), torch.sin(euler)
    o, l = torch.zeros_like(c[:, 0]), torch.ones_like(c[:, 0])

    def mat(rows):
        return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)
    rx = mat([[l, o, o], [o, c[:, 0], s[:, 0]], [o, -s[:, 0], c[:, 0]]])
    ry = mat([[c[:, 1], o, -s[:, 1]], [o, l, o], [s[:, 1], o, c[:, 1]]])
    rz = mat([[c[:, 2], s[:, 2], o], [-s[:, 2], c[:, 2], o], [o, o, l]])
    return rz @ ry @ rx


def angle_from_rot(R: torch.Tensor) -> torch.Tensor:
    """Inverse of rot_from_angle away from gimbal lock."""
    x = -torch.atan2(R[:, 2, 1], R[:, 2, 2])
    y = -torch.atan2(-R[:, 2, 0], torch.sqrt(R[:, 2, 1] ** 2 + R[:, 2, 2] ** 2))
    z = -torch.atan2(R[:, 1, 0], R[:, 0, 0])
    return torch.stack([x, y, z], dim=1)


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
