"""Small pure-tensor helpers of the camera model (host side, no kernels).

Mirrors the helper names of /root/reference model/camera_utils.py that the hot
path uses (`rotation2orth` :136, `intrinsic_param_to_K` :191,
`get_44_rotation_matrix_from_33_rotation_matrix` :184, `ortho2rotation` :78)."""
import torch


def rotation2orth(rot: torch.Tensor) -> torch.Tensor:
    """[C,3,3] -> [C,6]: the first two *columns* of each rotation, concatenated."""
    return torch.cat([rot[:, :, 0], rot[:, :, 1]], dim=-1)
