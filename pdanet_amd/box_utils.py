"""Box geometry the IA-SSD head needs (pcdet/utils/box_utils.py:28-53,145-158 and
pcdet/utils/common_utils.py rotate_points_along_z), torch only."""
import torch


def rotate_points_along_z(points, angle):
    """common_utils.rotate_points_along_z: points (B, N, 3+C), angle (B) -> rotated about z (x towards y)."""
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.matmul(points[:, :, 0:3], rot)
    return torch.cat((out, points[:, :, 3:]), dim=-1)


_CONST = {}


def const_tensor(like, key, values):
    """Small constant on `like`'s device/dtype, uploaded once (a per-call new_tensor(list) on the GPU is
    a pageable host-to-device copy, i.e. a host synchronisation in the middle of the step)."""
    k = (key, like.device, like.dtype)
    if k not in _CONST:
        _CONST[k] = torch.tensor(values, device=like.device, dtype=like.dtype)
    return _CONST[k]


_CORNER_TEMPLATE = ((1, 1, -1), (1, -1, -1), (-1, -1, -1), (-1, 1, -1), (1, 1, 1), (1, -1, 1), (-1, -1, 1), (-1, 1, 1))


def boxes_to_corners_3d(boxes3d):
    """box_utils.py:28-53: (N, 7) [x, y, z, dx, dy, dz, heading] -> (N, 8, 3) corners."""
    template = const_tensor(boxes3d, 'corners', _CORNER_TEMPLATE) / 2
    corners = boxes3d[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    corners = rotate_points_along_z(corners.view(-1, 8, 3), boxes3d[:, 6]).view(-1, 8, 3)
    return corners + boxes3d[:, None, 0:3]


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """box_utils.py:145-158."""
    large = boxes3d.clone()
    for k, w in enumerate(extra_width):      # python scalars travel as kernel arguments
        large[:, 3 + k] += float(w)
    return large
