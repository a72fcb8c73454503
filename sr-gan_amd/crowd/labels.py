"""Offline crowd labels from the annotated head positions (surface of reference crowd/database_preprocessor.py:64-101,
253-290): the point map (one unit of density at each head's pixel) on the host -- a few thousand
scalar additions -- and the ikNN maps ``1 / (mean distance to the k nearest heads + epsilon)`` on the device
(`srgan_crowd_iknn_map`), where the reference runs a scikit-learn ball tree over every pixel position."""
import numpy as np
import torch

from .. import _lib
from ..utility import current_device


def generate_point_density_map(head_positions, label_size):
    """(map, number of heads outside the map); positions are (y, x), rounded half-to-even like the reference's
    ``int(round(.))``; negative indices wrap as in the reference's NumPy indexing."""
    density_map = np.zeros(label_size)
    out_of_bounds_count = 0
    for y, x in head_positions:
        try:
            density_map[int(round(y)), int(round(x))] += 1
        except IndexError:
            out_of_bounds_count += 1
    return density_map, out_of_bounds_count


def generate_iknn_map(head_positions, label_size, number_of_neighbors=1, epsilon=1.0, upper_bound=None, device=None):
    """``1 / (generate_knn_map(head_positions, label_size, k, upper_bound) + epsilon)`` of the reference as a float32
    device tensor [H, W] (the preprocessor stores it as float16)."""
    device = device or current_device()
    heads = torch.as_tensor(np.ascontiguousarray(head_positions, dtype=np.float32)).to(device)
    if heads.ndim != 2 or heads.shape[1] != 2 or heads.shape[0] == 0:
        raise ValueError('head_positions must be a non-empty (M, 2) array of (y, x) pairs')
    height, width = int(label_size[0]), int(label_size[1])
    out = torch.empty((height, width), dtype=torch.float32, device=device)
    _lib.check(_lib.library().srgan_crowd_iknn_map(heads.data_ptr(), heads.shape[0], height, width,
                                                   int(number_of_neighbors), float(epsilon),
                                                   float(upper_bound) if upper_bound is not None else 0.0, out.data_ptr(),
                                                   torch.cuda.current_stream(device).cuda_stream), 'srgan_crowd_iknn_map')
    return out


def generate_density_label(head_positions, label_size, neighbor_deviation_beta=0.15, device=None):
    """The Gaussian density label of the reference's preprocessor for databases without a perspective map
    (``generate_density_label(positions, size, perspective_resizing=True, yx_order=True, neighbor_deviation_beta=beta)``,
    crowd/database_preprocessor.py:82-91,110-236): every head a Gaussian of sigma = beta x its mean distance to its 11
    nearest heads (itself included), windowed at int(2 sigma), normalised, and the whole label rescaled to the head
    count.  float32 device tensor [H, W]."""
    device = device or current_device()
    heads = torch.as_tensor(np.ascontiguousarray(head_positions, dtype=np.float32)).to(device)
    if heads.ndim != 2 or heads.shape[1] != 2 or heads.shape[0] < 2:
        raise ValueError('head_positions must be an (M >= 2, 2) array of (y, x) pairs')
    height, width = int(label_size[0]), int(label_size[1])
    out = torch.empty((height, width), dtype=torch.float32, device=device)
    workspace = torch.empty((heads.shape[0], 5), dtype=torch.float32, device=device)
    _lib.check(_lib.library().srgan_crowd_density_label(heads.data_ptr(), heads.shape[0], height, width,
                                                        float(neighbor_deviation_beta), workspace.data_ptr(), out.data_ptr(),
                                                        torch.cuda.current_stream(device).cuda_stream),
               'srgan_crowd_density_label')
    counted = (workspace[:, 4] > 0).sum()             # heads whose window reaches the image
    return out * (counted / out.sum())
