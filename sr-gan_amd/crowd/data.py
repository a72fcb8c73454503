"""Inference-side data handling of the crowd application: the sliding-window patches of one full
image (reference crowd/data.py:370-453,521-560) and the uint8 -> [-1, 1] normalisation (crowd/data.py:115-128).
Host-side NumPy only; the training datasets / preprocessors stay out of scope."""
import numpy as np
import torch


class CrowdExample:
    """Image (H, W, 3) and optional per-pixel label / map of one crowd scene (reference crowd/data.py:21-40)."""

    def __init__(self, image, label=None, roi=None, perspective=None, patch_center_y=None, patch_center_x=None, map_=None):
        self.image, self.label, self.roi, self.perspective, self.map = image, label, roi, perspective, map_
        self.patch_center_y, self.patch_center_x = patch_center_y, patch_center_x


def extract_padded_patch(image, y, x, patch_size):
    """The ``patch_size`` square centred on (y, x), zero-padded where it leaves the image, with the reference's
    order of operations (crowd/data.py:390-407: pad top / left first and move the centre, then pad bottom / right)."""
    half = int(patch_size // 2)
    if y - half < 0:
        image = np.pad(image, ((half - y, 0), (0, 0), (0, 0)), 'constant')
        y = half
    if y + half > image.shape[0]:
        image = np.pad(image, ((0, y + half - image.shape[0]), (0, 0), (0, 0)), 'constant')
    if x - half < 0:
        image = np.pad(image, ((0, 0), (half - x, 0), (0, 0)), 'constant')
        x = half
    if x + half > image.shape[1]:
        image = np.pad(image, ((0, 0), (0, x + half - image.shape[1]), (0, 0)), 'constant')
    return image[y - half:y + half, x - half:x + half, :]


def negative_one_to_one(image):
    """uint8 [0, 255] -> float32 [-1, 1] (crowd/data.py:127)."""
    return (image.astype(np.float32) / (255 / 2)) - 1


class ImageSlidingWindowDataset:
    """Every sliding-window patch of one full example: ``(image f32[3, P, P], x, y)`` per index (reference
    crowd/data.py:521-560).  Window centres run from half a patch in steps of ``window_step_size``; one more centre
    sits half a patch from the far edge, so an image smaller than a patch still yields one (padded) patch."""

    def __init__(self, full_example, image_patch_size=128, window_step_size=32):
        self.image = full_example.image
        self.window_step_size, self.image_patch_size = window_step_size, image_patch_size
        half = int(image_patch_size // 2)
        height, width = self.image.shape[0], self.image.shape[1]
        self.y_positions = list(range(half, height - half + 1, window_step_size))
        if height - half > 0:
            self.y_positions = sorted(set(self.y_positions + [height - half]))
        self.x_positions = list(range(half, width - half + 1, window_step_size))
        if width - half > 0:
            self.x_positions = sorted(set(self.x_positions + [width - half]))
        self.positions_shape = np.array([len(self.y_positions), len(self.x_positions)])
        self.length = int(self.positions_shape.prod())

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        y_index, x_index = np.unravel_index(index, self.positions_shape)
        y, x = self.y_positions[y_index], self.x_positions[x_index]
        patch = negative_one_to_one(extract_padded_patch(self.image, y, x, self.image_patch_size))
        return torch.from_numpy(np.ascontiguousarray(patch.transpose((2, 0, 1)))), x, y
