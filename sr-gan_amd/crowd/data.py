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


class PreprocessedCrowdDataset:
    """Full-image examples of a crowd database in the on-disk layout the reference's preprocessors write and its
    ``ShanghaiTechFullImageDataset`` / ``UcfQnrfFullImageDataset`` read (crowd/shanghai_tech_data.py:18-44,
    crowd/ucf_qnrf_data.py): ``<database_directory>/[<part>/]<dataset>_data/{images,labels,<map_directory_name>}/*.npy``
    (image u8[H, W, 3], head-count label f32[H, W], ikNN map f32[H, W]).  ``__getitem__`` returns ``(image, label,
    map)`` like the reference; ``examples()`` gives the ``CrowdExample``s that ``DeviceCrowdPatchLoader`` keeps resident
    (training) and ``CrowdExperiment.test_summaries`` walks (evaluation).  The database directory is passed in (the
    reference takes it from its preprocessor object)."""

    def __init__(self, database_directory, dataset='train', part=None, number_of_examples=None,
                 map_directory_name='knn_maps'):
        import os
        pieces = [database_directory] + ([part] if part else []) + ['{}_data'.format(dataset)]
        self.dataset_directory = os.path.join(*pieces)
        names = [name for name in os.listdir(os.path.join(self.dataset_directory, 'labels')) if name.endswith('.npy')]
        self.file_names = sorted(names)[:number_of_examples]       # (the reference keeps the directory's own order)
        self.length = len(self.file_names)
        self.map_directory_name = map_directory_name

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        import os
        file_name = self.file_names[index]
        return tuple(np.load(os.path.join(self.dataset_directory, directory, file_name))
                     for directory in ('images', 'labels', self.map_directory_name))

    def examples(self):
        return [CrowdExample(image=image, label=label, map_=map_) for image, label, map_ in
                (self[index] for index in range(self.length))]


class DeviceCrowdPatchLoader:
    """Endless training batches cut ON THE DEVICE from full crowd scenes that stay resident in HBM:
    the reference's ``ShanghaiTechTransformedDataset`` + ``DataLoader(num_workers=4)`` pipeline (random position
    uniform over every valid patch centre of every scene, ``RandomHorizontalFlip``, [-1, 1] normalisation, CHW layout;
    crowd/shanghai_tech_data.py:47-107) with one kernel launch per batch instead of per-example NumPy work.

    ``examples``: ``CrowdExample``s with uint8 ``image`` (H, W, 3) and float ``label`` / ``map`` (H, W).  Yields
    ``(image f32[B, 3, P, P], label f32[B, P, P], map f32[B, P, P])`` device tensors -- the batch contract of
    ``CrowdExperiment``.  Positions and flips come from a private ``numpy`` generator (the reference's
    multi-process workers make its own stream irreproducible anyway)."""

    def __init__(self, examples, batch_size, image_patch_size=224, seed=0, device=None, flip=True, dp=None):
        """``batch_size`` is the GLOBAL batch; under data parallelism (``dp``) every rank draws the positions of the whole
        batch from the shared seed and cuts only its own contiguous slice (same examples as one device would see)."""
        from .. import _lib
        self.dp = dp if dp is not None and dp.world_size > 1 else None
        if self.dp is not None:
            self.dp.local_batch(batch_size)           # raises unless the global batch divides over the ranks
        from ..utility import current_device
        self._lib = _lib
        self.device = device or current_device()
        self.batch_size, self.patch_size, self.flip = batch_size, image_patch_size, flip
        self.generator = np.random.RandomState(seed)
        half = image_patch_size // 2
        self.images, self.labels, self.maps, self.shapes, self.start_indexes = [], [], [], [], []
        self.length = 0
        for example in examples:
            height, width = example.image.shape[0], example.image.shape[1]
            self.images.append(torch.from_numpy(np.ascontiguousarray(example.image, dtype=np.uint8)).to(self.device))
            self.labels.append(torch.from_numpy(np.ascontiguousarray(example.label, dtype=np.float32)).to(self.device))
            self.maps.append(torch.from_numpy(np.ascontiguousarray(example.map, dtype=np.float32)).to(self.device))
            self.shapes.append((height, width))
            self.start_indexes.append(self.length)
            # every centre whose patch lies inside the scene; a scene smaller than a patch has none, as upstream
            self.length += max(height - 2 * half + 1, 0) * max(width - 2 * half + 1, 0)
        if self.length == 0:
            raise ValueError('no scene is as large as one patch')

    def draw_positions(self):
        """(scene index, y, x, flip) of one batch: a uniform draw over all valid centres (reference
        crowd/shanghai_tech_data.py:83-98) and a fair coin per example (crowd/data.py:104)."""
        half = self.patch_size // 2
        draws = []
        for _ in range(self.batch_size):
            index = int(self.generator.randint(self.length))
            scene = int(np.searchsorted(self.start_indexes, index, side='right') - 1)
            height, width = self.shapes[scene]
            columns = width - 2 * half + 1
            y_index, x_index = divmod(index - self.start_indexes[scene], columns)
            draws.append((scene, half + y_index, half + x_index, int(self.flip and self.generator.randint(2))))
        if self.dp is not None:
            local = self.batch_size // self.dp.world_size
            draws = draws[self.dp.rank * local:(self.dp.rank + 1) * local]
        return draws

    def batch_for(self, draws):
        """The device batch of explicit ``(scene, y, x, flip)`` draws."""
        count, size = len(draws), self.patch_size
        pointers = lambda tensors: torch.tensor([tensors[d[0]].data_ptr() for d in draws], dtype=torch.int64)
        table = torch.stack([pointers(self.images), pointers(self.labels), pointers(self.maps)]).to(self.device)
        numbers = torch.tensor([[self.shapes[d[0]][0] for d in draws], [self.shapes[d[0]][1] for d in draws],
                                [d[1] for d in draws], [d[2] for d in draws], [d[3] for d in draws]],
                               dtype=torch.int32).to(self.device)
        image = torch.empty((count, 3, size, size), dtype=torch.float32, device=self.device)
        label = torch.empty((count, size, size), dtype=torch.float32, device=self.device)
        map_ = torch.empty((count, size, size), dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream().cuda_stream
        self._lib.check(self._lib.library().srgan_crowd_extract_patches(
            table[0].data_ptr(), table[1].data_ptr(), table[2].data_ptr(), numbers[0].data_ptr(), numbers[1].data_ptr(),
            numbers[2].data_ptr(), numbers[3].data_ptr(), numbers[4].data_ptr(), count, size, image.data_ptr(),
            label.data_ptr(), map_.data_ptr(), stream), 'srgan_crowd_extract_patches')
        self._keep_alive = (table, numbers)          # the kernel reads the tables asynchronously
        return image, label, map_

    def __iter__(self):
        while True:
            yield self.batch_for(self.draw_positions())
