"""Host-side sliding-window bookkeeping of the crowd inference path (sr-gan_amd/crowd/data.py; reference
crowd/data.py:370-453,521-560).  The end-to-end result is pinned on the GPU against golden g9."""
import numpy as np


def test_sliding_window_positions_and_padding():
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.data import CrowdExample, ImageSlidingWindowDataset, extract_padded_patch, negative_one_to_one
    image = np.arange(100 * 150 * 3, dtype=np.int64).reshape(100, 150, 3) % 256
    image = image.astype(np.uint8)
    dataset = ImageSlidingWindowDataset(CrowdExample(image=image), image_patch_size=64, window_step_size=24)
    assert dataset.y_positions == [32, 56, 68] and dataset.x_positions == [32, 56, 80, 104, 118]
    assert len(dataset) == 15
    patch, x, y = dataset[7]                     # row 1, column 2
    assert (x, y) == (80, 56) and tuple(patch.shape) == (3, 64, 64)
    expected = negative_one_to_one(image[56 - 32:56 + 32, 80 - 32:80 + 32]).transpose(2, 0, 1)
    np.testing.assert_array_equal(patch.numpy(), expected)
    # a centre closer than half a patch to the top-left corner: zero padding, centre moved
    small = np.full((40, 90, 3), 200, dtype=np.uint8)
    padded = extract_padded_patch(small, 8, 32, 64)
    assert padded.shape == (64, 64, 3)
    assert (padded[:24] == 0).all() and (padded[24:] == 200).all()        # 32 - 8 rows of padding on top
    assert float(negative_one_to_one(np.array([[[0, 255, 127]]], dtype=np.uint8)).min()) == -1.0


def test_preprocessed_database_reader(tmp_path):
    """The on-disk layout of the reference's preprocessed crowd databases (crowd/shanghai_tech_data.py:18-44):
    <database>/<part>/<dataset>_data/{images,labels,<maps>}/<name>.npy -> (image, label, map) / CrowdExamples."""
    import os
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.data import PreprocessedCrowdDataset
    generator = np.random.RandomState(3)
    root = tmp_path / 'ShanghaiTech' / 'part_B' / 'test_data'
    scenes = {}
    for name, shape in (('IMG_2.npy', (30, 40)), ('IMG_1.npy', (25, 35)), ('IMG_3.npy', (20, 20))):
        scenes[name] = (generator.randint(0, 256, size=shape + (3,)).astype(np.uint8),
                        generator.rand(*shape).astype(np.float32), generator.rand(*shape).astype(np.float32))
        for directory, array in zip(('images', 'labels', 'density3e-1'), scenes[name]):
            os.makedirs(root / directory, exist_ok=True)
            np.save(root / directory / name, array)
    (root / 'labels' / 'notes.txt').write_text('ignored')
    dataset = PreprocessedCrowdDataset(str(tmp_path / 'ShanghaiTech'), dataset='test', part='part_B',
                                       map_directory_name='density3e-1')
    assert len(dataset) == dataset.length == 3 and dataset.file_names == ['IMG_1.npy', 'IMG_2.npy', 'IMG_3.npy']
    image, label, map_ = dataset[1]
    for got, expected in zip((image, label, map_), scenes['IMG_2.npy']):
        np.testing.assert_array_equal(got, expected)
    examples = dataset.examples()
    assert [e.image.shape[:2] for e in examples] == [(25, 35), (30, 40), (20, 20)]
    np.testing.assert_array_equal(examples[2].map, scenes['IMG_3.npy'][2])
    assert len(PreprocessedCrowdDataset(str(tmp_path / 'ShanghaiTech'), dataset='test', part='part_B',
                                        number_of_examples=2, map_directory_name='density3e-1')) == 2


def test_point_density_map_matches_the_reference():
    """generate_point_density_map against the reference's output (golden g12): host-side, runs without a GPU."""
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.labels import generate_point_density_map
    from helpers import load_golden
    g = load_golden('g12_crowd_labels')
    for index in range(3):
        density, outside = generate_point_density_map(g[f'scene{index}/heads_yx'], tuple(g[f'scene{index}/shape']))
        assert outside == 0
        np.testing.assert_array_equal(density, g[f'scene{index}/point_map'])
    _, outside = generate_point_density_map(np.array([[3.0, 99.0], [1.0, 1.0]]), (10, 10))
    assert outside == 1


def test_host_patch_transforms_against_the_reference_fixture():
    """``extract_padded_patch`` / ``negative_one_to_one`` + a left-right flip against golden g13, which the reference's own
    transforms produced (ExtractPatchForPosition(allow_padded=True), RandomHorizontalFlip, NegativeOneToOneNormalizeImage,
    NumpyArraysToTorchTensors; tests/golden/make_goldens.py g13): scenes larger / equal / smaller than a patch, centres
    inside, on the borders and in the corners."""
    import srgan_amd  # noqa: F401
    from srgan_amd.crowd.data import extract_padded_patch, negative_one_to_one
    from helpers import load_golden
    g = load_golden('g13_crowd_patches')
    size = int(g['patch_size'])
    for number, (scene, y, x, flip) in enumerate(g['draws']):
        image, label, map_ = (g[f'scene{scene}/{k}'] for k in ('image', 'label', 'map'))
        patch = negative_one_to_one(extract_padded_patch(image, int(y), int(x), size))
        patch_label = extract_padded_patch(label[:, :, None], int(y), int(x), size)[:, :, 0]
        patch_map = extract_padded_patch(map_[:, :, None], int(y), int(x), size)[:, :, 0]
        if flip:
            patch, patch_label, patch_map = (np.flip(a, axis=1) for a in (patch, patch_label, patch_map))
        np.testing.assert_array_equal(patch.transpose(2, 0, 1), g[f'patch{number}/image'])
        np.testing.assert_array_equal(patch_label, g[f'patch{number}/label'])
        np.testing.assert_array_equal(patch_map, g[f'patch{number}/map'])
