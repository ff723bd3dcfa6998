"""Synapse slice / volume reader and training augmentation.

Host-side counterpart of the reference's datasets/dataset_synapse.py:12-83: same public names, sample schema and -- this is
what makes a seeded run reproduce the reference's batches -- the same consumption order of the two RNG streams
(`random.random()` picks the branch, `np.random.randint` draws k / axis / angle).  tests/test_host_next_rows.py checks the
outputs against tests/golden/g9_augment.npz, which was produced by the reference itself.  The arithmetic (rot90, flip,
order-0 rotate, cubic / nearest zoom) is numpy / scipy, exactly the libraries the reference calls; nothing here touches the GPU.

Training slices: `<base_dir>/<name>.npz` holding `image` (H, W) float32 in [0, 1] and `label` (or `segmentation`) (H, W)
class ids (:62-69).  Test volumes: `<base_dir>/<name>.npy.h5` with `image`/`label` or `images`/`segmentations` (:70-77);
h5py is optional in this image, so `<name>.npz` volumes are accepted as well.
"""
import os
import random

import numpy as np
import torch
from scipy import ndimage
from torch.utils.data import Dataset

_NEAREST, _CUBIC = 0, 3


def random_rot_flip(image, label):
    """Quarter turns (k drawn from {0..3}) followed by a flip along a drawn axis; the same transform for both arrays (:12-19)."""
    quarter_turns = np.random.randint(0, 4)
    flip_axis = None
    out = []
    for arr in (image, label):
        arr = np.rot90(arr, quarter_turns)
        if flip_axis is None:
            flip_axis = np.random.randint(0, 2)         # drawn after the rotations, like the reference
        out.append(arr)
    return tuple(np.flip(arr, axis=flip_axis).copy() for arr in out)


def random_rotate(image, label):
    """Rotation by an integer angle from [-20, 20), nearest neighbour for image and label alike, shape kept (:22-26)."""
    degrees = np.random.randint(-20, 20)
    turn = lambda arr: ndimage.rotate(arr, degrees, order=_NEAREST, reshape=False)
    return turn(image), turn(label)


def _resize_pair(image, label, size):
    """Cubic zoom of the image and nearest zoom of the label to `size` when the shape differs (:41-43)."""
    h, w = image.shape
    if (h, w) == tuple(size):
        return image, label
    factors = (size[0] / h, size[1] / w)
    return ndimage.zoom(image, factors, order=_CUBIC), ndimage.zoom(label, factors, order=_NEAREST)


class RandomGenerator(object):
    """{'image': (H, W), 'label': (H, W)} -> {'image': float32 (1, h, w), 'label': int64 (h, w)} (:29-47)."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        pair = (sample['image'], sample['label'])
        # one uniform draw decides "rot90 + flip"; only if that fails a second draw decides "small rotation"
        if random.random() > 0.5:
            pair = random_rot_flip(*pair)
        elif random.random() > 0.5:
            pair = random_rotate(*pair)
        image, label = _resize_pair(*pair, self.output_size)
        image_t = torch.from_numpy(image.astype(np.float32)).unsqueeze(0)
        label_t = torch.from_numpy(label.astype(np.float32)).long()
        return {'image': image_t, 'label': label_t}


def _first_key(data, *names):
    for n in names:
        if n in data:
            return data[n][:]
    raise KeyError(f"none of {names} in {list(data.keys())}")


class Synapse_dataset(Dataset):
    """split == 'train': 2-D slices from .npz; any other split: whole volumes (:50-83).  `list_dir/<split>.txt` names the cases."""

    def __init__(self, base_dir, list_dir, split, transform=None, is_kits=False, is_lits=False):
        self.data_dir, self.split, self.transform, self.is_kits = base_dir, split, transform, is_kits
        with open(os.path.join(list_dir, split + '.txt')) as f:
            self.sample_list = f.readlines()

    def __len__(self):
        return len(self.sample_list)

    def _read_slice(self, name):
        data = np.load(os.path.join(self.data_dir, name + '.npz'))
        return data['image'], _first_key(data, 'label', 'segmentation')

    def _read_volume(self, name):
        h5 = self.data_dir + "/{}.npy.h5".format(name)
        if not os.path.exists(h5):
            data = np.load(os.path.join(self.data_dir, name + '.npz'))
            return _first_key(data, 'image', 'images'), _first_key(data, 'label', 'segmentations')
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError(f"{h5}: reading test volumes in HDF5 needs h5py, which is not installed; "
                               f"convert the volume to {name}.npz (image, label)") from e
        with h5py.File(h5, "r") as data:
            return _first_key(data, 'image', 'images'), _first_key(data, 'label', 'segmentations')

    def __getitem__(self, idx):
        name = self.sample_list[idx].strip('\n')
        image, label = (self._read_slice if self.split == "train" else self._read_volume)(name)
        sample = {'image': image, 'label': label}
        if self.transform:
            sample = self.transform(sample)
        sample['case_name'] = name
        return sample


def write_synthetic_synapse(root, n_slices=8, n_volumes=1, size=512, depth=6, num_classes=9, seed=1234):
    """A stand-in dataset in the Synapse schema (the real data is not in the container): blocky
    label maps and images correlated with them.  Returns (base_dir_train, base_dir_test, list_dir)."""
    rng = np.random.default_rng(seed)
    train, test, lists = (os.path.join(root, d) for d in ("train_npz", "test_vol", "lists"))
    for d in (train, test, lists):
        os.makedirs(d, exist_ok=True)

    def one(shape):
        coarse = rng.integers(0, num_classes, size=tuple(s // 32 for s in shape))
        lab = np.kron(coarse, np.ones((32, 32), np.int64)).astype(np.float32)
        img = (lab / (num_classes - 1) * 0.6 + 0.2 + 0.05 * rng.standard_normal(shape)).clip(0, 1).astype(np.float32)
        return img, lab

    names = []
    for i in range(n_slices):
        img, lab = one((size, size))
        names.append(f"case{i // 4:04d}_slice{i % 4:03d}")
        np.savez(os.path.join(train, names[-1] + ".npz"), image=img, label=lab)
    with open(os.path.join(lists, "train.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    vols = []
    for v in range(n_volumes):
        pairs = [one((size, size)) for _ in range(depth)]
        vols.append(f"case{100 + v:04d}")
        np.savez(os.path.join(test, vols[-1] + ".npz"), image=np.stack([p[0] for p in pairs]),
                 label=np.stack([p[1] for p in pairs]))
    with open(os.path.join(lists, "test_vol.txt"), "w") as f:
        f.write("\n".join(vols) + "\n")
    return train, test, lists
