"""Synapse slice / volume reader and training augmentation.

Host-side mirror of the reference's datasets/dataset_synapse.py:12-83 with the same names, sample schema and RNG
consumption order (``random.random`` for the branch choice, ``np.random.randint`` for k / axis / angle), so a seeded
run produces the same augmented batch as the reference (tests/test_host_next_rows.py checks this against
tests/golden/g9_augment.npz, generated from the reference itself).  The arithmetic (rot90, flip, order-0 rotate,
cubic / nearest zoom) is numpy / scipy, exactly the libraries the reference calls; nothing here runs on the GPU.

Training slices: ``<base_dir>/<name>.npz`` with ``image`` (H, W) float32 in [0, 1] and ``label`` (or ``segmentation``)
(H, W) class ids (:62-69).  Test volumes: ``<base_dir>/<name>.npy.h5`` with ``image``/``label`` (or
``images``/``segmentations``) (:70-77); h5py is optional in this image, so ``<name>.npz`` volumes are accepted too.
"""
import os
import random

import numpy as np
import torch
from scipy import ndimage
from scipy.ndimage import zoom
from torch.utils.data import Dataset


def random_rot_flip(image, label):
    """rot90 by k in {0..3}, then flip along a random axis (dataset_synapse.py:12-19)."""
    k = np.random.randint(0, 4)
    image, label = np.rot90(image, k), np.rot90(label, k)
    axis = np.random.randint(0, 2)
    return np.flip(image, axis=axis).copy(), np.flip(label, axis=axis).copy()


def random_rotate(image, label):
    """rotate by an integer angle in [-20, 20), nearest neighbour for both, same shape (:22-26)."""
    angle = np.random.randint(-20, 20)
    return (ndimage.rotate(image, angle, order=0, reshape=False), ndimage.rotate(label, angle, order=0, reshape=False))


class RandomGenerator(object):
    """sample {'image': (H, W), 'label': (H, W)} -> {'image': float32 (1, h, w), 'label': int64 (h, w)} (:29-47)."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = sample['image'], sample['label']
        if random.random() > 0.5:
            image, label = random_rot_flip(image, label)
        elif random.random() > 0.5:
            image, label = random_rotate(image, label)
        x, y = image.shape
        if x != self.output_size[0] or y != self.output_size[1]:
            image = zoom(image, (self.output_size[0] / x, self.output_size[1] / y), order=3)
            label = zoom(label, (self.output_size[0] / x, self.output_size[1] / y), order=0)
        image = torch.from_numpy(image.astype(np.float32)).unsqueeze(0)
        label = torch.from_numpy(label.astype(np.float32))
        return {'image': image, 'label': label.long()}


class Synapse_dataset(Dataset):
    """split == 'train': 2-D slices from .npz; otherwise whole volumes (:50-83).  ``list_dir/<split>.txt`` names the cases."""

    def __init__(self, base_dir, list_dir, split, transform=None, is_kits=False, is_lits=False):
        self.transform = transform
        self.split = split
        with open(os.path.join(list_dir, self.split + '.txt')) as f:
            self.sample_list = f.readlines()
        self.data_dir = base_dir
        self.is_kits = is_kits

    def __len__(self):
        return len(self.sample_list)

    @staticmethod
    def _pick(data, *names):
        for n in names:
            if n in data:
                return data[n][:]
        raise KeyError(f"none of {names} in {list(data.keys())}")

    def __getitem__(self, idx):
        name = self.sample_list[idx].strip('\n')
        if self.split == "train":
            data = np.load(os.path.join(self.data_dir, name + '.npz'))
            image, label = data['image'], self._pick(data, 'label', 'segmentation')
        else:
            h5 = self.data_dir + "/{}.npy.h5".format(name)
            if os.path.exists(h5):
                try:
                    import h5py
                except ImportError as e:
                    raise RuntimeError(f"{h5}: reading test volumes in HDF5 needs h5py, which is not installed; "
                                       f"convert the volume to {name}.npz (image, label)") from e
                with h5py.File(h5, "r") as data:
                    image = self._pick(data, 'image', 'images')
                    label = self._pick(data, 'label', 'segmentations')
            else:
                data = np.load(os.path.join(self.data_dir, name + '.npz'))
                image, label = self._pick(data, 'image', 'images'), self._pick(data, 'label', 'segmentations')
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
