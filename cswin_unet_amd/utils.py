"""Loss module, segmentation metrics and the volume inference loop.

Mirrors the reference's utils.py names and call signatures:
  * ``DiceLoss(n_classes)(inputs, target, weight=None, softmax=False)``  (utils.py:9-45)  -- on the HIP path the softmax,
    one-hot, per-class sums and the gradient are one fused pass (csrc/loss.hip); there is no eager fallback.
  * ``calculate_metric_percase(pred, gt)``  (utils.py:48-58)  -- Dice and HD95.  The reference gets both from
    medpy 0.4.0 (``metric.binary.dc`` / ``hd95``), which is not installed here and not part of /root/reference: they are
    restated from medpy's published definitions.  PARITY UNPINNED for these two functions (no reference fixture holds
    their outputs); tests check them against hand-computed cases only.
  * ``test_single_volume(image, label, net, classes, patch_size, ...)``  (utils.py:61-102)  -- same per-slice arithmetic
    (cubic zoom in, argmax, nearest zoom out), but the slices of a volume go through the network in batches instead of
    one launch per slice.
"""
import numpy as np
import torch
import torch.nn as nn
from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure, zoom

from . import ops


class DiceLoss(nn.Module):
    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, weight=None, softmax=False):
        """sum_c weight[c] * (1 - (2 sum(p t) + s) / (sum(p p) + sum(t t) + s)) / n_classes, sums over the whole batch
        (utils.py:22-45).  ``softmax=True`` (how trainer.py:56 calls it): inputs are logits, the softmax is fused;
        ``softmax=False``: inputs are taken as probabilities as they are.  ``weight``: per-class factors (default all 1)."""
        if inputs.shape[1] != self.n_classes:
            raise AssertionError('predict {} & target shape do not match ({} classes)'.format(tuple(inputs.shape), self.n_classes))
        cw = None
        if weight is not None:
            if len(weight) != self.n_classes:
                raise AssertionError('weight has {} entries for {} classes'.format(len(weight), self.n_classes))
            cw = torch.as_tensor([float(w) for w in weight], dtype=torch.float32, device=inputs.device)
        loss, _ = ops.ce_dice_loss(inputs, target, w_ce=0.0, w_dice=1.0, inputs_are_probs=not softmax, class_weight=cw)
        return loss


def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
    """Distances from the border voxels of `result` to the border of `reference` (medpy.metric.binary.__surface_distances)."""
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    if not result.any():
        raise RuntimeError('The first supplied array does not contain any binary object.')
    if not reference.any():
        raise RuntimeError('The second supplied array does not contain any binary object.')
    footprint = generate_binary_structure(result.ndim, connectivity)
    result_border = result ^ binary_erosion(result, structure=footprint, iterations=1)
    reference_border = reference ^ binary_erosion(reference, structure=footprint, iterations=1)
    dt = distance_transform_edt(~reference_border, sampling=voxelspacing)
    return dt[result_border]


def dice_coefficient(result, reference):
    """2 |A n B| / (|A| + |B|), 0 when both are empty (medpy.metric.binary.dc)."""
    result, reference = np.atleast_1d(result.astype(bool)), np.atleast_1d(reference.astype(bool))
    inter = np.count_nonzero(result & reference)
    size = np.count_nonzero(result) + np.count_nonzero(reference)
    return 2.0 * inter / float(size) if size else 0.0


def hd95(result, reference, voxelspacing=None, connectivity=1):
    """95th percentile of the symmetric surface distances (medpy.metric.binary.hd95)."""
    a = _surface_distances(result, reference, voxelspacing, connectivity)
    b = _surface_distances(reference, result, voxelspacing, connectivity)
    return float(np.percentile(np.hstack((a, b)), 95))


def calculate_metric_percase(pred, gt):
    """(dice, hd95) of one class; (1, 0) when only the prediction is non-empty, (0, 0) otherwise (utils.py:48-58)."""
    pred, gt = np.asarray(pred).copy(), np.asarray(gt).copy()
    pred[pred > 0] = 1
    gt[gt > 0] = 1
    if pred.sum() > 0 and gt.sum() > 0:
        return dice_coefficient(pred, gt), hd95(pred, gt)
    elif pred.sum() > 0 and gt.sum() == 0:
        return 1, 0
    return 0, 0


@torch.no_grad()
def predict_volume(image, net, patch_size=(224, 224), batch_slices=16, device="cuda"):
    """image (D, H, W) or (H, W) numpy -> integer class map of the same shape.  Per slice: cubic zoom to patch_size if the
    size differs, network, argmax over classes (softmax is monotonic, utils.py:75), nearest zoom back (:70-81)."""
    image = np.asarray(image)
    single = image.ndim == 2
    vol = image[None] if single else image
    D, x, y = vol.shape
    resize = x != patch_size[0] or y != patch_size[1]
    net.eval()
    pred = np.zeros((D, x, y), np.int64)
    for d0 in range(0, D, batch_slices):
        sl = vol[d0:d0 + batch_slices]
        if resize:
            sl = np.stack([zoom(s, (patch_size[0] / x, patch_size[1] / y), order=3) for s in sl])
        inp = torch.from_numpy(np.ascontiguousarray(sl)).unsqueeze(1).float().to(device)
        out = torch.argmax(net(inp), dim=1).cpu().numpy()
        for i, o in enumerate(out):
            pred[d0 + i] = zoom(o, (x / patch_size[0], y / patch_size[1]), order=0) if resize else o
    return pred[0] if single else pred


def test_single_volume(image, label, net, classes, patch_size=[256, 256], test_save_path=None, case=None, z_spacing=1,
                       batch_slices=16, device="cuda"):
    """Per-class (dice, hd95) of one volume, classes 1..classes-1 (utils.py:61-102).  image / label: (1, D, H, W) tensors
    as the DataLoader yields them.  With test_save_path the volumes are written as .npz (SimpleITK, which the reference
    uses for .nii.gz, is not installed here)."""
    image, label = image.squeeze(0).cpu().detach().numpy(), label.squeeze(0).cpu().detach().numpy()
    prediction = predict_volume(image, net, tuple(patch_size), batch_slices, device).astype(label.dtype)
    metric_list = [calculate_metric_percase(prediction == i, label == i) for i in range(1, classes)]
    if test_save_path is not None:
        np.savez_compressed(f"{test_save_path}/{case}_pred.npz", image=image.astype(np.float32),
                            prediction=prediction.astype(np.float32), label=label.astype(np.float32),
                            spacing=np.asarray((1, 1, z_spacing), np.float32))
    return metric_list


test_single_volume.__test__ = False     # not a pytest test (the name is the reference's)
