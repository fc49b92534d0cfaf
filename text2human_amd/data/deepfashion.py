"""Readers for the DeepFashion-MultiModal tree used by the sampling entry points.

Same constructor arguments, item keys, dtypes and value conventions as the reference's
`data/segm_attr_dataset.py` (DeepFashionAttrSegmDataset) and `data/pose_attr_dataset.py`
(DeepFashionAttrPoseDataset), so `torch.utils.data.DataLoader(dataset, batch_size=4)`
feeds `model.inference` unchanged:

  <ann_dir>/{upper,lower,outer}_fused.txt   "<image name> <attribute id>" per line, the three
                                            files list the same images in the same order
  <shape_ann_path>                          "<image name> a0 a1 ... a14" per line (pose set)
  <img_dir>/<name>                          RGB image            -> Lanczos /factor,   x/127.5 - 1
  <segm_dir>/<stem>_segm.png                class-id map         -> nearest /factor,   float32
  <pose_dir>/<stem>_densepose.png           IUV, only channel 2.. kept -> nearest, x/12 - 1

Texture mask rule (segm_attr_dataset.py:140-151): 0 = common codebook, attribute + 1 on the
pixels of the garment classes (upper {1,4}, lower {3,5,21}, outer {2}); attribute 17 = none.
"""
import os

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

GARMENT_CLASSES = (('upper', (1.0, 4.0)), ('lower', (3.0, 5.0, 21.0)), ('outer', (2.0, )))
NO_ATTRIBUTE = 17


def read_fused_annotations(ann_dir):
    """-> (names, {'upper': [...], 'lower': [...], 'outer': [...]}); the lower / outer files
    must list the upper file's images in the same order (the reference asserts this too)."""
    names, attrs = None, {}
    for part, _ in GARMENT_CLASSES:
        path = os.path.join(ann_dir, f'{part}_fused.txt')
        if not os.path.exists(path):
            raise AssertionError(f'missing annotation file {path}')
        rows = [line.split() for line in open(path, 'r') if line.strip()]
        part_names = [r[0] for r in rows]
        if names is None:
            names = part_names
        elif part_names != names:
            raise AssertionError(f'{path} does not list the same images as upper_fused.txt')
        attrs[part] = [int(r[1]) for r in rows]
    return names, attrs


def _load(path, factor, resample):
    with open(path, 'rb') as f:
        im = Image.open(f)
        if factor != 1:
            w, h = im.size
            im = im.resize(size=(w // factor, h // factor), resample=resample)
        return np.array(im)


def texture_mask(segm, upper, lower, outer):
    """segm float tensor [1, H, W] of class ids -> texture mask of the same shape."""
    mask = torch.zeros_like(segm)
    for (_, classes), attr in zip(GARMENT_CLASSES, (upper, lower, outer)):
        if attr != NO_ATTRIBUTE:
            for c in classes:
                mask[segm == c] = attr + 1
    return mask


class _FusedAttrBase(data.Dataset):

    def __init__(self, ann_dir, downsample_factor, xflip):
        self._image_fnames, attrs = read_fused_annotations(ann_dir)
        self.upper_fused_attrs, self.lower_fused_attrs, self.outer_fused_attrs = (
            attrs['upper'], attrs['lower'], attrs['outer'])
        self.downsample_factor = downsample_factor
        self.xflip = xflip

    def __len__(self):
        return len(self._image_fnames)

    def _densepose(self, pose_dir, fname):
        arr = _load(os.path.join(pose_dir, f'{fname[:-4]}_densepose.png'), self.downsample_factor, Image.NEAREST)
        return arr[:, :, 2:].transpose(2, 0, 1).astype(np.float32)  # the I channel of IUV, [1, H, W]

    def _flip(self):
        import random
        return self.xflip and random.random() > 0.5


class DeepFashionAttrSegmDataset(_FusedAttrBase):
    """Items: image f32 [3,H,W] in [-1,1], densepose f32 [1,H,W], segm f32 [1,H,W],
    texture_mask f32 [1,H,W], img_name."""

    def __init__(self, img_dir, segm_dir, pose_dir, ann_dir, downsample_factor=2, xflip=False):
        super().__init__(ann_dir, downsample_factor, xflip)
        self._img_path, self._segm_path, self._densepose_path = img_dir, segm_dir, pose_dir

    def __getitem__(self, index):
        fname = self._image_fnames[index]
        image = _load(os.path.join(self._img_path, fname), self.downsample_factor, Image.LANCZOS)
        image = (image[:, :, None] if image.ndim == 2 else image).transpose(2, 0, 1)
        pose = self._densepose(self._densepose_path, fname)
        segm = _load(os.path.join(self._segm_path, f'{fname[:-4]}_segm.png'), self.downsample_factor, Image.NEAREST)
        segm = segm[None].astype(np.float32)
        if self._flip():
            image, pose, segm = image[:, :, ::-1].copy(), pose[:, :, ::-1].copy(), segm[:, :, ::-1].copy()
        image, segm = torch.from_numpy(image), torch.from_numpy(segm)
        return {
            'image': image / 127.5 - 1,
            'densepose': pose / 12. - 1,
            'segm': segm,
            'texture_mask': texture_mask(segm, self.upper_fused_attrs[index], self.lower_fused_attrs[index],
                                         self.outer_fused_attrs[index]),
            'img_name': fname,
        }


class DeepFashionAttrPoseDataset(_FusedAttrBase):
    """Items: densepose f32 [1,H,W], shape_attr i64 [15], the three fused texture attributes,
    img_name (the target image name; the densepose file is <stem>_densepose.png)."""

    def __init__(self, pose_dir, texture_ann_dir, shape_ann_path, downsample_factor=2, xflip=False):
        super().__init__(texture_ann_dir, downsample_factor, xflip)
        self._densepose_path = pose_dir
        self._image_fnames_target = self._image_fnames
        self._image_fnames = [f'{n.split(".")[0]}.png' for n in self._image_fnames_target]
        if not os.path.exists(shape_ann_path):
            raise AssertionError(f'missing annotation file {shape_ann_path}')
        rows = [line.split() for line in open(shape_ann_path, 'r') if line.strip()]
        if [r[0] for r in rows] != self._image_fnames_target:
            raise AssertionError(f'{shape_ann_path} does not list the same images as upper_fused.txt')
        self.shape_attrs = [[int(v) for v in r[1:]] for r in rows]

    def __getitem__(self, index):
        pose = self._densepose(self._densepose_path, self._image_fnames[index])
        if self._flip():
            pose = pose[:, :, ::-1].copy()
        return {
            'densepose': pose / 12. - 1,
            'img_name': self._image_fnames_target[index],
            'shape_attr': torch.LongTensor(self.shape_attrs[index]),
            'upper_fused_attr': self.upper_fused_attrs[index],
            'lower_fused_attr': self.lower_fused_attrs[index],
            'outer_fused_attr': self.outer_fused_attrs[index],
        }
