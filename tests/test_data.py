"""Dataset readers on a tiny synthetic DeepFashion-style tree: item
contract, and -- where /root/reference exists -- equality with the reference's own loaders."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from text2human_amd import synthetic
from text2human_amd.data import DeepFashionAttrPoseDataset, DeepFashionAttrSegmDataset, deepfashion

REF = '/root/reference/data'


@pytest.fixture(scope='module')
def tree(tmp_path_factory):
    return synthetic.write_dataset_tree(str(tmp_path_factory.mktemp('df')), n=3, seed=7)


def test_segm_dataset_item_contract(tree):
    ds = DeepFashionAttrSegmDataset(img_dir=tree['test_img_dir'], segm_dir=tree['segm_dir'],
                                    pose_dir=tree['pose_dir'], ann_dir=tree['test_ann_file'])
    assert len(ds) == 3
    it = ds[1]
    assert it['img_name'] == tree['names'][1]
    assert it['image'].shape == (3, 512, 256) and it['image'].dtype == torch.float32
    assert -1.0 <= it['image'].min() and it['image'].max() <= 1.0
    assert it['segm'].shape == (1, 512, 256) and it['segm'].dtype == torch.float32
    assert it['densepose'].shape == (1, 512, 256) and it['densepose'].dtype == np.float32
    assert it['densepose'].min() >= -1.0 and it['densepose'].max() <= 1.0
    ref = synthetic.texture_mask_from_segm(it['segm'][None], [ds.upper_fused_attrs[1]], [ds.lower_fused_attrs[1]],
                                           [ds.outer_fused_attrs[1]])[0]
    assert torch.equal(it['texture_mask'], ref)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)))
    assert batch['segm'].shape == (2, 1, 512, 256) and list(batch['img_name']) == tree['names'][:2]


def test_pose_dataset_item_contract(tree):
    ds = DeepFashionAttrPoseDataset(pose_dir=tree['pose_dir'], texture_ann_dir=tree['texture_ann_file'],
                                    shape_ann_path=tree['shape_ann_path'])
    it = ds[2]
    assert it['img_name'] == tree['names'][2]
    assert it['densepose'].shape == (1, 512, 256)
    assert it['shape_attr'].dtype == torch.int64 and it['shape_attr'].shape == (15, )
    assert all(0 <= int(it[k]) <= 17 for k in ('upper_fused_attr', 'lower_fused_attr', 'outer_fused_attr'))


def test_annotation_files_must_agree(tree, tmp_path):
    import shutil
    bad = tmp_path / 'ann'
    shutil.copytree(tree['test_ann_file'], bad)
    lines = (bad / 'lower_fused.txt').read_text().splitlines()
    (bad / 'lower_fused.txt').write_text('\n'.join(lines[::-1]) + '\n')
    with pytest.raises(AssertionError):
        deepfashion.read_fused_annotations(str(bad))


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree absent')
def test_items_equal_the_reference_loaders(tree):
    def load(name):
        spec = importlib.util.spec_from_file_location(f'ref_{name}', os.path.join(REF, f'{name}.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    ours = DeepFashionAttrSegmDataset(img_dir=tree['test_img_dir'], segm_dir=tree['segm_dir'],
                                      pose_dir=tree['pose_dir'], ann_dir=tree['test_ann_file'])
    ref = load('segm_attr_dataset').DeepFashionAttrSegmDataset(
        img_dir=tree['test_img_dir'], segm_dir=tree['segm_dir'], pose_dir=tree['pose_dir'],
        ann_dir=tree['test_ann_file'])
    assert len(ours) == len(ref)
    for i in range(len(ref)):
        a, b = ours[i], ref[i]
        assert a.keys() == b.keys() and a['img_name'] == b['img_name']
        for k in ('image', 'segm', 'texture_mask'):
            assert torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k
        assert np.array_equal(a['densepose'], b['densepose'])
    ours_p = DeepFashionAttrPoseDataset(pose_dir=tree['pose_dir'], texture_ann_dir=tree['texture_ann_file'],
                                        shape_ann_path=tree['shape_ann_path'])
    ref_p = load('pose_attr_dataset').DeepFashionAttrPoseDataset(
        pose_dir=tree['pose_dir'], texture_ann_dir=tree['texture_ann_file'], shape_ann_path=tree['shape_ann_path'])
    for i in range(len(ref_p)):
        a, b = ours_p[i], ref_p[i]
        assert a.keys() == b.keys()
        assert np.array_equal(a['densepose'], b['densepose']) and torch.equal(a['shape_attr'], b['shape_attr'])
        for k in ('img_name', 'upper_fused_attr', 'lower_fused_attr', 'outer_fused_attr'):
            assert a[k] == b[k]
