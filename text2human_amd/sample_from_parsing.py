"""`python -m text2human_amd.sample_from_parsing -opt configs/sample_from_parsing.yml`

The reference's sample_from_parsing.py entry point on this package: same YAML, same
dataset tree, same output files ({results_root}/{img_name}).  `--batch-size` overrides the
reference's hard-coded 4 (larger batches are what the MI355X path is built for; note that
the images drawn for a batch depend on the batch composition through the shared RNG
stream, exactly as in the reference)."""
import argparse
import logging
import os.path as osp
import random

import torch

from . import options
from .data import DeepFashionAttrPoseDataset, DeepFashionAttrSegmDataset
from .models import create_model


def _setup(opt_path, log_name):
    opt = options.parse(opt_path, is_train=False)
    options.make_exp_dirs(opt)
    logger = logging.getLogger('base')
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('%(asctime)s %(levelname)s: %(message)s')
    for handler in (logging.StreamHandler(),
                    logging.FileHandler(osp.join(opt['path']['log'], f"{log_name}_{opt['name']}.log"))):
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    logger.info(options.dict2str(opt))
    opt = options.dict_to_nonedict(opt)
    seed = opt['manual_seed']
    if seed is None:
        seed = random.randint(1, 10000)
    logger.info(f'Random seed: {seed}')
    options.set_random_seed(seed)
    return opt, logger


def run(pose=False, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    ap.add_argument('--batch-size', type=int, default=4)
    args = ap.parse_args(argv)
    opt, logger = _setup(args.opt, 'test')
    if pose:
        dataset = DeepFashionAttrPoseDataset(pose_dir=opt['pose_dir'], texture_ann_dir=opt['texture_ann_file'],
                                             shape_ann_path=opt['shape_ann_path'])
    else:
        dataset = DeepFashionAttrSegmDataset(img_dir=opt['test_img_dir'], segm_dir=opt['segm_dir'],
                                             pose_dir=opt['pose_dir'], ann_dir=opt['test_ann_file'])
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=args.batch_size, shuffle=False)
    logger.info(f'Number of test set: {len(dataset)}.')
    model = create_model(opt)
    model.inference(loader, opt['path']['results_root'])


if __name__ == '__main__':
    run(pose=False)
