"""`python -m text2human_amd.sample_from_parsing -opt configs/sample_from_parsing.yml`

The reference's sample_from_parsing.py entry point on this package: same YAML, same
dataset tree, same output files ({results_root}/{img_name}).  `--batch-size` overrides the
reference's hard-coded 4 (larger batches are what the MI355X path is built for; note that
the images drawn for a batch depend on the batch composition through the shared RNG
stream, exactly as in the reference).

Several GPUs of one node: launch one process per GPU,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           -m text2human_amd.sample_from_parsing -opt configs/sample_from_parsing.yml --batch-size 32

Rank r takes the contiguous slice `shard.shard_range(len(dataset), r, world)` of the dataset (images are
independent: no data-path collective), seeds ITS shard with `manual_seed` -- so the oracle of rank r's images is
the reference run on that same slice with that seed --, runs `model.inference` over its own loader and writes its
own PNGs into the shared results directory, which rank 0 creates (an existing one is an error on every rank, like
the reference's `make_exp_dirs`).  The `.pth` files are read once, on rank 0, and reach the other ranks over RCCL
(`shard.broadcast_state_dicts`); the only other collectives are two barriers."""
import argparse
import logging
import os
import os.path as osp
import random

import torch

from . import options, shard, weights
from .data import DeepFashionAttrPoseDataset, DeepFashionAttrSegmDataset
from .models import create_model


def dist_env():
    """(rank, world, local_rank) of a torch.distributed.run launch; (0, 1, 0) for a plain `python -m`."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def init_dist(rank, world, local_rank):
    """One process per GPU: device = LOCAL_RANK, backend nccl (= RCCL over xGMI).  Without a GPU (the world-size-2
    tests of the host logic) gloo.  -> the torch.distributed module, or None for a single process."""
    if world == 1 and os.environ.get('T2H_FORCE_DIST') != '1':  # (T2H_FORCE_DIST=1: a one-rank group, test hook)
        return None
    import torch.distributed as dist
    if world == 1:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist.init_process_group('gloo')
    return dist


def _setup(opt_path, log_name, rank=0, world=1, dist=None):
    opt = options.parse(opt_path, is_train=False)
    # the results directory: made by rank 0 (FileExistsError if it exists, utils/util.py:22); every rank learns the
    # outcome before going on, so that no rank waits at a barrier for one that has raised
    err = [None]
    if rank == 0:
        try:
            options.make_exp_dirs(opt)
        except FileExistsError as e:
            err[0] = str(e)
    if dist is not None:
        dist.broadcast_object_list(err, src=0)
    if err[0] is not None:
        raise FileExistsError(err[0])
    logger = logging.getLogger('base')
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('%(asctime)s %(levelname)s: %(message)s' if world == 1 else
                            f'%(asctime)s %(levelname)s [rank {rank}/{world}]: %(message)s')
    handlers = [logging.StreamHandler()]
    if rank == 0:  # one log file, rank 0's (the reference's name)
        handlers.append(logging.FileHandler(osp.join(opt['path']['log'], f"{log_name}_{opt['name']}.log")))
    for handler in handlers:
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    if rank == 0:
        logger.info(options.dict2str(opt))
    opt = options.dict_to_nonedict(opt)
    seed = opt['manual_seed']
    if seed is None:
        seed = [random.randint(1, 10000)]
        if dist is not None:
            dist.broadcast_object_list(seed, src=0)
        seed = seed[0]
    logger.info(f'Random seed: {seed}')
    options.set_random_seed(seed)  # every rank seeds ITS shard with the run's seed
    return opt, logger


def run(pose=False, argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    ap.add_argument('--batch-size', type=int, default=4)
    args = ap.parse_args(argv)
    rank, world, local_rank = dist_env()
    dist = init_dist(rank, world, local_rank)
    try:
        opt, logger = _setup(args.opt, 'test', rank, world, dist)
        if pose:
            dataset = DeepFashionAttrPoseDataset(pose_dir=opt['pose_dir'], texture_ann_dir=opt['texture_ann_file'],
                                                 shape_ann_path=opt['shape_ann_path'])
        else:
            dataset = DeepFashionAttrSegmDataset(img_dir=opt['test_img_dir'], segm_dir=opt['segm_dir'],
                                                 pose_dir=opt['pose_dir'], ann_dir=opt['test_ann_file'])
        n_total = len(dataset)
        if dist is not None:
            lo, hi = shard.shard_range(n_total, rank, world)
            dataset = torch.utils.data.Subset(dataset, range(lo, hi))
            logger.info(f'Number of test set: {n_total}; this rank: items [{lo}, {hi}).')
        else:
            logger.info(f'Number of test set: {n_total}.')
        loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=args.batch_size, shuffle=False)
        if dist is not None:
            dev = torch.device('cuda', local_rank) if torch.cuda.is_available() else torch.device('cpu')
            # the `.pth` files are read by rank 0 alone; its outcome reaches every rank BEFORE the broadcast, so that a
            # missing / corrupt checkpoint ends all ranks with the error instead of leaving the others blocked in a
            # collective until the RCCL timeout
            sds0, err = None, [None]
            if rank == 0:
                try:
                    sds0 = load_state_dicts(opt)
                except Exception as e:  # noqa: BLE001 -- whatever the loader raised, every rank must learn it
                    err[0] = f'{type(e).__name__}: {e}'
            dist.broadcast_object_list(err, src=0)
            if err[0] is not None:
                raise RuntimeError(f'rank 0 could not load the checkpoints: {err[0]}')
            # (the helper's `world` argument is its "is distributed" switch)
            sds = shard.broadcast_state_dicts(sds0, 2, dev)
            model = create_model(opt, state_dicts=sds)
        else:
            model = create_model(opt)
        model.inference(loader, opt['path']['results_root'])
        if dist is not None:
            dist.barrier()  # every rank's files are on disk when any rank returns
    finally:
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()


def load_state_dicts(opt):
    """The `.pth` files named by the YAML, read once (rank 0 of a multi-GPU launch)."""
    return weights.load_checkpoints(opt)


if __name__ == '__main__':
    run(pose=False)
