"""The product's multi-GPU entry point (`python -m torch.distributed.run ... -m text2human_amd.sample_from_parsing`,
SURVEY.md 8(e); reference call site sample_from_parsing.py:38-49) with world_size 2 on gloo / CPU and a stub model
(tests/_sharded_entry_stub.py): the union of the written names is the single-process run's, no name is written twice,
every rank's files are the single-process result on ITS slice with the run's seed, the checkpoints are read once."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from text2human_amd import data, defaults, options, shard, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, 'tests', '_sharded_entry_stub.py')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _config(tmp_path, name, n=5):
    opt = defaults.sample_from_parsing()
    tree = synthetic.write_dataset_tree(str(tmp_path / 'data'), n=n, seed=11)
    names = tree.pop('names')
    opt.update(tree)
    opt.update(name=name, sample_steps=3, manual_seed=2021)
    return defaults.write_yaml(opt, str(tmp_path / f'{name}.yml')), names, opt


def _launch(nproc, cfg, cwd, reads, **extra_env):
    env = dict(os.environ, OMP_NUM_THREADS='1', T2H_STUB_READS_FILE=reads, **extra_env)
    if nproc == 1:
        cmd = [sys.executable, STUB, '-opt', cfg, '--batch-size', '2']
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(k, None)
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), STUB, '-opt', cfg, '--batch-size', '2']
    return subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=cwd)


def _expected(opt, lo, hi, seed=2021):
    """the stub's files for dataset items [lo, hi) in a process of its own seed: computed in-process"""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import _sharded_entry_stub as stub
    import tempfile
    ds = data.DeepFashionAttrSegmDataset(img_dir=opt['test_img_dir'], segm_dir=opt['segm_dir'],
                                         pose_dir=opt['pose_dir'], ann_dir=opt['test_ann_file'])
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, range(lo, hi)), batch_size=2, shuffle=False)
    options.set_random_seed(seed)
    with tempfile.TemporaryDirectory() as d:
        stub.StubModel(opt, state_dicts=stub.stub_state_dicts(opt)).inference(loader, d)
        return {n: open(os.path.join(d, n)).read() for n in os.listdir(d)}


@pytest.mark.timeout(300)
def test_two_ranks_write_the_union_once_each_on_their_own_slice(tmp_path):
    cfg, names, opt = _config(tmp_path, 'sharded')
    one_dir, two_dir = tmp_path / 'one', tmp_path / 'two'
    one_dir.mkdir()
    two_dir.mkdir()
    r1 = _launch(1, cfg, str(one_dir), str(tmp_path / 'reads1'))
    assert r1.returncode == 0, r1.stderr[-2000:]
    r2 = _launch(2, cfg, str(two_dir), str(tmp_path / 'reads2'))
    assert r2.returncode == 0, r2.stderr[-2000:]
    out1, out2 = one_dir / 'results' / 'sharded', two_dir / 'results' / 'sharded'
    log = 'test_sharded.log'
    f1 = sorted(f for f in os.listdir(out1) if f != log)
    f2 = sorted(f for f in os.listdir(out2) if f != log)
    assert f1 == sorted(names) and f2 == f1          # the union of the two ranks' names = the single-process run's
    assert os.path.exists(out2 / log)                 # rank 0's log, the reference's name
    # the checkpoints were read by ONE process (rank 0) and broadcast
    assert open(tmp_path / 'reads2').read().split() == ['0']
    # every rank's files = the single-process result on ITS contiguous slice, seeded with the run's seed
    nopt = options.dict_to_nonedict(opt)
    want = {}
    for r in range(2):
        lo, hi = shard.shard_range(len(names), r, 2)
        assert (lo, hi) == ((0, 3), (3, 5))[r]
        part = _expected(nopt, lo, hi)
        assert not set(part) & set(want)
        want.update(part)
    got = {n: open(out2 / n).read() for n in f2}
    assert got == want
    # and the single process = one slice over everything
    assert {n: open(out1 / n).read() for n in f1} == _expected(nopt, 0, len(names))
    # rank 1's second batch differs from what a single process draws for the same images (its own seeded stream)
    assert got[names[3]] != open(out1 / names[3]).read()
    # an existing results directory is an error on EVERY rank (no rank hangs at a barrier)
    r3 = _launch(2, cfg, str(two_dir), str(tmp_path / 'reads3'))
    assert r3.returncode != 0 and 'FileExistsError' in r3.stderr


@pytest.mark.timeout(300)
def test_a_checkpoint_rank_0_cannot_load_ends_every_rank_with_the_error(tmp_path):
    """rank 0 raising inside the checkpoint load used to leave the other ranks blocked in the broadcast
    until the collective's timeout; the outcome now reaches every rank first and all of them raise."""
    import time
    cfg, _, _ = _config(tmp_path, 'badload')
    d = tmp_path / 'run'
    d.mkdir()
    t0 = time.time()
    r = _launch(2, cfg, str(d), str(tmp_path / 'reads'), T2H_STUB_LOAD_FAILS='1')
    assert r.returncode != 0 and time.time() - t0 < 120
    assert 'rank 0 could not load the checkpoints: FileNotFoundError' in r.stderr
    assert r.stderr.count('rank 0 could not load the checkpoints') >= 2   # both ranks raised it
