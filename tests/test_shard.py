"""Multi-GPU batch sharding exercised with world_size 2 on the
gloo backend (CPU): contiguous ragged split, barrier, max-over-ranks timing
reduction and the gather of finished uint8 images."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from text2human_amd import shard, synthetic


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 8, 9, 256):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_batch_slices_every_entry():
    batch = synthetic.parsing_batch(5, seed=3)
    parts = [shard.shard_batch(batch, r, 2) for r in range(2)]
    assert [len(p['img_name']) for p in parts] == [3, 2]
    assert torch.equal(torch.cat([p['segm'] for p in parts]), batch['segm'])
    assert parts[0]['img_name'] + parts[1]['img_name'] == batch['img_name']


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_total = 5  # ragged: rank 0 gets 3 images, rank 1 gets 2
        lo, hi = shard.shard_range(n_total, rank, world)
        g = torch.Generator().manual_seed(7)
        full = torch.randint(0, 256, (n_total, 8, 4, 3), generator=g, dtype=torch.uint8)
        mine = full[lo:hi].clone()
        shard.barrier(world)
        t = shard.max_over_ranks(1.0 + rank, world, torch.device('cpu'))
        assert t == float(world)
        got = shard.gather_images(mine, world, dst=0)
        if rank == 0:
            assert torch.equal(got, full)
            open(os.path.join(out_dir, 'ok'), 'w').write('1')
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_shard_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok').exists()


def _bcast_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        src = {'a': {'w': torch.randn(3, 4, generator=g), 'cnt': torch.tensor(2**40 + 12345),
                     'idx': torch.arange(2**24, 2**24 + 9), 'flag': torch.tensor([True, False, True])},
               'b': {'h': torch.randn(5, generator=g).half(), 'bf': torch.randn(7, generator=g).bfloat16(),
                     'd': torch.randn(2, 2, generator=g).double()}}
        got = shard.broadcast_state_dicts(src if rank == 0 else None, world, torch.device('cpu'))
        assert list(got) == ['a', 'b'] and list(got['a']) == ['w', 'cnt', 'idx', 'flag']
        for m in src:
            for k, v in src[m].items():
                assert got[m][k].dtype == v.dtype and got[m][k].shape == v.shape and torch.equal(got[m][k], v), (m, k)
        open(os.path.join(out_dir, f'ok{rank}'), 'w').write('1')
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_broadcast_state_dicts_keeps_every_dtype_bit_for_bit(tmp_path):
    """int64 counters / indices above 2^24, bool, fp16 / bf16 / fp64 entries travel in their own dtype
    (one flat buffer per dtype), not through fp32."""
    port = _free_port()
    mp.spawn(_bcast_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()
