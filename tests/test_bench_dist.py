"""bench.py's own multi-rank code path (shard of the global batch, per-rank seed, weight
broadcast from rank 0, barrier + max-over-ranks timing, per-rank gathers, ONE JSON line on
rank 0) executed with world_size 2 on gloo / CPU through torch.distributed.run, exactly as the
driver launches it on GPUs -- with `--stub-model` standing in for the HIP model."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(nproc, extra=()):
    env = dict(os.environ, T2H_NO_PIN='0', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'),
           '--gpus', str(nproc), '--steps', '3', '--warmup', '1', '--stub-model', '--batch', '3', *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f'stdout must hold exactly one JSON line, got: {lines}'
    # the harness keeps the last 8 KB of stdout: the contract line stays far below (a 26.5 KB line was lost)
    assert len(lines[0]) < 6000, len(lines[0])
    assert 'bench detail: {' in r.stderr  # the full result goes to stderr + gpurun_out/bench_detail*.json
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_bench_two_ranks_gloo():
    out = _run(2)
    assert out['n_gpus'] == 2 and out['rccl_world'] == 2 and out['dist_backend'] == 'gloo'
    assert out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 6
    assert len(out['per_rank_ms_per_step']) == 2 and all(t > 0 for t in out['per_rank_ms_per_step'])
    # max over ranks: the reported step time is not below any rank's own
    assert out['ms_per_step'] >= max(out['per_rank_ms_per_step']) - 1e-6
    assert abs(out['value'] - 6 * 1000.0 / out['ms_per_step']) < 1e-6 * out['value']
    # every rank worked on ITS shard (different data -> different checksums) with broadcast weights
    cs = out['per_rank_image_checksum']
    assert len(cs) == 2 and cs[0] != cs[1]
    # the configs[3] leg ran on both ranks too (its barrier / max-over-ranks collectives are the headline's)
    leg = out['other_configs']['parsing_b32']
    assert leg['global_batch'] == 2 * 2 * 3 and leg['value'] > 0 and leg['steps'] == 5
    # the stub walks the graph-replay host logic (schedule.RoundTables) on every rank's own shard
    assert leg['launch_mode'] == 'graph'
    assert out['detail'] == 'gpurun_out/bench_detail_n2.json'
    # the shards are those of the single-process run over the same global batch
    one = _run(1, ('--batch', '6'))
    assert one['rccl_world'] == 1 and len(one['per_rank_image_checksum']) == 1


@pytest.mark.timeout(300)
def test_bench_eight_ranks_gloo_is_the_drivers_node_shape():
    """The driver's N = 8 launch (one rank per GPU of a node), on gloo with the stub model: eight shards of the global
    batch, eight different checksums, max-over-ranks timing, per-rank core blocks, ONE line on rank 0 -- no 8-GPU node was
    ever available, so this is the only execution of the eight-rank host logic."""
    out = _run(8, ('--batch', '2'))
    assert out['n_gpus'] == 8 and out['rccl_world'] == 8 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 16 and out['config']['parallelism'].startswith('batch shard x8')
    assert len(out['per_rank_ms_per_step']) == 8 and out['ms_per_step'] >= max(out['per_rank_ms_per_step']) - 1e-6
    assert abs(out['value'] - 16 * 1000.0 / out['ms_per_step']) < 1e-6 * out['value']
    assert len(set(out['per_rank_image_checksum'])) == 8            # every rank its own shard
    assert out['other_configs']['parsing_b32']['global_batch'] == 8 * 2 * 2
    assert out['detail'] == 'gpurun_out/bench_detail_n8.json'


@pytest.mark.timeout(300)
def test_plain_python_bench_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher -- the form the driver uses at N = 1 -- must
    produce a 2-rank line by itself (bench.spawn_ranks re-executes the command under torch.distributed.run)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE',
                                                             'MASTER_ADDR', 'MASTER_PORT', 'T2H_FORCE_DIST')}
    env.update(T2H_NO_PIN='0', OMP_NUM_THREADS='1')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--stub-model',
           '--batch', '3']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_world'] == 2 and out['config']['global_batch'] == 6
    assert len(out['per_rank_ms_per_step']) == 2


def test_a_launcher_whose_world_size_differs_from_gpus_is_refused():
    env = dict(os.environ, T2H_NO_PIN='0', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--stub-model', '--batch', '3']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()


def test_broadcast_state_dicts_roundtrip_single_process():
    from text2human_amd import shard
    sds = {'a': {'w': torch.randn(3, 4), 'i': torch.arange(5)}}
    assert shard.broadcast_state_dicts(sds, 1, torch.device('cpu')) is sds
