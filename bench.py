#!/usr/bin/env python
"""Benchmark of the Text2Human sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of `sample_from_parsing` over one batch of synthetic
512x256 parsing maps per GPU (BASELINE.json configs[1]: batch 8 per GPU):
segm tokenizer -> 256-step texture-aware transformer index sampler -> index
refinement -> hierarchical VQGAN decode -> uint8 images, all inputs resident in
HBM before the timed region.  For N > 1 the driver launches this file with
torch.distributed.run; images are independent so the batch is sharded across
ranks with no data-path collective (weak scaling, per-rank batch fixed).

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     -- dominant kernel (the split-precision GEMM of the sampler Linears:
                  three fp16 partial products per fp32 multiply) algorithmic FLOP/s
                  measured live with HIP events on the launch stream, vs the 2.5
                  PFLOP/s dense 16-bit matrix peak of gfx950 (its fp32-equivalent
                  rate and the 157.3 TFLOP/s fp32 matrix peak are reported beside it);
  cpu_baseline -- the oracle (CPU port of the reference path) timed on this
                  box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: fp16 / bf16 MFMA dense (NOT the 2:1-sparse figure)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    ap.add_argument('--sample-steps', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sampler-steps', type=int, default=8)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-baseline-worker', action='store_true')
    ap.add_argument('--no-exact-fp32', action='store_true',
                    help='skip the extra step on the exact-fp32 (v_mfma_f32_32x32x2_f32) sampler kernels')
    ap.add_argument('--eager-gpu-baseline', action='store_true',
                    help='also time the oracle sampler as eager PyTorch-ROCm ops on this GPU (SURVEY.md 8(d))')
    return ap.parse_args()


def cpu_baseline_worker(sample_steps, n_sub, threads):
    """Times the oracle (oracle/torch_ref.py = CPU port of the reference path)
    on a bounded sample: B=1, tokenizer + n_sub sampler steps (scaled to
    `sample_steps`) + refine + decode.  Runs in its own process (see below)."""
    from oracle import torch_ref as R
    from text2human_amd import defaults, options, synthetic
    torch.set_num_threads(threads)
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    batch = synthetic.parsing_batch(1, seed=2021)
    torch.manual_seed(2021)
    with torch.no_grad():
        R.transformer_logits(torch.zeros(1, 512, dtype=torch.long), torch.zeros(1, 512, dtype=torch.long),
                             torch.zeros(1, 512, dtype=torch.long), sds['sampler'], heads={0})  # warm up
        t0 = time.perf_counter()
        tok = R.segm_tokens(batch['segm'], sds['segm_encoder'], sds['segm_quant_conv'],
                            sds['segm_quantizer']['embedding.weight']).view(1, -1)
        t1 = time.perf_counter()
        top = R.sample_fn(tok, batch['texture_mask'], sds['sampler'], sample_steps=n_sub,
                          noise=R.TorchNoise('cpu'))
        t2 = time.perf_counter()
        R.refine_and_decode(top, batch['texture_mask'], sds)
        t3 = time.perf_counter()
    per_image = (t1 - t0) + (t2 - t1) * (sample_steps / n_sub) + (t3 - t2)
    return dict(value=1.0 / per_image, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=(f'B=1: tokenizer {t1 - t0:.2f}s + {n_sub} of {sample_steps} sampler steps '
                        f'{t2 - t1:.2f}s (scaled x{sample_steps / n_sub:g}) + refine/decode {t3 - t2:.2f}s'
                        f' -> {per_image:.1f} s/image'))


def cpu_baseline(sample_steps, n_sub):
    """Runs the worker in a fresh process (no GPU context, own OpenMP pool) with a
    hard time box so the benchmark always finishes; threads = min(cores, 32):
    torch's CPU kernels on B=1 shapes stop scaling (and degrade) far below the
    hundreds of hardware threads of the GPU hosts."""
    import subprocess
    threads = min(os.cpu_count() or 1, 32)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--sample-steps',
           str(sample_steps), '--cpu-sampler-steps', str(n_sub), '--cpu-threads', str(threads)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if line:
            return json.loads(line[-1])
        return dict(value=None, unit='images/s', cores=threads, kind='port',
                    sample=f'worker failed: {r.stderr[-300:]}')
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='images/s', cores=threads, kind='port',
                    sample='worker exceeded its 240 s time box')


def eager_gpu_baseline(model, batch, sds, n_sub, dev):
    """The un-tuned GPU baseline of SURVEY.md 8(d): the oracle's sampler (the
    reference's algorithm as eager PyTorch-ROCm fp32 ops: rocBLAS/hipBLASLt GEMMs,
    unfused softmax / LayerNorm / GELU / Categorical tail) on this same GPU and batch,
    n_sub steps, next to this package's sampler on the same n_sub steps."""
    from oracle import torch_ref as R
    sd = {k: v.to(dev) for k, v in sds['sampler'].items()}
    tok = model.segm_tokens.view(batch['segm'].shape[0], -1)

    def timed(fn):
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n_sub)
        torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t0) / n_sub

    with torch.no_grad():
        eager = timed(lambda n: R.sample_fn(tok, batch['texture_mask'], sd, sample_steps=n, noise=R.TorchNoise(dev)))
    ours = timed(lambda n: model.sample_fn(temp=1, sample_steps=n))
    return dict(kind='oracle sampler as eager PyTorch-ROCm fp32 on the same GPU', sampler_ms_per_step=eager,
                this_package_sampler_ms_per_step=ours, speedup=eager / ours,
                sample=f'B={tok.shape[0]}, {n_sub} sampler steps each (the sampler is 97% of the path)')


def pmc_traffic(kernel_label):
    """HBM-side bytes per launch of the dominant kernel.  PMC counters cannot be
    read from inside the benchmark; they come from the separate rocprofv3 --pmc
    passes of this same bench command committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in their own passes, gfx950 FETCH x2 correction, see
    tools/pmc_summary.py) -- launch-weighted over the kernel's shapes."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')
    m = __import__('re').match(r'gemm_kernel<(\d+)x(\d+)x(\d+)(w8)?,amode=(\d),pro=(\d),btrans=(\d)>', kernel_label)
    if not os.path.exists(path):
        return {'traffic': None}
    if kernel_label.startswith('gemm_split'):
        rows = [r for r in json.load(open(path)) if r['kernel'].startswith('gemm_split_kernel')]
    elif m:
        bm, bn, bk, w8, am, pro, bt = m.groups()
        wm, wn = ('4', '2') if w8 else (('4', '1') if bn == '32' else ('2', '2'))
        name = f'gemm_kernel<{bm}, {bn}, {bk}, {wm}, {wn}, {am}, {pro}, {"true" if bt == "1" else "false"}>'
        rows = [r for r in json.load(open(path)) if r['kernel'] == name]
    else:
        return {'traffic': None}
    if not rows:
        return {'traffic': None}
    n = sum(r['launches'] for r in rows)
    mb = sum(r['traffic_mb'] * r['launches'] for r in rows) / n
    util = sum(r['mfma_util'] * r['launches'] for r in rows) / n
    return {'traffic': mb * 1e6, 'traffic_unit': 'bytes/launch', 'mfma_util_pmc': util,
            'traffic_source': 'profiles/r01_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'}


def main():
    args = parse_args()
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(args.sample_steps, args.cpu_sampler_steps,
                                             args.cpu_threads or (os.cpu_count() or 1))), flush=True)
        return
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU path exists)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or os.environ.get("T2H_FORCE_DIST") == "1"  # test hook: RCCL init with 1 rank
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # RCCL prints a version banner on stdout at communicator creation; stdout is
        # reserved for the ONE JSON line, so route fd 1 to stderr while RCCL comes up.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group('nccl', device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            # the banner goes through C stdio, which block-buffers when stdout is a pipe and
            # would otherwise flush it to the restored fd 1 at exit: drain it to stderr now
            import ctypes
            ctypes.CDLL(None).fflush(None)
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from text2human_amd import defaults, ops, options, shard, synthetic
    from text2human_amd.models import SampleFromParsingModel

    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    opt['sample_steps'] = args.sample_steps
    sds = synthetic.make_state_dicts(opt, seed=1234)
    model = SampleFromParsingModel(opt, state_dicts=sds)

    # this rank's shard of the global batch (contiguous split)
    lo, hi = shard.shard_range(args.batch * world, rank, world)
    full = synthetic.parsing_batch(args.batch * world, seed=2021)
    batch = dict(segm=full['segm'][lo:hi].to(dev), texture_mask=full['texture_mask'][lo:hi].to(dev),
                 img_name=full['img_name'][lo:hi])

    def one_step():
        options.set_random_seed(2021)
        model.feed_data(batch)
        top = model.sample_fn(temp=1, sample_steps=args.sample_steps)
        _, u8 = model.decode_indices(top, want_u8=True)
        return u8

    for _ in range(args.warmup):
        one_step()
    shard.barrier(2 if use_dist else 1)
    torch.cuda.synchronize()
    ops.gemm_profile_start(every=37)  # HIP-event pairs around a sample of GEMM launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        u8 = one_step()
    torch.cuda.synchronize()
    shard.barrier(2 if use_dist else 1)
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, 2 if use_dist else 1, dev)
    prof = ops.gemm_profile_stop()
    assert u8.shape == (hi - lo, 512, 256, 3)

    if rank != 0:
        return
    n_img = args.batch * world * args.steps
    out = {
        'metric': '512x256 images/sec (sample_from_parsing)',
        'value': n_img / elapsed,
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': ('f32' if os.environ.get('T2H_SPLIT_GEMM', '1') == '0' else
                  'f32 (sampler Linears + attention as 2xfp16-split MFMA, 3 partial products, fp32 accumulate: '
                  'fp32-class accuracy, tokens bit-exact vs the fp32 oracle; everything else exact-fp32 MFMA)'),
        'data': 'synthetic',
        'config': {
            'workload': (f'sample_from_parsing.yml batch={args.batch}/GPU, {args.sample_steps} sampling '
                         'steps, top+bottom VQGAN decode + index sampler (BASELINE.json configs[1])'),
            'global_batch': args.batch * world,
            'sample_steps': args.sample_steps,
            'weights': 'synthetic seed 1234 (reference .pth layout)',
            'rng': 'torch global generator (reference contract)',
            'parallelism': f'batch shard x{world}, no data-path collective',
        },
    }
    # dominant kernel = the GEMM instantiation with the largest total sampled time
    if prof:
        dom = max(prof.values(), key=lambda r: r['ms'])
        eq = dom['flops'] / (dom['ms'] * 1e-3) / 1e12  # fp32-equivalent 2*M*N*K per launch / time
        split = dom['kernel'].startswith('gemm_split')
        # The split-precision kernel's algorithm is three fp16 x fp16 partial products per
        # fp32 multiply on v_mfma_f32_32x32x16_f16, so its matrix-core roofline is the
        # dense 16-bit peak and its algorithmic work 3 * 2*M*N*K; the fp32-equivalent rate
        # and its ratio to the fp32-MFMA peak are reported next to it.
        mult, peak = (3.0, BF16_MFMA_PEAK_TFLOPS) if split else (1.0, FP32_MFMA_PEAK_TFLOPS)
        ach = eq * mult
        out['roofline'] = {
            'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': ach, 'peak': peak,
            'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
            'fp32_equivalent_tflops': eq, 'frac_of_fp32_mfma_peak': eq / FP32_MFMA_PEAK_TFLOPS,
            'launches_sampled': dom['n'], 'avg_launch_us': 1000.0 * dom['ms'] / dom['n'],
            'flop_per_launch': mult * dom['flops'] / dom['n'],
            'all_gemm_kernels': {k: {'TFLOP/s': v['flops'] / (v['ms'] * 1e-3) / 1e12, 'n': v['n'],
                                     'avg_us': 1000.0 * v['ms'] / v['n']} for k, v in prof.items()},
        }
        out['roofline'].update(pmc_traffic(dom['kernel']))
    # whole-path arithmetic rate against the same peak (26.17 TFLOP / image)
    out['path_tflops'] = 26.17 * out['value'] / world
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args.sample_steps, args.cpu_sampler_steps)
    if world == 1 and not args.no_exact_fp32:
        # the same step with the sampler's Linears / attention on the exact-fp32 matrix
        # instructions instead of the split-precision kernels (T2H_SPLIT_GEMM=0), for reference
        from text2human_amd import engine
        fast = model.sampler_fn
        model.sampler_fn = engine.SamplerNet(model.P, model._tf_desc, opt['bert_n_head'], 'tf', split=False)
        one_step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        model.sampler_fn = fast
        out['exact_fp32_path'] = {'value': args.batch / dt, 'unit': 'images/s', 'ms_per_step': 1000.0 * dt,
                                  'note': 'sampler Linears and attention on v_mfma_f32_32x32x2_f32 (bitwise fp32 '
                                          'fma chains); 1 warm-up + 1 timed step'}
    if world == 1 and args.eager_gpu_baseline:
        out['eager_gpu_baseline'] = eager_gpu_baseline(model, batch, sds, 16, dev)
    print(json.dumps(out), flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
