#!/usr/bin/env python
"""Benchmark of the Text2Human sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config parsing|pose|hires]

One "step" = one full pass of the hot path over one batch of synthetic inputs per GPU, all
inputs resident in HBM before the timed region:

  parsing (default; BASELINE.json configs[1]): `sample_from_parsing`, batch 8 per GPU: segm
          tokenizer -> 256-step texture-aware transformer index sampler -> index refinement ->
          hierarchical VQGAN decode -> uint8 512x256 images;
  pose    (configs[2]): `sample_from_pose`, batch 32 per GPU: ShapeUNet parsing generator ->
          tokenizer -> sampler -> refine -> decode;
  hires   (configs[4]): the 1024x512 upscaled hierarchy (SURVEY.md 8(d) interpretation: sample at
          32x16, nearest-x2 both quantised latents, fully-convolutional decode), batch 8 per GPU.

The default run (`--config parsing`) also times, inside the same command, the other
configurations for 1 warm-up + 5 steps each and reports them under "other_configs": configs[3]'s
per-GPU share (sample_from_parsing at 32 images per GPU; its global batch at N = 8 IS configs[3]),
and at N = 1 configs[2] (pose, B = 32) and configs[4]'s per-GPU share (hires, B = 8).

N > 1: either the driver launches this file with torch.distributed.run (one rank per GPU; WORLD_SIZE must
equal --gpus), or plain `python bench.py --gpus N` spawns its own N ranks under torch.distributed.run
(spawn_ranks).  Images are independent, so the batch is sharded across ranks with no data-path collective
(weak scaling: the headline line keeps configs[1]'s 8 images per GPU at every N,
"other_configs.parsing_b32" keeps 32 per GPU); weights are synthesised on rank 0 and broadcast over RCCL.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     -- dominant kernel (the split-precision GEMM of the sampler Linears: per fp32 multiply the fp16
                  hi*hi product + both cross terms in one 8-bit instruction on x8 operands; three fp16
                  products with T2H_X8=0) measured live with HIP events that receive the kernel's own
                  start / end: `frac` = EXECUTED matrix FLOP/s over the dense peak of that instruction
                  mix (3750 TFLOP/s; 2500 on fp16 planes), `frac_useful` = the reference's fp32 FLOP
                  count over the fp16 peak; `worst_instantiation` = the tile configuration furthest
                  below that peak among those carrying >= 5 % of the kernel time;
  stages       -- per-stage times (HIP events) with the decode stage's compute AND HBM fractions;
  stages       also carry the sampler's schedule: (sample, step) pairs possible / needed / evaluated;
  parity       -- the split-precision step (sampler AND decoder) against the exact-fp32 step on the
                  same batch / seed: tokens, bottom indices, image;
  cpu_baseline -- kind "reference": the UNMODIFIED reference SampleFromParsingModel (oracle/_ref byte code through
                  oracle/ref_shim.py) on this box's host cores, all sampling steps once; kind "port"
                  (oracle/torch_ref.py, a bounded sample) only where the byte code is absent.
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
FP8_MFMA_PEAK_TFLOPS = 5000.0   # MI355X_MICROARCH.md: fp8 dense through the block-scaled K = 64 / 128 instructions
HBM_PEAK_TBS = 8.0              # MI355X_MICROARCH.md: HBM3E spec
MAX_CLOCK_GHZ = 2.4             # MI355X_MICROARCH.md: max clock, the clock the MFMA peaks are quoted at

# algorithmic work per image (exact, 2 * MAC) and algorithmic HBM bytes
GFLOP_IMAGE = dict(sampler_step=99.858, tokenizer=40.49, refine=2.19, decode=562.88, pose=239.86,
                   decode_hires=2380.0)
# of the 99.858 GFLOP of one reference sampler evaluation 9.664 are the 18 full [512, 1024] head
# projections (models/archs/transformer_arch.py:271); the HIP path runs the head of a token's own
# texture for the changed tokens only, so the work it EXECUTES per evaluation is the 24 layers
GFLOP_SAMPLER_LAYERS = 99.858 - 18 * 2 * 512 * 1024 * 512 / 1e9
# one evaluation of the 24 layers, fp32-equivalent (2 * MAC): the four Linears / the two attention products
GFLOP_SAMPLER_LINEARS = 24 * 2 * 512 * (512 * 1536 + 512 * 512 + 2 * 512 * 2048) / 1e9   # 77.31
GFLOP_SAMPLER_ATTN = 24 * 2 * 2 * 8 * 512 * 512 * 64 / 1e9                                 # 12.88


def sampler_matrix_time_frac(evaluations, seconds, x8, pmc=None):
    """Fraction of the sampler stage's time its matrix instructions need at their DENSE rates (the honest
    sampler-wide utilisation).  Per fp32 multiply-add the kernels issue three partial products;
    in units of the time ONE fp16 product takes on v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s):
      attention (always fp16 planes)            3 units  (hi*hi + two cross terms on the fp16 instruction)
      Linears on fp16 planes (T2H_X8=0)         3 units
      Linears on x8 operands (default)          2 units  (hi*hi: 1; BOTH cross terms in one 8-bit instruction at
                                                          twice the rate: 2 * 1/2)
    -> frac = (u_lin * 77.31 + 3 * 12.88) GFLOP * evaluations / seconds / 2.5 PFLOP/s.  `pmc` (the counter figure of
    the committed profile, SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE launch-weighted over the split GEMMs) is carried
    beside it for comparison."""
    u_lin = 2.0 if x8 else 3.0
    units = (u_lin * GFLOP_SAMPLER_LINEARS + 3.0 * GFLOP_SAMPLER_ATTN) * 1e9 * evaluations
    return {'matrix_time_frac': units / seconds / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            'matrix_time_units': {'linears': u_lin, 'attention': 3.0},
            'mfma_util_pmc_gemms': pmc}


DECODE_BYTES_IMAGE = dict(parsing=1.870e9, hires=7.48e9)
DECODE_WEIGHT_BYTES = 216.5e6
WORKLOADS = {
    'parsing': dict(batch=8, ref='BASELINE.json configs[1]', metric='512x256 images/sec (sample_from_parsing)',
                    desc='sample_from_parsing.yml, top+bottom VQGAN decode + index sampler'),
    'parsing_b32': dict(batch=32, ref="BASELINE.json configs[3]'s per-GPU share: batch 256 sharded over 8 GPUs",
                        metric='512x256 images/sec (sample_from_parsing)',
                        desc='sample_from_parsing.yml, top+bottom VQGAN decode + index sampler'),
    'pose': dict(batch=32, ref='BASELINE.json configs[2]', metric='512x256 images/sec (sample_from_pose)',
                 desc='sample_from_pose.yml (ParsingGen -> hierarchy VQGAN -> sampler end-to-end)'),
    'hires': dict(batch=8, ref="BASELINE.json configs[4]'s per-GPU share: batch 64 over 8 GPUs",
                  metric='1024x512 images/sec (upscaled hierarchy)',
                  desc='1024x512 upscaled hierarchy (32x16 sampling, nearest-x2 latents, 64x32 / 128x64 '
                       'token grids through the fully-convolutional decoders)'),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', choices=['parsing', 'pose', 'hires'], default='parsing')
    ap.add_argument('--batch', type=int, default=0, help='images per GPU per step (0 = the config default)')
    ap.add_argument('--sample-steps', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sampler-steps', type=int, default=32)
    ap.add_argument('--cpu-repeats', type=int, default=3)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-baseline-worker', action='store_true')
    ap.add_argument('--no-exact-fp32', action='store_true',
                    help='skip the extra steps on the exact-fp32 (v_mfma_f32_32x32x2_f32) sampler kernels')
    ap.add_argument('--exact-steps', type=int, default=3)
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the "other_configs" legs (configs[3] per-GPU share, pose B=32, hires B=8) of the default run')
    ap.add_argument('--no-eager-leg', action='store_true',
                    help='skip the individual-launches leg (T2H_GRAPH=0 + HIP-event sampling of the GEMM launches) '
                         'that the roofline object comes from')
    ap.add_argument('--other-steps', type=int, default=5, help='timed steps of each other_configs leg (after 1 warm-up)')
    ap.add_argument('--eager-steps', type=int, default=2, help='steps of the individual-launches leg (after 1 warm-up)')
    ap.add_argument('--no-eager-gpu-baseline', action='store_true',
                    help='skip the reference sampler as eager PyTorch-ROCm ops on this GPU (SURVEY.md 8(d))')
    ap.add_argument('--eager-gpu-steps', type=int, default=8)
    ap.add_argument('--stub-model', action='store_true',
                    help='TEST ONLY (tests/test_bench_dist.py): CPU + gloo + a stub model, to exercise the '
                         'multi-rank shard / seed / timing / gather logic of this file without a GPU')
    return ap.parse_args(argv)


# --------------------------------------------------------------------------- CPU baseline


def cpu_baseline_worker(sample_steps, n_sub, threads, repeats):
    """Runs in its own process (no GPU context, own OpenMP pool; see cpu_baseline()).  B=1, seed 2021, `threads`
    host threads.

    kind "reference": the UNMODIFIED reference `SampleFromParsingModel` (models/sample_model.py:215-328) through
    oracle/ref_shim.py -- from oracle/_ref's byte code on the GPU box (oracle/make_ref.py), from /root/reference in
    the build container -- on the reference's own call sequence `feed_data` + `sample_and_refine` (what
    `inference` runs per batch, sample_model.py:355-359), ALL `sample_steps` sampling steps, once, after a 2-step
    warm-up of the same call.
    kind "port": only where the reference is not available: oracle/torch_ref.py on a bounded sample (tokenizer +
    n_sub sampler steps scaled to `sample_steps` + refine + decode, median of `repeats`)."""
    import contextlib
    import io
    import tempfile
    from oracle import ref_shim, torch_ref as R
    from text2human_amd import defaults, options, synthetic
    torch.set_num_threads(threads)
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    batch = synthetic.parsing_batch(1, seed=2021)
    if ref_shim.available() and os.environ.get('T2H_CPU_BASELINE') != 'port':
        ns = ref_shim.load_reference('cpu')
        with tempfile.TemporaryDirectory() as d:
            o = synthetic.write_checkpoints(opt, d, seed=1234)
            with contextlib.redirect_stdout(io.StringIO()):
                model = ns.sample_model.SampleFromParsingModel(o)
        with torch.no_grad():
            model.sample_steps = 2
            model.feed_data(batch)
            model.sample_and_refine('/nonexistent', batch['img_name'])  # warm-up (the shim captures save_image)
            model.sample_steps = sample_steps
            ns.util.set_random_seed(2021)
            t0 = time.perf_counter()
            model.feed_data(batch)
            t1 = time.perf_counter()
            model.sample_and_refine('/nonexistent', batch['img_name'])
            t2 = time.perf_counter()
        img = ref_shim.saved_images[-1][0]
        per_image = t2 - t0
        return dict(value=1.0 / per_image, unit='images/s', cores=torch.get_num_threads(), kind='reference',
                    sample=(f'unmodified reference ({ref_shim.kind()}) SampleFromParsingModel, B=1, seed 2021: feed_data '
                            f'{t1 - t0:.2f}s + sample_and_refine (all {sample_steps} steps + refine + decode) {t2 - t1:.1f}s '
                            f'= {per_image:.1f} s/image, one run after a 2-step warm-up; image mean {float(img.mean()):.4f}'))
    sds = synthetic.make_state_dicts(opt, seed=1234)
    runs = []
    with torch.no_grad():
        R.transformer_logits(torch.zeros(1, 512, dtype=torch.long), torch.zeros(1, 512, dtype=torch.long),
                             torch.zeros(1, 512, dtype=torch.long), sds['sampler'], heads={0})  # warm up
        for _ in range(repeats):
            torch.manual_seed(2021)
            t0 = time.perf_counter()
            tok = R.segm_tokens(batch['segm'], sds['segm_encoder'], sds['segm_quant_conv'],
                                sds['segm_quantizer']['embedding.weight']).view(1, -1)
            t1 = time.perf_counter()
            top = R.sample_fn(tok, batch['texture_mask'], sds['sampler'], sample_steps=n_sub,
                              noise=R.TorchNoise('cpu'))
            t2 = time.perf_counter()
            R.refine_and_decode(top, batch['texture_mask'], sds)
            t3 = time.perf_counter()
            runs.append(((t1 - t0) + (t2 - t1) * (sample_steps / n_sub) + (t3 - t2), t1 - t0, t2 - t1, t3 - t2))
    runs.sort()
    per_image, tk, sm, dc = runs[len(runs) // 2]
    return dict(value=1.0 / per_image, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=(f'reference byte code (oracle/_ref) absent: the PORT oracle/torch_ref.py, median of {repeats}: B=1, '
                        f'tokenizer {tk:.2f}s + {n_sub} of {sample_steps} sampler steps '
                        f'{sm:.2f}s (scaled x{sample_steps / n_sub:g}) + refine/decode {dc:.2f}s'
                        f' -> {per_image:.1f} s/image'))


def cpu_baseline(sample_steps, n_sub, repeats):
    """Runs the worker in a fresh process (no GPU context, own OpenMP pool) with a hard time
    box so the benchmark always finishes; threads = min(cores, 32): torch's CPU kernels on B=1
    shapes stop scaling (and degrade) far below the hundreds of hardware threads of the hosts."""
    import subprocess
    threads = min(os.cpu_count() or 1, 32)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--sample-steps',
           str(sample_steps), '--cpu-sampler-steps', str(n_sub), '--cpu-threads', str(threads),
           '--cpu-repeats', str(repeats)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    fail = None
    for attempt in ('reference', 'port'):
        if attempt == 'port':
            env['T2H_CPU_BASELINE'] = 'port'  # the reference leg failed or ran out of its box: the bounded port
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300 if attempt == 'reference' else 200,
                               env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if line:
                out = json.loads(line[-1])
                if fail:
                    out['sample'] = f'({fail}) ' + out['sample']
                out['calibration'] = reference_calibration()
                return out
            fail = f'{attempt} worker failed: {r.stderr[-200:]}'
        except subprocess.TimeoutExpired:
            fail = f'{attempt} worker exceeded its time box'
        if env.get('T2H_CPU_BASELINE') == 'port':
            break
    return dict(value=None, unit='images/s', cores=threads, kind='port', sample=fail)


def reference_calibration():
    """How the port (oracle/torch_ref.py, what `cpu_baseline` times here) relates to the UNMODIFIED reference on the
    same cores: the reference cannot travel to the GPU box, so oracle/time_reference_vs_port.py times both in the
    build container (same synthetic checkpoints, parsing map and seed; tokens and image identical) and the committed
    result is quoted here.  ratio < 1: the port is FASTER than the reference, i.e. the baseline is generous."""
    path = os.path.join(ROOT, 'profiles', 'r04_cpu_reference_vs_port.json')
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    ref, port = d['reference'], d['port']
    return {'source': 'profiles/r04_cpu_reference_vs_port.json (oracle/time_reference_vs_port.py, build container)',
            'sampler_steps': d['config']['sampler_steps'], 'threads': d['config']['threads'], 'batch': d['config']['batch'],
            'port_over_reference_time': {'tokenizer': d['ratio_port_over_reference_tokenizer_s'],
                                         'sampler': d['ratio_port_over_reference_sampler_s'],
                                         'refine_decode': d['ratio_port_over_reference_refine_decode_s']},
            'reference_s': {k: ref[k] for k in ('tokenizer_s', 'sampler_s', 'refine_decode_s')},
            'port_s': {k: port[k] for k in ('tokenizer_s', 'sampler_s', 'refine_decode_s')},
            'tokens_equal': d['tokens_equal'], 'image_max_abs': d['image_max_abs']}


def eager_gpu_baseline(model, batch, sds, opt, n_sub, dev):
    """The un-tuned GPU baseline of SURVEY.md 8(d): the reference's sampler as eager PyTorch-ROCm fp32 ops on this
    same GPU and batch -- the UNMODIFIED reference model (`sample_fn`, models/sample_model.py:256-328, through
    oracle/ref_shim.py: oracle/_ref's byte code on the GPU box) where it is available, else the oracle port."""
    import contextlib
    import io
    import tempfile
    from oracle import ref_shim, torch_ref as R
    from text2human_amd import options, synthetic
    B = model.batch_size

    def timed(fn):
        fn(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n_sub)
        torch.cuda.synchronize()
        return 1000.0 * (time.perf_counter() - t0) / n_sub

    kind = None
    ns = None
    if ref_shim.available():
        try:
            ns = ref_shim.load_reference(dev)
        except Exception:  # noqa: BLE001 -- a baseline leg never takes the bench down
            ns = None
    if ns is not None:
        try:
            with tempfile.TemporaryDirectory() as d:
                o = synthetic.write_checkpoints(opt, d, seed=1234)
                with contextlib.redirect_stdout(io.StringIO()):
                    ref = ns.sample_model.SampleFromParsingModel(o)
            with torch.no_grad():
                ref.feed_data({k: batch[k] for k in ('segm', 'texture_mask', 'img_name')})
                options.set_random_seed(2021)
                eager = timed(lambda n: ref.sample_fn(temp=1, sample_steps=n))
            kind = f'unmodified reference sample_fn ({ref_shim.kind()}) as eager PyTorch-ROCm fp32 on the same GPU'
            del ref
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            kind = None
            print(f'eager_gpu_baseline: reference leg failed ({type(e).__name__}: {e})', file=sys.stderr)
    if kind is None:
        sd = {k: v.to(dev) for k, v in sds['sampler'].items()}
        tok = model.segm_tokens.view(B, -1)
        with torch.no_grad():
            eager = timed(lambda n: R.sample_fn(tok, model.texture_mask, sd, sample_steps=n, noise=R.TorchNoise(dev)))
        kind = 'oracle port sample_fn as eager PyTorch-ROCm fp32 on the same GPU'
    options.set_random_seed(2021)
    ours = timed(lambda n: model.sample_fn(temp=1, sample_steps=n))
    return dict(kind=kind, sampler_ms_per_step=eager, this_package_sampler_ms_per_step=ours, speedup=eager / ours,
                sample=f'B={B}, {n_sub} sampler steps each after 2 warm-up steps (the sampler is 97% of the path)')


# --------------------------------------------------------------------------- profile side data


def kernel_src_digest():
    """sha256 over the kernel sources + the C-ABI header: profile summaries under profiles/ carry
    the digest of the tree they were taken from (tools/pmc_summary.py, tools/rocprof_summary.py) and
    are only quoted here when it equals this tree's."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'text2human_amd', 'csrc')
    for f in sorted(os.listdir(csrc)):
        if f.endswith(('.hip', '.h')):
            h.update(open(os.path.join(csrc, f), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 't2h_hip.h'), 'rb').read())
    return h.hexdigest()[:16]


def profile_side_data(kernel_label, config):
    """What HIP events cannot give: HBM-side bytes per launch and matrix-pipe busy fraction of the
    dominant kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES in separate
    passes of this same bench command, gfx950 FETCH x2 correction, tools/pmc_summary.py) and its
    average duration by rocprofv3 --kernel-trace (tools/rocprof_summary.py) -- launch-weighted over
    the kernel's shapes.  Read from the committed round-4 summaries, and ONLY if they were taken
    from this tree's kernel sources (digest check); otherwise the fields are null."""
    out = {'traffic': None}
    if not kernel_label.startswith('gemm_split'):
        return out
    out['profile_kernel'] = kernel_label
    want = kernel_src_digest()
    # (parsing at 32 images per GPU runs the sampler GEMMs at the pose configuration's shapes, M = 16384: its rows are
    # taken from that configuration's passes)
    src_cfg = 'pose' if config == 'parsing_b32' else config
    if src_cfg != config:
        out['traffic_note'] = 'GEMM rows of the pose configuration (same sampler shapes, M = 16384)'
    tag = '' if src_cfg == 'parsing' else f'_{src_cfg}'
    rounds = ('r06', 'r05', 'r04')  # newest first: the first summary taken from THIS tree's kernel sources is quoted
    stale = None
    for rnd in rounds:
        path = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_summary{tag}.json')
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get('kernel_src_sha') != want:
            stale = stale or f'profiles/{os.path.basename(path)} is from other kernel sources ({d.get("kernel_src_sha")} != {want})'
            continue
        rows = [r for r in d['rows'] if r['kernel'].startswith('gemm_split')]
        n = sum(r['launches'] for r in rows)
        if n:
            out['per_kernel'] = {r['kernel']: {'mfma_util': r['mfma_util'], 'traffic_mb': r['traffic_mb']} for r in rows}
            out.update(traffic=sum(r['traffic_mb'] * r['launches'] for r in rows) / n * 1e6,
                       traffic_unit='bytes/launch',
                       mfma_util_pmc=sum(r['mfma_util'] * r['launches'] for r in rows) / n,
                       traffic_source=f'profiles/{os.path.basename(path)} (rocprofv3 --pmc, separate passes; '
                                      f'kernel sources {want})')
            break
    if out['traffic'] is None and stale:
        out['traffic_note'] = stale
    for rnd in rounds:
        path = os.path.join(ROOT, 'profiles', f'{rnd}_bench_{src_cfg}_kernel_stats.json')
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get('kernel_src_sha') != want:
            continue
        rows = [r for r in d['rows'] if r['kernel'].startswith('gemm_split')]
        n = sum(r['calls'] for r in rows)
        if n:
            out['avg_launch_us_rocprof'] = sum(r['avg_us'] * r['calls'] for r in rows) / n
            if d.get('total_kernel_ms'):
                for r in rows:  # (the two tables may list a kernel under several grids: shares add up)
                    e = out.setdefault('per_kernel', {}).setdefault(r['kernel'], {})
                    e['share'] = e.get('share', 0.0) + r['total_ms'] / d['total_kernel_ms']
            out['rocprof_source'] = f'profiles/{os.path.basename(path)}'
            break
    return out


# --------------------------------------------------------------------------- distributed glue


def init_dist(backend, dev):
    """RCCL prints a version banner on stdout at communicator creation; stdout is reserved for
    the ONE JSON line, so fd 1 is routed to stderr while the communicator comes up."""
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if int(os.environ.get('WORLD_SIZE', 1)) == 1:  # T2H_FORCE_DIST=1 without a launcher: a one-rank group
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29531')
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        dist.barrier()
        if backend == 'nccl':
            torch.cuda.synchronize()
    finally:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # the banner goes through C stdio: drain it to stderr now
        sys.stdout.flush()
        os.dup2(saved_fd, 1)
        os.close(saved_fd)
    return dist


def spawn_ranks(n, argv):
    """`python bench.py --gpus N ...` without a launcher: re-executes the same command under torch.distributed.run,
    one rank per GPU on this node (rendezvous on 127.0.0.1, a free port), stdout / stderr passed through -- rank 0
    prints the ONE JSON line.  Returns nothing; exits with the launcher's status if it is not 0."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # (dmabuf IPC: what RCCL needs on this pool's hosts)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *argv]
    sys.stdout.flush()
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def pin_launch_thread(local_rank, local_world):
    """One disjoint block of host cores per rank, so eight launch threads (each doing one host
    read per sampling step) never share or migrate between cores.  T2H_NO_PIN=1 disables."""
    if os.environ.get('T2H_NO_PIN') == '1' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
        k = max(1, len(cpus) // max(1, local_world))
        mine = cpus[local_rank * k:(local_rank + 1) * k] or cpus
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1]]
    except OSError:
        return None


class StubModel:
    """TEST ONLY: stands in for the model classes on CPU so that the multi-rank logic of this
    file (shard, per-rank seed, barrier, max-over-ranks, gathers, the JSON line) runs under gloo."""

    def __init__(self, sds):
        self.w = float(sds['sampler']['w'].sum())
        self.batch_size = 0

    def feed_data(self, batch):
        self.segm = batch['segm']
        self.batch_size = self.segm.shape[0]

    last_stats = None
    last_launch_mode = 'graph'

    def sample_fn(self, temp=1, sample_steps=None):
        """Walks the HOST logic of the product's graph-replay sampler (schedule.group_rounds -> schedule.RoundTables
        -> one "replay" per round, engine.RoundGraph.run) on this rank's shard with a schedule drawn from the
        (re-seeded) global generator: every token row must be sampled exactly once, in a round of its own sample."""
        import numpy as np
        from text2human_amd import schedule
        time.sleep(0.01)
        B, T, steps = self.batch_size, 16, 12
        u = torch.rand(steps, B * T).numpy()  # consumes the global generator like the reference's per-step rand
        step = np.zeros(B * T, dtype=np.int64)
        for i, t in enumerate(range(steps, 0, -1)):
            hit = (u[i] < 1.0 / t) & (step == 0)
            step[hit] = t
        order, start, round_steps = schedule.group_rounds(step, B, T, compact=True)
        maxr = -(-int(np.diff(start).max()) // 16) * 16
        tables = schedule.RoundTables(order, order.astype(np.int64) * 3 + 1, start, maxr)
        tok = np.full(B * T, -1, dtype=np.int64)

        def body(rows, vals):
            assert (vals == rows.astype(np.int64) * 3 + 1).all()
            tok[rows] = rows % 7          # (padding repeats a row of the same round: the same token twice)

        replays = tables.replay(body)
        assert replays == tables.n_rounds - 1 and (tok >= 0).all()
        StubModel.last_stats = dict(schedule.stats(round_steps, steps), replays=replays)
        self.sampler_fn = self  # (ConfigRun.timed reads sampler_fn.last_stats / last_launch_mode)
        r = torch.from_numpy(tok.reshape(B, T)[:, :4].astype(np.float64)) / 7.0
        return [(self.segm.reshape(self.batch_size, -1)[:, :4] + r + self.w).double()]

    def decode_indices(self, top, want_u8=True, upscale=False, return_inter=False):
        u8 = (top[0][:, :3].abs() * 1000).to(torch.int64).remainder(256).to(torch.uint8)
        return None, u8.view(self.batch_size, 1, 1, 3).expand(self.batch_size, 512, 256, 3).contiguous()


# --------------------------------------------------------------------------- one configuration


# tile configuration (ops.SPLIT_CFG_NAMES) -> the leading template arguments <BM, BN, WARPS_M, WARPS_N, KS, PP> of its
# gemm_split_kernel instantiation as rocprofv3 prints them (csrc/gemm_split.hip, t2h_gemm_split_f32's dispatch)
SPLIT_CFG_TEMPLATE = {'128x64': '128, 64, 2, 2, 1, 0', '128x128': '128, 128, 4, 2, 1, 0', '64x64': '64, 64, 2, 2, 1, 0',
                      '128x64, 8 waves': '128, 64, 4, 2, 1, 0', '128x256': '128, 256, 4, 2, 1, 0',
                      '128x64, 2 K groups': '128, 64, 2, 2, 2, 0', '256x128, ping-pong LDS-DMA': '256, 128, 4, 2, 1, 2',
                      '128x192, ping-pong LDS-DMA': '128, 192, 4, 2, 1, 2'}


def worst_instantiation(inst, side, x8):
    """The split-GEMM instantiation furthest below its roofline among those that carry time (>= 5 % of all kernel time
    by the committed kernel trace; all of them when no trace of these sources is committed):
    {name, frac (live, executed / peak of the instruction mix), mfma_util_pmc, share (of all kernel time)}."""
    rows = []
    for name, e in inst.items():
        m = name[len('gemm_split_kernel<'):-1]
        tmpl = SPLIT_CFG_TEMPLATE.get(m)
        full = f'gemm_split_kernel<{tmpl}, {1 if x8 else 0}>' if tmpl else None
        pm = (side.get('per_kernel') or {}).get(full, {})
        rows.append({'name': full or name, 'tile': m, 'frac': e['frac'], 'avg_us': e['avg_us'],
                     'mfma_util_pmc': pm.get('mfma_util'), 'share': pm.get('share')})
    if not rows:
        return None
    # (no trace of these sources committed: weigh by the live samples -- every 37th launch -- of the split GEMMs alone)
    tot = sum(e.get('n', 0) * e['avg_us'] for e in inst.values()) or 1.0
    for r, e in zip(rows, inst.values()):
        r['share_of_gemm_time_live'] = e.get('n', 0) * e['avg_us'] / tot
    heavy = [r for r in rows if (r['share'] if r['share'] is not None else r['share_of_gemm_time_live']) >= 0.05] or rows
    return min(heavy, key=lambda r: r['frac'])


def gemm_roofline(prof, config):
    """`roofline` object of the dominant GEMM instantiation (largest sampled time) from the HIP-event
    samples of ops.gemm_profile_*."""
    dom = max(prof.values(), key=lambda r: r['ms'])
    eq = dom['flops'] / (dom['ms'] * 1e-3) / 1e12  # fp32-equivalent 2*M*N*K per launch / time
    split = dom['kernel'].startswith('gemm_split')
    x8 = dom['kernel'] == 'gemm_split_kernel<x8>'
    # The split-precision kernel's algorithm is three partial products per fp32 multiply: its executed work is
    # 3 * 2*M*N*K (`frac`); the reference's own FLOP count against the fp16 peak is `frac_useful`.
    #   <2xfp16>: all three on v_mfma_f32_32x32x16_f16 -> the dense 16-bit peak;
    #   <x8>: hi*hi on that instruction, the two cross terms on v_mfma_scale_f32_32x32x64_f8f6f4 (8-bit operands, twice
    #         the rate) -> the peak of THAT instruction mix: 3 / (1 / 2500 + 2 / 5000) = 3750 TFLOP/s
    mult, peak = (3.0, BF16_MFMA_PEAK_TFLOPS) if split else (1.0, FP32_MFMA_PEAK_TFLOPS)
    if x8:
        peak = 3.0 / (1.0 / BF16_MFMA_PEAK_TFLOPS + 2.0 / FP8_MFMA_PEAK_TFLOPS)
    ach = eq * mult
    r = {
        'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': ach, 'peak': peak,
        'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
        'frac_basis': ('executed matrix instructions: fp16 hi*hi + both cross terms on the 8-bit instruction; peak = that mix (3750)'
                       if x8 else 'executed matrix instructions: 3 fp16 partial products per fp32 multiply'
                       if split else 'fp32 matrix instructions = the reference FLOP count'),
        'frac_useful': eq / (BF16_MFMA_PEAK_TFLOPS if split else peak),
        'fp32_equivalent_tflops': eq, 'frac_of_fp32_mfma_peak': eq / FP32_MFMA_PEAK_TFLOPS,
        'launches_sampled': dom['n'], 'avg_launch_us': 1000.0 * dom['ms'] / dom['n'],
        'avg_launch_us_basis': ('HIP events that receive the START and END of the kernel itself (hipExtLaunchKernelGGL, '
                                'every 37th launch of the timed region): the duration rocprofv3\'s kernel trace reports '
                                '(avg_launch_us_rocprof, from the committed profile of the same command).  '
                                'avg_stream_interval_us: events recorded on the stream around '
                                'OTHER sampled launches, i.e. from the end of the previous kernel: kernel + dependent-launch boundary'
                                if dom.get('kernel_timed') else
                                'HIP events recorded on the launch stream around every 37th launch: kernel time + the '
                                'dependent-launch boundary'),
        'avg_stream_interval_us': (1000.0 * dom['ms_stream'] / dom['n_stream'] if dom.get('n_stream') else None),
        'frac_incl_launch_boundary': (mult * dom['flops_stream'] / (dom['ms_stream'] * 1e-3) / 1e12 / peak
                                      if dom.get('n_stream') else None),
        'flop_per_launch': mult * dom['flops'] / dom['n'],
        'all_gemm_kernels': {k: {'TFLOP/s': v['flops'] / (v['ms'] * 1e-3) / 1e12, 'n': v['n'],
                                 'avg_us': 1000.0 * v['ms'] / v['n']} for k, v in prof.items()},
    }
    if dom.get('n_probe'):
        # phase stamps of the sampled launches (t2h_gemm_split_probe_next_launch): the clock the CUs really ran at
        ghz = dom['loop_ghz'] / dom['n_probe']
        r.update(main_loop_us=dom['loop_us'] / dom['n_probe'], main_loop_shader_clock_ghz=ghz,
                 peak_at_measured_clock=peak * ghz / MAX_CLOCK_GHZ, frac_of_peak_at_measured_clock=ach / (peak * ghz / MAX_CLOCK_GHZ),
                 clock_note=('s_memtime / s_memrealtime stamps around the main loop of the sampled launches, median over the '
                             'workgroups: the chip clocks to its power budget, and with the matrix pipes and the LDS-DMA stream '
                             f'both busy the CUs do not hold {MAX_CLOCK_GHZ} GHz (the {peak:.0f} TFLOP/s peak assumes they do); '
                             'profiles/r04_gemm_one_wave_per_simd_and_clock.log'))
    if split and dom.get('by_cfg'):
        # one entry per tile configuration of the dispatcher (= template instantiation of gemm_split_kernel)
        inst = {}
        for k, v in dom['by_cfg'].items():
            if not v['n']:
                continue
            e = {'n': v['n'], 'avg_us': 1000.0 * v['ms'] / v['n'], 'executed_TFLOP/s': mult * v['flops'] / (v['ms'] * 1e-3) / 1e12,
                 'frac': mult * v['flops'] / (v['ms'] * 1e-3) / 1e12 / peak}
            if v['n_stream']:
                e['avg_stream_interval_us'] = 1000.0 * v['ms_stream'] / v['n_stream']
            if v['n_probe']:
                g = v['loop_ghz'] / v['n_probe']
                e.update(main_loop_us=v['loop_us'] / v['n_probe'], main_loop_shader_clock_ghz=g,
                         frac_of_peak_at_measured_clock=e['frac'] * MAX_CLOCK_GHZ / g)
            inst[k] = e
        r['all_gemm_kernels'].update(inst)
    side = profile_side_data(dom['kernel'], config)
    r.update({k: v for k, v in side.items() if k != 'per_kernel'})
    if split and dom.get('by_cfg'):
        r['worst_instantiation'] = worst_instantiation(inst, side, x8)
    if r.get('avg_launch_us_rocprof'):
        r['frac_rocprof_kernel_time'] = r['frac'] * r['avg_launch_us'] / r['avg_launch_us_rocprof']
        r['rocprof_vs_live_kernel_time'] = r['avg_launch_us_rocprof'] / r['avg_launch_us']
    return r


class ConfigRun:
    """One BASELINE.json configuration on this rank: model + this rank's shard of the global batch."""

    def __init__(self, config, model, batch, sample_steps, set_seed, stub=False):
        self.config, self.model, self.batch = config, model, batch
        self.sample_steps, self.set_seed, self.stub = sample_steps, set_seed, stub
        self.upscale = config == 'hires'

    @property
    def x8(self):
        """the sampler's Linears run on x8 operands (fp16 hi*hi + both cross terms in one 8-bit instruction)"""
        net = getattr(self.model, 'sampler_fn', None)
        return bool(getattr(net, 'x8', False) and getattr(net, 'split', False) and getattr(net, '_x8', None) is not None)

    def step(self, events=None):
        """events: list that receives (stage name, start event, end event)."""
        model, batch = self.model, self.batch

        def mark(name, fn):
            if events is None or self.stub:
                return fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            events.append((name, e0, e1))
            return r

        self.set_seed(2021)
        if self.config == 'pose':
            def front():
                model.feed_data(batch)
                model.generate_parsing_map()
            mark('pose_front_end', front)

            def tok():
                model.generate_quantized_segm()
                model.generate_texture_map()
            mark('tokenizer', tok)
        else:
            mark('tokenizer', lambda: model.feed_data(batch))
        top = mark('sampler', lambda: model.sample_fn(temp=1, sample_steps=self.sample_steps))
        _, u8 = mark('refine_decode', lambda: model.decode_indices(top, want_u8=True, upscale=self.upscale))
        return top, u8

    def timed(self, steps, warmup, dworld, dev):
        """`warmup` untimed steps, then exactly `steps` steps of the DEFAULT product path (every sampling round
        one hipGraph replay; no per-launch instrumentation) between barrier + synchronize on both sides;
        returns the max-over-ranks time and this rank's stage samples."""
        from text2human_amd import shard
        sync = (lambda: None) if self.stub else torch.cuda.synchronize
        for _ in range(warmup):
            self.step()
        shard.barrier(dworld)
        sync()
        events = []
        marks = []  # one event per step boundary (no sync inside the timed region): the spread of the steps
        t0 = time.perf_counter()
        for _ in range(steps):
            if not self.stub:
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            top, u8 = self.step(events)
        if not self.stub:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
        sync()
        my_elapsed = time.perf_counter() - t0
        shard.barrier(dworld)
        elapsed = shard.max_over_ranks(time.perf_counter() - t0, dworld, dev)
        stage_ms = {}
        for name, e0, e1 in events:
            stage_ms[name] = stage_ms.get(name, 0.0) + e0.elapsed_time(e1) / steps
        net = getattr(self.model, 'sampler_fn', None)
        step_ms = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        return dict(elapsed=elapsed, my_elapsed=my_elapsed, top=top, u8=u8, stage_ms=stage_ms,
                    step_ms_median=(step_ms[len(step_ms) // 2] if step_ms else None),
                    step_ms_min_max=([step_ms[0], step_ms[-1]] if step_ms else None),
                    stats=getattr(net, 'last_stats', None), launch_mode=getattr(net, 'last_launch_mode', None))

    def eager_profile(self, steps):
        """The SAME step as individual launches from the host thread (T2H_GRAPH=0) with the HIP-event sampling of
        the GEMM launches armed (every 37th launch: events that receive the kernel's own start / end, and stream
        intervals on other launches): the source of the `roofline` object, kept OUT of the headline's timed
        region.  No collective in here (every rank may run it on its own).  -> dict(ms_per_step, prof, top, u8,
        host_calls_per_round)."""
        from text2human_amd import _lib, ops
        old = os.environ.get('T2H_GRAPH')
        os.environ['T2H_GRAPH'] = '0'
        try:
            self.step()  # (buffers / caches of the eager path)
            torch.cuda.synchronize()
            ops.gemm_profile_start(every=37)
            t0 = time.perf_counter()
            calls = []
            for _ in range(steps):
                self.set_seed(2021)
                c0 = _lib.n_calls
                top, u8 = self.step()
                calls.append(_lib.n_calls - c0)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            prof = ops.gemm_profile_stop()
        finally:
            if old is None:
                os.environ.pop('T2H_GRAPH', None)
            else:
                os.environ['T2H_GRAPH'] = old
        stats = getattr(getattr(self.model, 'sampler_fn', None), 'last_stats', None) or {}
        return dict(ms_per_step=1000.0 * dt, prof=prof, top=top, u8=u8,
                    host_calls_per_step=calls[-1], rounds=stats.get('rounds'))


def stage_view(stage_ms, b, sample_steps, upscale, stats, x8=False):
    """Per-stage times (HIP events on the launch stream) with the algorithmic rates of SURVEY.md 8(d)."""
    st = {k: {'ms_per_step': v} for k, v in stage_ms.items()}
    if 'sampler' in st:
        fl = GFLOP_IMAGE['sampler_step'] * sample_steps * b * 1e9
        t = stage_ms['sampler'] * 1e-3
        st['sampler'].update(tflops_fp32_equivalent=fl / t / 1e12,
                             frac_useful_of_16bit_peak=fl / t / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                             note='reference FLOP count = every (sample, step) pair evaluated, as the reference does')
        if stats:
            # the work actually launched: one transformer evaluation per sample and ROUND
            ex = GFLOP_SAMPLER_LAYERS * stats['sample_steps_launched'] * 1e9
            st['sampler'].update(
                sample_steps_possible=stats['sample_steps_possible'], sample_steps_needed=stats['sample_steps_needed'],
                sample_steps_evaluated=stats['sample_steps_launched'], rounds=stats['rounds'],
                ms_per_round=stage_ms['sampler'] / max(1, stats['rounds']),
                **sampler_matrix_time_frac(stats['sample_steps_launched'], t, x8),
                executed_products_frac_of_16bit_peak=3.0 * ex / t / 1e12 / BF16_MFMA_PEAK_TFLOPS,
                executed_note='matrix_time_frac: the time the matrix instructions of the evaluations launched need at '
                              'their dense rates over the stage time -- Linears 2 fp16-time units per multiply on x8 '
                              'operands (3 on fp16 planes), attention 3 (bench.sampler_matrix_time_frac).  '
                              'executed_products_frac_of_16bit_peak counts three PRODUCTS per multiply at the fp16 '
                              'rate whatever instruction ran them (the figure quoted until round 5; with x8 it '
                              'overstates the pipe time by 3 / 2 on the Linears).  A (sample, step) pair that changes '
                              'no token is not evaluated, and of the 18 head projections only the sampled rows are '
                              'computed (90.19 GFLOP fp32-equivalent per evaluation)')
    if 'refine_decode' in st:
        t = stage_ms['refine_decode'] * 1e-3
        fl = ((GFLOP_IMAGE['decode_hires'] if upscale else GFLOP_IMAGE['decode']) + GFLOP_IMAGE['refine']) * b * 1e9
        by = DECODE_BYTES_IMAGE['hires' if upscale else 'parsing'] * b + DECODE_WEIGHT_BYTES
        split_conv = os.environ.get('T2H_SPLIT_CONV', '1') != '0'
        st['refine_decode'].update(
            ms_per_image=1e3 * t / b, tflops=fl / t / 1e12,
            compute_frac_of_fp32_mfma_peak=fl / t / 1e12 / FP32_MFMA_PEAK_TFLOPS,
            compute_frac_of_16bit_mfma_peak=(3.0 if split_conv else 1.0) * fl / t / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            algorithmic_hbm_bytes=by, hbm_frac=by / t / (HBM_PEAK_TBS * 1e12),
            convs=('2xfp16-split MFMA, three products: executed = 3 x the reference FLOPs; the large levels by '
                   't2h_conv_halo_f32 (GroupNorm apply + swish + split folded into the halo staging), the small ones by '
                   't2h_gn_apply_split_f32 + t2h_conv_split_f32' if split_conv else 'exact-fp32 MFMA'),
            note='SURVEY.md 8(d) algorithmic FLOPs / bytes (flash-style attention, fused norms).  The stage '
                 'is matrix-bound: at 100% of the fp32 MFMA peak its HBM fraction would be 6.5%, at 100% of '
                 'the three-product fp16 rate 35%')
    if 'pose_front_end' in st:
        t = stage_ms['pose_front_end'] * 1e-3
        fl = GFLOP_IMAGE['pose'] * b * 1e9
        st['pose_front_end'].update(tflops=fl / t / 1e12, compute_frac_of_fp32_mfma_peak=fl / t / 1e12 / FP32_MFMA_PEAK_TFLOPS)
    return st


def side_config(name, run, steps, warmup, batch_per_gpu, world, dworld, dev, profile=True):
    """A further BASELINE.json configuration timed inside the same driver-run command (1 warm-up + 2
    steps by default): its own value, stages and dominant-kernel roofline."""
    r = run.timed(steps, warmup, dworld, dev)
    wl = WORKLOADS[name]
    out = {'metric': wl['metric'], 'value': batch_per_gpu * world * steps / r['elapsed'], 'unit': 'images/s',
           'ms_per_step': 1000.0 * r['elapsed'] / steps, 'steps': steps, 'warmup': warmup,
           'launch_mode': r['launch_mode'],
           **({'ms_per_step_median': r['step_ms_median'], 'ms_per_step_min_max': r['step_ms_min_max']}
              if r.get('step_ms_median') else {}),
           'config': {'workload': f'{wl["desc"]}, batch={batch_per_gpu}/GPU, {run.sample_steps} sampling steps ({wl["ref"]})',
                      'global_batch': batch_per_gpu * world},
           'stages': stage_view(r['stage_ms'], batch_per_gpu, run.sample_steps, run.upscale, r['stats'], x8=run.x8)}
    if not run.stub and profile:
        e = run.eager_profile(1)
        if e['prof']:
            out['roofline'] = gemm_roofline(e['prof'], name)
        out['eager_launches_ms_per_step'] = e['ms_per_step']
    return out



# --------------------------------------------------------------------------- the contract line


DTYPE_SPLIT = '2xf16-split operands (22 significant bits), f32 accumulate; exact-f32 number: exact_fp32_path'
DTYPE_X8 = 'f16 hi plane + e4m3 cross-term planes, f32 accumulate (hidden err 4e-5, tokens exact); exact-f32: exact_fp32_path'
COMPACT_LIMIT = 6000  # bytes; the driver keeps the last 8 KB of stdout and parses the last line


def _r(x, nd=4):
    """Floats to `nd` significant digits (the line is a record, not a data file); everything else unchanged."""
    if isinstance(x, float):
        return float(f'{x:.{nd}g}')
    return x


def _pick(d, keys, nd=4):
    return {k: _r(d[k], nd) for k in keys if d is not None and k in d and d[k] is not None}


def compact_line(out):
    """The ONE stdout line: the harness contract's keys and nothing wordy.  `out` is the full result dict of main()
    (written to gpurun_out/bench_detail.json and stderr); this keeps metric / value / unit / n_gpus / steps / warmup /
    ms_per_step / dtype / config, `roofline` and `cpu_baseline` reduced to numbers + short labels, one number per
    side measurement, and `other_configs` as {value, ms_per_step, steps, roofline_frac}.  Always < COMPACT_LIMIT bytes
    (tests/test_bench_line.py)."""
    c = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step',
                                    'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data') if k in out}
    c['dtype'] = str(out.get('dtype', ''))[:120]
    cfg = out.get('config', {})
    c['config'] = {'workload': str(cfg.get('workload', ''))[:200], **_pick(cfg, ('global_batch', 'sample_steps')),
                   'parallelism': str(cfg.get('parallelism', ''))[:60]}
    if 'ms_per_step_median' in out:
        c['ms_per_step_median'] = _r(out['ms_per_step_median'], 6)
    rf = out.get('roofline')
    if rf:
        c['roofline'] = {'bound': rf.get('bound'), 'kernel': str(rf.get('kernel', ''))[:60],
                         **_pick(rf, ('achieved', 'peak')), 'unit': rf.get('unit'), **_pick(rf, ('frac',)),
                         'traffic': _r(rf.get('traffic')),
                         **_pick(rf, ('frac_useful', 'avg_launch_us', 'avg_launch_us_rocprof', 'mfma_util_pmc',
                                      'main_loop_shader_clock_ghz', 'frac_of_peak_at_measured_clock',
                                      'launches_sampled', 'flop_per_launch'))}
        wi = rf.get('worst_instantiation')
        if wi:
            c['roofline']['worst_instantiation'] = {'name': str(wi.get('name', ''))[:60],
                                                    **_pick(wi, ('frac', 'mfma_util_pmc', 'share'))}
    cb = out.get('cpu_baseline')
    if cb:
        c['cpu_baseline'] = {'value': _r(cb.get('value')), 'unit': cb.get('unit'), 'cores': cb.get('cores'),
                             'kind': cb.get('kind'), 'sample': str(cb.get('sample', ''))[:240]}
    st = out.get('stages') or {}
    if st:
        c['stages_ms'] = {k: _r(v['ms_per_step']) for k, v in st.items()}
        sm, rd = st.get('sampler', {}), st.get('refine_decode', {})
        c['sampler'] = _pick(sm, ('rounds', 'ms_per_round', 'sample_steps_evaluated', 'matrix_time_frac', 'mfma_util_pmc_gemms'))
        c['decode'] = _pick(rd, ('ms_per_image', 'hbm_frac', 'compute_frac_of_16bit_mfma_peak'))
    if 'exact_fp32_path' in out:
        c['exact_fp32_path'] = _pick(out['exact_fp32_path'], ('value', 'ms_per_step'))
    if 'fp16_planes_path' in out:
        c['fp16_planes_path'] = _pick(out['fp16_planes_path'], ('value', 'tokens_equal', 'images_u8_equal'))
    if 'parity' in out:
        c['parity'] = _pick(out['parity'], ('tokens_equal', 'token_mismatches', 'bot_indices_equal', 'img_max_abs',
                                            'img_u8_max_abs'))
    if 'eager_launches' in out:
        c['eager_launches'] = _pick(out['eager_launches'], ('value', 'tokens_equal', 'images_u8_equal',
                                                            'host_calls_per_round'))
    if 'eager_gpu_baseline' in out:
        c['eager_gpu_baseline'] = {'kind': ('reference' if 'unmodified reference' in str(out['eager_gpu_baseline'].get('kind'))
                                            else 'port'),
                                   **_pick(out['eager_gpu_baseline'], ('sampler_ms_per_step',
                                                                       'this_package_sampler_ms_per_step', 'speedup'))}
    if out.get('launch_mode'):
        c['launch_mode'] = str(out['launch_mode']).split(':')[0][:40]
    for k in ('host_launches_per_round', 'rccl_world', 'dist_backend', 'path_tflops'):
        if out.get(k) is not None:
            c[k] = _r(out[k])
    for k in ('per_rank_ms_per_step', 'per_rank_image_checksum'):
        if out.get(k) is not None:
            c[k] = [_r(v, 10 if 'checksum' in k else 6) for v in out[k]]
    oc = out.get('other_configs')
    if oc:
        c['other_configs'] = {}
        for name, leg in oc.items():
            e = _pick(leg, ('value', 'ms_per_step', 'ms_per_step_median', 'steps'))
            e['global_batch'] = leg.get('config', {}).get('global_batch')
            if leg.get('launch_mode'):
                e['launch_mode'] = leg['launch_mode']
            if leg.get('roofline'):
                e['roofline_frac'] = _r(leg['roofline'].get('frac'))
            lst = leg.get('stages') or {}
            if 'refine_decode' in lst:
                e['decode_ms_per_image'] = _r(lst['refine_decode'].get('ms_per_image'))
            c['other_configs'][name] = e
    if out.get('detail'):
        c['detail'] = out['detail']
    line = json.dumps(c, separators=(',', ':'))
    if len(line) >= COMPACT_LIMIT:  # cannot happen with the fields above; never let the contract line grow again
        for k in ('per_rank_image_checksum', 'per_rank_ms_per_step', 'stages_ms', 'eager_launches', 'sampler', 'decode'):
            c.pop(k, None)
        line = json.dumps(c, separators=(',', ':'))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out):
    """Full result -> gpurun_out/bench_detail.json (+ stderr); the compact contract line -> stdout (the LAST line)."""
    try:
        d = os.path.join(ROOT, 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        name = 'bench_detail.json' if out.get('n_gpus', 1) == 1 else f'bench_detail_n{out["n_gpus"]}.json'
        with open(os.path.join(d, name), 'w') as f:
            json.dump(out, f, indent=1)
        out['detail'] = f'gpurun_out/{name}'
    except OSError:
        pass
    print('bench detail: ' + json.dumps(out), file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


# --------------------------------------------------------------------------- main


def main(argv=None):
    args = parse_args(argv)
    if args.cpu_baseline_worker:
        print(json.dumps(cpu_baseline_worker(args.sample_steps, args.cpu_sampler_steps,
                                             args.cpu_threads or (os.cpu_count() or 1), args.cpu_repeats)),
              flush=True)
        return
    launched = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ  # under torch.distributed.run (or T2H_FORCE_DIST's env)
    if not launched and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU) of the same
        # command -- the line printed is rank 0's, with n_gpus = rccl_world = N
        return spawn_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if launched and world != args.gpus and os.environ.get('T2H_FORCE_DIST') != '1':
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; '
                         f'launch with --nproc-per-node {args.gpus} (or run plain `python bench.py --gpus {args.gpus}`, '
                         'which spawns its own ranks)')
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
    stub = args.stub_model
    if stub:
        dev, backend = torch.device('cpu'), 'gloo'
    else:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU path exists)'
        torch.cuda.set_device(local_rank)
        dev, backend = torch.device('cuda', local_rank), 'nccl'
    pinned = pin_launch_thread(local_rank, local_world)
    use_dist = world > 1 or os.environ.get('T2H_FORCE_DIST') == '1'  # test hook: RCCL init with 1 rank
    dist = init_dist(backend, dev) if use_dist else None
    dworld = 2 if use_dist else 1  # "is distributed" switch of the shard helpers

    from text2human_amd import shard
    wl = WORKLOADS[args.config]
    batch_per_gpu = args.batch or wl['batch']

    # ---- weights: synthesised ONCE (rank 0) and broadcast, not 8x on the host cores
    t_w0 = time.perf_counter()
    pose = args.config == 'pose'
    if stub:
        sds = shard.broadcast_state_dicts({'sampler': {'w': torch.arange(6.0).view(2, 3)}} if rank == 0 else None,
                                          dworld, dev)
        model = StubModel(sds)
        set_seed = torch.manual_seed
        opt = None
    else:
        from text2human_amd import defaults, ops, options, synthetic
        from text2human_amd.models import SampleFromParsingModel, SampleFromPoseModel
        opt = options.dict_to_nonedict(defaults.sample_from_pose() if pose else defaults.sample_from_parsing())
        opt['sample_steps'] = args.sample_steps
        sds = synthetic.make_state_dicts(opt, seed=1234) if rank == 0 else None
        sds = shard.broadcast_state_dicts(sds, dworld, dev)
        model = (SampleFromPoseModel if pose else SampleFromParsingModel)(opt, state_dicts=sds)
        set_seed = options.set_random_seed
    t_weights = time.perf_counter() - t_w0

    def shard_of(config, per_gpu):
        """this rank's contiguous slice of the global batch; seed 2021 on every rank's OWN
        shard (the oracle of a sharded run is the reference on that shard)"""
        lo, hi = shard.shard_range(per_gpu * world, rank, world)
        if stub:
            g = torch.Generator().manual_seed(2021)
            full = dict(segm=torch.rand(per_gpu * world, 1, 8, 4, generator=g))
        elif config == 'pose':
            full = synthetic.pose_batch(per_gpu * world, seed=2021)
        else:
            full = synthetic.parsing_batch(per_gpu * world, seed=2021)
        return {k: (v[lo:hi].to(dev) if torch.is_tensor(v) else v[lo:hi]) for k, v in full.items()}, hi - lo

    batch, n_mine = shard_of(args.config, batch_per_gpu)
    run = ConfigRun(args.config, model, batch, args.sample_steps, set_seed, stub)
    res = run.timed(args.steps, args.warmup, dworld, dev)
    elapsed, top, u8 = res['elapsed'], res['top'], res['u8']
    hw = (1024, 512) if run.upscale else (512, 256)
    assert tuple(u8.shape) == (n_mine, hw[0], hw[1], 3), tuple(u8.shape)
    per_rank_ms = shard.gather_floats(1000.0 * res['my_elapsed'] / args.steps, dworld, dev)
    # ---- the same step as individual launches, with the per-launch event sampling armed (outside the headline's
    # timed region, no collective: every rank runs it on its own)
    eager = None
    if not stub and not args.no_eager_leg:
        eager = run.eager_profile(max(1, args.eager_steps))
    # a checksum of every rank's images reaches rank 0 (the optional image gather of SURVEY 8(e))
    sums = shard.gather_floats(float(u8.to(torch.float64).sum()), dworld, dev)

    # ---- the other BASELINE.json configurations, inside the same (driver-run) command
    other = {}
    if args.config == 'parsing' and not args.no_other_configs and (stub or not args.batch):
        # configs[3]'s per-GPU share (batch 256 over 8 GPUs = 32 per GPU) at every N: at N = 8 its global
        # batch IS configs[3]; the headline line above keeps configs[1]'s 8 per GPU at every N (weak scaling)
        # (--stub-model: the same leg with a small batch, so that its collectives run under gloo in the tests)
        per32 = 2 * batch_per_gpu if stub else 32
        b32, _ = shard_of('parsing', per32)
        other['parsing_b32'] = side_config('parsing_b32', ConfigRun('parsing', model, b32, args.sample_steps, set_seed, stub),
                                           args.other_steps, 1, per32, world, dworld, dev)
        if world == 1 and not stub:
            other['hires'] = side_config('hires', ConfigRun('hires', model, batch, args.sample_steps, set_seed),
                                         args.other_steps, 1, batch_per_gpu, 1, 1, dev)
            # sample_from_pose: the same five checkpoints + the parsing generator's three modules
            popt = options.dict_to_nonedict(defaults.sample_from_pose())
            popt['sample_steps'] = args.sample_steps
            schemas = synthetic.module_schemas(popt)
            psds = dict(sds)
            for name in ('shape_embedder', 'shape_encoder', 'shape_decoder'):
                psds[name] = synthetic.fill(schemas[name], 1234 * 1000 + synthetic._SEEDS[name])
            pmodel = SampleFromPoseModel(popt, state_dicts=psds)
            pb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic.pose_batch(32, seed=2021).items()}
            other['pose'] = side_config('pose', ConfigRun('pose', pmodel, pb, args.sample_steps, set_seed),
                                        args.other_steps, 1, 32, 1, 1, dev)
            del pmodel, pb
            torch.cuda.empty_cache()

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    n_img = batch_per_gpu * world * args.steps
    split_on = os.environ.get('T2H_SPLIT_GEMM', '1') != '0'
    out = {
        'metric': wl['metric'],
        'value': n_img / elapsed,
        'unit': 'images/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': 1000.0 * elapsed / args.steps,
        **({'ms_per_step_median': res['step_ms_median'], 'ms_per_step_min_max': res['step_ms_min_max']}
           if res.get('step_ms_median') else {}),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': ('stub' if stub else 'f32' if not split_on else
                  DTYPE_X8 if getattr(getattr(model, 'sampler_fn', None), 'x8', False) else DTYPE_SPLIT),
        'dtype_detail': ('sampler Linears + attention and the decoder convolutions as 3 fp16 partial products per multiply '
                         'on v_mfma_f32_32x32x16_f16 -- fp32-class accuracy, see "parity" and "exact_fp32_path" for the '
                         'strictly-fp32 number; tokenizer, index-prediction UNet, parsing generator exact-f32 MFMA; GELU '
                         'by a 3-ulp rational erf, softmax in the base-2 domain'),
        'data': 'synthetic',
        'config': {
            'workload': (f'{wl["desc"]}, batch={batch_per_gpu}/GPU, {args.sample_steps} sampling steps '
                         f'({wl["ref"]})'),
            'global_batch': batch_per_gpu * world,
            'sample_steps': args.sample_steps,
            'weights': 'synthetic seed 1234 (reference .pth layout), built on rank 0 and broadcast',
            'rng': 'torch global generator (reference contract), seed 2021 per rank shard',
            'parallelism': f'batch shard x{world}, no data-path collective',
        },
        'rccl_world': (dist.get_world_size() if use_dist else 1),
        'dist_backend': (backend if use_dist else None),
        'per_rank_ms_per_step': per_rank_ms,
        'per_rank_image_checksum': sums,
        'weights_s': t_weights,
        'launch_thread_cores': pinned,
    }
    if stub:
        if other:
            out['other_configs'] = other
        emit(out)
        dist and dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel = the GEMM instantiation with the largest sampled time
    out['launch_mode'] = {
        'graph': 'hipGraph replay: ONE host launch per sampling round (engine.RoundGraph, the product default)',
        'eager': 'individual launches from the host thread'}.get(res['launch_mode'], res['launch_mode'])
    if eager is not None:
        if eager['prof']:
            out['roofline'] = gemm_roofline(eager['prof'], args.config)
            out['roofline']['measured_in'] = ('the eager_launches leg of this command (same step, same kernels, individual '
                                              'launches): per-launch events cannot be attached to the nodes of a replayed graph')
        rounds = eager['rounds'] or 1
        out['eager_launches'] = {
            'value': batch_per_gpu * world / (eager['ms_per_step'] * 1e-3), 'unit': 'images/s (this rank x world)',
            'ms_per_step': eager['ms_per_step'],
            'tokens_equal': bool(torch.equal(torch.stack(eager['top']), torch.stack(top))),
            'images_u8_equal': bool(torch.equal(eager['u8'], u8)),
            'host_calls_per_step': eager['host_calls_per_step'],
            'host_calls_per_round': eager['host_calls_per_step'] / rounds,
            'note': f'T2H_GRAPH=0 with the GEMM event sampling armed, 1 warm-up + {max(1, args.eager_steps)} steps; the headline '
                    'issues 1 graph launch per round instead'}
        out['host_launches_per_round'] = 1 if res['launch_mode'] == 'graph' else eager['host_calls_per_step'] / rounds
    # ---- stage view (HIP events on the launch stream), incl. decode's compute AND HBM fractions
    if res['stage_ms']:
        out['stages'] = stage_view(res['stage_ms'], batch_per_gpu, args.sample_steps, run.upscale, res['stats'], x8=run.x8)
        if 'matrix_time_frac' in out['stages'].get('sampler', {}):
            out['stages']['sampler']['mfma_util_pmc_gemms'] = (out.get('roofline') or {}).get('mfma_util_pmc')
    # whole-path arithmetic rate on the reference FLOP count
    per_image = (GFLOP_IMAGE['sampler_step'] * args.sample_steps + GFLOP_IMAGE['tokenizer'] + GFLOP_IMAGE['refine']
                 + (GFLOP_IMAGE['decode_hires'] if run.upscale else GFLOP_IMAGE['decode'])
                 + (GFLOP_IMAGE['pose'] if pose else 0.0)) / 1e3
    out['path_tflops'] = per_image * out['value'] / world
    if other:
        out['other_configs'] = other
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args.sample_steps, args.cpu_sampler_steps, args.cpu_repeats)
    if world == 1 and not args.no_exact_fp32 and split_on:
        # the same step with the sampler's Linears / attention AND the decoders' convolutions on the exact-fp32
        # matrix instructions (T2H_SPLIT_GEMM=0 + T2H_SPLIT_CONV=0): timing over >= 3 steps, and the parity of
        # the default path against it -- tokens, bottom indices and the image, each through its own kernels
        from text2human_amd import engine
        fast = model.sampler_fn
        model.sampler_fn = engine.SamplerNet(model.P, model._tf_desc, opt['bert_n_head'], 'tf', split=False)
        model.decoder.use_split = model.bot_decoder_res.use_split = False
        try:
            run.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.exact_steps):
                top_x, u8_x = run.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / args.exact_steps
            img_x, _, int_x = model.decode_indices(top_x, return_inter=True, upscale=run.upscale)
            img_sx, _, _ = model.decode_indices(top, return_inter=True, upscale=run.upscale)  # exact decoder, default tokens
        finally:
            model.sampler_fn = fast
            model.decoder.use_split = model.bot_decoder_res.use_split = True
        out['exact_fp32_path'] = {'value': batch_per_gpu / dt, 'unit': 'images/s', 'ms_per_step': 1000.0 * dt,
                                  'note': 'sampler Linears / attention and decoder convolutions on '
                                          'v_mfma_f32_32x32x2_f32 (bitwise fp32 fma chains); '
                                          f'1 warm-up + {args.exact_steps} timed steps'}
        ts, tx = torch.stack(top), torch.stack(top_x)
        diff = (u8.to(torch.int16) - u8_x.to(torch.int16)).abs()
        n_tok = int((ts != tx).sum())
        img_s, _, int_s = model.decode_indices(top, return_inter=True, upscale=run.upscale)
        n_bot = sum(int((a['bot_lists'] != c['bot_lists']).sum()) for a, c in zip(int_s, int_x))
        out['parity'] = {
            'what': 'default step (split-precision sampler + decoder) vs the exact-fp32 step (both on '
                    f'v_mfma_f32_32x32x2_f32), same batch, same seed, free-running ({args.sample_steps} steps, '
                    f'B={batch_per_gpu}); oracle-side parity on this configuration: tests/test_gpu_bench_parity.py',
            'tokens_equal': n_tok == 0, 'token_mismatches': n_tok, 'tokens': int((ts >= 0).sum()),
            'bot_indices_equal': n_bot == 0, 'bot_index_mismatches': n_bot,
            'img_max_abs': float((img_s - img_x).abs().max()),
            'img_max_abs_same_tokens': float((img_s - img_sx).abs().max()),
            'images_u8_equal': bool(int(diff.max()) == 0), 'img_u8_max_abs': int(diff.max()),
            'img_u8_frac_differing': float((diff != 0).float().mean()),
        }
    if world == 1 and split_on and not args.no_exact_fp32 and getattr(model.sampler_fn, 'x8', False):
        # the same step with the sampler's Linears on the two-fp16-plane operands (22 significant bits, the default
        # until round 5): what the 8-bit cross-term planes buy on THIS box, and that the tokens are the same
        net = model.sampler_fn
        net.x8, net._graphs = False, {}
        try:
            run.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.exact_steps):
                top_p, u8_p = run.step()
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t1) / args.exact_steps
        finally:
            net.x8, net._graphs = True, {}
        out['fp16_planes_path'] = {'value': batch_per_gpu / dtp, 'unit': 'images/s', 'ms_per_step': 1000.0 * dtp,
                                   'tokens_equal': bool(torch.equal(torch.stack(top_p), torch.stack(top))),
                                   'images_u8_equal': bool(torch.equal(u8_p, u8)),
                                   'note': 'T2H_X8=0: three fp16 partial products per multiply (22-bit operands) instead of '
                                           f'fp16 hi*hi + 8-bit cross terms; 1 warm-up + {args.exact_steps} timed steps'}
    if world == 1 and not args.no_eager_gpu_baseline and args.config != 'pose':
        out['eager_gpu_baseline'] = eager_gpu_baseline(model, batch, sds, opt, args.eager_gpu_steps, dev)
    emit(out)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
