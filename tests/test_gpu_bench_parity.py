"""Token / index / image parity ON THE BENCHMARKED CONFIGURATION (-m gpu).

bench.py times BASELINE.json configs[1]: batch 8, 256 sampling steps, seed 2021, the
split-precision (2 x fp16, three products) sampler.  These tests run exactly that
workload -- not a short or small-batch stand-in -- against the oracle executed as eager
PyTorch-ROCm fp32 on the same GPU with the same seed (the reference's RNG contract:
`rand([B,512])` per step + one full `[B*512,1024]` `exponential_` per active head on
torch's device generator, models/sample_model.py:286,301-306).

Method.  Sampling is autoregressive, so one differently decided token makes every later
state differ and a plain comparison of the final tokens cannot tell one float near-tie
from a broken kernel.  The HIP sampler is therefore TEACHER-FORCED on the oracle's
trajectory (engine.sample_tokens(step_hook=...)): after every step its tokens are compared
with the oracle's for that step and then overwritten with them, so all 256 transformer
evaluations see exactly the oracle's partially unmasked states and every one of the
B*512 = 4096 categorical decisions is checked on identical inputs.  Every mismatch is
accounted for explicitly: with the step's noise re-drawn from the recorded generator state,
`gap` = log-ratio by which the oracle's own arithmetic prefers its token over ours, and
`dl` = max |logit difference| between the two implementations on that row; a mismatch is
"explained" iff gap <= 2 dl (the two logits that decide the race moved by at most dl each)
AND dl is inside the activation tolerance.  Counts are REPORTED (gpurun_out/ JSON + the
assertion message), never asserted away.

Both schedules of engine.sample_tokens are forced this way: the reference's synchronous steps, and the
product path's default in which every sample advances through its own active steps (a round then
holds samples at different steps; engine.sample_tokens docstring).

Variants: (a) synthetic default weights; (b) the peaked-logits variant SURVEY.md 8(d)
prescribes (`head_list.*` x50, `conv_seg_head_list.*` x50), where decision margins look
like a trained model's and a logit error is NOT drowned by the noise.  Both the default
split-precision path and the exact-fp32 path (T2H_SPLIT_GEMM=0) are checked.
"""
import json
import os

import pytest
import torch

from oracle import torch_ref as R
from text2human_amd import defaults, engine, ops, options, synthetic
from text2human_amd.models import SampleFromParsingModel

from parity_util import ACT_TOL, DEV, account, forced_run, odev, oracle_run, osds, seed_all, vq_mismatch_accounting

pytestmark = pytest.mark.gpu
B, STEPS, SEED = 8, 256, 2021          # bench.py defaults (BASELINE.json configs[1])
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')


def _full_parity(batch_size, peaked, tag, all_paths=True):
    """The whole sample_from_parsing path at `batch_size`, 256 steps, seed 2021 against the oracle:
    tokenizer exact, sampler teacher-forced (every decision accounted for) and free-running, bottom
    indices exact, image within ACT_TOL.  -> report dict (also written under gpurun_out/)."""
    scale = 50.0 if peaked else 1.0
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234, head_scale=scale, argmax_scale=scale)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    batch = synthetic.parsing_batch(batch_size, seed=SEED)
    report = dict(config=f'B={batch_size}, {STEPS} steps, seed {SEED}, head/argmax scale {scale:g}', paths={})

    model = SampleFromParsingModel(opt, state_dicts=sds)  # default: split-precision sampler
    assert model.sampler_fn.split and model.sampler_fn.split_mha
    model.feed_data(batch)
    od = osds(sds)          # (the oracle's convolutional stages: parity_util.ORACLE_DEV)
    with torch.no_grad():   # tokenizer on non-degenerate maps: exact against the oracle at this batch size
        tok_ref = R.segm_tokens(odev(batch['segm']), od['segm_encoder'], od['segm_quant_conv'],
                                od['segm_quantizer']['embedding.weight']).view(batch_size, -1)
    report['segm_token_mismatches'] = int((model.segm_tokens.cpu() != tok_ref.cpu()).sum())
    if report['segm_token_mismatches']:
        # a codebook decision that differs must be a near-tie of THAT row (two exact-fp32 evaluations of 20 convolution
        # layers in different summation orders; parity_util.vq_mismatch_accounting against the measured latent error);
        # the oracle's tokens are then injected so that everything downstream is compared on identical inputs
        # (tools/parity_more_seeds.py: 0 at seeds 2021 / 1 / 99 / 12345, one accounted near-tie at seed 7)
        import torch.nn.functional as F
        with torch.no_grad():
            one_hot = F.one_hot(odev(batch['segm']).squeeze(1).long(), 24).permute(0, 3, 1, 2).float()
            z_ref = F.conv2d(R.encoder(one_hot, od['segm_encoder']), od['segm_quant_conv']['weight'], od['segm_quant_conv']['bias'])
            z_ref = z_ref.permute(0, 2, 3, 1).reshape(-1, z_ref.shape[1]).cpu()
        x = ops.onehot_nhwc(model.segm.to(torch.float32).reshape(-1), 24, model.segm_cin_pad)
        z_hip, _, _ = model.segm_encoder.encode(x, batch_size, 512, 256)
        z_hip = ops.gemm(z_hip, model.P['segm.qc.w'], bias=model.P['segm.qc.b'])
        acc_tok = vq_mismatch_accounting(z_hip, z_ref, sds['segm_quantizer']['embedding.weight'], model.segm_tokens, tok_ref)
        report['segm_token_accounting'] = acc_tok
        assert all(a['explained'] for a in acc_tok), acc_tok
        model.segm_tokens = tok_ref.to(DEV).view_as(model.segm_tokens).contiguous()
    ref, trace, rng_state = oracle_run(model.segm_tokens, batch['texture_mask'], sd_dev, STEPS, SEED)
    ref_t = torch.stack(ref)
    assert (ref_t >= 0).sum().item() == batch_size * 512  # every token sampled exactly once

    exact = engine.SamplerNet(model.P, model._tf_desc, opt['bert_n_head'], 'tf', split=False)
    split = model.sampler_fn
    paths = [('split_2xfp16', split, True)]
    if all_paths:
        paths += [('split_2xfp16_synchronous_steps', split, False), ('exact_fp32', exact, True)]
    for name, net, compact in paths:
        model.sampler_fn = net
        mism, stats = forced_run(model, trace, STEPS, SEED, compact)
        acc = account(model, sd_dev, batch['texture_mask'], trace, rng_state, mism, STEPS, scale)
        report['paths'][name] = dict(decisions=batch_size * 512, mismatches=len(mism), accounted=acc, schedule=stats)
    model.sampler_fn = split
    st = report['paths']['split_2xfp16']['schedule']
    # a (sample, step) pair is evaluated iff it changes a token; ~13.5 % of them change none
    assert st['sample_steps_needed'] < 0.9 * st['sample_steps_possible']
    if all_paths:
        assert st['rounds'] < STEPS
        assert report['paths']['split_2xfp16_synchronous_steps']['schedule']['rounds'] == len(
            [t for t in trace if trace[t]['active']])

    # free-running (not teacher-forced) runs of the product paths: what bench.py times
    free = {}
    for name, net in (('split_2xfp16', split), ('exact_fp32', exact))[:2 if all_paths else 1]:
        model.sampler_fn = net
        seed_all(SEED)
        free[name] = torch.stack(model.sample_fn(temp=1, sample_steps=STEPS))
    model.sampler_fn = split
    report['free_running'] = dict(split_vs_oracle_mismatches=int((free['split_2xfp16'] != ref_t).sum()))
    if all_paths:
        report['free_running'].update(
            split_vs_exact_mismatches=int((free['split_2xfp16'] != free['exact_fp32']).sum()),
            exact_vs_oracle_mismatches=int((free['exact_fp32'] != ref_t).sum()))

    # refine + decode on the oracle's tokens: bottom indices exact, image within tolerance -- every image of the batch
    with torch.no_grad():
        ref_img, inter = R.refine_and_decode(odev(ref), odev(batch['texture_mask']), od)
    img, _, inters = model.decode_indices(ref, want_u8=True, return_inter=True)
    bot = torch.cat([d['bot_lists'].view(18, -1, 32, 16) for d in inters], 1).cpu()
    ref_bot = torch.stack(inter['bot_idx']).view(18, batch_size, 32, 16).cpu()
    report['decode'] = dict(bot_index_mismatches=int((bot != ref_bot).sum()), bot_indices=int((ref_bot >= 0).sum()),
                            img_max_abs_err=float((img.cpu() - ref_img.cpu()).abs().max()), images=batch_size)

    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, f'parity_{tag}.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))

    assert report['segm_token_mismatches'] <= 4, report['segm_token_mismatches']   # (each one accounted above)
    for name, r in report['paths'].items():
        unexplained = [a for a in r['accounted'] if not a['explained']]
        assert not unexplained, f'{name}: {len(unexplained)} of {r["mismatches"]} mismatches are not float near-ties: {unexplained[:5]}'
    assert report['decode']['bot_index_mismatches'] == 0, report['decode']
    assert report['decode']['bot_indices'] == batch_size * 512, report['decode']
    assert report['decode']['img_max_abs_err'] < ACT_TOL, report['decode']
    return report


@pytest.mark.parametrize('peaked', [False, True], ids=['default_weights', 'peaked_logits_x50'])
def test_bench_config_parity(peaked):
    report = _full_parity(B, peaked, 'bench_config_' + ('peaked' if peaked else 'default'))
    assert report['segm_token_mismatches'] == 0, report   # (the benchmarked batch, seed 2021: exact)
    if not peaked:
        # near-uniform logits: the race is decided by the noise, no near-tie is expected at all
        assert report['paths']['split_2xfp16']['mismatches'] == 0, report['paths']['split_2xfp16']
        assert report['free_running']['split_vs_oracle_mismatches'] == 0, report['free_running']
        assert report['free_running']['split_vs_exact_mismatches'] == 0, report['free_running']


def test_parsing_batch_32_full_parity():
    """BASELINE.json configs[3]'s per-GPU share (bench.py other_configs.parsing_b32): 32 parsing maps, ALL 256
    sampling steps -- M = 16384 rows through every sampler Linear (the 256x128 / 128x128 dispatch, not the
    headline's), attention over 32 x 8 heads, up to 256 changed rows per round in the sampling tail, 4 decode
    chunks.  Teacher-forced 0 unexplained of 16384 decisions; free-running tokens == the eager-GPU oracle's;
    bottom indices 16384 / 16384; all 32 images within tolerance."""
    report = _full_parity(32, False, 'parsing_b32', all_paths=False)
    assert report['segm_token_mismatches'] == 0, report
    assert report['paths']['split_2xfp16']['mismatches'] == 0, report['paths']['split_2xfp16']
    assert report['free_running']['split_vs_oracle_mismatches'] == 0, report['free_running']


def outlier_state_dicts(opt, seed=1234):
    """A 'trained-like' fixture: the synthetic checkpoints' activations are near-Gaussian, a trained
    transformer's residual stream is not.  Per layer: three channels of each LayerNorm gain x20..50 (outlier channels
    in every Linear's input), two rows of `mlp.2` x10 (channels of the residual stream that receive massive updates),
    plus the x50 heads / argmax heads of the peaked-logits variant (decision margins like a trained model's)."""
    sds = synthetic.make_state_dicts(opt, seed=seed, head_scale=50.0, argmax_scale=50.0)
    sd = sds['sampler']
    g = torch.Generator().manual_seed(99)
    n_layers = len([k for k in sd if k.endswith('.ln1.weight')])
    for i in range(n_layers):
        for ln in ('ln1', 'ln2'):
            ch = torch.randperm(512, generator=g)[:3]
            w = sd[f'blocks.{i}.{ln}.weight'].clone()
            w[ch] *= torch.empty(3).uniform_(20.0, 50.0, generator=g)
            sd[f'blocks.{i}.{ln}.weight'] = w
        rows = torch.randperm(512, generator=g)[:2]
        w = sd[f'blocks.{i}.mlp.2.weight'].clone()
        w[rows] *= 10.0
        sd[f'blocks.{i}.mlp.2.weight'] = w
    return sds


def test_outlier_channels():
    """x8 operands (per-tensor static scales, 4-bit mantissas in the cross terms) and the fp16 planes on heavy-tailed
    activation statistics, at the bench shape (B = 8, 256 steps, seed 2021): hidden-state error of both against the
    oracle, every categorical decision teacher-forced on the oracle's trajectory and accounted for, the free-running
    tokens, and the x8 range fall-back fired on purpose on this fixture.  Counts -> gpurun_out/parity_outliers.json
    (committed as profiles/r06_parity_outliers.json)."""
    import torch.nn.functional as F
    import warnings
    from text2human_amd.models import sample_model as SM
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = outlier_state_dicts(opt)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    batch = synthetic.parsing_batch(B, seed=SEED)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    x8net = model.sampler_fn
    assert x8net.x8 and x8net._x8 is not None
    planes = engine.SamplerNet(model.P, model._tf_desc, opt['bert_n_head'], 'tf', split=True, x8=False)
    model.feed_data(batch)
    report = dict(config=f'B={B}, {STEPS} steps, seed {SEED}; per layer 3 LayerNorm gains x20..50 (both norms), 2 mlp.2 rows '
                         'x10; heads / argmax heads x50', paths={})
    # the statistics really are heavy-tailed: per-tensor maxima of the calibration against a plain fixture's
    mx = x8net._x8['act_max']
    report['calibration_max'] = {r: [round(min(mx[i, r] for i in range(24)), 2), round(max(mx[i, r] for i in range(24)), 2)]
                                 for r in ('h1', 'y', 'h2', 'u')}
    assert report['calibration_max']['h1'][1] > 40.0   # (plain synthetic weights: ~5)

    # ---- hidden state of one evaluation (a half-unmasked state of the oracle's trajectory comes later; here a random one)
    gen = torch.Generator().manual_seed(14)
    idx = torch.randint(0, 18433, (2, 512), generator=gen).to(DEV)
    tex = model._texture_tokens(model.texture_mask)[:2]
    seg = model.segm_tokens[:2].contiguous()
    with torch.no_grad():
        ref_h = R.transformer_hidden(idx, seg, tex, sd_dev)   # after ln_f, eager PyTorch-ROCm fp32
    lnf = lambda t: F.layer_norm(t.view(2, 512, 512), (512, ), sd_dev['ln_f.weight'], sd_dev['ln_f.bias'], 1e-5)
    e8 = float((lnf(x8net.hidden(idx, seg, tex).clone()) - ref_h).abs().max())
    e1 = float((lnf(planes.hidden(idx, seg, tex).clone()) - ref_h).abs().max())
    report['hidden_max_abs_err'] = dict(x8=e8, fp16_planes=e1, ref_max_abs=float(ref_h.abs().max()))

    # ---- every decision, teacher-forced
    ref, trace, rng_state = oracle_run(model.segm_tokens, batch['texture_mask'], sd_dev, STEPS, SEED)
    ref_t = torch.stack(ref)
    for name, net in (('x8', x8net), ('fp16_planes', planes)):
        model.sampler_fn = net
        mism, stats = forced_run(model, trace, STEPS, SEED, True)
        acc = account(model, sd_dev, batch['texture_mask'], trace, rng_state, mism, STEPS, 50.0)
        report['paths'][name] = dict(decisions=B * 512, mismatches=len(mism), accounted=acc,
                                     overflow_bits=int(ops.split_overflow_bits(reset=True)))
    model.sampler_fn = x8net

    # ---- free-running, with the fall-back machinery live: did the range bit fire on these statistics by itself?
    SM._warned.discard('index sampler (x8 range)')
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter('always')
        seed_all(SEED)
        free = torch.stack(model.sample_fn(temp=1, sample_steps=STEPS))
    report['free_running'] = dict(x8_vs_oracle_mismatches=int((free != ref_t).sum()),
                                  range_fallback_fired_by_itself=any('fp16-plane' in str(w.message) for w in wlist))
    # ... and fired on purpose (the attention output's scales 64x too large): the call falls back, same tokens as the
    # fp16-plane net free-running, generator where the oracle's stands
    model.sampler_fn = planes
    seed_all(SEED)
    free_planes = torch.stack(model.sample_fn(temp=1, sample_steps=24))
    state = torch.cuda.get_rng_state(DEV)
    model.sampler_fn = x8net
    good = dict(x8net._x8['a'])
    for k in x8net._x8['a']:
        if k[1] == 'y':
            x8net._x8['a'][k] *= 64.0
    x8net._graphs = {}
    SM._warned.discard('index sampler (x8 range)')
    try:
        with pytest.warns(UserWarning, match='fp16-plane'):
            seed_all(SEED)
            forced = torch.stack(model.sample_fn(temp=1, sample_steps=24))
    finally:
        x8net._x8['a'].update(good)
        x8net._graphs = {}
    report['forced_range_fallback'] = dict(tokens_equal_fp16_planes=bool(torch.equal(forced, free_planes)),
                                           generator_equal=bool(torch.equal(torch.cuda.get_rng_state(DEV), state)))

    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, 'parity_outliers.json'), 'w') as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))
    for name, r in report['paths'].items():
        unexplained = [a for a in r['accounted'] if not a['explained']]
        assert not unexplained, f'{name}: {len(unexplained)} of {r["mismatches"]} mismatches are not float near-ties: {unexplained[:5]}'
        assert r['overflow_bits'] == 0, (name, r['overflow_bits'])
    # the hidden state: relative to the tensor's own scale (outlier channels make ln_f's output O(10), not O(1))
    scale = max(1.0, report['hidden_max_abs_err']['ref_max_abs'])
    assert e1 < 1e-5 * scale and e8 < ACT_TOL * scale, report['hidden_max_abs_err']
    assert report['forced_range_fallback'] == dict(tokens_equal_fp16_planes=True, generator_equal=True)
    # free-running: tokens of a run that forks at an accounted near-tie differ from there on; otherwise equal
    if report['paths']['x8']['mismatches'] == 0:
        assert report['free_running']['x8_vs_oracle_mismatches'] == 0, report['free_running']


def test_split_overflow_is_loud():
    """An activation beyond fp16's range must raise -- no inf / NaN planes, no silent fallback."""
    from text2human_amd import ops
    ops.split_overflow(reset=True)
    x = torch.randn(64, 512, device=DEV)
    g, b = torch.ones(512, device=DEV), torch.zeros(512, device=DEV)
    out = ops.split_rows_empty(64, 512, DEV)
    ops.layernorm_split(x, g, b, out)
    assert not ops.split_overflow(reset=True)
    ops.layernorm_split(x, g * 3.0e4, b, out)  # LN output up to ~1e5 > 65504
    assert ops.split_overflow(reset=False) and ops.split_overflow(reset=True)
    assert not ops.split_overflow(reset=True)
    ops.split_rows(x * 1.0e5)
    assert ops.split_overflow(reset=True)

    # end to end: a sampler whose first LayerNorm gain is huge must make sample_fn raise
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    sds['sampler']['blocks.3.ln2.weight'] = sds['sampler']['blocks.3.ln2.weight'] * 1.0e5
    model = SampleFromParsingModel(opt, state_dicts=sds)
    model.feed_data(synthetic.parsing_batch(1, seed=3))
    os.environ['T2H_OVERFLOW_FALLBACK'] = '0'
    try:
        with pytest.raises(engine.SplitOverflowError, match='65504'):
            model.sample_fn(temp=1, sample_steps=2)
    finally:
        os.environ.pop('T2H_OVERFLOW_FALLBACK')
    assert not ops.split_overflow(reset=True)  # the check consumed the flag


def test_split_overflow_falls_back_to_exact_fp32_with_the_reference_tokens():
    """Default behaviour on a checkpoint whose activations leave fp16's range: the call is re-run on the exact-fp32
    kernels from the generator state it started with, so the tokens (and the generator afterwards) are the ones the
    fp32 reference gives (transformer_arch.py:91-99 computes any checkpoint) -- here against the oracle as eager
    PyTorch-ROCm on the same generator.  The split path's captured graphs are not touched."""
    from text2human_amd import ops
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    sds['sampler']['blocks.3.ln2.weight'] = sds['sampler']['blocks.3.ln2.weight'] * 1.0e5
    # (keep the residual stream finite: the block's MLP output is scaled back down)
    sds['sampler']['blocks.3.mlp.0.weight'] = sds['sampler']['blocks.3.mlp.0.weight'] * 1.0e-5
    model = SampleFromParsingModel(opt, state_dicts=sds)
    batch = synthetic.parsing_batch(2, seed=3)
    model.feed_data(batch)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    steps = 12
    seed_all(77)
    with torch.no_grad():
        ref = R.sample_fn(model.segm_tokens, batch['texture_mask'].to(DEV), sd_dev, sample_steps=steps,
                          noise=R.TorchNoise(DEV))
    ref_state = torch.cuda.get_rng_state(DEV)
    seed_all(77)
    with pytest.warns(UserWarning, match='exact-fp32'):
        from text2human_amd.models import sample_model as SM
        SM._warned.discard('index sampler')
        top = model.sample_fn(temp=1, sample_steps=steps)
    assert torch.equal(torch.stack(top), torch.stack(ref))
    assert torch.equal(torch.cuda.get_rng_state(DEV), ref_state)
    assert not ops.split_overflow(reset=True)
    assert model.sampler_fn.split and model._sampler_exact is not None and not model._sampler_exact.split
    # and the decode stage: a decoder whose activations overflow the split rows re-runs on the fp32 convolutions
    sds2 = synthetic.make_state_dicts(opt, seed=1234)
    sds2['decoder']['mid.block_1.norm2.weight'] = sds2['decoder']['mid.block_1.norm2.weight'] * 3.0e5
    sds2['decoder']['mid.block_1.conv2.weight'] = sds2['decoder']['mid.block_1.conv2.weight'] * (1.0 / 3.0e5)
    m2 = SampleFromParsingModel(opt, state_dicts=sds2)
    m2.feed_data(batch)
    seed_all(5)
    top2 = m2.sample_fn(temp=1, sample_steps=3)
    SM._warned.discard('VQGAN refine / decode')
    with pytest.warns(UserWarning, match='exact-fp32'):
        img, _ = m2.decode_indices(top2)
    with torch.no_grad():
        want, _ = R.refine_and_decode(odev(top2), odev(batch['texture_mask']), osds(sds2))
    assert (img.cpu() - want.cpu()).abs().max().item() < 2e-4
    assert m2.decoder.use_split and m2.bot_decoder_res.use_split  # the next call starts on the split path again


@pytest.mark.parametrize('temp', [0.7, 1.3])
def test_sampling_at_a_temperature_other_than_one(temp):
    """`logits / temp` before the categorical draw (models/sample_model.py:300-306; the UI passes other values): the
    HIP sampler teacher-forced on the oracle's trajectory at this temperature, both schedules, and the free-running
    tokens.  Head weights x5: calibrated with the CPU oracle so that the temperature MATTERS (a quarter of the tokens
    differ between temp 1 and 0.7 / 1.3 under one seed; with the default weights the noise decides nearly everything,
    with x50 the logits do) -- a kernel that ignored or mis-applied `temp` fails by hundreds of tokens.  B=4, 96 steps,
    device generator."""
    Bt, steps = 4, 96
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234, head_scale=5.0)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    batch = synthetic.parsing_batch(Bt, seed=11)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    model.feed_data(batch)
    ref, trace, _ = oracle_run(model.segm_tokens, batch['texture_mask'], sd_dev, steps, SEED, temp=temp)
    ref_state = torch.cuda.get_rng_state(DEV)
    ref_t = torch.stack(ref)
    # the temperature matters on this fixture: the same seed at temp 1 gives other tokens
    ref1, _, _ = oracle_run(model.segm_tokens, batch['texture_mask'], sd_dev, steps, SEED, temp=1.0)
    assert (torch.stack(ref1) != ref_t).sum().item() > 0.1 * Bt * 512
    for compact in (True, False):
        mism, _ = forced_run(model, trace, steps, SEED, compact, temp=temp)
        assert len(mism) <= 2, (compact, mism[:5])   # (a float near-tie at most; a wrong temperature gives hundreds)
    seed_all(SEED)
    top = model.sample_fn(temp=temp, sample_steps=steps)
    diff = int((torch.stack(top) != ref_t).sum())
    assert torch.equal(torch.cuda.get_rng_state(DEV), ref_state)
    assert diff == 0, f'{diff} of {Bt * 512} free-running tokens differ at temp {temp}'


def test_bad_texture_id_is_rejected():
    from text2human_amd import _lib
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    model = SampleFromParsingModel(opt, state_dicts=synthetic.make_state_dicts(opt, seed=1234))
    batch = synthetic.parsing_batch(1, seed=3)
    batch['texture_mask'] = batch['texture_mask'].clone()
    batch['texture_mask'][0, 0, :16, :16] = 18.0
    model.feed_data(batch)
    with pytest.raises(_lib.T2HError, match='texture ids'):
        model.sample_fn(temp=1, sample_steps=2)
