"""Parity (-m gpu) on the BASELINE.json configurations that are not the bench
line: configs[2] sample_from_pose end to end, configs[4] the 1024x512 upscaled
hierarchy (SURVEY.md 8(d) interpretation: nearest-x2 of both quantised latents,
then the fully convolutional decoders)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as R
from text2human_amd import defaults, options, synthetic
from text2human_amd.models import SampleFromPoseModel

from parity_util import odev, osds

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def opt():
    return options.dict_to_nonedict(defaults.sample_from_pose())


@pytest.fixture(scope='module')
def sds(opt):
    return synthetic.make_state_dicts(opt, seed=4321)


@pytest.fixture(scope='module')
def model(opt, sds):
    return SampleFromPoseModel(opt, state_dicts=sds)


def test_sample_from_pose_end_to_end(model, sds, opt):
    """densepose + attributes -> parsing -> tokens -> 2 sampling steps -> image,
    stage by stage against the oracle (the oracle's parsing map is injected after
    the parsing stage so that a near-tie argmax cannot cascade)."""
    pb = synthetic.pose_batch(2, seed=8)
    model.feed_data(pb)
    model.generate_parsing_map()
    od = osds(sds)   # (the oracle's convolutional stages run where parity_util.ORACLE_DEV says)
    with torch.no_grad():
        segm_ref, logits = R.parsing_from_pose(odev(pb['densepose']), odev(pb['shape_attr']), od['shape_embedder'],
                                               od['shape_encoder'], od['shape_decoder'],
                                               opt['shape_attr_class_num'])
    segm_ref, logits = segm_ref.cpu(), logits.cpu()
    bad = model.segm.cpu() != segm_ref
    t2 = logits.topk(2, dim=1).values
    margin = (t2[:, 0] - t2[:, 1]).unsqueeze(1)
    assert (margin[bad] < 1e-4).all(), f'{int(bad.sum())} parsing pixels differ beyond a near-tie'
    assert bad.float().mean() < 1e-3
    model.segm = segm_ref.to(DEV)                      # identical upstream for the next stages
    model.generate_quantized_segm()
    model.generate_texture_map()
    with torch.no_grad():
        tok_ref = R.segm_tokens(odev(segm_ref), od['segm_encoder'], od['segm_quant_conv'],
                                od['segm_quantizer']['embedding.weight']).view(2, -1).cpu()
        mask_ref = R.texture_map(segm_ref, pb['upper_fused_attr'], pb['lower_fused_attr'],
                                 pb['outer_fused_attr'])
    assert torch.equal(model.segm_tokens.cpu(), tok_ref)
    assert torch.equal(model.texture_mask.cpu(), mask_ref)
    model.noise = R.SeededNoise(17, 'cpu')
    try:
        top = model.sample_fn(temp=1, sample_steps=2)
    finally:
        model.noise = None
    with torch.no_grad():
        top_ref = R.sample_fn(tok_ref, mask_ref, sds['sampler'], 2, noise=R.SeededNoise(17, 'cpu'))
        img_ref, _ = R.refine_and_decode(odev(top_ref), odev(mask_ref), od)
    assert torch.equal(torch.stack(top).cpu(), torch.stack(top_ref))
    img, _ = model.decode_indices(top)
    assert (img.cpu() - img_ref.cpu()).abs().max().item() < 2e-4


def test_upscaled_hierarchy_1024x512(model, sds):
    g = torch.Generator().manual_seed(31)
    tex = torch.randint(0, 18, (1, 512), generator=g)
    val = torch.randint(0, 1024, (1, 512), generator=g)
    top = [torch.where(tex == h, val, torch.full_like(val, -1)) for h in range(18)]
    mask = tex.view(1, 1, 32, 16).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
    model.texture_mask, model.batch_size = mask.to(DEV), 1
    img, _, inter = model.decode_indices([t.to(DEV) for t in top], return_inter=True, upscale=True)
    assert img.shape == (1, 3, 1024, 512)
    od, top_o, mask_o = osds(sds), odev(top), odev(mask)
    with torch.no_grad():
        pq, bq = od['top_post_quant_conv'], od['bot_post_quant_conv']
        tq = F.conv2d(R.top_codebook_entry(top_o, mask_o, od['top_quantize']), pq['weight'], pq['bias'])
        bot_idx = R.bot_index_prediction(tq, mask_o, od['guidance_encoder'], od['index_decoder'])
        qb = F.conv2d(R.bot_codebook_entry(bot_idx, mask_o, od['bot_quantize']), bq['weight'], bq['bias'])
        up = lambda t: F.interpolate(t, scale_factor=2.0, mode='nearest')
        bh = R.decoder_res(up(qb), od['bot_decoder_res'])
        dec = R.decoder(up(tq), od['decoder'], bot_h=bh)
        ref = ((dec + 1) / 2).clamp(0, 1)
    got_bot = inter[0]['bot_lists'].view(18, 1, 32, 16).cpu()
    assert torch.equal(got_bot, torch.stack(bot_idx).cpu())
    assert (img.cpu() - ref.cpu()).abs().max().item() < 2e-4


# ---------------------------------------------------------------------------------------------
# Parity AT THE TIMED SHAPES of the configurations bench.py reports under "other_configs":
# configs[2] sample_from_pose at batch 32 (M = 16384 rows through every sampler Linear: the
# 256x128 ping-pong tiles for q|k|v / fc1 / fc2, 128x128 for proj -- another dispatch than the
# headline's M = 4096) and configs[4]'s per-GPU share, the 1024x512 decode of a batch of 8.


def test_sample_from_pose_batch_32_teacher_forced(model, sds, opt):
    """B = 32 through the whole pose path (BASELINE.json configs[2], the shape bench.py times) on a fixture with
    non-degenerate parsing maps: parsing maps vs the oracle (every differing pixel must be a near-tie of the two
    best classes); tokenizer: latents within the activation tolerance and every token that differs from the
    oracle's accounted for as a codebook near-tie against the measured latent error of its row
    (parity_util.vq_mismatch_accounting); texture map exact; then ALL 256 sampling steps teacher-forced on the
    oracle's trajectory (each of the 32 x 512 categorical decisions is taken on the oracle's own partially
    unmasked state -- the oracle's sampler as eager PyTorch-ROCm fp32 on this GPU, with the torch device
    generator on both sides), and free-running."""
    from parity_util import (account, balanced_pose_state_dicts, forced_run, oracle_run, seed_all,
                             vq_mismatch_accounting)
    from text2human_amd import ops
    Bp, steps = 32, 256
    # non-degenerate parsing maps (all 24 classes, parity_util.balanced_pose_state_dicts): the plain synthetic
    # weights map every pose to ONE constant class, an ill-conditioned input for the tokenizer's GroupNorms
    sds = balanced_pose_state_dicts(sds, opt)
    model = SampleFromPoseModel(opt, state_dicts=sds)
    pb = synthetic.pose_batch(Bp, seed=2021)
    model.feed_data(pb)
    model.generate_parsing_map()
    dv = lambda d: {k: v.to(DEV) for k, v in d.items()}
    od = osds(sds)
    with torch.no_grad():   # (convolutional stages of the oracle: parity_util.ORACLE_DEV)
        segm_ref, logits = R.parsing_from_pose(odev(pb['densepose']), odev(pb['shape_attr']), od['shape_embedder'],
                                               od['shape_encoder'], od['shape_decoder'],
                                               opt['shape_attr_class_num'])
    segm_ref, logits = segm_ref.cpu(), logits.cpu()
    assert len(torch.unique(segm_ref)) >= 20, 'the fixture must give non-degenerate parsing maps'
    bad = model.segm.cpu() != segm_ref
    t2 = logits.topk(2, dim=1).values
    margin = (t2[:, 0] - t2[:, 1]).unsqueeze(1)
    # every differing pixel is accounted for: the oracle's own margin between its two best classes must be a
    # rounding-level near-tie (the class logits are O(0.1), their spatial variation O(1e-3))
    assert (margin[bad] < 1e-5).all(), f'{int(bad.sum())} parsing pixels differ beyond a near-tie'
    assert bad.float().mean() < 1e-2
    print(f'pose B=32 parsing maps: {int(bad.sum())} of {bad.numel()} pixels differ, all with margin < 1e-5')
    # Tokenizer on the HIP path's own parsing maps (identical input on both sides): every token that differs
    # from the CPU oracle's is accounted for as a codebook near-tie against the measured latent error of its row
    model.generate_quantized_segm()
    model.generate_texture_map()
    segm_cpu = model.segm.cpu()
    with torch.no_grad():
        one_hot = F.one_hot(odev(segm_cpu).squeeze(1).long(), 24).permute(0, 3, 1, 2).float()
        qc = od['segm_quant_conv']
        z_ref = F.conv2d(R.encoder(one_hot, od['segm_encoder']), qc['weight'], qc['bias'])
        z_ref = z_ref.permute(0, 2, 3, 1).reshape(-1, z_ref.shape[1])
        tok_ref = R.vq_l2_argmin(z_ref, od['segm_quantizer']['embedding.weight']).view(Bp, -1).cpu()
        z_ref = z_ref.cpu()
        book = sds['segm_quantizer']['embedding.weight']
        mask_ref = R.texture_map(segm_cpu, pb['upper_fused_attr'], pb['lower_fused_attr'], pb['outer_fused_attr'])
    x = ops.onehot_nhwc(model.segm.to(torch.float32).reshape(-1), 24, model.segm_cin_pad)
    z_hip, _, _ = model.segm_encoder.encode(x, Bp, 512, 256)
    z_hip = ops.gemm(z_hip, model.P['segm.qc.w'], bias=model.P['segm.qc.b'])
    # (the latent is the output of 20 GroupNorm'ed conv layers over a pixel-noisy 24-class map: two exact-fp32
    # evaluations with different summation orders differ by a few 1e-4 there; measured 2.6e-4)
    lat_err = float((z_hip.cpu() - z_ref).abs().max())
    assert lat_err < 6e-4 * max(1.0, float(z_ref.abs().max())), lat_err
    acc = vq_mismatch_accounting(z_hip, z_ref, book, model.segm_tokens, tok_ref)
    unexplained = [a for a in acc if not a['explained']]
    print(f'pose B=32 tokenizer: {len(acc)} of {Bp * 512} tokens differ from the CPU oracle, all near-ties: '
          f'{not unexplained}; latent max abs err {lat_err:.2e}')
    assert not unexplained, f'{len(unexplained)} of {len(acc)} differing tokens are not codebook near-ties: {unexplained[:5]}'
    assert len(acc) <= 16, acc[:5]     # (non-degenerate maps: 0 expected, a handful of ties tolerated and listed)
    assert torch.equal(model.texture_mask.cpu(), mask_ref)

    sd_dev = dv(sds['sampler'])
    ref, trace, rng_state = oracle_run(model.segm_tokens, model.texture_mask, sd_dev, steps, 2021)
    assert model.sampler_fn.split and model.sampler_fn.split_mha
    mism, stats = forced_run(model, trace, steps, 2021, compact=True)
    accs = account(model, sd_dev, model.texture_mask, trace, rng_state, mism, steps)
    bad_s = [a for a in accs if not a['explained']]
    assert not bad_s, f'{len(bad_s)} of {len(mism)} sampler mismatches are not float near-ties: {bad_s[:5]}'
    assert len(mism) == 0, f'{len(mism)} of {Bp * 512} decisions differ from the oracle: {accs[:5]}'
    assert stats['rounds'] <= steps
    # free-running
    seed_all(2021)
    top = torch.stack(model.sample_fn(temp=1, sample_steps=steps))
    assert torch.equal(top, torch.stack(ref))


def test_upscaled_hierarchy_batch_of_8(model, sds):
    """configs[4]'s per-GPU share: 8 images of 1024x512 through refine + decode (chunks of 2 images, the
    split-precision convolutions at 512x256x... pixel counts, spatial attention over 2048 / 8192
    positions).  Bottom indices of all 8 exact; all 8 images against the CPU oracle within the image
    tolerance."""
    Bh = 8
    g = torch.Generator().manual_seed(41)
    tex = torch.randint(0, 18, (Bh, 512), generator=g)
    val = torch.randint(0, 1024, (Bh, 512), generator=g)
    top = [torch.where(tex == h, val, torch.full_like(val, -1)) for h in range(18)]
    mask = tex.view(Bh, 1, 32, 16).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
    model.texture_mask, model.batch_size = mask.to(DEV), Bh
    img, u8, inter = model.decode_indices([t.to(DEV) for t in top], want_u8=True, return_inter=True, upscale=True)
    assert img.shape == (Bh, 3, 1024, 512) and u8.shape == (Bh, 1024, 512, 3)
    got_bot = torch.cat([d['bot_lists'].view(18, -1, 32, 16) for d in inter], 1).cpu()
    up = lambda t: F.interpolate(t, scale_factor=2.0, mode='nearest')
    od, top_o, mask_o = osds(sds), odev(top), odev(mask)
    pq, bq = od['top_post_quant_conv'], od['bot_post_quant_conv']
    with torch.no_grad():
        tq = F.conv2d(R.top_codebook_entry(top_o, mask_o, od['top_quantize']), pq['weight'], pq['bias'])
        bot_idx = R.bot_index_prediction(tq, mask_o, od['guidance_encoder'], od['index_decoder'])
        assert torch.equal(got_bot, torch.stack(bot_idx).cpu())
        qb = F.conv2d(R.bot_codebook_entry(bot_idx, mask_o, od['bot_quantize']), bq['weight'], bq['bias'])
        for i in range(Bh):
            bh = R.decoder_res(up(qb[i:i + 1]), od['bot_decoder_res'])
            dec = R.decoder(up(tq[i:i + 1]), od['decoder'], bot_h=bh)
            ref = ((dec + 1) / 2).clamp(0, 1)
            err = (img[i:i + 1].cpu() - ref.cpu()).abs().max().item()
            assert err < 2e-4, (i, err)
