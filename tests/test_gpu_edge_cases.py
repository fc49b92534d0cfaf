"""Edge cases of the HIP path (-m gpu): odd batch sizes, decode chunk
boundaries, non-default resolutions (the decoder is fully convolutional),
degenerate schedules and argument validation."""
import pytest
import torch

from oracle import torch_ref as R
from text2human_amd import defaults, ops, options, synthetic
from text2human_amd._lib import T2HError
from text2human_amd.models import SampleFromParsingModel
from text2human_amd.models import sample_model as sm

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def opt():
    return options.dict_to_nonedict(defaults.sample_from_parsing())


@pytest.fixture(scope='module')
def sds(opt):
    return synthetic.make_state_dicts(opt, seed=77)


@pytest.fixture(scope='module')
def model(opt, sds):
    return SampleFromParsingModel(opt, state_dicts=sds)


def _rand_indices(b, seed):
    g = torch.Generator().manual_seed(seed)
    tex = torch.randint(0, 18, (b, 512), generator=g)
    top = torch.full((18, b, 512), -1, dtype=torch.int64)
    val = torch.randint(0, 1024, (b, 512), generator=g)
    for h in range(18):
        top[h][tex == h] = val[tex == h]
    mask = tex.view(b, 1, 32, 16).float().repeat_interleave(16, 2).repeat_interleave(16, 3)
    return top, mask


def test_batch_of_one_matches_oracle(model, sds):
    batch = synthetic.parsing_batch(1, seed=5)
    model.feed_data(batch)
    model.noise = R.SeededNoise(3, 'cpu')
    try:
        top = model.sample_fn(temp=1, sample_steps=2)
    finally:
        model.noise = None
    with torch.no_grad():
        tok = R.segm_tokens(batch['segm'], sds['segm_encoder'], sds['segm_quant_conv'],
                            sds['segm_quantizer']['embedding.weight']).view(1, -1)
        ref = R.sample_fn(tok, batch['texture_mask'], sds['sampler'], 2, noise=R.SeededNoise(3, 'cpu'))
    assert torch.equal(model.segm_tokens.cpu(), tok)
    assert torch.equal(torch.stack(top).cpu(), torch.stack(ref))


def test_decode_chunk_boundary_is_invisible(model):
    """B = 3 with a chunk size of 2: chunked batched decode == per-image decode."""
    top, mask = _rand_indices(3, seed=9)
    model.texture_mask, model.batch_size = mask.to(DEV), 3
    old = sm.DECODE_CHUNK
    try:
        sm.DECODE_CHUNK = 2
        img, u8 = model.decode_indices([t.to(DEV) for t in top], want_u8=True)
    finally:
        sm.DECODE_CHUNK = old
    assert img.shape == (3, 3, 512, 256) and u8.shape == (3, 512, 256, 3)
    for i in range(3):
        model.texture_mask, model.batch_size = mask[i:i + 1].to(DEV), 1
        one, _ = model.decode_indices([t[i:i + 1].to(DEV) for t in top])
        assert torch.equal(one[0], img[i]), f'image {i} depends on its batch neighbours'


def test_decoder_is_resolution_agnostic(model, sds):
    """16x8 latents (256x128 image): same kernels, oracle is the same functional decoder."""
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 256, 16, 8, generator=g) * 0.05
    zb = torch.randn(2, 256, 32, 16, generator=g) * 0.05
    with torch.no_grad():
        bh = R.decoder_res(zb, sds['bot_decoder_res'])
        ref = R.decoder(z, sds['decoder'], bot_h=bh)
    bot_h = model.bot_decoder_res.decode_res(ops.nchw_to_nhwc(zb.to(DEV)), 2, 32, 16)
    dec, ho, wo = model.decoder.decode(ops.nchw_to_nhwc(z.to(DEV)), 2, 16, 8, bot_h=bot_h)
    assert (ho, wo) == (256, 128)
    got = ops.nhwc_to_nchw(dec, 2, 256, 128).cpu()
    assert (got - ref).abs().max().item() < 2e-4


def test_single_step_schedule_unmasks_everything(model):
    """sample_steps = 1: t = 1 -> rand < 1 is always true, every token is drawn once."""
    batch = synthetic.parsing_batch(2, seed=21)
    model.feed_data(batch)
    top = torch.stack(model.sample_fn(temp=1, sample_steps=1))
    tex = model._texture_tokens(model.texture_mask)
    chosen = top.gather(0, tex.unsqueeze(0)).squeeze(0)
    assert (chosen >= 0).all() and (chosen < 1024).all()
    assert ((top >= 0).sum(0) == 1).all()          # exactly one head owns each token


def test_all_background_mask_uses_head_zero_only(model):
    batch = synthetic.parsing_batch(2, seed=22)
    batch['texture_mask'] = torch.zeros_like(batch['texture_mask'])
    model.feed_data(batch)
    top = torch.stack(model.sample_fn(temp=1, sample_steps=3))
    assert (top[0] >= 0).all() and (top[1:] == -1).all()


def test_argument_validation_raises():
    a = torch.zeros(64, 48, device=DEV)
    with pytest.raises(T2HError, match='multiple of 32'):
        ops.gemm(a, torch.zeros(32, 48, device=DEV))
    with pytest.raises(TypeError):
        ops.gemm(a.double(), torch.zeros(32, 48, device=DEV))
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(64, 64, device=DEV).t(), torch.zeros(32, 64, device=DEV))
    with pytest.raises(T2HError, match='T='):
        ops.mha_noncausal(torch.zeros(2 * 100, 1536, device=DEV), 2, 100, 8)
    with pytest.raises(T2HError, match='unsupported'):
        ops.layernorm(torch.zeros(8, 384, device=DEV), torch.ones(384, device=DEV), torch.zeros(384, device=DEV))


@pytest.mark.parametrize('noise', ['torch_device_generator', 'explicit_cpu_draws'])
def test_compact_rounds_give_the_synchronous_loops_tokens(model, monkeypatch, noise):
    """engine.sample_tokens: every sample walking through its OWN active steps (the default) samples
    exactly the tokens of the reference's synchronous loop -- for the in-kernel Philox reproduction of
    torch's draws and for explicit draws -- with fewer transformer evaluations; the torch generator
    ends at the same offset either way."""
    from text2human_amd import engine
    batch = synthetic.parsing_batch(3, seed=31)
    model.feed_data(batch)
    tex_tok = model._texture_tokens(model.texture_mask)
    runs, stats, ends = {}, {}, {}
    for compact in (True, False):
        torch.cuda.manual_seed_all(123)
        nz = R.SeededNoise(5, 'cpu') if noise == 'explicit_cpu_draws' else None
        runs[compact] = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 40,
                                             model.mask_id, noise=nz, compact=compact).clone()
        stats[compact] = dict(model.sampler_fn.last_stats)
        ends[compact] = torch.cuda.default_generators[torch.cuda.current_device()].get_offset()
    assert torch.equal(runs[True], runs[False])
    assert ((runs[True] >= 0).sum(0) == 1).all()                      # every token sampled once, by one head
    assert ends[True] == ends[False]
    assert stats[True]['sample_steps_needed'] == stats[False]['sample_steps_needed']
    assert stats[True]['rounds'] <= stats[False]['rounds'] <= 40
    assert stats[True]['sample_steps_launched'] >= stats[True]['sample_steps_needed']
    if noise == 'torch_device_generator':                             # and T2H_COMPACT_ROUNDS=0 selects the other one
        monkeypatch.setenv('T2H_COMPACT_ROUNDS', '0')
        torch.cuda.manual_seed_all(123)
        again = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 40, model.mask_id)
        assert model.sampler_fn.last_stats['rounds'] == stats[False]['rounds'] and torch.equal(again, runs[False])


def test_emulated_torch_draws_fall_back_to_real_ones(model, monkeypatch):
    """If the in-kernel reproduction of ATen's Philox kernels ever disagreed with the installed torch
    (TorchDeviceNoise.emulation_ok), the schedule is built from real torch draws: same tokens, same
    final generator offset."""
    from text2human_amd import engine
    model.feed_data(synthetic.parsing_batch(2, seed=32))
    tex_tok = model._texture_tokens(model.texture_mask)
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    torch.cuda.manual_seed_all(7)
    a = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 12, model.mask_id).clone()
    end_a = gen.get_offset()
    monkeypatch.setattr(engine.TorchDeviceNoise, 'emulation_ok', lambda self, n, k: False)
    torch.cuda.manual_seed_all(7)
    b = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 12, model.mask_id)
    assert torch.equal(a, b) and gen.get_offset() == end_a


def test_stale_overflow_flag_is_not_this_runs_and_decode_checks_its_own(opt, sds, model, monkeypatch):
    """The sticky split-overflow flag: a flag left by an earlier producer must not fail the next valid
    sampling run, and a decoder activation outside fp16's range must raise from decode_indices, naming
    the decode stage."""
    from text2human_amd import engine
    model.feed_data(synthetic.parsing_batch(1, seed=33))
    ops.split_rows(torch.full((32, 64), 1.0e5, device=DEV))          # raises the flag of this stream
    top = model.sample_fn(temp=1, sample_steps=2)                     # ... which sample_fn clears at its start
    assert not ops.split_overflow(reset=False)
    model.decode_indices(top)
    bad = {k: dict(v) for k, v in sds.items()}
    bad['top_post_quant_conv']['weight'] = sds['top_post_quant_conv']['weight'] * 1.0e12
    m2 = SampleFromParsingModel(opt, state_dicts=bad)
    m2.feed_data(synthetic.parsing_batch(1, seed=33))
    monkeypatch.setenv('T2H_OVERFLOW_FALLBACK', '0')   # (default: the stage is re-run on the exact-fp32 kernels)
    with pytest.raises(engine.SplitOverflowError, match='VQGAN refine / decode.*T2H_SPLIT_CONV'):
        m2.decode_indices(top)
    assert not ops.split_overflow(reset=True)


def test_last_layer_trim_on_and_off_sample_the_same_tokens_on_the_bench_config(monkeypatch):
    """T2H_TRIM_LAST_LAYER: the last layer's row-wise tail on the changed rows only (few-rows GEMM
    kernel, another summation order) vs on all rows -- equal to rounding, so the tokens of the
    benchmarked configuration (B = 8, 256 steps, seed 2021) must agree up to float near-ties; on the
    synthetic default weights (noise-decided races) none is expected."""
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    model = SampleFromParsingModel(opt, state_dicts=synthetic.make_state_dicts(opt, seed=1234))
    model.feed_data(synthetic.parsing_batch(8, seed=2021))
    toks = {}
    for trim in ('1', '0'):
        monkeypatch.setenv('T2H_TRIM_LAST_LAYER', trim)
        options.set_random_seed(2021)
        toks[trim] = torch.stack(model.sample_fn(temp=1, sample_steps=256))
    assert int((toks['1'] != toks['0']).sum()) == 0


def test_graph_replay_of_the_sampling_rounds_gives_the_same_tokens(model, monkeypatch):
    """T2H_GRAPH=1 (the default): one round = one replay of a captured launch sequence (engine.RoundGraph; row lists
    padded to a fixed length with duplicates, seed read from device memory).  Same tokens and the same
    final generator offset as the launch-by-launch loop -- on the run that captures, on a replay-only
    run with ANOTHER seed, and after a change of batch size dropped the captured graph."""
    from text2human_amd import engine
    gen = torch.cuda.default_generators[torch.cuda.current_device()]

    def run(seed, graph, b):
        model.feed_data(synthetic.parsing_batch(b, seed=31))
        tex_tok = model._texture_tokens(model.texture_mask)
        monkeypatch.setenv('T2H_GRAPH', '1' if graph else '0')
        torch.cuda.manual_seed_all(seed)
        out = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 48, model.mask_id).clone()
        return out, gen.get_offset()

    for seed, b in ((123, 3), (456, 3), (789, 2), (123, 3), (-3, 3)):   # (-3: a uint64 seed >= 2^63, held as int64 on the device)
        want, off_w = run(seed, False, b)
        got, off_g = run(seed, True, b)
        assert torch.equal(got, want) and off_g == off_w, (seed, b)
        assert ((got >= 0).sum(0) == 1).all()
    assert len(model.sampler_fn._graphs) == 1      # the B = 2 call replaced the buffers and the B = 3 graph with them
    assert model.sampler_fn.last_launch_mode == 'graph'



def test_graph_capture_and_replay_beside_a_live_rccl_communicator(opt, sds, monkeypatch):
    """The product default (one captured hipGraph per sampling round) next to a live RCCL communicator, as every
    rank of a multi-GPU run has it (bench.py / shard.py: weights broadcast, barriers, gathers): a one-rank NCCL
    process group is created in this process, a collective runs on it, and a FRESH model then captures its round
    graph and replays it, with another collective between two runs.  Tokens must equal the launch-by-launch loop's."""
    import socket
    import torch.distributed as dist
    from text2human_amd import engine
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv('MASTER_ADDR', '127.0.0.1')
    monkeypatch.setenv('MASTER_PORT', str(port))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        t = torch.ones(1024, device=DEV)
        dist.all_reduce(t)                 # the communicator really exists
        dist.barrier()
        model = SampleFromParsingModel(opt, state_dicts=sds)
        model.feed_data(synthetic.parsing_batch(3, seed=41))
        tex_tok = model._texture_tokens(model.texture_mask)

        def run(seed, graph):
            monkeypatch.setenv('T2H_GRAPH', '1' if graph else '0')
            torch.cuda.manual_seed_all(seed)
            return engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 40, model.mask_id).clone()

        for seed in (5, 6):
            got = run(seed, True)          # seed 5 captures, seed 6 only replays
            assert model.sampler_fn.last_launch_mode == 'graph'
            dist.all_reduce(t)             # a collective between the graph runs
            assert torch.equal(got, run(seed, False)), seed
        torch.cuda.synchronize()
        assert float(t[0]) == 1.0
    finally:
        dist.destroy_process_group()


def test_decoder_attention_flash_fallback_matches_the_materialised_default(model, monkeypatch):
    """The decoders' AttnBlocks run the materialised bmm -> softmax -> bmm form unless the N x N score tensor exceeds
    T2H_ATTN_SCORE_MB; with the budget at 0 every one of them takes the flash-style kernel instead.  Same image."""
    top, mask = _rand_indices(2, seed=77)
    model.texture_mask, model.batch_size = mask.to(DEV), 2
    img_a, _ = model.decode_indices([t.to(DEV) for t in top])
    monkeypatch.setenv('T2H_ATTN_SCORE_MB', '0')
    img_b, _ = model.decode_indices([t.to(DEV) for t in top])
    assert (img_a - img_b).abs().max().item() < 2e-4
    assert not torch.equal(img_a, img_b)     # (another kernel really ran)


@pytest.mark.parametrize('graph', ['0', '1'])
def test_finished_samples_leave_the_batch_same_tokens(model, monkeypatch, graph):
    """T2H_SHRINK_BATCH (default on): the samples are reordered by their number of rounds (schedule.leave_order) and
    the transformer of a round runs on the samples that still have a step left -- a prefix of the reordered batch;
    every row keeps its own element of the reference's noise tensors (t2h_sample_heads_args.rng_rows).  Tokens, in
    the caller's sample order, and the final generator offset equal the full-batch schedule's, launch by launch and
    as per-k captured graphs."""
    from text2human_amd import engine
    gen = torch.cuda.default_generators[torch.cuda.current_device()]
    model.feed_data(synthetic.parsing_batch(5, seed=43))
    tex_tok = model._texture_tokens(model.texture_mask)
    monkeypatch.setenv('T2H_GRAPH', graph)

    def run(seed, shrink):
        monkeypatch.setenv('T2H_SHRINK_BATCH', '1' if shrink else '0')
        torch.cuda.manual_seed_all(seed)
        out = engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, 128, model.mask_id).clone()
        return out, gen.get_offset(), dict(model.sampler_fn.last_stats)

    for seed in (3, 4):
        want, off_w, st_w = run(seed, False)
        got, off_g, st_g = run(seed, True)
        assert torch.equal(got, want) and off_g == off_w, seed
        assert st_g['sample_steps_launched'] == st_g['sample_steps_needed'] <= st_w['sample_steps_launched']
    assert st_g['sample_steps_launched'] < st_w['sample_steps_launched']   # (128 steps, 5 samples: somebody finishes early)
