"""The x8 operand format of the split GEMM (-m gpu): fp16 plane + two e4m3 planes per (row, 32-wide K tile), both
cross terms of a K tile in ONE 8-bit matrix instruction (csrc/common.h, csrc/gemm_split.hip FMT = 1).  Producers
against a host restatement of the format bit for bit; the GEMM in every tile configuration built for it against an
fp64 EMULATION OF ITS OWN ARITHMETIC (tight: proves the plane pairing, scales and accumulation) and against the true
fp64 product (loose: what the format costs); the sampler network against the oracle; the range fallback."""
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import _lib, engine, ops, synthetic, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def _x8_planes_host(x, s):
    """(hi, h8, l8) as the numbers they stand for: hi = fp16(x), h8 = e4m3(hi s) / s, l8 = e4m3((x - hi) 2048 s) / s"""
    hi = x.half().float()
    return hi, _e4m3(hi * s) / s, _e4m3((x - hi) * 2048.0 * s) / s


def _emulate(a, w, sa, sw):
    """the kernel's arithmetic in fp64: ah.bh + (ah8.bl8 + al8.bh8) / 2048"""
    ah, ah8, al8 = [t.double() for t in _x8_planes_host(a, sa)]
    bh, bh8, bl8 = [t.double() for t in _x8_planes_host(w, sw)]
    return ah @ bh.t() + (ah8 @ bl8.t() + al8 @ bh8.t()) / 2048.0


def test_split_rows_x8_planes_and_range_bits():
    x = _rnd(96, 512, seed=1) * 1.7
    x[3, 5], x[7, 100], x[9, 9] = 0.0, 1.0e-6, -13.25
    s = 16.0
    ops.split_overflow(reset=True)
    got = ops.split_rows_x8(x.to(DEV), s)
    assert ops.split_overflow_bits(reset=True) == 0
    hi, h8, l8 = ops.unpack_x8_rows_host(got, 96, 512, s)
    whi, wh8, wl8 = _x8_planes_host(x, s)
    assert torch.equal(hi, whi) and torch.equal(h8, wh8) and torch.equal(l8, wl8)
    # |x| s >= 448 raises bit 1 (the fp16 plane is still exact), |x| >= 65504 bit 0
    x2 = x.clone()
    x2[0, 0] = 30.0
    got = ops.split_rows_x8(x2.to(DEV), s)
    assert ops.split_overflow_bits(reset=True) == 2
    assert torch.equal(ops.unpack_x8_rows_host(got, 96, 512, s)[0], x2.half().float())
    x2[0, 0] = 1.0e5
    ops.split_rows_x8(x2.to(DEV), s)
    assert ops.split_overflow_bits(reset=True) & 1
    assert ops.x8_scale_for(1.0) == 16.0 and ops.x8_scale_for(31.9) == 1.0 and ops.x8_scale_for(32.0) == 0.5 and ops.x8_scale_for(0.02, 256.0) == 8192.0


@pytest.mark.parametrize('cfg', [0, 2, 6, 8, 10])
@pytest.mark.parametrize('M,N,K', [(512, 1536, 512), (1024, 512, 2048), (256, 2048, 512)])
def test_gemm_x8_every_tile_configuration(cfg, M, N, K):
    if cfg == 10 and N % 192:
        pytest.skip('128x192 tiles need N % 192 == 0')
    a, w, b = _rnd(M, K, seed=2) * 1.3, _rnd(N, K, seed=3, scale=0.05), _rnd(N, seed=4)
    sa, sw = ops.x8_scale_for(float(a.abs().max())), ops.x8_scale_for(float(w.abs().max()), 256.0)
    a_s, w_s = ops.split_rows_x8(a.to(DEV), sa), ops.split_rows_x8(w.to(DEV), sw)
    lib = _lib.load()
    lib.t2h_gemm_split_force_config(cfg)
    try:
        out = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=out, bias=b.to(DEV), x8=(sa, sw))
    finally:
        lib.t2h_gemm_split_force_config(-1)
    emu = _emulate(a, w, sa, sw) + b.double()
    true = a.double() @ w.double().t() + b.double()
    got = out.cpu().double()
    scale = (a.double().abs() @ w.double().abs().t())            # sum |a||b| of every element
    assert ((got - emu).abs() <= 2e-6 * scale + 1e-6).all(), ((got - emu).abs() / scale).max().item()
    # the format itself: 4-bit cross-term factors at weight 2^-11 -> a random walk of ~2^-15 |a||b| terms
    # (measured max 8.2e-5 sum|a||b| / sqrt K, i.e. ~5e-5 of a typical output)
    assert ((got - true).abs() <= 1.5e-4 * scale / K**0.5 + 1e-6).all(), ((got - true).abs() * K**0.5 / scale).max().item()


@pytest.mark.parametrize('M,N,K', [(1, 512, 512), (17, 2048, 512), (64, 512, 2048), (5, 48, 96)])
def test_gemm_x8_few_rows_kernel_and_epilogues(M, N, K):
    a, w, b, r = _rnd(M, K, seed=5) * 1.1, _rnd(N, K, seed=6, scale=0.07), _rnd(N, seed=7), _rnd(M, N, seed=8)
    sa, sw = ops.x8_scale_for(float(a.abs().max())), ops.x8_scale_for(float(w.abs().max()), 256.0)
    a_s, w_s = ops.split_rows_x8(a.to(DEV), sa), ops.split_rows_x8(w.to(DEV), sw)
    lib = _lib.load()
    g = _lib.GemmSplitArgs()
    out = torch.empty(M, N, device=DEV)
    ops.gemm_split(a_s, w_s, M, N, K, out=out, bias=b.to(DEV), residual=r.to(DEV), x8=(sa, sw))
    emu = _emulate(a, w, sa, sw) + b.double() + r.double()
    scale = a.double().abs() @ w.double().abs().t()
    assert ((out.cpu().double() - emu).abs() <= 2e-6 * scale + 1e-6).all()
    if N % 32 == 0:  # GELU + x8 output (fc1 of the tail): the planes of the fp32 result
        so = 8.0
        o_x = ops.split_rows_empty(M, N, DEV)
        o32 = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=o32, out_split=o_x, bias=b.to(DEV), act=ops.ACT_GELU, x8=(sa, sw),
                       out_x8_scale=so)
        hi, h8, l8 = ops.unpack_x8_rows_host(o_x, M, N, so)
        whi, wh8, wl8 = _x8_planes_host(o32.cpu(), so)
        assert torch.equal(hi, whi) and torch.equal(h8, wh8) and torch.equal(l8, wl8)
        assert (o32.cpu().double() - F.gelu(_emulate(a, w, sa, sw) + b.double())).abs().max().item() < 1e-5


def test_gemm_x8_fc1_epilogue_and_qkv_routing_on_the_big_tiles():
    """M = 4096 (the dispatcher's own choice: 256x128 / 128x192 ping-pong tiles): GELU + x8 rows out (fc1); q|k as
    fp16-plane rows + transposed value planes (q|k|v: the attention kernel's operands keep their format)."""
    M, C = 4096, 512
    a = _rnd(M, C, seed=9) * 1.2
    sa = ops.x8_scale_for(float(a.abs().max()))
    a_s = ops.split_rows_x8(a.to(DEV), sa)
    w1, b1 = _rnd(4 * C, C, seed=10, scale=0.05), _rnd(4 * C, seed=11)
    sw1 = ops.x8_scale_for(float(w1.abs().max()), 256.0)
    so = 16.0
    o_x, o32 = ops.split_rows_empty(M, 4 * C, DEV), torch.empty(M, 4 * C, device=DEV)
    ops.gemm_split(a_s, ops.split_rows_x8(w1.to(DEV), sw1), M, 4 * C, C, out=o32, out_split=o_x, bias=b1.to(DEV),
                   act=ops.ACT_GELU, x8=(sa, sw1), out_x8_scale=so)
    ref = F.gelu(_emulate(a, w1, sa, sw1) + b1.double())
    assert (o32.cpu().double() - ref).abs().max().item() < 1e-5
    hi, h8, l8 = ops.unpack_x8_rows_host(o_x, M, 4 * C, so)
    whi, wh8, wl8 = _x8_planes_host(o32.cpu(), so)
    assert torch.equal(hi, whi) and torch.equal(h8, wh8) and torch.equal(l8, wl8)
    # q|k|v
    B, T, H = M // 512, 512, 8
    wq, bq = _rnd(3 * C, C, seed=12, scale=0.06), _rnd(3 * C, seed=13)
    swq = ops.x8_scale_for(float(wq.abs().max()), 256.0)
    wq_s = ops.split_rows_x8(wq.to(DEV), swq)
    full = torch.empty(M, 3 * C, device=DEV)
    ops.gemm_split(a_s, wq_s, M, 3 * C, C, out=full, bias=bq.to(DEV), x8=(sa, swq))
    qk_s, vt = ops.split_rows_empty(M, 3 * C, DEV), ops.vt_empty(B, H, T, DEV)
    qk_s.zero_()
    ops.gemm_split(a_s, wq_s, M, 3 * C, C, out_split=qk_s, bias=bq.to(DEV), vt=vt, vt_col0=2 * C, vt_T=T,
                   x8=(sa, swq))
    want = ops.pack_split_rows_host(full.cpu())
    assert torch.equal(qk_s.cpu()[:, :2 * C // 32], want[:, :2 * C // 32])
    from test_gpu_split import _pack_vt_host
    assert torch.equal(vt.cpu(), _pack_vt_host(full[:, 2 * C:].contiguous().cpu(), B, T, H))
    assert (full.cpu().double() - (_emulate(a, wq, sa, swq) + bq.double())).abs().max().item() < 1e-5


def test_layernorm_and_attention_write_the_x8_planes_of_their_fp16_plane_results():
    rows, C, s = 1024, 512, 4.0
    x = _rnd(rows, C, seed=14) * 2.0 + 0.3
    g, b = _rnd(C, seed=15) * 0.2 + 1.0, _rnd(C, seed=16) * 0.1
    v1 = ops.layernorm_split(x.to(DEV), g.to(DEV), b.to(DEV), ops.split_rows_empty(rows, C, DEV))
    x8 = ops.layernorm_x8(x.to(DEV), g.to(DEV), b.to(DEV), ops.split_rows_empty(rows, C, DEV), s)
    y = ops.unsplit_rows_host(v1, rows, C)                    # the fp32 result to 22 bits
    hi, h8, l8 = ops.unpack_x8_rows_host(x8, rows, C, s)
    assert torch.equal(hi, v1.cpu().view(torch.float16).view(rows, C // 32, 2, 32)[:, :, 0].float().reshape(rows, C))
    assert torch.equal(h8, _e4m3(hi * s) / s)
    lo16 = v1.cpu().view(torch.float16).view(rows, C // 32, 2, 32)[:, :, 1].float().reshape(rows, C)  # fp16((x - h) 2048)
    assert ((l8 - lo16).abs() <= lo16.abs() / 16 + 2.0**-9 / s * 1.01).all()   # e4m3 of the same residual
    ref = F.layer_norm(x, (C, ), g, b, 1e-5)
    assert (y - ref).abs().max().item() < 1e-5
    # attention
    B, T, H = 2, 512, 8
    qkv = _rnd(B * T, 3 * C, seed=17) * 1.3
    from test_gpu_split import _pack_vt_host
    qk_s = ops.split_rows(qkv.to(DEV))
    vt = _pack_vt_host(qkv[:, 2 * C:].contiguous(), B, T, H).to(DEV)
    y1 = ops.mha_split(qk_s, 3 * C, vt, B, T, H, out_split=ops.split_rows_empty(B * T, C, DEV))
    y8 = ops.mha_split_x8(qk_s, 3 * C, vt, B, T, H, ops.split_rows_empty(B * T, C, DEV), 64.0)
    hi, h8, _ = ops.unpack_x8_rows_host(y8, B * T, C, 64.0)
    assert torch.equal(hi, y1.cpu().view(torch.float16).view(B * T, C // 32, 2, 32)[:, :, 0].float().reshape(B * T, C))
    assert torch.equal(h8, _e4m3(hi * 64.0) / 64.0)
    assert ops.split_overflow_bits(reset=True) == 0


def test_sampler_net_x8_against_the_oracle_and_the_fp16_plane_path():
    from oracle import torch_ref as R
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 6, 512, 18), seed=12)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    gen = torch.Generator().manual_seed(14)
    idx = torch.randint(0, 18433, (3, 512), generator=gen)
    seg = torch.randint(0, 1024, (3, 512), generator=gen)
    tex = torch.randint(0, 18, (3, 512), generator=gen)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    n8 = engine.SamplerNet(P, desc, 8, 'tf', split=True, x8=True)
    n1 = engine.SamplerNet(P, desc, 8, 'tf', split=True, x8=False)
    ops.split_overflow(reset=True)
    a = n8.hidden(*args).clone().cpu()
    assert n8._x8 is not None and ops.split_overflow_bits(reset=True) == 0
    b = n1.hidden(*args).clone().cpu()
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sd)
    ln = lambda t: F.layer_norm(t.view(3, 512, 512), (512, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)
    e8, e1 = (ln(a) - ref).abs().max().item(), (ln(b) - ref).abs().max().item()
    assert e1 < 1e-5 and e8 < 1e-4, (e8, e1)
    assert e8 > e1                     # (the 8-bit cross terms are really in use)
    # the scales put every calibrated maximum into [16, 32)
    for k, m in n8._x8['act_max'].items():
        assert 16.0 <= m * n8._x8['a'][k] < 32.0, (k, m)
    # the deferred tail on a compact row list = those rows of the full evaluation (few-rows kernel, x8 operands)
    rows = torch.randperm(3 * 512, generator=gen)[:37].to(torch.int32).to(DEV)
    full = n8.hidden(*args).clone()
    n8.hidden(*args, defer_tail=True)
    got, compact = n8.finish_tail(rows, 37)
    assert compact and (got - full[rows.long()]).abs().max().item() < 5e-5


def test_x8_range_overflow_falls_back_to_the_fp16_planes():
    from parity_util import seed_all
    from text2human_amd import defaults, options
    from text2human_amd.models import SampleFromParsingModel, sample_model as SM
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    batch = synthetic.parsing_batch(2, seed=3)
    ref_model = SampleFromParsingModel(opt, state_dicts=sds)
    ref_model.sampler_fn.x8 = False
    ref_model.feed_data(batch)
    seed_all(9)
    want = torch.stack(ref_model.sample_fn(temp=1, sample_steps=10))
    state = torch.cuda.get_rng_state(DEV)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    assert model.sampler_fn.x8
    model.feed_data(batch)
    seed_all(9)
    got = torch.stack(model.sample_fn(temp=1, sample_steps=10))     # x8: same tokens on this fixture
    assert torch.equal(got, want) and model.sampler_fn.x8 and model.sampler_fn.last_launch_mode == 'graph'
    # scales 64x too large: every producer saturates -> bit 1 -> THIS call is re-run on the fp16 planes
    good = dict(model.sampler_fn._x8['a'])
    for k in model.sampler_fn._x8['a']:
        model.sampler_fn._x8['a'][k] *= 64.0
    model.sampler_fn._graphs = {}
    SM._warned.discard('index sampler (x8 range)')
    seed_all(9)
    with pytest.warns(UserWarning, match='fp16-plane'):
        got = torch.stack(model.sample_fn(temp=1, sample_steps=10))
    assert torch.equal(got, want)
    assert torch.equal(torch.cuda.get_rng_state(DEV), state)
    # the fall-back is per call: the net is still an x8 net, and with its scales back the next call runs
    # on x8 operands again (a call's result never depends on an earlier call)
    assert model.sampler_fn.x8
    model.sampler_fn._x8['a'].update(good)
    model.sampler_fn._graphs = {}
    seed_all(9)
    got = torch.stack(model.sample_fn(temp=1, sample_steps=10))
    assert torch.equal(got, want) and model.sampler_fn.x8 and model.sampler_fn.last_launch_mode == 'graph'


def test_x8_result_is_a_pure_function_of_checkpoint_input_and_seed():
    """the x8 scales come from the checkpoint alone (engine.SamplerNet.calibrate_x8: fixed
    synthetic states at load), so a model object's history cannot change how a value is rounded.  Batch Y on a fresh
    model, on a model that first sampled batch X (another batch size, another seed), and on a model that first met an
    x8 range fall-back: bitwise-equal tokens, bitwise-equal uint8 images, the same generator state -- and the scales of
    all models are equal."""
    from parity_util import seed_all
    from text2human_amd import defaults, options
    from text2human_amd.models import SampleFromParsingModel, sample_model as SM
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    X, Y = synthetic.parsing_batch(3, seed=41), synthetic.parsing_batch(2, seed=3)

    def run_y(model):
        model.feed_data(Y)
        seed_all(2021)
        top = model.sample_fn(temp=1, sample_steps=24)
        _, u8 = model.decode_indices(top, want_u8=True)
        assert model.sampler_fn.x8 and model.sampler_fn.last_launch_mode == 'graph'
        return torch.stack(top).cpu(), u8.cpu(), torch.cuda.get_rng_state(DEV)

    fresh = SampleFromParsingModel(opt, state_dicts=sds)
    assert fresh.sampler_fn._x8 is not None   # calibrated at load, before any input
    want = run_y(fresh)

    seasoned = SampleFromParsingModel(opt, state_dicts=sds)
    seasoned.feed_data(X)
    seed_all(7)
    seasoned.sample_fn(temp=1, sample_steps=9)
    assert seasoned.sampler_fn._x8['a'] == fresh.sampler_fn._x8['a'] and seasoned.sampler_fn._x8['w'] == fresh.sampler_fn._x8['w']
    got = run_y(seasoned)
    assert all(torch.equal(a, b) for a, b in zip(got, want))

    # a model whose previous call fell back to the fp16 planes (scales sabotaged for that one call)
    fell = SampleFromParsingModel(opt, state_dicts=sds)
    good = dict(fell.sampler_fn._x8['a'])
    for k in fell.sampler_fn._x8['a']:
        fell.sampler_fn._x8['a'][k] *= 64.0
    fell.feed_data(X)
    seed_all(7)
    SM._warned.discard('index sampler (x8 range)')
    with pytest.warns(UserWarning, match='fp16-plane'):
        fell.sample_fn(temp=1, sample_steps=5)
    fell.sampler_fn._x8['a'].update(good)
    fell.sampler_fn._graphs = {}
    got = run_y(fell)
    assert all(torch.equal(a, b) for a, b in zip(got, want))

    # the calibration states are a function of the checkpoint's shapes only
    a, b = fresh.sampler_fn.x8_calibration_states(), seasoned.sampler_fn.x8_calibration_states()
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert int(a[0].max()) == 18432 and (a[0][0] == 18432).all() and (a[0][1] != 18432).all()


def test_x8_result_does_not_depend_on_the_rank_that_computes_it():
    """...nor on the process: inside a one-rank RCCL group (the sharded entry point's set-up: shard.broadcast_state_dicts,
    a live communicator) the model has the plain model's scales and gives its tokens.  (A rank of a larger world
    calibrates on the same checkpoint-only states -- there is no input it could differ by.)"""
    import os
    import torch.distributed as dist
    from parity_util import seed_all
    from text2human_amd import defaults, options, shard
    from text2human_amd.models import SampleFromParsingModel
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    Y = synthetic.parsing_batch(2, seed=3)
    plain = SampleFromParsingModel(opt, state_dicts=sds)
    plain.feed_data(Y)
    seed_all(2021)
    want = torch.stack(plain.sample_fn(temp=1, sample_steps=12)).cpu()
    if dist.is_initialized():
        pytest.skip('a process group is already up in this process')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        t = torch.ones(4, device=DEV)
        dist.all_reduce(t)                      # (a live communicator beside the model's graphs)
        sds_b = shard.broadcast_state_dicts(sds, dist.get_world_size(), torch.device('cuda', 0))
        ranked = SampleFromParsingModel(opt, state_dicts=sds_b)
        assert ranked.sampler_fn._x8['a'] == plain.sampler_fn._x8['a']
        ranked.feed_data(Y)
        seed_all(2021)
        got = torch.stack(ranked.sample_fn(temp=1, sample_steps=12)).cpu()
    finally:
        dist.destroy_process_group()
    assert torch.equal(got, want)
