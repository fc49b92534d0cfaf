"""Split-precision path (-m gpu): three-bf16-plane operands, six-product GEMM on
the bf16 matrix cores, split-row producers.  Accuracy bar = the exact-fp32 path's
(summation-order-level error against an fp64 reference)."""
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import _lib, engine, ops, synthetic, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _unsplit(s, rows, c):
    return s.view(torch.bfloat16).view(rows, c // 32, 3, 32).float().sum(2).reshape(rows, c)


def test_split3_is_exact_to_fp32_and_matches_host_pack():
    x = torch.cat([_rnd(300, 512, seed=1) * 3, _rnd(300, 512, seed=2) * 1e-3, _rnd(8, 512, seed=3) * 1e4])
    s = ops.split3(x.to(DEV))
    assert torch.equal(s.cpu(), ops.pack_split_rows_host(x).view_as(s.cpu()))
    back = _unsplit(s.cpu(), x.shape[0], 512)
    assert ((back - x).abs() <= x.abs() * 2.0**-23).all()


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize('M,N,K', [(4096, 512, 512), (1024, 1536, 512), (512, 512, 2048), (130, 96, 64),
                                   (40, 32, 32)])
def test_gemm_split(cfg, M, N, K):
    if cfg == 6 and K % 64:
        pytest.skip('the in-block K split needs K % 64 == 0')
    a, w, b, r = _rnd(M, K, seed=4) * 1.3, _rnd(N, K, seed=5, scale=0.08), _rnd(N, seed=6), _rnd(M, N, seed=7)
    ref = a.double() @ w.double().t() + b.double()
    a_s, w_s = ops.split3(a.to(DEV)), ops.pack_split_rows_host(w).to(DEV)
    lib = _lib.load()
    lib.t2h_gemm_split_force_config(cfg)
    try:
        out = torch.empty(M, N, device=DEV)
        ops.gemm_split(a_s, w_s, M, N, K, out=out, bias=b.to(DEV), residual=r.to(DEV))
        err = (out.cpu().double() - (ref + r.double())).abs()
        assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), err.max().item()
        if N % 32 == 0:
            o_s = ops.split_rows_empty(M, N, DEV)
            ops.gemm_split(a_s, w_s, M, N, K, out_split=o_s, bias=b.to(DEV), act=ops.ACT_GELU)
            got = _unsplit(o_s.cpu(), M, N).double()
            err = (got - F.gelu(ref)).abs()
            assert (err <= 2e-5 + 2e-5 * ref.abs()).all(), err.max().item()
    finally:
        lib.t2h_gemm_split_force_config(-1)


def test_split_producers_are_bitwise_the_split_of_the_fp32_result():
    x, g, b = _rnd(777, 512, seed=8) * 2 + 0.1, _rnd(512, seed=9) * 0.1 + 1, _rnd(512, seed=10) * 0.1
    ln = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV))
    ln_s = ops.layernorm_split(x.to(DEV), g.to(DEV), b.to(DEV), ops.split_rows_empty(777, 512, DEV))
    assert torch.equal(ln_s, ops.split3(ln))
    qkv = (_rnd(2 * 512, 1536, seed=11) * 1.2).to(DEV)
    y = ops.mha_noncausal(qkv, 2, 512, 8)
    y_s = ops.mha_noncausal_split(qkv, 2, 512, 8, ops.split_rows_empty(1024, 512, DEV))
    assert torch.equal(y_s, ops.split3(y))


def test_sampler_net_split_matches_oracle_and_fp32_path():
    from oracle import torch_ref as R
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 4, 512, 18), seed=12)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    gen = torch.Generator().manual_seed(13)
    idx = torch.randint(0, 18433, (2, 512), generator=gen)
    seg = torch.randint(0, 1024, (2, 512), generator=gen)
    tex = torch.randint(0, 18, (2, 512), generator=gen)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    a = engine.SamplerNet(P, desc, 8, 'tf', split=False).hidden(*args).clone().cpu()
    b = engine.SamplerNet(P, desc, 8, 'tf', split=True).hidden(*args).clone().cpu()
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sd)
    ln = lambda t: F.layer_norm(t.view(2, 512, 512), (512, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)
    ea, eb = (ln(a) - ref).abs().max().item(), (ln(b) - ref).abs().max().item()
    assert ea < 1e-4 and eb < 1e-4, (ea, eb)
    assert eb < 3 * ea + 1e-6, f'split path error {eb:.2e} vs fp32 path {ea:.2e}'
