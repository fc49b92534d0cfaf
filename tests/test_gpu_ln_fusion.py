"""Fused LayerNorm (-m gpu): slab statistics produced by `t2h_row_stats_f32` /
GEMM epilogues, consumed by the next GEMM's operand staging."""
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import ops

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _slab_stats(x):
    r, c = x.shape
    xs = x.double().view(r, c // 32, 32)
    return torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1)


def test_row_stats():
    x = _rnd(1001, 512, seed=1) * 3 + 0.5
    got = ops.row_stats(x.to(DEV)).cpu().double()
    ref = _slab_stats(x)
    assert (got - ref).abs().max().item() < 1e-3 and ((got - ref).abs() / (ref.abs() + 1)).max().item() < 1e-5


@pytest.mark.parametrize('M,N', [(4096, 1536), (640, 2048), (130, 512)])
def test_gemm_with_fused_layernorm(M, N):
    K = 512
    x = _rnd(M, K, seed=2) * 2 + 0.3
    w, b = _rnd(N, K, seed=3, scale=0.05), _rnd(N, seed=4)
    g, be = _rnd(K, seed=5) * 0.1 + 1, _rnd(K, seed=6) * 0.1
    ref = F.linear(F.layer_norm(x.double(), (K, ), g.double(), be.double(), 1e-5), w.double(), b.double())
    w_ln = (w.double() * g.double()[None]).float()
    b_ln = (b.double() + w.double() @ be.double()).float()
    st = ops.row_stats(x.to(DEV))
    out = ops.gemm(x.to(DEV), w_ln.to(DEV), bias=b_ln.to(DEV), ln_stats_in=st)
    err = (out.cpu().double() - ref).abs()
    assert (err <= 3e-5 + 3e-5 * ref.abs()).all(), err.max().item()


def test_sampler_net_fused_equals_unfused():
    from oracle import torch_ref as R
    from text2human_amd import engine, synthetic, weights
    sd = synthetic.fill(synthetic.transformer_schema(18432, 1024, 18, 512, 3, 512, 18), seed=9)
    P = weights.Params(DEV)
    desc = weights.pack_transformer(P, sd, 'tf')
    g = torch.Generator().manual_seed(3)
    idx = torch.randint(0, 18433, (2, 512), generator=g)
    seg = torch.randint(0, 1024, (2, 512), generator=g)
    tex = torch.randint(0, 18, (2, 512), generator=g)
    args = (idx.to(DEV), seg.to(DEV), tex.to(DEV))
    a = engine.SamplerNet(P, desc, 8, 'tf', fuse_ln=False).hidden(*args).clone()
    b = engine.SamplerNet(P, desc, 8, 'tf', fuse_ln=True).hidden(*args).clone()
    with torch.no_grad():
        ref = R.transformer_hidden(idx, seg, tex, sd)  # includes ln_f
    ln = lambda t: F.layer_norm(t.cpu().view(2, 512, 512), (512, ), sd['ln_f.weight'], sd['ln_f.bias'], 1e-5)
    assert (ln(a) - ref).abs().max().item() < 1e-4
    assert (ln(b) - ref).abs().max().item() < 1e-4


def test_gemm_epilogue_statistics_feed_the_next_gemm():
    """proj-like GEMM (+bias +residual, in place) emits stats; an fc1-like GEMM
    consumes them: equals LN(x_new) @ W^T."""
    M, C = 1024, 512
    x = _rnd(M, C, seed=7).to(DEV)
    y = _rnd(M, C, seed=8).to(DEV)
    wp, bp = _rnd(C, C, seed=9, scale=0.05).to(DEV), _rnd(C, seed=10).to(DEV)
    w1 = _rnd(2048, C, seed=11, scale=0.05)
    x_new_ref = x.cpu().double() + y.cpu().double() @ wp.cpu().double().t() + bp.cpu().double()
    st = torch.full((M, C // 32, 2), float('nan'), device=DEV)
    ops.gemm(y, wp, out=x, bias=bp, residual=x, ln_stats_out=st)
    assert (x.cpu().double() - x_new_ref).abs().max().item() < 1e-4
    ref_st = _slab_stats(x.cpu())
    assert not torch.isnan(st).any()
    assert ((st.cpu().double() - ref_st).abs() / (ref_st.abs() + 1)).max().item() < 1e-5
    out = ops.gemm(x, w1.to(DEV), act=ops.ACT_GELU, ln_stats_in=st)
    ref = F.gelu(F.layer_norm(x.cpu().double(), (C, )) @ w1.double().t())
    err = (out.cpu().double() - ref).abs()
    assert (err <= 3e-5 + 3e-5 * ref.abs()).all(), err.max().item()
