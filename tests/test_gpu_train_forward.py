"""Sampler training-time forward (-m gpu): q_sample, the routed masked
cross entropy and _train_loss against the oracle (which is pinned to the reference's
TransformerTextureAwareModel._train_loss in tests/test_oracle_vs_reference.py)."""
import pytest
import torch
import torch.nn.functional as F

from text2human_amd import defaults, ops, options, synthetic
from text2human_amd.models import TransformerTextureAwareModel

from parity_util import odev, osds  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_q_sample_and_masked_ce_kernels():
    B, T, C, K, H = 2, 512, 512, 1024, 18
    g = torch.Generator().manual_seed(3)
    x0 = torch.randint(0, 18432, (B, T), generator=g)
    u = torch.rand(B, T, generator=g)
    t = torch.tensor([1, 200])
    x_t, mask = ops.q_sample(x0.to(DEV), u.to(DEV), t.to(DEV), 512, 18432)
    ref_mask = u < (t.float().unsqueeze(-1) / 512)
    assert torch.equal(mask.cpu().bool(), ref_mask)
    assert torch.equal(x_t.cpu(), torch.where(ref_mask, torch.full_like(x0, 18432), x0))
    hidden = torch.randn(B * T, C, generator=g) * 2
    lg, lb = torch.randn(C, generator=g) * 0.1 + 1, torch.randn(C, generator=g) * 0.1
    w = torch.randn(H, K, C, generator=g) * 0.1
    tex = torch.randint(0, H, (B * T, ), generator=g)
    code = torch.randint(0, K, (B * T, ), generator=g)
    gt = torch.full((H, B * T), -1, dtype=torch.long)
    gt[tex, torch.arange(B * T)] = code
    gt[tex[7], 7] = -1                       # a masked row without a target contributes nothing
    rows, samples = ops.masked_ce_heads(hidden.to(DEV), lg.to(DEV), lb.to(DEV), w.to(DEV), tex.to(DEV),
                                        mask.reshape(-1), gt.to(DEV), B, T)
    h = F.layer_norm(hidden.double(), (C, ), lg.double(), lb.double(), 1e-5)
    logits = torch.einsum('nc,nkc->nk', h, w.double()[tex])
    ce = F.cross_entropy(logits, gt[tex, torch.arange(B * T)], ignore_index=-1, reduction='none')
    ce = ce * ref_mask.reshape(-1)
    assert (rows.cpu().double() - ce).abs().max().item() < 1e-4
    assert (samples.cpu().double() - ce.view(B, T).sum(1)).abs().max().item() < 2e-3


@pytest.mark.parametrize('loss_type', ['reweighted_elbo', 'elbo', 'mlm'])
def test_train_loss_matches_oracle(loss_type):
    from oracle import torch_ref as R
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    opt['loss_type'] = loss_type
    sds = synthetic.make_state_dicts(opt, seed=1234, encode=True)
    model = TransformerTextureAwareModel(opt, state_dicts=sds)
    B = 2
    gen = torch.Generator().manual_seed(31)
    batch = synthetic.parsing_batch(B, seed=2021)
    image = torch.rand(B, 3, 512, 256, generator=gen) * 2 - 1
    model.feed_data(dict(image=image, segm=batch['segm'], texture_mask=batch['texture_mask']))
    t = torch.tensor([37, 800])
    u = torch.rand(B, 512, generator=gen)
    loss, vb = model._train_loss(model.input_indices, model.gt_indices_list, t=t, u=u)
    with torch.no_grad():
        od = osds(sds)   # (the oracle's encoders: parity_util.ORACLE_DEV; its transformer loss on the CPU)
        _, top_idx = R.top_encode(odev(image), odev(batch['texture_mask']), od)
        top_idx = odev(top_idx, 'cpu')
        tex = R.texture_tokens(batch['texture_mask'])
        seg = R.segm_tokens(odev(batch['segm']), od['segm_encoder'], od['segm_quant_conv'],
                            od['segm_quantizer']['embedding.weight']).view(B, -1).cpu()
        gt_list = [i.view(B, -1) for i in top_idx]
        own = torch.stack(gt_list).gather(0, tex[None])[0]
        x_0 = own + 1024 * tex
        ref_loss, ref_vb, ref_ce = R.train_loss(x_0, gt_list, seg, tex, sds['sampler'], t, u, num_timesteps=1000,
                                                loss_type=loss_type)
    assert torch.equal(model.input_indices.cpu(), x_0)
    assert (model.cross_entropy_loss.cpu() - ref_ce).abs().max().item() < 2e-2 + 1e-5 * ref_ce.abs().max().item()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    assert abs(vb.item() - ref_vb.item()) < 1e-4 * max(1.0, abs(ref_vb.item()))
    # device-RNG form: same seed, same draws as the reference order (randint, rand_like)
    torch.manual_seed(5)
    l1, _ = model._train_loss(model.input_indices, model.gt_indices_list)
    torch.manual_seed(5)
    tt, _ = model.sample_time(B, model.device)
    uu = torch.rand_like(model.input_indices.float())
    l2, _ = model._train_loss(model.input_indices, model.gt_indices_list, t=tt, u=uu)
    assert l1.item() == l2.item()
