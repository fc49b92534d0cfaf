"""Kernel parity (-m gpu) of the pose front-end glue: attribute embedding MLPs
and the per-pixel tap-bias map of ShapeUNet's broadcast attribute channels."""
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_ref as R
from text2human_amd import defaults, ops, synthetic, weights

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def test_shape_attr_embed_matches_oracle():
    opt = defaults.sample_from_pose()
    cls = opt['shape_attr_class_num']
    sd = synthetic.fill(synthetic.shape_embedder_schema(8, 128, cls), seed=5)
    P = weights.Params(DEV)
    emb = weights.pack_shape_embedder(P, sd, 'semb', cls)
    g = torch.Generator().manual_seed(1)
    attr = torch.stack([torch.randint(0, c, (37, ), generator=g) for c in cls], 1)
    ref = R.shape_attr_embedding(attr, sd, cls)
    got = ops.shape_attr_embed(attr.to(DEV), emb).cpu()
    assert (got - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize('h,w', [(8, 4), (16, 16), (2, 1), (1, 1)])
def test_tap_bias_map_equals_conv_of_constant_channels(h, w):
    """conv3x3(pad 1) over a spatially constant A-channel map == tap-bias map."""
    B, A, cout = 3, 128, 64
    g = torch.Generator().manual_seed(2)
    wattr = torch.randn(cout, A, 3, 3, generator=g) * 0.05
    attr = torch.randn(B, A, generator=g)
    ref = F.conv2d(attr.double().view(B, A, 1, 1).expand(B, A, h, w), wattr.double(), None, 1, 1)
    packed = wattr.permute(0, 2, 3, 1).reshape(cout * 9, A).contiguous()   # [(co,tap), A]
    tapc = ops.gemm(attr.to(DEV), packed.to(DEV))
    got = ops.tap_bias_map(tapc, B, h, w, cout).cpu().view(B, h, w, cout).permute(0, 3, 1, 2)
    assert (got.double() - ref).abs().max().item() < 1e-4
