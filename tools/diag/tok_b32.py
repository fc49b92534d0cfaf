"""Diagnostic (GPU): tokenizer at B=32 vs B=2 vs the CPU oracle on identical parsing maps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from oracle import torch_ref as R
from text2human_amd import defaults, options, synthetic, ops
from text2human_amd.models import SampleFromPoseModel
DEV = 'cuda'
opt = options.dict_to_nonedict(defaults.sample_from_pose())
sds = synthetic.make_state_dicts(opt, seed=4321)
model = SampleFromPoseModel(opt, state_dicts=sds)
pb = synthetic.pose_batch(32, seed=2021)
model.feed_data(pb)
model.generate_parsing_map()
segm = model.segm.clone()
print('distinct parsing maps among the 32:', len({segm[i].cpu().numpy().tobytes() for i in range(32)}))
toks = {}
for B in (32, 8, 2, 1):
    toks[B] = model.get_quantized_segm(segm[:B]).view(B, -1).cpu()
for B in (8, 2, 1):
    print(f'B=32 vs B={B} on the first {B} samples: {(toks[32][:B] != toks[B]).sum().item()} tokens differ')
with torch.no_grad():
    s2 = segm[:2].cpu()
    one_hot = F.one_hot(s2.squeeze(1).long(), 24).permute(0, 3, 1, 2).float()
    z = R.encoder(one_hot, sds['segm_encoder'])
    z = F.conv2d(z, sds['segm_quant_conv']['weight'], sds['segm_quant_conv']['bias'])
    zr = z.permute(0, 2, 3, 1).reshape(-1, z.shape[1])
    cb = sds['segm_quantizer']['embedding.weight']
    d = (zr ** 2).sum(1, keepdim=True) + (cb ** 2).sum(1)[None] - 2 * zr @ cb.t()
    ref = d.argmin(1).view(2, -1)
    two = d.topk(2, dim=1, largest=False).values
    margin = ((two[:, 1] - two[:, 0]) / two[:, 0].abs().clamp_min(1e-30)).view(2, -1)
for B in (32, 2):
    bad = toks[B][:2] != ref
    print(f'HIP B={B} vs CPU oracle (2 samples): {bad.sum().item()} differ; margins of those: {margin[bad][:10].tolist()}')
print('oracle margin quantiles:', torch.quantile(margin.flatten(), torch.tensor([0.0, 0.01, 0.1, 0.5])).tolist())
print('|z| range', float(zr.abs().max()), 'z std', float(zr.std()))
# the HIP latent rows against the oracle's
x = ops.onehot_nhwc(segm[:2].to(torch.float32).reshape(-1), 24, model.segm_cin_pad)
zh, h, w = model.segm_encoder.encode(x, 2, 512, 256)
zh = ops.gemm(zh, model.P['segm.qc.w'], bias=model.P['segm.qc.b']).cpu()
print('latent max abs err HIP(B=2) vs oracle:', float((zh - zr).abs().max()))
x = ops.onehot_nhwc(segm.to(torch.float32).reshape(-1), 24, model.segm_cin_pad)
zh32, h, w = model.segm_encoder.encode(x, 32, 512, 256)
zh32 = ops.gemm(zh32, model.P['segm.qc.w'], bias=model.P['segm.qc.b']).cpu()
print('latent max abs err HIP(B=32)[:2] vs oracle:', float((zh32[:1024] - zr).abs().max()), ' vs HIP(B=2):', float((zh32[:1024] - zh).abs().max()))
