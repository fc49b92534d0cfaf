"""Can the decode-side oracle of the long GPU parity cases run as eager PyTorch-ROCm instead of on the host cores?
(the driver's GPU step is mostly the CPU oracle's side of the full-length cases.)  Times the
oracle's tokenizer, refine + decode and the 1024 x 512 decoder on the CPU and on cuda:0 (cold + warm) and compares the
two executions of the SAME oracle source: integer outputs must be equal, images within 5e-5.  GPU only; test tool.

    python tools/oracle_device_check.py [batch=8]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import torch_ref as R  # noqa: E402
from text2human_amd import defaults, options, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DEV = 'cuda'


def to_dev(o, dev):
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: to_dev(v, dev) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(to_dev(v, dev) for v in o)
    return o


def timed(fn, sync=False):
    t0 = time.time()
    with torch.no_grad():
        r = fn()
    if sync:
        torch.cuda.synchronize()
    return r, time.time() - t0


opt = options.dict_to_nonedict(defaults.sample_from_parsing())
sds = synthetic.make_state_dicts(opt, seed=1234)
sds_d = to_dev(sds, DEV)
batch = synthetic.parsing_batch(B, seed=2021)
g = torch.Generator().manual_seed(5)
mask = batch['texture_mask']
tex = R.texture_tokens(mask, (32, 16)).view(B, -1)
idx = torch.randint(0, 1024, (B, 512), generator=g)
top = [torch.where(tex == h, idx, torch.full_like(idx, -1)).view(-1) for h in range(18)]

tok_c, t_c = timed(lambda: R.segm_tokens(batch['segm'], sds['segm_encoder'], sds['segm_quant_conv'], sds['segm_quantizer']['embedding.weight']))
f = lambda: R.segm_tokens(batch['segm'].to(DEV), sds_d['segm_encoder'], sds_d['segm_quant_conv'], sds_d['segm_quantizer']['embedding.weight'])
tok_g, t_g0 = timed(f, True)
_, t_g1 = timed(f, True)
print(f'segm_tokens B={B}: cpu {t_c:.2f} s | cuda cold {t_g0:.2f} s warm {t_g1:.3f} s | tokens differ {int((tok_c != tok_g.cpu()).sum())} of {tok_c.numel()}', flush=True)

(img_c, in_c), t_c = timed(lambda: R.refine_and_decode(top, mask, sds))
f = lambda: R.refine_and_decode(to_dev(top, DEV), mask.to(DEV), sds_d)
(img_g, in_g), t_g0 = timed(f, True)
_, t_g1 = timed(f, True)
bc, bg = torch.stack(in_c['bot_idx']), torch.stack(in_g['bot_idx']).cpu()
print(f'refine_and_decode B={B}: cpu {t_c:.2f} s | cuda cold {t_g0:.2f} s warm {t_g1:.3f} s | bottom indices differ {int((bc != bg).sum())} of '
      f'{int((bc >= 0).sum())} | image max abs diff {(img_c - img_g.cpu()).abs().max().item():.3e}', flush=True)

up = lambda t: torch.nn.functional.interpolate(t, scale_factor=2, mode='nearest')
tq, qb = in_c['top_quant'][:1], in_c['quant_bot'][:1]
f_c = lambda: R.decoder(up(tq), sds['decoder'], bot_h=R.decoder_res(up(qb), sds['bot_decoder_res']))
f_g = lambda: R.decoder(up(tq.to(DEV)), sds_d['decoder'], bot_h=R.decoder_res(up(qb.to(DEV)), sds_d['bot_decoder_res']))
d_c, t_c = timed(f_c)
d_g, t_g0 = timed(f_g, True)
_, t_g1 = timed(f_g, True)
print(f'1024x512 decoder, one image: cpu {t_c:.2f} s | cuda cold {t_g0:.2f} s warm {t_g1:.3f} s | max abs diff {(d_c - d_g.cpu()).abs().max().item():.3e} '
      f'on values up to {d_c.abs().max().item():.2f}', flush=True)

opt_p = options.dict_to_nonedict(defaults.sample_from_pose())
sds_p = synthetic.make_state_dicts(opt_p, seed=4321)
sds_pd = to_dev(sds_p, DEV)
pb = synthetic.pose_batch(B, seed=8)
args = lambda d, dev: (to_dev(pb['densepose'], dev), to_dev(pb['shape_attr'], dev), d['shape_embedder'], d['shape_encoder'], d['shape_decoder'],
                       opt_p['shape_attr_class_num'])
(seg_c, lg_c), t_c = timed(lambda: R.parsing_from_pose(*args(sds_p, 'cpu')))
(seg_g, lg_g), t_g0 = timed(lambda: R.parsing_from_pose(*args(sds_pd, DEV)), True)
_, t_g1 = timed(lambda: R.parsing_from_pose(*args(sds_pd, DEV)), True)
print(f'parsing_from_pose B={B}: cpu {t_c:.2f} s | cuda cold {t_g0:.2f} s warm {t_g1:.3f} s | pixels differ {int((seg_c != seg_g.cpu()).sum())} of {seg_c.numel()} '
      f'| logits max abs diff {(lg_c - lg_g.cpu()).abs().max().item():.3e}', flush=True)
