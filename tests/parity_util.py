"""Teacher-forced sampler parity machinery shared by the -m gpu parity tests (bench configuration,
parsing B = 32, pose B = 32).  Method and vocabulary: tests/test_gpu_bench_parity.py's docstring.

The oracle (oracle/torch_ref.py) runs as eager PyTorch-ROCm fp32 on the same GPU and generator; the
HIP sampler is forced onto the oracle's trajectory, so every categorical decision is compared on
identical inputs; every mismatch is accounted for (`account`) instead of being averaged away."""
import torch

from oracle import torch_ref as R
from text2human_amd import engine

DEV = 'cuda'
ACT_TOL = 2e-4  # activations, on O(1) values

# Where the ORACLE's convolutional stages (tokenizer, refine + decode, pose front end, encode side) execute in the
# -m gpu tests.  Round 6: 'cuda' -- the same oracle source (oracle/torch_ref.py) as eager
# PyTorch-ROCm fp32 on the box's GPU: refine + decode of 8 images 0.07 s instead of 14 s on the host cores, 32 images /
# the 1024 x 512 decoder in proportion (profiles/r06_oracle_device_check.log) -- the driver's GPU step was mostly the
# oracle's CPU side.  The chain stays pinned: oracle on the CPU == the unmodified reference (tests/
# test_oracle_vs_reference.py, CPU suite); oracle on cuda:0 == oracle on the CPU (tests/test_gpu_oracle_device.py:
# integer outputs equal, images within 5e-5); HIP path vs oracle on cuda:0 (everything else).
# T2H_TEST_ORACLE_DEVICE=cpu runs every comparison against the CPU execution again (slow; the arbiter if the two
# executions of the oracle ever disagree on a near-tie).
import os
ORACLE_DEV = os.environ.get('T2H_TEST_ORACLE_DEVICE', 'cuda')
_SDS_ON_DEV = {}


def odev(o, dev=None):
    """tensors / dicts / lists of tensors -> the oracle's device"""
    dev = dev or ORACLE_DEV
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, dict):
        return {k: odev(v, dev) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(odev(v, dev) for v in o)
    return o


def osds(sds):
    """the state dicts on the oracle's device (one copy per dict object, kept while the dict lives in the cache)"""
    if ORACLE_DEV == 'cpu':
        return sds
    hit = _SDS_ON_DEV.get(id(sds))
    if hit is None or hit[0] is not sds:
        if len(_SDS_ON_DEV) >= 4:
            _SDS_ON_DEV.clear()
        hit = _SDS_ON_DEV[id(sds)] = (sds, odev(sds))
    return hit[1]


class RecordingNoise(R.TorchNoise):
    """TorchNoise that remembers the device generator state at the start of every step."""

    def __init__(self, device):
        super().__init__(device)
        self.state = {}

    def uniform(self, step, shape):
        self.state[step] = torch.cuda.get_rng_state(self.device)
        return super().uniform(step, shape)


def seed_all(s):
    torch.manual_seed(s)
    torch.cuda.manual_seed_all(s)


def oracle_run(segm_tokens, texture_mask, sd_dev, steps, seed, temp=1.0):
    """-> (list of 18 token tensors, {t: trace dict}, {t: generator state at the start of step t})"""
    noise, trace = RecordingNoise(DEV), []
    seed_all(seed)
    with torch.no_grad():
        ref = R.sample_fn(segm_tokens, texture_mask.to(DEV), sd_dev, sample_steps=steps, temp=temp, noise=noise, trace=trace)
    return ref, {d['t']: d for d in trace}, noise.state


def forced_run(model, trace, steps, seed, compact=True, temp=1.0):
    """HIP sampler on the oracle's trajectory; -> (list of (t, row, ours, oracle's), schedule stats).

    compact=False: one round per step, all samples at that step (the reference's loop).
    compact=True (the default schedule of the product path): every sample walks through its OWN
    active steps, so a round holds samples at different steps; sample b at step t must agree with --
    and is then forced to -- the oracle's state of sample b after step t."""
    mism = []
    T = trace[steps]['x_t'].shape[1]

    def step_hook(t, x_t, out):
        want = trace[t]['x_t']
        for b, j in (x_t != want).nonzero().tolist():
            mism.append((t, b * T + j, int(x_t[b, j]), int(want[b, j])))
        x_t.copy_(want)

    def round_hook(r, st, x_t, out):
        st_l = st.tolist()
        want = torch.stack([trace[t]['x_t'][b] if t else x_t[b] for b, t in enumerate(st_l)])
        for b, j in (x_t != want).nonzero().tolist():
            mism.append((st_l[b], b * T + j, int(x_t[b, j]), int(want[b, j])))
        x_t.copy_(want)

    seed_all(seed)
    tex_tok = model._texture_tokens(model.texture_mask)
    kw = dict(round_hook=round_hook, compact=True) if compact else dict(step_hook=step_hook)
    engine.sample_tokens(model.sampler_fn, model.segm_tokens.contiguous(), tex_tok, steps, model.mask_id, temp=temp, **kw)
    return mism, dict(model.sampler_fn.last_stats)


def account(model, sd_dev, texture_mask, trace, rng_state, mism, steps, scale=1.0):
    """gap / dl of every mismatch: with the step's noise re-drawn from the recorded generator state,
    `gap` = log-ratio by which the oracle's own arithmetic prefers its token over ours, `dl` = max |logit
    difference| between the two implementations on that row; "explained" iff gap <= 2 dl and dl is
    inside the activation tolerance."""
    rows = []
    tex_tok = R.texture_tokens(texture_mask.cpu(), (32, 16)).to(DEV)
    n = tex_tok.numel()
    B = tex_tok.shape[0]
    for t, row, ours, theirs in mism:
        prev = trace[t + 1]['x_t'] if t < steps else torch.full_like(trace[t]['x_t'], model.mask_id)
        head = int(tex_tok.view(-1)[row])
        with torch.no_grad():
            lo = R.transformer_logits(prev, model.segm_tokens, tex_tok, sd_dev, heads={head})[head]
        lm = model.sampler_fn.logits(prev, model.segm_tokens.contiguous(), tex_tok, heads={head})[head]
        lo_r, lm_r = lo.reshape(n, -1)[row].double(), lm.reshape(n, -1)[row].double()
        torch.cuda.set_rng_state(rng_state[t], DEV)
        torch.rand((B, n // B), device=DEV)
        expo = None
        for cb in trace[t]['active']:
            e = torch.empty((n, lo_r.numel()), device=DEV).exponential_(1.0)
            if cb == head:
                expo = e[row].double()
        score = torch.log_softmax(lo_r, -1) - expo.log()
        a, c = theirs - 1024 * head, ours - 1024 * head
        rows.append(dict(step=t, row=row, head=head, ours=c, oracle=a,
                         gap=float(score[a] - score[c]), dl=float((lo_r - lm_r).abs().max()),
                         logit_range=float(lo_r.max() - lo_r.min())))
    for r in rows:
        r['explained'] = bool(r['gap'] <= 2.0 * r['dl'] + 1e-7 and r['dl'] <= ACT_TOL * scale)
    return rows


def vq_mismatch_accounting(z_hip, z_ref, codebook, tok_hip, tok_ref):
    """Codebook argmin decisions that differ between the HIP tokenizer and the oracle, each accounted for.

    z_hip / z_ref: latent rows [n, D] the two implementations quantise (fp32); tokens [n].  For a row whose
    tokens differ (oracle picks a, HIP picks c) the oracle's latent gives, in fp64, gap = |z - e_c|^2 -
    |z - e_a|^2 >= 0.  HIP chose c on ITS latent z + e, so gap + 2 e.(e_a - e_c) <= 0 up to the rounding of
    the fp32 distance evaluation: the mismatch is a near-tie iff gap <= 2 |e| |e_a - e_c| + slack, with |e|
    the measured latent error of THAT row and slack = 8 fp32 ulps of the distance terms (both
    implementations evaluate the expanded form z.z + e.e - 2 z.e in fp32).
    -> list of dicts (row, ours, oracle, gap, bound, explained)."""
    z_hip, z_ref, cb = z_hip.double().cpu(), z_ref.double().cpu(), codebook.double().cpu()
    tok_hip, tok_ref = tok_hip.reshape(-1).cpu(), tok_ref.reshape(-1).cpu()
    out = []
    eps = 2.0**-23
    for row in (tok_hip != tok_ref).nonzero().flatten().tolist():
        z, e = z_ref[row], z_hip[row] - z_ref[row]
        a, c = int(tok_ref[row]), int(tok_hip[row])
        da, dc = float(((z - cb[a])**2).sum()), float(((z - cb[c])**2).sum())
        mag = float((z**2).sum() + max((cb[a]**2).sum(), (cb[c]**2).sum()) + 2 * z.abs() @ torch.maximum(cb[a].abs(), cb[c].abs()))
        bound = 2.0 * float(e.norm()) * float((cb[a] - cb[c]).norm()) + 8 * eps * mag
        out.append(dict(row=row, ours=c, oracle=a, gap=dc - da, bound=bound, latent_err=float(e.abs().max()),
                        explained=bool(dc - da <= bound)))
    return out


def balanced_pose_state_dicts(sds, opt, n=2, seed=99):
    """Synthetic sample_from_pose weights whose parsing generator gives NON-degenerate maps.

    With plain random weights the 24 class logits of the FCNHead differ by ~0.1 in their means and by
    ~1e-3 over space, so every pose maps to one constant class -- and a constant map is an ill-conditioned
    input for the tokenizer (GroupNorm of a constant plane amplifies rounding differences by orders of
    magnitude: 1e-2 latent differences between two correct fp32 implementations).  Here the head's bias is
    re-centred so that the class means are equal on a small calibration batch (the CPU oracle computes
    them): the argmax is then decided by the spatial variation and the map has all 24 classes.  Returns a
    shallow copy of `sds` with a new 'shape_decoder'."""
    from text2human_amd import synthetic
    pb = synthetic.pose_batch(n, seed=seed)
    with torch.no_grad():
        _, logits = R.parsing_from_pose(pb['densepose'], pb['shape_attr'], sds['shape_embedder'], sds['shape_encoder'],
                                        sds['shape_decoder'], opt['shape_attr_class_num'])
    out = dict(sds)
    out['shape_decoder'] = dict(sds['shape_decoder'])
    out['shape_decoder']['conv_seg.bias'] = sds['shape_decoder']['conv_seg.bias'] - logits.mean((0, 2, 3))
    return out
