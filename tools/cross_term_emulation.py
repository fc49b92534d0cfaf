"""VERDICT r04 item 6, the accuracy side, on the CPU (no kernel needed to answer it): what do cheaper cross terms do
to the sampler's logits?  The split GEMM computes a.b = ah.bh + 2^-11 (ah.bl + al.bh) with all planes fp16
(csrc/gemm_split.hip); two thirds of its matrix instructions are the cross terms.  Candidates that would run them
faster: the fp8 matrix instructions (2x the fp16 rate; e4m3 planes with a per-(row, 32-wide K tile) scale), or
narrower planes in general.  This tool runs ONE transformer evaluation of the benchmark model (24 layers, the
oracle's own forward, oracle/torch_ref.py) with every Linear replaced by an fp64 emulation of

    ah.bh (fp16 planes, exact)  +  2^-11 ( q(ah).q(bl) + q(al).q(bh) ),     q = the candidate format

and reports the max / rms difference of the final hidden state and of the logits of one head against the exact
fp32 forward -- next to the same figure for the shipped arithmetic (q = identity: fp16 planes) and the tolerance the
parity tests allow (ACT_TOL = 2e-4, tests/parity_util.py).  A sampled token flips when the logit error exceeds
the gap of the two leading Gumbel scores; with the default weights that gap has density O(1) near 0, so the
expected number of flipped decisions per 4096 is about 4096 x (rms logit error x sqrt 2)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_ref as R  # noqa: E402
from text2human_amd import defaults, options, synthetic  # noqa: E402

LO = 2048.0


def planes(x):
    h = x.to(torch.float16)
    l = ((x - h.float()) * LO).to(torch.float16)
    return h.double(), l.double()


def q_identity(p):
    return p


def q_e4m3(p):
    """fp8 e4m3 with one scale per (row, 32-wide K tile): max |x| of the tile -> 448 (the format's maximum)"""
    r, k = p.shape
    t = p.view(r, k // 32, 32)
    s = t.abs().amax(-1, keepdim=True).clamp_min(1e-30) / 448.0
    q = (t / s).float().to(torch.float8_e4m3fn).double() * s
    return q.view(r, k)


def q_e5m2(p):
    """fp8 e5m2 (bf8), no scale: fp16's exponent range with 2 mantissa bits"""
    return p.float().to(torch.float8_e5m2).double()


def q_mx(emax, mbits, vmax):
    """OCP MX block format: one power-of-two scale per (row, 32-wide K tile) = 2^(floor(log2 max) - emax), elements
    with `mbits` mantissa bits, subnormals below 2^(1 - bias) (bias 1 for e2m3, 7 for e4m3), saturating at vmax"""
    emin = {2: 0, 8: -6}[emax]  # exponent of the smallest normal

    def q(p):
        r, k = p.shape
        t = p.view(r, k // 32, 32)
        mx = t.abs().amax(-1, keepdim=True).clamp_min(1e-300)
        s = torch.exp2(torch.floor(torch.log2(mx)) - emax)
        v = t / s
        e = torch.floor(torch.log2(v.abs().clamp_min(1e-300))).clamp_min(emin)
        step = torch.exp2(e - mbits)
        out = (torch.round(v / step) * step).clamp(-vmax, vmax) * s
        return out.view(r, k)
    return q


def q_bits(nbits):
    def q(p):  # keep `nbits` significant bits (round to nearest), exponent unbounded: an upper bound for any n-bit format
        m, e = torch.frexp(p)
        return torch.ldexp(torch.round(m * 2.0**nbits) / 2.0**nbits, e)
    return q


def make_linear(q):
    def linear(x, w, b=None):
        shp = x.shape
        xh, xl = planes(x.reshape(-1, shp[-1]).float())
        wh, wl = planes(w.float())
        y = xh @ wh.t() + (q(xh) @ q(wl).t() + q(xl) @ q(wh).t()) / LO
        if b is not None:
            y = y + b.double()
        return y.float().view(*shp[:-1], w.shape[0])
    return linear


def main():
    torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', '16')))
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    for scale in (1.0, 50.0):
        sds = synthetic.make_state_dicts(opt, seed=1234, head_scale=scale)
        sd = sds['sampler']
        batch = synthetic.parsing_batch(1, seed=2021)
        g = torch.Generator().manual_seed(3)
        tex = R.texture_tokens(batch['texture_mask'], (32, 16))
        seg = torch.randint(0, 1024, (1, 512), generator=g)
        idx = torch.where(torch.rand(1, 512, generator=g) < 0.5, torch.full((1, 512), 18432),
                          torch.randint(0, 1024, (1, 512), generator=g) + 1024 * tex)  # a half-unmasked state
        head = int(tex.view(-1)[0])
        with torch.no_grad():
            ref_h = R.transformer_hidden(idx, seg, tex, sd)
            ref_l = F.linear(ref_h, sd[f'head_list.{head}.weight'])
            print(f'head weights x{scale:g}: logits of head {head}: range {float(ref_l.max() - ref_l.min()):.2f}, '
                  f'std {float(ref_l.std()):.3f}', flush=True)
            real = F.linear
            variants = {'a': (('fp16 planes (shipped)', q_identity), ('fp8 e4m3, per-tile scale', q_e4m3),
                              ('8 significant bits (bf16-like)', q_bits(8)), ('6 significant bits', q_bits(6))),
                        'b': (('fp8 e5m2, no scale', q_e5m2), ('MXFP8 e4m3 (E8M0 block scale)', q_mx(8, 3, 448.0)),
                              ('MXFP6 e2m3 (E8M0 block scale)', q_mx(2, 3, 7.5)))}[os.environ.get('T2H_EMU_SET', 'a')]
            for name, q in variants:
                F.linear = make_linear(q)
                try:
                    h = R.transformer_hidden(idx, seg, tex, sd)
                finally:
                    F.linear = real
                lg = F.linear(h, sd[f'head_list.{head}.weight'])
                dh, dl = (h - ref_h).abs(), (lg - ref_l).abs()
                rms = float(dl.pow(2).mean().sqrt())
                print(f'  cross terms in {name:32s}: hidden max {float(dh.max()):.2e}  logits max {float(dl.max()):.2e} '
                      f'rms {rms:.2e}  -> ~{4096 * rms * 2**0.5:.2f} flipped decisions per 4096 (tolerance 2e-4 x {scale:g})',
                      flush=True)


if __name__ == '__main__':
    main()
