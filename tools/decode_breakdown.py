"""Per-kernel account of the refine + decode stage ("5 ms of the stage's 18.2 are not the 3x3
convolutions and have no per-kernel account").

    rocprofv3 --kernel-trace -d DIR -o p -- python tools/decode_breakdown.py run [batch=8] [upscale=0]
    python tools/decode_breakdown.py summarize DIR/.../p_results.db profiles/r06_decode_breakdown.md [decodes=5]

`run`: refine + decode of `batch` images, 1 warm-up + 4 more, NOTHING else on the GPU (no tokenizer, no sampler: the
texture tokens and top indices are synthetic).  `summarize`: kernels grouped by (name, grid) and by role, microseconds
per decode of the batch."""
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_DECODES = 5


def run(B, up):
    import torch
    from text2human_amd import defaults, options, synthetic
    from text2human_amd.models import SampleFromParsingModel
    opt = options.dict_to_nonedict(defaults.sample_from_parsing())
    sds = synthetic.make_state_dicts(opt, seed=1234)
    batch = synthetic.parsing_batch(B, seed=2021)
    model = SampleFromParsingModel(opt, state_dicts=sds)
    model.texture_mask = batch['texture_mask'].to(model.device)
    model.batch_size = B
    g = torch.Generator().manual_seed(3)
    tex = model._texture_tokens(model.texture_mask)
    idx = torch.randint(0, 1024, (B, 512), generator=g).cuda()
    top = [torch.where(tex == h, idx, torch.full_like(idx, -1)) for h in range(18)]
    torch.cuda.synchronize()
    for _ in range(N_DECODES):
        model.decode_indices(top, want_u8=True, upscale=up)
    torch.cuda.synchronize()


ROLES = (('conv_halo', '3x3 convolutions, halo-staged (large levels)'),
         ('conv_split', '3x3 (small levels) and 1x1 convolutions, split-precision implicit GEMM'),
         ('gn_apply_split', 'GroupNorm apply + swish + split (small levels, 1x1 operands)'),
         ('gn_finalize', 'GroupNorm finalize (statistics -> tables)'),
         ('gn_partial', 'GroupNorm statistics pass (inputs without a producing conv epilogue)'),
         ('conv3x3_small', 'conv_out (128 -> 3 channels), vector ALU'),
         ('softmax_rows', 'AttnBlock softmax (materialised form)'),
         ('gemm_kernel', 'exact-fp32 GEMMs: AttnBlock q k^T / p v (batched), index-prediction UNet + heads, post-quant 1x1s'),
         ('spatial_attention', 'AttnBlock attention, flash-style fallback'),
         ('codebook_gather', 'codebook gathers (top / bottom, fold)'),
         ('routed_head_argmax', 'index-prediction heads: routed 1x1 + argmax'),
         ('maxpool2', 'UNet max-pool'), ('bilinear_up2', 'UNet bilinear x2'), ('image_epilogue', 'image epilogue (clamp, uint8)'))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:80]


def summarize(db_path, out_path, n_dec):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    grid = 'grid_size' if 'grid_size' in cols else ('grid_x * grid_y * grid_z' if 'grid_x' in cols else '0')
    # only what lies between the FIRST and the LAST decode's image epilogue (one per decode): the model's construction
    # (weight packing, the x8 calibration of the sampler) and the first decode's warm-up are not the stage's
    ts = 'start' if 'start' in cols else 'start_timestamp'
    marks = [r[0] for r in db.execute(f"select {ts} from kernels where name like '%image_epilogue%' order by {ts}")]
    assert marks and len(marks) % n_dec == 0, (len(marks), n_dec)  # (a chunked decode has one epilogue per chunk)
    marks = marks[len(marks) // n_dec - 1:]
    n_dec -= 1
    rows = list(db.execute(f'select name, {grid}, count(*), sum(duration), avg(duration) from kernels '
                           f'where {ts} > {marks[0]} and {ts} <= {marks[-1]} group by name, {grid}'))
    total = sum(r[3] for r in rows)
    by_role, detail = {}, []
    for name, gsz, n, tot, avg in rows:
        k = short(name)
        role = next((d for key, d in ROLES if key in k), 'other (torch copies / fills, layout)')
        a = by_role.setdefault(role, [0, 0.0])
        a[0] += n
        a[1] += tot
        detail.append((tot, k, gsz, n, avg, role))
    import bench
    out = [f'# refine + decode of one batch: per-kernel account\n\nsource: `{db_path}` (rocprofv3 --kernel-trace over '
           f'`tools/decode_breakdown.py run`: {n_dec + 1} decodes, the first left out, nothing else on the GPU); kernel sources {bench.kernel_src_digest()}; '
           f'kernel time per decode {total / n_dec / 1e6:.2f} ms\n\n## by role (per decode of the batch)\n\n'
           '| role | launches | us | % |\n|---|---:|---:|---:|']
    for role, (n, tot) in sorted(by_role.items(), key=lambda kv: -kv[1][1]):
        out.append(f'| {role} | {n / n_dec:.0f} | {tot / n_dec / 1e3:.0f} | {100 * tot / total:.1f} |')
    out.append('\n## by kernel and grid (per decode; rows above 0.3 %)\n\n| kernel | grid (threads) | launches | avg us | us per decode | % |\n'
               '|---|---:|---:|---:|---:|---:|')
    for tot, k, gsz, n, avg, role in sorted(detail, reverse=True):
        if tot / total < 0.003:
            continue
        out.append(f'| `{k}` | {gsz} | {n / n_dec:.1f} | {avg / 1e3:.1f} | {tot / n_dec / 1e3:.0f} | {100 * tot / total:.1f} |')
    open(out_path, 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out[:24]))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8, bool(int(sys.argv[3])) if len(sys.argv) > 3 else False)
    else:
        summarize(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else N_DECODES)
