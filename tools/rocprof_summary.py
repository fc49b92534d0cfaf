"""Condenses a rocprofv3 (rocpd sqlite) result into a small per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof2/r2_results.db profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name)
    base = m.group(1) if m else name
    if 'distribution_elementwise_grid_stride_kernel' in name:
        base = 'at::native::distribution_kernel(' + ('exponential' if 'exponential' in name else 'uniform') + ')'
    return base[:110]


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute(
        'select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
        'max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name'))
    total = sum(r[2] for r in rows)
    agg = {}
    for name, n, tot, avg, mn, mx, vg, ag, lds in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0, vg, ag, lds])
        a[0] += n
        a[1] += tot
        a[2] = min(a[2], mn)
        a[3] = max(a[3], mx)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |',
             '|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|']
    for k, (n, tot, mn, mx, vg, ag, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'| `{k}` | {n} | {tot / 1e6:.2f} | {tot / n / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | '
                     f'{100 * tot / total:.2f} | {vg} | {ag} | {lds} |')
    hdr = f'# rocprofv3 --kernel-trace --stats summary\n\nsource: `{db_path}`; total kernel time {total / 1e6:.1f} ms\n\n'
    open(out_path, 'w').write(hdr + '\n'.join(lines) + '\n')
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
