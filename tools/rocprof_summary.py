"""Condenses a rocprofv3 (rocpd sqlite) result into a small per-kernel table.

    python tools/rocprof_summary.py gpurun_out/prof2/r2_results.db profiles/r03_bench_parsing_kernel_stats.md

Writes the .md table and, beside it, a .json with the same rows plus the digest of the kernel sources
(bench.kernel_src_digest) and the git HEAD, which bench.py checks before quoting a duration.
"""
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    import bench
    head = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
    sha = bench.kernel_src_digest()
    hdr = (f'# rocprofv3 --kernel-trace --stats summary\n\nsource: `{db_path}`; total kernel time {total / 1e6:.1f} ms; '
           f'kernel sources {sha}, HEAD {head or "n/a"}\n\n')
    open(out_path, 'w').write(hdr + '\n'.join(lines) + '\n')
    rows_j = [dict(kernel=k, calls=n, total_ms=tot / 1e6, avg_us=tot / n / 1e3, min_us=mn / 1e3, max_us=mx / 1e3)
              for k, (n, tot, mn, mx, vg, ag, lds) in sorted(agg.items(), key=lambda kv: -kv[1][1])]
    json.dump(dict(kernel_src_sha=sha, git_head=head, total_kernel_ms=total / 1e6, rows=rows_j),
              open(os.path.splitext(out_path)[0] + '.json', 'w'), indent=1)
    print('\n'.join(lines[:14]))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
