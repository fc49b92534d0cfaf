"""Per-kernel PMC summary from separate rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES/GRBM_GUI_ACTIVE), as MI355X_MICROARCH.md
prescribes: counters in their own passes; FETCH_SIZE/WRITE_SIZE are KiB;
on gfx950 FETCH_SIZE reports 1/2 of a wide (16 B/lane) coalesced read stream
-> read bytes = 2 * FETCH_SIZE * 1024 (WRITE_SIZE checked exact against the
known output bytes of the GEMMs: 4096x2048x4 B = 32768 KiB).

    python tools/pmc_summary.py gpurun_out profiles/r03_pmc_summary ["bench.py arguments of the passes"]

The JSON carries the digest of the kernel sources it was measured on (bench.kernel_src_digest) and
the git HEAD; bench.py quotes it only while that digest equals the tree's.
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
    return re.match(r'([A-Za-z0-9_:]+(<[^(]*>)?)', name).group(1)[:80]


def load(db, counters):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for kn, grid, cn, n, avg, dur in cur.execute(
            'select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) '
            'from counters_collection group by kernel_name, grid_size, counter_name'):
        if cn in counters:
            out[(short(kn), grid, cn)] = (n, avg, dur / 1e3)
    return out


def main(root, out_prefix, what='bench.py --steps 1 --warmup 1'):
    f = load(f'{root}/pmc_FETCH_SIZE/p_results.db', {'FETCH_SIZE'})
    w = load(f'{root}/pmc_WRITE_SIZE/p_results.db', {'WRITE_SIZE'})
    m = load(f'{root}/pmc_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db',
             {'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'})
    rows = []
    for (k, grid, _), (n, fetch, us) in f.items():
        wr = w.get((k, grid, 'WRITE_SIZE'), (0, 0.0, 0))[1]
        busy = m.get((k, grid, 'SQ_VALU_MFMA_BUSY_CYCLES'), (0, 0.0, 0))[1]
        act = m.get((k, grid, 'GRBM_GUI_ACTIVE'), (0, 0.0, 0))[1]
        util = busy / 1024.0 / (act / 8.0) if act else 0.0  # 1024 SIMDs; GUI_ACTIVE summed over 8 XCDs
        rows.append(dict(kernel=k, grid_threads=grid, launches=n, avg_us=us, fetch_kib=fetch, write_kib=wr,
                         traffic_mb=(2 * fetch + wr) * 1024 / 1e6, mfma_util=util, total_us=n * us))
    rows.sort(key=lambda r: -r['total_us'])
    rows = rows[:24]
    import bench
    head = subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
    sha = bench.kernel_src_digest()
    json.dump(dict(kernel_src_sha=sha, git_head=head, command=what, rows=rows), open(out_prefix + '.json', 'w'), indent=1)
    lines = [f'# rocprofv3 --pmc summary (separate passes; {what}; kernel sources {sha}, HEAD {head or "n/a"})', '',
             'traffic MB = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 FETCH correction); '
             'mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)', '',
             '| kernel | grid threads | launches | avg us | FETCH KiB | WRITE KiB | traffic MB | MFMA util |',
             '|---|---:|---:|---:|---:|---:|---:|---:|']
    for r in rows:
        lines.append(f"| `{r['kernel']}` | {r['grid_threads']} | {r['launches']} | {r['avg_us']:.1f} | "
                     f"{r['fetch_kib']:.0f} | {r['write_kib']:.0f} | {r['traffic_mb']:.1f} | {r['mfma_util']:.2f} |")
    open(out_prefix + '.md', 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[4:14]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
