"""Dump per-kernel averages of every counter found in rocprofv3 --pmc result DBs.

    python tools/pmc_dump.py gpurun_out/pmc2/*/p_results.db > gpurun_out/pmc2/summary.txt
    PMC_DUMP_FILTER=conv,gn_apply python tools/pmc_dump.py ...     (kernel-name substrings; default gemm,mha)
"""
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:60]


def main(paths):
    rows = {}
    for db in paths:
        try:
            cur = sqlite3.connect(db).cursor()
            q = cur.execute('select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) '
                            'from counters_collection group by kernel_name, grid_size, counter_name')
            for kn, grid, cn, n, avg, dur in q:
                rows.setdefault((short(kn), grid), {})[cn] = (n, avg, dur / 1e3)
        except Exception as e:  # a pass whose counter is not available on this box
            print(f'# {db}: {e}')
    for (k, grid), cs in sorted(rows.items()):
        if not any(f in k for f in os.environ.get('PMC_DUMP_FILTER', 'gemm,mha').split(',')):
            continue
        print(f'{k} grid={grid}')
        for cn, (n, avg, us) in sorted(cs.items()):
            print(f'    {cn:34s} n={n:4d} avg={avg:16.1f} dur_us={us:8.1f}')


if __name__ == '__main__':
    main(sys.argv[1:])
