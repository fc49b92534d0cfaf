"""The bench-configuration parity case (tests/test_gpu_bench_parity.py::_full_parity: B = 8, 256 steps, tokenizer exact,
sampler teacher-forced on the oracle's trajectory with every mismatch accounted, free-running tokens, bottom indices,
images) at MORE sampling seeds than the suite's 2021, default and x50-peaked weights -- evidence, not a test: the
equalities the suite asserts are properties of float arithmetic on the tested seeds, not identities.

    python tools/parity_more_seeds.py [seeds=1,7,99,12345] -> gpurun_out/parity_more_seeds.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
import test_gpu_bench_parity as T  # noqa: E402

seeds = [int(s) for s in (sys.argv[1].split(',') if len(sys.argv) > 1 else '1,7,99,12345'.split(','))]
out = []
for seed in seeds:
    for peaked in (False, True):
        T.SEED = seed
        try:
            r = T._full_parity(8, peaked, f'more_seeds_{seed}_{"peaked" if peaked else "default"}')
            row = dict(seed=seed, peaked=peaked, ok=True, segm_token_mismatches=r['segm_token_mismatches'],
                       segm_token_accounting=r.get('segm_token_accounting', []),
                       paths={k: dict(mismatches=v['mismatches'], unexplained=len([a for a in v['accounted'] if not a['explained']]))
                              for k, v in r['paths'].items()},
                       free_running=r['free_running'], decode=r['decode'])
        except AssertionError as e:
            row = dict(seed=seed, peaked=peaked, ok=False, error=str(e)[:500])
        out.append(row)
        print(json.dumps(row), flush=True)
with open(os.path.join(ROOT, 'gpurun_out', 'parity_more_seeds.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(f'{sum(r["ok"] for r in out)} of {len(out)} cases passed every assertion')
