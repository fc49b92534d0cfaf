"""The contract line of bench.py: the driver keeps the last 8 KB of stdout and parses the LAST line, so the line
must stay small whatever the detail dict holds (the 26.5 KB line left BENCH_r04.parsed null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _canned():
    """a full result dict as main() builds it: the round-4 driver-command run (26.5 KB as one line)"""
    return json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_driver_cmd.json')))


def test_compact_line_is_small_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert len(line) < 6000 and '\n' not in line
    c = json.loads(line)
    for k in CONTRACT:
        assert k in c, k
    assert c['value'] == round(full['value'], 4) or abs(c['value'] - full['value']) < 1e-4 * full['value']
    assert len(c['dtype']) <= 120
    assert set(c['config']) >= {'workload', 'global_batch', 'sample_steps'}
    rf = c['roofline']
    for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_useful', 'avg_launch_us',
              'avg_launch_us_rocprof', 'mfma_util_pmc', 'main_loop_shader_clock_ghz'):
        assert k in rf, k
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 2e-3 and rf['frac'] < 1
    cb = c['cpu_baseline']
    assert set(cb) == {'value', 'unit', 'cores', 'kind', 'sample'} and cb['kind'] in ('reference', 'port')
    assert c['exact_fp32_path']['value'] > 0 and c['parity']['tokens_equal'] is True
    for name in ('parsing_b32', 'pose', 'hires'):
        leg = c['other_configs'][name]
        assert leg['value'] > 0 and leg['ms_per_step'] > 0 and leg['steps'] >= 1 and 0 < leg['roofline_frac'] < 1
    assert c['rccl_world'] == 1 and c['launch_mode'].startswith('hipGraph')


def test_compact_line_survives_bloat():
    """whatever prose or tables a later round adds to the detail dict, the line does not grow"""
    full = _canned()
    full['roofline']['all_gemm_kernels'] = {f'k{i}': {'x': 'y' * 200} for i in range(200)}
    full['cpu_baseline']['sample'] = 'z' * 5000
    full['config']['workload'] = 'w' * 5000
    full['dtype'] = 'd' * 1000
    full['per_rank_ms_per_step'] = [1.23456789] * 8
    full['per_rank_image_checksum'] = [123456789.0] * 8
    line = bench.compact_line(full)
    assert len(line) < 6000
    assert json.loads(line)['value'] > 0


def test_compact_line_without_optional_parts():
    """N > 1 runs carry no cpu_baseline / exact path; a failed eager leg carries no roofline"""
    full = _canned()
    for k in ('roofline', 'cpu_baseline', 'exact_fp32_path', 'parity', 'eager_launches', 'stages', 'other_configs'):
        full.pop(k)
    c = json.loads(bench.compact_line(full))
    assert c['value'] > 0 and 'roofline' not in c


def test_committed_counter_summaries_belong_to_these_kernel_sources():
    """roofline.traffic is quoted from profiles/*_pmc_summary*.json only when that summary was taken from this tree's
    kernel sources (digest over csrc/ and include/t2h_hip.h).  An edit to a kernel file without a new counter pass
    turns the field null in the driver's line: this test says so before the round ends."""
    for cfg in ('parsing', 'pose', 'parsing_b32'):
        sd = bench.profile_side_data('gemm_split_kernel<x8>', cfg)
        assert sd['traffic'] is not None, sd.get('traffic_note')
        assert bench.kernel_src_digest() in sd['traffic_source']
        assert 20e6 < sd['traffic'] < 400e6 and 0.05 < sd['mfma_util_pmc'] < 1
        assert 5 < sd['avg_launch_us_rocprof'] < 200
    # a kernel the summaries do not describe gets no figures
    assert bench.profile_side_data('some_other_kernel', 'parsing') == {'traffic': None}


def test_sampler_utilisation_weights_the_cross_terms_by_their_instruction_rate():
    """with x8 operands the Linears' two cross terms run in ONE 8-bit instruction at twice the fp16
    rate, so a Linear multiply costs 2 fp16-time units, not 3; attention keeps 3.  Pinned on round 5's canned run
    (profiles/r05_bench_driver_cmd_detail.json: 1776 evaluations in 655.5 ms, quoted then as 0.293)."""
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_driver_cmd_detail.json')))
    sm = full['stages']['sampler']
    t, ev = sm['ms_per_step'] * 1e-3, sm['sample_steps_evaluated']
    off = bench.sampler_matrix_time_frac(ev, t, x8=False)
    on = bench.sampler_matrix_time_frac(ev, t, x8=True, pmc=0.18)
    assert abs(off['matrix_time_frac'] - sm['executed_frac_of_16bit_peak']) < 1e-3      # fp16 planes: the old figure
    lin, att = bench.GFLOP_SAMPLER_LINEARS, bench.GFLOP_SAMPLER_ATTN
    assert abs(lin - 77.31) < 0.01 and abs(att - 12.88) < 0.01 and abs(lin + att - bench.GFLOP_SAMPLER_LAYERS) < 1e-3
    want = sm['executed_frac_of_16bit_peak'] * (2 * lin + 3 * att) / (3 * (lin + att))
    assert abs(on['matrix_time_frac'] - want) < 1e-3 and 0.205 < on['matrix_time_frac'] < 0.22   # "0.21", not 0.29
    assert on['matrix_time_units'] == {'linears': 2.0, 'attention': 3.0} and on['mfma_util_pmc_gemms'] == 0.18
    # and the stage view carries it (x8 on / off), next to the old three-products figure under its own name
    stats = dict(sample_steps_possible=2048, sample_steps_needed=1776, sample_steps_launched=ev, rounds=226)
    for x8, f in ((True, on), (False, off)):
        st = bench.stage_view({'sampler': sm['ms_per_step']}, 8, 256, False, stats, x8=x8)['sampler']
        assert abs(st['matrix_time_frac'] - f['matrix_time_frac']) < 1e-9
        assert abs(st['executed_products_frac_of_16bit_peak'] - sm['executed_frac_of_16bit_peak']) < 1e-3
    c = json.loads(bench.compact_line({**full, 'stages': {'sampler': {**st, 'ms_per_step': 1.0}}}))
    assert 'matrix_time_frac' in c['sampler'] and 'executed_frac_of_16bit_peak' not in c['sampler']


def test_worst_instantiation_of_the_compact_line():
    """roofline.worst_instantiation{name, frac, mfma_util_pmc, share}: the tile configuration furthest below its peak
    among those carrying >= 5 % of the kernel time -- on round 5's canned run q|k|v's 128x192 ping-pong tile (live
    frac 0.187, 18 % of the kernel time; proj / fc2's 128x64 K-split tile follows at 0.199 with 30 %)."""
    full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_driver_cmd_detail.json')))
    inst = {k: v for k, v in full['roofline']['all_gemm_kernels'].items() if 'frac' in v}
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r05_pmc_summary.json')))
    ks = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_parsing_kernel_stats.json')))
    side = {'per_kernel': {}}
    for r in pmc['rows']:
        if r['kernel'].startswith('gemm_split'):
            side['per_kernel'][r['kernel']] = {'mfma_util': r['mfma_util']}
    for r in ks['rows']:
        if r['kernel'].startswith('gemm_split'):
            side['per_kernel'].setdefault(r['kernel'], {})['share'] = r['total_ms'] / ks['total_kernel_ms']
    wi = bench.worst_instantiation(inst, side, x8=True)
    assert wi['name'] == 'gemm_split_kernel<128, 192, 4, 2, 1, 2, 1>' and abs(wi['frac'] - 0.187) < 1e-3
    assert abs(wi['mfma_util_pmc'] - 0.17) < 0.01 and abs(wi['share'] - 0.182) < 0.01
    # an instantiation below 5 % of the kernel time is not the one reported, however low its frac
    inst2 = dict(inst, **{'gemm_split_kernel<64x64>': {'frac': 0.01, 'avg_us': 5.0}})
    side['per_kernel']['gemm_split_kernel<64, 64, 2, 2, 1, 0, 1>'] = {'share': 0.001, 'mfma_util': 0.01}
    assert bench.worst_instantiation(inst2, side, x8=True)['name'] == wi['name']
    # no committed profile of these sources: every instantiation counts, the side fields are null
    wi0 = bench.worst_instantiation(inst, {}, x8=True)
    assert wi0['name'] == wi['name'] and wi0['share'] is None and wi0['mfma_util_pmc'] is None
    full['roofline']['worst_instantiation'] = wi
    c = json.loads(bench.compact_line(full))
    assert set(c['roofline']['worst_instantiation']) == {'name', 'frac', 'mfma_util_pmc', 'share'}
    assert len(bench.compact_line(full)) < 6000
