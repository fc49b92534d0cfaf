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
