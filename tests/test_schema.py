"""Checkpoint / config surface."""
import json
import os

import torch

from text2human_amd import defaults, options, synthetic

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def test_schema_matches_reference_dump():
    ref = json.load(open(os.path.join(GOLD, 'state_dict_schema.json')))
    mine = synthetic.module_schemas(defaults.sample_from_pose())
    for name, want in ref.items():
        got = {k: list(v[0]) for k, v in mine[name].items()}
        assert list(got.keys()) == list(want.keys()), name
        assert got == want, name


def test_yaml_roundtrip_and_nonedict(tmp_path):
    p = defaults.write_yaml(defaults.sample_from_parsing(), str(tmp_path / 'c.yml'))
    opt = options.parse(p, is_train=False, root=str(tmp_path))
    assert opt['model_type'] == 'SampleFromParsingModel'
    assert opt['path']['results_root'].endswith('results/sample_from_parsing')
    nd = options.dict_to_nonedict(opt)
    assert nd['no_such_key'] is None and nd['path']['nope'] is None
    assert nd['top_ch_mult'] == [1, 1, 2, 2, 4] and nd['sample_steps'] == 256


def test_checkpoint_files_layout(tmp_path):
    opt = defaults.sample_from_pose()
    # shrink nothing: layout test only inspects keys of the small files
    o = synthetic.write_checkpoints(opt, str(tmp_path), seed=1)
    top = torch.load(o['top_vae_path'], weights_only=False)
    assert set(top) == {'decoder', 'quantize', 'post_quant_conv'}
    bot = torch.load(o['bot_vae_path'], weights_only=False)
    assert set(bot) == {'bot_decoder_res', 'decoder', 'bot_quantize', 'bot_post_quant_conv'}
    assert set(torch.load(o['segm_token_path'], weights_only=False)) == {'encoder', 'quantize', 'quant_conv'}
    assert set(torch.load(o['pretrained_index_network'], weights_only=False)) == {'guidance_encoder', 'index_decoder'}
    assert 'head_list.17.weight' in torch.load(o['pretrained_sampler'], weights_only=False)
    assert set(torch.load(o['pretrained_parsing_gen'], weights_only=False)) == {'embedder', 'encoder', 'decoder'}
