"""YAML option surface of the sampling path.

Mirrors the contract of the reference's utils/options.py:33-129 (same keys in,
same derived keys out) so `configs/sample_from_parsing.yml` /
`sample_from_pose.yml` parse unchanged:

* ``parse(path, is_train=False)`` -> ordered dict, plus ``is_train`` and
  ``opt['path'][root|results_root|log|visualization]`` (utils/options.py:56-79)
* ``dict_to_nonedict`` -> missing keys read as ``None`` (utils/options.py:105-129)
* ``gpu_ids`` is only echoed unless ``set_CUDA_VISIBLE_DEVICES`` is truthy
  (utils/options.py:47-52); on ROCm the variable that matters is
  ``HIP_VISIBLE_DEVICES`` and it is set alongside.
"""
import os
import os.path as osp
from collections import OrderedDict

import yaml


class NoneDict(dict):
    """dict whose missing keys read as None."""

    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict((k, dict_to_nonedict(v)) for k, v in opt.items())
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def _ordered_load(stream):

    class _Loader(yaml.SafeLoader):
        pass

    def _construct(loader, node):
        loader.flatten_mapping(node)
        return OrderedDict(loader.construct_pairs(node))

    _Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                            _construct)
    return yaml.load(stream, Loader=_Loader)


def parse(opt_path, is_train=False, root=None):
    with open(opt_path, 'r') as f:
        opt = _ordered_load(f)

    gpu_list = ','.join(str(x) for x in opt.get('gpu_ids', []) or [])
    if opt.get('set_CUDA_VISIBLE_DEVICES', None):
        os.environ['CUDA_VISIBLE_DEVICES'] = gpu_list
        os.environ['HIP_VISIBLE_DEVICES'] = gpu_list
        print('export CUDA_VISIBLE_DEVICES=' + gpu_list, flush=True)
    else:
        print('gpu_list: ', gpu_list, flush=True)

    opt['is_train'] = is_train
    root = root or osp.abspath(os.getcwd())
    paths = OrderedDict(root=root)
    if is_train:
        raise NotImplementedError(
            'text2human_amd implements the sampling path only '
            '(training is out of scope, see DESIGN.md)')
    results_root = osp.join(root, 'results', opt['name'])
    paths['results_root'] = results_root
    paths['log'] = results_root
    paths['visualization'] = osp.join(results_root, 'visualization')
    opt['path'] = paths
    return opt


def dict2str(opt, indent_level=1):
    msg = ''
    pad = ' ' * (indent_level * 2)
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += f'{pad}{k}:[\n{dict2str(v, indent_level + 1)}{pad}]\n'
        else:
            msg += f'{pad}{k}: {v}\n'
    return msg


def make_exp_dirs(opt):
    """utils/util.py:13-22: raises FileExistsError if results_root exists."""
    os.makedirs(opt['path']['results_root'])


def set_random_seed(seed):
    """utils/util.py:25-31."""
    import random

    import numpy as np
    import torch
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
