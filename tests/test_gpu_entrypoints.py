"""Entry points on a real dataset tree (-m gpu): YAML + .pth files +
DeepFashion-style directories -> `python -m text2human_amd.sample_from_parsing / _pose`
write one PNG per listed image, like the reference scripts."""
import os

import numpy as np
import pytest
from PIL import Image

from text2human_amd import defaults, sample_from_parsing, synthetic

pytestmark = pytest.mark.gpu


def _config(tmp_path, pose, name):
    opt = defaults.sample_from_pose() if pose else defaults.sample_from_parsing()
    opt = synthetic.write_checkpoints(opt, str(tmp_path / 'ckpt'), seed=1234)
    tree = synthetic.write_dataset_tree(str(tmp_path / 'data'), n=3, seed=11)
    names = tree.pop('names')
    opt.update(tree)
    opt.update(name=name, sample_steps=3, manual_seed=2021)
    return defaults.write_yaml(opt, str(tmp_path / f'{name}.yml')), names


@pytest.mark.parametrize('pose', [False, True])
def test_entry_point_writes_one_png_per_image(tmp_path, monkeypatch, pose):
    name = 'pose_run' if pose else 'parsing_run'
    cfg, names = _config(tmp_path, pose, name)
    monkeypatch.chdir(tmp_path)  # results/<name>/ is created under the working directory
    sample_from_parsing.run(pose=pose, argv=['-opt', cfg, '--batch-size', '2'])
    out = tmp_path / 'results' / name
    for n in names:
        img = np.array(Image.open(out / n)) if n.endswith('.png') else np.array(Image.open(out / n))
        assert img.shape == (512, 256, 3) and img.dtype == np.uint8 and img.std() > 1.0
    assert os.path.exists(out / f'test_{name}.log')
    with pytest.raises(FileExistsError):  # utils/util.py:22: an existing results dir is an error
        sample_from_parsing.run(pose=pose, argv=['-opt', cfg])
