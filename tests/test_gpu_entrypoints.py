"""Entry points on a real dataset tree (-m gpu): YAML + .pth files +
DeepFashion-style directories -> `python -m text2human_amd.sample_from_parsing / _pose`
write one PNG per listed image, like the reference scripts (sample_from_parsing.py:38-49,
sample_from_pose.py:32-48) -- and every PNG is the ORACLE's image for that name: the oracle
replays the same dataset tree, loader batches, seed and generator."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import torch_ref as R
from text2human_amd import data, defaults, options, sample_from_parsing, synthetic
from text2human_amd.models import sample_model

from parity_util import odev, osds

pytestmark = pytest.mark.gpu
DEV = 'cuda'
STEPS, BATCH, SEED = 6, 2, 2021


def _config(tmp_path, pose, name):
    opt = defaults.sample_from_pose() if pose else defaults.sample_from_parsing()
    sds = synthetic.make_state_dicts(options.dict_to_nonedict(opt), seed=1234)
    if pose:  # a parsing generator whose maps are not one constant class (parity_util.balanced_pose_state_dicts)
        from parity_util import balanced_pose_state_dicts
        sds = balanced_pose_state_dicts(sds, opt)
    opt = synthetic.write_checkpoints(opt, str(tmp_path / 'ckpt'), state_dicts=sds)
    tree = synthetic.write_dataset_tree(str(tmp_path / 'data'), n=3, seed=11)
    names = tree.pop('names')
    opt.update(tree)
    opt.update(name=name, sample_steps=STEPS, manual_seed=SEED)
    return defaults.write_yaml(opt, str(tmp_path / f'{name}.yml')), names, opt, sds


def _oracle_images(opt, sds, pose, recorded):
    """name -> uint8 HWC image of the oracle on the same loader batches / seed / device generator.  For the pose
    entry the oracle continues from the RECORDED parsing map and tokens of each batch (checked against the
    oracle's own separately below): an argmax near-tie of the parsing generator cannot cascade into the comparison."""
    if pose:
        ds = data.DeepFashionAttrPoseDataset(pose_dir=opt['pose_dir'], texture_ann_dir=opt['texture_ann_file'],
                                             shape_ann_path=opt['shape_ann_path'])
    else:
        ds = data.DeepFashionAttrSegmDataset(img_dir=opt['test_img_dir'], segm_dir=opt['segm_dir'],
                                             pose_dir=opt['pose_dir'], ann_dir=opt['test_ann_file'])
    loader = torch.utils.data.DataLoader(dataset=ds, batch_size=BATCH, shuffle=False)
    sd_dev = {k: v.to(DEV) for k, v in sds['sampler'].items()}
    od = osds(sds)   # (the oracle's convolutional stages: parity_util.ORACLE_DEV)
    out, checks = {}, []
    options.set_random_seed(SEED)
    with torch.no_grad():
        for i, batch in enumerate(loader):
            rec = recorded[i]
            if pose:
                segm_ref, logits = R.parsing_from_pose(odev(batch['densepose']), odev(batch['shape_attr']), od['shape_embedder'],
                                                       od['shape_encoder'], od['shape_decoder'],
                                                       opt['shape_attr_class_num'])
                segm_ref, logits = segm_ref.cpu(), logits.cpu()
                bad = rec['segm'].cpu() != segm_ref
                t2 = logits.topk(2, dim=1).values
                assert ((t2[:, 0] - t2[:, 1]).unsqueeze(1)[bad] < 1e-4).all() and bad.float().mean() < 1e-3
                segm = rec['segm'].cpu()
                mask = R.texture_map(segm, batch['upper_fused_attr'], batch['lower_fused_attr'], batch['outer_fused_attr'])
                assert torch.equal(rec['texture_mask'].cpu(), mask)
                tok_ref = R.segm_tokens(odev(segm), od['segm_encoder'], od['segm_quant_conv'],
                                        od['segm_quantizer']['embedding.weight']).view(segm.shape[0], -1).cpu()
                checks.append(float((rec['segm_tokens'].cpu() != tok_ref).float().mean()))
                tok = rec['segm_tokens']
            else:
                mask = batch['texture_mask']
                assert torch.equal(rec['texture_mask'].cpu(), mask)
                tok_ref = R.segm_tokens(odev(batch['segm']), od['segm_encoder'], od['segm_quant_conv'],
                                        od['segm_quantizer']['embedding.weight']).view(mask.shape[0], -1).cpu()
                assert torch.equal(rec['segm_tokens'].cpu(), tok_ref)      # piece-wise constant maps: exact
                tok = tok_ref.to(DEV)
            top = R.sample_fn(tok, mask.to(DEV), sd_dev, sample_steps=STEPS, noise=R.TorchNoise(DEV))
            assert torch.equal(torch.stack(top), torch.stack(rec['top'])), (i, 'sampled tokens differ from the oracle')
            img, _ = R.refine_and_decode(odev(top), odev(mask), od)
            u8 = R.to_uint8(img).cpu().numpy()
            for j, n in enumerate(batch['img_name']):
                assert n not in out
                out[n] = u8[j]
    assert all(c < 5e-3 for c in checks), checks  # pose tokenizer on pixel-noisy maps: near-ties only (test_gpu_configs.py)
    return out


@pytest.mark.parametrize('pose', [False, True])
def test_entry_point_writes_the_oracles_image_per_name(tmp_path, monkeypatch, pose):
    name = 'pose_run' if pose else 'parsing_run'
    cfg, names, opt, sds = _config(tmp_path, pose, name)
    monkeypatch.chdir(tmp_path)  # results/<name>/ is created under the working directory
    recorded = []
    real = sample_model.BaseSampleModel.sample_fn

    def spy(self, *a, **k):  # the state each loader batch is sampled from, and its tokens
        top = real(self, *a, **k)
        recorded.append(dict(segm=self.segm.clone(), segm_tokens=self.segm_tokens.clone(),
                             texture_mask=self.texture_mask.clone(), top=[t.clone() for t in top]))
        return top

    monkeypatch.setattr(sample_model.BaseSampleModel, 'sample_fn', spy)
    written = {}   # name -> the uint8 image handed to the file writer (the dataset's names end in .jpg: the file
    real_save = sample_model.save_u8_images   # itself is JPEG-compressed, as the reference's save_image would)

    def save_spy(u8, save_dir, img_name):
        for i, n in enumerate(img_name):
            assert n not in written
            written[n] = u8[i].cpu().numpy().copy()
        return real_save(u8, save_dir, img_name)

    monkeypatch.setattr(sample_model, 'save_u8_images', save_spy)
    sample_from_parsing.run(pose=pose, argv=['-opt', cfg, '--batch-size', str(BATCH)])
    monkeypatch.setattr(sample_model.BaseSampleModel, 'sample_fn', real)
    monkeypatch.setattr(sample_model, 'save_u8_images', real_save)
    out = tmp_path / 'results' / name
    assert sorted(f for f in os.listdir(out) if not f.endswith('.log')) == sorted(names)  # one file per listed image
    assert len(recorded) == 2                             # 3 images in loader batches of 2
    want = _oracle_images(options.dict_to_nonedict(opt), sds, pose, recorded)
    assert sorted(want) == sorted(names)
    import io
    for n in names:
        # what was handed to the writer under this name = the oracle's image for this name (uint8, +-1 on < 0.5 %)
        d = np.abs(written[n].astype(np.int16) - want[n].astype(np.int16))
        assert d.max() <= 1 and (d != 0).mean() < 5e-3, (n, int(d.max()), float((d != 0).mean()))
        # and the file holds it, through the encoder its extension selects (PIL, like torchvision's save_image)
        img = np.array(Image.open(out / n))
        assert img.shape == (512, 256, 3) and img.dtype == np.uint8
        buf = io.BytesIO()
        Image.fromarray(written[n]).save(buf, format=Image.registered_extensions()[os.path.splitext(n)[1].lower()])
        assert np.array_equal(img, np.array(Image.open(io.BytesIO(buf.getvalue())))), n
    # the images of different names differ (a per-name mix-up cannot pass)
    assert not np.array_equal(want[names[0]], want[names[1]])
    assert os.path.exists(out / f'test_{name}.log')
    with pytest.raises(FileExistsError):  # utils/util.py:22: an existing results dir is an error
        sample_from_parsing.run(pose=pose, argv=['-opt', cfg])


def test_entry_point_under_torchrun_with_a_one_rank_rccl_group(tmp_path):
    """The multi-GPU form of the entry point on the hardware there is: `python -m torch.distributed.run
    --nproc-per-node 1 -m text2human_amd.sample_from_parsing` with T2H_FORCE_DIST=1 -- RCCL initialised, the checkpoints
    broadcast through it, the dataset "sharded" over one rank, graphs replayed beside the live communicator -- writes
    the same PNGs as the plain single-process run (same slice, same seed)."""
    import socket
    import subprocess
    import sys
    cfg, names, opt, sds = _config(tmp_path, False, 'rccl_run')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    outs = {}
    for tag, cmd, extra in (
            ('plain', [sys.executable, '-m', 'text2human_amd.sample_from_parsing'], {}),
            ('rccl', [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr',
                      '127.0.0.1', '--master-port', str(port), '-m', 'text2human_amd.sample_from_parsing'],
             {'T2H_FORCE_DIST': '1'})):
        cwd = tmp_path / tag
        cwd.mkdir()
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), **extra)
        r = subprocess.run(cmd + ['-opt', cfg, '--batch-size', '2'], capture_output=True, text=True, timeout=600,
                           env=env, cwd=str(cwd))
        assert r.returncode == 0, r.stderr[-3000:]
        out = cwd / 'results' / 'rccl_run'
        outs[tag] = {n: np.array(Image.open(out / n)) for n in names}
        assert os.path.exists(out / 'test_rccl_run.log')
        if tag == 'rccl':
            assert 'this rank: items [0, 3)' in r.stderr
    for n in names:
        assert np.array_equal(outs['plain'][n], outs['rccl'][n]), n
