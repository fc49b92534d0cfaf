"""TEST ONLY (tests/test_sharded_entry.py): runs the product entry point `text2human_amd.sample_from_parsing.run`
under torch.distributed.run on CPU / gloo with a stub in place of the HIP model, to exercise its multi-rank host
logic -- dataset shard, per-rank seed, rank-0 results directory and log, one read of the checkpoints + broadcast,
per-rank files -- without a GPU.  The stub "image" of a name is a digest of the item's data, the broadcast weights
and the rank's seeded random stream, so the test can recompute it for any slice."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from text2human_amd import sample_from_parsing as entry  # noqa: E402

READS = os.environ.get('T2H_STUB_READS_FILE')


class StubModel:

    def __init__(self, opt, state_dicts=None):
        assert state_dicts is not None or int(os.environ.get('WORLD_SIZE', 1)) == 1
        sds = state_dicts if state_dicts is not None else stub_state_dicts(opt)
        self.w = float(sum(v.double().sum() for sd in sds.values() for v in sd.values()))
        self.n_int = int(sds['sampler']['steps'].item())

    def inference(self, loader, save_dir):
        for data in loader:
            draw = torch.rand(len(data['img_name']), 4)  # the global generator, like the reference's sampler
            for i, name in enumerate(data['img_name']):
                h = hashlib.sha256()
                h.update(data['segm'][i].numpy().tobytes())
                h.update(data['texture_mask'][i].numpy().tobytes())
                h.update(draw[i].numpy().tobytes())
                h.update(repr((self.w, self.n_int)).encode())
                with open(os.path.join(save_dir, name), 'x') as f:  # 'x': a name written twice is an error
                    f.write(h.hexdigest())


def stub_state_dicts(opt):
    if os.environ.get('T2H_STUB_LOAD_FAILS') == '1':   # a missing / corrupt .pth on rank 0
        raise FileNotFoundError('stub: pretrained_sampler.pth is not there')
    if READS:  # how many processes read the "checkpoints"
        with open(READS, 'a') as f:
            f.write(f"{os.environ.get('RANK', 0)}\n")
    g = torch.Generator().manual_seed(5)
    return {'sampler': {'w': torch.randn(7, 3, generator=g), 'steps': torch.tensor(256)},
            'decoder': {'b': torch.randn(4, generator=g).half(), 'flag': torch.tensor([True, False])}}


if __name__ == '__main__':
    entry.create_model = lambda opt, state_dicts=None: StubModel(opt, state_dicts)
    entry.load_state_dicts = stub_state_dicts
    entry.run(pose=False, argv=sys.argv[1:])
