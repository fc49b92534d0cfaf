"""Builds text2human_amd/libt2h_hip.so (gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the
gpurun snapshot (it is git-ignored, not gpurun-ignored)."""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libt2h_hip.so')
OBJ_DIR = os.path.join(HERE, 'csrc', 'build')
SOURCES = ['api.hip', 'gemm.hip', 'gemm_split.hip', 'conv_split.hip', 'conv_halo.hip', 'conv_small.hip', 'attention.hip', 'spatial_attn.hip', 'norm.hip', 'sampler.hip',
           'vq.hip', 'misc.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-comment']


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(path):
    h = hashlib.sha256()
    for f in [path, os.path.join(CSRC, 'common.h'),
              os.path.join(HERE, '..', 'include', 't2h_hip.h')]:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    hipcc = _hipcc()
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
    stamp = obj + '.sha'
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    # the per-kernel resource report (registers, spills, scratch, LDS) is kept next to the object:
    # tests/test_build_resources.py fails the build check if a hot kernel needs scratch memory
    r = subprocess.run([hipcc, *FLAGS, '-Rpass-analysis=kernel-resource-usage', '-c', path, '-o', obj],
                       stderr=subprocess.PIPE, text=True)
    with open(obj + '.resources.log', 'w') as f:
        f.write(r.stderr)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise subprocess.CalledProcessError(r.returncode, r.args)
    with open(stamp, 'w') as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(OUT):
        subprocess.run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC',
                        *objs, '-o', OUT], check=True)
        if verbose:
            print(f'built {OUT}', file=sys.stderr)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
