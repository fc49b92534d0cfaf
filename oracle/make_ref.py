"""TEST INFRASTRUCTURE (build container; needs /root/reference): compiles the reference's own modules of the
sampling path -- from the sources WHERE THEY LIE under /root/reference, unmodified -- into CPython byte code under
the git-ignored `oracle/_ref/`, so that the reference itself (not the port) can be executed on the GPU box, where
`/root/reference` does not exist: as the pin of the oracle there and as `bench.py`'s `cpu_baseline` with
`kind: "reference"`.

    python oracle/make_ref.py            # also run by __graft_entry__.build() when /root/reference is present

Only compiler OUTPUT goes into `oracle/_ref/` (`<module>.pyc` where `<module>.py` would sit: CPython imports
source-less byte code from there); no reference source is copied anywhere.  The build container and the GPU box
run the same image, hence the same interpreter; `oracle/ref_shim.py` checks the byte code's magic number before use
and reports the reference as unavailable otherwise.  `oracle/_ref/` is listed in `.gitignore` (it stays out of the
history) and NOT in `.gpurunignore` (it travels with the snapshot like the built `.so`).

Nothing under `text2human_amd/` imports, links or executes anything from `oracle/` (tests/test_cabi.py)."""
import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
SRC = os.environ.get('T2H_REFERENCE_SRC', '/root/reference')

# the modules oracle/ref_shim.load_reference() imports, plus the encode-side / training-time model
# files the oracle-vs-reference tests load by name
MODULES = [
    'models/sample_model.py',
    'models/hierarchy_inference_model.py',
    'models/transformer_model.py',
    'models/archs/__init__.py',
    'models/archs/vqgan_arch.py',
    'models/archs/transformer_arch.py',
    'models/archs/unet_arch.py',
    'models/archs/fcn_arch.py',
    'models/archs/shape_attr_embedding_arch.py',
    'utils/__init__.py',
    'utils/options.py',
    'utils/util.py',
]


def build(verbose=True):
    """-> manifest dict, or None when the reference sources are not here (the GPU box: uses the prebuilt files)."""
    if not os.path.isdir(os.path.join(SRC, 'models', 'archs')):
        return None
    manifest = {'source_root': SRC, 'magic': importlib.util.MAGIC_NUMBER.hex(), 'python': sys.version.split()[0],
                'modules': {}}
    for rel in MODULES:
        src = os.path.join(SRC, rel)
        if not os.path.exists(src):
            continue
        dst = os.path.join(OUT, rel[:-3] + '.pyc')
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: tracebacks name the reference file the byte code came from
        py_compile.compile(src, cfile=dst, dfile=src, doraise=True,
                           invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        manifest['modules'][rel] = hashlib.sha256(open(src, 'rb').read()).hexdigest()[:16]
    with open(os.path.join(OUT, 'MANIFEST.json'), 'w') as f:
        json.dump(manifest, f, indent=1)
    if verbose:
        print(f'oracle/_ref: byte code of {len(manifest["modules"])} reference modules '
              f'(python {manifest["python"]}, magic {manifest["magic"]})')
    return manifest


if __name__ == '__main__':
    if build() is None:
        sys.exit(f'reference sources not found under {SRC}')
