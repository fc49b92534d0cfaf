"""CPU-side checks of the drop-in boundary: libt2h_hip.so builds/loads and
exports every symbol include/t2h_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

from text2human_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, 'include', 't2h_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(t2h_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from text2human_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert decl, 'no symbols parsed from the header'
    assert sorted(_lib.SIGNATURES) == decl


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.t2h_version() >= 100


def test_argument_validation_without_gpu(lib):
    # NULL args are rejected before any HIP call -> safe on a GPU-less box
    assert lib.t2h_gemm_f32(None, None) == -1
    assert b'NULL' in lib.t2h_last_error()
    assert lib.t2h_groupnorm_workspace_bytes(2, 4096, 128) == 2 * 4 * 2 * 128 * 8


def test_gemm_args_layout_matches_header():
    # field order of the ctypes mirror == field order of struct t2h_gemm_args
    src = open(os.path.join(ROOT, 'include', 't2h_hip.h')).read()
    body = src[src.index('typedef struct t2h_gemm_args {'):src.index('} t2h_gemm_args;')]
    body = re.sub(r'/\*.*?\*/', '', body[body.index('{') + 1:], flags=re.S)
    fields = []
    for m in re.finditer(r'(?:const\s+float\*|float\*|double\*|int32_t|int64_t|float)\s+([^;]+);', body):
        fields += [n.strip() for n in m.group(1).split(',')]
    assert fields == [f[0] for f in _lib.GemmArgs._fields_]


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'text2human_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt, f
