"""A race detector for the kernels' LDS protocols: the kernel sources, compiled for the host workgroup emulator
(tests/emu), built with ThreadSanitizer and run through their product entry points by small C++ drivers
(tests/emu/tsan/*.cpp).  The threads of a workgroup are OS threads and `__syncthreads()` / the wave-collective
restatements are pthread barriers, so a shared-memory access that is not ordered by a barrier -- a fragment read of a
buffer another wave is already refilling, a staging tile read before its writer's barrier -- is a data race
ThreadSanitizer reports, with the offset inside the kernel's LDS array.

Expected: none, except where the halo convolution sends the pieces beyond the halo (its third piece exists for 272 of
512 threads) to overlapping scratch slots instead of predicating the stores off -- write / write on bytes nobody reads.
The negative control removes the per-tap barrier of that kernel from the emulated source and must be reported.

What this cannot see: the asynchronous landing of LDS-DMA and global loads (`s_waitcnt` counts): requests are
synchronous copies here."""
import hashlib
import os
import re
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(HERE, 'emu', 'tsan')


def _tsan_available():
    cc = build_emu._host_clang()
    if cc is None:
        return False
    os.makedirs(build_emu.OUT, exist_ok=True)
    src = os.path.join(build_emu.OUT, 'tsan_probe.cpp')
    open(src, 'w').write('int main() { return 0; }\n')
    r = subprocess.run([cc, '-fsanitize=thread', src, '-o', os.path.join(build_emu.OUT, 'tsan_probe')], capture_output=True)
    return r.returncode == 0


pytestmark = pytest.mark.skipif(not _tsan_available(), reason='no host clang++ with ThreadSanitizer')


def build_driver(kernel, transform=None, tag=''):
    """tests/emu/tsan/<kernel>.cpp + the emulation source of csrc/<kernel>.hip (optionally transformed) -> executable"""
    build_emu.build(f'{kernel}.hip')   # (re)generates tests/emu/_build/<kernel>_emu.cpp
    emu_cpp = os.path.join(build_emu.OUT, f'{kernel}_emu.cpp')
    text = open(emu_cpp).read()
    if transform is not None:
        text = transform(text)
        emu_cpp = os.path.join(build_emu.OUT, f'{kernel}_emu_{tag}.cpp')
        open(emu_cpp, 'w').write(text)
    drv = os.path.join(DRV, f'{kernel}.cpp')
    dig = hashlib.sha256((text + open(drv).read() + open(os.path.join(DRV, 'common.h')).read()).encode()).hexdigest()[:16]
    exe = os.path.join(build_emu.OUT, f'tsan_{kernel}{tag}_{dig}')
    if not os.path.exists(exe):
        subprocess.run([build_emu._host_clang(), '-O1', '-g', '-std=c++17', '-pthread', '-fsanitize=thread', '-Wno-everything',
                        f'-DEMU_SOURCE="{emu_cpp}"', f'-I{DRV}', drv, '-o', exe], check=True)
    return exe


def races(exe, *args):
    """-> list of (LDS offset or None, LDS array size or None, report) of one run"""
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS='history_size=4 exitcode=0'))
    assert 'rc 0' in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    out = []
    for rep in r.stderr.split('WARNING: ThreadSanitizer: ')[1:]:
        m = re.search(r' of size \d+ at (0x[0-9a-f]+) by thread', rep)
        g = re.search(r"Location is global '[^']*::(smem\w*)' of size (\d+) at (0x[0-9a-f]+)", rep)
        out.append((int(m.group(1), 16) - int(g.group(3), 16), int(g.group(2))) + (rep[:400],) if (m and g) else (None, None, rep[:400]))
    return out


@pytest.mark.parametrize('kernel,args', [('conv_split', (0,)), ('gemm_split', (8, 0)), ('gemm_split', (8, 1)),
                                         ('gemm_split', (6, 0)), ('gemm_split', (9, 0)), ('attention', (2,)),
                                         ('attention', (1,)), ('vq', ()), ('spatial_attn', ())])
def test_no_lds_race(kernel, args):
    """conv_split: 128-row tiles; gemm_split: the ping-pong LDS-DMA loop with fp16-plane and x8 operands, the in-block
    K split, the few-rows kernel; attention: key halves (polled LDS-counter barriers + merge) and all keys; vq: both
    codebook argmins; spatial_attn: the flash-style AttnBlock attention.  (tests/emu/tsan/gemm.cpp -- the exact-fp32
    GEMM / implicit convolution, also clean -- is kept out of the suite for its half-minute build.)"""
    found = races(build_driver(kernel), *args)
    assert not found, found[:3]


@pytest.mark.parametrize('variant', [1, 0])
def test_halo_convolution_races_only_on_its_scratch_slots(variant):
    found = races(build_driver('conv_halo'), variant)
    halo_b, weights_b = 18 * 2816, (3 * 128 * 128 if variant == 1 else 2 * 128 * 144)
    lo = 2 * halo_b + weights_b                    # CH_DMA_LOOP_B / CH_LOOP_B: where the scratch slots begin
    hi = lo + (3 * 512 - 324 * 4) * 16 + 64 + 16   # + CH_DUMMY_B
    # (usually five / two reports; none is fine too -- whether two unordered writes are caught depends on how far apart
    # they happen.  That the detector sees a real one is the negative control below.)
    for off, size, rep in found:
        assert off is not None and lo <= off < hi, (off, lo, hi, rep)


def test_a_removed_barrier_is_reported():
    """negative control: kernel 2 of conv_halo.hip without its per-tap barrier -- the fragment reads of a weight tile
    buffer / halo buffer are no longer ordered against the next requests and conversions"""
    def drop_tap_barrier(text):
        a = text.index('void conv_halo_dma_kernel(')
        marker = 'HALO_STAMP(st[4]);\n      __syncthreads();'
        b = text.index(marker, a)
        return text[:b] + 'HALO_STAMP(st[4]);' + text[b + len(marker):]
    found = races(build_driver('conv_halo', transform=drop_tap_barrier, tag='_no_tap_barrier'), 1)
    lo = 2 * 18 * 2816 + 3 * 128 * 128
    assert any(off is not None and off < lo for off, _, _ in found), 'a missing barrier went unnoticed'
