"""The sampler's random draws and unmasking schedule on the CPU (csrc/sampler.hip through tests/emu): the in-kernel
Philox4x32-10 against an independent numpy implementation that is itself pinned to Random123's published known-answer
vectors; the element <-> (subsequence, counter, word) mapping and the uniform transform against that reference; the
whole-run unmasking schedule (S-1, models/sample_model.py:279-292,301-306: which token is unmasked at which step, which
heads sample, where the generator stands) against a plain restatement of the reference's loop on the same draws.
(That the draws ARE torch's is checked on the GPU: tests/test_gpu_kernels.py compares full tensors with torch.rand /
exponential_.)"""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'emu'))
import build_emu  # noqa: E402

pytestmark = pytest.mark.skipif(not build_emu.available(), reason='no host clang++ for the emulation build')
c_vp, c_i32, c_i64, c_u32, c_u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64


@pytest.fixture(scope='module')
def lib():
    so = ctypes.CDLL(build_emu.build('sampler.hip'))
    for name, args in {'t2h_philox_uniform_f32': [c_u64, c_u64, c_u32, c_vp, c_i64, c_vp],
                       't2h_philox_exponential_f32': [c_u64, c_u64, c_u32, c_vp, c_i64, c_vp],
                       't2h_unmask_schedule': [c_u64, c_u64, c_u32, c_u32, c_u32, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]}.items():
        getattr(so, name).restype = ctypes.c_int
        getattr(so, name).argtypes = args
    so.emu_last_error.restype = ctypes.c_char_p
    return so


def philox4x32_10(counter, key):
    """Random123's philox4x32-10 (Salmon et al., SC'11): counter [..., 4] and key [..., 2] of uint32 -> [..., 4]"""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.asarray(key[..., 0], dtype=np.uint64), np.asarray(key[..., 1], dtype=np.uint64)
    M0, M1, W0, W1, MASK = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = c[0] * M0, c[2] * M1
        hi0, lo0, hi1, lo1 = p0 >> 32, p0 & MASK, p1 >> 32, p1 & MASK
        c = [(hi1 ^ c[1] ^ k0) & MASK, lo1, (hi0 ^ c[3] ^ k1) & MASK, lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def test_the_numpy_philox_matches_the_published_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want


def raw_words(seed, offset, grid_threads, numel):
    """word of element e as ATen's grid-stride kernel assigns it (sampler.hip's header comment)"""
    e = np.arange(numel, dtype=np.uint64)
    idx, m = e % np.uint64(grid_threads), e // np.uint64(grid_threads)
    ctr = np.uint64(offset // 4) + (m >> np.uint64(2))
    counter = np.stack([ctr & np.uint64(0xFFFFFFFF), ctr >> np.uint64(32), idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32)],
                       axis=-1).astype(np.uint32)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32), (numel, 2))
    words = philox4x32_10(counter, key)
    return words[np.arange(numel), (m & np.uint64(3)).astype(np.int64)]


def curand_uniform(words):
    """rocRAND / cuRAND's uniform of a 32-bit word: fma((float)v, 2^-32, 2^-32) -- the word is rounded to fp32 first,
    the fma rounds once more: (0, 1]"""
    f = words.astype(np.float32).astype(np.float64)
    return (f * 2.0**-32 + 2.0**-32).astype(np.float32)


def uniform_of(words):
    u = curand_uniform(words)
    return np.where(u == np.float32(1.0), np.float32(0.0), u)   # uniform_(0, 1) reverses the bounds


@pytest.mark.parametrize('seed,offset,grid', [(0, 0, 256), (2021, 4 * 77, 1024), (0x1234567890ABCDEF, 1 << 34, 256 * 8)])
def test_emulated_uniform_and_exponential_draws(lib, seed, offset, grid):
    numel = 5 * grid + 123   # every thread makes two calls, the second one partly used
    out = torch.full((numel,), float('nan'))
    assert lib.t2h_philox_uniform_f32(seed, offset, grid, out.data_ptr(), numel, None) == 0, lib.emu_last_error()
    words = raw_words(seed, offset, grid, numel)
    assert np.array_equal(out.numpy(), uniform_of(words))
    if seed == 0:  # elements 0, G, 2 G, 3 G are the four words of the first known-answer vector
        assert [int(w) for w in words[[0, grid, 2 * grid, 3 * grid]]] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    ex = torch.full((numel,), float('nan'))
    assert lib.t2h_philox_exponential_f32(seed, offset, grid, ex.data_ptr(), numel, None) == 0, lib.emu_last_error()
    u = curand_uniform(words).astype(np.float64)
    want = np.where(u >= 1.0 - 2.0**-24, 2.0**-24, -np.log(u))
    # (the kernel's log is v_log_f32 x ln 2 in two pieces, as ATen's build compiles it: the emulator's log2f is the
    # correctly rounded one, the hardware's an approximation -- compared to a few ulp here, bit for bit on the GPU)
    assert np.all(np.abs(ex.numpy().astype(np.float64) - want) <= 4e-7 * np.maximum(1.0, np.abs(want)))
    assert (ex.numpy() > 0).all()


def test_emulated_unmask_schedule_is_the_references_loop(lib):
    n, steps, n_heads = 2 * 512, 24, 18
    seed, offset, grid, rand_inc, expo_inc = 2021, 4 * 10, 1024, 4, 4 * 1024
    g = torch.Generator().manual_seed(4)
    tex = torch.randint(0, n_heads, (n,), generator=g)
    step_of_row = torch.full((n,), -1, dtype=torch.int32)
    head_mask = torch.zeros(steps + 1, dtype=torch.int32)
    rc = lib.t2h_unmask_schedule(seed, offset, grid, rand_inc, expo_inc, tex.data_ptr(), n, steps, n_heads,
                                 step_of_row.data_ptr(), head_mask.data_ptr(), None)
    assert rc == 0, lib.emu_last_error()
    # the reference: for t = steps .. 1: changes = rand(n) < 1 / t on the still-masked tokens; one exponential_ draw per
    # head that has a changed token, in between -- the generator moves by rand_inc + active heads * expo_inc per step
    want_step = np.zeros(n, dtype=np.int32)
    want_mask = np.zeros(steps + 1, dtype=np.uint32)
    off = offset
    for t in range(steps, 0, -1):
        r = uniform_of(raw_words(seed, off, grid, n))
        change = (want_step == 0) & (r < np.float32(1.0) / np.float32(t))
        want_step[change] = t
        heads = np.unique(tex.numpy()[change])
        want_mask[t] = sum(1 << int(h) for h in heads)
        off += rand_inc + len(heads) * expo_inc
    assert np.array_equal(step_of_row.numpy(), want_step) and (want_step > 0).all()
    assert np.array_equal(head_mask.numpy().view(np.uint32), want_mask)


def test_emulated_sampling_tail_is_the_exponential_race_of_the_reference():
    """S-3 (models/sample_model.py:300-306): ln_f, the row's own head, softmax(logits / temp), Categorical.sample() =
    argmax(p / Exp(1)).  t2h_sample_heads in its one-launch form and its two-launch form on explicit noise, and the
    two-launch form drawing the noise itself (element (row, class) of the tensor torch would have drawn at the head's
    generator offset) -- against fp64 on the same noise."""
    from text2human_amd._lib import SampleHeadsArgs
    so = ctypes.CDLL(build_emu.build('sampler.hip'))
    so.t2h_sample_heads.restype = ctypes.c_int
    so.t2h_sample_heads.argtypes = [ctypes.POINTER(SampleHeadsArgs), c_vp]
    so.emu_last_error.restype = ctypes.c_char_p
    n, C, n_class, n_heads, temp = 24, 512, 128, 4, 0.9
    g = torch.Generator().manual_seed(12)
    hidden = torch.randn(n, C, generator=g) * 1.5 + 0.2
    gamma, beta = torch.randn(C, generator=g) * 0.2 + 1.0, torch.randn(C, generator=g) * 0.1
    w = torch.randn(n_heads, n_class, C, generator=g) * 0.08
    tex = torch.randint(0, n_heads, (n,), generator=g)
    rows = torch.tensor([0, 3, 9, 10, 17, 23], dtype=torch.int32)
    seed, grid, inc = 2021, 256 * 12, 4   # (n * n_class = 3072 elements: 12 blocks of 256 threads, one call each)
    offs = [4 * 100 + h * inc for h in range(n_heads)]
    expo = [torch.from_numpy(-np.log(curand_uniform(raw_words(seed, offs[h], grid, n * n_class)).astype(np.float64))
                             .astype(np.float32)).view(n, n_class).contiguous() for h in range(n_heads)]

    def run(two_launch, in_kernel_noise):
        a = SampleHeadsArgs()
        x_t = torch.full((n,), -5, dtype=torch.int64)
        out = torch.full((n_heads, n), -1, dtype=torch.int64)
        ws = torch.zeros(len(rows), n_class)
        a.hidden, a.lnf_gamma, a.lnf_beta, a.w_heads = hidden.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr()
        a.rows, a.tex, a.x_t, a.out_idx = rows.data_ptr(), tex.data_ptr(), x_t.data_ptr(), out.data_ptr()
        a.temp, a.n_rows, a.n, a.C, a.n_class, a.n_heads = temp, len(rows), n, C, n_class, n_heads
        if two_launch:
            a.logits_ws = ws.data_ptr()
        if in_kernel_noise:
            a.philox_seed, a.philox_grid_threads = seed, grid
            for h in range(n_heads):
                a.philox_offset[h] = offs[h]
        else:
            for h in range(n_heads):
                a.expo[h] = expo[h].data_ptr()
        assert so.t2h_sample_heads(ctypes.byref(a), None) == 0, so.emu_last_error()
        return x_t, out

    y = torch.nn.functional.layer_norm(hidden.double(), (C,), gamma.double(), beta.double(), 1e-5)
    results = [run(False, False), run(True, False), run(True, True)]
    for x_t, out in results:
        untouched = torch.ones(n, dtype=torch.bool)
        untouched[rows.long()] = False
        assert (x_t[untouched] == -5).all() and (out[:, untouched] == -1).all()
    for r in rows.tolist():
        h = int(tex[r])
        logits = (w[h].double() @ y[r]) / temp
        score = torch.exp(logits - logits.max()) / expo[h][r].double()
        top2 = score.topk(2)
        clear = (top2.values[0] - top2.values[1]) > 1e-4 * top2.values[0]
        for k, (x_t, out) in enumerate(results):
            got = int(out[h, r])
            assert x_t[r] == got + n_class * h and (out[:, r] >= 0).sum() == 1
            if clear:   # (the third form's noise comes through the emulator's log2f: a few ulp from the table's)
                assert got == int(top2.indices[0]), (r, k, got, top2)
            else:
                assert got in top2.indices.tolist()
    assert torch.equal(results[0][1], results[1][1])   # one launch == two launches: the same fma chains
