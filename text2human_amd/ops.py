"""Thin tensor -> pointer wrappers over the C ABI (include/t2h_hip.h).

PyTorch-ROCm is used for device memory, streams and nothing else: every
function here checks dtype / device / contiguity, takes raw device pointers and
the current HIP stream, and calls into libt2h_hip.so.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import GemmArgs, check

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
PRO_NONE, PRO_SWISH = 0, 1


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- the sticky overflow word of the split-row producers (include/t2h_hip.h): owned HERE, one int32 in
# device memory per (device, stream) plus a pinned host word the asynchronous read-back lands in.  Models
# driven on different streams (or threads, each on its own stream) neither see nor clear each other's flag.
_ovf_slots = {}


def _ovf_slot():
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    slot = _ovf_slots.get(key)
    if slot is None:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.T2HError('the overflow flag of this stream must exist before a capture begins: call '
                                'ops.split_overflow(reset=True) on the stream first')
        slot = _ovf_slots[key] = (torch.zeros(1, dtype=torch.int32, device=f'cuda:{dev}'),
                                  torch.zeros(1, dtype=torch.int32).pin_memory())
    return slot


def overflow_flag():
    """Device pointer of the current stream's overflow word (allocated, zeroed, at first use)."""
    return ctypes.c_void_p(_ovf_slot()[0].data_ptr())


def release_overflow_flag(stream=None):
    """Forgets the overflow word of `stream` (default: the current one), e.g. before the stream is destroyed:
    a later stream that reuses the handle then starts from a fresh, zeroed word."""
    st = stream if stream is not None else torch.cuda.current_stream()
    _ovf_slots.pop((torch.cuda.current_device(), st.cuda_stream), None)


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_cuda:
            raise TypeError(f'expected CUDA float32 tensor, got {t.dtype} on {t.device}')


def _chk_i64(*ts):
    for t in ts:
        if t.dtype != torch.int64 or not t.is_cuda or not t.is_contiguous():
            raise TypeError('expected contiguous CUDA int64 tensor')


def _rows(t):
    """(ptr tensor, leading dim) of a 2-D row-major view (stride(1) == 1)."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f'need a 2-D tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}')
    return t.stride(0)


GEMM_CFG_NAMES = ['64x64x32', '128x32x32', '128x64x32', '128x128x32', '128x64x64', '128x128x64',
                  '128x64x32w8', '128x64x64w8', '128x128x64w8']


def gemm_force_config(cfg):
    return _lib.load().t2h_gemm_force_config(int(cfg))


# ---- optional HIP-event profiling of GEMM launches (bench.py roofline leg) ----
_prof = None


def gemm_profile_start(every=1):
    """Bracket every `every`-th t2h_gemm_f32 launch with HIP events on the
    launch stream (torch's current stream)."""
    global _prof
    _prof = dict(every=max(1, int(every)), count=0, recs=[])


def gemm_profile_active():
    return _prof is not None


SPLIT_CFG_NAMES = {0: '128x64', 1: '128x128', 2: '64x64', 3: '128x64, 8 waves', 5: '128x256', 6: '128x64, 2 K groups',
                   8: '256x128, ping-pong LDS-DMA', 9: 'few rows (16x16 per workgroup, K over 8 waves)',
                   10: '128x192, ping-pong LDS-DMA', 11: '128x128, ping-pong LDS-DMA'}


def _probe_clock(buf):
    """(median main-loop time in us, median shader clock in GHz over the main loop) from a phase-stamp buffer of
    t2h_gemm_split_probe_next_launch; None if the launch left no stamps (few-rows kernel)."""
    t = buf.view(-1, 16).cpu()
    t = t[(t[:, 1] != 0) & (t[:, 2] != 0)].double()
    if t.shape[0] == 0:
        return None
    us = (t[:, 2] - t[:, 1]) * 0.01
    ghz = (t[:, 10] - t[:, 9]) / (us * 1e3)
    return float(us.median()), float(ghz.median())


def gemm_profile_stop():
    """-> {kernel label: dict(kernel, n, ms, flops, ...)} (synchronises).  The split GEMM appears once as
    'gemm_split_kernel<2xfp16>' (all instantiations, launch-weighted) and once per tile configuration of the
    dispatcher under that record's 'by_cfg'; records that carried a phase-stamp buffer add the main loop's
    duration and the shader clock it ran at."""
    global _prof
    p, _prof = _prof, None
    if not p:
        return {}
    torch.cuda.synchronize()
    out = {}

    def add(r, flops, e0, e1, kind, probe):
        if kind == 'stream':  # interval on the stream: kernel + launch boundary
            r['n_stream'] += 1
            r['ms_stream'] += e0.elapsed_time(e1)
            r['flops_stream'] += flops
            return
        r['kernel_timed'] = r['kernel_timed'] or kind == 'kernel'
        r['n'] += 1
        r['ms'] += e0.elapsed_time(e1)
        r['flops'] += flops
        if probe is not None:
            r['n_probe'] += 1
            r['loop_us'] += probe[0]
            r['loop_ghz'] += probe[1]

    blank = lambda label: dict(kernel=label, n=0, ms=0.0, flops=0.0, n_stream=0, ms_stream=0.0, flops_stream=0.0,
                               kernel_timed=False, n_probe=0, loop_us=0.0, loop_ghz=0.0)
    for rec in p['recs']:
        label, flops, e0, e1 = rec[:4]
        kind = rec[4] if len(rec) > 4 else None
        cfg = rec[5] if len(rec) > 5 else None
        probe = _probe_clock(rec[6]) if len(rec) > 6 and rec[6] is not None else None
        r = out.setdefault(label, blank(label))
        add(r, flops, e0, e1, kind, probe)
        if cfg is not None:
            name = f'gemm_split_kernel<{SPLIT_CFG_NAMES.get(cfg, cfg)}>'
            add(r.setdefault('by_cfg', {}).setdefault(name, blank(name)), flops, e0, e1, kind, probe)
    return out


def _launch_gemm(g, what):
    lib = _lib.load()
    if _prof is not None:
        _prof['count'] += 1
        if _prof['count'] % _prof['every'] == 0:
            cfg = lib.t2h_gemm_tile_config(ctypes.byref(g))
            label = (f"gemm_kernel<{GEMM_CFG_NAMES[cfg]},amode={g.a_mode},"
                     f"pro={int(bool(g.pro_scale))},btrans={g.b_trans}>")
            flops = 2.0 * g.M * g.N * g.K * max(1, g.batch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.t2h_gemm_f32(ctypes.byref(g), _stream()), what)
            e1.record()
            _prof['recs'].append((label, flops, e0, e1))
            return
    check(lib.t2h_gemm_f32(ctypes.byref(g), _stream()), what)


def gemm(a, w, out=None, bias=None, residual=None, act=ACT_NONE, alpha=1.0, pro=None,
         b_trans=False):
    """out[M,N] = act(alpha * pro(a)[M,K] @ w[N,K]^T + bias) + residual.

    a, w, out, residual: 2-D views with unit inner stride (row stride free).
    pro = (scale[n_img,K], shift[n_img,K], rows_per_img, pro_act).
    b_trans: w is given as [K,N]."""
    _chk_f32(a, w, out, bias, residual)
    M, K = a.shape
    N = w.shape[1] if b_trans else w.shape[0]
    assert (w.shape[0] if b_trans else w.shape[1]) == K, (a.shape, w.shape)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    g = GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.residual = residual.data_ptr() if residual is not None else None
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = _rows(a), _rows(w), _rows(out)
    g.ldr = _rows(residual) if residual is not None else 0
    g.a_mode, g.b_trans, g.epi_act, g.alpha = 0, int(b_trans), act, alpha
    if pro is not None:
        sc, sh, rows, pact = pro
        _chk_f32(sc, sh)
        g.pro_scale, g.pro_shift = sc.data_ptr(), sh.data_ptr()
        g.pro_rows, g.pro_ld, g.pro_act = rows, sc.shape[1], pact
    g.batch = 1
    _launch_gemm(g, 't2h_gemm_f32')
    return out


def _launch_conv_split(g, flops):
    lib = _lib.load()
    if _prof is not None:
        _prof['count'] += 1
        if _prof['count'] % _prof['every'] == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.t2h_conv_split_f32(ctypes.byref(g), _stream()), 't2h_conv_split_f32')
            e1.record()
            _prof['recs'].append(('conv_split_kernel<2xfp16>', flops, e0, e1))
            return
    check(lib.t2h_conv_split_f32(ctypes.byref(g), _stream()), 't2h_conv_split_f32')


def conv_split_ok(n_pix_per_img, mode='same', act=ACT_NONE):
    """Shapes t2h_conv_split_f32 serves: stride-1 convolutions whose images hold a multiple of
    128 pixels (a workgroup never straddles two images)."""
    return mode in ('same', 'up') and n_pix_per_img % 128 == 0 and act in (ACT_NONE, ACT_RELU)


def gn_apply_split(x, scale=None, shift=None, rows_per_img=0, act=PRO_NONE, out=None):
    """fp32 pixel rows x [rows, C] -> split rows of act(x * scale[img] + shift[img]) (GroupNorm apply
    + swish in one pass; scale None = plain split): the activation operand of conv_split."""
    _chk_f32(x, scale, shift)
    rows, C = x.shape
    if out is None:
        out = split_rows_empty(rows, C, x.device)
    check(_lib.load().t2h_gn_apply_split_f32(_p(x), _rows(x), _p(scale), _p(shift),
                                             scale.shape[1] if scale is not None else 0, _p(out), rows,
                                             rows_per_img, C, act, overflow_flag(), _stream()), 't2h_gn_apply_split_f32')
    return out


def conv_split(xs, w_split, n_img, hin, win, cin, cout, taps=9, out=None, bias=None, residual=None,
               act=ACT_NONE, mode='same', res_pre=False, gn_stats=False):
    """Stride-1 convolution (taps 9: 3x3 'same' or, mode 'up', 3x3 after nearest x2; taps 1: 1x1) of
    split-row activations xs [n_img*hin*win, cin/32, 2, 32] with split-row weights
    w_split [cout, taps*cin/32, 2, 32] on the fp16 matrix cores; fp32 rows out [M, cout]."""
    _chk_f32(out, bias, residual)
    assert mode in ('same', 'up') and taps in (1, 9)
    ups = 1 if mode == 'up' else 0
    hout, wout = hin << ups, win << ups
    M = n_img * hout * wout
    assert xs.numel() == n_img * hin * win * cin * 2 and w_split.numel() == cout * taps * cin * 2, \
        (tuple(xs.shape), tuple(w_split.shape), n_img, hin, win, cin, cout, taps)
    if out is None:
        out = torch.empty((M, cout), device=xs.device, dtype=torch.float32)
    g = GemmArgs()
    g.A, g.B, g.C = xs.data_ptr(), w_split.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.residual = residual.data_ptr() if residual is not None else None
    g.M, g.N, g.K = M, cout, taps * cin
    g.lda, g.ldb, g.ldc = 0, 0, _rows(out)
    g.ldr = _rows(residual) if residual is not None else 0
    g.a_mode, g.epi_act, g.alpha, g.res_pre = 1, act, 1.0, int(res_pre)
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout = hin, win, cin, hout, wout
    g.stride, g.pad, g.ups, g.batch = 1, (1 if taps == 9 else 0), ups, 1
    part = None
    if gn_stats:  # per-(image, 128-row tile, channel) fp64 (sum, sum of squares) of the output, from the epilogue
        part = torch.empty((n_img, hout * wout // 128, 2, cout), device=xs.device, dtype=torch.float64)
        g.gn_part_out = part.data_ptr()
    _launch_conv_split(g, 2.0 * M * cout * taps * cin)
    if part is not None:
        out._t2h_gn_part = part  # groupnorm_tables(out, ...) then only reduces these partials
    return out


def conv_halo_ok(n_img, hout, wout, cin, cout, mode='same'):
    """Shapes t2h_conv_halo_f32 serves AND is meant for: 3x3 'same' / nearest-x2 convolutions whose 16 x 16-pixel
    tiles x 128-channel column tiles give at least 32 workgroups PER IMAGE (a chunk of 8 images then gives every CU
    one: the decoders' levels from 64 x 32 / 512 channels up); the small levels keep the 128-pixel tiles of
    t2h_conv_split_f32.  Deliberately NOT a function of n_img: the two kernels sum K in different orders, and an
    image must not depend on how many neighbours it is decoded with (tests/test_gpu_edge_cases.py:
    test_decode_chunk_boundary_is_invisible compares bit for bit).  T2H_HALO_CONV=0 switches the kernel off, =2 uses
    it wherever it serves the shape (tests)."""
    knob = os.environ.get('T2H_HALO_CONV', '1')
    if knob == '0':
        return False
    served = mode in ('same', 'up') and hout % 16 == 0 and wout % 16 == 0 and cin % 32 == 0 and cout % 8 == 0 and cin <= 512
    # (the kernel addresses an image and the weights with 32-bit byte offsets: beyond 2 GiB per image the two-kernel
    # path serves the layer -- 4096 x 2048 pixels of 128 channels; the decoders' largest is 1024 x 512)
    served = served and hout * wout * cin * 4 < 2**31 and cout * 9 * cin * 4 < 2**31
    return served and (knob == '2' or (hout // 16) * (wout // 16) * ((cout + 127) // 128) >= 32)


def conv_halo(x, w_split, n_img, hin, win, cin, cout, out=None, bias=None, residual=None, mode='same', pro=None,
              gn_stats=False):
    """3x3 convolution (mode 'same', or 'up': after nearest x2) of fp32 NHWC rows x [n_img*hin*win, >= cin] with
    split-row weights w_split [cout, 9*cin/32, 2, 32]; pro = (scale, shift) [n_img, C] GroupNorm tables applied with
    swish while the operand is staged (None: plain convolution).  fp32 rows out [M, cout]; gn_stats as conv_split."""
    _chk_f32(x, out, bias, residual)
    assert mode in ('same', 'up')
    ups = 1 if mode == 'up' else 0
    hout, wout = hin << ups, win << ups
    M = n_img * hout * wout
    assert x.shape[0] == n_img * hin * win and x.shape[1] >= cin and w_split.numel() == cout * 9 * cin * 2, \
        (tuple(x.shape), tuple(w_split.shape), n_img, hin, win, cin, cout)
    if out is None:
        out = torch.empty((M, cout), device=x.device, dtype=torch.float32)
    g = GemmArgs()
    g.A, g.B, g.C = x.data_ptr(), w_split.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.residual = residual.data_ptr() if residual is not None else None
    g.M, g.N, g.K = M, cout, 9 * cin
    g.lda, g.ldb, g.ldc = _rows(x), 0, _rows(out)
    g.ldr = _rows(residual) if residual is not None else 0
    g.a_mode, g.epi_act, g.alpha, g.res_pre = 1, ACT_NONE, 1.0, 0
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout = hin, win, cin, hout, wout
    g.stride, g.pad, g.ups, g.batch = 1, 1, ups, 1
    if pro is not None:
        sc, sh = pro
        _chk_f32(sc, sh)
        g.pro_scale, g.pro_shift, g.pro_ld, g.pro_act = sc.data_ptr(), sh.data_ptr(), sc.shape[1], PRO_SWISH
    part = None
    if gn_stats:
        part = torch.empty((n_img, hout * wout // 128, 2, cout), device=x.device, dtype=torch.float64)
        g.gn_part_out = part.data_ptr()
    lib = _lib.load()
    flops = 2.0 * M * cout * 9 * cin
    if _prof is not None:
        _prof['count'] += 1
        if _prof['count'] % _prof['every'] == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.t2h_conv_halo_f32(ctypes.byref(g), overflow_flag(), _stream()), 't2h_conv_halo_f32')
            e1.record()
            _prof['recs'].append(('conv_halo_kernel<2xfp16>', flops, e0, e1))
            if part is not None:
                out._t2h_gn_part = part
            return out
    check(lib.t2h_conv_halo_f32(ctypes.byref(g), overflow_flag(), _stream()), 't2h_conv_halo_f32')
    if part is not None:
        out._t2h_gn_part = part
    return out


def bgemm(a, w, out, alpha=1.0, b_trans=False):
    """Batched: a [b,M,K], w [b,N,K] (or [b,K,N] if b_trans), out [b,M,N]; 3-D
    views with unit inner stride (batch / row strides free)."""
    _chk_f32(a, w, out)
    nb, M, K = a.shape
    N = w.shape[2] if b_trans else w.shape[1]
    g = GemmArgs()
    g.A, g.B, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    g.M, g.N, g.K = M, N, K
    assert a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1
    g.lda, g.ldb, g.ldc = a.stride(1), w.stride(1), out.stride(1)
    g.strideA, g.strideB, g.strideC = a.stride(0), w.stride(0), out.stride(0)
    g.batch, g.alpha, g.b_trans = nb, alpha, int(b_trans)
    _launch_gemm(g, 't2h_gemm_f32(batched)')
    return out


def conv3x3(x, w, n_img, hin, win, cin, out=None, bias=None, residual=None, act=ACT_NONE,
            pro=None, mode='same', res_pre=False, ksplit=None):
    """3x3 convolution of an NHWC image held as pixel rows x [n_img*hin*win, >=cin]
    with packed weights w [Cout, 9*cin] ([tap][cin] order).

    mode: 'same' (stride 1 pad 1), 'up' (nearest x2 then same conv),
          'down' (zero-pad right/bottom by 1, stride 2).
    ksplit: K slices on their own workgroups (t2h_gemm_args.ksplit): None = the library's choice from the layer's
          geometry (few pixels per image: the deep UNet levels), 1 = one pass over K, n > 1 = n slices."""
    _chk_f32(x, w, out, bias, residual)
    if mode == 'same':
        hout, wout, stride, pad, ups = hin, win, 1, 1, 0
    elif mode == 'up':
        hout, wout, stride, pad, ups = 2 * hin, 2 * win, 1, 1, 1
    elif mode == 'down':
        hout, wout, stride, pad, ups = hin // 2, win // 2, 2, 0, 0
    else:
        raise ValueError(mode)
    M, N = n_img * hout * wout, w.shape[0]
    assert w.shape[1] == 9 * cin and x.shape[0] == n_img * hin * win
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    g = GemmArgs()
    g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    g.bias = bias.data_ptr() if bias is not None else None
    g.residual = residual.data_ptr() if residual is not None else None
    g.M, g.N, g.K = M, N, 9 * cin
    g.lda, g.ldb, g.ldc = _rows(x), _rows(w), _rows(out)
    g.ldr = _rows(residual) if residual is not None else 0
    g.a_mode, g.epi_act, g.alpha, g.res_pre = 1, act, 1.0, int(res_pre)
    g.Hin, g.Win, g.Cin, g.Hout, g.Wout = hin, win, cin, hout, wout
    g.stride, g.pad, g.ups, g.batch = stride, pad, ups, 1
    if pro is not None:
        sc, sh, pact = pro
        _chk_f32(sc, sh)
        g.pro_scale, g.pro_shift = sc.data_ptr(), sh.data_ptr()
        g.pro_ld, g.pro_act = sc.shape[1], pact
    # few pixels per image (the deep UNet levels): K split across workgroups, partial tiles in a workspace of this
    # call (stream-ordered like every torch allocation); the slice count depends on the layer's geometry only
    if ksplit is None:
        g.splitk_ws = ctypes.c_void_p(1)  # (non-NULL: ask what the library would do with a workspace)
        ksplit = _lib.load().t2h_gemm_ksplit(ctypes.byref(g))
        g.splitk_ws = None
    if ksplit is not None and ksplit > 1:
        ws = torch.empty(ksplit * M * N, device=x.device, dtype=torch.float32)
        g.splitk_ws, g.splitk_ws_floats, g.ksplit = ws.data_ptr(), ws.numel(), int(ksplit)
    else:
        g.ksplit = 1
    _launch_gemm(g, 't2h_gemm_f32(conv)')
    return out


def conv3x3_small(x, w, n_img, h, wd, cin, bias=None, pro=None, out=None):
    """3x3 'same' convolution with <= 4 output channels (conv_out) on the vector ALU, exact fp32;
    pro = (scale[n_img, C], shift[n_img, C], pro_act) as for conv3x3."""
    _chk_f32(x, w, bias, out)
    cout = w.shape[0]
    assert w.shape[1] == 9 * cin and x.shape[0] == n_img * h * wd and w.is_contiguous()
    if out is None:
        out = torch.empty((n_img * h * wd, cout), device=x.device, dtype=torch.float32)
    sc, sh, pact = pro if pro is not None else (None, None, PRO_NONE)
    _chk_f32(sc, sh)
    check(_lib.load().t2h_conv3x3_small_f32(_p(x), _rows(x), _p(w), _p(bias), _p(sc), _p(sh),
                                            sc.shape[1] if sc is not None else 0, int(pact), _p(out), _rows(out), n_img, h,
                                            wd, cin, cout, _stream()), 't2h_conv3x3_small_f32')
    return out


def conv3x3_small_ok(cin, cout, mode):
    return mode == 'same' and cout <= 4 and cin % 32 == 0


def layernorm(x, gamma, beta, out=None, eps=1e-5):
    _chk_f32(x, gamma, beta, out)
    assert x.is_contiguous()
    rows, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().t2h_layernorm_f32(_p(x), _p(gamma), _p(beta), _p(out), rows, C, eps, _stream()),
          't2h_layernorm_f32')
    return out


def split_rows_empty(rows, C, device):
    """Uninitialised split-row buffer [rows][C/32][2][32] fp16 (as int16)."""
    return torch.empty((rows, C // 32, 2, 32), device=device, dtype=torch.int16)


SPLIT_LO_SCALE = 2048.0


def split_planes_host(x):
    """(hi, lo) fp16 planes of an fp32 tensor: x = hi + lo / 2048 (round-to-nearest-even,
    bit-identical to the device split)."""
    x = x.float()
    if x.numel() and float(x.abs().max()) >= 65504.0:
        raise ValueError('split rows carry fp16 planes: |x| must stay below 65504 '
                         f'(got {float(x.abs().max()):.3g}); use the exact-fp32 kernels (T2H_SPLIT_GEMM=0)')
    hi = x.half()
    lo = ((x - hi.float()) * SPLIT_LO_SCALE).half()
    return hi, lo


def unsplit_rows_host(s, rows, C):
    """split rows (int16 view, any device) -> fp32 [rows, C] on the CPU"""
    pl = s.cpu().view(torch.float16).view(rows, C // 32, 2, 32).float()
    return (pl[:, :, 0] + pl[:, :, 1] / SPLIT_LO_SCALE).reshape(rows, C)


def split_rows(x, out=None):
    """fp32 x [rows, C] (unit inner stride) -> split rows (x = hi + lo / 2048, fp16 planes)."""
    _chk_f32(x)
    rows, C = x.shape
    if out is None:
        out = split_rows_empty(rows, C, x.device)
    check(_lib.load().t2h_split_rows_f32(_p(x), _rows(x), _p(out), rows, C, overflow_flag(), _stream()),
          't2h_split_rows_f32')
    return out


def pack_split_rows_host(w):
    """Host-side (torch CPU) repack of an fp32 matrix [N, K] into split rows."""
    hi, lo = split_planes_host(w)
    n, k = w.shape
    planes = torch.stack([hi, lo], 0).view(2, n, k // 32, 32).permute(1, 2, 0, 3).contiguous()
    return planes.view(torch.int16)


X8_TARGET_MAX = 32.0  # a tensor's calibration maximum is scaled into [16, 32): 14x headroom below e4m3's 448


def x8_scale_for(max_abs, target=X8_TARGET_MAX):
    """The power of two s with max_abs * s in [target / 2, target): the scale of a tensor's 8-bit planes (x8 format,
    csrc/common.h).  Weights use target 256 (their maximum is known exactly when they are packed)."""
    import math
    m = float(max_abs)
    if not (m > 0.0) or not math.isfinite(m):
        return 1.0
    # (never below 2^-7: 448 / s then stays under fp16's 65504, so a value the 8-bit planes can hold never overflows
    # the fp16 hi plane first)
    return max(2.0 ** (math.ceil(math.log2(target / m)) - 1), 2.0 ** -7)


def absmax_f32(x, out_bits):
    """atomicMax of the fp32 bits of max |x| into out_bits[0] (int32 / uint32 device word the caller zeroed)."""
    _chk_f32(x)
    rows, C = x.shape
    check(_lib.load().t2h_absmax_f32(_p(x), _rows(x), rows, C, _p(out_bits), _stream()), 't2h_absmax_f32')


def split_rows_absmax(rows_split, rows, C, out_bits):
    """The same over the fp16 hi plane of split rows / x8 rows."""
    check(_lib.load().t2h_split_rows_absmax(_p(rows_split), rows, C, _p(out_bits), _stream()), 't2h_split_rows_absmax')


def split_rows_x8(x, scale, out=None):
    """fp32 x [rows, C] -> x8 rows [rows][C/32][hi16 | hi8 | lo8] (same byte size as split rows), 8-bit planes
    scaled by the power of two `scale`."""
    _chk_f32(x)
    rows, C = x.shape
    if out is None:
        out = split_rows_empty(rows, C, x.device)
    check(_lib.load().t2h_split_rows_x8_f32(_p(x), _rows(x), _p(out), rows, C, float(scale), overflow_flag(), _stream()),
          't2h_split_rows_x8_f32')
    return out


def unpack_x8_rows_host(s, rows, C, scale):
    """x8 rows (int16 view, any device) -> (hi fp32 [rows, C], hi8 / scale, lo8 / scale) on the CPU: the three planes
    as the numbers they stand for (tests)."""
    raw = s.cpu().contiguous().view(torch.uint8).view(rows, C // 32, 128)
    hi = raw[:, :, :64].contiguous().view(torch.float16).float().reshape(rows, C)
    h8 = raw[:, :, 64:96].contiguous().view(torch.float8_e4m3fn).float().reshape(rows, C) / scale
    l8 = raw[:, :, 96:].contiguous().view(torch.float8_e4m3fn).float().reshape(rows, C) / scale
    return hi, h8, l8


def layernorm_x8(x, gamma, beta, out_x8, scale, eps=1e-5):
    """LayerNorm whose result is written in the x8 format."""
    _chk_f32(x, gamma, beta)
    assert x.is_contiguous()
    rows, C = x.shape
    check(_lib.load().t2h_layernorm_x8_f32(_p(x), _p(gamma), _p(beta), _p(out_x8), rows, C, eps, float(scale),
                                           overflow_flag(), _stream()), 't2h_layernorm_x8_f32')
    return out_x8


def mha_split_x8(qk_split, ld_cols, vt, B, T, n_head, out_x8, scale):
    """mha_split with the output written in the x8 format."""
    check(_lib.load().t2h_mha_split_x8_f32(_p(qk_split), ld_cols, _p(vt), _p(out_x8), float(scale), B, T, n_head,
                                           overflow_flag(), _stream()), 't2h_mha_split_x8_f32')
    return out_x8


def gemm_split(a_split, w_split, M, N, K, out=None, out_split=None, bias=None, residual=None, act=ACT_NONE,
               vt=None, vt_col0=0, vt_T=0, vt_hd=64, x8=None, out_x8_scale=None, ksplit=0):
    """C = act(A @ W^T + bias) + residual on the fp16 matrix cores at fp32-class
    accuracy; a_split / w_split are split rows.  Writes fp32 `out` and / or the
    split-row form `out_split` of the result.  With `vt` the output columns from
    `vt_col0` on go to the transposed value planes of mha_split instead
    (t2h_gemm_split_args.Vt).
    x8 = (scale_A, scale_B): the operands are x8 rows (fp16 plane + two e4m3 planes scaled by those powers of two),
    the cross terms run on the 8-bit matrix instruction; out_x8_scale: `out_split` is written in the x8 format."""
    _chk_f32(out, bias, residual)
    g = _lib.GemmSplitArgs()
    if x8 is not None:
        g.fmt = 1
        g.lo_mul = 1.0 / (SPLIT_LO_SCALE * float(x8[0]) * float(x8[1]))
    if out_x8_scale is not None:
        g.out_fmt, g.out_scale = 1, float(out_x8_scale)
    g.ksplit = int(ksplit)  # (> 1: `out` holds ksplit * M rows of partial tiles; experiment, t2h_hip.h)
    g.A, g.B = a_split.data_ptr(), w_split.data_ptr()
    g.C = out.data_ptr() if out is not None else None
    g.C_split = out_split.data_ptr() if out_split is not None else None
    g.bias = bias.data_ptr() if bias is not None else None
    g.residual = residual.data_ptr() if residual is not None else None
    g.M, g.N, g.K = M, N, K
    g.ldc = _rows(out) if out is not None else 0
    g.ldr = _rows(residual) if residual is not None else 0
    g.epi_act = act
    if vt is not None:
        g.Vt, g.vt_col0, g.vt_T, g.vt_hd = vt.data_ptr(), vt_col0, vt_T, vt_hd
    if out_split is not None or vt is not None:
        g.overflow_flag = _ovf_slot()[0].data_ptr()
    lib = _lib.load()
    if _prof is not None:
        _prof['count'] += 1
        phase = _prof['count'] % _prof['every']
        # algorithmic (fp32-equivalent) FLOPs; the kernel issues 3 fp16 products per multiply.  Two kinds of
        # samples, on DIFFERENT launches (timing a kernel through hipExtLaunchKernelGGL adds packets around it):
        if phase == 0 or phase == _prof['every'] // 2:
            cfg = lib.t2h_gemm_split_tile_config(ctypes.byref(g))
        if phase == 0:
            # (k0, k1) receive the kernel's OWN start / end (t2h_gemm_split_time_next_launch): kernel time; the same
            # launch stores its phase stamps (main loop duration + the shader clock it ran at)
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record()  # (creates the hipEvent_t handles the hook hands to the launch)
            k1.record()
            # (sized for the smallest tile any configuration uses, 64 x 64: the kernel stores 16 words per workgroup unbounded)
            stamps = torch.zeros(16 * (-(-M // 64)) * (-(-N // 64)), dtype=torch.int64, device=a_split.device)
            check(lib.t2h_gemm_split_time_next_launch(ctypes.c_void_p(k0.cuda_event), ctypes.c_void_p(k1.cuda_event)),
                  't2h_gemm_split_time_next_launch')
            check(lib.t2h_gemm_split_probe_next_launch(_p(stamps)), 't2h_gemm_split_probe_next_launch')
            check(lib.t2h_gemm_split_f32(ctypes.byref(g), _stream()), 't2h_gemm_split_f32')
            _prof['recs'].append((_split_label(g), 2.0 * M * N * K, k0, k1, 'kernel', cfg, stamps))
            return out if out is not None else out_split
        if phase == _prof['every'] // 2:
            # (e0, e1) are recorded on the stream around the launch and so run from the end of the previous
            # kernel: kernel + dependent-launch boundary
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.t2h_gemm_split_f32(ctypes.byref(g), _stream()), 't2h_gemm_split_f32')
            e1.record()
            _prof['recs'].append((_split_label(g), 2.0 * M * N * K, e0, e1, 'stream', cfg, None))
            return out if out is not None else out_split
    check(lib.t2h_gemm_split_f32(ctypes.byref(g), _stream()), 't2h_gemm_split_f32')
    return out if out is not None else out_split


def _split_label(g):
    """profile label of a split-GEMM launch: '<2xfp16>' = three fp16 partial products, '<x8>' = fp16 hi*hi + both
    cross terms in one 8-bit instruction"""
    return 'gemm_split_kernel<x8>' if g.fmt == 1 else 'gemm_split_kernel<2xfp16>'


def split_overflow_async(reset=True):
    """Enqueues the read-back (and optional clear) of the current stream's overflow word; -> the pinned host
    tensor the value lands in, valid once the stream has been synchronised.  No host wait here."""
    flag, host = _ovf_slot()
    check(_lib.load().t2h_split_overflow_async(_p(flag), _p(host), int(bool(reset)), _stream()), 't2h_split_overflow_async')
    return host


def split_overflow_bits(reset=True):
    """The sticky overflow word of the current stream's split-row producers since the last reset (include/t2h_hip.h):
    bit 0: a value outside fp16's range (|x| >= 65504); bit 1: a value outside the 8-bit planes' range of an x8
    tensor (|x| s >= 448).  Synchronises the current stream -- here, in the host code, not inside the library."""
    host = split_overflow_async(reset)
    torch.cuda.current_stream().synchronize()
    return int(host[0])


def split_overflow(reset=True):
    """True if any split-row producer flagged a value since the last reset (any bit of split_overflow_bits)."""
    return bool(split_overflow_bits(reset))


def vt_empty(B, n_head, T, device, hd=64):
    """Transposed value planes [B][H][2][hd][T] fp16 (as int16) for mha_split."""
    return torch.empty(B, n_head, 2, hd, T, dtype=torch.int16, device=device)


def vt_key_positions(T):
    """Position of key k inside the Vt planes (include/t2h_hip.h, t2h_gemm_split_args.Vt)."""
    k = torch.arange(T)
    w = k & 31
    return (k & ~31) + 16 * (w >> 4) + 8 * ((w >> 2) & 1) + 4 * ((w >> 3) & 1) + (w & 3)


def mha_split(qk_split, ld_cols, vt, B, T, n_head, out=None, out_split=None):
    """Attention with q, k read as split rows and v as transposed planes; both
    products as three fp16 partial products (fp32-class accuracy)."""
    _chk_f32(out)
    check(_lib.load().t2h_mha_split_f32(_p(qk_split), ld_cols, _p(vt), _p(out) if out is not None else None,
                                        _p(out_split) if out_split is not None else None, B, T, n_head,
                                        overflow_flag() if out_split is not None else None, _stream()), 't2h_mha_split_f32')
    return out if out is not None else out_split


def layernorm_split(x, gamma, beta, out_split, eps=1e-5):
    """LayerNorm whose result is written as split rows."""
    _chk_f32(x, gamma, beta)
    assert x.is_contiguous()
    rows, C = x.shape
    check(_lib.load().t2h_layernorm_split_f32(_p(x), _p(gamma), _p(beta), _p(out_split), rows, C, eps,
                                              overflow_flag(), _stream()), 't2h_layernorm_split_f32')
    return out_split


def mha_noncausal_split(qkv, B, T, n_head, out_split):
    """Attention whose output is written as split rows."""
    _chk_f32(qkv)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * n_head * 64)
    check(_lib.load().t2h_mha_noncausal_split_f32(_p(qkv), _p(out_split), B, T, n_head, overflow_flag(), _stream()),
          't2h_mha_noncausal_split_f32')
    return out_split


def groupnorm_tables(x, gamma, beta, n_img, hw, groups=32, eps=1e-6):
    """GroupNorm statistics of x [n_img*hw, C] -> (scale, shift) [n_img, C]."""
    _chk_f32(x, gamma, beta)
    C = gamma.shape[0]
    lib = _lib.load()
    scale = torch.empty((n_img, C), device=x.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    part = getattr(x, '_t2h_gn_part', None)
    if part is not None and tuple(part.shape[:1] + part.shape[2:]) == (n_img, 2, C) and x.shape[0] == n_img * hw:
        # the producing conv already left per-tile partial sums: no pass over x
        check(lib.t2h_groupnorm_finalize_f32(_p(part), part.shape[1], _p(gamma), _p(beta), _p(scale), _p(shift),
                                             n_img, hw, C, groups, eps, _stream()), 't2h_groupnorm_finalize_f32')
        return scale, shift
    ws = torch.empty(lib.t2h_groupnorm_workspace_bytes(n_img, hw, C) // 8, device=x.device,
                     dtype=torch.float64)
    check(lib.t2h_groupnorm_tables_f32(_p(x), _rows(x), _p(gamma), _p(beta), _p(scale), _p(shift),
                                       n_img, hw, C, groups, eps, _p(ws), _stream()),
          't2h_groupnorm_tables_f32')
    return scale, shift


def spatial_attention(qkv, n_img, n, c, out=None):
    """AttnBlock attention without the N x N tensor: qkv rows [n_img * n, 3c] (q | k | v) -> rows [n_img * n, c]."""
    _chk_f32(qkv, out)
    assert qkv.shape == (n_img * n, 3 * c) and c in (256, 512) and n % 32 == 0
    if out is None:
        out = torch.empty((n_img * n, c), device=qkv.device, dtype=torch.float32)
    check(_lib.load().t2h_spatial_attention_f32(_p(qkv), _rows(qkv), _p(out), _rows(out), n_img, n, c, float(int(c)**(-0.5)),
                                                _stream()), 't2h_spatial_attention_f32')
    return out


def spatial_attention_ok(n, c):
    return c in (256, 512) and n % 32 == 0


def softmax_rows_(x):
    _chk_f32(x)
    x2 = x.view(-1, x.shape[-1])
    check(_lib.load().t2h_softmax_rows_f32(_p(x2), x2.shape[0], x2.shape[1], x2.stride(0), _stream()),
          't2h_softmax_rows_f32')
    return x


def embed_sum4(idx, segm, tex, tok_emb, pos_emb, segm_emb, tex_emb, out=None):
    _chk_i64(idx, segm, tex)
    _chk_f32(tok_emb, pos_emb, segm_emb, tex_emb, out)
    B, T = idx.shape
    C = tok_emb.shape[1]
    if out is None:
        out = torch.empty((B * T, C), device=idx.device, dtype=torch.float32)
    check(_lib.load().t2h_embed_sum4_f32(_p(idx), _p(segm), _p(tex), _p(tok_emb), _p(pos_emb),
                                         _p(segm_emb), _p(tex_emb), _p(out), B, T, C, _stream()),
          't2h_embed_sum4_f32')
    return out


def mha_noncausal(qkv, B, T, n_head, out=None):
    _chk_f32(qkv, out)
    assert qkv.is_contiguous() and qkv.shape == (B * T, 3 * n_head * 64)
    if out is None:
        out = torch.empty((B * T, n_head * 64), device=qkv.device, dtype=torch.float32)
    check(_lib.load().t2h_mha_noncausal_f32(_p(qkv), _p(out), B, T, n_head, _stream()),
          't2h_mha_noncausal_f32')
    return out


def unmask_step(rand, t, unmasked, changes, tex, head_count, changed_rows=None, n_heads=0):
    """head_count: int32 [n_heads (+1)]; with changed_rows (int32 [n]) the changed rows are
    appended to it and counted in head_count[n_heads]."""
    _chk_f32(rand)
    n = rand.numel()
    if changed_rows is not None:
        assert changed_rows.dtype == torch.int32 and changed_rows.numel() >= n
        assert head_count.numel() > n_heads
    check(_lib.load().t2h_unmask_step(_p(rand), int(t), _p(unmasked), _p(changes), _p(tex),
                                      _p(head_count), n, _p(changed_rows), int(n_heads), _stream()),
          't2h_unmask_step')


def torch_draw_geometry(numel, device=None):
    """(threads of the grid ATen launches for an elementwise random draw of `numel` floats, generator
    offset increment of that draw) -- calc_execution_policy of ATen/native/cuda/DistributionTemplates.h:
    block 256, grid = min(#CU * (max threads per CU / 256), ceil(numel / 256)), unroll 4."""
    prop = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device())
    grid = min(prop.multi_processor_count * (prop.max_threads_per_multi_processor // 256), (numel + 255) // 256)
    return 256 * grid, ((numel - 1) // (256 * grid * 4) + 1) * 4


def philox_exponential(seed, offset, numel, device):
    """What `torch.empty(numel, device=device).exponential_()` returns on a generator with this seed /
    offset, computed element by element (t2h_philox_exponential_f32)."""
    out = torch.empty(numel, device=device, dtype=torch.float32)
    gt, _ = torch_draw_geometry(numel, device)
    check(_lib.load().t2h_philox_exponential_f32(int(seed), int(offset), gt, _p(out), numel, _stream()),
          't2h_philox_exponential_f32')
    return out


def philox_uniform(seed, offset, numel, device):
    """What `torch.rand(numel, device=device)` returns on a generator with this seed / offset
    (t2h_philox_uniform_f32)."""
    out = torch.empty(numel, device=device, dtype=torch.float32)
    gt, _ = torch_draw_geometry(numel, device)
    check(_lib.load().t2h_philox_uniform_f32(int(seed), int(offset), gt, _p(out), numel, _stream()),
          't2h_philox_uniform_f32')
    return out


def unmask_schedule(seed, offset, tex, steps, n_heads, n_class):
    """The whole unmasking schedule of a sampling run on torch's device generator (seed, offset before
    the first draw) in one launch (t2h_unmask_schedule).  tex int64 [n].  Returns
    (step_of_row int32 [n], head_mask int32 [steps + 1], rand_inc, expo_inc) -- device tensors."""
    _chk_i64(tex)
    n = tex.numel()
    dev = tex.device
    rand_gt, rand_inc = torch_draw_geometry(n, dev)
    _, expo_inc = torch_draw_geometry(n * n_class, dev)
    step_of_row = torch.empty(n, dtype=torch.int32, device=dev)
    head_mask = torch.empty(steps + 1, dtype=torch.int32, device=dev)
    check(_lib.load().t2h_unmask_schedule(int(seed), int(offset), rand_gt, rand_inc, expo_inc, _p(tex), n, int(steps),
                                          int(n_heads), _p(step_of_row), _p(head_mask), _stream()),
          't2h_unmask_schedule')
    return step_of_row, head_mask, rand_inc, expo_inc


def schedule_advance(rows_tbl, aux64_tbl, aux32_tbl, round_ctr, cur_rows, cur_aux64, cur_aux32, maxr):
    """Round cursor (t2h_schedule_advance): cur_* <- round *round_ctr of the padded tables; *round_ctr += 1."""
    check(_lib.load().t2h_schedule_advance(_p(rows_tbl), _p(aux64_tbl), _p(aux32_tbl), _p(round_ctr), _p(cur_rows),
                                           _p(cur_aux64), _p(cur_aux32), int(maxr), _stream()), 't2h_schedule_advance')


def gather_rows(src, rows, n_rows, out=None):
    """out[i] = src[rows[i]] (rows int32 on the device, first n_rows used); src 2-D+ contiguous, any dtype
    whose row is a multiple of 16 bytes."""
    assert src.is_contiguous() and rows.dtype == torch.int32
    row_bytes = src[0].numel() * src.element_size()
    if out is None:
        out = torch.empty((int(n_rows), ) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check(_lib.load().t2h_gather_rows(_p(src), _p(rows), _p(out), int(n_rows), row_bytes, _stream()), 't2h_gather_rows')
    return out


def sample_heads(hidden, lnf_g, lnf_b, w_heads, expo_by_head, rows, n_rows, tex, temp, x_t, out_idx, split=True,
                 philox=None, hidden_compact=False, row_noise=None, logits_ws=None):
    """All heads in one launch: `rows` (int32, first n_rows valid) are the changed token
    rows, expo_by_head {head: [n, n_class] Exp(1) draw}, w_heads [n_heads, n_class, C],
    out_idx [n_heads, n].  philox = (seed, {head: generator offset}): the noise of the listed heads is
    computed in the kernel as the corresponding elements of torch's full-tensor exponential_ draws
    instead of being read from expo_by_head.  row_noise (row lists that mix sampling steps):
    ('philox', seed, offsets int64 [>= n_rows][, rng_rows int32 [>= n_rows]]) = per-listed-row generator offsets (and
    the rows of the reference's draw they belong to, if the samples were reordered), or
    ('explicit', expo_rows f32 [*, n_class], slots int32 [>= n_rows]) = per-listed-row explicit draws."""
    _chk_f32(hidden, lnf_g, lnf_b, w_heads, *expo_by_head.values())
    if int(n_rows) == 0:
        return
    C = hidden.shape[1]
    n = out_idx.shape[1]
    assert hidden.shape[0] == (int(n_rows) if hidden_compact else n)
    n_heads, n_class = w_heads.shape[0], w_heads.shape[1]
    assert w_heads.is_contiguous() and out_idx.is_contiguous() and out_idx.shape == (n_heads, n)
    a = _lib.SampleHeadsArgs()
    a.hidden, a.lnf_gamma, a.lnf_beta, a.w_heads = (hidden.data_ptr(), lnf_g.data_ptr(), lnf_b.data_ptr(),
                                                    w_heads.data_ptr())
    for h, e in expo_by_head.items():
        assert e.shape == (n, n_class) and e.is_contiguous()
        a.expo[h] = e.data_ptr()
    a.rows, a.tex, a.x_t, a.out_idx = rows.data_ptr(), tex.data_ptr(), x_t.data_ptr(), out_idx.data_ptr()
    a.temp, a.n_rows, a.n, a.C, a.n_class, a.n_heads = float(temp), int(n_rows), n, C, n_class, n_heads
    a.hidden_compact = int(bool(hidden_compact))
    if (split or philox is not None or hidden_compact or row_noise is not None) and n_rows > 0:
        # logits scratch: 8 workgroups per row share the weight stream
        ws = logits_ws if logits_ws is not None else torch.empty((int(n_rows), n_class), device=hidden.device,
                                                                 dtype=torch.float32)
        assert ws.numel() >= int(n_rows) * n_class and ws.dtype == torch.float32
        a.logits_ws = ws.data_ptr()
    if row_noise is not None:
        if row_noise[0] == 'philox':
            _, seed, offs = row_noise[:3]
            assert offs.dtype == torch.int64 and offs.is_cuda and offs.numel() >= int(n_rows)
            a.row_philox_offset = offs.data_ptr()
            if len(row_noise) > 3 and row_noise[3] is not None:  # rows of the reference's draw (reordered samples)
                rr = row_noise[3]
                assert rr.dtype == torch.int32 and rr.is_cuda and rr.numel() >= int(n_rows)
                a.rng_rows = rr.data_ptr()
            if torch.is_tensor(seed):  # the seed lives in device memory (graph replay)
                assert seed.dtype == torch.int64 and seed.is_cuda and seed.numel() == 1
                a.philox_seed_dev = seed.data_ptr()
            else:
                a.philox_seed = int(seed)
            a.philox_grid_threads = torch_draw_geometry(n * n_class, hidden.device)[0]
        else:
            _, erows, slots = row_noise
            _chk_f32(erows)
            assert erows.is_contiguous() and erows.shape[-1] == n_class
            assert slots.dtype == torch.int32 and slots.is_cuda and slots.numel() >= int(n_rows)
            a.expo_rows, a.expo_slot = erows.data_ptr(), slots.data_ptr()
    elif philox is not None:
        seed, offsets = philox
        a.philox_seed = int(seed)
        for h, off in offsets.items():
            a.philox_offset[h] = int(off)
        a.philox_grid_threads = torch_draw_geometry(n * n_class, hidden.device)[0]
    check(_lib.load().t2h_sample_heads(ctypes.byref(a), _stream()), 't2h_sample_heads')


def sample_head(hidden, lnf_g, lnf_b, w_head, expo, changes, tex, head, temp, x_t, out_idx):
    _chk_f32(hidden, lnf_g, lnf_b, w_head, expo)
    n, C = hidden.shape
    n_class = w_head.shape[0]
    assert expo.shape == (n, n_class) and expo.is_contiguous() and w_head.is_contiguous()
    check(_lib.load().t2h_sample_head(_p(hidden), _p(lnf_g), _p(lnf_b), _p(w_head), _p(expo),
                                      _p(changes), _p(tex), int(head), float(temp), _p(x_t),
                                      _p(out_idx), n, C, n_class, _stream()), 't2h_sample_head')


def q_sample(x0, u, t, num_timesteps, mask_id):
    """x0 int64 [B, T], u f32 [B, T] uniform draw, t int64 [B] -> (x_t int64 [B, T], mask uint8 [B, T])."""
    _chk_i64(x0, t)
    _chk_f32(u)
    B, T = x0.shape
    x_t = torch.empty_like(x0)
    mask = torch.empty((B, T), dtype=torch.uint8, device=x0.device)
    check(_lib.load().t2h_q_sample(_p(x0.contiguous()), _p(u.contiguous()), _p(t.contiguous()), int(num_timesteps),
                                   int(mask_id), _p(x_t), _p(mask), B, T, _stream()), 't2h_q_sample')
    return x_t, mask


def masked_ce_heads(hidden, lnf_g, lnf_b, w_heads, tex, mask, gt_lists, B, T):
    """-> (ce_rows f32 [B*T], ce_samples f32 [B]): summed 18-head cross entropy of the masked tokens."""
    _chk_f32(hidden, lnf_g, lnf_b, w_heads)
    _chk_i64(tex, gt_lists)
    n, C = hidden.shape
    n_heads, n_class = w_heads.shape[0], w_heads.shape[1]
    assert n == B * T and gt_lists.shape == (n_heads, n) and gt_lists.is_contiguous() and w_heads.is_contiguous()
    ce_rows = torch.empty(n, dtype=torch.float32, device=hidden.device)
    ce_samples = torch.empty(B, dtype=torch.float32, device=hidden.device)
    check(_lib.load().t2h_masked_ce_heads(_p(hidden), _p(lnf_g), _p(lnf_b), _p(w_heads), _p(tex), _p(mask),
                                          _p(gt_lists), _p(ce_rows), _p(ce_samples), B, T, C, n_class, n_heads,
                                          _stream()), 't2h_masked_ce_heads')
    return ce_rows, ce_samples


def vq_l2_argmin(z, codebook):
    _chk_f32(z, codebook)
    assert z.is_contiguous() and codebook.is_contiguous()
    n, d = z.shape
    idx = torch.empty((n, ), device=z.device, dtype=torch.int64)
    check(_lib.load().t2h_vq_l2_argmin_f32(_p(z), _p(codebook), _p(idx), n, codebook.shape[0], d,
                                           _stream()), 't2h_vq_l2_argmin_f32')
    return idx


def vq_argmin_tex(z, books, tex, fold_hw=None):
    """Texture-routed codebook argmin (encode side).  z: latent rows [n, d], or with
    fold_hw=(h, w) the NHWC map rows [B*2h*2w, d/4] whose 2x2 patches are quantised;
    books [n_books, n_e, d]; tex int64 [n].  Returns idx_lists int64 [n_books, n] (-1
    where the row is not of that texture)."""
    _chk_f32(z, books)
    _chk_i64(tex)
    nb, n_e, d = books.shape
    n = tex.numel()
    fh, fw = fold_hw or (0, 0)
    assert z.is_contiguous() and books.is_contiguous()
    assert z.numel() == n * d, (tuple(z.shape), n, d)
    out = torch.empty((nb, n), device=z.device, dtype=torch.int64)
    check(_lib.load().t2h_vq_argmin_tex_f32(_p(z), _p(books), _p(tex), _p(out), n, nb, n_e, d, fh, fw,
                                            _stream()), 't2h_vq_argmin_tex_f32')
    return out


def codebook_gather_tex(idx_lists, tex, books):
    """idx_lists [18, n] i64, tex [n] i64, books [18, n_e, e_dim] -> [n, e_dim]."""
    _chk_i64(idx_lists, tex)
    _chk_f32(books)
    nb, n_e, e_dim = books.shape
    n = tex.numel()
    out = torch.empty((n, e_dim), device=books.device, dtype=torch.float32)
    check(_lib.load().t2h_codebook_gather_tex_f32(_p(idx_lists), _p(tex), _p(books), _p(out), n, nb,
                                                  n_e, e_dim, _stream()),
          't2h_codebook_gather_tex_f32')
    return out


def codebook_gather_fold(idx_lists, tex, books, B, h, w):
    """books [18, n_e, C*4] -> NHWC rows [B*2h*2w, C]."""
    _chk_i64(idx_lists, tex)
    _chk_f32(books)
    nb, n_e, e4 = books.shape
    C = e4 // 4
    out = torch.empty((B * 2 * h * 2 * w, C), device=books.device, dtype=torch.float32)
    check(_lib.load().t2h_codebook_gather_fold_f32(_p(idx_lists), _p(tex), _p(books), _p(out), B, h, w,
                                                   nb, n_e, C, _stream()),
          't2h_codebook_gather_fold_f32')
    return out


def routed_head_argmax(feat, w, b, tex, n_heads, cf, n_class):
    """feat [n, n_heads*cf]; w [n_heads, n_class, cf]; b [n_heads, n_class]
    -> out_lists [n_heads, n] i64 (-1 off-texture)."""
    _chk_f32(feat, w, b)
    _chk_i64(tex)
    n = feat.shape[0]
    out = torch.empty((n_heads, n), device=feat.device, dtype=torch.int64)
    check(_lib.load().t2h_routed_head_argmax(_p(feat), _rows(feat), _p(w), _p(b), _p(tex), _p(out), n,
                                             n_heads, cf, n_class, _stream()),
          't2h_routed_head_argmax')
    return out


def onehot_nhwc(segm, n_cls, cpad):
    _chk_f32(segm)
    n_pix = segm.numel()
    out = torch.empty((n_pix, cpad), device=segm.device, dtype=torch.float32)
    check(_lib.load().t2h_onehot_nhwc_f32(_p(segm.contiguous()), _p(out), n_pix, n_cls, cpad, _stream()),
          't2h_onehot_nhwc_f32')
    return out


def nchw_to_nhwc(x, cpad=None):
    """x [B,C,H,W] -> rows [B*H*W, cpad or C] (extra channels zero)."""
    _chk_f32(x)
    B, C, H, W = x.shape
    ld = cpad or C
    out = (torch.zeros if ld != C else torch.empty)((B * H * W, ld), device=x.device,
                                                   dtype=torch.float32)
    check(_lib.load().t2h_nchw_to_nhwc_f32(_p(x.contiguous()), _p(out), B, C, H * W, ld, _stream()),
          't2h_nchw_to_nhwc_f32')
    return out


def nhwc_to_nchw(x, B, H, W, C=None):
    _chk_f32(x)
    C = C or x.shape[1]
    out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    check(_lib.load().t2h_nhwc_to_nchw_f32(_p(x), _rows(x), _p(out), B, C, H * W, _stream()),
          't2h_nhwc_to_nchw_f32')
    return out


def maxpool2(x, B, H, W):
    _chk_f32(x)
    C = x.shape[1]
    out = torch.empty((B * (H // 2) * (W // 2), C), device=x.device, dtype=torch.float32)
    check(_lib.load().t2h_maxpool2_nhwc_f32(_p(x), _rows(x), _p(out), B, H, W, C, _stream()),
          't2h_maxpool2_nhwc_f32')
    return out


def bilinear_up2(x, B, H, W):
    _chk_f32(x)
    assert x.is_contiguous()
    C = x.shape[1]
    out = torch.empty((B * 4 * H * W, C), device=x.device, dtype=torch.float32)
    check(_lib.load().t2h_bilinear_up2_nhwc_f32(_p(x), _p(out), B, H, W, C, _stream()),
          't2h_bilinear_up2_nhwc_f32')
    return out


def argmax_rows(x, n=None):
    _chk_f32(x)
    rows = x.shape[0]
    n = n or x.shape[1]
    out = torch.empty((rows, ), device=x.device, dtype=torch.int64)
    check(_lib.load().t2h_argmax_rows_f32(_p(x), _rows(x), _p(out), rows, n, _stream()),
          't2h_argmax_rows_f32')
    return out


def image_epilogue(dec, B, H, W, want_u8=False):
    """dec rows [B*H*W, >=3] -> (img f32 [B,3,H,W] in [0,1], u8 [B,H,W,3] or None)."""
    _chk_f32(dec)
    img = torch.empty((B, 3, H, W), device=dec.device, dtype=torch.float32)
    u8 = torch.empty((B, H, W, 3), device=dec.device, dtype=torch.uint8) if want_u8 else None
    check(_lib.load().t2h_image_epilogue(_p(dec), _rows(dec), _p(img), _p(u8), B, H * W, _stream()),
          't2h_image_epilogue')
    return img, u8


def texture_map(segm, upper, lower, outer):
    """segm i64 [B,1,H,W] -> f32 mask [B,1,H,W]."""
    _chk_i64(segm, upper, lower, outer)
    B = segm.shape[0]
    hw = segm.numel() // B
    mask = torch.empty(segm.shape, device=segm.device, dtype=torch.float32)
    check(_lib.load().t2h_texture_map(_p(segm), _p(upper), _p(lower), _p(outer), _p(mask), B, hw,
                                      _stream()), 't2h_texture_map')
    return mask


def shape_attr_embed(attr, emb):
    """attr i64 [B, n_attr]; emb = dict of packed tensors (weights.pack_shape_embedder)."""
    _chk_i64(attr)
    B, n_attr = attr.shape
    out = torch.empty((B, emb['out_dim']), device=attr.device, dtype=torch.float32)
    check(_lib.load().t2h_shape_attr_embed_f32(
        _p(attr), _p(emb['cls_off']), _p(emb['w0t']), _p(emb['b0']), _p(emb['w1']), _p(emb['b1']),
        _p(emb['f0']), _p(emb['fb0']), _p(emb['f1']), _p(emb['fb1']), _p(out), B, n_attr, emb['dim'],
        emb['out_dim'], _stream()), 't2h_shape_attr_embed_f32')
    return out


def tap_bias_map(tapc, B, H, W, cout):
    """tapc f32 [B, cout, 9] -> rows [B*H*W, cout]."""
    _chk_f32(tapc)
    assert tapc.is_contiguous()
    out = torch.empty((B * H * W, cout), device=tapc.device, dtype=torch.float32)
    check(_lib.load().t2h_tap_bias_map_f32(_p(tapc), _p(out), B, H, W, cout, _stream()),
          't2h_tap_bias_map_f32')
    return out
