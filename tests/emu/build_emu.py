"""Builds a HOST shared library out of a product kernel's source for tests/emu/hip_emu.h (test infrastructure).

The kernel file and csrc/common.h are used as they are, minus what only a GPU can assemble: the
`#include <hip/hip_runtime.h>` line, and with their few inline-assembly idioms (16-byte loads / write-through stores,
LDS-DMA requests, counted waits) replaced by what they mean (`rules` below; an idiom without a rule fails the build).  Output: tests/emu/_build/lib<name>_emu.so exporting the kernel file's extern "C"
entry points with their product signatures."""
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'text2human_amd', 'csrc')
OUT = os.path.join(HERE, '_build')


def _host_clang():
    for cand in ('/opt/rocm/lib/llvm/bin/clang++', shutil.which('amdclang++'), shutil.which('clang++')):
        if cand and os.path.exists(cand):
            return cand
    return None


def available():
    return _host_clang() is not None


def build(kernel_file, transform=None, tag=''):
    """-> path of the emulation library of csrc/<kernel_file> (rebuilt when a source changes).  transform(text) -> text
    edits the kernel file's text first (negative controls: a dropped barrier, a wait count one too permissive); tag
    names that build."""
    name = kernel_file.replace('.hip', '') + tag
    common = open(os.path.join(CSRC, 'common.h')).read()
    common = common.replace('#include <hip/hip_runtime.h>', '').replace('#pragma once', '')
    common = common.replace('#include "../../include/t2h_hip.h"', f'#include "{os.path.join(ROOT, "include", "t2h_hip.h")}"')
    common = common.replace('void t2h_set_error(const char* fmt, ...);', '')
    kern = open(os.path.join(CSRC, kernel_file)).read().replace('#include "common.h"', '').replace('#include <hip/hip_ext.h>', '')
    if transform is not None:
        kern = transform(kern)
    # the inline-assembly idioms of common.h and of the kernel files, restated for the host: a 16-byte global load /
    # write-through store becomes a copy (synchronous here), an LDS-DMA request a copy into the emulated LDS
    # (lane-linear, 16 bytes per lane), a counted wait and an opaque register copy become nothing
    rules = [
        # explicit vector-memory requests and counted waits -> the emulator's in-order request queue (hip_emu.h: executed
        # at once, or -- emu_set_deferred(1) -- only when a wait forces them: the latest the hardware may land them)
        (r'asm volatile\("global_load_dwordx4 %0, %1, %2" : "=v"\((\w+)\) : "v"\((\w+)\), "s"\((\w+)\) : "memory"\);',
         r'emu_vm_issue(&\1, \3 + \2, 16);'),
        (r'asm volatile\("global_load_dwordx4 %0, %1, off" : "=v"\((\w+)\) : "v"\((\w+)\) : "memory"\);',
         r'emu_vm_issue(&\1, \2, 16);'),
        (r'asm volatile\("global_store_dwordx4 %0, %1, off sc1\\n\\ts_nop 1" ::"v"\((\w+)\), "v"\((\w+)\) : "memory"\);',
         r'memcpy(\1, &\2, 16);'),
        (r'asm volatile\("global_store_dwordx2 %0, %1, off sc1\\n\\ts_nop 0" ::"v"\((\w+)\), "v"\((\w+)\) : "memory"\);',
         r'memcpy(\1, &\2, 8);'),
        (r'asm volatile\("s_mov_b32 %0, m0[^;]*global_load_lds_dwordx4[^;]*: "=&s"\(keep\) : "v"\((\w+)\), "s"\((\w+)\), "s"\(dst\) : "memory"\);',
         r'emu_vm_issue(lds_dst + (emu_tid & 63) * 16, \2 + \1, 16);'),
        (r'asm volatile\("s_waitcnt vmcnt\(%1\)" : "\+v"\(\w+\) : "n"\((\w+)\)\);', r'emu_vm_wait(\1);'),
        (r'asm volatile\("s_waitcnt vmcnt\(%0\)" ::"n"\((\w+)\) : "memory"\);', r'emu_vm_wait(\1);'),
        (r'asm volatile\("s_waitcnt vmcnt\(0\)" ::: "memory"\);', 'emu_vm_wait(0);'),
        (r'asm volatile\("" : "\+v"\(\w+\)\);', ';'),
        # gemm_split.hip / attention.hip: LDS addresses are absolute there (the address of the kernel's LDS array, LDS_NAME
        # below): offsets from that array here
        (r'\(unsigned\)\(uintptr_t\)LDS_NAME\b', '0u'),
        (r'asm volatile\("s_mov_b32 %0, m0[^;]*?global_load_lds_dwordx4[^;]*?: "=&s"\(keep\)\s*: "v"\(([\w\[\]]+)\), "s"\((\w+)\), "s"\((\w+)\)\s*: "memory"\);',
         r'emu_vm_issue(LDS_NAME + \3 + (emu_tid & 63) * 16, \2 + \1, 16);'),
        # a wave waiting for its OWN LDS traffic is a point all of its lanes are at (the hardware runs them in lockstep;
        # attention.hip signals other waves right behind it): a wave-level rendezvous here.  (Only where every lane of
        # the wave reaches it, which holds for the two places that use this form.)
        (r'asm volatile\("s_waitcnt lgkmcnt\(0\)" ::: "memory"\);', 'emu_wave_sync();'),
        (r'asm volatile\("" ::: "memory"\);', '__atomic_thread_fence(__ATOMIC_SEQ_CST);'),
        (r'asm volatile\(""[^;]*?\);', ';'),   # empty templates (ablation stubs of never-defined switches, register pins)
    ]
    m = re.search(r'\(unsigned\)\(uintptr_t\)(smem\w*)', kern)
    lds_name = m.group(1) if m else 'smem'
    for pat, rep in rules:
        pat, rep = pat.replace('LDS_NAME', lds_name), rep.replace('LDS_NAME', lds_name)
        common = re.sub(pat, rep, common)
        kern = re.sub(pat, rep, kern)
    # dynamic LDS: the whole 160 KiB (one workgroup is alive at a time)
    kern = re.sub(r'extern __shared__ (?:__attribute__\(\(aligned\(\d+\)\)\) )?(\w+) (\w+)\[\];',
                  r'static __attribute__((aligned(16))) \1 \2[163840 / sizeof(\1)];', kern)
    for name_, text in (('common.h', common), (kernel_file, kern)):
        assert 'asm volatile' not in text, f'an inline-assembly idiom of {name_} has no host restatement'
    src = f'#include "{os.path.join(HERE, "hip_emu.h")}"\n' + common + '\n' + kern
    dig = hashlib.sha256((src + open(os.path.join(HERE, 'hip_emu.h')).read()).encode()).hexdigest()[:16]
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f'lib{name}_emu_{dig}.so')
    if os.path.exists(lib):
        return lib
    for old in os.listdir(OUT):  # earlier builds of this kernel file
        if old.startswith(f'lib{name}_emu_') and old.endswith('.so'):
            os.remove(os.path.join(OUT, old))
    cpp = os.path.join(OUT, f'{name}_emu.cpp')
    with open(cpp, 'w') as f:
        f.write(src)
    subprocess.run([_host_clang(), '-O1', '-std=c++17', '-fPIC', '-shared', '-pthread', '-Wno-everything', cpp, '-o', lib],
                   check=True)
    return lib


def load(kernel_file):
    """-> ctypes handle of the emulation library of csrc/<kernel_file>, every product entry point it exports typed
    from text2human_amd._lib.SIGNATURES (the table the product's own loader uses)"""
    import ctypes
    import sys
    sys.path.insert(0, ROOT)
    from text2human_amd import _lib
    so = ctypes.CDLL(build(kernel_file))
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(so, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, args
    so.emu_last_error.restype = ctypes.c_char_p
    return so
