"""Build check: no kernel of libt2h_hip.so may need scratch memory (register spills).

A spill anywhere in a kernel -- even in a path that never runs -- makes every launch of it pay for
the scratch set-up: in round 2 a peeled loop tail of the 256x128 split GEMM spilled 423 registers and
the q|k|v / fc1 launches went from 33 to 45 us.  text2human_amd.build keeps hipcc's
-Rpass-analysis=kernel-resource-usage report next to each object; this test parses it."""
import os
import re

from text2human_amd import build as t2h_build


def test_no_kernel_uses_scratch_memory():
    t2h_build.build(verbose=False)
    seen = 0
    for src in t2h_build.SOURCES:
        log = os.path.join(t2h_build.OBJ_DIR, src.replace('.hip', '.o') + '.resources.log')
        if not os.path.exists(log):  # object older than this check: rebuild that file
            os.remove(os.path.join(t2h_build.OBJ_DIR, src.replace('.hip', '.o') + '.sha'))
            t2h_build.build(verbose=False)
        txt = open(log).read()
        for m in re.finditer(r'Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?VGPRs Spill: (\d+)', txt, re.S):
            seen += 1
            assert int(m.group(2)) == 0 and int(m.group(3)) == 0, \
                f'{src}: kernel {m.group(1)} uses {m.group(2)} B/lane of scratch ({m.group(3)} spilled VGPRs)'
    assert seen >= 40, f'only {seen} kernels found in the resource reports'
