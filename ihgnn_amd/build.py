"""Build recipe for libihgnn_hip.so (hipcc, gfx950 only, in-tree so the .so travels with the snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(CSRC, 'libihgnn_hip.so')
SOURCES = [os.path.join(CSRC, name) for name in ('host.hip', 'aggregate.hip', 'interact.hip', 'split_arith.hip', 'dense.hip', 'tail.hip', 'eval.hip')]
HEADERS = [os.path.join(REPO, 'include', 'ihgnn_hip.h'), os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'split.hpp'), os.path.join(CSRC, 'ablate.hpp')]
ARCH = 'gfx950'


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libihgnn_hip.so cannot be built (ROCm toolchain required)')
    return exe


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if missing or older than its sources; returns its path."""
    if not force and not stale():
        return LIB
    cmd = [hipcc(), f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall',
           '-Wno-unused-function', '-I', os.path.join(REPO, 'include'), '-I', CSRC, '-o', LIB] + SOURCES
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + proc.stdout + proc.stderr)
    if verbose:
        sys.stderr.write(proc.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
