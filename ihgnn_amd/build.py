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
SOURCES = [os.path.join(CSRC, name) for name in ('host.hip', 'aggregate.hip', 'interact.hip', 'split_arith.hip', 'split_node.hip', 'narrow.hip', 'dense.hip', 'tail.hip', 'eval.hip')]
HEADERS = [os.path.join(REPO, 'include', 'ihgnn_hip.h'), os.path.join(CSRC, 'common.hpp'), os.path.join(CSRC, 'split.hpp'), os.path.join(CSRC, 'split_common.hpp'), os.path.join(CSRC, 'ablate.hpp'), os.path.join(CSRC, 'narrow.hpp')]
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
    # one compile per translation unit, side by side (the two split-arithmetic files are 12 - 20 s each, the rest seconds), then the link
    import concurrent.futures
    import tempfile
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function', '-I', os.path.join(REPO, 'include'), '-I', CSRC]
    if verbose:
        flags.insert(0, '-Rpass-analysis=kernel-resource-usage')
    with tempfile.TemporaryDirectory(prefix='ihgnn_build_') as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.basename(src) + '.o')
            cmd = [hipcc()] + flags + ['-c', '-o', obj, src]
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + proc.stdout + proc.stderr)
            return obj, proc.stderr
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
            results = list(pool.map(compile_one, SOURCES))
        if verbose:
            for _, err in results:
                sys.stderr.write(err)
        cmd = [hipcc(), f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + [obj for obj, _ in results]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError('hipcc (link) failed:\n' + ' '.join(cmd) + '\n' + proc.stdout + proc.stderr)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='-v' in sys.argv))
