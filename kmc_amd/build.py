"""Build kmc_amd/libkmc_hip.so in-tree with hipcc for gfx950 (cross-compiles
without a GPU).  ``python -m kmc_amd.build``"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'kmc_hip.hip')
OUT = os.path.join(HERE, 'libkmc_hip.so')
DEPS = [SRC] + [os.path.join(HERE, 'csrc', f) for f in ('kmc_device.h', 'kmc_hop_lds.h', 'kmc_hop_lds64.h', 'kmc_hop_q64.h', 'kmc_hop_lat.h', 'kmc_hop_global.h', 'kmc_hop_big.h', 'kmc_exact_par.h')] + \
       [os.path.join(os.path.dirname(HERE), 'include', 'kmc_hip.h')]
# -enable-ipra=0: the hop kernels call their (rare) exact path as a real function; with inter-procedural
# register allocation the callee's large clobber set makes the kernels spill scalar registers inside the
# hop loop (measured: 4.81 -> 4.55 ms per launch of hop_q64_kernel without it)
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-shared',
         '-Wno-unused-value', '-Wno-inline-asm', '-mllvm', '-enable-ipra=0']


def source_sha256():
    """Hash of everything the library is built from (sources, headers, flags): stands in for the commit id where
    there is no .git (the GPU box), so that bench.py can tell whether a committed counter pass (profiles/) was
    taken on the code it is timing."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        h.update(os.path.basename(d).encode() + b'\0')
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = [hipcc()] + FLAGS + ['-o', OUT, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    if '--source-sha256' in sys.argv:
        print(source_sha256())
    else:
        build(force='--force' in sys.argv, verbose=True)
