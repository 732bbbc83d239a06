"""Builds libodrift_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'odrift.hip')
DEPS = [SRC] + [os.path.join(HERE, 'csrc', f) for f in
                ('odr_kernels.hip.h', 'odr_field.hip.h', 'odr_geodesic.hip.h', 'odr_oil.hip.h', 'odr_mesh.h')] + \
    [os.path.join(os.path.dirname(HERE), 'include', 'odrift.h')]
LIB = os.path.join(HERE, 'libodrift_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')


def build(force=False, verbose=False):
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS)):
        return LIB
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
           '-I' + os.path.join(os.path.dirname(HERE), 'include'), '-o', LIB, SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
