"""Builds libodrift_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

The library is nine translation units (csrc/odr_comm.hip, csrc/odrift.hip, odr_step.hip, odr_step_noise.hip, odr_step_fast.hip, odr_step_fast_noise.hip,
odr_step_mix.hip, odr_mix.hip, odr_step_tile.hip) compiled in parallel and
linked into one shared object; objects are rebuilt only when a source they include changed."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
UNITS = ['odr_step_noise.hip', 'odr_step_fast_noise.hip', 'odr_step.hip', 'odr_step_fast.hip', 'odrift.hip', 'odr_step_mix.hip', 'odr_mix.hip',
         'odr_step_tile.hip', 'odr_comm.hip']
HEADERS = [os.path.join(CSRC, f) for f in ('odr_host.h', 'odr_step_launch.h', 'odr_kernels.hip.h', 'odr_field.hip.h', 'odr_geodesic.hip.h',
                                           'odr_oil.hip.h', 'odr_mesh.h', 'odr_tile.hip.h')] + \
    [os.path.join(os.path.dirname(HERE), 'include', 'odrift.h')]
LIB = os.path.join(HERE, 'libodrift_hip.so')
OBJDIR = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
         '-I' + os.path.join(os.path.dirname(HERE), 'include')]


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, extra_flags=(), lib=LIB, objdir=OBJDIR):
    """extra_flags / lib / objdir: A/B builds of the same sources (tools/vbuild.sh)."""
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for u in UNITS:
        src = os.path.join(CSRC, u)
        obj = os.path.join(objdir, u.replace('.hip', '.o'))
        if force or _stale(obj, [src] + HEADERS):
            jobs.append([HIPCC] + FLAGS + list(extra_flags) + ['-c', src, '-o', obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(' '.join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(len(jobs)) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, u.replace('.hip', '.o')) for u in UNITS]
    if jobs or _stale(lib, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    import sys
    print(build(force='--force' in sys.argv, verbose=True))


def build_variant(tag, units, extra_flags, outdir=None):
    """A/B library tools/_lib<tag>.so: only `units` are compiled with `extra_flags`, the other translation units are the
    default build's objects (tools/vbuild_tu.sh; a what-if flag usually touches one kernel family)."""
    build()
    outdir = outdir or os.path.join(os.path.dirname(HERE), 'tools')
    objdir = os.path.join(outdir, '_obj' + tag)
    os.makedirs(objdir, exist_ok=True)

    def run(u):
        obj = os.path.join(objdir, u.replace('.hip', '.o'))
        subprocess.check_call([HIPCC] + FLAGS + list(extra_flags) + ['-c', os.path.join(CSRC, u), '-o', obj])
        return obj
    with ThreadPoolExecutor(len(units)) as ex:
        mine = dict(zip(units, ex.map(run, units)))
    objs = [mine.get(u, os.path.join(OBJDIR, u.replace('.hip', '.o'))) for u in UNITS]
    lib = os.path.join(outdir, '_lib%s.so' % tag)
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs)
    return lib
