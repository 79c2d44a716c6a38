"""Builds synthsr_amd/libsynthsr_hip.so (gfx950 only) with hipcc, in-tree.

    python -m synthsr_amd.build [--force]

generator.hip is compiled with -ffp-contract=off (bit-exact label indexing, see the file header).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libsynthsr_hip.so')
ARCH = 'gfx950'

SOURCES = [('generator.hip', ['-ffp-contract=off']),
           ('unet_pointwise.hip', []),
           ('ssim.hip', []),
           ('critic.hip', []),
           ('conv_bf16.hip', []),
           ('conv_split.hip', []),
           ('conv3d.hip', [])]


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    deps = [os.path.join(CSRC, 'common.h'), os.path.join(os.path.dirname(HERE), 'include', 'synthsr_hip.h'),
            os.path.abspath(__file__)]
    objs = []
    procs = []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(CSRC, name.replace('.hip', '.o'))
        objs.append(obj)
        if force or _newer([src] + deps, obj):
            cmd = [hipcc, '-x', 'hip', '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj] + extra
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('compile failed: ' + ' '.join(cmd))
    if force or procs or _newer(objs, LIB):
        cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
