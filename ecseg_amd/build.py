"""Build libecseg_hip.so (gfx950) in-tree with hipcc.  ``python -m ecseg_amd.build [--force]``."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libecseg_hip.so')
SOURCES = ['api.hip', 'unet_kernels.hip', 'layer_kernels.hip', 'wino4_kernel.hip', 'wino4s_kernel.hip', 'wino4r_kernel.hip', 'convs_kernel.hip', 'wino16_kernel.hip', 'post_kernels.hip', 'host_codec.cpp', 'host_io.cpp', 'comm.hip']
# packed f32 VALU ops stall the SIMD beside MFMAs: keep the transform arithmetic of the Winograd kernels scalar
EXTRA_FLAGS = {'wino4_kernel.hip': ['-fno-slp-vectorize'], 'wino4s_kernel.hip': ['-fno-slp-vectorize'], 'wino4r_kernel.hip': ['-fno-slp-vectorize']}
HEADERS = [os.path.join(CSRC, f) for f in ('common.h', 'device_util.h', 'wino4_consts.inc', 'wino4_region.inc', 'wino4_combine.inc', 'wino4_head.inc')] + \
    [os.path.join(HERE, '..', 'include', 'ecseg_hip.h')]


def source_hash():
    """sha256 over the device-side sources of the library: ties a committed PMC summary (tools/pmc_summary.py) to the kernels
    it was measured on - bench.py reports a summary of other sources as stale instead of quoting it."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.h', '.inc')))
    for f in names:
        h.update(f.encode() + b'\0')
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library next to this file."""
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + '.o')
        cmd = [_hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function'] + \
            EXTRA_FLAGS.get(s, []) + ['-c', os.path.join(CSRC, s), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-lz', '-ldl']     # zlib: PNG / Deflate-TIFF in host_io.cpp; dl: RCCL is loaded on first use (comm.hip)
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
