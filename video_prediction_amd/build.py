"""Build libsavp_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so that the
library travels with the repository snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
OUT = os.path.join(HERE, 'libsavp_hip.so')
BUILD = os.path.join(CSRC, 'build')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-Wno-unused-value'] + \
        os.environ.get('SAVP_EXTRA_FLAGS', '').split()


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def build(force=False, verbose=False):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(BUILD, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    headers = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)] + \
              [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(BUILD, s[:-4] + '.o')
        objs.append(obj)
        if force or _newer(src, obj, headers):
            jobs.append([hipcc] + FLAGS + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(OUT) or force:
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-ldl'])
    build_io(force=force, verbose=verbose)
    return OUT


def build_io(force=False, verbose=False):
    """libsavp_io.so: the host-side C++ input pipeline (include/savp_io.h), plain g++."""
    src = os.path.join(HERE, 'csrc_host', 'tfrecord_pipeline.cpp')
    out = os.path.join(HERE, 'libsavp_io.so')
    hdr = os.path.join(INCLUDE, 'savp_io.h')
    if force or _newer(src, out, (hdr,)):
        cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-pthread', '-I' + INCLUDE, src, '-o', out]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('g++ failed:\n' + r.stdout)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
