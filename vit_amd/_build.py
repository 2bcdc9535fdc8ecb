"""Build libvitk.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Called by ``__graft_entry__.build()`` and by ``python -m vit_amd._build``.  Objects are
rebuilt only when their source (or common.h) is newer; the .so lives in ``vit_amd/lib`` so
it travels with the tree to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libvitk.so')
SOURCES = ['runtime.hip', 'gemm.hip', 'gemm256.hip', 'norm_act.hip', 'attention.hip', 'attention_bwd.hip', 'attention_band.hip', 'attention_lai.hip', 'favor.hip', 'favor_fast.hip', 'optim.hip', 'loss.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-Wno-unused-result']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def build(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdr_mtime = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith('.h'))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_mtime):
            jobs.append([hipcc, *FLAGS, '-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB, *objs])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
