"""Builds libsrgan_hip.so (every HIP translation unit in csrc/) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting shared object is
git-ignored but travels to the GPU box with the working tree."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIBRARY = os.path.join(CSRC, 'libsrgan_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-I' + CSRC]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]


def is_current():
    if not os.path.exists(LIBRARY):
        return False
    built = os.path.getmtime(LIBRARY)
    return all(os.path.getmtime(p) <= built for p in _sources() + _headers())


def build(force=False, verbose=True):
    """Compile (if stale) and return the library path.  Raises on compiler failure."""
    if not force and is_current():
        return LIBRARY
    hipcc = os.environ.get('HIPCC', 'hipcc')
    newest_header = max([os.path.getmtime(h) for h in _headers()] or [0.0])

    def compile_one(source):
        obj = source[:-4] + '.o'
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(source)
                and os.path.getmtime(obj) >= newest_header):
            return obj
        if verbose:
            print('[srgan_amd] hipcc', os.path.basename(source), flush=True)
        subprocess.check_call([hipcc] + FLAGS + ['-c', source, '-o', obj])
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objects = list(pool.map(compile_one, _sources()))
    # The library NEEDs libamdhip64.so.7 -- the SONAME of the HIP runtime PyTorch-ROCm bundles.  _lib.library()
    # imports torch first, so the loader binds us to that already-loaded runtime (one runtime per process: it
    # owns the allocations and streams we are handed).
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objects + ['-o', LIBRARY])
    return LIBRARY
