"""Builds libsrgan_hip.so (every HIP translation unit in csrc/) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting shared object is
git-ignored but travels to the GPU box with the working tree."""
import hashlib
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


def source_id():
    """sha256 over the names and contents of every csrc/*.hip and *.h: the identity of the kernel sources.  It is
    compiled into the library (``srgan_build_id()``) and compared by ``_lib.library()`` when the library is loaded, so
    a stale prebuilt .so is detected by content, not by file times (which do not survive the copy to the GPU box)."""
    digest = hashlib.sha256()
    for path in sorted(_sources() + _headers()):
        digest.update(os.path.basename(path).encode())
        with open(path, 'rb') as handle:
            digest.update(handle.read())
    return digest.hexdigest()[:16]


def library_id(path=LIBRARY):
    """The source id a built library carries (read from the file: no GPU, no dlopen needed)."""
    try:
        with open(path, 'rb') as handle:
            blob = handle.read()
    except OSError:
        return None
    marker = b'SRGAN_BUILD_ID='
    at = blob.find(marker)
    return blob[at + len(marker):at + len(marker) + 16].decode('ascii', 'replace') if at >= 0 else None


def is_current():
    return os.path.exists(LIBRARY) and library_id() == source_id()


def build(force=False, verbose=True):
    """Compile (if stale) and return the library path.  Raises on compiler failure."""
    if not force and is_current():
        return LIBRARY
    hipcc = os.environ.get('HIPCC', 'hipcc')
    headers = hashlib.sha256()
    for path in sorted(_headers()):
        with open(path, 'rb') as handle:
            headers.update(handle.read())

    def compile_one(source):
        """An object is reused only when the sidecar next to it records the same source + header contents + flags."""
        obj, sidecar = source[:-4] + '.o', source[:-4] + '.o.id'
        with open(source, 'rb') as handle:
            wanted = hashlib.sha256(handle.read() + headers.digest() + ' '.join(FLAGS).encode()).hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(sidecar) and open(sidecar).read() == wanted:
            return obj
        if verbose:
            print('[srgan_amd] hipcc', os.path.basename(source), flush=True)
        subprocess.check_call([hipcc] + FLAGS + ['-c', source, '-o', obj])
        with open(sidecar, 'w') as handle:
            handle.write(wanted)
        return obj

    with ThreadPoolExecutor(max_workers=4) as pool:
        objects = list(pool.map(compile_one, _sources()))
    # the identity of the sources, as a tiny translation unit of its own (rebuilt every time anything changed)
    stamp_source, stamp_object = os.path.join(CSRC, 'build_id.cc'), os.path.join(CSRC, 'build_id.o')
    with open(stamp_source, 'w') as handle:
        handle.write('extern "C" const char* srgan_build_id(void) { static const char id[] = "SRGAN_BUILD_ID=%s"; '
                     'return id + 15; }\n' % source_id())
    subprocess.check_call([hipcc, '-O1', '-fPIC', '-c', '-x', 'c++', stamp_source, '-o', stamp_object])
    os.remove(stamp_source)
    objects.append(stamp_object)
    # The library NEEDs libamdhip64.so.7 -- the SONAME of the HIP runtime PyTorch-ROCm bundles.  _lib.library()
    # imports torch first, so the loader binds us to that already-loaded runtime (one runtime per process: it
    # owns the allocations and streams we are handed).
    # Linked under a temporary name and renamed into place: a rank that waits for the file never maps a half-written one.
    temporary = f'{LIBRARY}.tmp.{os.getpid()}'
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objects + ['-o', temporary])
    os.replace(temporary, LIBRARY)
    return LIBRARY
