"""The C-ABI library on the CPU (no GPU, no compute calls): it loads, exports every entry point include/srgan_hip.h
declares (and the ctypes table binds exactly those), reports its capabilities, refuses a stale build, and rejects bad
arguments before touching a device."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'srgan_hip.h')


def declared_entry_points():
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.DOTALL)
    return sorted(set(re.findall(r'\b(srgan_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    import srgan_amd  # noqa: F401
    from srgan_amd import _lib
    return _lib


def test_every_declared_entry_point_is_exported_and_bound(lib):
    from srgan_amd import _build
    declared = declared_entry_points()
    assert len(declared) >= 45
    exported = subprocess.run(['nm', '-D', '--defined-only', _build.LIBRARY], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r'\bT (srgan_[a-z0-9_]+)$', exported, flags=re.MULTILINE)))
    assert exported == declared, (sorted(set(declared) - set(exported)), sorted(set(exported) - set(declared)))
    assert sorted(lib.SIGNATURES) == declared           # the Python binding covers the whole ABI
    library = lib.library()
    for name in declared:
        assert getattr(library, name) is not None


def test_capabilities_and_build_identity(lib):
    from srgan_amd import _build
    caps = lib.capabilities()
    assert caps.abi_version == lib.library().srgan_version() == 110
    assert caps.struct_bytes == ctypes.sizeof(lib.Capabilities) and caps.arch == b'gfx950'
    assert caps.dtypes == 0x7 and caps.features & 0x7 == 0x7
    assert caps.workspace_bytes == lib.library().srgan_workspace_bytes() == 256 << 20      # (round 5: the partial tiles of every K split)
    assert caps.max_tensor_elements == 2 ** 31 - 1
    assert lib.library().srgan_build_id().decode() == _build.source_id() == _build.library_id()
    assert lib.library().srgan_capabilities(None, 0) == lib.EINVAL


def test_a_library_built_from_other_sources_is_refused(lib, monkeypatch):
    """a stale prebuilt .so must not run silently -- the loader compares the source id compiled into the
    library with the kernel sources next to it."""
    monkeypatch.setattr(lib, '_library', None)
    monkeypatch.setattr(lib, 'source_id', lambda: '0123456789abcdef')
    monkeypatch.delenv('SRGAN_ALLOW_STALE_LIBRARY', raising=False)
    with pytest.raises(lib.HipLibraryError, match='other kernel sources'):
        lib.library()


def test_argument_errors_are_reported_before_any_device_work(lib):
    library = lib.library()
    huge = lib.ConvDesc(4096, 512, 1024, 1024, 64, 3, 3, 1, 1, 1, 1, 1024, 1024, 0, 0, 0)        # 2^41 input elements
    assert library.srgan_conv2d_fwd(ctypes.byref(huge), 16, 16, None, 16, 0, None) == lib.ERANGE
    assert b'2^31' in library.srgan_last_error()
    malformed = lib.ConvDesc(1, 8, 8, 8, 8, 3, 3, 0, 1, 1, 1, 8, 8, 0, 0, 0)                       # stride 0
    assert library.srgan_conv2d_fwd(ctypes.byref(malformed), 16, 16, None, 16, 0, None) == lib.EINVAL
    wrong_dtype = lib.ConvDesc(1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 1, 8, 8, 0, 0, 7)
    assert library.srgan_conv2d_fwd(ctypes.byref(wrong_dtype), 16, 16, None, 16, 0, None) == lib.EINVAL
    assert library.srgan_conv2d_bnrelu_supported(ctypes.byref(huge), 0) == 0
    assert library.srgan_set_workspace(16, 1024, None) == lib.EINVAL                              # smaller than required
    assert library.srgan_set_workspace(8, 256 << 20, None) == lib.EINVAL                          # misaligned
    # the RCCL entry points check their arguments before they resolve / reach RCCL
    assert library.srgan_comm_unique_id(None) == lib.EINVAL
    assert library.srgan_comm_init(None, 1, 0, None) == lib.EINVAL
    holder = ctypes.c_void_p()
    assert library.srgan_comm_init(ctypes.byref(holder), 2, 2, b'\0' * 128) == lib.EINVAL          # rank outside the world
    assert library.srgan_all_reduce_sum(None, 16, 16, 4, 0, None) == lib.EINVAL
    assert library.srgan_reduce_scatter_sum(16, 16, 16, 4, 5, None) == lib.EINVAL                 # unknown wire dtype
    assert library.srgan_comm_available() in (0, 1)
