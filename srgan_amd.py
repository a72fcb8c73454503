"""Import shim: makes the package that lives in ``sr-gan_amd/`` (a directory name Python cannot import
directly because of the hyphen) importable as ``srgan_amd``.  After ``import srgan_amd`` the name refers
to the real package (this module replaces itself in ``sys.modules``)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_package_dir = os.path.join(_here, 'sr-gan_amd')
_spec = importlib.util.spec_from_file_location('srgan_amd', os.path.join(_package_dir, '__init__.py'),
                                               submodule_search_locations=[_package_dir])
_module = importlib.util.module_from_spec(_spec)
sys.modules['srgan_amd'] = _module
_spec.loader.exec_module(_module)
