"""lingodb_amd — ctypes harness over liblingodb_gpu.so (MI355X-native LingoDB sub-operator runtime).

The package directory is `lingo-db_amd/` (not importable as-is because of the hyphen); put that
directory on sys.path and `import lingodb_amd` (tests/conftest.py, bench.py and
__graft_entry__.py do exactly that).
"""
from . import capi  # noqa: F401
from .api import *  # noqa: F401,F403
