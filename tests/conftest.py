import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lingo-db_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

# GPU test modules that are re-run with every kernel specialised at run time (hiprtc) and every
# filter fused lazily into its consumer — the code path the SF100 bench numbers come from, which
# the default thresholds (>= 4 M / >= 1 M rows) would never reach at test sizes.
SPEC_MODULES = {"test_gpu_parity", "test_gpu_joins_more", "test_gpu_tpch_more", "test_gpu_z_golden", "test_gpu_z_tpch_q10", "test_gpu_tpch_new",
                "test_gpu_new_ops"}


# GPU suite: specialisations compile synchronously, so that the 'spec' passes below launch the specialised kernels on their FIRST call (the library
# default — jit_async = 1 — answers "still compiling" and launches the generic kernel; tests/test_gpu_jit_async.py covers that mode)
os.environ.setdefault("LDB_JIT_ASYNC", "0")
os.environ.setdefault("LDB_JIT_MIN_ROWS", "4000000")  # the library default is 256 K rows since round 6: at test sizes that would compile hundreds of shapes one after the other

try:  # torch first: it ships its own HIP runtime / RCCL copies, which must be the ones the process binds (see api.Comm)
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        import ctypes as C

        from lingodb_amd import capi

        h = C.c_void_p()
        st = capi.gpu_lib().ldb_gpu_ctx_create(0, None, C.byref(h))
        if st == capi.LDB_OK:
            capi.gpu_lib().ldb_gpu_ctx_destroy(h)
            return True, ""
        return False, capi.gpu_lib().ldb_gpu_last_error().decode(errors="replace")
    except Exception as e:  # library not built
        return False, str(e)


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    ok, why = _gpu_available()
    if ok:
        return
    skip = pytest.mark.skip(reason=f"no MI355X / HIP library here ({why}); GPU tests run with -m gpu on the GPU box")
    for it in gpu_items:
        it.add_marker(skip)


def pytest_generate_tests(metafunc):
    if "kernel_mode" in metafunc.fixturenames and metafunc.module.__name__ in SPEC_MODULES:
        metafunc.parametrize("kernel_mode", ["generic", "spec"], indirect=True, scope="module")


@pytest.fixture(scope="module")
def kernel_mode(request):
    """'generic': the library's defaults (ahead-of-time kernels at test sizes); 'spec': thresholds 0 —
    every launch specialised by hiprtc, every base-table filter lazy and fused into its consumer."""
    mode = getattr(request, "param", "generic")
    if mode == "generic":
        yield mode
        return
    import ctypes as C

    from lingodb_amd import capi

    lib = capi.gpu_lib()
    if mode == "spec":
        lib.ldb_gpu_set_option(b"jit_min_rows", 0)
        lib.ldb_gpu_set_option(b"lazy_min_rows", 0)
        n0, h0, ms0 = C.c_int64(), C.c_int64(), C.c_double()
        lib.ldb_gpu_jit_stats(C.byref(n0), C.byref(h0), C.byref(ms0))
    yield mode
    if mode == "spec":
        n1, h1, ms1 = C.c_int64(), C.c_int64(), C.c_double()
        lib.ldb_gpu_jit_stats(C.byref(n1), C.byref(h1), C.byref(ms1))
        lib.ldb_gpu_set_option(b"jit_min_rows", 4000000)
        lib.ldb_gpu_set_option(b"lazy_min_rows", 1 << 20)
        assert n1.value + h1.value > n0.value + h0.value, "spec mode ran without a single specialised kernel launch"


@pytest.fixture(autouse=True)
def _apply_kernel_mode(kernel_mode):
    # every test depends on kernel_mode; only the modules listed in SPEC_MODULES get both modes
    yield


@pytest.fixture(scope="session")
def oracle():
    import oracle_bind

    return oracle_bind.load()


@pytest.fixture(scope="session")
def ctx():
    import lingodb_amd as ldb

    c = ldb.Context(0)
    yield c
    c.close()
