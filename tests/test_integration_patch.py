"""f1, producer half: the emitter extensions are a real diff against the reference's tool and lowering (integration/mlir-subop-to-json.patch), not
prose.  Where the reference checkout exists the patch must apply cleanly (`patch --dry-run`: nothing is written); everywhere it must carry every
extension the consumer's manifest check asks for and emit the fields the golden dumps (tools/write_subop_dumps*.py) contain."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "mlir-subop-to-json.patch")
REF = "/root/reference"


def added_lines():
    with open(PATCH) as f:
        return [l[1:] for l in f if l.startswith("+") and not l.startswith("+++")]


def test_patch_emits_what_the_consumer_reads():
    text = "\n".join(added_lines())
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import write_subop_dumps as W

    # the manifest the patched tool writes is the one the dump writers write
    for e in W.MANIFEST["extensions"]:
        assert '"%s"' % e in text, e
    assert '"emitter_manifest"' in text and "gpu-manifest" in text
    # every extension field of the golden dumps has an emitter line
    for field in ("lowerInclusive", "upperInclusive", "sortBy", "maxRows", "direction", "primaryKey", "combine_tuple", "aggregates", "keys", "materialized", '" - "'):
        assert field in text, field
    # and the golden dumps use nothing beyond them: any key of a sub-operator that the unpatched tool does not print is one of the extension fields
    known = {"ref", "type", "outerEdges", "accesses", "subop", "subops", "inputs", "results", "innerEdges", "meta", "resultType", "stateType", "mapping", "elem", "reference",
             "generated", "computed", "semantic", "columns", "renamed", "leftRef", "rightRef", "between", "offset", "newRef", "optionalRef", "updated", "operator", "streams"}
    ext = {"sortBy", "maxRows", "aggregates", "keys", "materialized"}
    gold = os.path.join(ROOT, "tests", "golden")

    def subops(node):
        for op in node.get("subops", []):
            yield op
            yield from subops(op)

    for name in sorted(os.listdir(gold)):
        if name.startswith("subop_") and name.endswith(".json"):
            with open(os.path.join(gold, name)) as f:
                doc = json.load(f)
            for step in doc[:-1]:
                for op in subops(step):
                    assert set(op) <= known | ext, (name, sorted(set(op) - known - ext))


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference checkout and patch(1)")
def test_patch_applies_to_the_reference_checkout():
    with open(PATCH) as f:
        r = subprocess.run(["patch", "--dry-run", "-p1", "-d", REF], stdin=f, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tools/ct/mlir-subop-to-json.cpp" in r.stdout and "RelAlgToSubOp.cpp" in r.stdout and "FAILED" not in r.stdout and "fuzz" not in r.stdout


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference checkout and patch(1)")
def test_backend_patch_applies_on_top_of_the_emitter_patch(tmp_path):
    """integration/gpu-execution-backend.patch — the reference-side ExecutionBackend for ExecutionMode::GPU (src/execution/Execution.cpp:428-431 selects it,
    the step walk it replaces is SubOpToControlFlow.cpp:4363-4395) — is a diff on top of the emitter patch: both are applied, in order, to a scratch
    copy of the files they touch; the backend source must call the runtime through the entry points the headers declare"""
    import re

    backend = os.path.join(ROOT, "integration", "gpu-execution-backend.patch")
    touched = set()
    for path in (PATCH, backend):
        with open(path) as f:
            for line in f:
                m = re.match(r"^(?:---|\+\+\+) [ab]/(\S+)", line)
                if m:
                    touched.add(m.group(1))
    assert "src/execution/HIPOperatorBackend.cpp" in touched and "src/execution/Execution.cpp" in touched
    for rel in touched:
        src = os.path.join(REF, rel)
        if os.path.exists(src):
            os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
            shutil.copy(src, tmp_path / rel)
    for path in (PATCH, backend):
        with open(path) as f:
            r = subprocess.run(["patch", "-p1", "-d", str(tmp_path)], stdin=f, capture_output=True, text=True)
        assert r.returncode == 0 and "FAILED" not in r.stdout, r.stdout + r.stderr
    src = (tmp_path / "src/execution/HIPOperatorBackend.cpp").read_text()
    with open(os.path.join(ROOT, "include", "lingodb_gpu.h")) as f:
        abi = f.read()
    with open(os.path.join(ROOT, "lingo-db_amd", "host", "ldb_host.hpp")) as f:
        abi += f.read()
    called = set(re.findall(r"\b(ldb_[a-z0-9_]+)\(", src))
    assert {"ldb_subop_translate", "ldb_plan_prepare", "ldb_plan_execute", "ldb_gpu_table_load_ipc", "ldb_gpu_export"} <= called
    for fn in called:
        assert re.search(r"\b%s\(" % fn, abi), fn + " is not declared by the runtime's headers"
    tool = (tmp_path / "tools/ct/mlir-subop-to-json.cpp").read_text()
    assert "planToJson" in tool and "LINGODB_SUBOP_TO_JSON_NO_MAIN" in tool and '" - "' in tool
