#!/usr/bin/env bash
# Builds oracle/_ref/libldb_ref.so from the REFERENCE'S OWN sources where they lie under
# /root/reference (never copied into this repo) + the glue/shims in this directory.
# Only the runtime files that compile offline are used; the reference's own build
# system (cmake + LLVM/MLIR 20.1 + Boost + bison/flex) is NOT run — the full system is unbuildable
# here.  Output goes to oracle/_ref/ only (git-ignored, travels to the GPU box with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LDB_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/../_ref"
[ -d "$REF/src/runtime" ] || { echo "reference checkout not found at $REF" >&2; exit 3; }
PA_INC="$(python3 -c 'import pyarrow as pa; print(pa.get_include())')"
PA_LIB="$(python3 -c 'import pyarrow as pa; print(pa.get_library_dirs()[0])')"
mkdir -p "$OUT/obj"
CXX="${CXX:-g++}"
FLAGS="-std=c++20 -O2 -fPIC -DENABLE_REFCOUNT=1 -I$HERE/shim -I$REF/include -I$REF/vendored -I$PA_INC -w"
SRCS=(
  src/runtime/Hash.cpp
  src/runtime/helpers.cpp
  src/runtime/ArrowView.cpp
  src/runtime/storage/Restrictions.cpp
  src/runtime/Buffer.cpp
  src/runtime/GrowingBuffer.cpp
  src/runtime/LazyJoinHashtable.cpp
  src/runtime/PreAggregationHashtable.cpp
  src/runtime/ThreadLocal.cpp
  src/runtime/ExecutionContext.cpp
  src/runtime/Sorting.cpp
  src/runtime/Heap.cpp
  src/runtime/SimpleState.cpp
  src/runtime/Hashtable.cpp
  src/runtime/HashMultiMap.cpp
  src/runtime/SegmentTreeView.cpp
  src/runtime/StringRuntime.cpp
  src/runtime/ListRuntime.cpp
  src/runtime/DateRuntime.cpp
  src/utility/Tracer.cpp
  src/utility/Setting.cpp
)
OBJS=()
for s in "${SRCS[@]}"; do
  o="$OUT/obj/$(echo "$s" | tr '/' '_').o"
  if [ ! -f "$o" ] || [ "$REF/$s" -nt "$o" ]; then $CXX $FLAGS -c "$REF/$s" -o "$o"; fi
  OBJS+=("$o")
done
for s in sched_shim.cpp ref_glue.cpp; do
  o="$OUT/obj/$s.o"
  # (-fno-access-control: the glue reads HashMultiMap's private entry / value structs as the generated code does by offset)
  $CXX $FLAGS -fno-access-control -c "$HERE/$s" -o "$o"
  OBJS+=("$o")
done
ARROW_SO="$(ls "$PA_LIB"/libarrow.so.* | head -1)"
$CXX -shared -fPIC -o "$OUT/libldb_ref.so" "${OBJS[@]}" "$ARROW_SO" -Wl,-rpath,"$PA_LIB" -lpthread
echo "built $OUT/libldb_ref.so"
