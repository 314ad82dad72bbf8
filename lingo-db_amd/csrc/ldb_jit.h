// ldb_jit.h — run-time specialisation of pipeline kernels with hiprtc.
// The reference compiles every query pipeline with LLVM (src/execution/LLVMBackends.cpp:219-406);
// the MI355X runtime keeps ONE hand-written kernel source per operator and, for large inputs,
// re-compiles it with the launch descriptor's metadata as a compile-time constant so the
// compiler folds the type/op dispatch and unrolls the descriptor loops.
#pragma once
#include "ldb_internal.h"
#include "ldb_gb_kernel.h"

// true when a launch over n_rows rows should use a specialised kernel
// (env LDB_JIT=0 disables, LDB_JIT_MIN_ROWS overrides the default threshold of 4 M rows)
bool ldb_jit_wanted(int64_t n_rows);

// Specialised group-by kernel for the metadata of `h` (addresses / sizes are NOT baked in).
// Returns nullptr (and records the reason in *why) when hiprtc is unavailable or compilation
// fails — the caller then launches the generic ahead-of-time kernel.
hipFunction_t ldb_jit_groupby(const DGroupBy* h, std::string* why);

// statistics for tests / bench: kernels compiled, cache hits, total compile milliseconds
extern "C" int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms);
