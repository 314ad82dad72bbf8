// ldb_jit.h — run-time specialisation of pipeline kernels with hiprtc.
// The reference compiles every query pipeline with LLVM (src/execution/LLVMBackends.cpp:219-406);
// the MI355X runtime keeps ONE hand-written kernel source per operator (ldb_*_kernel.h) and, for
// large inputs, re-compiles it with the launch descriptor's metadata as a compile-time constant so
// the compiler folds the type/op dispatch and unrolls the descriptor loops.
#pragma once
#include "ldb_internal.h"
#include "ldb_gb_kernel.h"

// true when a launch over n_rows rows should use a specialised kernel
// (env LDB_JIT=0 disables, LDB_JIT_MIN_ROWS overrides the default threshold of 4 M rows)
bool ldb_jit_wanted(int64_t n_rows);

// Generic entry: compile (or fetch from the cache) a module made of
//    #include "<header>";  constexpr <struct_name> LDB_META = <meta bytes>;  <kernels_src>
// and return `kernel_name` from it, loaded into `device` (one module per device and architecture).  `meta` must be pointer-free metadata (addresses stripped with
// the helpers below).  Returns nullptr (reason in *why) when hiprtc is unavailable or the
// compilation fails — callers then launch their generic ahead-of-time kernel.
hipFunction_t ldb_jit_kernel(int device, const char* header, const char* struct_name, const char* kernels_src, const char* kernel_name, const void* meta, size_t meta_bytes,
                             std::string* why);
// compile only (no device needed, nothing cached): used by the device-less build check
bool ldb_jit_compile_only(const char* header, const char* struct_name, const char* kernels_src, const void* meta, size_t meta_bytes, std::string* log);
// per-operator compile checks with a representative descriptor (ldb_scan.hip, ldb_join.hip)
bool ldb_scan_jit_check(std::string* log);
bool ldb_join_jit_check(std::string* log);
bool ldb_expr_jit_check(std::string* log);
// addresses → presence flags (0/1); what remains of a column / predicate / key set is metadata
void ldb_jit_strip_col(DCol& c);
void ldb_jit_strip_pred(DPred& p);
void ldb_jit_strip_keys(DKeys& k);

// specialised group-by kernel for the metadata of `h`
hipFunction_t ldb_jit_groupby(int device, const DGroupBy* h, std::string* why);
hipFunction_t ldb_jit_groupby_kernel(int device, const DGroupBy* h, const char* kernel, std::string* why); // any kernel of the group-by translation unit

// statistics for tests / bench: kernels compiled, cache hits, total compile milliseconds
extern "C" int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms);
extern "C" int32_t ldb_gpu_jit_wait(int64_t timeout_ms, int64_t* pending);
extern "C" int32_t ldb_gpu_jit_info(int64_t* vals, int32_t n);
