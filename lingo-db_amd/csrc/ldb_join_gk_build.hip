// ldb_join_gk_build.hip — generic (ahead-of-time) join kernel(s), one translation unit per kernel so that the library
// builds in parallel: each of these instantiates ldb_join_kernel.h with every layout / pipeline branch live at run time,
// which is minutes of register allocation apiece.  Declared in ldb_join.hip; bodies in ldb_join_kernel.h.
#include "ldb_internal.h"
#include "ldb_join_kernel.h"

__global__ void k_join_build(const DJoin* __restrict__ d) { join_build_body(*d, d); }
__global__ void k_join_key_range(const DJoin* __restrict__ d, long long* __restrict__ out) { join_key_range_body(*d, d, out); }
__global__ void k_join_key_bits(const DJoin* __restrict__ d) { join_key_bits_body(*d, d); }
__global__ void k_join_rank_bits(const DJoin* __restrict__ d) { join_rank_bits_body(*d, d); }
__global__ void k_join_rank_perm(const DJoin* __restrict__ d) { join_rank_perm_body(*d, d); }
