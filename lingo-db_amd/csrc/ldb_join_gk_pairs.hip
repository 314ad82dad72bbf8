// ldb_join_gk_pairs.hip — generic (ahead-of-time) join kernel(s), one translation unit per kernel so that the library
// builds in parallel: each of these instantiates ldb_join_kernel.h with every layout / pipeline branch live at run time,
// which is minutes of register allocation apiece.  Declared in ldb_join.hip; bodies in ldb_join_kernel.h.
#include "ldb_internal.h"
#include "ldb_join_kernel.h"

__global__ void k_join_probe_pairs(const DJoin* __restrict__ d) { join_probe_pairs_body(*d, d); }
