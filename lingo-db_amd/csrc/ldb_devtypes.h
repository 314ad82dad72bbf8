// ldb_devtypes.h — plain-old-data descriptors shared by host code and device kernels.
// No host headers: this file (with ldb_device.h / ldb_keys.h / ldb_gb_kernel.h) is also compiled
// at run time by hiprtc when a pipeline kernel is specialised (ldb_jit.hip).
// Device addresses are carried as uint64_t so every descriptor is pointer-free and can be
// turned into a compile-time constant with __builtin_bit_cast in a specialised kernel.
#pragma once
#ifdef __HIPCC_RTC__
#define LDB_NO_STD_HEADERS
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
typedef unsigned long size_t;
#define INT64_MAX 9223372036854775807L
#define INT64_MIN (-9223372036854775807L - 1)
#endif
#include "lingodb_gpu.h"

// One column as the kernels see it.  `rowids` belongs to the relation side the column is read
// through (0 = identity), so a kernel reads logical row i at values[rowids ? rowids[i] : i].
struct DCol {
   uint64_t values; // device address
   uint64_t offsets; // utf8: const int64_t*
   uint64_t validity; // Arrow bitmap or 0
   uint64_t rowids; // const uint32_t* or 0
   int32_t type; // ldb_type
   int32_t width; // bytes per value on the device
   int32_t precision;
   int32_t scale;
};

#define LDB_MAX_PREDS 8
#define LDB_MAX_IN 8
#define LDB_STR_INLINE 48
#define LDB_LIKE_MAX_SEG 4 // literal segments of a "simple" LIKE pattern (host: ldb_like_plan)
struct DPred {
   DCol col;
   DCol rhs;
   int32_t op;
   int32_t rhs_kind;
   uint64_t lo;
   int64_t hi;
   double f;
   int32_t str_len;
   int32_t n_in;
   char str[LDB_STR_INLINE];
   // IN lists: ints as lo/hi; strings packed into in_blob with in_off[k]..in_off[k+1]
   uint64_t in_lo[LDB_MAX_IN];
   int64_t in_hi[LDB_MAX_IN];
   int32_t in_off[LDB_MAX_IN + 1];
   int32_t same_col; // lhs column identical to the previous conjunct's (host: ldb_mark_same_col)
   char in_blob[LDB_MAX_IN * 16];
   // zone map of the column (ldb_column_zones): int64 min / max per LDB_ZONE_ROWS physical rows, or 0.  Attached only to
   // column-vs-constant comparisons over a dense column whose zones are selective (clustered / sorted data): a row whose
   // zone cannot satisfy the comparison fails without its value being loaded.
   uint64_t zmin, zmax;
};
#define LDB_ZONE_SHIFT 14
#define LDB_ZONE_ROWS (1u << LDB_ZONE_SHIFT)

#define LDB_MAX_KEYS 8
struct DKeys {
   int32_t n_keys;
   int32_t pad;
   DCol cols[LDB_MAX_KEYS];
};
