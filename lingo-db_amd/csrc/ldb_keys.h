// ldb_keys.h — key-column sets: hashing and equality on the device.
#pragma once
#include "ldb_device.h"

#define LDB_MAX_KEYS 8
struct DKeys {
   int32_t n_keys;
   int32_t pad;
   DCol cols[LDB_MAX_KEYS];
};

// db.hash(keys) of logical row i; NULL parts are skipped (reference LowerToStd.cpp:1118-1132).
// `*any_null` reports whether some key part is NULL (join keys with NULL never match).
__device__ inline uint64_t d_hash_keys(const DKeys& k, uint64_t i, bool* any_null = nullptr) {
   uint64_t total = 0; // combine(h, 0) == h, so the empty running hash is 0
   bool nul = false;
   const int nk = k.n_keys;
   for (int j = 0; j < nk; j++) {
      uint32_t row = d_phys_row(k.cols[j], i);
      if (!d_valid(k.cols[j], row)) {
         nul = true;
         continue;
      }
      total = d_hash_part(k.cols[j], row, total);
   }
   if (any_null) *any_null = nul;
   return total;
}

// all key parts equal between logical row ia of ka and ib of kb.
// nulls_equal: group-by semantics (NULL = NULL, `isa`); else join semantics (createEqFn,
// reference src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:1035-1066).
__device__ inline bool d_keys_equal(const DKeys& ka, uint64_t ia, const DKeys& kb, uint64_t ib, bool nulls_equal) {
   const int nk = ka.n_keys;
   for (int j = 0; j < nk; j++) {
      uint32_t ra = d_phys_row(ka.cols[j], ia), rb = d_phys_row(kb.cols[j], ib);
      bool va = d_valid(ka.cols[j], ra), vb = d_valid(kb.cols[j], rb);
      if (!va || !vb) {
         if (nulls_equal && !va && !vb) continue;
         return false;
      }
      if (!d_key_part_equal(ka.cols[j], ra, kb.cols[j], rb)) return false;
   }
   return true;
}

int32_t ldb_make_dkeys(const ldb_rel* r, const ldb_colref* keys, int32_t n_keys, DKeys* out);
