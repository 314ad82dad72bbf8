// ldb_keys.h — key-column sets: hashing and equality on the device.
#pragma once
#include "ldb_device.h"

struct KV {
   const DKeys& m;
   const DKeys& p;
   __device__ __forceinline__ KV(const DKeys& k) : m(k), p(k) {}
   __device__ __forceinline__ KV(const DKeys& m_, const DKeys& p_) : m(m_), p(p_) {}
   __device__ __forceinline__ CV col(int j) const { return CV(m.cols[j], p.cols[j]); }
};

// db.hash(keys) of logical row i; NULL parts are skipped (reference LowerToStd.cpp:1118-1132).
// `*any_null` reports whether some key part is NULL (join keys with NULL never match).
__device__ __forceinline__ uint64_t d_hash_keys(KV k, uint64_t i, bool* any_null = nullptr) {
   uint64_t total = 0; // combine(h, 0) == h, so the empty running hash is 0
   bool nul = false;
   const int nk = k.m.n_keys;
   LDB_UNROLL
   for (int j = 0; j < nk; j++) {
      const CV c = k.col(j);
      uint32_t row = d_phys_row(c, i);
      if (!d_valid(c, row)) {
         nul = true;
      } else {
         total = d_hash_part(c, row, total);
      }
   }
   if (any_null) *any_null = nul;
   return total;
}

// all key parts equal between logical row ia of ka and ib of kb.
// nulls_equal: group-by semantics (NULL = NULL, `isa`); else join semantics (createEqFn,
// reference src/compiler/Conversion/RelAlgToSubOp/RelAlgToSubOp.cpp:1035-1066).
__device__ __forceinline__ bool d_keys_equal(KV ka, uint64_t ia, KV kb, uint64_t ib, bool nulls_equal) {
   const int nk = ka.m.n_keys;
   bool eq = true;
   LDB_UNROLL
   for (int j = 0; j < nk; j++) {
      if (eq) {
         const CV ca = ka.col(j), cb = kb.col(j);
         uint32_t ra = d_phys_row(ca, ia), rb = d_phys_row(cb, ib);
         bool va = d_valid(ca, ra), vb = d_valid(cb, rb);
         if (!va || !vb) {
            if (!(nulls_equal && !va && !vb)) eq = false;
         } else if (!d_key_part_equal(ca, ra, cb, rb)) {
            eq = false;
         }
      }
   }
   return eq;
}
