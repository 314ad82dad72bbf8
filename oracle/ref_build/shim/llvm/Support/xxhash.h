// Shim for <llvm/Support/xxhash.h> (LLVM 20.1 is not installed): llvm::xxHash64(ArrayRef<uint8_t>)
// is the standard XXH64 with seed 0; forwarded to Arrow's vendored xxhash (shipped with pyarrow).
// Used ONLY to compile the reference's src/runtime/Hash.cpp into oracle/_ref.
#pragma once
#include <cstddef>
#include <cstdint>
#define XXH_INLINE_ALL
#include "arrow/vendored/xxhash.h"
namespace llvm {
template <typename T>
class ArrayRef {
   const T* p;
   size_t n;

   public:
   ArrayRef(const T* p, size_t n) : p(p), n(n) {}
   const T* data() const { return p; }
   size_t size() const { return n; }
};
inline uint64_t xxHash64(ArrayRef<uint8_t> d) { return XXH64(d.data(), d.size(), 0); }
} // namespace llvm
