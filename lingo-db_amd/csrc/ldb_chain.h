// ldb_chain.h — building blocks of the single-launch (chained, decoupled look-back) scans: ldb_core.hip's exclusive scans and
// bitmap compaction, the rank-table prefix pass of ldb_join.hip, the occupancy scan of ldb_gbhost.hip.
// A tile of CHAIN_TILE elements per workgroup of 256; tile numbers come from a ticket counter (a tile only ever waits for
// tiles whose workgroups are already running); per-tile status words {epoch, state, value} are never cleared: every call owns
// a fresh epoch (ldb_chain_begin), a word of another epoch reads as "not there yet".
// Status word: value in the low VBITS bits, state (1 = the tile's own sum, 2 = inclusive prefix) above it, epoch on top.
#pragma once
#include "ldb_internal.h"
#define CHAIN_ITEMS 8
#define CHAIN_TILE (256 * CHAIN_ITEMS)
struct ChainCall {
   unsigned long long* status;
   unsigned long long* ticket;
   unsigned long long ticket_base;
   unsigned long long epoch;
};
// reserves n_tiles tickets and an epoch for one launch (wide = 42-bit values / 20-bit epochs instead of 32 / 30)
int32_t ldb_chain_begin(ldb_ctx* ctx, uint64_t n_tiles, bool wide, ChainCall* c);
// after a launch that did not happen: re-bases the ticket counter (or every later chain would wait for tiles that never run)
int32_t ldb_chain_failed(ldb_ctx* ctx);

template <typename TO, int VBITS>
__device__ __forceinline__ unsigned long long d_chain_pack(unsigned long long epoch, unsigned state, TO v) {
   return (epoch << (VBITS + 2)) | ((unsigned long long) state << VBITS) | ((unsigned long long) v & ((1ull << VBITS) - 1));
}
// called by all lanes of ONE wave of the workgroup that owns `tile`: publishes the tile's sum, walks back over the
// predecessors' status words (64 at a time) and returns the exclusive prefix of the tile (valid in every lane)
template <typename TO, int VBITS>
__device__ __forceinline__ TO d_chain_prefix(unsigned long long* __restrict__ status, uint64_t tile, unsigned long long epoch, TO agg, uint32_t lane) {
   TO prefix = 0;
   if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&status[0], d_chain_pack<TO, VBITS>(epoch, 2, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return prefix;
   }
   if (lane == 0) __hip_atomic_store(&status[tile], d_chain_pack<TO, VBITS>(epoch, 1, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   int64_t look = (int64_t) tile - 1;
   for (;;) {
      const int64_t idx = look - (int64_t) lane;
      unsigned long long w;
      for (;;) {
         w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : d_chain_pack<TO, VBITS>(epoch, 2, (TO) 0);
         const bool ready = (w >> (VBITS + 2)) == epoch && ((w >> VBITS) & 3u) != 0;
         if (__ballot(!ready) == 0) break;
         __builtin_amdgcn_s_sleep(1);
      }
      const unsigned long long incl_mask = __ballot(((w >> VBITS) & 3u) == 2u);
      const int first = incl_mask ? __builtin_ctzll(incl_mask) : 64;
      TO contrib = (int) lane <= first ? (TO) (w & ((1ull << VBITS) - 1)) : (TO) 0;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) contrib += (TO) __shfl_xor((long long) contrib, off);
      prefix += contrib;
      if (incl_mask) break;
      look -= 64;
   }
   if (lane == 0) __hip_atomic_store(&status[tile], d_chain_pack<TO, VBITS>(epoch, 2, (TO) (prefix + agg)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   return prefix;
}
// workgroup of 256: exclusive scan of one value per thread; returns the thread's exclusive prefix inside the workgroup and
// the workgroup's total in *agg (s_wave: 4 words of LDS)
template <typename TO>
__device__ __forceinline__ TO d_block_scan256(TO sum, TO* s_wave, TO* agg) {
   const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
   TO incl = sum;
#pragma unroll
   for (int off = 1; off < 64; off <<= 1) {
      const TO t = (TO) __shfl_up((long long) incl, off);
      if ((int) lane >= off) incl += t;
   }
   if (lane == 63) s_wave[wave] = incl;
   __syncthreads();
   TO wave_off = 0;
   for (uint32_t w = 0; w < wave; w++) wave_off += s_wave[w];
   *agg = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
   return wave_off + incl - sum;
}
