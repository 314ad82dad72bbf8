// ldb_comm.hip — the multi-GPU exchange behind the C-ABI: RCCL over xGMI, one rank per GPU.
// the path has exactly two exchange shapes — replicate a small relation on every rank
// (filtered dimension tables, partial aggregates: ldb_gpu_allgather) and re-partition a relation by
// the reference hash of its key (joins / high-cardinality group-bys that are not co-partitioned:
// ldb_gpu_shuffle = ldb_gpu_partition + ldb_gpu_alltoall).  The reference has no counterpart (one
// process, morsel-driven threads); its LingoDB-side caller would be the GPU ExecutionBackend.
//
// MI355X design: on one node every peer pair has its own xGMI link (7 x ~153 GB/s per GPU), so both
// shapes are issued as ONE grouped batch of point-to-point transfers per exchange
// (ncclGroupStart … ncclSend / ncclRecv … ncclGroupEnd) carrying every column of the table — no ring,
// no per-column collective — on the context's stream, receiving straight into the column buffers
// of the result table.  Row counts travel first (one small all-gather + one D2H read: the receive
// buffers have to be sized); no staging tensors, no host round trip per column.
#include "ldb_internal.h"
#include "ldb_device.h"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <memory>
#include <vector>

// librccl is bound at run time, on the first communicator call: a process that already carries an
// RCCL (torch ships its own librccl.so.1) must end up with ONE copy — two copies abort at exit — and a
// single-GPU user of the library does not need it at all.
namespace {
struct Rccl {
   decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
   decltype(&::ncclCommInitRank) CommInitRank = nullptr;
   decltype(&::ncclCommDestroy) CommDestroy = nullptr;
   decltype(&::ncclGetErrorString) GetErrorString = nullptr;
   decltype(&::ncclGroupStart) GroupStart = nullptr;
   decltype(&::ncclGroupEnd) GroupEnd = nullptr;
   decltype(&::ncclSend) Send = nullptr;
   decltype(&::ncclRecv) Recv = nullptr;
   decltype(&::ncclAllGather) AllGather = nullptr;
   bool ok = false;
};
Rccl& rccl() {
   static Rccl r;
   static bool tried = false;
   if (!tried) {
      tried = true;
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); // the copy the process already has
      if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (h) {
         r.GetUniqueId = (decltype(r.GetUniqueId)) dlsym(h, "ncclGetUniqueId");
         r.CommInitRank = (decltype(r.CommInitRank)) dlsym(h, "ncclCommInitRank");
         r.CommDestroy = (decltype(r.CommDestroy)) dlsym(h, "ncclCommDestroy");
         r.GetErrorString = (decltype(r.GetErrorString)) dlsym(h, "ncclGetErrorString");
         r.GroupStart = (decltype(r.GroupStart)) dlsym(h, "ncclGroupStart");
         r.GroupEnd = (decltype(r.GroupEnd)) dlsym(h, "ncclGroupEnd");
         r.Send = (decltype(r.Send)) dlsym(h, "ncclSend");
         r.Recv = (decltype(r.Recv)) dlsym(h, "ncclRecv");
         r.AllGather = (decltype(r.AllGather)) dlsym(h, "ncclAllGather");
         r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GetErrorString && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather;
      }
   }
   return r;
}
} // namespace
#define ncclGetUniqueId rccl().GetUniqueId
#define ncclCommInitRank rccl().CommInitRank
#define ncclCommDestroy rccl().CommDestroy
#define ncclGetErrorString rccl().GetErrorString
#define ncclGroupStart rccl().GroupStart
#define ncclGroupEnd rccl().GroupEnd
#define ncclSend rccl().Send
#define ncclRecv rccl().Recv
#define ncclAllGather rccl().AllGather

struct ldb_comm {
   ncclComm_t comm = nullptr;
   int32_t rank = 0, world = 1;
};

#define LDB_NCCL(expr)                                                                                   \
   do {                                                                                                  \
      ncclResult_t r_ = (expr);                                                                          \
      if (r_ != ncclSuccess) {                                                                           \
         ldb_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__);      \
         return LDB_ERR_HIP;                                                                             \
      }                                                                                                  \
   } while (0)

extern "C" int32_t ldb_gpu_comm_unique_id(void* id128) {
   if (!id128) LDB_FAIL(LDB_ERR_INVALID, "comm_unique_id: NULL argument");
   if (!rccl().ok) LDB_FAIL(LDB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "symbols missing");
   static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
   ncclUniqueId id;
   LDB_NCCL(ncclGetUniqueId(&id));
   memcpy(id128, &id, sizeof(id));
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_comm_create(ldb_ctx* ctx, int32_t rank, int32_t world, const void* id128, ldb_comm** out) {
   if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) LDB_FAIL(LDB_ERR_INVALID, "comm_create: bad argument");
   if (!rccl().ok) LDB_FAIL(LDB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded");
   LDB_HIP(hipSetDevice(ctx->device));
   auto c = std::make_unique<ldb_comm>();
   c->rank = rank;
   c->world = world;
   ncclUniqueId id;
   memcpy(&id, id128, sizeof(id));
   LDB_NCCL(ncclCommInitRank(&c->comm, world, id, rank));
   *out = c.release();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_comm_destroy(ldb_comm* c) {
   if (!c) return LDB_OK;
   if (c->comm) (void) ncclCommDestroy(c->comm);
   delete c;
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_comm_rank(const ldb_comm* c) { return c ? c->rank : 0; }
extern "C" int32_t ldb_gpu_comm_world(const ldb_comm* c) { return c ? c->world : 1; }

// ---------------------------------------------------------------- small kernels
__global__ void k_offsets_to_lens(const int64_t* __restrict__ offs, int64_t* __restrict__ lens, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) lens[i] = offs[i + 1] - offs[i];
}
__global__ void k_bits_to_bytes(const uint8_t* __restrict__ bitmap, uint8_t* __restrict__ bytes, uint64_t n) {
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x)
      bytes[i] = bitmap ? ((bitmap[i >> 3] >> (i & 7)) & 1) : 1;
}
__global__ void k_bytes_to_bits(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bitmap, uint64_t n) {
   const uint64_t nb = (n + 7) / 8;
   for (uint64_t b = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; b < nb; b += (uint64_t) gridDim.x * blockDim.x) {
      uint8_t m = 0;
      for (int k = 0; k < 8; k++)
         if (b * 8 + k < n && bytes[b * 8 + k]) m |= (uint8_t) (1u << k);
      bitmap[b] = m;
   }
}
__global__ void k_count_zero_bytes(const uint8_t* __restrict__ bytes, uint64_t n, unsigned long long* __restrict__ out) {
   unsigned long long c = 0;
   for (uint64_t i = blockIdx.x * (uint64_t) blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) c += bytes[i] ? 0 : 1;
   for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
   if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ---------------------------------------------------------------- validity as bytes (exchanges done outside the library)
// The torch.distributed test double of the exchange (tpch_dist.replicate) moves flat byte buffers: a
// column's validity travels as one byte per row and is re-attached on arrival.
extern "C" int32_t ldb_gpu_table_validity_bytes(ldb_ctx* ctx, const ldb_table* t, int32_t col, void* d_bytes) {
   if (!ctx || !t || !d_bytes || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "table_validity_bytes: bad argument");
   const int64_t n = t->n_rows;
   if (n) hipLaunchKernelGGL(k_bits_to_bytes, dim3(ldb_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) t->cols[(size_t) col].validity, (uint8_t*) d_bytes, (uint64_t) n);
   LDB_HIP(hipGetLastError());
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_table_set_validity_bytes(ldb_ctx* ctx, ldb_table* t, int32_t col, const void* d_bytes) {
   if (!ctx || !t || !d_bytes || col < 0 || (size_t) col >= t->cols.size()) LDB_FAIL(LDB_ERR_INVALID, "table_set_validity_bytes: bad argument");
   ldb_column& c = t->cols[(size_t) col];
   if (!c.owned) LDB_FAIL(LDB_ERR_INVALID, "table_set_validity_bytes: the table does not own its buffers");
   const int64_t n = t->n_rows;
   unsigned long long* d_nulls = (unsigned long long*) ctx->d_scratch;
   LDB_HIP(hipMemsetAsync(d_nulls, 0, 8, ctx->stream));
   if (n) hipLaunchKernelGGL(k_count_zero_bytes, dim3(ldb_grid_for(ctx, n, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) d_bytes, (uint64_t) n, d_nulls);
   uint64_t nulls = 0;
   LDB_TRY(ldb_read_u64(ctx, d_nulls, &nulls));
   ldb_dev_free(ctx, c.validity);
   c.validity = nullptr;
   c.null_count = (int64_t) nulls;
   if (nulls) {
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &c.validity, (size_t) ((n + 7) / 8 + 1)));
      hipLaunchKernelGGL(k_bytes_to_bits, dim3(ldb_grid_for(ctx, (n + 7) / 8, 256, 8)), dim3(256), 0, ctx->stream, (const uint8_t*) d_bytes, c.validity, (uint64_t) n);
      LDB_HIP(hipGetLastError());
   }
   return LDB_OK;
}

// ---------------------------------------------------------------- the exchange
// Every rank sends the rows [send_off[p], send_off[p] + send_cnt[p]) of `t` to peer p and receives
// recv_cnt[p] rows from it; the result holds the received rows in peer order.  allgather = every peer
// gets all rows; alltoall = ldb_gpu_partition's layout.
static int32_t exchange(ldb_ctx* ctx, ldb_comm* c, const ldb_table* t, const std::vector<int64_t>& send_off, const std::vector<int64_t>& send_cnt, const char* name,
                        ldb_table** out) {
   const int world = c->world;
   const int nc = (int) t->cols.size();
   std::vector<int> ucols; // utf8 columns
   for (int k = 0; k < nc; k++)
      if (t->cols[(size_t) k].type.type == LDB_T_UTF8) ucols.push_back(k);
   const int nu = (int) ucols.size();
   // ---- metadata: per destination peer [rows, bytes of every utf8 column, validity flag of every column]
   const int mw = 1 + nu + nc; // words per (sender, receiver) pair
   std::vector<int64_t> meta((size_t) world * mw, 0);
   std::vector<std::vector<int64_t>> h_offs((size_t) nu);
   for (int u = 0; u < nu; u++) { // byte ranges of the utf8 columns per peer: the offsets at the run boundaries
      const ldb_column& col = t->cols[(size_t) ucols[(size_t) u]];
      h_offs[(size_t) u].resize((size_t) world * 2);
      for (int p = 0; p < world; p++) {
         int64_t b[2] = {0, 0};
         if (send_cnt[(size_t) p] > 0) {
            LDB_HIP(hipMemcpyAsync(&b[0], col.offsets + send_off[(size_t) p], 8, hipMemcpyDeviceToHost, ctx->stream));
            LDB_HIP(hipMemcpyAsync(&b[1], col.offsets + send_off[(size_t) p] + send_cnt[(size_t) p], 8, hipMemcpyDeviceToHost, ctx->stream));
            LDB_HIP(hipStreamSynchronize(ctx->stream));
         }
         h_offs[(size_t) u][(size_t) p * 2] = b[0];
         h_offs[(size_t) u][(size_t) p * 2 + 1] = b[1];
      }
   }
   for (int p = 0; p < world; p++) {
      int64_t* m = &meta[(size_t) p * mw];
      m[0] = send_cnt[(size_t) p];
      for (int u = 0; u < nu; u++) m[1 + u] = h_offs[(size_t) u][(size_t) p * 2 + 1] - h_offs[(size_t) u][(size_t) p * 2];
      for (int k = 0; k < nc; k++) m[1 + nu + k] = t->cols[(size_t) k].validity ? 1 : 0;
   }
   // all-to-all of the metadata rows: meta[p] goes to peer p
   int64_t *d_send, *d_recv;
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &d_send, 8 * meta.size()));
   LDB_TRY(ldb_dev_alloc(ctx, (void**) &d_recv, 8 * meta.size()));
   LDB_HIP(hipMemcpyAsync(d_send, meta.data(), 8 * meta.size(), hipMemcpyHostToDevice, ctx->stream));
   LDB_HIP(hipStreamSynchronize(ctx->stream)); // (meta is a host vector)
   LDB_NCCL(ncclGroupStart());
   for (int p = 0; p < world; p++) {
      LDB_NCCL(ncclSend(d_send + (size_t) p * mw, (size_t) mw, ncclInt64, p, c->comm, ctx->stream));
      LDB_NCCL(ncclRecv(d_recv + (size_t) p * mw, (size_t) mw, ncclInt64, p, c->comm, ctx->stream));
   }
   LDB_NCCL(ncclGroupEnd());
   std::vector<int64_t> rmeta((size_t) world * mw);
   LDB_HIP(hipMemcpyAsync(rmeta.data(), d_recv, 8 * rmeta.size(), hipMemcpyDeviceToHost, ctx->stream));
   LDB_HIP(hipStreamSynchronize(ctx->stream));
   ldb_dev_free(ctx, d_send);
   ldb_dev_free(ctx, d_recv);
   // ---- result table
   std::vector<int64_t> recv_cnt((size_t) world), recv_off((size_t) world + 1, 0);
   for (int p = 0; p < world; p++) {
      recv_cnt[(size_t) p] = rmeta[(size_t) p * mw];
      recv_off[(size_t) p + 1] = recv_off[(size_t) p] + recv_cnt[(size_t) p];
   }
   const int64_t n_all = recv_off[(size_t) world];
   if (n_all >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "exchange: %ld received rows exceed uint32 row ids", (long) n_all);
   std::vector<ldb_coltype> types((size_t) nc);
   std::vector<const char*> names((size_t) nc);
   std::vector<int64_t> data_bytes((size_t) nc, 0);
   std::vector<bool> any_valid((size_t) nc, false);
   bool narrow = false;
   for (int k = 0; k < nc; k++) {
      types[(size_t) k] = t->cols[(size_t) k].type;
      names[(size_t) k] = t->cols[(size_t) k].name.c_str();
      if (types[(size_t) k].type == LDB_T_DECIMAL128 && t->cols[(size_t) k].width == 8) narrow = true;
      for (int p = 0; p < world; p++) any_valid[(size_t) k] = any_valid[(size_t) k] || rmeta[(size_t) p * mw + 1 + nu + k] != 0;
      // a column is nullable on arrival when ANY sender has NULLs in it: every rank must take part in the
      // validity exchange of such a column, so the senders' flags are agreed on by one more tiny exchange
   }
   for (int u = 0; u < nu; u++)
      for (int p = 0; p < world; p++) data_bytes[(size_t) ucols[(size_t) u]] += rmeta[(size_t) p * mw + 1 + u];
   {  // agree on the validity flags: OR over all ranks (a sender without NULLs still sends all-ones bytes)
      std::vector<int64_t> mine((size_t) nc), all((size_t) nc * (size_t) world);
      for (int k = 0; k < nc; k++) mine[(size_t) k] = t->cols[(size_t) k].validity ? 1 : 0;
      int64_t *dm, *da;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &dm, 8 * (size_t) nc));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &da, 8 * (size_t) nc * (size_t) world));
      LDB_HIP(hipMemcpyAsync(dm, mine.data(), 8 * (size_t) nc, hipMemcpyHostToDevice, ctx->stream));
      LDB_HIP(hipStreamSynchronize(ctx->stream));
      LDB_NCCL(ncclAllGather(dm, da, (size_t) nc, ncclInt64, c->comm, ctx->stream));
      LDB_HIP(hipMemcpyAsync(all.data(), da, 8 * all.size(), hipMemcpyDeviceToHost, ctx->stream));
      LDB_HIP(hipStreamSynchronize(ctx->stream));
      ldb_dev_free(ctx, dm);
      ldb_dev_free(ctx, da);
      for (int k = 0; k < nc; k++) {
         any_valid[(size_t) k] = false;
         for (int p = 0; p < world; p++) any_valid[(size_t) k] = any_valid[(size_t) k] || all[(size_t) p * nc + k] != 0;
      }
   }
   ldb_table* res;
   LDB_TRY(ldb_gpu_table_alloc(ctx, name ? name : "exchanged", nc, types.data(), names.data(), n_all, data_bytes.data(), narrow ? 1 : 0, &res));
   const int64_t n_mine = t->n_rows;
   const int grid_in = ldb_grid_for(ctx, n_mine, 256, 8), grid_out = ldb_grid_for(ctx, n_all, 256, 8);
   // ---- staging for strings (lengths) and validity (one byte per row)
   std::vector<int64_t*> lens_in((size_t) nu, nullptr), lens_out((size_t) nu, nullptr);
   std::vector<uint8_t*> vb_in((size_t) nc, nullptr), vb_out((size_t) nc, nullptr);
   for (int u = 0; u < nu; u++) {
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens_in[(size_t) u], 8 * (size_t) (n_mine + 1)));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &lens_out[(size_t) u], 8 * (size_t) (n_all + 1)));
      if (n_mine) hipLaunchKernelGGL(k_offsets_to_lens, dim3(grid_in), dim3(256), 0, ctx->stream, (const int64_t*) t->cols[(size_t) ucols[(size_t) u]].offsets, lens_in[(size_t) u], (uint64_t) n_mine);
   }
   for (int k = 0; k < nc; k++) {
      if (!any_valid[(size_t) k]) continue;
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb_in[(size_t) k], (size_t) (n_mine + 1)));
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &vb_out[(size_t) k], (size_t) (n_all + 1)));
      if (n_mine) hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid_in), dim3(256), 0, ctx->stream, (const uint8_t*) t->cols[(size_t) k].validity, vb_in[(size_t) k], (uint64_t) n_mine);
   }
   LDB_HIP(hipGetLastError());
   // ---- ONE grouped batch: every column's values (+ string lengths + validity bytes) to / from every peer
   std::vector<int64_t> ubyte_off((size_t) nu, 0); // running byte displacement of the received string data per utf8 column
   LDB_NCCL(ncclGroupStart());
   for (int p = 0; p < world; p++) {
      const int64_t so = send_off[(size_t) p], sn = send_cnt[(size_t) p], ro = recv_off[(size_t) p], rn = recv_cnt[(size_t) p];
      int u = 0;
      for (int k = 0; k < nc; k++) {
         const ldb_column& col = t->cols[(size_t) k];
         ldb_column& dst = res->cols[(size_t) k];
         if (col.type.type == LDB_T_UTF8) {
            const int64_t sb0 = h_offs[(size_t) u][(size_t) p * 2], sbytes = h_offs[(size_t) u][(size_t) p * 2 + 1] - sb0, rbytes = rmeta[(size_t) p * mw + 1 + u];
            if (sn) LDB_NCCL(ncclSend(lens_in[(size_t) u] + so, (size_t) sn, ncclInt64, p, c->comm, ctx->stream));
            if (rn) LDB_NCCL(ncclRecv(lens_out[(size_t) u] + ro, (size_t) rn, ncclInt64, p, c->comm, ctx->stream));
            if (sbytes) LDB_NCCL(ncclSend((const uint8_t*) col.values + sb0, (size_t) sbytes, ncclUint8, p, c->comm, ctx->stream));
            if (rbytes) LDB_NCCL(ncclRecv((uint8_t*) dst.values + ubyte_off[(size_t) u], (size_t) rbytes, ncclUint8, p, c->comm, ctx->stream));
            ubyte_off[(size_t) u] += rbytes;
            u++;
         } else {
            const size_t w = (size_t) col.width;
            if (sn) LDB_NCCL(ncclSend((const uint8_t*) col.values + (size_t) so * w, (size_t) sn * w, ncclUint8, p, c->comm, ctx->stream));
            if (rn) LDB_NCCL(ncclRecv((uint8_t*) dst.values + (size_t) ro * w, (size_t) rn * w, ncclUint8, p, c->comm, ctx->stream));
         }
         if (any_valid[(size_t) k]) {
            if (sn) LDB_NCCL(ncclSend(vb_in[(size_t) k] + so, (size_t) sn, ncclUint8, p, c->comm, ctx->stream));
            if (rn) LDB_NCCL(ncclRecv(vb_out[(size_t) k] + ro, (size_t) rn, ncclUint8, p, c->comm, ctx->stream));
         }
      }
   }
   LDB_NCCL(ncclGroupEnd());
   // ---- rebuild offsets / validity bitmaps on arrival
   for (int u = 0; u < nu; u++) {
      ldb_column& dst = res->cols[(size_t) ucols[(size_t) u]];
      LDB_TRY(ldb_exclusive_scan_i64(ctx, lens_out[(size_t) u], dst.offsets, n_all, dst.offsets + n_all));
      ldb_dev_free(ctx, lens_in[(size_t) u]);
      ldb_dev_free(ctx, lens_out[(size_t) u]);
   }
   bool counted = false;
   unsigned long long* d_nulls = (unsigned long long*) (ctx->d_scratch + 48);
   for (int k = 0; k < nc; k++) {
      if (!any_valid[(size_t) k]) continue;
      ldb_column& dst = res->cols[(size_t) k];
      LDB_TRY(ldb_dev_alloc(ctx, (void**) &dst.validity, (size_t) ((n_all + 7) / 8 + 1)));
      if (n_all) hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid_out), dim3(256), 0, ctx->stream, (const uint8_t*) vb_out[(size_t) k], dst.validity, (uint64_t) n_all);
      LDB_HIP(hipMemsetAsync(d_nulls, 0, 8, ctx->stream));
      if (n_all) hipLaunchKernelGGL(k_count_zero_bytes, dim3(grid_out), dim3(256), 0, ctx->stream, (const uint8_t*) vb_out[(size_t) k], (uint64_t) n_all, d_nulls);
      uint64_t nulls = 0;
      LDB_TRY(ldb_read_u64(ctx, d_nulls, &nulls));
      dst.null_count = (int64_t) nulls;
      dst.type.nullable = 1;
      counted = true;
      ldb_dev_free(ctx, vb_in[(size_t) k]);
      ldb_dev_free(ctx, vb_out[(size_t) k]);
   }
   (void) counted;
   LDB_HIP(hipGetLastError());
   *out = res;
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_allgather(ldb_ctx* ctx, ldb_comm* c, const ldb_table* t, const char* name, ldb_table** out) {
   if (!ctx || !c || !t || !out) LDB_FAIL(LDB_ERR_INVALID, "allgather: NULL argument");
   std::vector<int64_t> off((size_t) c->world, 0), cnt((size_t) c->world, t->n_rows);
   return exchange(ctx, c, t, off, cnt, name, out);
}
extern "C" int32_t ldb_gpu_alltoall(ldb_ctx* ctx, ldb_comm* c, const ldb_table* t, const int64_t* send_counts, const char* name, ldb_table** out) {
   if (!ctx || !c || !t || !send_counts || !out) LDB_FAIL(LDB_ERR_INVALID, "alltoall: NULL argument");
   std::vector<int64_t> off((size_t) c->world, 0), cnt((size_t) c->world, 0);
   int64_t at = 0;
   for (int p = 0; p < c->world; p++) {
      if (send_counts[p] < 0) LDB_FAIL(LDB_ERR_INVALID, "alltoall: negative count");
      off[(size_t) p] = at;
      cnt[(size_t) p] = send_counts[p];
      at += send_counts[p];
   }
   if (at != t->n_rows) LDB_FAIL(LDB_ERR_INVALID, "alltoall: the counts add up to %ld rows, the table has %ld", (long) at, (long) t->n_rows);
   return exchange(ctx, c, t, off, cnt, name, out);
}
extern "C" int32_t ldb_gpu_shuffle(ldb_ctx* ctx, ldb_comm* c, ldb_rel* in, const ldb_colref* keys, int32_t n_keys, const ldb_colref* cols, int32_t n_cols, const char* name,
                                   ldb_table** out) {
   if (!ctx || !c || !in || !out) LDB_FAIL(LDB_ERR_INVALID, "shuffle: NULL argument");
   ldb_table* packed = nullptr;
   std::vector<int64_t> counts((size_t) c->world, 0);
   LDB_TRY(ldb_gpu_partition(ctx, in, keys, n_keys, c->world, cols, n_cols, &packed, counts.data()));
   const int32_t st = ldb_gpu_alltoall(ctx, c, packed, counts.data(), name, out);
   ldb_gpu_table_release(ctx, packed);
   return st;
}
