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
#include <atomic>
#include <chrono>
#include <cctype>
#include <cerrno>
#include <dlfcn.h>
#include <fcntl.h>
#include <memory>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
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
   bool ok = false;
   std::string why;
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
         r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GetErrorString && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
         if (!r.ok) r.why = "symbols missing";
      } else {
         const char* e = dlerror(); // (a second dlerror() call returns NULL: keep the first answer)
         r.why = e ? e : "dlopen failed";
      }
   }
   return r;
}

#define LDB_NCCL(expr)                                                                                   \
   do {                                                                                                  \
      ncclResult_t r_ = (expr);                                                                          \
      if (r_ != ncclSuccess) {                                                                           \
         ldb_set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(r_), __FILE__, __LINE__);   \
         return LDB_ERR_HIP;                                                                             \
      }                                                                                                  \
   } while (0)

// ---------------------------------------------------------------- transports
// The exchange needs exactly one primitive: a GROUP of point-to-point byte transfers between the ranks
// (every rank lists its sends and receives; transfers of one (sender, receiver) pair match in order).
// Two transports implement it:
//   RCCL  — ncclGroupStart … ncclSend / ncclRecv … ncclGroupEnd on the ctx stream: one rank per GPU, xGMI.
//   SHM   — host-staged through POSIX shared memory: ranks are processes of one node that may SHARE a GPU
//           (RCCL refuses two ranks on one device).  The sender copies its bytes device → a /dev/shm segment
//           per (sender, receiver) pair, a barrier in a shared control segment publishes them, the receiver
//           copies segment → device.  It exists so that every world > 1 path of this file — metadata
//           all-to-all, displacement arithmetic, offset / bitmap rebuild on arrival — runs under test on a
//           one-GPU box, and as the fallback when RCCL cannot be initialised; it is not a fast path.
struct Transport {
   // what went over the links since the last reset (ldb_gpu_comm_stats): transfers to / from OTHER ranks only — a rank's
   // block for itself never leaves the device
   int self = 0;
   int64_t groups = 0, bytes_out = 0, bytes_in = 0;
   std::vector<int64_t> out_peer;
   double host_ms = 0, dev_ms = 0;
   ldb_ctx* ev_ctx = nullptr; // device communicators: every group is bracketed by two events on the ctx stream
   std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
   std::vector<hipEvent_t> ev_free;
   std::chrono::steady_clock::time_point t_open;
   void note(bool is_send, size_t bytes, int peer) {
      if (peer == self) return;
      if (is_send) {
         bytes_out += (int64_t) bytes;
         if ((size_t) peer >= out_peer.size()) out_peer.resize((size_t) peer + 1, 0);
         out_peer[(size_t) peer] += (int64_t) bytes;
      } else {
         bytes_in += (int64_t) bytes;
      }
   }
   hipEvent_t event() {
      hipEvent_t e = nullptr;
      if (!ev_free.empty()) {
         e = ev_free.back();
         ev_free.pop_back();
      } else if (hipEventCreate(&e) != hipSuccess) {
         e = nullptr;
      }
      return e;
   }
   void timed_open() {
      t_open = std::chrono::steady_clock::now();
      if (ev_ctx) {
         // a process that never asks for statistics must not pile up events: pairs the device has finished are folded as we go
         while (ev_pending.size() > 32 && ev_pending.front().second && hipEventQuery(ev_pending.front().second) == hipSuccess) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev_pending.front().first, ev_pending.front().second) == hipSuccess) dev_ms += ms;
            ev_free.push_back(ev_pending.front().first);
            ev_free.push_back(ev_pending.front().second);
            ev_pending.erase(ev_pending.begin());
         }
         hipEvent_t a = event();
         if (a && hipEventRecord(a, ev_ctx->stream) == hipSuccess) ev_pending.push_back({a, nullptr});
         else if (a) ev_free.push_back(a);
      }
   }
   void timed_close() {
      groups++;
      host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_open).count();
      if (ev_ctx && !ev_pending.empty() && !ev_pending.back().second) {
         hipEvent_t b = event();
         if (b && hipEventRecord(b, ev_ctx->stream) == hipSuccess) {
            ev_pending.back().second = b;
         } else {
            if (b) ev_free.push_back(b);
            ev_free.push_back(ev_pending.back().first);
            ev_pending.pop_back();
         }
      }
   }
   void fold_events() { // (waits for the last group to finish on the device)
      for (auto& pr : ev_pending) {
         float ms = 0;
         if (pr.second && hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) dev_ms += ms;
         ev_free.push_back(pr.first);
         if (pr.second) ev_free.push_back(pr.second);
      }
      ev_pending.clear();
   }
   void reset_stats() {
      fold_events();
      groups = bytes_out = bytes_in = 0;
      out_peer.clear();
      host_ms = dev_ms = 0;
   }
   void drop_events() {
      fold_events();
      for (auto e : ev_free) (void) hipEventDestroy(e);
      ev_free.clear();
   }
   virtual ~Transport() {}
   virtual const char* name() const = 0;
   virtual int32_t group_start() = 0;
   virtual int32_t send(const void* d_ptr, size_t bytes, int peer) = 0;
   virtual int32_t recv(void* d_ptr, size_t bytes, int peer) = 0;
   virtual int32_t group_end() = 0; // the transfers are complete or ordered on the ctx stream afterwards
   virtual void group_abort() = 0; // error path: leave no group open
};

struct RcclTransport : Transport {
   ldb_ctx* ctx;
   ncclComm_t comm = nullptr;
   bool open = false;
   explicit RcclTransport(ldb_ctx* c) : ctx(c) {}
   ~RcclTransport() override {
      drop_events();
      if (comm) (void) rccl().CommDestroy(comm);
   }
   const char* name() const override { return "rccl"; }
   int32_t group_start() override {
      LDB_NCCL(rccl().GroupStart());
      open = true;
      return LDB_OK;
   }
   int32_t send(const void* p, size_t bytes, int peer) override {
      LDB_NCCL(rccl().Send(p, bytes, ncclUint8, peer, comm, ctx->stream));
      note(true, bytes, peer);
      return LDB_OK;
   }
   int32_t recv(void* p, size_t bytes, int peer) override {
      LDB_NCCL(rccl().Recv(p, bytes, ncclUint8, peer, comm, ctx->stream));
      note(false, bytes, peer);
      return LDB_OK;
   }
   int32_t group_end() override {
      open = false;
      LDB_NCCL(rccl().GroupEnd());
      return LDB_OK;
   }
   void group_abort() override {
      if (open) (void) rccl().GroupEnd();
      open = false;
   }
};

static const char SHM_MAGIC[8] = {'L', 'D', 'B', 'S', 'H', 'M', '0', '1'};
struct ShmControl { // zero-filled on creation = the valid initial state
   std::atomic<uint32_t> arrived;
   std::atomic<uint32_t> generation;
   std::atomic<uint32_t> failed; // a rank gave up: everybody else stops waiting
};
struct ShmTransport : Transport {
   ldb_ctx* ctx;
   int rank, world;
   std::string token; // from the unique id
   ShmControl* ctl = nullptr;
   uint64_t seq = 0;
   struct Op {
      bool is_send;
      void* ptr;
      size_t bytes;
      int peer;
   };
   std::vector<Op> ops;
   ShmTransport(ldb_ctx* c, int r, int w, std::string t) : ctx(c), rank(r), world(w), token(std::move(t)) {}
   std::string ctl_name() const { return "/ldbcomm_" + token + "_ctl"; }
   std::string seg_name(uint64_t s, int from, int to) const { return "/ldbcomm_" + token + "_" + std::to_string(s) + "_" + std::to_string(from) + "_" + std::to_string(to); }
   int32_t open_control() {
      const int fd = shm_open(ctl_name().c_str(), O_CREAT | O_RDWR, 0600);
      if (fd < 0) LDB_FAIL(LDB_ERR_HIP, "shm transport: shm_open(%s) failed: %s", ctl_name().c_str(), strerror(errno));
      if (ftruncate(fd, sizeof(ShmControl)) != 0) {
         close(fd);
         LDB_FAIL(LDB_ERR_HIP, "shm transport: ftruncate failed: %s", strerror(errno));
      }
      void* p = mmap(nullptr, sizeof(ShmControl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) LDB_FAIL(LDB_ERR_HIP, "shm transport: mmap failed: %s", strerror(errno));
      ctl = (ShmControl*) p;
      return LDB_OK;
   }
   ~ShmTransport() override {
      drop_events();
      if (ctl) {
         (void) barrier(); // nobody unlinks while a peer may still open
         munmap(ctl, sizeof(ShmControl));
         if (rank == 0) shm_unlink(ctl_name().c_str());
      }
   }
   const char* name() const override { return "shm"; }
   // sense-reversing barrier over the ranks; bounded (a lost rank must not hang the others — or a GPU box)
   int32_t barrier() {
      const int64_t limit_ms = ldb_option("comm_timeout_ms", 120000);
      const uint32_t gen = ctl->generation.load(std::memory_order_acquire);
      if (ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t) world) {
         ctl->arrived.store(0, std::memory_order_relaxed);
         ctl->generation.fetch_add(1, std::memory_order_acq_rel);
         return LDB_OK;
      }
      const auto t0 = std::chrono::steady_clock::now();
      for (uint64_t spin = 0;; spin++) {
         if (ctl->generation.load(std::memory_order_acquire) != gen) return LDB_OK;
         if (ctl->failed.load(std::memory_order_acquire)) LDB_FAIL(LDB_ERR_HIP, "shm transport: a peer rank failed");
         if (spin > 1000) usleep(50);
         if ((spin & 1023) == 1023 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double) limit_ms) {
            ctl->failed.store(1, std::memory_order_release);
            LDB_FAIL(LDB_ERR_HIP, "shm transport: barrier timed out after %ld ms (rank %d of %d)", (long) limit_ms, rank, world);
         }
      }
   }
   int32_t group_start() override {
      ops.clear();
      return LDB_OK;
   }
   int32_t send(const void* p, size_t bytes, int peer) override {
      ops.push_back({true, const_cast<void*>(p), bytes, peer});
      note(true, bytes, peer);
      return LDB_OK;
   }
   int32_t recv(void* p, size_t bytes, int peer) override {
      ops.push_back({false, p, bytes, peer});
      note(false, bytes, peer);
      return LDB_OK;
   }
   void group_abort() override {
      ops.clear();
      if (ctl) ctl->failed.store(1, std::memory_order_release);
   }
   struct Mapping { // RAII: one mapped segment
      void* p = MAP_FAILED;
      size_t bytes = 0;
      ~Mapping() {
         if (p != MAP_FAILED) munmap(p, bytes);
      }
   };
   int32_t map_segment(const std::string& name, size_t bytes, bool create, Mapping* m) {
      const int fd = shm_open(name.c_str(), create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDONLY, 0600);
      if (fd < 0) LDB_FAIL(LDB_ERR_HIP, "shm transport: shm_open(%s) failed: %s", name.c_str(), strerror(errno));
      if (create && ftruncate(fd, (off_t) bytes) != 0) {
         close(fd);
         shm_unlink(name.c_str());
         LDB_FAIL(LDB_ERR_HIP, "shm transport: ftruncate(%zu) failed: %s", bytes, strerror(errno));
      }
      if (create) {
         // reserve the pages now: ftruncate alone leaves a sparse tmpfs file, and a /dev/shm smaller than one round's
         // volume (a container's default is 64 MB) would then kill the rank with SIGBUS in the middle of the copy while the
         // peers wait out the barrier timeout.  A refused reservation is an ordinary error instead, posted to the peers.
         int rc;
         do rc = posix_fallocate(fd, 0, (off_t) bytes);
         while (rc == EINTR);
         if (rc != 0) {
            close(fd);
            shm_unlink(name.c_str());
            if (ctl) ctl->failed.store(1, std::memory_order_release);
            LDB_FAIL(rc == ENOSPC ? LDB_ERR_OOM : LDB_ERR_HIP, "shm transport: cannot reserve %zu bytes of shared memory for one exchange round (%s): enlarge /dev/shm or use the RCCL transport", bytes,
                     strerror(rc));
         }
      }
      m->p = mmap(nullptr, bytes, create ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0);
      close(fd);
      if (m->p == MAP_FAILED) LDB_FAIL(LDB_ERR_HIP, "shm transport: mmap(%s, %zu) failed: %s", name.c_str(), bytes, strerror(errno));
      m->bytes = bytes;
      return LDB_OK;
   }
   // ctx == NULL: host mode (ldb_gpu_comm_create_host) — the "device" pointers are host memory; the protocol
   // (segments, barriers, ordering of a pair's transfers) is what the CPU tests exercise with world 2 and 3
   int32_t copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
      if (!ctx) {
         memcpy(dst, src, bytes);
         return LDB_OK;
      }
      // on the ctx stream, never the NULL stream: the ctx stream does not synchronise with it, and a device-to-device
      // hipMemcpy may return before it has run — what the caller enqueues next on the ctx stream (the read-back of the
      // received metadata) then raced with the copy and saw the buffer's previous contents
      LDB_HIP(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
      if (kind != hipMemcpyDeviceToDevice) LDB_HIP(hipStreamSynchronize(ctx->stream)); // host side is a mapped segment: done before it is unmapped / published
      return LDB_OK;
   }
   int32_t group_end_inner() {
      if (ctx) LDB_HIP(hipStreamSynchronize(ctx->stream)); // what is sent was produced on the ctx stream
      std::vector<size_t> out_bytes((size_t) world, 0), in_bytes((size_t) world, 0);
      for (auto& o : ops) (o.is_send ? out_bytes : in_bytes)[(size_t) o.peer] += o.bytes;
      // 1. write: one segment per receiving peer, the sends to it back to back in issue order
      for (int p = 0; p < world; p++) {
         if (p == rank || out_bytes[(size_t) p] == 0) continue;
         Mapping m;
         LDB_TRY(map_segment(seg_name(seq, rank, p), out_bytes[(size_t) p], true, &m));
         size_t at = 0;
         for (auto& o : ops)
            if (o.is_send && o.peer == p && o.bytes) {
               LDB_TRY(copy((char*) m.p + at, o.ptr, o.bytes, hipMemcpyDeviceToHost));
               at += o.bytes;
            }
      }
      LDB_TRY(barrier()); // every segment of this round is written
      // 2. read: transfers of a pair match in order
      for (int p = 0; p < world; p++) {
         if (in_bytes[(size_t) p] == 0) continue;
         if (p == rank) { // to myself: device to device, sends and receives pair up in order
            size_t si = 0;
            for (auto& r : ops) {
               if (r.is_send || r.peer != rank || !r.bytes) continue;
               while (si < ops.size() && !(ops[si].is_send && ops[si].peer == rank && ops[si].bytes)) si++;
               if (si == ops.size() || ops[si].bytes != r.bytes) LDB_FAIL(LDB_ERR_INVALID, "shm transport: self transfer sizes do not pair up");
               LDB_TRY(copy(r.ptr, ops[si].ptr, r.bytes, hipMemcpyDeviceToDevice));
               si++;
            }
            continue;
         }
         Mapping m;
         LDB_TRY(map_segment(seg_name(seq, p, rank), in_bytes[(size_t) p], false, &m));
         size_t at = 0;
         for (auto& o : ops)
            if (!o.is_send && o.peer == p && o.bytes) {
               LDB_TRY(copy(o.ptr, (const char*) m.p + at, o.bytes, hipMemcpyHostToDevice));
               at += o.bytes;
            }
      }
      LDB_TRY(barrier()); // every segment of this round is read: the writers remove theirs
      return LDB_OK;
   }
   int32_t group_end() override {
      const int32_t st = group_end_inner();
      for (int p = 0; p < world; p++)
         if (p != rank) shm_unlink(seg_name(seq, rank, p).c_str()); // (absent when nothing was sent: ignored)
      seq++;
      ops.clear();
      if (st != LDB_OK && ctl) ctl->failed.store(1, std::memory_order_release);
      return st;
   }
};
} // namespace

struct ldb_comm {
   std::unique_ptr<Transport> t;
   int32_t rank = 0, world = 1;
};

extern "C" int32_t ldb_gpu_comm_unique_id(void* id128) {
   if (!id128) LDB_FAIL(LDB_ERR_INVALID, "comm_unique_id: NULL argument");
   memset(id128, 0, 128);
   if (ldb_option("comm_transport", 0) == 1) { // host-staged transport: the id names the shared segments
      memcpy(id128, SHM_MAGIC, 8);
      unsigned char rnd[12] = {0};
      if (FILE* f = fopen("/dev/urandom", "rb")) {
         (void) !fread(rnd, 1, sizeof(rnd), f);
         fclose(f);
      }
      char* tok = (char*) id128 + 8;
      snprintf(tok, 64, "%08x", (unsigned) getpid());
      for (int i = 0; i < 12; i++) snprintf(tok + 8 + 2 * i, 3, "%02x", rnd[i]);
      return LDB_OK;
   }
   if (!rccl().ok) LDB_FAIL(LDB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", rccl().why.c_str());
   static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
   ncclUniqueId id;
   LDB_NCCL(rccl().GetUniqueId(&id));
   memcpy(id128, &id, sizeof(id));
   return LDB_OK;
}
// 1 when librccl can be loaded in this process (a cheap probe every rank runs BEFORE any of them waits for an id)
extern "C" int32_t ldb_gpu_comm_available(void) { return rccl().ok ? 1 : 0; }
extern "C" int32_t ldb_gpu_comm_create(ldb_ctx* ctx, int32_t rank, int32_t world, const void* id128, ldb_comm** out) {
   if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) LDB_FAIL(LDB_ERR_INVALID, "comm_create: bad argument");
   LDB_HIP(hipSetDevice(ctx->device));
   auto c = std::make_unique<ldb_comm>();
   c->rank = rank;
   c->world = world;
   if (!memcmp(id128, SHM_MAGIC, 8)) {
      char tok[65] = {0};
      memcpy(tok, (const char*) id128 + 8, 64);
      for (char* q = tok; *q; q++)
         if (!isxdigit((unsigned char) *q)) LDB_FAIL(LDB_ERR_INVALID, "comm_create: malformed shm transport id");
      auto t = std::make_unique<ShmTransport>(ctx, rank, world, tok);
      LDB_TRY(t->open_control());
      c->t = std::move(t);
      c->t->self = rank;
      c->t->ev_ctx = ctx;
   } else {
      if (!rccl().ok) LDB_FAIL(LDB_ERR_UNSUPPORTED, "librccl.so.1 could not be loaded: %s", rccl().why.c_str());
      auto t = std::make_unique<RcclTransport>(ctx);
      ncclUniqueId id;
      memcpy(&id, id128, sizeof(id));
      LDB_NCCL(rccl().CommInitRank(&t->comm, world, id, rank));
      c->t = std::move(t);
      c->t->self = rank;
      c->t->ev_ctx = ctx;
   }
   *out = c.release();
   return LDB_OK;
}
// a communicator over the host-staged transport WITHOUT a device: transfers move host memory.  For CPU tests
// of the transport's protocol and for exchanging host-side metadata between the ranks' host programs.
extern "C" int32_t ldb_gpu_comm_create_host(int32_t rank, int32_t world, const void* id128, ldb_comm** out) {
   if (!id128 || !out || world < 1 || rank < 0 || rank >= world) LDB_FAIL(LDB_ERR_INVALID, "comm_create_host: bad argument");
   if (memcmp(id128, SHM_MAGIC, 8)) LDB_FAIL(LDB_ERR_INVALID, "comm_create_host: the id is not a host-staged transport id (set option comm_transport = 1 before ldb_gpu_comm_unique_id)");
   char tok[65] = {0};
   memcpy(tok, (const char*) id128 + 8, 64);
   for (char* q = tok; *q; q++)
      if (!isxdigit((unsigned char) *q)) LDB_FAIL(LDB_ERR_INVALID, "comm_create_host: malformed id");
   auto c = std::make_unique<ldb_comm>();
   c->rank = rank;
   c->world = world;
   auto t = std::make_unique<ShmTransport>(nullptr, rank, world, tok);
   LDB_TRY(t->open_control());
   c->t = std::move(t);
   c->t->self = rank;
   *out = c.release();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_comm_destroy(ldb_comm* c) {
   delete c; // the transport's destructor closes the communicator / the shared segments
   return LDB_OK;
}
// traffic and time of the transfer groups since the last reset.  device_ms = between two events on the ctx stream around every
// group (RCCL: the grouped send / recv kernels, waits for the peers included); host_ms = host wall time inside the groups (the
// host-staged transport does all its copying there).  max_peer_bytes_out is what the busiest of this rank's links carried:
// on one MI355X node every peer pair has its own xGMI link (≈ 153 GB/s), so max_peer_bytes_out / device_ms is the link rate.
extern "C" int32_t ldb_gpu_comm_stats(ldb_comm* c, ldb_comm_stats* out, int32_t reset) {
   if (!c || !out) LDB_FAIL(LDB_ERR_INVALID, "comm_stats: NULL argument");
   Transport* t = c->t.get();
   t->fold_events();
   out->groups = t->groups;
   out->bytes_out = t->bytes_out;
   out->bytes_in = t->bytes_in;
   out->max_peer_bytes_out = 0;
   for (int64_t b : t->out_peer) out->max_peer_bytes_out = std::max(out->max_peer_bytes_out, b);
   out->host_ms = t->host_ms;
   out->device_ms = t->dev_ms;
   if (reset) t->reset_stats();
   return LDB_OK;
}
extern "C" int32_t ldb_gpu_comm_rank(const ldb_comm* c) { return c ? c->rank : 0; }
extern "C" int32_t ldb_gpu_comm_world(const ldb_comm* c) { return c ? c->world : 1; }
extern "C" const char* ldb_gpu_comm_transport(const ldb_comm* c) { return c && c->t ? c->t->name() : "none"; }

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

#define PICK_MAX 32
struct PickList {
   int32_t n, pad;
   int64_t idx[PICK_MAX];
};
// out[i] = src[idx[i]] (0 for idx < 0): the string offsets at the run boundaries of an exchange, gathered for one read-back
__global__ void k_pick_i64(const int64_t* __restrict__ src, PickList pl, int64_t* __restrict__ out) {
   const int i = (int) threadIdx.x;
   if (i < pl.n) out[i] = pl.idx[i] >= 0 ? src[pl.idx[i]] : 0;
}

// ---------------------------------------------------------------- the exchange
// Every rank sends the rows [send_off[p], send_off[p] + send_cnt[p]) of `t` to peer p and receives
// recv_cnt[p] rows from it; the result holds the received rows in peer order.  allgather = every peer
// gets all rows; alltoall = ldb_gpu_partition's layout.
namespace {
struct DevBufs { // device temporaries of one exchange: released on every return path
   ldb_ctx* ctx;
   std::vector<void*> ptrs;
   explicit DevBufs(ldb_ctx* c) : ctx(c) {}
   ~DevBufs() {
      for (void* p : ptrs) ldb_dev_free(ctx, p);
   }
   template <typename T>
   int32_t alloc(T** out, size_t bytes) {
      void* p = nullptr;
      LDB_TRY(ldb_dev_alloc(ctx, &p, bytes ? bytes : 8));
      ptrs.push_back(p);
      *out = (T*) p;
      return LDB_OK;
   }
};
struct TableGuard { // the partially built result table
   ldb_ctx* ctx;
   ldb_table* t = nullptr;
   explicit TableGuard(ldb_ctx* c) : ctx(c) {}
   ~TableGuard() {
      if (t) ldb_gpu_table_release(ctx, t);
   }
   ldb_table* release() {
      ldb_table* r = t;
      t = nullptr;
      return r;
   }
};
struct GroupGuard { // never leave a transfer group open on an error return
   Transport* t;
   bool open = false;
   explicit GroupGuard(Transport* tr) : t(tr) {}
   int32_t start() {
      t->timed_open();
      LDB_TRY(t->group_start());
      open = true;
      return LDB_OK;
   }
   int32_t end() {
      open = false;
      const int32_t st = t->group_end();
      t->timed_close();
      return st;
   }
   ~GroupGuard() {
      if (open) t->group_abort();
   }
};
} // namespace

// a result column with the SOURCE column's width (a table may mix 8-byte narrowed decimals with the
// 16-byte decimals group-by and map produce: one `narrow` flag per table cannot describe it)
static int32_t alloc_like(ldb_ctx* ctx, const ldb_table* t, const char* name, int64_t n_rows, const std::vector<int64_t>& data_bytes, ldb_table** out) {
   const int nc = (int) t->cols.size();
   std::vector<ldb_coltype> types((size_t) nc);
   std::vector<const char*> names((size_t) nc);
   for (int k = 0; k < nc; k++) {
      types[(size_t) k] = t->cols[(size_t) k].type;
      names[(size_t) k] = t->cols[(size_t) k].name.c_str();
   }
   ldb_table* res = nullptr;
   LDB_TRY(ldb_gpu_table_alloc(ctx, name, nc, types.data(), names.data(), n_rows, data_bytes.data(), 0, &res));
   for (int k = 0; k < nc; k++) {
      const ldb_column& src = t->cols[(size_t) k];
      ldb_column& dst = res->cols[(size_t) k];
      if (src.type.type == LDB_T_UTF8 || dst.width == src.width) continue;
      ldb_dev_free(ctx, dst.values);
      dst.values = nullptr;
      dst.width = src.width;
      dst.value_bytes = n_rows * (int64_t) src.width;
      const int32_t st = ldb_dev_alloc(ctx, &dst.values, (size_t) (dst.value_bytes ? dst.value_bytes : 8));
      if (st != LDB_OK) {
         ldb_gpu_table_release(ctx, res);
         return st;
      }
   }
   *out = res;
   return LDB_OK;
}

static int32_t exchange(ldb_ctx* ctx, ldb_comm* c, const ldb_table* t, const std::vector<int64_t>& send_off, const std::vector<int64_t>& send_cnt, const char* name,
                        ldb_table** out) {
   const int world = c->world;
   const int nc = (int) t->cols.size();
   for (auto& col : t->cols) // the exchange ships string bytes: dictionary-coded (lazy) columns are written out first
      if (ldb_column_is_lazy(col)) LDB_TRY(ldb_column_strings(ctx, col, t->n_rows));
   Transport* tr = c->t.get();
   DevBufs bufs(ctx);
   std::vector<int> ucols; // utf8 columns
   for (int k = 0; k < nc; k++)
      if (t->cols[(size_t) k].type.type == LDB_T_UTF8) ucols.push_back(k);
   const int nu = (int) ucols.size();
   // ---- metadata per destination peer: [rows, bytes of every utf8 column, per column: validity flag, width, type]
   const int mw = 1 + nu + 3 * nc; // words per (sender, receiver) pair
   std::vector<int64_t> meta((size_t) world * mw, 0);
   // byte ranges of the utf8 columns per peer: the offsets at the run boundaries, picked on the device and read back ONCE for all columns
   // (a read-back like any count: recorded / replayed inside a trace)
   std::vector<std::vector<int64_t>> h_offs((size_t) nu);
   for (int u = 0; u < nu; u++) h_offs[(size_t) u].assign((size_t) world * 2, 0);
   if (nu) {
      int64_t* d_pick;
      const size_t n_pick = (size_t) nu * 2 * (size_t) world;
      LDB_TRY(bufs.alloc(&d_pick, 8 * n_pick));
      for (int u = 0; u < nu; u++) {
         const ldb_column& col = t->cols[(size_t) ucols[(size_t) u]];
         for (int p0 = 0; p0 < world; p0 += PICK_MAX / 2) {
            PickList pl;
            pl.n = 0;
            for (int p = p0; p < world && pl.n + 2 <= PICK_MAX; p++) {
               const bool any = send_cnt[(size_t) p] > 0;
               pl.idx[pl.n++] = any ? send_off[(size_t) p] : -1;
               pl.idx[pl.n++] = any ? send_off[(size_t) p] + send_cnt[(size_t) p] : -1;
            }
            hipLaunchKernelGGL(k_pick_i64, dim3(1), dim3(64), 0, ctx->stream, (const int64_t*) col.offsets, pl, d_pick + ((size_t) u * world + (size_t) p0) * 2);
         }
      }
      LDB_HIP(hipGetLastError());
      std::vector<int64_t> picked(n_pick, 0);
      LDB_TRY(LDB_READBACK(ctx, picked.data(), d_pick, 8 * n_pick));
      for (int u = 0; u < nu; u++)
         for (int k = 0; k < 2 * world; k++) h_offs[(size_t) u][(size_t) k] = picked[(size_t) u * 2 * world + (size_t) k];
   }
   for (int p = 0; p < world; p++) {
      int64_t* m = &meta[(size_t) p * mw];
      m[0] = send_cnt[(size_t) p];
      for (int u = 0; u < nu; u++) m[1 + u] = h_offs[(size_t) u][(size_t) p * 2 + 1] - h_offs[(size_t) u][(size_t) p * 2];
      for (int k = 0; k < nc; k++) {
         m[1 + nu + 3 * k] = t->cols[(size_t) k].validity ? 1 : 0;
         m[1 + nu + 3 * k + 1] = t->cols[(size_t) k].width;
         m[1 + nu + 3 * k + 2] = t->cols[(size_t) k].type.type;
      }
   }
   // all-to-all of the metadata rows: meta[p] goes to peer p
   int64_t *d_send, *d_recv;
   LDB_TRY(bufs.alloc(&d_send, 8 * meta.size()));
   LDB_TRY(bufs.alloc(&d_recv, 8 * meta.size()));
   LDB_TRY(ldb_h2d_small(ctx, d_send, meta.data(), 8 * meta.size())); // (through the pinned ring: `meta` may die before the copy runs)
   if (8 * meta.size() > LDB_RING_BYTES / 8) LDB_HIP(hipStreamSynchronize(ctx->stream)); // (too large for the ring: copied from the vector itself)
   GroupGuard grp(tr);
   LDB_TRY(grp.start());
   for (int p = 0; p < world; p++) {
      LDB_TRY(tr->send(d_send + (size_t) p * mw, 8 * (size_t) mw, p));
      LDB_TRY(tr->recv(d_recv + (size_t) p * mw, 8 * (size_t) mw, p));
   }
   LDB_TRY(grp.end());
   // what the peers send: a read-back like any count — a replaying plan takes the recorded metadata and queues the data transfers at once
   // (every rank replays the same execution: sender and receiver use the same recorded sizes), the real words are checked at the trace's end
   std::vector<int64_t> rmeta((size_t) world * mw);
   LDB_TRY(LDB_READBACK(ctx, rmeta.data(), d_recv, 8 * rmeta.size()));
   // ---- every rank must have passed the same schema (count is implied by mw: a mismatch garbles the words below)
   for (int p = 0; p < world; p++)
      for (int k = 0; k < nc; k++) {
         const int64_t w = rmeta[(size_t) p * mw + 1 + nu + 3 * k + 1], ty = rmeta[(size_t) p * mw + 1 + nu + 3 * k + 2];
         if (w != t->cols[(size_t) k].width || ty != t->cols[(size_t) k].type.type)
            LDB_FAIL(LDB_ERR_INVALID, "exchange: rank %d sends column %d as type %ld / %ld bytes, rank %d has type %d / %d bytes", p, k, (long) ty, (long) w, c->rank,
                     (int) t->cols[(size_t) k].type.type, t->cols[(size_t) k].width);
      }
   // ---- result table
   std::vector<int64_t> recv_cnt((size_t) world), recv_off((size_t) world + 1, 0);
   for (int p = 0; p < world; p++) {
      recv_cnt[(size_t) p] = rmeta[(size_t) p * mw];
      if (recv_cnt[(size_t) p] < 0) LDB_FAIL(LDB_ERR_INVALID, "exchange: negative row count from rank %d", p);
      recv_off[(size_t) p + 1] = recv_off[(size_t) p] + recv_cnt[(size_t) p];
   }
   const int64_t n_all = recv_off[(size_t) world];
   if (n_all >= (int64_t) LDB_NULL_ROW) LDB_FAIL(LDB_ERR_UNSUPPORTED, "exchange: %ld received rows exceed uint32 row ids", (long) n_all);
   std::vector<int64_t> data_bytes((size_t) nc, 0);
   // a column is nullable on arrival when ANY sender has NULLs in it.  Every sender reports its flags to every
   // peer, so all ranks compute the same OR and take part in the validity transfer of the same columns.
   std::vector<bool> any_valid((size_t) nc, false);
   for (int k = 0; k < nc; k++)
      for (int p = 0; p < world; p++) any_valid[(size_t) k] = any_valid[(size_t) k] || rmeta[(size_t) p * mw + 1 + nu + 3 * k] != 0;
   for (int u = 0; u < nu; u++)
      for (int p = 0; p < world; p++) data_bytes[(size_t) ucols[(size_t) u]] += rmeta[(size_t) p * mw + 1 + u];
   TableGuard res(ctx);
   LDB_TRY(alloc_like(ctx, t, name ? name : "exchanged", n_all, data_bytes, &res.t));
   const int64_t n_mine = t->n_rows;
   const int grid_in = ldb_grid_for(ctx, n_mine, 256, 8), grid_out = ldb_grid_for(ctx, n_all, 256, 8);
   // ---- staging for strings (lengths) and validity (one byte per row)
   std::vector<int64_t*> lens_in((size_t) nu, nullptr), lens_out((size_t) nu, nullptr);
   std::vector<uint8_t*> vb_in((size_t) nc, nullptr), vb_out((size_t) nc, nullptr);
   for (int u = 0; u < nu; u++) {
      LDB_TRY(bufs.alloc(&lens_in[(size_t) u], 8 * (size_t) (n_mine + 1)));
      LDB_TRY(bufs.alloc(&lens_out[(size_t) u], 8 * (size_t) (n_all + 1)));
      if (n_mine) hipLaunchKernelGGL(k_offsets_to_lens, dim3(grid_in), dim3(256), 0, ctx->stream, (const int64_t*) t->cols[(size_t) ucols[(size_t) u]].offsets, lens_in[(size_t) u], (uint64_t) n_mine);
   }
   for (int k = 0; k < nc; k++) {
      if (!any_valid[(size_t) k]) continue;
      LDB_TRY(bufs.alloc(&vb_in[(size_t) k], (size_t) (n_mine + 1)));
      LDB_TRY(bufs.alloc(&vb_out[(size_t) k], (size_t) (n_all + 1)));
      if (n_mine) hipLaunchKernelGGL(k_bits_to_bytes, dim3(grid_in), dim3(256), 0, ctx->stream, (const uint8_t*) t->cols[(size_t) k].validity, vb_in[(size_t) k], (uint64_t) n_mine);
   }
   LDB_HIP(hipGetLastError());
   // ---- ONE grouped batch: every column's values (+ string lengths + validity bytes) to / from every peer
   std::vector<int64_t> ubyte_off((size_t) nu, 0); // running byte displacement of the received string data per utf8 column
   LDB_TRY(grp.start());
   for (int p = 0; p < world; p++) {
      const int64_t so = send_off[(size_t) p], sn = send_cnt[(size_t) p], ro = recv_off[(size_t) p], rn = recv_cnt[(size_t) p];
      int u = 0;
      for (int k = 0; k < nc; k++) {
         const ldb_column& col = t->cols[(size_t) k];
         ldb_column& dst = res.t->cols[(size_t) k];
         if (col.type.type == LDB_T_UTF8) {
            const int64_t sb0 = h_offs[(size_t) u][(size_t) p * 2], sbytes = h_offs[(size_t) u][(size_t) p * 2 + 1] - sb0, rbytes = rmeta[(size_t) p * mw + 1 + u];
            if (sn) LDB_TRY(tr->send(lens_in[(size_t) u] + so, 8 * (size_t) sn, p));
            if (rn) LDB_TRY(tr->recv(lens_out[(size_t) u] + ro, 8 * (size_t) rn, p));
            if (sbytes) LDB_TRY(tr->send((const uint8_t*) col.values + sb0, (size_t) sbytes, p));
            if (rbytes) LDB_TRY(tr->recv((uint8_t*) dst.values + ubyte_off[(size_t) u], (size_t) rbytes, p));
            ubyte_off[(size_t) u] += rbytes;
            u++;
         } else {
            const size_t w = (size_t) col.width; // == dst.width (alloc_like) == every sender's width (checked above)
            if (sn) LDB_TRY(tr->send((const uint8_t*) col.values + (size_t) so * w, (size_t) sn * w, p));
            if (rn) LDB_TRY(tr->recv((uint8_t*) dst.values + (size_t) ro * w, (size_t) rn * w, p));
         }
         if (any_valid[(size_t) k]) {
            if (sn) LDB_TRY(tr->send(vb_in[(size_t) k] + so, (size_t) sn, p));
            if (rn) LDB_TRY(tr->recv(vb_out[(size_t) k] + ro, (size_t) rn, p));
         }
      }
   }
   LDB_TRY(grp.end());
   // ---- rebuild offsets / validity bitmaps on arrival
   for (int u = 0; u < nu; u++) {
      ldb_column& dst = res.t->cols[(size_t) ucols[(size_t) u]];
      LDB_TRY(ldb_exclusive_scan_i64(ctx, lens_out[(size_t) u], dst.offsets, n_all, dst.offsets + n_all));
   }
   // NULL counts of all nullable columns: counted on the device, ONE read-back
   std::vector<int> vcols;
   for (int k = 0; k < nc; k++)
      if (any_valid[(size_t) k]) vcols.push_back(k);
   if (!vcols.empty()) {
      unsigned long long* d_nulls;
      LDB_TRY(bufs.alloc(&d_nulls, 8 * vcols.size()));
      LDB_HIP(hipMemsetAsync(d_nulls, 0, 8 * vcols.size(), ctx->stream));
      for (size_t v = 0; v < vcols.size(); v++) {
         ldb_column& dst = res.t->cols[(size_t) vcols[v]];
         LDB_TRY(ldb_dev_alloc(ctx, (void**) &dst.validity, (size_t) ((n_all + 7) / 8 + 1)));
         if (n_all) {
            hipLaunchKernelGGL(k_bytes_to_bits, dim3(grid_out), dim3(256), 0, ctx->stream, (const uint8_t*) vb_out[(size_t) vcols[v]], dst.validity, (uint64_t) n_all);
            hipLaunchKernelGGL(k_count_zero_bytes, dim3(grid_out), dim3(256), 0, ctx->stream, (const uint8_t*) vb_out[(size_t) vcols[v]], (uint64_t) n_all, d_nulls + v);
         }
      }
      std::vector<unsigned long long> nulls(vcols.size(), 0);
      LDB_TRY(LDB_READBACK(ctx, nulls.data(), d_nulls, 8 * vcols.size()));
      for (size_t v = 0; v < vcols.size(); v++) {
         ldb_column& dst = res.t->cols[(size_t) vcols[v]];
         dst.null_count = (int64_t) nulls[v];
         dst.type.nullable = 1;
      }
   }
   LDB_HIP(hipGetLastError());
   *out = res.release();
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
   TableGuard packed(ctx);
   std::vector<int64_t> counts((size_t) c->world, 0);
   LDB_TRY(ldb_gpu_partition(ctx, in, keys, n_keys, c->world, cols, n_cols, &packed.t, counts.data()));
   return ldb_gpu_alltoall(ctx, c, packed.t, counts.data(), name, out);
}

extern "C" int32_t ldb_gpu_comm_agree(ldb_ctx* ctx, ldb_comm* c, int32_t mine, int32_t* all_min) {
   if (!c || !all_min) LDB_FAIL(LDB_ERR_INVALID, "comm_agree: NULL argument");
   const int world = c->world;
   if (world == 1) {
      *all_min = mine;
      return LDB_OK;
   }
   if (world > 63) LDB_FAIL(LDB_ERR_UNSUPPORTED, "comm_agree: more than 63 ranks");
   std::vector<int64_t> h((size_t) world + 1, 0);
   h[0] = mine;
   int64_t *send = h.data(), *recv = h.data() + 1;
   void* dev = nullptr;
   if (ctx) { // a device communicator moves device memory: [mine | one word per peer]
      LDB_TRY(ldb_dev_alloc(ctx, &dev, 8 * ((size_t) world + 1)));
      send = (int64_t*) dev;
      recv = send + 1;
      const int32_t st = ldb_h2d_small(ctx, send, h.data(), 8);
      if (st != LDB_OK) {
         ldb_dev_free(ctx, dev);
         return st;
      }
   }
   int32_t st = LDB_OK;
   {
      GroupGuard grp(c->t.get());
      st = grp.start();
      for (int p = 0; p < world && st == LDB_OK; p++) {
         st = c->t->send(send, 8, p);
         if (st == LDB_OK) st = c->t->recv(recv + p, 8, p);
      }
      if (st == LDB_OK) st = grp.end();
   }
   if (st == LDB_OK && ctx) {
      hipError_t e = hipMemcpyAsync(ctx->h_scratch, recv, 8 * (size_t) std::min(world, 63), hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) {
         ldb_set_error("comm_agree: %s", hipGetErrorString(e));
         st = LDB_ERR_HIP;
      } else {
         memcpy(h.data() + 1, ctx->h_scratch, 8 * (size_t) std::min(world, 63));
      }
   }
   if (ctx) ldb_dev_free(ctx, dev);
   if (st != LDB_OK) return st;
   int64_t m = mine;
   for (int p = 0; p < world; p++) m = std::min(m, h[(size_t) p + 1]);
   *all_min = (int32_t) m;
   return LDB_OK;
}

// raw all-to-all of bytes: send_bytes[p] bytes from `send` (peer runs back to back) to peer p, recv_bytes[p]
// bytes from peer p into `recv` (peer runs back to back) — one group of point-to-point transfers
extern "C" int32_t ldb_gpu_comm_alltoall_bytes(ldb_comm* c, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes) {
   if (!c || !send_bytes || !recv_bytes) LDB_FAIL(LDB_ERR_INVALID, "comm_alltoall_bytes: NULL argument");
   GroupGuard grp(c->t.get());
   LDB_TRY(grp.start());
   size_t so = 0, ro = 0;
   for (int p = 0; p < c->world; p++) {
      if (send_bytes[p] < 0 || recv_bytes[p] < 0) LDB_FAIL(LDB_ERR_INVALID, "comm_alltoall_bytes: negative size");
      if (send_bytes[p]) LDB_TRY(c->t->send((const char*) send + so, (size_t) send_bytes[p], p));
      if (recv_bytes[p]) LDB_TRY(c->t->recv((char*) recv + ro, (size_t) recv_bytes[p], p));
      so += (size_t) send_bytes[p];
      ro += (size_t) recv_bytes[p];
   }
   return grp.end();
}
