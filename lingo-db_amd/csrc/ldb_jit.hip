// ldb_jit.hip — run-time kernel specialisation with hiprtc (see ldb_jit.h).
#include "ldb_jit.h"
#include <hip/hiprtc.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <ctime>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

// the kernel headers, embedded at build time (Makefile → ldb_jit_sources.inc)
struct EmbeddedHeader {
   const char* name;
   const char* text;
};
static const EmbeddedHeader g_headers[] = {
#include "ldb_jit_sources.inc"
};
static const int g_n_headers = (int) (sizeof(g_headers) / sizeof(g_headers[0]));

// One entry per (device, key).  A specialisation is COMPILED on a worker thread (hiprtc takes 0.3 – 3 s per kernel; the reference answers the same
// problem with a baseline backend that emits code in milliseconds and an optimising one behind it, include/lingodb/execution/Execution.h:103-104): the
// operator that asked launches its generic ahead-of-time kernel meanwhile and picks the specialised one up on a later call.  Code objects are kept
// on disk under a content hash, so a second process start compiles nothing.
struct JitModule {
   std::string key; // device | arch | header | struct | kernel source | metadata bytes
   enum State { PENDING, CODE_READY, LOADED, FAILED };
   State state = PENDING;
   int device = 0;
   hipModule_t module = nullptr;
   std::unordered_map<std::string, hipFunction_t> fns;
   std::vector<char> code;
   std::string error; // FAILED: compilation failed once, do not retry
   bool from_disk = false;
};
struct JitJob {
   JitModule* mod;
   std::string src, arch, disk_path, dump_name;
};
static std::mutex g_mu;
static std::condition_variable g_cv; // an entry left PENDING / the queue changed
static std::unordered_map<uint64_t, std::vector<std::unique_ptr<JitModule>>> g_cache;
static std::deque<JitJob> g_queue;
static int g_workers = 0, g_running = 0;
static bool g_stop = false;
static int64_t g_compiled = 0, g_hits = 0, g_disk_hits = 0, g_disk_writes = 0, g_failed = 0, g_async_misses = 0;
static double g_compile_ms = 0;

// (round 6: from 256 K rows on — 4 M until then.  With compilation off the critical path and the code objects on disk the threshold only decides
// how many shapes the compiler sees: 94 instead of 80 over the 22 TPC-H plans, + 35 s of background compilation on a cold start, for Q11 1.41 → 1.14,
// Q17 1.92 → 1.70, Q20 3.62 → 3.29, Q3 4.49 → 4.22 ms — geomean 3.70 → 3.57 ms; `tools/r06_run36.sh`.  The test suite pins 4 M: tests/conftest.py.)
bool ldb_jit_wanted(int64_t n_rows) { return ldb_option("jit", 1) != 0 && n_rows >= ldb_option("jit_min_rows", 262144); }

// 64-bit content hash, eight bytes per step (a lookup hashes a whole descriptor — up to 16 KB — on every operator call; one
// byte per step was 20 µs of host time per group-by)
static uint64_t hash_bytes(const unsigned char* p, size_t n, uint64_t h = 1469598103934665603ull) {
   size_t i = 0;
   for (; i + 8 <= n; i += 8) {
      uint64_t w;
      memcpy(&w, p + i, 8);
      h = (h ^ w) * 0xFF51AFD7ED558CCDull;
      h ^= h >> 32;
   }
   for (; i < n; i++) h = (h ^ p[i]) * 1099511628211ull;
   return h;
}

// addresses become presence flags, per-launch sizes are cleared: what remains is the metadata
void ldb_jit_strip_col(DCol& c) {
   c.values = 0;
   c.offsets = c.offsets ? 1 : 0;
   c.validity = c.validity ? 1 : 0;
   c.rowids = c.rowids ? 1 : 0;
}
void ldb_jit_strip_pred(DPred& p) {
   ldb_jit_strip_col(p.col);
   ldb_jit_strip_col(p.rhs);
   p.zmin = p.zmin ? 1 : 0;
   p.zmax = p.zmax ? 1 : 0;
}
void ldb_jit_strip_keys(DKeys& k) {
   for (int j = 0; j < LDB_MAX_KEYS; j++) ldb_jit_strip_col(k.cols[j]);
}

static std::string build_source(const char* header, const char* struct_name, const char* kernels, const unsigned char* meta, size_t n) {
   std::string s;
   s.reserve(n * 5 + 4096);
   s += "#define LDB_JIT_SPECIALIZED 1\n#include \"";
   s += header;
   s += "\"\nstruct LdbMetaBytes { unsigned char b[" + std::to_string(n) + "]; };\n";
   s += "static constexpr LdbMetaBytes LDB_META_BYTES = {{";
   char buf[8];
   for (size_t i = 0; i < n; i++) {
      snprintf(buf, sizeof(buf), "%u,", (unsigned) meta[i]);
      s += buf;
      if ((i & 63) == 63) s += "\n";
   }
   s += "}};\n__device__ static constexpr ";
   s += struct_name;
   s += " LDB_META = __builtin_bit_cast(";
   s += struct_name;
   s += ", LDB_META_BYTES);\n";
   s += kernels;
   return s;
}

// compile only (no device needed); code object into *code
static bool compile(const std::string& src, std::vector<char>* code, std::string* err, const char* arch = "gfx950") {
   std::vector<const char*> names, texts;
   for (int i = 0; i < g_n_headers; i++) {
      names.push_back(g_headers[i].name);
      texts.push_back(g_headers[i].text);
   }
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src.c_str(), "ldb_spec.hip", g_n_headers, texts.data(), names.data()) != HIPRTC_SUCCESS) {
      *err = "hiprtcCreateProgram failed";
      return false;
   }
   std::vector<std::string> extra; // tuning experiments: LDB_JIT_DEFINES="-DGB_ROWS=8 -DGB_PRED_BATCH=0"
   if (const char* defs = getenv("LDB_JIT_DEFINES")) {
      std::string s(defs), tok;
      for (size_t i = 0; i <= s.size(); i++) {
         if (i == s.size() || s[i] == ' ') {
            if (!tok.empty()) extra.push_back(tok);
            tok.clear();
         } else {
            tok += s[i];
         }
      }
   }
   // The descriptor loops (LDB_UNROLL) must unroll completely or nothing folds: before unrolling
   // their bodies hold the whole generic interpreter, which exceeds the default pragma-unroll
   // size limit and silently leaves a generic loop reading the constexpr descriptor from memory.
   const std::string arch_opt = std::string("--offload-arch=") + arch;
   std::vector<const char*> opts = {arch_opt.c_str(), "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-pragma-unroll-threshold=4000000"};
   for (auto& e : extra) opts.push_back(e.c_str());
   hiprtcResult r = hiprtcCompileProgram(prog, (int) opts.size(), opts.data());
   if (r != HIPRTC_SUCCESS) {
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, '\0');
      if (ls) hiprtcGetProgramLog(prog, log.data());
      *err = std::string("hiprtc: ") + hiprtcGetErrorString(r) + ": " + log.substr(0, 3000);
      hiprtcDestroyProgram(&prog);
      return false;
   }
   size_t cs = 0;
   hiprtcGetCodeSize(prog, &cs);
   code->resize(cs);
   hiprtcGetCode(prog, code->data());
   hiprtcDestroyProgram(&prog);
   if (const char* dir = getenv("LDB_JIT_DUMP_ALL")) { // every code object, numbered (offline ISA / register-usage inspection)
      static std::atomic<int> seq{0}; // (compilations run on several worker threads)
      char path[512];
      snprintf(path, sizeof(path), "%s/ldb_spec_%03d.co", dir, seq.fetch_add(1));
      if (FILE* f = fopen(path, "wb")) {
         fwrite(code->data(), 1, code->size(), f);
         fclose(f);
      }
   }
   return cs > 0;
}

bool ldb_jit_compile_only(const char* header, const char* struct_name, const char* kernels_src, const void* meta, size_t meta_bytes, std::string* log) {
   std::vector<char> code;
   const bool ok = compile(build_source(header, struct_name, kernels_src, (const unsigned char*) meta, meta_bytes), &code, log);
   if (ok)
      if (const char* dir = getenv("LDB_JIT_DUMP_DIR")) { // offline ISA inspection of the check shapes (no GPU needed)
         static int seq = 0;
         char path[512];
         snprintf(path, sizeof(path), "%s/check_%s_%d.co", dir, struct_name, seq++);
         if (FILE* f = fopen(path, "wb")) {
            fwrite(code.data(), 1, code.size(), f);
            fclose(f);
         }
      }
   return ok;
}

// gcnArchName of a device ("gfx950:sramecc+:xnack-"), cached
static std::string device_arch(int device) {
   static std::mutex mu;
   static std::unordered_map<int, std::string> archs;
   std::lock_guard<std::mutex> lock(mu);
   auto it = archs.find(device);
   if (it != archs.end()) return it->second;
   hipDeviceProp_t prop;
   std::string a = "gfx950";
   if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.gcnArchName[0]) a = prop.gcnArchName;
   if (const char* o = getenv("LDB_JIT_ARCH")) a = o; // experiments: e.g. the processor name without target features
   archs[device] = a;
   return a;
}

// ---------------------------------------------------------------- on-disk cache of code objects
// <dir>/<arch>/<hash of (arch, header name, struct, kernel source, metadata, the embedded headers' TEXT, extra defines)>.co ; `dir` = $LDB_JIT_CACHE_DIR,
// else $HOME/.cache/ldb_jit, else /tmp/ldb_jit_<uid>; option jit_disk_cache = 0 (LDB_JIT_DISK_CACHE=0) turns it off.  Files are written to a
// temporary name and renamed, so a reader never sees half a file; a file that does not load is removed and compiled again.
static uint64_t headers_hash() {
   static const uint64_t h = [] {
      uint64_t x = 0x9E3779B97F4A7C15ull;
      for (int i = 0; i < g_n_headers; i++) {
         x = hash_bytes((const unsigned char*) g_headers[i].name, strlen(g_headers[i].name), x);
         x = hash_bytes((const unsigned char*) g_headers[i].text, strlen(g_headers[i].text), x);
      }
      if (const char* defs = getenv("LDB_JIT_DEFINES")) x = hash_bytes((const unsigned char*) defs, strlen(defs), x);
      return x;
   }();
   return h;
}
static std::string cache_dir() {
   if (ldb_option("jit_disk_cache", 1) == 0) return "";
   if (const char* d = getenv("LDB_JIT_CACHE_DIR")) return *d ? std::string(d) : std::string();
   if (const char* home = getenv("HOME"))
      if (*home) return std::string(home) + "/.cache/ldb_jit";
   return "/tmp/ldb_jit_" + std::to_string((unsigned) getuid());
}
static bool make_dirs(const std::string& path) {
   for (size_t i = 1; i <= path.size(); i++)
      if (i == path.size() || path[i] == '/') {
         const std::string sub = path.substr(0, i);
         if (mkdir(sub.c_str(), 0755) != 0 && errno != EEXIST) return false;
      }
   return true;
}
static std::string disk_path_for(const std::string& arch, const std::string& key_nodev) {
   const std::string dir = cache_dir();
   if (dir.empty()) return "";
   std::string a = arch;
   for (char& c : a)
      if (!(isalnum((unsigned char) c) || c == '+' || c == '-')) c = '_';
   const uint64_t h1 = hash_bytes((const unsigned char*) key_nodev.data(), key_nodev.size(), headers_hash());
   const uint64_t h2 = hash_bytes((const unsigned char*) key_nodev.data(), key_nodev.size(), headers_hash() ^ 0xD6E8FEB86659FD93ull);
   char name[64];
   snprintf(name, sizeof(name), "%016llx%016llx.co", (unsigned long long) h1, (unsigned long long) h2);
   return dir + "/" + a + "/" + name;
}
static bool disk_read(const std::string& path, std::vector<char>* code) {
   if (path.empty()) return false;
   FILE* f = fopen(path.c_str(), "rb");
   if (!f) return false;
   bool ok = false;
   if (fseek(f, 0, SEEK_END) == 0) {
      const long n = ftell(f);
      if (n > 64 && n < (64l << 20) && fseek(f, 0, SEEK_SET) == 0) {
         code->resize((size_t) n);
         ok = fread(code->data(), 1, (size_t) n, f) == (size_t) n && memcmp(code->data(), "\x7f" "ELF", 4) == 0;
      }
   }
   fclose(f);
   if (!ok) code->clear();
   return ok;
}
static bool disk_write(const std::string& path, const std::vector<char>& code) {
   if (path.empty()) return false;
   const size_t slash = path.rfind('/');
   if (slash == std::string::npos || !make_dirs(path.substr(0, slash))) return false;
   char suffix[64];
   snprintf(suffix, sizeof(suffix), ".tmp.%d.%llx", (int) getpid(), (unsigned long long) std::hash<std::thread::id>()(std::this_thread::get_id()));
   const std::string tmp = path + suffix;
   FILE* f = fopen(tmp.c_str(), "wb");
   if (!f) return false;
   const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
   if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) {
      unlink(tmp.c_str());
      return false;
   }
   return true;
}

// ---------------------------------------------------------------- compile workers
static void stop_workers();
// Several processes on one host share the disk cache and meet the same shapes at the same time (the ranks of a multi-GPU run: eight ranks compiling
// the same hundred kernels each is eight times the compiler work on the same cores).  A worker CLAIMS a shape before compiling it — `<hash>.co.lock`,
// created exclusively —; a worker that finds the claim of another process waits for that process's code object instead (polling; a claim older than
// five minutes, or one that disappears without a code object, is ignored and the shape compiled here after all).
static std::atomic<int64_t> g_shared_from_peers{0};
static bool claim_or_wait(const std::string& path, std::vector<char>* code, bool* claimed) {
   *claimed = false;
   if (path.empty() || ldb_option("jit_share_compiles", 1) == 0) return false;
   const size_t slash = path.rfind('/');
   if (slash == std::string::npos || !make_dirs(path.substr(0, slash))) return false;
   const std::string lock = path + ".lock";
   for (int attempt = 0; attempt < 2; attempt++) {
      const int fd = open(lock.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644);
      if (fd >= 0) {
         close(fd);
         *claimed = true;
         return false;
      }
      if (errno != EEXIST) return false;
      struct stat st;
      if (stat(lock.c_str(), &st) == 0 && time(nullptr) - st.st_mtime > 300) { // a claim left behind by a process that died
         unlink(lock.c_str());
         continue;
      }
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(300)) {
         if (disk_read(path, code)) return true;
         if (stat(lock.c_str(), &st) != 0) return disk_read(path, code); // the claim is gone: its owner finished (or failed)
         {
            std::lock_guard<std::mutex> lk(g_mu);
            if (g_stop) return false;
         }
         std::this_thread::sleep_for(std::chrono::milliseconds(25));
      }
      return false;
   }
   return false;
}
static void run_job(JitJob& job) {
   auto t0 = std::chrono::steady_clock::now();
   std::vector<char> code;
   std::string err;
   bool claimed = false;
   const bool from_peer = claim_or_wait(job.disk_path, &code, &claimed);
   const bool ok = from_peer || compile(job.src, &code, &err, job.arch.c_str());
   const double ms = from_peer ? 0.0 : std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
   bool wrote = false;
   if (from_peer) g_shared_from_peers.fetch_add(1);
   if (ok && !from_peer) {
      wrote = disk_write(job.disk_path, code);
      if (const char* dir = getenv("LDB_JIT_DUMP_DIR")) { // code objects for llvm-objdump inspection
         const std::string path = std::string(dir) + "/" + job.dump_name + ".co";
         if (FILE* f = fopen(path.c_str(), "wb")) {
            fwrite(code.data(), 1, code.size(), f);
            fclose(f);
         }
      }
   }
   if (claimed) unlink((job.disk_path + ".lock").c_str());
   atexit(stop_workers); // (see stop_workers: ahead of the statics this compilation created)
   std::lock_guard<std::mutex> lock(g_mu);
   g_compile_ms += ms;
   if (ok) {
      job.mod->code = std::move(code);
      job.mod->state = JitModule::CODE_READY;
      if (from_peer) job.mod->from_disk = true; // (not compiled here: ldb_gpu_jit_info counts it under "taken from a peer process")
      g_disk_writes += wrote ? 1 : 0;
   } else {
      job.mod->error = err;
      job.mod->state = JitModule::FAILED;
      g_failed++;
   }
}
static void worker_main() {
   std::unique_lock<std::mutex> lock(g_mu);
   for (;;) {
      g_cv.wait(lock, [] { return g_stop || !g_queue.empty(); });
      if (g_stop) break;
      JitJob job = std::move(g_queue.front());
      g_queue.pop_front();
      g_running++;
      lock.unlock();
      run_job(job);
      lock.lock();
      g_running--;
      g_cv.notify_all();
   }
   g_workers--;
   g_cv.notify_all();
}
// at process exit: queued jobs are dropped, running compilations finish (hiprtc must not be torn down under a worker).  Exit handlers run in
// reverse order of registration and hiprtc / LLVM create statics lazily DURING compilations, so the handler is registered again after every finished
// compilation (it is idempotent): it then runs before the destructors of everything the finished compilations created.  Hosts that can should call
// ldb_gpu_jit_shutdown() themselves before they exit (the Python binding does, from `atexit`).
static void stop_workers() {
   std::unique_lock<std::mutex> lock(g_mu);
   g_stop = true;
   g_queue.clear();
   g_cv.notify_all();
   g_cv.wait_for(lock, std::chrono::seconds(60), [] { return g_workers == 0; });
}
extern "C" int32_t ldb_gpu_jit_shutdown(void) {
   stop_workers();
   return LDB_OK;
}
static void ensure_workers_locked() {
   if (g_workers > 0 || g_stop) return;
   atexit(stop_workers);
   int64_t n = ldb_option("jit_threads", 0);
   if (n <= 0) n = std::max<int64_t>(1, std::min<int64_t>(8, (int64_t) std::thread::hardware_concurrency() / 2));
   for (int64_t i = 0; i < n; i++) {
      g_workers++;
      std::thread(worker_main).detach();
   }
}

// load a CODE_READY entry into its device (the calling thread; g_mu held)
static void load_locked(JitModule* mod) {
   int prev = -1;
   (void) hipGetDevice(&prev);
   (void) hipSetDevice(mod->device);
   const hipError_t le = hipModuleLoadData(&mod->module, mod->code.data());
   if (prev >= 0 && prev != mod->device) (void) hipSetDevice(prev);
   if (le != hipSuccess) {
      mod->error = "hipModuleLoadData failed for the specialised kernel";
      mod->module = nullptr;
      mod->state = JitModule::FAILED;
      g_failed++;
   } else {
      mod->state = JitModule::LOADED;
      if (!mod->from_disk) g_compiled++;
   }
}

// the entry for (device, header, struct, kernels, meta): LOADED into `device` (load = true) or at least CODE_READY (load = false: the device-less
// self-test), or nullptr with *why — also while the compilation is still running and the caller does not want to wait (async)
static JitModule* jit_acquire(std::unique_lock<std::mutex>& lock, int device, const std::string& arch, const char* header, const char* struct_name, const char* kernels_src,
                              const char* kernel_name, const void* meta, size_t meta_bytes, bool async, bool load, std::string* why) {
   std::string key_nodev;
   key_nodev.reserve(meta_bytes + 256);
   key_nodev += arch;
   key_nodev += '|';
   key_nodev += header;
   key_nodev += '|';
   key_nodev += struct_name;
   key_nodev += '|';
   key_nodev += kernels_src;
   key_nodev += '|';
   key_nodev.append((const char*) meta, meta_bytes);
   const std::string key = std::to_string(device) + "|" + key_nodev;
   const uint64_t h = hash_bytes((const unsigned char*) key.data(), key.size());
   auto& bucket = g_cache[h];
   JitModule* mod = nullptr;
   for (auto& e : bucket)
      if (e->key == key) mod = e.get();
   if (!mod) {
      auto e = std::make_unique<JitModule>();
      e->key = key;
      e->device = device;
      bucket.push_back(std::move(e));
      mod = bucket.back().get();
      const std::string path = disk_path_for(arch, key_nodev);
      if (disk_read(path, &mod->code)) { // compiled by an earlier process (or for another device of this one)
         mod->from_disk = true;
         mod->state = JitModule::CODE_READY;
         if (load) load_locked(mod);
         if (mod->state != JitModule::FAILED) {
            g_disk_hits++;
         } else { // a damaged file: forget it and compile
            unlink(path.c_str());
            mod->from_disk = false;
            mod->error.clear();
            mod->code.clear();
            mod->state = JitModule::PENDING;
            g_failed--;
         }
      }
      if (mod->state == JitModule::PENDING) {
         JitJob job;
         job.mod = mod;
         job.arch = arch;
         job.disk_path = path;
         job.src = build_source(header, struct_name, kernels_src, (const unsigned char*) meta, meta_bytes);
         char dump[160];
         snprintf(dump, sizeof(dump), "%s_%016llx", kernel_name, (unsigned long long) h);
         job.dump_name = dump;
         if (const char* dir = getenv("LDB_JIT_DUMP_DIR")) { // the specialised translation unit, for offline ISA inspection
            const std::string p = std::string(dir) + "/" + dump + ".hip";
            if (FILE* f = fopen(p.c_str(), "wb")) {
               fwrite(job.src.data(), 1, job.src.size(), f);
               fclose(f);
            }
         }
         ensure_workers_locked();
         if (g_workers > 0) {
            g_queue.push_back(std::move(job));
            g_cv.notify_all();
         } else { // (no worker could be started: compile here)
            lock.unlock();
            run_job(job);
            lock.lock();
         }
      }
   } else if (mod->state == JitModule::LOADED) {
      g_hits++;
   }
   if (mod->state == JitModule::PENDING) {
      if (async) { // the caller launches its generic kernel now and finds the specialised one on a later call
         g_async_misses++;
         if (why) *why = "the specialised kernel is being compiled in the background";
         return nullptr;
      }
      g_cv.wait(lock, [&] { return mod->state != JitModule::PENDING; });
   }
   if (mod->state == JitModule::CODE_READY && load) load_locked(mod);
   if (mod->state == JitModule::FAILED) {
      if (why) *why = mod->error;
      return nullptr;
   }
   return mod;
}

hipFunction_t ldb_jit_kernel(int device, const char* header, const char* struct_name, const char* kernels_src, const char* kernel_name, const void* meta, size_t meta_bytes,
                             std::string* why) {
   // a module is loaded into ONE device: the cache is keyed by device too, and the load happens
   // with that device current (a process may hold contexts on several GPUs)
   const std::string arch = device_arch(device);
   const bool async = ldb_option("jit_async", 1) != 0;
   std::unique_lock<std::mutex> lock(g_mu);
   JitModule* mod = jit_acquire(lock, device, arch, header, struct_name, kernels_src, kernel_name, meta, meta_bytes, async, true, why);
   if (!mod) return nullptr;
   auto it = mod->fns.find(kernel_name);
   if (it != mod->fns.end()) return it->second;
   hipFunction_t fn = nullptr;
   if (hipModuleGetFunction(&fn, mod->module, kernel_name) != hipSuccess) {
      if (why) *why = std::string(kernel_name) + " not found in the specialised module";
      return nullptr;
   }
   mod->fns[kernel_name] = fn;
   return fn;
}

// block until no specialisation is queued or compiling (or `timeout_ms` passed; < 0 = no limit).  *pending = what is still outstanding
extern "C" int32_t ldb_gpu_jit_wait(int64_t timeout_ms, int64_t* pending) {
   std::unique_lock<std::mutex> lock(g_mu);
   auto idle = [] { return g_queue.empty() && g_running == 0; };
   if (timeout_ms < 0)
      g_cv.wait(lock, idle);
   else
      g_cv.wait_for(lock, std::chrono::milliseconds(timeout_ms), idle);
   if (pending) *pending = (int64_t) g_queue.size() + g_running;
   return LDB_OK;
}
// vals[0..n): kernels compiled in this process, in-memory hits, code objects taken from the disk cache, written to it, compilations outstanding,
// failed, calls answered "still compiling" (the caller ran its generic kernel), worker threads
extern "C" int32_t ldb_gpu_jit_info(int64_t* vals, int32_t n) {
   if (!vals || n < 0) LDB_FAIL(LDB_ERR_INVALID, "jit_info: NULL argument");
   std::lock_guard<std::mutex> lock(g_mu);
   const int64_t all[9] = {g_compiled, g_hits, g_disk_hits, g_disk_writes, (int64_t) g_queue.size() + g_running, g_failed, g_async_misses, (int64_t) g_workers, g_shared_from_peers.load()};
   for (int32_t i = 0; i < n; i++) vals[i] = i < 9 ? all[i] : 0;
   return LDB_OK;
}

// ---------------------------------------------------------------- group-by
static const char* GB_SPEC_SRC =
   "extern __shared__ __attribute__((aligned(16))) unsigned long long gb_lds_dyn[];\n"
   "#ifndef GB_ROWS\n#define GB_ROWS (LDB_META.batch_rows > 0 ? LDB_META.batch_rows : 4)\n#endif\n"
   "extern \"C\" __global__ __launch_bounds__(GB_BLOCK) void k_groupby_spec(const DGroupBy* __restrict__ d) { gb_body<GB_ROWS>(LDB_META, d, gb_lds_dyn); }\n"
   "extern \"C\" __global__ void k_gb_sorted_heads_spec(const DGroupBy* __restrict__ d, uint32_t* __restrict__ chunk_cnt) { gb_sorted_heads_body(LDB_META, d, chunk_cnt); }\n";

static void gb_meta(const DGroupBy* h, DGroupBy* m) {
   memcpy(m, h, sizeof(DGroupBy));
   m->n_rows = 0;
   m->g_cap = 0;
   m->g_keys = m->g_acc = m->g_flags = 0;
   m->direct_keys_out = 0;
   m->lds_slots = m->lds_reps = 0;
   m->kmin = 0;
   m->kmult = 0;
   m->chunk_off = 0;
   m->rep_rows_out = m->cross_flags = m->dense_groups = 0;
   ldb_jit_strip_keys(m->keys);
   for (int p = 0; p < LDB_MAX_PREDS; p++) ldb_jit_strip_pred(m->preds[p]);
   for (int p = 0; p < GB_MAX_CPREDS; p++) ldb_jit_strip_pred(m->cpreds[p]);
   for (int c = 0; c < GB_MAX_COLS; c++) ldb_jit_strip_col(m->cols[c]);
   for (int o = 0; o < GB_MAX_OUT; o++) m->outs[o].out_values = m->outs[o].out_valid = 0;
}

hipFunction_t ldb_jit_groupby_kernel(int device, const DGroupBy* h, const char* kernel, std::string* why) {
   auto meta = std::make_unique<DGroupBy>();
   gb_meta(h, meta.get());
   return ldb_jit_kernel(device, "ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, kernel, meta.get(), sizeof(DGroupBy), why);
}
hipFunction_t ldb_jit_groupby(int device, const DGroupBy* h, std::string* why) { return ldb_jit_groupby_kernel(device, h, "k_groupby_spec", why); }

// Compile-only check (no device needed): specialise the group-by kernel for a TPC-H-Q1-shaped
// descriptor and report the hiprtc log.  Used by the CPU-side tests and __graft_entry__.build().
extern "C" int32_t ldb_gpu_jit_compile_check(char* log, int32_t cap) {
   auto h = std::make_unique<DGroupBy>();
   memset(h.get(), 0, sizeof(DGroupBy));
   h->n_preds = 1;
   h->preds[0].col.type = LDB_T_DATE32;
   h->preds[0].col.width = 4;
   h->preds[0].op = LDB_F_LTE;
   h->preds[0].lo = 10471;
   h->keys.n_keys = 2;
   for (int k = 0; k < 2; k++) {
      h->keys.cols[k].type = LDB_T_CHAR4;
      h->keys.cols[k].width = 4;
   }
   h->n_cols = 2;
   for (int c = 0; c < 2; c++) {
      h->cols[c].type = LDB_T_DECIMAL128;
      h->cols[c].width = 16;
      h->cols[c].precision = 12;
      h->cols[c].scale = 2;
   }
   h->n_accs = 3;
   h->accs[0].kind = ACC_SUM64;
   h->accs[0].e.n_terms = 1;
   h->accs[0].e.t[0].n_factors = 1;
   h->accs[0].e.t[0].f[0] = {1, 0, 0, 1};
   h->accs[1].kind = ACC_SUM128;
   h->accs[1].word = 1;
   h->accs[1].e.n_terms = 1;
   h->accs[1].e.t[0].n_factors = 2;
   h->accs[1].e.t[0].f[0] = {1, 0, 0, 1};
   h->accs[1].e.t[0].f[1] = {1, 1, 100, -1};
   h->accs[2].kind = ACC_COUNT;
   h->accs[2].word = 3;
   h->accs[2].count_rows = 1;
   h->n_words = 4;
   h->use_lds = 1;
   auto meta = std::make_unique<DGroupBy>();
   gb_meta(h.get(), meta.get());
   std::vector<char> code;
   std::string err;
   bool ok = compile(build_source("ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, (const unsigned char*) meta.get(), sizeof(DGroupBy)), &code, &err);
   if (ok) { // the sorted-key high-cardinality shape (TPC-H Q18: one int32 key, SUM of a decimal, no LDS table, rows final inside the wave)
      auto q = std::make_unique<DGroupBy>();
      memset(q.get(), 0, sizeof(DGroupBy));
      q->keys.n_keys = 1;
      q->keys.cols[0].type = LDB_T_INT32;
      q->keys.cols[0].width = 4;
      q->n_cols = 1;
      q->cols[0].type = LDB_T_DECIMAL128;
      q->cols[0].width = 16;
      q->cols[0].precision = 12;
      q->cols[0].scale = 2;
      q->n_accs = 2;
      q->accs[0].kind = ACC_SUM128;
      q->accs[0].e.n_terms = 1;
      q->accs[0].e.t[0].n_factors = 1;
      q->accs[0].e.t[0].f[0] = {1, 0, 0, 1};
      q->accs[1].kind = ACC_COUNT;
      q->accs[1].word = 2;
      q->n_words = 3;
      q->n_outs = 1;
      q->outs[0].fn = LDB_AGG_SUM;
      q->outs[0].acc = 0;
      q->outs[0].cnt_acc = 1;
      q->outs[0].cnt_rows_acc = q->outs[0].cnt_pass_acc = -1;
      q->outs[0].wide = 1;
      q->outs[0].out_width = 16;
      q->dense_sorted = q->dense_out = 1;
      q->batch_rows = 4;
      gb_meta(q.get(), meta.get());
      code.clear();
      ok = compile(build_source("ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, (const unsigned char*) meta.get(), sizeof(DGroupBy)), &code, &err);
   }
   if (ok) ok = ldb_scan_jit_check(&err);
   if (ok) ok = ldb_join_jit_check(&err);
   if (ok) ok = ldb_expr_jit_check(&err);
   if (log && cap > 0) snprintf(log, (size_t) cap, "%s", err.c_str());
   if (ok) {
      if (const char* dump = getenv("LDB_JIT_DUMP")) { // code object for llvm-objdump inspection
         if (FILE* f = fopen(dump, "wb")) {
            fwrite(code.data(), 1, code.size(), f);
            fclose(f);
         }
      }
   }
   return ok ? LDB_OK : LDB_ERR_HIP;
}

// Device-less self-test of the specialiser's plumbing (CPU test suite): a TPC-H-Q6-shaped scan-free group-by descriptor is requested
// asynchronously (the first answer must be "still compiling"), awaited, found on disk under `cache_dir`, forgotten, and requested again — the second
// answer must come from the disk cache without a compilation.  Nothing is loaded into a device.
extern "C" int32_t ldb_gpu_jit_cache_selftest(char* log, int32_t cap) {
   auto say = [&](const std::string& m) {
      if (log && cap > 0) snprintf(log, (size_t) cap, "%s", m.c_str());
      return LDB_ERR_HIP;
   };
   if (cache_dir().empty()) return say("the disk cache is disabled");
   auto h = std::make_unique<DGroupBy>();
   memset(h.get(), 0, sizeof(DGroupBy));
   h->n_cols = 1;
   h->cols[0].type = LDB_T_DECIMAL128;
   h->cols[0].width = 16;
   h->cols[0].precision = 12;
   h->cols[0].scale = 2;
   h->n_accs = 1;
   h->accs[0].kind = ACC_SUM64;
   h->accs[0].e.n_terms = 1;
   h->accs[0].e.t[0].n_factors = 1;
   h->accs[0].e.t[0].f[0] = {1, 0, 0, 1};
   h->n_words = 1;
   h->use_lds = 1;
   auto meta = std::make_unique<DGroupBy>();
   gb_meta(h.get(), meta.get());
   meta->n_preds = 0;
   const std::string arch = "gfx950";
   const int device = -1; // an entry of its own: never loaded
   int64_t before[8], after[8];
   ldb_gpu_jit_info(before, 8);
   {
      std::unique_lock<std::mutex> lock(g_mu);
      std::string why;
      JitModule* m = jit_acquire(lock, device, arch, "ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, "k_groupby_spec", meta.get(), sizeof(DGroupBy), true, false, &why);
      if (m && !m->from_disk) return say("the first asynchronous request returned a module at once");
      if (m) return say("cache directory not empty for this key: use a fresh LDB_JIT_CACHE_DIR");
   }
   int64_t pending = -1;
   ldb_gpu_jit_wait(120000, &pending);
   if (pending != 0) return say("compilation still outstanding after 120 s");
   {
      std::unique_lock<std::mutex> lock(g_mu);
      std::string why;
      JitModule* m = jit_acquire(lock, device, arch, "ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, "k_groupby_spec", meta.get(), sizeof(DGroupBy), true, false, &why);
      if (!m || m->state != JitModule::CODE_READY || m->code.empty()) return say("no code object after the wait: " + why);
      // forget the entry: the next request must be answered from the disk
      for (auto& kv : g_cache)
         for (size_t i = 0; i < kv.second.size(); i++)
            if (kv.second[i].get() == m) {
               kv.second.erase(kv.second.begin() + (long) i);
               break;
            }
      JitModule* again = jit_acquire(lock, device, arch, "ldb_gb_kernel.h", "DGroupBy", GB_SPEC_SRC, "k_groupby_spec", meta.get(), sizeof(DGroupBy), true, false, &why);
      if (!again || !again->from_disk || again->code.empty()) return say("the second request was not answered from the disk cache: " + why);
   }
   ldb_gpu_jit_info(after, 8);
   if (after[2] != before[2] + 1 || after[3] != before[3] + 1 || after[6] != before[6] + 1) return say("statistics do not show one disk write, one disk hit and one asynchronous miss");
   return LDB_OK;
}

extern "C" int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms) {
   std::lock_guard<std::mutex> lock(g_mu);
   if (compiled) *compiled = g_compiled + g_disk_hits; // (modules made available to this process: compiled here or taken from the disk cache)
   if (cache_hits) *cache_hits = g_hits;
   if (compile_ms) *compile_ms = g_compile_ms;
   return LDB_OK;
}
