// ldb_jit.hip — run-time kernel specialisation with hiprtc (see ldb_jit.h).
#include "ldb_jit.h"
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdlib>
#include <memory>
#include <mutex>
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

struct JitModule {
   std::string key; // header | struct | kernel source | metadata bytes
   hipModule_t module = nullptr;
   std::unordered_map<std::string, hipFunction_t> fns;
   std::vector<char> code;
   std::string error; // non-empty: compilation failed once, do not retry
};
static std::mutex g_mu;
static std::unordered_map<uint64_t, std::vector<std::unique_ptr<JitModule>>> g_cache;
static int64_t g_compiled = 0, g_hits = 0;
static double g_compile_ms = 0;

bool ldb_jit_wanted(int64_t n_rows) { return ldb_option("jit", 1) != 0 && n_rows >= ldb_option("jit_min_rows", 4000000); }

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
      static int seq = 0;
      char path[512];
      snprintf(path, sizeof(path), "%s/ldb_spec_%03d.co", dir, seq++);
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

hipFunction_t ldb_jit_kernel(int device, const char* header, const char* struct_name, const char* kernels_src, const char* kernel_name, const void* meta, size_t meta_bytes,
                             std::string* why) {
   // a module is loaded into ONE device: the cache is keyed by device too, and the load happens
   // with that device current (a process may hold contexts on several GPUs)
   const std::string arch = device_arch(device);
   std::string key;
   key.reserve(meta_bytes + 256);
   key += std::to_string(device);
   key += '|';
   key += arch;
   key += '|';
   key += header;
   key += '|';
   key += struct_name;
   key += '|';
   key += kernels_src;
   key += '|';
   key.append((const char*) meta, meta_bytes);
   const uint64_t h = hash_bytes((const unsigned char*) key.data(), key.size());
   std::lock_guard<std::mutex> lock(g_mu);
   auto& bucket = g_cache[h];
   JitModule* mod = nullptr;
   for (auto& e : bucket)
      if (e->key == key) mod = e.get();
   if (!mod) {
      auto e = std::make_unique<JitModule>();
      e->key = key;
      auto t0 = std::chrono::steady_clock::now();
      const std::string src = build_source(header, struct_name, kernels_src, (const unsigned char*) meta, meta_bytes);
      if (const char* dir = getenv("LDB_JIT_DUMP_DIR")) { // the specialised translation unit, for offline ISA inspection
         char path[512];
         snprintf(path, sizeof(path), "%s/%s_%016llx.hip", dir, kernel_name, (unsigned long long) h);
         if (FILE* f = fopen(path, "wb")) {
            fwrite(src.data(), 1, src.size(), f);
            fclose(f);
         }
      }
      bool ok = compile(src, &e->code, &e->error, arch.c_str());
      g_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ok) {
         if (const char* dir = getenv("LDB_JIT_DUMP_DIR")) { // code objects for llvm-objdump inspection
            char path[512];
            snprintf(path, sizeof(path), "%s/%s_%016llx.co", dir, kernel_name, (unsigned long long) h);
            if (FILE* f = fopen(path, "wb")) {
               fwrite(e->code.data(), 1, e->code.size(), f);
               fclose(f);
            }
         }
         int prev = -1;
         (void) hipGetDevice(&prev);
         (void) hipSetDevice(device);
         const hipError_t le = hipModuleLoadData(&e->module, e->code.data());
         if (prev >= 0 && prev != device) (void) hipSetDevice(prev);
         if (le != hipSuccess) {
            e->error = "hipModuleLoadData failed for the specialised kernel";
            e->module = nullptr;
         } else {
            g_compiled++;
         }
      }
      bucket.push_back(std::move(e));
      mod = bucket.back().get();
   } else if (mod->module) {
      g_hits++;
   }
   if (!mod->module) {
      if (why) *why = mod->error;
      return nullptr;
   }
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

extern "C" int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms) {
   std::lock_guard<std::mutex> lock(g_mu);
   if (compiled) *compiled = g_compiled;
   if (cache_hits) *cache_hits = g_hits;
   if (compile_ms) *compile_ms = g_compile_ms;
   return LDB_OK;
}
