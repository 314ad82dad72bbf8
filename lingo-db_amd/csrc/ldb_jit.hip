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

struct JitEntry {
   std::vector<unsigned char> key;
   hipModule_t module = nullptr;
   hipFunction_t fn = nullptr;
   std::string error; // non-empty: compilation failed once, do not retry
};
static std::mutex g_mu;
static std::unordered_map<uint64_t, std::vector<JitEntry>> g_cache;
static int64_t g_compiled = 0, g_hits = 0;
static double g_compile_ms = 0;

bool ldb_jit_wanted(int64_t n_rows) {
   static int enabled = -1;
   static int64_t min_rows = 4000000;
   if (enabled < 0) {
      const char* e = getenv("LDB_JIT");
      enabled = (e && e[0] == '0') ? 0 : 1;
      if (const char* m = getenv("LDB_JIT_MIN_ROWS")) min_rows = atoll(m);
   }
   return enabled && n_rows >= min_rows;
}

static uint64_t fnv1a(const unsigned char* p, size_t n) {
   uint64_t h = 1469598103934665603ull;
   for (size_t i = 0; i < n; i++) {
      h ^= p[i];
      h *= 1099511628211ull;
   }
   return h;
}

// addresses become presence flags, per-launch sizes are cleared: what remains is the metadata
static void strip_col(DCol& c) {
   c.values = 0;
   c.offsets = c.offsets ? 1 : 0;
   c.validity = c.validity ? 1 : 0;
   c.rowids = c.rowids ? 1 : 0;
}
static void strip_pred(DPred& p) {
   strip_col(p.col);
   strip_col(p.rhs);
}
static void make_meta(const DGroupBy* h, DGroupBy* m) {
   memcpy(m, h, sizeof(DGroupBy));
   m->n_rows = 0;
   m->g_cap = 0;
   m->g_keys = m->g_acc = m->g_flags = 0;
   m->lds_slots = m->lds_reps = 0;
   for (int k = 0; k < LDB_MAX_KEYS; k++) strip_col(m->keys.cols[k]);
   for (int p = 0; p < LDB_MAX_PREDS; p++) strip_pred(m->preds[p]);
   for (int p = 0; p < GB_MAX_CPREDS; p++) strip_pred(m->cpreds[p]);
   for (int c = 0; c < GB_MAX_COLS; c++) strip_col(m->cols[c]);
   for (int o = 0; o < GB_MAX_OUT; o++) m->outs[o].out_values = m->outs[o].out_valid = 0;
}

static std::string build_source(const unsigned char* meta, size_t n) {
   std::string s;
   s.reserve(n * 5 + 2048);
   s += "#define LDB_JIT_SPECIALIZED 1\n#include \"ldb_gb_kernel.h\"\n";
   s += "struct LdbMetaBytes { unsigned char b[" + std::to_string(n) + "]; };\n";
   s += "static constexpr LdbMetaBytes LDB_META_BYTES = {{";
   char buf[8];
   for (size_t i = 0; i < n; i++) {
      snprintf(buf, sizeof(buf), "%u,", (unsigned) meta[i]);
      s += buf;
      if ((i & 63) == 63) s += "\n";
   }
   s += "}};\n";
   s += "__device__ static constexpr DGroupBy LDB_META = __builtin_bit_cast(DGroupBy, LDB_META_BYTES);\n";
   s += "extern __shared__ __attribute__((aligned(16))) unsigned long long gb_lds_dyn[];\n";
   s += "extern \"C\" __global__ __launch_bounds__(GB_BLOCK) void k_groupby_spec(const DGroupBy* __restrict__ d) { gb_body(LDB_META, d, gb_lds_dyn); }\n";
   return s;
}

static bool compile(const std::string& src, JitEntry* e) {
   std::vector<const char*> names, texts;
   for (int i = 0; i < g_n_headers; i++) {
      names.push_back(g_headers[i].name);
      texts.push_back(g_headers[i].text);
   }
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src.c_str(), "ldb_groupby_spec.hip", g_n_headers, texts.data(), names.data()) != HIPRTC_SUCCESS) {
      e->error = "hiprtcCreateProgram failed";
      return false;
   }
   const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
   hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
   if (r != HIPRTC_SUCCESS) {
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, '\0');
      if (ls) hiprtcGetProgramLog(prog, log.data());
      e->error = std::string("hiprtc: ") + hiprtcGetErrorString(r) + ": " + log.substr(0, 1500);
      hiprtcDestroyProgram(&prog);
      return false;
   }
   size_t cs = 0;
   hiprtcGetCodeSize(prog, &cs);
   std::vector<char> code(cs);
   hiprtcGetCode(prog, code.data());
   hiprtcDestroyProgram(&prog);
   if (hipModuleLoadData(&e->module, code.data()) != hipSuccess) {
      e->error = "hipModuleLoadData failed for the specialised kernel";
      return false;
   }
   if (hipModuleGetFunction(&e->fn, e->module, "k_groupby_spec") != hipSuccess) {
      e->error = "k_groupby_spec not found in the specialised module";
      return false;
   }
   return true;
}

hipFunction_t ldb_jit_groupby(const DGroupBy* h, std::string* why) {
   auto meta = std::make_unique<DGroupBy>();
   make_meta(h, meta.get());
   const unsigned char* bytes = (const unsigned char*) meta.get();
   const uint64_t key = fnv1a(bytes, sizeof(DGroupBy));
   std::lock_guard<std::mutex> lock(g_mu);
   auto& bucket = g_cache[key];
   for (auto& e : bucket) {
      if (e.key.size() == sizeof(DGroupBy) && !memcmp(e.key.data(), bytes, sizeof(DGroupBy))) {
         if (e.fn) g_hits++;
         else if (why) *why = e.error;
         return e.fn;
      }
   }
   JitEntry e;
   e.key.assign(bytes, bytes + sizeof(DGroupBy));
   auto t0 = std::chrono::steady_clock::now();
   bool ok = compile(build_source(bytes, sizeof(DGroupBy)), &e);
   g_compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
   if (ok) g_compiled++;
   else if (why) *why = e.error;
   bucket.push_back(std::move(e));
   return bucket.back().fn;
}

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
   make_meta(h.get(), meta.get());
   std::string src = build_source((const unsigned char*) meta.get(), sizeof(DGroupBy));
   std::vector<const char*> names, texts;
   for (int i = 0; i < g_n_headers; i++) {
      names.push_back(g_headers[i].name);
      texts.push_back(g_headers[i].text);
   }
   hiprtcProgram prog;
   if (hiprtcCreateProgram(&prog, src.c_str(), "ldb_groupby_spec.hip", g_n_headers, texts.data(), names.data()) != HIPRTC_SUCCESS) {
      if (log && cap > 0) snprintf(log, (size_t) cap, "hiprtcCreateProgram failed");
      return LDB_ERR_HIP;
   }
   const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics"};
   hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
   size_t ls = 0;
   hiprtcGetProgramLogSize(prog, &ls);
   std::string l(ls, '\0');
   if (ls) hiprtcGetProgramLog(prog, l.data());
   if (log && cap > 0) snprintf(log, (size_t) cap, "%s%s", r == HIPRTC_SUCCESS ? "" : hiprtcGetErrorString(r), l.c_str());
   size_t cs = 0;
   if (r == HIPRTC_SUCCESS) hiprtcGetCodeSize(prog, &cs);
   if (cs) {
      if (const char* dump = getenv("LDB_JIT_DUMP")) { // code object for llvm-objdump inspection
         std::vector<char> code(cs);
         hiprtcGetCode(prog, code.data());
         if (FILE* f = fopen(dump, "wb")) {
            fwrite(code.data(), 1, cs, f);
            fclose(f);
         }
      }
   }
   hiprtcDestroyProgram(&prog);
   return (r == HIPRTC_SUCCESS && cs > 0) ? LDB_OK : LDB_ERR_HIP;
}

extern "C" int32_t ldb_gpu_jit_stats(int64_t* compiled, int64_t* cache_hits, double* compile_ms) {
   std::lock_guard<std::mutex> lock(g_mu);
   if (compiled) *compiled = g_compiled;
   if (cache_hits) *cache_hits = g_hits;
   if (compile_ms) *compile_ms = g_compile_ms;
   return LDB_OK;
}
