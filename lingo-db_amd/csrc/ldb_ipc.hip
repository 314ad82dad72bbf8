// ldb_ipc.hip — Arrow IPC FILE → HBM, inside the library.
//
// The reference keeps one `<table>.arrow` IPC file per table and reads all its record batches
// (LingoDBTable::ensureLoaded, src/runtime/storage/LingoDBTable.cpp:27-54: arrow::ipc::RecordBatchFileReader over a
// memory-mapped file).  This is the same step without libarrow: the file is mapped, the footer / schema / record-batch
// messages — flatbuffers, read with the four accessors below — are walked, every batch becomes an Arrow C-data-interface
// array whose buffers point INTO the mapping, and ldb_gpu_table_register copies them to the device (one hipMemcpy per
// buffer, straight from the page cache).  Format: Arrow columnar format 1.0 "IPC File Format" (magic ARROW1, footer with
// Blocks, encapsulated messages with the 0xFFFFFFFF continuation marker); flat primitive / decimal / date / utf8 /
// fixed-size-binary columns, uncompressed, little-endian, no dictionaries — what LingoDB's own tables contain.  Anything
// else is LDB_ERR_UNSUPPORTED with the field named, never a misread buffer: every offset is bounds-checked.
#include "ldb_internal.h"
#include <fcntl.h>
#include <memory>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

struct Bad {
   int32_t status;
   std::string what;
};
[[noreturn]] void bad(const std::string& what, int32_t status = LDB_ERR_INVALID) { throw Bad{status, what}; }

// ---- flatbuffers: a table is an int32 back-offset to its vtable; vtable = {u16 vtable bytes, u16 table bytes, u16 field offsets…}
struct Buf {
   const uint8_t* p = nullptr;
   size_t n = 0;
   template <typename T>
   T rd(size_t at) const {
      if (at + sizeof(T) > n || at + sizeof(T) < at) bad("IPC file: offset outside the file");
      T v;
      memcpy(&v, p + at, sizeof(T));
      return v;
   }
};
struct Tab {
   const Buf* b = nullptr;
   size_t pos = 0; // 0 = absent
   explicit operator bool() const { return pos != 0; }
   size_t field(int i) const { // absolute position of field i, 0 if absent
      const int32_t back = b->rd<int32_t>(pos);
      const size_t vt = (size_t) ((int64_t) pos - back);
      const uint16_t vbytes = b->rd<uint16_t>(vt);
      const size_t slot = 4 + 2 * (size_t) i;
      if (slot + 2 > vbytes) return 0;
      const uint16_t off = b->rd<uint16_t>(vt + slot);
      return off ? pos + off : 0;
   }
   template <typename T>
   T scalar(int i, T dflt) const {
      const size_t f = field(i);
      return f ? b->rd<T>(f) : dflt;
   }
   Tab table(int i) const {
      const size_t f = field(i);
      return f ? Tab{b, f + b->rd<uint32_t>(f)} : Tab{b, 0};
   }
   // vector field: position of element 0 and the element count
   size_t vec(int i, uint32_t* n) const {
      const size_t f = field(i);
      if (!f) {
         *n = 0;
         return 0;
      }
      const size_t v = f + b->rd<uint32_t>(f);
      *n = b->rd<uint32_t>(v);
      return v + 4;
   }
   std::string str(int i) const {
      uint32_t n;
      const size_t v = vec(i, &n);
      if (!v) return "";
      if (v + n > b->n) bad("IPC file: string outside the file");
      return std::string((const char*) b->p + v, n);
   }
};

// Schema.fbs: Type union member ids
enum { T_Int = 2, T_FloatingPoint = 3, T_Utf8 = 5, T_Bool = 6, T_Decimal = 7, T_Date = 8, T_FixedSizeBinary = 15, T_LargeUtf8 = 20 };

struct Col {
   std::string name, format;
   bool nullable = true;
   int n_buffers = 2; // validity + data; utf8: validity + offsets + data
};

std::string format_of(const Tab& field, const std::string& name, int* n_buffers) {
   const uint8_t tt = field.scalar<uint8_t>(2, 0);
   const Tab ty = field.table(3);
   if (field.table(4)) bad("IPC file: column '" + name + "' is dictionary-encoded", LDB_ERR_UNSUPPORTED);
   uint32_t n_children;
   field.vec(5, &n_children);
   if (n_children) bad("IPC file: column '" + name + "' is nested", LDB_ERR_UNSUPPORTED);
   *n_buffers = 2;
   switch (tt) {
      case T_Int: {
         const int32_t bits = ty ? ty.scalar<int32_t>(0, 0) : 0;
         const bool sign = ty ? ty.scalar<uint8_t>(1, 0) != 0 : false;
         if (!sign) bad("IPC file: column '" + name + "' is an unsigned integer", LDB_ERR_UNSUPPORTED);
         switch (bits) {
            case 8: return "c";
            case 16: return "s";
            case 32: return "i";
            case 64: return "l";
         }
         bad("IPC file: column '" + name + "': integer width " + std::to_string(bits), LDB_ERR_UNSUPPORTED);
      }
      case T_FloatingPoint: {
         const int16_t prec = ty ? ty.scalar<int16_t>(0, 0) : 0;
         if (prec == 1) return "f";
         if (prec == 2) return "g";
         bad("IPC file: column '" + name + "' is a half float", LDB_ERR_UNSUPPORTED);
      }
      case T_Utf8: *n_buffers = 3; return "u";
      case T_LargeUtf8: *n_buffers = 3; return "U";
      case T_Decimal: {
         const int32_t p = ty ? ty.scalar<int32_t>(0, 0) : 0, s = ty ? ty.scalar<int32_t>(1, 0) : 0, bits = ty ? ty.scalar<int32_t>(2, 128) : 128;
         if (bits != 128) bad("IPC file: column '" + name + "': decimal" + std::to_string(bits), LDB_ERR_UNSUPPORTED);
         return "d:" + std::to_string(p) + "," + std::to_string(s);
      }
      case T_Date: {
         const int16_t unit = ty ? ty.scalar<int16_t>(0, 1) : 1; // default MILLISECOND
         if (unit == 0) return "tdD";
         bad("IPC file: column '" + name + "' is date64", LDB_ERR_UNSUPPORTED);
      }
      case T_FixedSizeBinary: return "w:" + std::to_string(ty ? ty.scalar<int32_t>(0, 0) : 0);
      case T_Bool: bad("IPC file: column '" + name + "' is a bit-packed bool", LDB_ERR_UNSUPPORTED);
      default: bad("IPC file: column '" + name + "' has Arrow type id " + std::to_string(tt), LDB_ERR_UNSUPPORTED);
   }
}

struct Mapping {
   int fd = -1;
   void* p = MAP_FAILED;
   size_t n = 0;
   ~Mapping() {
      if (p != MAP_FAILED) munmap(p, n);
      if (fd >= 0) close(fd);
   }
};

struct Batch { // one record batch as C-data-interface structs (buffers point into the mapping)
   ArrowArray top;
   std::vector<ArrowArray> kids;
   std::vector<ArrowArray*> kid_ptrs;
   std::vector<std::vector<const void*>> bufs;
   const void* top_bufs[1] = {nullptr};
};

struct Parsed { // everything register needs; the C-data-interface structs point into these members and into the mapping
   Mapping m;
   std::vector<Col> cols;
   std::vector<ArrowSchema> ks;
   std::vector<ArrowSchema*> kp;
   ArrowSchema top;
   std::vector<std::unique_ptr<Batch>> batches;
   Batch empty;
   std::vector<ArrowArray*> bp;
};

void parse(const char* path, Parsed& P) {
   Mapping& m = P.m;
   m.fd = open(path, O_RDONLY);
   if (m.fd < 0) bad(std::string("IPC file: cannot open '") + path + "'");
   struct stat st;
   if (fstat(m.fd, &st) != 0 || st.st_size < 8 + 8 + 4) bad("IPC file: too short");
   m.n = (size_t) st.st_size;
   m.p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
   if (m.p == MAP_FAILED) bad("IPC file: mmap failed");
   const Buf file{(const uint8_t*) m.p, m.n};
   if (memcmp(file.p, "ARROW1", 6) != 0 || memcmp(file.p + m.n - 6, "ARROW1", 6) != 0) bad("IPC file: not an Arrow IPC file (magic ARROW1 missing; a stream is not a file)");
   const int32_t flen = file.rd<int32_t>(m.n - 10);
   if (flen <= 0 || (size_t) flen + 10 + 8 > m.n) bad("IPC file: bad footer length");
   const size_t fstart = m.n - 10 - (size_t) flen;
   const Tab footer{&file, fstart + file.rd<uint32_t>(fstart)};
   const Tab schema = footer.table(1);
   if (!schema) bad("IPC file: footer without a schema");
   if (schema.scalar<int16_t>(0, 0) != 0) bad("IPC file: big-endian data", LDB_ERR_UNSUPPORTED);
   uint32_t n_fields;
   const size_t fields = schema.vec(1, &n_fields);
   if (n_fields == 0 || n_fields > 4096) bad("IPC file: " + std::to_string(n_fields) + " columns");
   std::vector<Col>& cols = P.cols;
   cols.resize(n_fields);
   for (uint32_t c = 0; c < n_fields; c++) {
      const size_t at = fields + 4 * (size_t) c;
      const Tab f{&file, at + file.rd<uint32_t>(at)};
      cols[c].name = f.str(0);
      cols[c].nullable = f.scalar<uint8_t>(1, 0) != 0;
      cols[c].format = format_of(f, cols[c].name, &cols[c].n_buffers);
   }
   uint32_t n_dict;
   footer.vec(2, &n_dict);
   if (n_dict) bad("IPC file: dictionary batches", LDB_ERR_UNSUPPORTED);
   // ---- the schema as C data interface structs
   std::vector<ArrowSchema>& ks = P.ks;
   std::vector<ArrowSchema*>& kp = P.kp;
   ks.resize(n_fields);
   kp.resize(n_fields);
   for (uint32_t c = 0; c < n_fields; c++) {
      memset(&ks[c], 0, sizeof(ArrowSchema));
      ks[c].format = cols[c].format.c_str();
      ks[c].name = cols[c].name.c_str();
      ks[c].flags = cols[c].nullable ? 2 : 0; // ARROW_FLAG_NULLABLE
      kp[c] = &ks[c];
   }
   ArrowSchema& top = P.top;
   memset(&top, 0, sizeof(top));
   top.format = "+s";
   top.name = "";
   top.n_children = n_fields;
   top.children = kp.data();
   // ---- record batches
   uint32_t n_blocks;
   const size_t blocks = footer.vec(3, &n_blocks);
   std::vector<std::unique_ptr<Batch>>& batches = P.batches;
   for (uint32_t k = 0; k < n_blocks; k++) {
      const size_t bl = blocks + 24 * (size_t) k; // struct Block { offset: long; metaDataLength: int; (pad) bodyLength: long }
      const int64_t off = file.rd<int64_t>(bl), body_len = file.rd<int64_t>(bl + 16);
      const int32_t meta_len = file.rd<int32_t>(bl + 8);
      if (off < 8 || meta_len < 8 || body_len < 0 || (uint64_t) off + (uint64_t) meta_len + (uint64_t) body_len > m.n) bad("IPC file: record batch block outside the file");
      size_t mp = (size_t) off;
      if (file.rd<uint32_t>(mp) == 0xFFFFFFFFu) mp += 4; // continuation marker (format ≥ 0.15)
      mp += 4; // int32 flatbuffer size
      const Tab msg{&file, mp + file.rd<uint32_t>(mp)};
      if (msg.scalar<uint8_t>(1, 0) != 3) bad("IPC file: block " + std::to_string(k) + " is not a RecordBatch message"); // MessageHeader.RecordBatch = 3
      const Tab rb = msg.table(2);
      if (!rb) bad("IPC file: RecordBatch message without a header");
      if (rb.table(3)) bad("IPC file: compressed record batches", LDB_ERR_UNSUPPORTED);
      const int64_t length = rb.scalar<int64_t>(0, 0);
      if (length < 0 || length > (int64_t) 1 << 40) bad("IPC file: record batch " + std::to_string(k) + " claims " + std::to_string(length) + " rows");
      uint32_t n_nodes, n_bufs;
      const size_t nodes = rb.vec(1, &n_nodes), bufs = rb.vec(2, &n_bufs);
      if (n_nodes != n_fields) bad("IPC file: record batch with " + std::to_string(n_nodes) + " field nodes for " + std::to_string(n_fields) + " columns");
      const uint8_t* body = file.p + (size_t) off + (size_t) meta_len;
      auto b = std::make_unique<Batch>();
      b->kids.resize(n_fields);
      b->kid_ptrs.resize(n_fields);
      b->bufs.resize(n_fields);
      uint32_t bi = 0;
      for (uint32_t c = 0; c < n_fields; c++) {
         ArrowArray& a = b->kids[c];
         memset(&a, 0, sizeof(a));
         a.length = file.rd<int64_t>(nodes + 16 * (size_t) c);
         a.null_count = file.rd<int64_t>(nodes + 16 * (size_t) c + 8);
         if (a.length != length) bad("IPC file: column '" + cols[c].name + "' has " + std::to_string(a.length) + " rows in a batch of " + std::to_string(length));
         if (a.null_count < -1 || a.null_count > length) bad("IPC file: column '" + cols[c].name + "' claims " + std::to_string(a.null_count) + " NULLs in " + std::to_string(length) + " rows");

         a.n_buffers = cols[c].n_buffers;
         b->bufs[c].resize((size_t) a.n_buffers);
         for (int j = 0; j < a.n_buffers; j++, bi++) {
            if (bi >= n_bufs) bad("IPC file: record batch lists too few buffers");
            const int64_t bo = file.rd<int64_t>(bufs + 16 * (size_t) bi), blen = file.rd<int64_t>(bufs + 16 * (size_t) bi + 8);
            if (bo < 0 || blen < 0 || (uint64_t) bo + (uint64_t) blen > (uint64_t) body_len) bad("IPC file: buffer of column '" + cols[c].name + "' outside the batch body");
            // how many bytes the column needs from this buffer (the register call reads exactly these)
            uint64_t need = 0;
            const std::string& f = cols[c].format;
            if (j == 0 && a.null_count < 0) a.null_count = blen && length ? 1 : 0; // "unknown": a validity buffer, if there is one, decides (the register call counts)
            if (j == 0) need = a.null_count ? ((uint64_t) length + 7) / 8 : 0;
            else if (a.n_buffers == 3 && j == 1) need = ((uint64_t) length + 1) * (f == "U" ? 8 : 4);
            else if (a.n_buffers == 2) {
               const int64_t fw = f[0] == 'w' ? (int64_t) atoll(f.c_str() + 2) : 0;
               if (f[0] == 'w' && (fw <= 0 || fw > 1 << 20)) bad("IPC file: fixed_size_binary column '" + cols[c].name + "' of width " + std::to_string(fw));
               uint64_t w = f == "c" ? 1 : f == "s" ? 2 : (f == "i" || f == "f" || f == "tdD") ? 4 : (f == "l" || f == "g") ? 8 : f[0] == 'd' ? 16 : f[0] == 'w' ? (uint64_t) fw : 0;
               need = w * (uint64_t) length;
            }
            // several Arrow writers emit a 0-byte offsets buffer for a zero-length array (RecordBatchFileReader accepts it)
            const bool empty_offsets = a.n_buffers == 3 && j == 1 && length == 0 && blen == 0;
            if ((uint64_t) blen < need && !empty_offsets) bad("IPC file: buffer of column '" + cols[c].name + "' is shorter than its " + std::to_string(length) + " rows need");
            b->bufs[c][(size_t) j] = blen == 0 || (j == 0 && a.null_count == 0) ? nullptr : body + bo;
            if (empty_offsets) {
               static const int64_t zero_offsets[2] = {0, 0};
               b->bufs[c][1] = zero_offsets;
            }
            if (a.n_buffers == 3 && j == 2 && length > 0) { // every offset must stay inside the data buffer, in non-decreasing order
               const uint8_t* offs = body + file.rd<int64_t>(bufs + 16 * (size_t) (bi - 1));
               const bool wide = f == "U";
               auto at = [&](int64_t i) -> int64_t {
                  if (wide) {
                     int64_t v;
                     memcpy(&v, offs + 8 * (size_t) i, 8);
                     return v;
                  }
                  int32_t v;
                  memcpy(&v, offs + 4 * (size_t) i, 4);
                  return v;
               };
               int64_t prev = at(0);
               if (prev < 0) bad("IPC file: negative string offset in column '" + cols[c].name + "'");
               for (int64_t i = 1; i <= length; i++) {
                  const int64_t cur = at(i);
                  if (cur < prev) bad("IPC file: string offsets of column '" + cols[c].name + "' decrease at row " + std::to_string(i - 1));
                  prev = cur;
               }
               if (prev > blen) bad("IPC file: string offsets of column '" + cols[c].name + "' exceed its data buffer");
               if (!b->bufs[c][2]) b->bufs[c][2] = body + bo; // an empty data buffer still needs a valid base
            }
         }
         a.buffers = b->bufs[c].data();
         b->kid_ptrs[c] = &a;
      }
      memset(&b->top, 0, sizeof(ArrowArray));
      b->top.length = length;
      b->top.n_buffers = 1;
      b->top.buffers = b->top_bufs;
      b->top.n_children = n_fields;
      b->top.children = b->kid_ptrs.data();
      batches.push_back(std::move(b));
   }
   std::vector<ArrowArray*>& bp = P.bp;
   for (auto& b : batches) bp.push_back(&b->top);
   // an empty file (schema only) registers a zero-row table through one empty batch
   Batch& empty = P.empty;
   if (bp.empty()) {
      empty.kids.resize(n_fields);
      empty.kid_ptrs.resize(n_fields);
      empty.bufs.resize(n_fields);
      for (uint32_t c = 0; c < n_fields; c++) {
         memset(&empty.kids[c], 0, sizeof(ArrowArray));
         empty.kids[c].n_buffers = cols[c].n_buffers;
         empty.bufs[c].assign((size_t) cols[c].n_buffers, nullptr);
         static const int64_t zero_offsets[2] = {0, 0};
         if (cols[c].n_buffers == 3) empty.bufs[c][1] = zero_offsets;
         empty.kids[c].buffers = empty.bufs[c].data();
         empty.kid_ptrs[c] = &empty.kids[c];
      }
      memset(&empty.top, 0, sizeof(ArrowArray));
      empty.top.n_buffers = 1;
      empty.top.buffers = empty.top_bufs;
      empty.top.n_children = n_fields;
      empty.top.children = empty.kid_ptrs.data();
      bp.push_back(&empty.top);
   }
}

int32_t load(ldb_ctx* ctx, const char* name, const char* path, int32_t narrow, ldb_table** out) {
   Parsed P;
   parse(path, P);
   const int32_t rc = ldb_gpu_table_register(ctx, name, &P.top, P.bp.data(), (int64_t) P.bp.size(), narrow, out);
   (void) hipStreamSynchronize(ctx->stream); // the copies read the mapping: finish them before it goes away (also after a failed registration: some were queued)
   return rc;
}

} // namespace

extern "C" int32_t ldb_gpu_table_load_ipc(ldb_ctx* ctx, const char* name, const char* path, int32_t narrow_decimals, ldb_table** out) {
   if (!ctx || !name || !path || !out) LDB_FAIL(LDB_ERR_INVALID, "table_load_ipc: NULL argument");
   try {
      return load(ctx, name, path, narrow_decimals, out);
   } catch (const Bad& b) {
      LDB_FAIL(b.status, "%s", b.what.c_str());
   } catch (const std::exception& e) {
      LDB_FAIL(LDB_ERR_INVALID, "table_load_ipc: %s", e.what());
   }
}
// the parse alone, no device: {"columns": [{"name", "format", "nullable"}], "batches": [rows, …], "rows": total} — what the
// loader would register (format = Arrow C data interface format string)
extern "C" int32_t ldb_gpu_ipc_describe(const char* path, char* out, int64_t cap) {
   if (!path || !out || cap < 2) LDB_FAIL(LDB_ERR_INVALID, "ipc_describe: NULL argument");
   try {
      Parsed P;
      parse(path, P);
      std::string j = "{\"columns\": [";
      for (size_t c = 0; c < P.cols.size(); c++) {
         std::string nm;
         for (char ch : P.cols[c].name) {
            if (ch == '"' || ch == '\\') nm += '\\';
            nm += (unsigned char) ch < 0x20 ? ' ' : ch;
         }
         j += std::string(c ? ", " : "") + "{\"name\": \"" + nm + "\", \"format\": \"" + P.cols[c].format + "\", \"nullable\": " + (P.cols[c].nullable ? "true" : "false") + "}";
      }
      j += "], \"batches\": [";
      int64_t total = 0;
      for (size_t b = 0; b < P.batches.size(); b++) {
         j += (b ? ", " : "") + std::to_string(P.batches[b]->top.length);
         total += P.batches[b]->top.length;
      }
      j += "], \"rows\": " + std::to_string(total) + "}";
      if ((int64_t) j.size() + 1 > cap) LDB_FAIL(LDB_ERR_INVALID, "ipc_describe: buffer of %lld bytes needed", (long long) j.size() + 1);
      memcpy(out, j.c_str(), j.size() + 1);
      return LDB_OK;
   } catch (const Bad& b) {
      LDB_FAIL(b.status, "%s", b.what.c_str());
   } catch (const std::exception& e) {
      LDB_FAIL(LDB_ERR_INVALID, "ipc_describe: %s", e.what());
   }
}
