// ldb_json.hpp — the minimal JSON reader shared by the plan interpreter (ldb_plan.cpp) and the
// sub-operator dump consumer (ldb_subop.cpp).  Bounded nesting, \u escapes, integers kept exact.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ldbjson {

// ================================================================== minimal JSON
struct J {
   enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
   bool b = false;
   double num = 0;
   int64_t inum = 0;
   bool isInt = false;
   std::string str;
   std::vector<J> arr;
   std::vector<std::pair<std::string, J>> obj;
   const J* get(const char* key) const {
      for (auto& kv : obj)
         if (kv.first == key) return &kv.second;
      return nullptr;
   }
   const J& at(const char* key) const {
      const J* v = get(key);
      if (!v) throw std::runtime_error(std::string("plan: missing field '") + key + "'");
      return *v;
   }
   const std::string& s(const char* key) const {
      const J& v = at(key);
      if (v.kind != STR) throw std::runtime_error(std::string("plan: field '") + key + "' must be a string");
      return v.str;
   }
   std::string sOr(const char* key, const std::string& d) const {
      const J* v = get(key);
      return v && v->kind == STR ? v->str : d;
   }
   int64_t iOr(const char* key, int64_t d) const {
      const J* v = get(key);
      return v && v->kind == NUM ? (v->isInt ? v->inum : (int64_t) v->num) : d;
   }
   bool bOr(const char* key, bool d) const {
      const J* v = get(key);
      return v && v->kind == BOOL ? v->b : d;
   }
};
struct JParser {
   const char* p;
   explicit JParser(const char* text) : p(text) {}
   [[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string("plan JSON: ") + what + " near '" + std::string(p).substr(0, 24) + "'"); }
   void ws() {
      while (*p && isspace((unsigned char) *p)) p++;
   }
   int depth = 0; // nesting of the value being parsed (plans nest ~6 deep; a bound keeps a hostile text from exhausting the stack)
   struct Nest {
      JParser& jp;
      explicit Nest(JParser& q) : jp(q) {
         if (++jp.depth > 64) jp.fail("nesting deeper than 64");
      }
      ~Nest() { jp.depth--; }
   };
   J value() {
      Nest nest(*this);
      ws();
      J j;
      if (*p == '{') {
         p++;
         j.kind = J::OBJ;
         ws();
         if (*p == '}') {
            p++;
            return j;
         }
         for (;;) {
            ws();
            if (*p != '"') fail("object key expected");
            std::string k = string();
            ws();
            if (*p++ != ':') fail("':' expected");
            j.obj.emplace_back(std::move(k), value());
            ws();
            if (*p == ',') {
               p++;
               continue;
            }
            if (*p == '}') {
               p++;
               return j;
            }
            fail("',' or '}' expected");
         }
      }
      if (*p == '[') {
         p++;
         j.kind = J::ARR;
         ws();
         if (*p == ']') {
            p++;
            return j;
         }
         for (;;) {
            j.arr.push_back(value());
            ws();
            if (*p == ',') {
               p++;
               continue;
            }
            if (*p == ']') {
               p++;
               return j;
            }
            fail("',' or ']' expected");
         }
      }
      if (*p == '"') {
         j.kind = J::STR;
         j.str = string();
         return j;
      }
      if (!strncmp(p, "true", 4)) {
         p += 4;
         j.kind = J::BOOL;
         j.b = true;
         return j;
      }
      if (!strncmp(p, "false", 5)) {
         p += 5;
         j.kind = J::BOOL;
         return j;
      }
      if (!strncmp(p, "null", 4)) {
         p += 4;
         return j;
      }
      if (*p == '-' || isdigit((unsigned char) *p)) {
         const char* b = p;
         if (*p == '-') p++;
         if (!isdigit((unsigned char) *p)) fail("digit expected after '-'");
         while (isdigit((unsigned char) *p)) p++;
         bool isInt = true;
         if (*p == '.' || *p == 'e' || *p == 'E') {
            isInt = false;
            while (*p && (isdigit((unsigned char) *p) || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) p++;
         }
         j.kind = J::NUM;
         j.isInt = isInt;
         std::string t(b, p);
         try {
            if (isInt) j.inum = std::stoll(t);
            j.num = std::stod(t);
         } catch (const std::exception&) {
            fail("number out of range or malformed");
         }
         return j;
      }
      fail("value expected");
   }
   std::string string() {
      std::string out;
      p++; // opening quote
      while (*p && *p != '"') {
         if (*p == '\\') {
            p++;
            if (!*p) fail("unterminated escape"); // (never step over the terminating NUL)
            switch (*p) {
               case 'n': out += '\n'; break;
               case 't': out += '\t'; break;
               case 'r': out += '\r'; break;
               case 'b': out += '\b'; break;
               case 'f': out += '\f'; break;
               case 'u': { // \uXXXX → UTF-8; a surrogate pair \uD8xx\uDCxx is one code point (4 bytes), a lone surrogate is an error
                  auto hex4 = [&](const char* q) -> unsigned {
                     unsigned v = 0;
                     for (int i = 1; i <= 4; i++) {
                        if (!isxdigit((unsigned char) q[i])) fail("\\u needs four hex digits"); // (also stops at the NUL)
                        v = v * 16 + (unsigned) (isdigit((unsigned char) q[i]) ? q[i] - '0' : (tolower(q[i]) - 'a' + 10));
                     }
                     return v;
                  };
                  unsigned cp = hex4(p);
                  p += 4;
                  if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone low surrogate in a string");
                  if (cp >= 0xD800 && cp <= 0xDBFF) {
                     if (p[1] != '\\' || p[2] != 'u') fail("high surrogate without its low surrogate");
                     const unsigned lo = hex4(p + 2);
                     if (lo < 0xDC00 || lo > 0xDFFF) fail("high surrogate without its low surrogate");
                     p += 6;
                     cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                     out += (char) (0xF0 | (cp >> 18));
                     out += (char) (0x80 | ((cp >> 12) & 0x3F));
                     out += (char) (0x80 | ((cp >> 6) & 0x3F));
                     out += (char) (0x80 | (cp & 0x3F));
                     break;
                  }
                  if (cp < 0x80) out += (char) cp;
                  else if (cp < 0x800) {
                     out += (char) (0xC0 | (cp >> 6));
                     out += (char) (0x80 | (cp & 0x3F));
                  } else {
                     out += (char) (0xE0 | (cp >> 12));
                     out += (char) (0x80 | ((cp >> 6) & 0x3F));
                     out += (char) (0x80 | (cp & 0x3F));
                  }
                  break;
               }
               default: out += *p; break; // \" \\ \/
            }
            p++;
         } else {
            out += *p++;
         }
      }
      if (*p != '"') fail("unterminated string");
      p++;
      return out;
   }
};

inline std::string quote(const std::string& s) { // a JSON string literal
   std::string o = "\"";
   for (unsigned char c : s) {
      if (c == '"' || c == '\\') {
         o += '\\';
         o += (char) c;
      } else if (c < 0x20) {
         char buf[8];
         snprintf(buf, sizeof(buf), "\\u%04x", c);
         o += buf;
      } else {
         o += (char) c;
      }
   }
   return o + "\"";
}

} // namespace ldbjson
