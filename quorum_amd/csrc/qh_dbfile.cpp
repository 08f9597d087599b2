// qh_dbfile.cpp -- header writer/reader of the "binary/quorum_db" container
// (see qh_dbfile.h for what it replaces upstream).
#include "qh_dbfile.h"
#include "qh_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <limits.h>
#include <map>
#include <memory>

// ------------------------------------------------------------------ mini JSON

namespace {

struct jval {
  enum kind_t { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  bool b = false;
  uint64_t u = 0;     // integer value (all numbers in our headers are unsigned integers)
  double d = 0;
  std::string s;
  std::vector<jval> a;
  std::vector<std::pair<std::string, jval>> o;
  const jval* get(const char* key) const {
    for (auto& kv : o)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
};

struct jparser {
  const char* p;
  const char* e;
  std::string err;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  bool str(std::string& out) {
    if (p >= e || *p != '"') return fail("expected string");
    ++p;
    out.clear();
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) return fail("bad escape");
        switch (*p) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            if (e - p < 5) return fail("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            if (cp < 0x80) out += (char)cp;
            else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
            else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
            p += 4;
            break;
          }
          default: out += *p;
        }
        ++p;
      } else {
        out += *p++;
      }
    }
    if (p >= e) return fail("unterminated string");
    ++p;
    return true;
  }
  bool value(jval& v, int depth = 0) {
    if (depth > 32) return fail("nesting too deep");
    ws();
    if (p >= e) return fail("unexpected end");
    if (*p == '{') {
      v.kind = jval::OBJ;
      ++p; ws();
      if (p < e && *p == '}') { ++p; return true; }
      for (;;) {
        ws();
        std::string k;
        if (!str(k)) return false;
        ws();
        if (p >= e || *p != ':') return fail("expected ':'");
        ++p;
        jval c;
        if (!value(c, depth + 1)) return false;
        v.o.emplace_back(std::move(k), std::move(c));
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == '}') { ++p; return true; }
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      v.kind = jval::ARR;
      ++p; ws();
      if (p < e && *p == ']') { ++p; return true; }
      for (;;) {
        jval c;
        if (!value(c, depth + 1)) return false;
        v.a.push_back(std::move(c));
        ws();
        if (p < e && *p == ',') { ++p; continue; }
        if (p < e && *p == ']') { ++p; return true; }
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') { v.kind = jval::STR; return str(v.s); }
    if (e - p >= 4 && !strncmp(p, "true", 4)) { v.kind = jval::BOOL; v.b = true; p += 4; return true; }
    if (e - p >= 5 && !strncmp(p, "false", 5)) { v.kind = jval::BOOL; v.b = false; p += 5; return true; }
    if (e - p >= 4 && !strncmp(p, "null", 4)) { v.kind = jval::NUL; p += 4; return true; }
    if (*p == '-' || (*p >= '0' && *p <= '9')) {
      const char* s = p;
      if (*p == '-') ++p;
      while (p < e && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-')) ++p;
      std::string t(s, p);
      v.kind = jval::NUM;
      v.d = strtod(t.c_str(), nullptr);
      v.u = t.find_first_of(".eE-") == std::string::npos ? strtoull(t.c_str(), nullptr, 10) : (uint64_t)v.d;
      return true;
    }
    return fail("unexpected character");
  }
};

void jescape(const std::string& s, std::string& out) {
  out += '"';
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\t': out += "\\t"; break;
      case '\r': out += "\\r"; break;
      default:
        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
        else out += (char)c;
    }
  }
  out += '"';
}

void jkv(std::string& o, const char* k, const std::string& v, bool comma = true) {
  jescape(k, o); o += ':'; jescape(v, o); if (comma) o += ',';
}
void jkv(std::string& o, const char* k, uint64_t v, bool comma = true) {
  jescape(k, o); o += ':'; o += std::to_string(v); if (comma) o += ',';
}

uint64_t ju(const jval& root, const char* k, uint64_t dflt = 0) {
  const jval* v = root.get(k);
  return v && v->kind == jval::NUM ? v->u : dflt;
}
std::string js(const jval& root, const char* k) {
  const jval* v = root.get(k);
  return v && v->kind == jval::STR ? v->s : std::string();
}

constexpr int kDigits = 9;  // jellyfish::file_header length prefix

}  // namespace

// --------------------------------------------------------------- serialisation

void qh_header_serialize(qh_file_header& h, std::string& out) {
  // keys in alphabetical order, one line, like jsoncpp's FastWriter
  std::string j = "{";
  jkv(j, "alignment", (uint64_t)8);
  jkv(j, "bits", (uint64_t)h.bits);
  jescape("cmdline", j); j += ":[";
  for (size_t i = 0; i < h.cmdline.size(); ++i) { if (i) j += ','; jescape(h.cmdline[i], j); }
  j += "],";
  jkv(j, "exe_path", h.exe_path);
  jkv(j, "format", h.format);
  jkv(j, "hostname", h.hostname);
  jkv(j, "key_bytes", h.key_bytes);
  jkv(j, "key_len", (uint64_t)h.key_len);
  const bool jf = h.layout == QH_LAYOUT_JF;
  if (jf) {  // RectangularBinaryMatrix as the hash's "matrix1" 
    jescape("matrix1", j); j += ":{";
    jkv(j, "c", (uint64_t)h.matrix_c);
    jescape("columns", j); j += ":[";
    for (size_t i = 0; i < h.matrix_columns.size(); ++i) { if (i) j += ','; j += std::to_string(h.matrix_columns[i]); }
    j += "],";
    jkv(j, "r", (uint64_t)h.matrix_r, false);
    j += "},";
  }
  jkv(j, "max_reprobe", (uint64_t)h.max_reprobe);
  jkv(j, "pwd", h.pwd);
  if (!jf) {
    jescape("qhip", j); j += ":{";
    jkv(j, "hash", h.hash);
    jkv(j, "layout", h.layout);
    jkv(j, "probe_bits", (uint64_t)h.probe_bits);
    jkv(j, "shard_bits", (uint64_t)h.shard_bits);
    jkv(j, "window_bits", (uint64_t)h.window_bits, false);
    j += "},";
  }
  jescape("reprobes", j); j += ":[";
  if (jf) {
    for (size_t i = 0; i < h.reprobes.size(); ++i) { if (i) j += ','; j += std::to_string(h.reprobes[i]); }
  } else {  // linear probing inside a window: offsets 0..max_reprobe
    for (uint32_t i = 0; i <= h.max_reprobe; ++i) { if (i) j += ','; j += std::to_string(i); }
  }
  j += "],";
  jkv(j, "size", h.size);
  jkv(j, "time", h.time);
  jkv(j, "val_len", (uint64_t)h.val_len);
  jkv(j, "value_bytes", h.value_bytes, false);
  j += "}\n";
  size_t pad = (8 - (kDigits + j.size()) % 8) % 8;
  const size_t hlen = j.size() + pad;
  char pre[16];
  snprintf(pre, sizeof pre, "%0*zu", kDigits, hlen);
  out.assign(pre, kDigits);
  out += j;
  out.append(pad, '\0');
  h.offset = out.size();
}

bool qh_header_parse(const char* data, size_t len, qh_file_header& h, std::string& err) {
  if (len < (size_t)kDigits) { err = "file shorter than the header length prefix"; return false; }
  uint64_t hlen = 0;
  for (int i = 0; i < kDigits; ++i) {
    if (data[i] < '0' || data[i] > '9') { err = "header length prefix is not numeric"; return false; }
    hlen = hlen * 10 + (uint64_t)(data[i] - '0');
  }
  if (kDigits + hlen > len) { err = "truncated header"; return false; }
  jparser jp{data + kDigits, data + kDigits + hlen, {}};
  jval root;
  if (!jp.value(root) || root.kind != jval::OBJ) { err = "header JSON: " + (jp.err.empty() ? std::string("not an object") : jp.err); return false; }
  h = qh_file_header();
  h.offset = kDigits + hlen;
  h.format = js(root, "format");
  h.size = ju(root, "size");
  h.key_len = (uint32_t)ju(root, "key_len");
  h.val_len = (uint32_t)ju(root, "val_len");
  h.max_reprobe = (uint32_t)ju(root, "max_reprobe");
  h.bits = (uint32_t)ju(root, "bits");
  h.key_bytes = ju(root, "key_bytes");
  h.value_bytes = ju(root, "value_bytes");
  h.hostname = js(root, "hostname"); h.pwd = js(root, "pwd"); h.time = js(root, "time"); h.exe_path = js(root, "exe_path");
  if (const jval* c = root.get("cmdline"))
    for (auto& x : c->a) if (x.kind == jval::STR) h.cmdline.push_back(x.s);
  if (const jval* q = root.get("qhip")) {
    h.layout = js(*q, "layout"); h.hash = js(*q, "hash");
    h.window_bits = (uint32_t)ju(*q, "window_bits");
    h.probe_bits = (uint32_t)ju(*q, "probe_bits");
    h.shard_bits = (uint32_t)ju(*q, "shard_bits");
  } else if (const jval* m = root.get("matrix1")) {
    h.layout = QH_LAYOUT_JF;
    h.matrix_r = (uint32_t)ju(*m, "r");
    h.matrix_c = (uint32_t)ju(*m, "c");
    if (const jval* c = m->get("columns"))
      for (auto& x : c->a) if (x.kind == jval::NUM) h.matrix_columns.push_back(x.u);
    if (const jval* r = root.get("reprobes"))
      for (auto& x : r->a) if (x.kind == jval::NUM) h.reprobes.push_back(x.u);
  }
  return true;
}

bool qh_header_read_file(const char* path, qh_file_header& h, std::string& err) {
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("Can't open '") + path + "' for reading"; return false; }  // mer_database.hpp:259-260
  char pre[kDigits];
  bool ok = fread(pre, 1, kDigits, f) == (size_t)kDigits;
  uint64_t hlen = 0;
  for (int i = 0; ok && i < kDigits; ++i) { ok = pre[i] >= '0' && pre[i] <= '9'; hlen = hlen * 10 + (uint64_t)(pre[i] - '0'); }
  std::string buf;
  if (ok && hlen < (1u << 26)) {
    buf.assign(pre, kDigits);
    buf.resize(kDigits + hlen);
    ok = fread(&buf[kDigits], 1, hlen, f) == hlen;
  } else ok = false;
  fclose(f);
  if (!ok) { err = std::string("Can't parse header of file '") + path + "'"; return false; }  // :262-263
  if (!qh_header_parse(buf.data(), buf.size(), h, err)) { err = std::string("Can't parse header of file '") + path + "': " + err; return false; }
  return true;
}

void qh_header_fill_standard(qh_file_header& h, int argc, const char* const* argv) {
  char host[256] = {0};
  if (gethostname(host, sizeof host - 1) == 0) h.hostname = host;
  char cwd[PATH_MAX];
  if (getcwd(cwd, sizeof cwd)) h.pwd = cwd;
  time_t t = time(nullptr);
  char tb[64];
  struct tm tmv;
  localtime_r(&t, &tmv);
  strftime(tb, sizeof tb, "%a %b %e %H:%M:%S %Y", &tmv);
  h.time = tb;
  char exe[PATH_MAX];
  ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
  if (n > 0) { exe[n] = 0; h.exe_path = exe; }
  h.cmdline.clear();
  for (int i = 0; i < argc; ++i) h.cmdline.push_back(argv && argv[i] ? argv[i] : "");
}

void qh_header_from_geom(const qh_geom& g, qh_file_header& h) {
  h.format = "binary/quorum_db";
  h.size = 1ull << (g.lsize + g.shard_bits);
  h.key_len = 2 * g.k;
  h.val_len = 0;
  h.max_reprobe = g.max_reprobe;
  h.bits = g.bits;
  h.key_bytes = h.size * 8;
  h.value_bytes = qh_vals_bytes(h.size, g.bits + 1);
  h.layout = g.wide ? QH_LAYOUT_WIDE : "slot64/1";
  h.hash = "feistel3-core/2";
  h.window_bits = g.wbits;
  h.probe_bits = g.pbits;
  h.shard_bits = g.shard_bits;
}

bool qh_geom_from_header(const qh_file_header& h, qh_geom& g, std::string& err) {
  if (h.format != "binary/quorum_db") { err = "Wrong type '" + h.format + "'"; return false; }  // mer_database.hpp:264-265
  if ((h.layout != "slot64/1" && h.layout != QH_LAYOUT_WIDE) || h.hash != "feistel3-core/2") {
    err = "keys image layout '" + h.layout + "'/hash '" + h.hash + "' is not the native slot64/1 + feistel3-core/2 (upstream Jellyfish images are not readable)";
    return false;
  }
  if (h.size == 0 || (h.size & (h.size - 1))) { err = "size is not a power of two"; return false; }
  if (h.shard_bits > 8 || (h.size >> h.shard_bits) == 0) { err = "invalid shard_bits"; return false; }
  // the file holds all shards back to back
  if (!qh_make_geom(h.key_len / 2, h.bits, h.size >> h.shard_bits, 0, h.max_reprobe, (int)h.shard_bits, 0,
                    (int)h.window_bits, g, err)) return false;
  g.all_shards = 1;
  if (qh_table_slots(g) != h.size) { err = "size below the minimum table size"; return false; }
  if ((uint32_t)g.pbits != h.probe_bits) { err = "probe_bits mismatch"; return false; }
  if (h.key_bytes != h.size * 8) { err = "key_bytes does not match size"; return false; }
  if (h.value_bytes != qh_vals_bytes(h.size, g.bits + 1)) { err = "value_bytes does not match size/bits"; return false; }
  if ((h.layout == QH_LAYOUT_WIDE) != (g.wide != 0)) { err = "keys image layout '" + h.layout + "' does not match the geometry"; return false; }
  return true;
}

void qh_vals_pack(const uint64_t* slots, uint64_t n, uint64_t first, int vbits, uint64_t* image) {
  const uint64_t vpw = 64 / vbits, vmask = qh_low_mask(vbits);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t v = slots[i] & vmask;
    if (!v) continue;
    const uint64_t gi = first + i;
    image[gi / vpw] |= v << ((gi % vpw) * vbits);
  }
}
