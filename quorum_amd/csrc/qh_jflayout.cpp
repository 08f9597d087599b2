// qh_jflayout.cpp -- see qh_jflayout.h: the Jellyfish large_hash::array keys image as recalled (PARITY UNPINNED).
#include <new>
#include "qh_jflayout.h"
#include "qh_dbfile.h"
#include "qh_host.h"

#include <errno.h>
#include <stdio.h>
#include <string.h>

namespace {

uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

int bitsize(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

// a GF(2) matrix as byte-sliced tables: out = XOR over the set bits i of x of col[i]
struct gf2_lut {
  uint64_t t[8][256];
  void build(const uint64_t* col, int nbits) {
    memset(t, 0, sizeof t);
    for (int b = 0; b < 8; ++b)
      for (int v = 1; v < 256; ++v) {
        const int low = __builtin_ctz(v), i = 8 * b + low;
        t[b][v] = t[b][v & (v - 1)] ^ (i < nbits ? col[i] : 0);
      }
  }
  uint64_t mul(uint64_t x) const {
    uint64_t r = 0;
    for (int b = 0; b < 8; ++b) r ^= t[b][(x >> (8 * b)) & 0xFF];
    return r;
  }
};

// column i of the r x r block that acts on key bit i (i < r): columns[c - 1 - i]
bool invert_low_block(const qh_jf_geom& g, std::vector<uint64_t>& inv_cols) {
  const int r = g.lsize, c = g.key_len;
  // rows of [A | I]
  std::vector<uint64_t> a(r, 0), id(r, 0);
  for (int t = 0; t < r; ++t) {
    for (int i = 0; i < r; ++i) a[t] |= ((g.columns[c - 1 - i] >> t) & 1ull) << i;
    id[t] = 1ull << t;
  }
  for (int col = 0; col < r; ++col) {
    int piv = -1;
    for (int t = col; t < r; ++t) if ((a[t] >> col) & 1ull) { piv = t; break; }
    if (piv < 0) return false;
    std::swap(a[piv], a[col]);
    std::swap(id[piv], id[col]);
    for (int t = 0; t < r; ++t)
      if (t != col && ((a[t] >> col) & 1ull)) { a[t] ^= a[col]; id[t] ^= id[col]; }
  }
  // id now holds the rows of A^-1; as columns: inv_cols[j] bit t = id[t] bit j
  inv_cols.assign(r, 0);
  for (int t = 0; t < r; ++t)
    for (int j = 0; j < r; ++j) inv_cols[j] |= ((id[t] >> j) & 1ull) << t;
  return true;
}

struct packed {
  uint64_t* w;
  int ebits;
  uint64_t get(uint64_t e) const {
    const uint64_t bit = e * (uint64_t)ebits, wi = bit >> 6;
    const int off = (int)(bit & 63);
    uint64_t v = w[wi] >> off;
    if (off + ebits > 64) v |= w[wi + 1] << (64 - off);
    return ebits == 64 ? v : v & ((1ull << ebits) - 1);
  }
  void set(uint64_t e, uint64_t v) {
    const uint64_t bit = e * (uint64_t)ebits, wi = bit >> 6;
    const int off = (int)(bit & 63);
    w[wi] |= v << off;
    if (off + ebits > 64) w[wi + 1] |= v >> (64 - off);
  }
};

// times(key): key bit i selects columns[c - 1 - i]
void key_columns(const qh_jf_geom& g, std::vector<uint64_t>& by_bit) {
  by_bit.resize(g.key_len);
  for (int i = 0; i < g.key_len; ++i) by_bit[i] = g.columns[g.key_len - 1 - i];
}

}  // namespace

bool qh_jf_check_geom(qh_jf_geom& g, std::string& err) {
  if (g.lsize < 1 || g.lsize > 48 || g.key_len < 2 || g.key_len > 62) { err = "jf layout: unsupported size or key length"; return false; }
  if (g.key_len <= g.lsize) { err = "jf layout: the table has more entries than there are keys (key_len <= log2 size)"; return false; }
  if ((int)g.columns.size() != g.key_len) { err = "jf layout: the matrix does not have key_len columns"; return false; }
  const uint64_t size = 1ull << g.lsize;
  // (max_reprobe comes from a file header: checked before anything is sized by it)
  if (g.max_reprobe && (uint64_t)g.max_reprobe * ((uint64_t)g.max_reprobe + 1) / 2 >= size) { err = "jf layout: reprobe offsets reach past the table"; return false; }
  g.reprobes.resize((size_t)g.max_reprobe + 1);
  for (uint64_t i = 0; i <= g.max_reprobe; ++i) g.reprobes[i] = i * (i + 1) / 2;
  g.reprobe_len = bitsize((uint64_t)g.max_reprobe + 1);
  g.entry_bits = 1 + g.key_len - g.lsize + g.reprobe_len;
  if (g.entry_bits > 64) { err = "jf layout: an entry of more than 64 bits (tiny table for this mer length)"; return false; }
  for (auto& c : g.columns) c &= size - 1;
  std::vector<uint64_t> inv;
  if (!invert_low_block(g, inv)) { err = "jf layout: the matrix is not invertible on the low key bits"; return false; }
  return true;
}

bool qh_jf_make_geom(int lsize, int key_len, int reprobe_limit, uint64_t seed, qh_jf_geom& g, std::string& err) {
  g = qh_jf_geom();
  g.lsize = lsize;
  g.key_len = key_len;
  if (lsize < 1 || lsize > 48 || key_len <= lsize || key_len > 62) { err = "jf layout: unsupported size / key length"; return false; }
  const uint64_t size = 1ull << lsize;
  uint64_t mr = reprobe_limit < 0 ? 0 : (uint64_t)reprobe_limit;
  while (mr && mr * (mr + 1) / 2 >= size) --mr;   // large_hash::array lowers max_reprobe until the offsets fit
  g.max_reprobe = (uint32_t)mr;
  uint64_t s = seed ? seed : 0x5EEDull;
  for (int tries = 0; tries < 1000; ++tries) {
    g.columns.resize(key_len);
    for (auto& c : g.columns) c = splitmix(s) & (size - 1);
    std::string e;
    if (qh_jf_check_geom(g, e)) return true;
    if (e.find("invertible") == std::string::npos) { err = e; return false; }
  }
  err = "jf layout: no invertible matrix found";
  return false;
}

bool qh_jf_build(const qh_jf_geom& g, const uint64_t* mers, const uint64_t* vals, uint64_t n,
                 std::vector<uint64_t>& keys_image, std::vector<uint64_t>& values) {
  const uint64_t size = 1ull << g.lsize, mask = size - 1;
  keys_image.assign(g.key_bytes() / 8 + 1, 0);
  values.assign(size, 0);
  std::vector<uint64_t> by_bit;
  key_columns(g, by_bit);
  gf2_lut M;
  M.build(by_bit.data(), g.key_len);
  packed P = {keys_image.data(), g.entry_bits};
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t key = mers[i], pos = M.mul(key);
    bool placed = false;
    for (uint32_t rp = 0; rp <= g.max_reprobe && !placed; ++rp) {
      const uint64_t id = (pos + g.reprobes[rp]) & mask;
      if (P.get(id) == 0) {
        // [large flag = 0][(key >> lsize) << reprobe_len | reprobe + 1]
        P.set(id, ((((key >> g.lsize) << g.reprobe_len) | (uint64_t)(rp + 1)) << 1));
        values[id] = vals[i];
        placed = true;
      }
    }
    if (!placed) return false;
  }
  keys_image.resize(g.key_bytes() / 8);
  return true;
}

bool qh_jf_scan(const qh_jf_geom& g, const uint64_t* keys_image, std::vector<uint64_t>& mers, std::vector<uint64_t>& ids,
                std::string& err) {
  const uint64_t size = 1ull << g.lsize, mask = size - 1;
  std::vector<uint64_t> by_bit, inv;
  key_columns(g, by_bit);
  if (!invert_low_block(g, inv)) { err = "jf layout: the matrix is not invertible on the low key bits"; return false; }
  // contribution of the stored high bits: key bit lsize + j selects by_bit[lsize + j]
  gf2_lut H, I;
  H.build(by_bit.data() + g.lsize, g.key_len - g.lsize);
  I.build(inv.data(), g.lsize);
  // the image is read one word past a straddling last entry at most: copy with a guard word
  std::vector<uint64_t> img(keys_image, keys_image + g.key_bytes() / 8);
  img.push_back(0);
  packed P = {img.data(), g.entry_bits};
  const uint64_t rmask = (1ull << g.reprobe_len) - 1;
  mers.clear();
  ids.clear();
  for (uint64_t id = 0; id < size; ++id) {
    const uint64_t e = P.get(id);
    if (!e) continue;
    if (e & 1ull) { err = "jf layout: an entry with the large flag set (not written by this library)"; return false; }
    const uint64_t f = e >> 1, rp1 = f & rmask;
    if (rp1 == 0 || rp1 - 1 > g.max_reprobe) { err = "jf layout: invalid reprobe field"; return false; }
    const uint64_t high = f >> g.reprobe_len;
    const uint64_t pos = (id - g.reprobes[rp1 - 1]) & mask;
    const uint64_t low = I.mul(pos ^ H.mul(high));
    mers.push_back((high << g.lsize) | low);
    ids.push_back(id);
  }
  return true;
}

// ------------------------------------------------------------------------------------------------- whole files

bool qh_jf_write_pairs(const char* path, int k, int bits, uint64_t size, int reprobe_limit, const uint64_t* mers,
                       const uint64_t* vals, uint64_t n, int argc, const char* const* argv, std::string& err) {
  if (k < 1 || k > 31 || bits < 1 || bits > 31) { err = "jf layout: invalid mer length or bits"; return false; }
  int lsize = 1;
  while (lsize < 48 && (1ull << lsize) < size) ++lsize;
  while (lsize < 48 && (1ull << lsize) < n) ++lsize;
  const int vbits = bits + 1;
  const uint64_t kmask = 2 * k == 64 ? ~0ull : (1ull << (2 * k)) - 1, vmask = (1ull << vbits) - 1;
  for (uint64_t i = 0; i < n; ++i)
    if ((mers[i] & ~kmask) || (vals[i] & ~vmask) || !vals[i]) { err = "jf layout: a mer or value does not fit its field"; return false; }
  qh_jf_geom g;
  std::vector<uint64_t> keys, values;
  for (int tries = 0;; ++tries, ++lsize) {
    if (!qh_jf_make_geom(lsize, 2 * k, reprobe_limit, 0x9E3779B9ull + (uint64_t)lsize, g, err)) return false;
    if (qh_jf_build(g, mers, vals, n, keys, values)) break;
    if (tries == 6) { err = "Hash is full"; return false; }
  }
  qh_file_header fh;
  fh.format = "binary/quorum_db";
  fh.size = 1ull << g.lsize;
  fh.key_len = (uint32_t)(2 * k);
  fh.val_len = 0;
  fh.max_reprobe = g.max_reprobe;
  fh.bits = (uint32_t)bits;
  fh.key_bytes = g.key_bytes();
  fh.value_bytes = qh_vals_bytes(fh.size, vbits);
  fh.layout = QH_LAYOUT_JF;
  fh.matrix_r = (uint32_t)g.lsize;
  fh.matrix_c = (uint32_t)g.key_len;
  fh.matrix_columns = g.columns;
  fh.reprobes = g.reprobes;
  qh_header_fill_standard(fh, argc, argv);
  std::string hdr;
  qh_header_serialize(fh, hdr);
  std::vector<uint64_t> vimg(fh.value_bytes / 8, 0);
  qh_vals_pack(values.data(), fh.size, 0, vbits, vimg.data());
  FILE* f = fopen(path, "wb");
  if (!f) { err = std::string("Failed to open output file '") + path + "'."; return false; }
  bool ok = fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
  ok = ok && fwrite(keys.data(), 1, fh.key_bytes, f) == fh.key_bytes;
  ok = ok && fwrite(vimg.data(), 1, fh.value_bytes, f) == fh.value_bytes;
  if (fclose(f) != 0) ok = false;
  if (!ok) { err = std::string("Failed to write output file '") + path + "': " + strerror(errno); return false; }
  return true;
}

bool qh_jf_read_pairs(const char* path, int* k, int* bits, uint64_t* size, std::vector<uint64_t>& mers,
                      std::vector<uint64_t>& vals, std::string& err) {
  qh_file_header fh;
  if (!qh_header_read_file(path, fh, err)) return false;
  if (fh.format != "binary/quorum_db") { err = "Wrong type '" + fh.format + "'"; return false; }
  if (fh.layout != QH_LAYOUT_JF) { err = "the file is not in the jellyfish layout"; return false; }
  if (fh.size == 0 || (fh.size & (fh.size - 1)) || fh.key_len < 2 || fh.key_len > 62 || (fh.key_len & 1) || fh.bits < 1 || fh.bits > 31) {
    err = "invalid size, key_len or bits"; return false;
  }
  qh_jf_geom g;
  while ((1ull << g.lsize) < fh.size) ++g.lsize;
  g.key_len = (int)fh.key_len;
  g.max_reprobe = fh.max_reprobe;
  g.columns = fh.matrix_columns;
  if (fh.matrix_r != (uint32_t)g.lsize || fh.matrix_c != fh.key_len) { err = "matrix1 does not match size / key_len"; return false; }
  const std::vector<uint64_t> file_reprobes = fh.reprobes;
  if (!qh_jf_check_geom(g, err)) return false;
  if (file_reprobes != g.reprobes) { err = "reprobes[] are not the quadratic offsets i(i+1)/2"; return false; }
  const int vbits = (int)fh.bits + 1;
  if (fh.key_bytes != g.key_bytes()) { err = "key_bytes does not match size / key_len / max_reprobe"; return false; }
  if (fh.value_bytes != qh_vals_bytes(fh.size, vbits)) { err = "value_bytes does not match size/bits"; return false; }
  FILE* f = fopen(path, "rb");
  if (!f) { err = std::string("Can't open '") + path + "' for reading"; return false; }
  std::vector<uint64_t> keys(fh.key_bytes / 8), vimg(fh.value_bytes / 8);
  bool ok = fseeko(f, (off_t)fh.offset, SEEK_SET) == 0;
  ok = ok && fread(keys.data(), 1, fh.key_bytes, f) == fh.key_bytes;
  ok = ok && fread(vimg.data(), 1, fh.value_bytes, f) == fh.value_bytes;
  fclose(f);
  if (!ok) { err = std::string("Failed to read in file '") + path + "'"; return false; }
  std::vector<uint64_t> ids;
  if (!qh_jf_scan(g, keys.data(), mers, ids, err)) return false;
  const uint64_t vpw = 64 / (uint64_t)vbits, vmask = (1ull << vbits) - 1;
  vals.resize(ids.size());
  for (size_t i = 0; i < ids.size(); ++i) vals[i] = (vimg[ids[i] / vpw] >> ((ids[i] % vpw) * vbits)) & vmask;
  if (k) *k = (int)fh.key_len / 2;
  if (bits) *bits = (int)fh.bits;
  if (size) *size = fh.size;
  return true;
}

// host-only C ABI over the two (include/quorum_hip.h)
extern "C" int qh_jf_write_file(const char* path, int k, int bits, uint64_t size, int reprobe_limit, const uint64_t* mers,
                                const uint64_t* vals, uint64_t n, int argc, const char* const* argv) {
  if (!path || (n && (!mers || !vals))) return qh_fail(QH_EBADARG, "null argument");
  std::string err;
  try {
    if (!qh_jf_write_pairs(path, k, bits, size, reprobe_limit, mers, vals, n, argc, argv, err))
      return qh_fail(err == "Hash is full" ? QH_EFULL : err.find("output file") != std::string::npos ? QH_EIO : QH_EBADARG, "%s", err.c_str());
  } catch (const std::bad_alloc&) {  // no exception crosses the C ABI
    return qh_fail(QH_ENOMEM, "out of host memory building the image for '%s'", path);
  }
  return QH_OK;
}

extern "C" int qh_jf_read_file(const char* path, int* k, int* bits, uint64_t* size, uint64_t* mers, uint64_t* vals,
                               uint64_t cap, uint64_t* n) {
  if (!path || !n) return qh_fail(QH_EBADARG, "null argument");
  std::string err;
  std::vector<uint64_t> m, v;
  try {
    if (!qh_jf_read_pairs(path, k, bits, size, m, v, err)) return qh_fail(QH_EFORMAT, "%s for file '%s'", err.c_str(), path);
  } catch (const std::bad_alloc&) {  // a header that asks for more than the host has
    return qh_fail(QH_ENOMEM, "out of host memory reading '%s'", path);
  }
  *n = m.size();
  if (m.size() > cap) return mers || vals ? qh_fail(QH_EBADARG, "buffer too small: %zu entries", m.size()) : QH_OK;
  if (mers) memcpy(mers, m.data(), m.size() * 8);
  if (vals) memcpy(vals, v.data(), v.size() * 8);
  return QH_OK;
}
