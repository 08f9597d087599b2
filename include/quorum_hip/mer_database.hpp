// quorum_hip/mer_database.hpp -- header-only C++ shim with the class shapes of the reference's
// src/mer_database.hpp (hash_with_quality :65-188, database_header :43-63, database_query
// :251-362) over the libquorum_hip C ABI (include/quorum_hip.h), plus the slice of
// jellyfish::mer_dna those classes are used with (k <= 31, one 64-bit word).
//
// Same names, argument meaning and error behaviour as upstream: add() returns false when the
// table is full and cannot grow, every thread calls done(), open/parse failures throw
// std::runtime_error. add() may be called concurrently (calls are batched under a mutex and
// handed to the GPU a million at a time); the const methods of database_query are thread-safe and run
// concurrently (the library gives every call a stream of its own; begin() / end() take their one snapshot of
// the table under std::call_once).
//
// One difference in WHEN a full table shows: upstream's add() returns false on the very call that cannot
// be placed (src/mer_database.hpp:97-99). Here a batch of up to 2^20 adds is placed at once, so add()
// returns false from the call that flushes the failing batch (and from every call after it), and a
// failure in the last, partial batch surfaces in done(), which then throws std::runtime_error("Hash is
// full") -- the exception create_database.cc:86-87 throws when add() fails -- so that a caller ported
// from upstream cannot miss it. write() flushes too and throws the same.
#ifndef QUORUM_HIP_MER_DATABASE_HPP
#define QUORUM_HIP_MER_DATABASE_HPP

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <istream>
#include <iterator>
#include <ostream>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../quorum_hip.h"

namespace quorum_hip {

// jellyfish::mer_dna as used at src/create_database.cc:57,73-86 and src/kmer.hpp:15-50:
// A=0 C=1 G=2 T=3 (either case), base 0 = the two low bits = the LAST character.
class mer_dna {
  uint64_t w_;
  static unsigned& k_ref() { static unsigned k = 0; return k; }

public:
  static unsigned k() { return k_ref(); }
  static unsigned k(unsigned nk) { std::swap(nk, k_ref()); return nk; }
  static int code(char c) {
    switch (c) {
      case 'A': case 'a': return 0;
      case 'C': case 'c': return 1;
      case 'G': case 'g': return 2;
      case 'T': case 't': return 3;
      case 'N': case 'n': return -1;
      case '\n': return -2;
      default: return -3;
    }
  }
  static bool not_dna(int c) { return c < 0; }
  static int complement(int c) { return 3 - c; }
  static char rev_code(int c) { return "ACGT"[c & 3]; }
  static uint64_t mask() { return k() >= 32 ? ~0ull : ((1ull << (2 * k())) - 1); }

  mer_dna() : w_(0) {}
  explicit mer_dna(uint64_t w) : w_(w & mask()) {}
  explicit mer_dna(const std::string& s) : w_(0) { *this = s; }
  mer_dna& operator=(const std::string& s) {
    if (s.size() < k()) throw std::length_error("Input string is to short");
    w_ = 0;
    for (unsigned i = 0; i < k(); ++i) {
      const int c = code(s[i]);
      if (c < 0) throw std::domain_error("Invalid character in mer");
      w_ = (w_ << 2) | (uint64_t)c;
    }
    return *this;
  }
  mer_dna& operator=(const char* s) { return *this = std::string(s); }

  uint64_t word() const { return w_; }
  int code(int i) const { return (int)((w_ >> (2 * i)) & 3); }
  void replace(unsigned i, int c) { w_ = (w_ & ~(3ull << (2 * i))) | ((uint64_t)(c & 3) << (2 * i)); }
  void shift_left(int c) { w_ = ((w_ << 2) | (uint64_t)(c & 3)) & mask(); }
  void shift_right(int c) { w_ = (w_ >> 2) | ((uint64_t)(c & 3) << (2 * (k() - 1))); }
  bool operator<(const mer_dna& o) const { return w_ < o.w_; }
  bool operator==(const mer_dna& o) const { return w_ == o.w_; }
  bool operator!=(const mer_dna& o) const { return w_ != o.w_; }
  mer_dna get_reverse_complement() const {
    uint64_t m = w_, r = 0;
    for (unsigned i = 0; i < k(); ++i) { r = (r << 2) | (3 - (m & 3)); m >>= 2; }
    return mer_dna(r);
  }
  void canonicalize() { const mer_dna r = get_reverse_complement(); if (r < *this) *this = r; }
  mer_dna get_canonical() const { mer_dna m(*this); m.canonicalize(); return m; }
  std::string to_str() const {
    std::string s(k(), 'A');
    for (unsigned i = 0; i < k(); ++i) s[i] = rev_code(code((int)(k() - 1 - i)));
    return s;
  }
};

// database_header (src/mer_database.hpp:43-63) -- read side
class database_header {
  qh_db_info info_;

public:
  database_header() { memset(&info_, 0, sizeof info_); }
  bool read(const char* path) { return qh_db_header_info(path, &info_) == QH_OK || info_.size != 0; }
  // bool read(std::istream&) (jellyfish::file_header::read as src/mer_database.hpp:261 calls it): parses the header
  // and leaves the stream at offset(), where the keys image starts
  bool read(std::istream& is) {
    char pre[9];
    if (!is.read(pre, 9)) return false;
    size_t hlen = 0;
    for (int i = 0; i < 9; ++i) {
      if (pre[i] < '0' || pre[i] > '9') return false;
      hlen = hlen * 10 + (size_t)(pre[i] - '0');
    }
    if (hlen >= ((size_t)1 << 26)) return false;  // as the library's own reader: no header is that long
    std::string buf(pre, 9);
    buf.resize(9 + hlen);
    if (hlen && !is.read(&buf[9], (std::streamsize)hlen)) return false;
    return qh_db_header_parse(buf.data(), buf.size(), &info_) == QH_OK || info_.size != 0;
  }
  uint32_t bits() const { return info_.bits; }
  size_t value_bytes() const { return info_.value_bytes; }
  size_t key_bytes() const { return info_.key_bytes; }
  size_t size() const { return info_.size; }
  unsigned key_len() const { return info_.key_len; }
  unsigned val_len() const { return info_.val_len; }
  unsigned max_reprobe() const { return info_.max_reprobe; }
  size_t offset() const { return info_.offset; }
  std::string format() const { return info_.format; }
  bool check_format() const { return format() == "binary/quorum_db"; }
};

// hash_with_quality (src/mer_database.hpp:65-188)
class hash_with_quality {
  qh_build* h_;
  std::mutex mu_;
  std::vector<uint64_t> mers_;
  std::vector<uint8_t> quals_;
  bool full_;
  static const size_t batch = 1u << 20;

  bool flush_locked() {
    if (mers_.empty()) return !full_;
    int rc = qh_build_add_mers(h_, mers_.data(), quals_.data(), mers_.size());
    if (rc == QH_OK) rc = qh_build_sync(h_);
    mers_.clear();
    quals_.clear();
    if (rc == QH_EFULL) full_ = true;
    else if (rc != QH_OK) throw std::runtime_error(qh_last_error());
    return !full_;
  }

public:
  hash_with_quality(size_t size, uint16_t key_len, int bits, uint16_t /*nb_threads*/, uint16_t reprobe_limit = 126,
                    int device = 0)
      : h_(0), full_(false) {
    if (qh_build_create(&h_, key_len / 2, bits, size, 0, reprobe_limit, device, 0, 0) != QH_OK)
      throw std::runtime_error(qh_last_error());
  }
  ~hash_with_quality() { qh_build_destroy(h_); }
  hash_with_quality(const hash_with_quality&) = delete;
  hash_with_quality& operator=(const hash_with_quality&) = delete;

  // bool add(const mer_dna& key, unsigned int quality) (:94); false = "Hash is full"
  bool add(const mer_dna& key, unsigned int quality) {
    std::lock_guard<std::mutex> lock(mu_);
    if (full_) return false;
    mers_.push_back(key.word());
    quals_.push_back(quality ? 1 : 0);
    return mers_.size() < batch || flush_locked();
  }
  // void done() (:128): every inserting thread calls it once. Places what is still batched; a table that cannot
  // take it throws (see the top of this file)
  void done() {
    std::lock_guard<std::mutex> lock(mu_);
    if (!flush_locked()) throw std::runtime_error("Hash is full");
  }
  // the C ABI writes by path
  void write(const char* path, int argc = 0, const char* const* argv = 0) {
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (!flush_locked()) throw std::runtime_error("Hash is full");
    }
    if (qh_build_write(h_, path, argc, argv) != QH_OK) throw std::runtime_error(qh_last_error());
  }
  // void write(std::ostream& os, database_header* header = 0) const (:115-126): header, keys image, values image.
  // The library serialises to a file (the table is copied out of HBM in pieces), so the stream form goes through a
  // temporary file in $TMPDIR; *header receives the fields of what was written (bits, key_bytes, value_bytes, offset...).
  void write(std::ostream& os, database_header* header = 0) {
    const char* dir = getenv("TMPDIR");
    std::string tmp = std::string(dir && *dir ? dir : "/tmp") + "/quorum_hip_db_XXXXXX";
    const int fd = mkstemp(&tmp[0]);
    if (fd < 0) throw std::runtime_error("Failed to create a temporary file for the database image");
    close(fd);
    try {
      write(tmp.c_str());
      if (header && !header->read(tmp.c_str())) throw std::runtime_error("Can't parse the header just written");
      FILE* f = fopen(tmp.c_str(), "rb");
      if (!f) throw std::runtime_error("Can't reopen the temporary database image");
      std::vector<char> buf(1 << 22);
      size_t got;
      while ((got = fread(buf.data(), 1, buf.size(), f)) > 0) os.write(buf.data(), (std::streamsize)got);
      fclose(f);
    } catch (...) {
      unlink(tmp.c_str());
      throw;
    }
    unlink(tmp.c_str());
    if (!os) throw std::runtime_error("Failed to write the database to the stream");
  }
  qh_build* handle() { return h_; }
};

// database_query (src/mer_database.hpp:251-362)
class database_query {
  qh_query* q_;
  database_header header_;

public:
  // database_query(const char* filename, bool no_mmap = false) (:270); throws on open/parse failure
  explicit database_query(const char* filename, bool /*no_mmap*/ = false, int device = 0) : q_(0) {
    // parse_header (:257-267): upstream builds the "Wrong type" error at :264-265 and forgets to throw it; throwing
    // is what it means to do, and what happens here, before anything of the file is loaded
    if (!header_.read(filename)) throw std::runtime_error(qh_last_error());
    if (!header_.check_format())
      throw std::runtime_error(std::string("Wrong type '") + header_.format() + "' for file '" + filename + "'");
    if (qh_query_open(&q_, filename, device) != QH_OK) throw std::runtime_error(qh_last_error());
  }
  ~database_query() { qh_query_close(q_); }
  database_query(const database_query&) = delete;
  database_query& operator=(const database_query&) = delete;

  const database_header& header() const { return header_; }

  // std::pair<uint64_t,int> operator[](const mer_dna&) const (:284)
  std::pair<uint64_t, int> operator[](const mer_dna& m) const {
    uint64_t key = m.word(), v = 0;
    if (qh_query_batch(q_, &key, 1, &v) != QH_OK) throw std::runtime_error(qh_last_error());
    return std::pair<uint64_t, int>(v >> 1, (int)(v & 1));
  }
  // uint64_t get_val(const mer_dna&) const (:296)
  uint64_t get_val(const mer_dna& m) const {
    const std::pair<uint64_t, int> v = (*this)[m];
    return v.second ? v.first : 0;
  }
  // batched forms: one GPU call for many mers
  void lookup(const std::vector<uint64_t>& canonical_mers, std::vector<uint64_t>& values) const {
    values.resize(canonical_mers.size());
    if (qh_query_batch(q_, canonical_mers.data(), canonical_mers.size(), values.data()) != QH_OK)
      throw std::runtime_error(qh_last_error());
  }

  // template<typename mer_type> int get_best_alternatives(mer_type& m, uint64_t counts[4], int& ucode,
  // int& level) const (:302-329). mer_type needs code(0), replace(0, x) and canonical() like
  // forward_mer / backward_mer (src/kmer.hpp:70-109). The four sibling lookups are one GPU call.
  template <typename mer_type>
  int get_best_alternatives(mer_type& m, uint64_t counts[4], int& ucode, int& level) const {
    int count = 0;
    memset(counts, '\0', sizeof(uint64_t) * 4);
    level = 0;
    const int ori_code = m.code(0);
    uint64_t keys[4], vals[4];
    for (int i = 0; i < 4; ++i) {
      m.replace(0, i);
      keys[i] = m.canonical().word();
    }
    m.replace(0, ori_code);  // Reset m to original value
    if (qh_query_batch(q_, keys, 4, vals) != QH_OK) throw std::runtime_error(qh_last_error());
    for (int i = 0; i < 4; ++i) {
      const uint64_t first = vals[i] >> 1;
      const int second = (int)(vals[i] & 1);
      if (first > 0) {
        if (second >= level) {
          if (second > level && count > 0) {
            for (int j = 0; j < i; ++j) counts[j] = 0;
            count = 0;
          }
          counts[i] = first;
          ucode = i;
          level = second;
          ++count;
        }
      }
    }
    return count;
  }

  // const_iterator / begin() / end() (:331-361): a snapshot of every (mer, (count, quality))
  typedef std::pair<mer_dna, std::pair<uint64_t, int> > value_type;
  typedef std::vector<value_type>::const_iterator const_iterator;
  const_iterator begin() const { load(); return content_.begin(); }
  const_iterator end() const { load(); return content_.end(); }

  // compute_poisson_cutoff__'s reduction over vals() (src/error_correct_reads.cc:650-659)
  void hq_stats(uint64_t& distinct, uint64_t& total) const {
    if (qh_query_stats(q_, &distinct, &total) != QH_OK) throw std::runtime_error(qh_last_error());
  }
  qh_query* handle() const { return q_; }

private:
  mutable std::vector<value_type> content_;
  mutable std::once_flag loaded_;
  void load() const { std::call_once(loaded_, [this]() { load_once(); }); }  // a throw leaves the flag unset
  void load_once() const {
    uint64_t n = 0;
    if (qh_query_dump(q_, 0, 0, 0, &n) != QH_OK) throw std::runtime_error(qh_last_error());
    std::vector<uint64_t> mers(n), vals(n);
    if (n && qh_query_dump(q_, mers.data(), vals.data(), n, &n) != QH_OK) throw std::runtime_error(qh_last_error());
    content_.reserve(n);
    for (uint64_t i = 0; i < n; ++i)
      content_.push_back(value_type(mer_dna(mers[i]), std::pair<uint64_t, int>(vals[i] >> 1, (int)(vals[i] & 1))));
  }
};

}  // namespace quorum_hip

#endif
