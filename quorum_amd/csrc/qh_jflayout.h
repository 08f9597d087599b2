// qh_jflayout.h -- the keys image in the layout of jellyfish::large_hash::array, isolated in one module.
//
// PARITY UNPINNED, and said so wherever this is used. Upstream's hash_with_quality::write (src/mer_database.hpp:115-126)
// dumps the raw memory of a Jellyfish 2.x large_hash::array; database_query (:273-277) wraps such an image. Jellyfish is
// not in the reference tree, not installed here and not fetchable, and the reference holds no database file, so nothing
// this module writes can be checked against an upstream reader or writer. It follows what SURVEY.md section 8(c) records
// of the library's published design:
//   * size 2^lsize; position of a key = M x key over GF(2), M an lsize x key_len binary matrix kept as key_len 64-bit
//     columns (key bit i selects column key_len-1-i), whose part acting on the LOW lsize key bits is invertible;
//   * reprobe offsets i(i+1)/2, max_reprobe lowered until the largest offset is below the size;
//   * an entry stores only  (key >> lsize) << reprobe_len | (reprobe + 1)  (0 = empty) behind one "large" flag bit, the
//     low key bits follow from the entry's position: key_low = Minv x (position ^ M_high x key_high);
//   * entries are bit-packed back to back; the values are a jellyfish::atomic_bits_array beside it, one (bits+1)-wide
//     field per entry, floor(64/(bits+1)) fields per word;
//   * header keys size, key_len, val_len, max_reprobe, reprobes[], matrix1{r, c, columns[]}.
// KNOWN DEVIATION: Jellyfish reserves a "set" bit in a word that holds the tail of an entry straddling two words (its
// lock-free writers need it); the exact rule could not be recalled, so entries are packed here without such bits. An
// upstream reader will therefore NOT decode this image as it stands: the module exists so that the one place to fix is
// known and self-tested (write -> read -> identical content) for the day a Jellyfish install is at hand.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

struct qh_jf_geom {
  int lsize = 0;          // log2(entries)
  int key_len = 0;        // 2k
  int reprobe_len = 0;    // bits of the reprobe + 1 field
  int entry_bits = 0;     // 1 (large flag) + key_len - lsize + reprobe_len
  uint32_t max_reprobe = 0;
  std::vector<uint64_t> reprobes;   // i (i + 1) / 2
  std::vector<uint64_t> columns;    // the matrix: key_len columns of lsize bits
  uint64_t key_bytes() const { return ((((uint64_t)1 << lsize) * (uint64_t)entry_bits + 63) / 64) * 8; }
};

// geometry for 2^lsize entries of key_len-bit keys; the matrix is drawn from `seed`. false: unsupported (key_len <= lsize)
bool qh_jf_make_geom(int lsize, int key_len, int reprobe_limit, uint64_t seed, qh_jf_geom& g, std::string& err);
// checks a geometry read from a header (matrix invertible on the low key bits, ...)
bool qh_jf_check_geom(qh_jf_geom& g, std::string& err);

// (mer, value) pairs -> keys image + per-entry values (values[id], 0 for empty entries). false: a key found no place.
bool qh_jf_build(const qh_jf_geom& g, const uint64_t* mers, const uint64_t* vals, uint64_t n,
                 std::vector<uint64_t>& keys_image, std::vector<uint64_t>& values);
// keys image -> the (mer, entry id) of every occupied entry
bool qh_jf_scan(const qh_jf_geom& g, const uint64_t* keys_image, std::vector<uint64_t>& mers, std::vector<uint64_t>& ids,
                std::string& err);

// --- whole files (header of qh_dbfile.h with "matrix1" instead of "qhip") --------------------------------------------
// Writes n distinct (mer, value) pairs as a database file in this layout: a table of max(size, 2^ceil) entries, doubled
// (a few times at most) when a key finds no place within the reprobe limit.
bool qh_jf_write_pairs(const char* path, int k, int bits, uint64_t size, int reprobe_limit, const uint64_t* mers,
                       const uint64_t* vals, uint64_t n, int argc, const char* const* argv, std::string& err);
// Reads such a file back: every occupied entry's mer and value, in entry order; k and bits from its header.
bool qh_jf_read_pairs(const char* path, int* k, int* bits, uint64_t* size, std::vector<uint64_t>& mers,
                      std::vector<uint64_t>& vals, std::string& err);
