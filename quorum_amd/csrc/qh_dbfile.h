// qh_dbfile.h -- the on-disk "binary/quorum_db" container.
//
// Replaces database_header + hash_with_quality::write (src/mer_database.hpp:43-63,
// 115-126) and the header part of database_query's constructor (:257-278).
//
// File = [9 ASCII digits: length L of what follows up to the data]
//        [JSON header][zero padding so that 9 + L is a multiple of 8]
//        [keys image : key_bytes][vals image : value_bytes]
// The invariants the reference's test pins are kept: offset % 8 == 0 and
// file length == offset + key_bytes + value_bytes (unit_tests/test_mer_database.cc:102-106),
// header.bits() and header.key_len() round-trip (:113-114).
//
// keys image : the native table, one little-endian 64-bit slot per entry
//              (layout in qh_spec.h). It is NOT Jellyfish's bit-packed
//              large_hash::array image: that layout cannot be verified offline
//               and the JSON says so in "qhip.layout". Behind a switch
//              (qh_build_write_jf, --jf-layout) the image is written in the recalled
//              Jellyfish layout instead (qh_jflayout.h, parity unpinned): such a header
//              has no "qhip" object and carries "matrix1" and the quadratic "reprobes".
// vals image : jellyfish::atomic_bits_array layout -- (bits+1)-wide fields
//              "count<<1|hq", floor(64/(bits+1)) per 64-bit word, entry i in word
//              i / vpw at shift (i % vpw) * (bits+1); what
//              compute_poisson_cutoff__ scans (src/error_correct_reads.cc:650-659).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "qh_spec.h"

#define QH_LAYOUT_JF "jellyfish/large_hash"   // internal name; never written to a file
// keys image of the wide form (qh_spec.h): slots hold [remainder | probe+1] only, every value lives in the vals image
// (large -b, or long mers in small tables: what would not fit one 64-bit word together)
#define QH_LAYOUT_WIDE "slot64-keys/1"

struct qh_file_header {
  std::string format;        // "binary/quorum_db"
  uint64_t size = 0;         // total slots (all shards)
  uint32_t key_len = 0;      // 2k
  uint32_t val_len = 0;      // 0, as upstream (mer_database.hpp:79)
  uint32_t max_reprobe = 0;
  uint32_t bits = 0;         // count bits (value field is bits+1 wide)
  uint64_t key_bytes = 0, value_bytes = 0;
  uint64_t offset = 0;       // start of the keys image
  // native layout parameters ("qhip" object)
  std::string layout;        // "slot64/1", or QH_LAYOUT_WIDE
  uint32_t window_bits = 0, probe_bits = 0, shard_bits = 0;
  std::string hash;          // "feistel3-core/2"
  // layout "jellyfish/large_hash" (qh_jflayout.h; no "qhip" object in the JSON): the matrix1 object and reprobes[]
  uint32_t matrix_r = 0, matrix_c = 0;
  std::vector<uint64_t> matrix_columns;
  std::vector<uint64_t> reprobes;
  std::vector<std::string> cmdline;
  std::string hostname, pwd, time, exe_path;
};

// Serialises the header (length prefix + JSON + padding) into `out`; sets h.offset.
void qh_header_serialize(qh_file_header& h, std::string& out);
// Parses a header from the first bytes of a file. Returns false and fills err on failure.
bool qh_header_parse(const char* data, size_t len, qh_file_header& h, std::string& err);
// reads the header of `path`
bool qh_header_read_file(const char* path, qh_file_header& h, std::string& err);

void qh_header_fill_standard(qh_file_header& h, int argc, const char* const* argv);

// geometry <-> header
void qh_header_from_geom(const qh_geom& g, qh_file_header& h);
bool qh_geom_from_header(const qh_file_header& h, qh_geom& g, std::string& err);

inline uint64_t qh_vals_bytes(uint64_t slots, int vbits) {
  const uint64_t vpw = 64 / vbits;
  return (slots + vpw - 1) / vpw * 8;
}
// packs the value fields of `n` slots (starting at global slot index first) into a
// zero-initialised atomic_bits_array image
void qh_vals_pack(const uint64_t* slots, uint64_t n, uint64_t first, int vbits, uint64_t* image);
