// The reference's own test, TEST_P(MerDatabase, WriteRead) (unit_tests/test_mer_database.cc:47-135),
// rewritten against the header-only shim include/quorum_hip/mer_database.hpp: ten threads insert
// hq2, hq2, hq1, lqhq(0), lqhq(1), hqlq(1), hqlq(0), lq1, lq2, lq2 concurrently, the database is
// written, read back with database_query and checked against the known answers
// (2,1) (1,1) (1,1) (1,1) (1,0) (2,0). k = 31 instead of upstream's 33 (one 64-bit word).
// Usage: test_mer_database_shim <param> <output file>   (param = 1, 10, 20 or 40 as upstream :138)
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <atomic>
#include <fstream>
#include <map>
#include <string>
#include <thread>

#include "../../include/quorum_hip/mer_database.hpp"

using namespace quorum_hip;

static int failures = 0;
#define EXPECT(cond)                                                             \
  do {                                                                           \
    if (!(cond)) {                                                               \
      if (++failures < 20) fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
    }                                                                            \
  } while (0)

static uint64_t rng_state = 0x853c49e6748fea9bull;
static unsigned random_bits2() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (unsigned)(rng_state >> 62);
}

static std::string generate_sequence(const size_t s) {
  std::string res(s, 'A');
  for (size_t i = 0; i < s; ++i) res[i] = mer_dna::rev_code(random_bits2());
  return res;
}

static void insert_sequence(hash_with_quality* hash, const std::string& seq, const unsigned int quality) {
  mer_dna m;
  for (size_t i = 0; i <= seq.size() - mer_dna::k(); ++i) {
    m = seq.substr(i, mer_dna::k());
    if (!hash->add(m, quality)) { fprintf(stderr, "Hash is full\n"); exit(2); }
  }
  hash->done();
}

typedef std::map<uint64_t, std::pair<uint64_t, int> > mer_map_t;

static void test_sequence(const database_query& hash, const std::string& seq, const uint64_t count, const int quality,
                          const char* msg, mer_map_t& mer_map) {
  // one batched lookup per class; a few through operator[] itself
  std::vector<uint64_t> keys, vals;
  mer_dna m;
  for (size_t i = 0; i <= seq.size() - mer_dna::k(); ++i) {
    m = seq.substr(i, mer_dna::k());
    keys.push_back(m.word());
  }
  hash.lookup(keys, vals);
  size_t bad = 0;
  for (size_t i = 0; i < keys.size(); ++i) {
    if ((vals[i] >> 1) != count || (int)(vals[i] & 1) != quality) ++bad;
    mer_map[keys[i]] = std::make_pair(vals[i] >> 1, (int)(vals[i] & 1));
  }
  if (bad) fprintf(stderr, "%s: %zu of %zu mers wrong\n", msg, bad, keys.size());
  EXPECT(bad == 0);
  for (size_t i = 0; i < keys.size(); i += keys.size() / 5 + 1) {
    const std::pair<uint64_t, int> res = hash[mer_dna(keys[i])];
    EXPECT(res.first == count);
    EXPECT(res.second == quality);
    EXPECT(hash.get_val(mer_dna(keys[i])) == (quality ? count : 0));
  }
}

// the forward_mer shape get_best_alternatives is instantiated with (src/kmer.hpp:70-89).
// canonical() is the identity here: like upstream's, this test stores the mers as they come,
// not canonicalized.
struct fwd_mer {
  mer_dna f;
  int code(int i) const { return f.code(i); }
  void replace(int i, int x) { f.replace(i, x); }
  mer_dna canonical() const { return f; }
};

int main(int argc, char* argv[]) {
  if (argc != 3) { fprintf(stderr, "usage: %s param file\n", argv[0]); return 1; }
  const int param = atoi(argv[1]);
  const char* path = argv[2];
  static const size_t sequence_len = 100000;
  static const unsigned int bits = 4;
  std::string hq2 = generate_sequence(sequence_len), hq1 = generate_sequence(sequence_len),
              lqhq = generate_sequence(sequence_len), hqlq = generate_sequence(sequence_len),
              lq1 = generate_sequence(sequence_len), lq2 = generate_sequence(sequence_len);
  mer_dna::k(31);
  {
    hash_with_quality database(param * sequence_len, mer_dna::k() * 2, bits, 10);
    {
      std::thread th_hq2_1(insert_sequence, &database, hq2, 1);
      std::thread th_hq2_2(insert_sequence, &database, hq2, 1);
      std::thread th_hq1_1(insert_sequence, &database, hq1, 1);
      std::thread th_lqhq_0(insert_sequence, &database, lqhq, 0);
      std::thread th_lqhq_1(insert_sequence, &database, lqhq, 1);
      std::thread th_lqhq_2(insert_sequence, &database, hqlq, 1);
      std::thread th_lqhq_3(insert_sequence, &database, hqlq, 0);
      std::thread th_lq1_1(insert_sequence, &database, lq1, 0);
      std::thread th_lq2_1(insert_sequence, &database, lq2, 0);
      std::thread th_lq2_2(insert_sequence, &database, lq2, 0);
      th_hq2_1.join(); th_hq2_2.join(); th_hq1_1.join(); th_lqhq_0.join(); th_lqhq_1.join();
      th_lqhq_2.join(); th_lqhq_3.join(); th_lq1_1.join(); th_lq2_1.join(); th_lq2_2.join();
    }
    // the stream form, as unit_tests/test_mer_database.cc:97-101: write(os, &header) fills the caller's header
    std::ofstream file(path);
    database_header wheader;
    database.write(file, &wheader);
    file.close();
    EXPECT(file.good());
    EXPECT(wheader.bits() == bits);
    EXPECT(wheader.offset() % sizeof(uint64_t) == 0);
  }
  database_header header;
  EXPECT(header.read(path));
  EXPECT(header.offset() % sizeof(uint64_t) == 0);                                             // :102
  {
    // database_header::read(std::istream&), as database_query::parse_header does (src/mer_database.hpp:257-267)
    std::ifstream in(path);
    database_header sheader;
    EXPECT(sheader.read(in));
    EXPECT(sheader.check_format());
    EXPECT(sheader.offset() == header.offset() && sheader.key_bytes() == header.key_bytes() && sheader.bits() == header.bits());
    EXPECT((size_t)in.tellg() == header.offset());   // the stream stands where the keys image starts
  }
  struct stat st;
  EXPECT(stat(path, &st) == 0);
  EXPECT((size_t)st.st_size == header.offset() + header.key_bytes() + header.value_bytes());   // :103-106

  database_query database(path);
  mer_map_t mer_map;
  EXPECT(database.header().bits() == bits);                                                    // :113
  EXPECT(database.header().key_len() == mer_dna::k() * 2);                                     // :114
  test_sequence(database, hq2, 2, 1, "hq2", mer_map);                                          // :115-120
  test_sequence(database, hq1, 1, 1, "hq1", mer_map);
  test_sequence(database, lqhq, 1, 1, "lqhq", mer_map);
  test_sequence(database, hqlq, 1, 1, "hqlq", mer_map);
  test_sequence(database, lq1, 1, 0, "lq1", mer_map);
  test_sequence(database, lq2, 2, 0, "lq2", mer_map);
  {
    // the const lookups are called concurrently by the corrector's threads upstream (error_correct_reads.cc:114,173-175)
    std::atomic<int> wrong(0);
    auto hammer = [&](const std::string* seq, uint64_t count, int quality, unsigned seed) {
      mer_dna m;
      for (int i = 0; i < 300; ++i) {
        seed = seed * 1664525u + 1013904223u;
        m = seq->substr(seed % (seq->size() - mer_dna::k()), mer_dna::k());
        const std::pair<uint64_t, int> res = database[m];
        if (res.first != count || res.second != quality) ++wrong;
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < 12; ++t) {
      if (t % 3 == 0) th.emplace_back(hammer, &hq2, 2, 1, t);
      else if (t % 3 == 1) th.emplace_back(hammer, &lq1, 1, 0, t);
      else th.emplace_back(hammer, &hqlq, 1, 1, t);
    }
    for (auto& t : th) t.join();
    EXPECT(wrong.load() == 0);
  }
  {
    // begin() / end() from two threads at once on a handle nobody has iterated yet: one snapshot, both see all of it
    std::atomic<size_t> seen[2];
    seen[0] = 0; seen[1] = 0;
    auto walk = [&](int t) {
      size_t c = 0;
      for (database_query::const_iterator it = database.begin(); it != database.end(); ++it) ++c;
      seen[t] = c;
    };
    std::thread a(walk, 0), b(walk, 1);
    a.join(); b.join();
    EXPECT(seen[0].load() == mer_map.size() && seen[1].load() == mer_map.size());
  }
  size_t n = 0;
  for (database_query::const_iterator it = database.begin(); it != database.end(); ++it, ++n) {  // :123-134
    mer_map_t::const_iterator counts = mer_map.find(it->first.word());
    EXPECT(counts != mer_map.end());
    if (counts != mer_map.end()) EXPECT(counts->second == it->second);
  }
  EXPECT(n == mer_map.size());
  // get_best_alternatives on a stored HQ mer: its own base must be among the kept alternatives
  fwd_mer fm;
  fm.f = hq2.substr(5, mer_dna::k());
  uint64_t counts[4];
  int ucode = 0, level = 0;
  const int kept = database.get_best_alternatives(fm, counts, ucode, level);
  EXPECT(kept >= 1 && level == 1 && counts[fm.code(0)] == 2);
  EXPECT(fm.f.to_str() == hq2.substr(5, mer_dna::k()));  // m restored (:327)
  bool threw = false;
  try { database_query missing("/nonexistent/db"); } catch (const std::runtime_error&) { threw = true; }  // :259-260
  EXPECT(threw);
  {
    // a file whose header parses but names another format: check_format() is false and database_query refuses it
    // (what src/mer_database.hpp:264-265 means to do)
    const std::string other = std::string(path) + ".sorted";
    const std::string j = "{\"format\":\"binary/sorted\",\"size\":16,\"key_len\":62}";
    char pre[16];
    const size_t pad = (8 - (9 + j.size()) % 8) % 8;
    snprintf(pre, sizeof pre, "%09zu", j.size() + pad);
    std::ofstream o(other.c_str());
    o << pre << j << std::string(pad, '\0') << std::string(256, '\0');
    o.close();
    database_header oh;
    EXPECT(oh.read(other.c_str()) && !oh.check_format() && oh.format() == "binary/sorted");
    bool refused = false;
    std::string what;
    try { database_query wrong(other.c_str()); } catch (const std::runtime_error& e) { refused = true; what = e.what(); }
    EXPECT(refused && what.find("Wrong type 'binary/sorted'") != std::string::npos);
    unlink(other.c_str());
  }
  printf("%s: %d failure(s), %zu distinct mers\n", failures ? "FAILED" : "PASSED", failures, n);
  return failures ? 1 : 0;
}
