// Host-only exercise of the isolated Jellyfish-layout module (quorum_amd/csrc/qh_jflayout.cpp + qh_dbfile.cpp), built
// by tests/test_jf_layout.py with g++ -fsanitize=address,undefined: write -> read round trips over a few geometries,
// growth at a tight reprobe limit, rejected inputs. No GPU, no libquorum_hip: qh_fail / qh_set_error are stubbed here.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../quorum_amd/csrc/qh_jflayout.h"
#include "../../quorum_amd/csrc/qh_host.h"

static std::string g_err;
void qh_set_error(const char* fmt, ...) { char b[1024]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); g_err = b; }
int qh_fail(int status, const char* fmt, ...) { char b[1024]; va_list ap; va_start(ap, fmt); vsnprintf(b, sizeof b, fmt, ap); va_end(ap); g_err = b; return status; }

// (only the native layout's reader calls this; it lives in qh_build.hip)
bool qh_make_geom(int, int, uint64_t, int, int, int, int, int, qh_geom&, std::string& err) { err = "not linked"; return false; }

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, g_err.c_str()); return 1; } } while (0)

static int round_trip(const char* path, int k, int bits, uint64_t size, uint64_t n, int reprobe) {
  std::map<uint64_t, uint64_t> want;
  const uint64_t kmask = (1ull << (2 * k)) - 1, vmask = (1ull << (bits + 1)) - 1;
  while (want.size() < n) { const uint64_t v = rnd() & vmask; want[rnd() & kmask] = v ? v : 1; }
  std::vector<uint64_t> mers, vals;
  for (auto& kv : want) { mers.push_back(kv.first); vals.push_back(kv.second); }
  for (size_t i = mers.size(); i > 1; --i) { const size_t j = rnd() % i; std::swap(mers[i - 1], mers[j]); std::swap(vals[i - 1], vals[j]); }
  const char* argv[] = {"test", "-x"};
  std::string err;
  CHECK(qh_jf_write_pairs(path, k, bits, size, reprobe, mers.data(), vals.data(), n, 2, argv, err));
  int k2 = 0, b2 = 0; uint64_t size2 = 0;
  std::vector<uint64_t> m2, v2;
  CHECK(qh_jf_read_pairs(path, &k2, &b2, &size2, m2, v2, err));
  CHECK(k2 == k && b2 == bits && size2 >= size && m2.size() == n);
  for (size_t i = 0; i < m2.size(); ++i) { auto it = want.find(m2[i]); CHECK(it != want.end() && it->second == v2[i]); want.erase(it); }
  CHECK(want.empty());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  if (round_trip((dir + "/a.jf").c_str(), 24, 7, 1 << 14, 9000, 126)) return 1;
  if (round_trip((dir + "/b.jf").c_str(), 31, 3, 1 << 10, 700, 126)) return 1;   // 64-bit entries nearly: 1 + 52 + 7
  if (round_trip((dir + "/c.jf").c_str(), 13, 12, 1 << 8, 256, 62)) return 1;     // full table
  if (round_trip((dir + "/d.jf").c_str(), 24, 7, 1 << 10, 1000, 1)) return 1;     // two probes only: the table must grow
  if (round_trip((dir + "/e.jf").c_str(), 24, 7, 1 << 10, 0, 126)) return 1;      // empty
  std::string err;
  uint64_t one = 1, val = 3;
  const char* av[] = {"t"};
  CHECK(!qh_jf_write_pairs((dir + "/x").c_str(), 5, 7, 1 << 12, 126, &one, &val, 1, 1, av, err));    // 2k <= log2 size
  val = 0;
  CHECK(!qh_jf_write_pairs((dir + "/x").c_str(), 24, 7, 1 << 12, 126, &one, &val, 1, 1, av, err));   // value 0 = empty
  // a truncated file and a header without a matrix are rejected, not read past
  { FILE* f = fopen((dir + "/t.jf").c_str(), "wb"); fputs("000000016{\"format\":\"x\"}\n", f); fputc(0, f); fclose(f); }
  int k2, b2; uint64_t s2; std::vector<uint64_t> m2, v2;
  CHECK(!qh_jf_read_pairs((dir + "/t.jf").c_str(), &k2, &b2, &s2, m2, v2, err));
  { // chop the values image off a good file
    FILE* f = fopen((dir + "/a.jf").c_str(), "rb"); std::vector<char> all(1 << 20); const size_t got = fread(all.data(), 1, all.size(), f); fclose(f);
    f = fopen((dir + "/u.jf").c_str(), "wb"); fwrite(all.data(), 1, got - 100, f); fclose(f);
  }
  CHECK(!qh_jf_read_pairs((dir + "/u.jf").c_str(), &k2, &b2, &s2, m2, v2, err));
  puts("ok");
  return 0;
}
