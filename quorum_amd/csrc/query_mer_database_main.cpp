// query_mer_database -- mirrors src/query_mer_database.cc:7-25 on libquorum_hip:
// prints k, then "<arg>:<canonical mer> val:<count> qual:<hq>" per mer argument.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/quorum_hip.h"

static int code(char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
  }
  return -1;
}

int main(int argc, char* argv[]) {
  if (argc < 3) {
    fprintf(stderr, "Usage: %s db mer ...\n", argv[0]);
    return 1;
  }
  qh_query* q = nullptr;
  if (qh_query_open(&q, argv[1], 0) != QH_OK) {
    fprintf(stderr, "%s\n", qh_last_error());
    return 1;
  }
  int k = 0;
  qh_query_info(q, &k, nullptr, nullptr);
  printf("%d\n", k);
  std::vector<uint64_t> mers;
  for (int i = 2; i < argc; ++i) {
    if ((int)strlen(argv[i]) != k) {
      fprintf(stderr, "Invalid mer '%s': length is not %d\n", argv[i], k);
      return 1;
    }
    uint64_t m = 0, rm = 0;
    for (int j = 0; j < k; ++j) {
      const int c = code(argv[i][j]);
      if (c < 0) { fprintf(stderr, "Invalid mer '%s'\n", argv[i]); return 1; }
      m = (m << 2) | (uint64_t)c;
      rm = (rm >> 2) | ((uint64_t)(3 - c) << (2 * (k - 1)));
    }
    mers.push_back(m < rm ? m : rm);  // m.canonicalize()
  }
  std::vector<uint64_t> vals(mers.size());
  if (qh_query_batch(q, mers.data(), mers.size(), vals.data()) != QH_OK) {
    fprintf(stderr, "%s\n", qh_last_error());
    return 1;
  }
  for (size_t i = 0; i < mers.size(); ++i) {
    std::string s(k, 'A');
    for (int j = 0; j < k; ++j) s[j] = "ACGT"[(mers[i] >> (2 * (k - 1 - j))) & 3];
    printf("%s:%s val:%llu qual:%d\n", argv[i + 2], s.c_str(), (unsigned long long)(vals[i] >> 1), (int)(vals[i] & 1));
  }
  qh_query_close(q);
  return 0;
}
