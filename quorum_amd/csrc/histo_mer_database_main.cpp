// histo_mer_database -- mirrors src/histo_mer_database.cc:8-29 on libquorum_hip:
// prints "count lowq highq" for every count (capped at 1000) present in the table.
#include <stdint.h>
#include <stdio.h>
#include <vector>

#include "../../include/quorum_hip.h"

int main(int argc, char* argv[]) {
  if (argc != 2) {
    fprintf(stderr, "Usage: %s db\n", argv[0]);
    return 1;
  }
  qh_query* q = nullptr;
  if (qh_query_open(&q, argv[1], 0) != QH_OK) {
    fprintf(stderr, "%s\n", qh_last_error());
    return 1;
  }
  std::vector<uint64_t> histo(1001 * 2);
  if (qh_query_histo(q, histo.data()) != QH_OK) {
    fprintf(stderr, "%s\n", qh_last_error());
    return 1;
  }
  for (size_t i = 0; i < 1001; ++i)
    if (histo[2 * i] || histo[2 * i + 1])
      printf("%zu %llu %llu\n", i, (unsigned long long)histo[2 * i], (unsigned long long)histo[2 * i + 1]);
  qh_query_close(q);
  return 0;
}
