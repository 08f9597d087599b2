// qh_host.h -- host-side helpers shared by the libquorum_hip translation units.
#pragma once
#include <stdarg.h>
#include <stdint.h>
#include <string>
#include "qh_spec.h"
#include "../../include/quorum_hip.h"

// thread-local last error text (qh_last_error)
void qh_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int qh_fail(int status, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define QH_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return qh_fail(QH_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// Table geometry from the create_database arguments (src/create_database.cc:105-118,
// src/mer_database.hpp:78-87). size is rounded up to a power of two like
// jellyfish::large_hash::array; wbits_override < 0 selects min(lsize, QH_WINDOW_BITS).
bool qh_make_geom(int k, int bits, uint64_t size, int min_qual, int reprobe_limit, int shard_bits,
                  int shard_id, int wbits_override, qh_geom& g, std::string& err);
