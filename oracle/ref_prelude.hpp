// ref_prelude.hpp — TEST INFRASTRUCTURE (oracle).  Included before any /root/reference header by
// oracle/ref_harness.cpp; contains no reference code.
//
// Purpose: make the genuine reference arithmetic deterministic.  random_double()
// (/root/reference/common.hpp:29-34) instantiates `static std::mt19937 gen(rd())`; by including every
// standard header first and then `#define mt19937 zr_oracle_engine`, that line instantiates the
// injected counter engine below instead, whose state is thread_local and set by the harness per
// (pixel, sample).  The stream contract is include/zr_rng.h.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/zr_rng.h"

struct zr_oracle_tls_t {
    uint64_t key = 0;        // main-stream key (zr_stream_key)
    uint64_t k = 0;          // next main-stream draw index
    uint64_t draws = 0;      // main-stream draws since last reset (draw-count traces)
    uint64_t medium_draws = 0;
    uint32_t bounce = 0;     // index of the closest-hit query in flight (0 = primary)
    uint32_t medium_id = 0;
    bool in_medium = false;  // a constant_medium::hit is executing: draws are keyed off-stream
    // optional tape: when non-null, draws are read from here instead (known-answer tests)
    const double* tape = nullptr;
    size_t tape_n = 0, tape_i = 0;
    uint64_t tape_bits_dummy = 0;
};
inline thread_local zr_oracle_tls_t zr_oracle_tls;

inline void zr_oracle_seed(uint64_t seed, uint64_t pixel, uint64_t sample) {
    zr_oracle_tls.key = zr_stream_key(seed, pixel, sample);
    zr_oracle_tls.k = 0;
    zr_oracle_tls.bounce = 0;
    zr_oracle_tls.in_medium = false;
}

namespace std {
// UniformRandomBitGenerator with a 64-bit range: libstdc++'s uniform_real_distribution<double> then
// consumes exactly one word per draw and returns double(x) * 2^-64 (clamped below 1).
class zr_oracle_engine {
public:
    using result_type = uint64_t;
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return ~result_type(0); }
    explicit zr_oracle_engine(unsigned) {}
    result_type operator()() {
        zr_oracle_tls_t& s = zr_oracle_tls;
        if (s.in_medium) {
            s.medium_draws++;
            return zr_medium_bits(s.key, s.bounce, s.medium_id);
        }
        s.draws++;
        return zr_stream_bits(s.key, s.k++);
    }
};
}  // namespace std

#define mt19937 zr_oracle_engine
