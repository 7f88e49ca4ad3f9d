// Checks rt_fastdiv.h: floor(n / d) for divisors 1 .. 70000, a spread of large ones, and for
// each a sweep of n (multiples of d and their neighbours, powers of two, random, 2^32 - 1).
//   g++ -O2 -I ray-tracer_amd/csrc tools/check_fastdiv.cpp -o /tmp/check_fastdiv && /tmp/check_fastdiv
#include <cstdint>
#include <cstdio>
#include <vector>

#include "rt_fastdiv.h"

int main() {
  std::vector<unsigned> ds;
  for (unsigned d = 1; d <= 70000; ++d) ds.push_back(d);
  for (unsigned k = 17; k < 32; ++k)
    for (int o = -2; o <= 2; ++o) ds.push_back((1u << k) + (unsigned)o);
  ds.push_back(0xffffffffu);
  ds.push_back(0xfffffffeu);
  ds.push_back(3000000019u);
  uint64_t rng = 88172645463325252ull, checked = 0;
  for (unsigned d : ds) {
    const rtdev::FastDiv f = rtdev::make_fastdiv(d);
    auto check = [&](unsigned n) {
      ++checked;
      if (rtdev::fastdiv_apply(n, f) != n / d) {
        std::printf("MISMATCH n=%u d=%u: %u != %u\n", n, d, rtdev::fastdiv_apply(n, f), n / d);
        return false;
      }
      return true;
    };
    bool ok = check(0) && check(1) && check(0xffffffffu) && check(0xfffffffeu) && check(d) && check(d - 1);
    for (unsigned k = 0; ok && k < 32; ++k) ok = check(1u << k) && check((1u << k) - 1u);
    for (int i = 0; ok && i < 64; ++i) {
      rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
      const unsigned n = (unsigned)(rng >> 16);
      const unsigned m = (unsigned)(((uint64_t)(n / d)) * d);  // a multiple of d and its neighbours
      ok = check(n) && check(m) && check(m - 1u) && check(m + (d - 1u));
    }
    if (!ok) return 1;
  }
  std::printf("fastdiv ok: %zu divisors, %llu quotients\n", ds.size(), (unsigned long long)checked);
  return 0;
}
