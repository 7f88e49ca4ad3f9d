// Host-generated constants for dividing by launch constants in the kernels' hot loop.
// Plain C++ (no HIP): tools/check_fastdiv.cpp tests it on the CPU.
#pragma once

namespace rtdev {

// Division of a 32-bit unsigned by a launch constant without the ~25-instruction expansion of
// `/`: floor(n / d) = (n * (2^32 + magic)) >> (33 + shift) for every n < 2^32 (the round-up
// method with a 33-bit multiplier; Granlund & Montgomery 1994).  d == 1 is passed through.
struct FastDiv {
  unsigned magic, shift, d, pad;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f{0u, 0u, d, 0u};
  if (d < 2) return f;
  unsigned k = 31;
  while (!((d >> k) & 1u)) --k;  // floor(log2 d)
  if ((d & (d - 1)) == 0) {      // power of two: magic 0, the formula's fixed >> 1 plus k - 1 more
    f.shift = k - 1;
    return f;
  }
  // M = floor(2^(33+k) / d) + 1 is a 33-bit number; its low 32 bits are the magic
  const unsigned __int128 one = 1;
  unsigned __int128 M = (one << (33 + k)) / d + 1;
  f.magic = (unsigned)(M & 0xffffffffu);
  f.shift = k;
  return f;
}

// the kernels' evaluation (rt_kernels.inl fdiv), restated for the host-side check
inline unsigned fastdiv_apply(unsigned n, const FastDiv &f) {
  const unsigned q = (unsigned)(((unsigned long long)n * f.magic) >> 32);
  const unsigned t = ((n - q) >> 1) + q;
  return f.d == 1u ? n : (t >> f.shift);
}

}  // namespace rtdev
