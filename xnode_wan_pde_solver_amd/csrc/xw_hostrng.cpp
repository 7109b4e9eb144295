// Host-side helper of the samplers (src/dataset.py:248-272: every sample is torch.rand / uniform_ on torch's global CPU
// generator, and "same seeds" means the same stream): float32 uniform fills straight from the generator's state blob.
//
// torch's CPU uniform_ walks a scalar mt19937 (at::mt19937, ATen/core/MT19937RNGEngine.h) element by element: 2.8 ns per
// number, and a training iteration draws 0.66 M of them on one thread in the reference's order -- the floor of train() once
// the GPU side is pipelined.  Here the 624-word state is regenerated and tempered in vectorisable loops and the
// numbers are converted in blocks: same words, same order, same arithmetic as at::uniform_real_distribution<float>
// ((y & (2^24 - 1)) * 2^-24 * (to - from) + from in float; `fused` says whether the last two operations are one fma -- the
// caller finds out which by comparing with torch once at import and falls back to torch if neither matches).
//
// State blob = torch.get_rng_state() (CPUGeneratorImplStateLegacy, 5056 bytes):
//   u64 seed | i32 left | i32 seeded | u64 next | u64 state[624] (32-bit words) | normal-distribution cache (untouched)
#include <cstdint>
#include <cstring>
#include <cmath>

namespace {
constexpr int N = 624, M = 397;
constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;

inline uint32_t twist(uint32_t u, uint32_t v) { return (((u & UPPER) | (v & LOWER)) >> 1) ^ ((v & 1u) ? MATRIX : 0u); }

// the three classic loops of the regeneration: no loop-carried dependence shorter than M / N - M words, so they vectorise
static inline __attribute__((always_inline)) void regenerate_body(uint32_t* s) {
  for (int j = 0; j < N - M; ++j) s[j] = s[j + M] ^ twist(s[j], s[j + 1]);
  for (int j = N - M; j < N - 1; ++j) s[j] = s[j + M - N] ^ twist(s[j], s[j + 1]);
  s[N - 1] = s[M - 1] ^ twist(s[N - 1], s[0]);
}
static inline __attribute__((always_inline)) void emit_body(const uint32_t* s, float* out, int k, float span, float from, int fused) {
  if (fused) {
    for (int i = 0; i < k; ++i) {
      uint32_t y = s[i];
      y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
      out[i] = __builtin_fmaf((float)(y & 0xffffffu) * 5.9604644775390625e-8f, span, from);
    }
  } else {
    for (int i = 0; i < k; ++i) {
      uint32_t y = s[i];
      y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
      const float x = (float)(y & 0xffffffu) * 5.9604644775390625e-8f * span;
      out[i] = x + from;
    }
  }
}
// two builds of the loops, picked once by what the CPU has (AVX2 + FMA: 8-wide, the fused form is one instruction;
// baseline x86-64: SSE2, the fused form goes through libm's exact fmaf)
__attribute__((target("avx2,fma"))) void regenerate_v3(uint32_t* s) { regenerate_body(s); }
__attribute__((target("avx2,fma"))) void emit_v3(const uint32_t* s, float* out, int k, float span, float from, int fused) {
  emit_body(s, out, k, span, from, fused);
}
void regenerate_v1(uint32_t* s) { regenerate_body(s); }
void emit_v1(const uint32_t* s, float* out, int k, float span, float from, int fused) { emit_body(s, out, k, span, from, fused); }
const bool wide = (__builtin_cpu_init(), __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"));
inline void regenerate(uint32_t* s) { wide ? regenerate_v3(s) : regenerate_v1(s); }
inline void emit(const uint32_t* s, float* out, int k, float span, float from, int fused) {
  wide ? emit_v3(s, out, k, span, from, fused) : emit_v1(s, out, k, span, from, fused);
}
}  // namespace

extern "C" int xw_mt19937_uniform_f32(void* state_blob, long blob_bytes, float* out, long n, float from, float to, int fused) {
  if (!state_blob || blob_bytes < 24 + 8 * N || (!out && n > 0) || n < 0) return -2;
  uint8_t* blob = static_cast<uint8_t*>(state_blob);
  int32_t left, seeded;
  uint64_t next;
  std::memcpy(&left, blob + 8, 4);
  std::memcpy(&seeded, blob + 12, 4);
  std::memcpy(&next, blob + 16, 8);
  if (!seeded || left < 1 || left > N || next > (uint64_t)N) return -2;
  uint32_t s[N];
  uint64_t w;
  for (int j = 0; j < N; ++j) { std::memcpy(&w, blob + 24 + 8 * j, 8); s[j] = (uint32_t)w; }
  const float span = to - from;
  long done = 0;
  while (done < n) {
    // at::mt19937::operator(): if (--left == 0) next_state() [left = 624, next = 0]; y = state[next++]
    if (left == 1) { regenerate(s); left = N + 1; next = 0; }
    const long avail = left - 1;
    const int k = (int)(n - done < avail ? n - done : avail);
    emit(s + next, out + done, k, span, from, fused);
    next += (uint64_t)k; left -= k; done += k;
  }
  std::memcpy(blob + 8, &left, 4);
  std::memcpy(blob + 16, &next, 8);
  for (int j = 0; j < N; ++j) { w = s[j]; std::memcpy(blob + 24 + 8 * j, &w, 8); }
  return 0;
}

// ---- numpy's legacy normal stream ------------------------------------------------------------------------------------------
// The ball domains draw their points with np.random.normal on numpy's GLOBAL RandomState (src/dataset.py:65, 180:
// "same seeds" = this stream too): legacy_gauss of numpy/random/src/legacy/legacy-distributions.c, the polar method on pairs of
// 53-bit doubles of the mt19937 words, the second value of a pair cached in the state.  numpy walks it value by value, 18.6 ns
// each, 0.3 M per hourglass sample -- what bounded the ball domains' train().  Here the words of a state block are
// tempered and turned into candidate pairs in vectorisable loops, the rejection scan only compacts (branch-free), and the log /
// sqrt of the accepted pairs -- libm's scalar log, the function numpy calls: same bits -- run back to back over a block: 4.5 ns
// per value.  Same values, same order, same state left behind (key, pos, has_gauss, cached value); the caller verifies that
// against numpy once per process.
namespace {
inline uint32_t temper(uint32_t y) {
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static inline __attribute__((always_inline)) void candidates_body(const uint32_t* s, int attempts, double* x1, double* x2, double* r2) {
  for (int i = 0; i < attempts; ++i) {
    const uint32_t a1 = temper(s[4 * i]) >> 5, b1 = temper(s[4 * i + 1]) >> 6, a2 = temper(s[4 * i + 2]) >> 5, b2 = temper(s[4 * i + 3]) >> 6;
    const double u1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0;      // mt19937_next_double
    const double u2 = ((double)a2 * 67108864.0 + (double)b2) / 9007199254740992.0;
    const double p = 2.0 * u1 - 1.0, q = 2.0 * u2 - 1.0;
    x1[i] = p; x2[i] = q; r2[i] = p * p + q * q;
  }
}
__attribute__((target("avx2"))) void candidates_v3(const uint32_t* s, int n, double* x1, double* x2, double* r2) { candidates_body(s, n, x1, x2, r2); }
void candidates_v1(const uint32_t* s, int n, double* x1, double* x2, double* r2) { candidates_body(s, n, x1, x2, r2); }

}  // namespace

extern "C" int xw_mt19937_legacy_normal_f64(uint32_t* key, int* pos_io, int* has_gauss_io, double* cached_io, double* out, long n) {
  if (!key || !pos_io || !has_gauss_io || !cached_io || (!out && n > 0) || n < 0) return -2;
  int pos = *pos_io;
  if (pos < 0 || pos > N) return -2;
  long done = 0;
  if (n > 0 && *has_gauss_io) { out[done++] = 0.0 + 1.0 * *cached_io; *has_gauss_io = 0; *cached_io = 0.0; }
  const long want = (n - done + 1) / 2;                 // pairs still to accept
  // words of the current state block that have not been handed out, behind up to three words left over from the block before
  constexpr int A = N / 4 + 2;
  uint32_t w[N + 4];
  double x1[A], x2[A], r2[A], a1[A], a2[A], ar[A], vals[2 * A];
  int carry = 0;
  long got = 0;
  while (got < want) {
    if (pos == N) { regenerate(key); pos = 0; }
    const int fresh = N - pos;
    std::memcpy(w + carry, key + pos, sizeof(uint32_t) * (size_t)fresh);
    const int have = carry + fresh, attempts = have / 4;
    wide ? candidates_v3(w, attempts, x1, x2, r2) : candidates_v1(w, attempts, x1, x2, r2);
    int i = 0, k = 0;
    if (want - got >= attempts) {                        // the whole block is needed: compact without branches
      for (; i < attempts; ++i) {
        a1[k] = x1[i]; a2[k] = x2[i]; ar[k] = r2[i];
        k += !(r2[i] >= 1.0 || r2[i] == 0.0);
      }
    } else {
      for (; i < attempts && got + k < want; ++i)
        if (!(r2[i] >= 1.0 || r2[i] == 0.0)) { a1[k] = x1[i]; a2[k] = x2[i]; ar[k] = r2[i]; ++k; }
    }
    for (int j = 0; j < k; ++j) {
      const double f = std::sqrt(-2.0 * std::log(ar[j]) / ar[j]);
      vals[2 * j] = 0.0 + 1.0 * (f * a2[j]);             // legacy_normal: loc + scale * gauss, loc = 0, scale = 1
      vals[2 * j + 1] = 0.0 + 1.0 * (f * a1[j]);         // (the value legacy_gauss caches and hands out next)
    }
    got += k;
    const long room = n - done, m = 2L * k < room ? 2L * k : room;
    std::memcpy(out + done, vals, sizeof(double) * (size_t)m);
    done += m;
    if (m < 2L * k) { *has_gauss_io = 1; *cached_io = vals[2 * k - 1]; }     // (only the very last pair can be cut)
    const int used = 4 * i;                              // words consumed, the carried ones first
    if (got == want) { pos = N - (have - used); break; }  // (the unused words all belong to this block: carry < 4 <= used)
    carry = have - used;
    std::memmove(w, w + used, sizeof(uint32_t) * (size_t)carry);
    pos = N;
  }
  *pos_io = pos;
  return 0;
}
