// Host-side publication primitives of the one-node parameter exchange (xmris_amd/sharding.py::ShmExchange): the O(1)
// per-dataset hand-off between the ranks of a node that autophase's GLOBAL arg-max and its single (p0, p1) need
// (reference processing/phasing.py:229, 276-290) goes through a shared-memory page of sequence-numbered slots.  The
// payload is written with plain stores; the sequence word that publishes it is written with a RELEASE store and read
// with ACQUIRE loads here, so the protocol does not lean on x86's store ordering or on what the Python interpreter
// happens to do between two numpy stores.  Host code only.
#include <chrono>
#include <cstdint>

#if defined(__x86_64__)
#include <immintrin.h>
static inline void cpu_relax() { _mm_pause(); }
#else
static inline void cpu_relax() {}
#endif

extern "C" {

int64_t xm_atomic_load_acquire_i64(const int64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }

void xm_atomic_store_release_i64(int64_t* p, int64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

// 1 as soon as every one of `count` words, `stride_words` apart, holds a value >= `value` (acquire loads); 0 when that
// has not happened within `spin_us` microseconds of busy polling (the caller then sleeps between calls).
int xm_atomic_wait_all_ge_i64(const int64_t* p, int stride_words, int count, int64_t value, int spin_us) {
  if (!p || count < 0 || stride_words < 1) return -1;
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us < 0 ? 0 : spin_us);
  for (;;) {
    int ok = 1;
    for (int i = 0; i < count && ok; ++i) ok = __atomic_load_n(p + (long)i * stride_words, __ATOMIC_ACQUIRE) >= value;
    if (ok) return 1;
    for (int k = 0; k < 32; ++k) cpu_relax();
    if (std::chrono::steady_clock::now() >= t_end) return 0;
  }
}

}  // extern "C"
