// rocrand_harness.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// cuRAND, whose XORWOW the reference calls (integrator.h:63-64, 210, 279, 384-385), is not in this image; rocRAND
// (/opt/rocm/include/rocrand/rocrand_xorwow.h, ROCRAND_VERSION 400200 in ROCm 7.2) is a third party's implementation of the
// same generator: Marsaglia's xorwow recurrence, the Weyl sequence d += 362437, output d + x[4], and subsequences of 2^67
// draws skipped with precomputed GF(2) matrix powers.  Its SEEDING differs from cuRAND's on purpose (other scramble
// constants) and so does its float conversion, so what this harness can pin is the step and the 2^67-skip - through a
// subclass that sets and reads the engine's raw state.  tests/test_rng_vs_rocrand.py compares the oracle's
// xorwow_next / jump tables with it.
#include <rocrand/rocrand_version.h>
#include <rocrand/rocrand_xorwow.h>

namespace {
struct OpenEngine : rocrand_device::xorwow_engine {
    void set(const unsigned int* x, unsigned int d) { for (int i = 0; i < 5; i++) m_state.x[i] = x[i]; m_state.d = d; }
    void get(unsigned int* x, unsigned int* d) const { for (int i = 0; i < 5; i++) x[i] = m_state.x[i]; *d = m_state.d; }
};
}  // namespace

extern "C" {
int rr_version(void) { return ROCRAND_VERSION; }
// n draws from the state (x, d); the state is updated in place
void rr_next(unsigned int x[5], unsigned int* d, int n, unsigned int* out) {
    OpenEngine e; e.set(x, *d);
    for (int i = 0; i < n; i++) out[i] = e.next();
    e.get(x, d);
}
// skip n subsequences of 2^67 draws (rocRAND's own precomputed h_xorwow_sequence_jump_matrices)
void rr_skip_subsequences(unsigned int x[5], unsigned int* d, unsigned long long n) {
    OpenEngine e; e.set(x, *d);
    e.discard_subsequence(n);
    e.get(x, d);
}
// skip n draws (h_xorwow_jump_matrices)
void rr_skip(unsigned int x[5], unsigned int* d, unsigned long long n) {
    OpenEngine e; e.set(x, *d);
    e.discard(n);
    e.get(x, d);
}
}
