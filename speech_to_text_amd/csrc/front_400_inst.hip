// The n_fft = 16 A kernels of front_400.h, instantiated: this file is compiled once per part
// (-DMFCC_16A_PART=0 .. MFCC_16A_PARTS-1), each part holding every MFCC_16A_PARTS-th length, so
// that a parallel make builds them side by side.
#include "front_400.h"

#ifndef MFCC_16A_PART
#error "compile with -DMFCC_16A_PART=<0 .. MFCC_16A_PARTS-1>"
#endif
static_assert(MFCC_16A_PART >= 0 && MFCC_16A_PART < MFCC_16A_PARTS, "no such part");

namespace {
template <int A, int W>
Pow2Fn pick(bool pw2, bool pcm16) {
    if (pw2) return pcm16 ? k_melspec_16a<A, W, 0, true> : k_melspec_16a<A, W, 0, false>;
    return pcm16 ? k_melspec_16a<A, W, 1, true> : k_melspec_16a<A, W, 1, false>;
}
template <int A>
Pow2Fn of_length(int waves, bool pw2, bool pcm16) {
    if constexpr (k16a_part(A) == MFCC_16A_PART) {
        if (waves == k16a_waves(A)) return pick<A, k16a_waves(A)>(pw2, pcm16);
        if constexpr (k16a_fallback_waves(A) != 0) {
            if (waves == k16a_fallback_waves(A)) return pick<A, k16a_fallback_waves(A)>(pw2, pcm16);
        }
    }
    return nullptr;
}
}  // namespace

#define MFCC_16A_PASTE2(a, b) a##b
#define MFCC_16A_PASTE(a, b) MFCC_16A_PASTE2(a, b)
Pow2Fn MFCC_16A_PASTE(k16a_lookup_part, MFCC_16A_PART)(int n_fft, int waves, bool pw2, bool pcm16) {
#define MFCC_16A_CASE(A) if (n_fft == 16 * A) return of_length<A>(waves, pw2, pcm16);
    MFCC_16A_LENGTHS(MFCC_16A_CASE)
#undef MFCC_16A_CASE
    return nullptr;
}
