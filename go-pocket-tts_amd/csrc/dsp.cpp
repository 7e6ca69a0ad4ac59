// dsp.cpp -- SURVEY.md 8f N3, the optional post-processing of a finished utterance (internal/audio/dsp.go:12-78; applied by the CLI
// in the order normalise -> DC block -> fade in -> fade out, cmd/pockettts/synth.go:361-390).  Host code on host samples: the
// filter is a sample-by-sample recurrence over one utterance, there is nothing for the GPU in it.
//
// PeakNormalize, FadeIn and FadeOut follow dsp.go operation for operation (float32 products, the same gain expressions): bit-exact
// against the oracle.  DCBlock delegates, in the reference, to github.com/cwbudde/algo-dsp (design.Highpass(20 Hz, Q 0.707) +
// biquad.Section, go.mod), which is not in the tree: the published RBJ cookbook high-pass in direct form II transposed with float64
// state is used, and it is held to the properties the reference's own tests state (dsp_test.go:69-107), not to its bits: PARITY
// UNPINNED for this one function.
#include <cmath>

#include "runtime.h"

namespace ptts {

void dsp_peak_normalize(float* s, int64_t n) {   // dsp.go:12-34
    float peak = 0.0f;
    for (int64_t i = 0; i < n; i++) {
        const float a = (float)std::fabs((double)s[i]);
        if (a > peak) peak = a;
    }
    if (peak == 0.0f) return;
    const float gain = 1.0f / peak;   // `gain := 1.0 / peak` is a float32 division (untyped constant, float32 operand)
    for (int64_t i = 0; i < n; i++) s[i] = s[i] * gain;
}

void dsp_dc_block(float* s, int64_t n, int sample_rate) {   // dsp.go:38-48 (20 Hz, Q 0.707)
    const double w0 = 2.0 * M_PI * 20.0 / (double)sample_rate, q = 0.707;
    const double cw = std::cos(w0), alpha = std::sin(w0) / (2.0 * q);
    const double a0 = 1.0 + alpha;
    const double b0 = (1.0 + cw) / 2.0 / a0, b1 = -(1.0 + cw) / a0, b2 = b0, a1 = -2.0 * cw / a0, a2 = (1.0 - alpha) / a0;
    double z1 = 0.0, z2 = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const double x = (double)s[i];
        const double y = b0 * x + z1;
        z1 = b1 * x - a1 * y + z2;
        z2 = b2 * x - a2 * y;
        s[i] = (float)y;
    }
}

void dsp_fade_in(float* s, int64_t n, int sample_rate, double ms) {   // dsp.go:51-63
    const int64_t fade = std::min<int64_t>((int64_t)(ms / 1000.0 * (double)sample_rate), n);
    for (int64_t i = 0; i < fade; i++) s[i] = s[i] * ((float)i / (float)fade);
}

void dsp_fade_out(float* s, int64_t n, int sample_rate, double ms) {   // dsp.go:66-80
    const int64_t fade = std::min<int64_t>((int64_t)(ms / 1000.0 * (double)sample_rate), n);
    for (int64_t i = n - fade; i < n; i++) {
        const int64_t remaining = n - 1 - i;
        s[i] = s[i] * ((float)remaining / (float)fade);
    }
}

}  // namespace ptts
