"""TEST INFRASTRUCTURE (CPU oracle): restatement of internal/audio/dsp.go:12-78 in numpy float32 arithmetic.
PeakNormalize / FadeIn / FadeOut follow the Go code operation for operation.  DCBlock's biquad lives in a third-party module
(github.com/cwbudde/algo-dsp) that is not in the tree: not restated -- parity unpinned; the product's filter is held to the
properties of the reference's own tests (dsp_test.go:69-107)."""
import numpy as np


def peak_normalize(s):   # :12-34
    s = np.asarray(s, np.float32)
    peak = np.float32(np.abs(s.astype(np.float64)).max()) if s.size else np.float32(0)
    if peak == 0:
        return s.copy()
    gain = np.float32(1.0) / peak
    return (s * gain).astype(np.float32)


def fade_in(s, sample_rate, ms):   # :51-63
    s = np.asarray(s, np.float32)
    fade = min(int(ms / 1000.0 * float(sample_rate)), s.size)
    out = s.copy()
    for i in range(fade):
        out[i] = s[i] * (np.float32(i) / np.float32(fade))
    return out


def fade_out(s, sample_rate, ms):   # :66-80
    s = np.asarray(s, np.float32)
    fade = min(int(ms / 1000.0 * float(sample_rate)), s.size)
    out = s.copy()
    for i in range(s.size - fade, s.size):
        out[i] = s[i] * (np.float32(s.size - 1 - i) / np.float32(fade))
    return out
