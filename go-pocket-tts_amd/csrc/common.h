// common.h -- shared declarations of libptts_hip (MI355X / gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ptts.h"

namespace ptts {

// ---- errors: thrown inside the library, converted to codes at the C ABI ----
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

std::string strfmt(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void set_last_error(const std::string& m);

#define PTTS_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            throw ::ptts::Error(PTTS_ENODEVICE, ::ptts::strfmt("hip: %s failed: %s (%s:%d)", #expr,     \
                                                               hipGetErrorString(_e), __FILE__, __LINE__)); \
    } while (0)

// ---- bf16 helpers (host) ----
static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); // keep NaN a NaN
    uint32_t r = ((u >> 16) & 1u) + 0x7fffu;
    return (uint16_t)((u + r) >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// ---- safetensors (internal/safetensors/store.go) ----
struct StEntry {
    std::string dtype;  // upper-case
    std::vector<int64_t> shape;
    size_t begin = 0, end = 0;  // absolute byte range in the file image
    int64_t count() const;
};
struct StFile {
    std::vector<uint8_t> owned;  // when read from a path
    const uint8_t* data = nullptr;
    size_t size = 0;
    std::map<std::string, StEntry> entries;  // sorted by name, like Store.names
    bool has(const std::string& n) const { return entries.count(n) != 0; }
    const StEntry& at(const std::string& n) const;
    // store.go:339-395: every dtype decodes to float32
    void decode_f32(const std::string& name, float* out) const;
};
void st_parse(StFile& f);                       // store.go:65-184, 246-271
void st_open_path(const std::string& path, StFile& f);

}  // namespace ptts
