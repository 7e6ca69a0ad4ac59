// safetensors.cpp -- reader for the on-disk checkpoint / voice format.
// Follows internal/safetensors/store.go: header parse (:246-271), entry validation
// (:288-308), byte-range checks (:127-160), dtype decode to f32 (:339-395), f16
// conversion (:397-431).  A minimal JSON reader for the header (objects, arrays,
// strings, integers) replaces encoding/json.
#include <cmath>
#include <cstdarg>
#include <fstream>

#include "common.h"

namespace ptts {

std::string strfmt(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return std::string(buf);
}

static thread_local std::string g_last_error;
void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error_ref() { return g_last_error; }

int64_t StEntry::count() const {
    int64_t n = 1;
    for (int64_t d : shape) {
        if (d == 0) return 0;
        n *= d;
    }
    return n;
}

const StEntry& StFile::at(const std::string& n) const {
    auto it = entries.find(n);
    if (it == entries.end()) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" not found", n.c_str()));
    return it->second;
}

namespace {

struct JsonCursor {
    const char* p;
    const char* e;
    void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    [[noreturn]] void fail(const char* what) { throw Error(PTTS_EFORMAT, strfmt("safetensors: parse header: %s", what)); }
    void expect(char c) { ws(); if (p >= e || *p != c) fail("unexpected character"); p++; }
    bool peek(char c) { ws(); return p < e && *p == c; }
    std::string str() {
        ws();
        if (p >= e || *p != '"') fail("expected string");
        p++;
        std::string s;
        while (p < e && *p != '"') {
            if (*p == '\\') {
                p++;
                if (p >= e) fail("bad escape");
                switch (*p) {
                    case 'n': s += '\n'; break;
                    case 't': s += '\t'; break;
                    case 'r': s += '\r'; break;
                    case 'b': s += '\b'; break;
                    case 'f': s += '\f'; break;
                    case 'u': {  // keep ASCII range only; names never need more
                        if (e - p < 5) fail("bad \\u escape");
                        unsigned v = 0;
                        for (int i = 1; i <= 4; i++) {
                            char c = p[i];
                            v <<= 4;
                            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
                            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
                            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
                            else fail("bad \\u escape");
                        }
                        s += (char)(v & 0x7f);
                        p += 4;
                        break;
                    }
                    default: s += *p;
                }
                p++;
            } else {
                s += *p++;
            }
        }
        if (p >= e) fail("unterminated string");
        p++;
        return s;
    }
    int64_t integer() {
        ws();
        bool neg = false;
        if (p < e && *p == '-') { neg = true; p++; }
        if (p >= e || *p < '0' || *p > '9') fail("expected integer");
        int64_t v = 0;
        while (p < e && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        return neg ? -v : v;
    }
    void skip_value() {  // for __metadata__ and unknown keys
        ws();
        if (p >= e) fail("truncated");
        if (*p == '"') { str(); return; }
        if (*p == '{') {
            p++;
            if (peek('}')) { p++; return; }
            for (;;) { str(); expect(':'); skip_value(); if (peek(',')) { p++; continue; } expect('}'); return; }
        }
        if (*p == '[') {
            p++;
            if (peek(']')) { p++; return; }
            for (;;) { skip_value(); if (peek(',')) { p++; continue; } expect(']'); return; }
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') p++;  // number / literal
    }
};

float f16_to_f32(uint16_t h) {  // store.go:397-431
    uint32_t sign = (h >> 15) & 1u, exp = (h >> 10) & 0x1fu, frac = h & 0x3ffu, bits;
    if (exp == 0) {
        if (frac == 0) bits = sign << 31;
        else {
            uint32_t e32 = 113;
            while ((frac & 0x400u) == 0) { frac <<= 1; e32--; }
            frac &= 0x3ffu;
            bits = (sign << 31) | (e32 << 23) | (frac << 13);
        }
    } else if (exp == 0x1f) bits = (sign << 31) | 0x7f800000u | (frac << 13);
    else bits = (sign << 31) | ((exp + 112) << 23) | (frac << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

int dtype_bytes(const std::string& dt) {
    if (dt == "F32") return 4;
    if (dt == "F16" || dt == "BF16") return 2;
    if (dt == "I64") return 8;
    return 0;
}

}  // namespace

void st_parse(StFile& f) {
    if (f.size < 8) throw Error(PTTS_EFORMAT, strfmt("safetensors: file too short (%zu bytes)", f.size));
    uint64_t hlen = 0;
    std::memcpy(&hlen, f.data, 8);  // little-endian host
    if (hlen > f.size - 8) throw Error(PTTS_EFORMAT, strfmt("safetensors: header length %llu exceeds file size %zu", (unsigned long long)hlen, f.size));
    size_t header_end = 8 + (size_t)hlen;
    JsonCursor c{(const char*)f.data + 8, (const char*)f.data + header_end};
    c.expect('{');
    if (!c.peek('}')) {
        for (;;) {
            std::string name = c.str();
            c.expect(':');
            if (name == "__metadata__") {
                c.skip_value();
            } else {
                StEntry en;
                int64_t o0 = -1, o1 = -1;
                c.expect('{');
                for (;;) {
                    std::string key = c.str();
                    c.expect(':');
                    if (key == "dtype") {
                        en.dtype = c.str();
                        for (auto& ch : en.dtype) ch = (char)toupper((unsigned char)ch);
                    } else if (key == "shape") {
                        c.expect('[');
                        if (!c.peek(']')) for (;;) { en.shape.push_back(c.integer()); if (c.peek(',')) { c.p++; continue; } break; }
                        c.expect(']');
                    } else if (key == "data_offsets") {
                        c.expect('['); o0 = c.integer(); c.expect(','); o1 = c.integer(); c.expect(']');
                    } else c.skip_value();
                    if (c.peek(',')) { c.p++; continue; }
                    c.expect('}');
                    break;
                }
                // validateHeaderEntry store.go:288-308
                if (dtype_bytes(en.dtype) == 0) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" has unsupported dtype \"%s\"", name.c_str(), en.dtype.c_str()));
                if (o0 < 0 || o1 < o0) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" has invalid data offsets [%lld %lld]", name.c_str(), (long long)o0, (long long)o1));
                for (int64_t d : en.shape) if (d < 0) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" has negative shape dimension", name.c_str()));
                en.begin = header_end + (size_t)o0;
                en.end = header_end + (size_t)o1;
                if (en.end > f.size) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" data [%zu:%zu] exceeds file size %zu", name.c_str(), en.begin, en.end, f.size));
                size_t need = (size_t)en.count() * (size_t)dtype_bytes(en.dtype);
                if (en.end - en.begin < need) throw Error(PTTS_EFORMAT, strfmt("safetensors: tensor \"%s\" needs %zu bytes but data has %zu", name.c_str(), need, en.end - en.begin));
                // trim like strings.TrimSpace(mapped) in store.go:118
                size_t a = name.find_first_not_of(" \t\n\r"), b = name.find_last_not_of(" \t\n\r");
                std::string mapped = a == std::string::npos ? "" : name.substr(a, b - a + 1);
                if (mapped.empty()) throw Error(PTTS_EFORMAT, strfmt("safetensors: remapped tensor name for \"%s\" is empty", name.c_str()));
                f.entries.emplace(mapped, std::move(en));  // first one wins on collision (lenient mode)
            }
            if (c.peek(',')) { c.p++; continue; }
            c.expect('}');
            break;
        }
    } else c.p++;
    if (f.entries.empty()) throw Error(PTTS_EFORMAT, "safetensors: no tensors found");
}

void st_open_path(const std::string& path, StFile& f) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in) throw Error(PTTS_EIO, strfmt("safetensors: read %s: cannot open", path.c_str()));
    std::streamsize n = in.tellg();
    in.seekg(0);
    f.owned.resize((size_t)n);
    if (n > 0 && !in.read((char*)f.owned.data(), n)) throw Error(PTTS_EIO, strfmt("safetensors: read %s: short read", path.c_str()));
    f.data = f.owned.data();
    f.size = f.owned.size();
    st_parse(f);
}

void StFile::decode_f32(const std::string& name, float* out) const {
    const StEntry& en = at(name);
    const uint8_t* raw = data + en.begin;
    int64_t n = en.count();
    if (en.dtype == "F32") {
        std::memcpy(out, raw, (size_t)n * 4);
    } else if (en.dtype == "F16") {
        for (int64_t i = 0; i < n; i++) { uint16_t b; std::memcpy(&b, raw + i * 2, 2); out[i] = f16_to_f32(b); }
    } else if (en.dtype == "BF16") {
        for (int64_t i = 0; i < n; i++) { uint16_t b; std::memcpy(&b, raw + i * 2, 2); out[i] = bf16_to_f32(b); }
    } else if (en.dtype == "I64") {
        for (int64_t i = 0; i < n; i++) { int64_t v; std::memcpy(&v, raw + i * 8, 8); out[i] = (float)v; }
    } else {
        throw Error(PTTS_EFORMAT, strfmt("unsupported dtype \"%s\"", en.dtype.c_str()));
    }
}

}  // namespace ptts
