// text.cpp -- SURVEY.md 8f N2: the text front end of Synthesize in the product (internal/text/prepare.go, chunk.go).
//
// It decides what the runtime is asked to do: how the input is cut into <= 50-token chunks, the step budget of a chunk
// (EstimateMaxFrames) and its EOS tail (FramesAfterEOS).  Pure host string logic; the SentencePiece encoder is the
// caller's (the reference takes it as an interface, prepare.go:12-16), passed here as a callback.
//
// Unicode: Go's unicode.IsSpace is restated exactly.  ToUpper / IsLetter / IsDigit are exact for ASCII, Latin-1,
// Latin Extended-A, Greek and Cyrillic; other scripts are classified by block (letters of caseless scripts: CJK, kana,
// Hangul, Arabic, Hebrew, Devanagari ..; the decimal digit runs Nd of the common scripts).  The reference model is
// English-only (PLAN.md); its own tests are ASCII.
#include <cmath>
#include <cstring>

#include "runtime.h"

namespace ptts {

namespace {

struct Rune { uint32_t r; int size; };

Rune decode_rune(const std::string& s, size_t i) {   // utf8.DecodeRuneInString: invalid encodings give U+FFFD, width 1
    const unsigned char c0 = (unsigned char)s[i];
    if (c0 < 0x80) return {c0, 1};
    auto cont = [&](size_t k) { return i + k < s.size() && (((unsigned char)s[i + k]) & 0xC0) == 0x80; };
    if (c0 >= 0xC2 && c0 <= 0xDF && cont(1)) return {(uint32_t)((c0 & 0x1F) << 6 | ((unsigned char)s[i + 1] & 0x3F)), 2};
    if (c0 >= 0xE0 && c0 <= 0xEF && cont(1) && cont(2)) {
        uint32_t r = (uint32_t)((c0 & 0x0F) << 12 | ((unsigned char)s[i + 1] & 0x3F) << 6 | ((unsigned char)s[i + 2] & 0x3F));
        if (r >= 0x800 && !(r >= 0xD800 && r <= 0xDFFF)) return {r, 3};
    }
    if (c0 >= 0xF0 && c0 <= 0xF4 && cont(1) && cont(2) && cont(3)) {
        uint32_t r = (uint32_t)((c0 & 0x07) << 18 | ((unsigned char)s[i + 1] & 0x3F) << 12 | ((unsigned char)s[i + 2] & 0x3F) << 6 | ((unsigned char)s[i + 3] & 0x3F));
        if (r >= 0x10000 && r <= 0x10FFFF) return {r, 4};
    }
    return {0xFFFD, 1};
}

void append_rune(std::string& out, uint32_t r) {
    if (r < 0x80) out.push_back((char)r);
    else if (r < 0x800) { out.push_back((char)(0xC0 | r >> 6)); out.push_back((char)(0x80 | (r & 0x3F))); }
    else if (r < 0x10000) { out.push_back((char)(0xE0 | r >> 12)); out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F))); }
    else { out.push_back((char)(0xF0 | r >> 18)); out.push_back((char)(0x80 | ((r >> 12) & 0x3F))); out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F))); }
}

bool is_space(uint32_t r) {   // unicode.IsSpace: White_Space property
    switch (r) {
        case '\t': case '\n': case '\v': case '\f': case '\r': case ' ': case 0x85: case 0xA0: case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
            return true;
        default:
            return r >= 0x2000 && r <= 0x200A;
    }
}

uint32_t to_upper(uint32_t r) {
    if (r < 0x80) return (r >= 'a' && r <= 'z') ? r - 32 : r;
    if (r == 0xB5) return 0x39C;                                   // micro sign -> GREEK CAPITAL MU
    if (r >= 0xE0 && r <= 0xFE && r != 0xF7) return r - 32;        // Latin-1
    if (r == 0xFF) return 0x178;
    if (r >= 0x100 && r <= 0x17F) {                                // Latin Extended-A
        if (r == 0x131) return 'I';
        if (r == 0x17F) return 'S';
        if (r == 0x138 || r == 0x149) return r;
        const bool odd_lower = (r <= 0x137) || (r >= 0x14A && r <= 0x177);   // pairs (even upper, odd lower)
        if (odd_lower) return (r & 1) ? r - 1 : r;
        return (r & 1) ? r : ((r == 0x178) ? r : r - 1);           // 0x139-0x148, 0x179-0x17E: (odd upper, even lower)
    }
    if (r >= 0x3B1 && r <= 0x3C9) return r == 0x3C2 ? 0x3A3 : r - 32;   // Greek
    if (r >= 0x3AC && r <= 0x3AF) return r == 0x3AC ? 0x386 : r - 37;   // tonos
    if (r == 0x3CC) return 0x38C;
    if (r == 0x3CD || r == 0x3CE) return r - 63;
    if (r >= 0x430 && r <= 0x44F) return r - 32;                   // Cyrillic
    if (r >= 0x450 && r <= 0x45F) return r - 80;
    if (r >= 0x460 && r <= 0x481) return (r & 1) ? r - 1 : r;
    if (r >= 0x48A && r <= 0x4BF) return (r & 1) ? r - 1 : r;
    return r;
}

bool in(uint32_t r, uint32_t lo, uint32_t hi) { return r >= lo && r <= hi; }

bool is_digit(uint32_t r) {   // unicode.IsDigit: category Nd
    if (r < 0x80) return r >= '0' && r <= '9';
    static const uint32_t zero[] = {0x660, 0x6F0, 0x7C0, 0x966, 0x9E6, 0xA66, 0xAE6, 0xB66, 0xBE6, 0xC66, 0xCE6, 0xD66, 0xE50, 0xED0, 0xF20, 0x1040, 0x17E0, 0x1810, 0xFF10};
    for (uint32_t z : zero) if (in(r, z, z + 9)) return true;
    return false;
}

bool is_letter(uint32_t r) {  // unicode.IsLetter: categories L*
    if (r < 0x80) return (r >= 'a' && r <= 'z') || (r >= 'A' && r <= 'Z');
    if (r < 0x100) return r == 0xAA || r == 0xB5 || r == 0xBA || (r >= 0xC0 && r != 0xD7 && r != 0xF7);
    if (in(r, 0x100, 0x2C1) || in(r, 0x2C6, 0x2D1) || in(r, 0x2E0, 0x2E4)) return true;
    if (in(r, 0x370, 0x3FF)) return !(r == 0x375 || r == 0x378 || r == 0x379 || r == 0x37E || in(r, 0x380, 0x385) || r == 0x387 || r == 0x38B || r == 0x38D || r == 0x3A2 || r == 0x3F6);
    if (in(r, 0x400, 0x481) || in(r, 0x48A, 0x52F)) return true;                     // Cyrillic
    if (in(r, 0x531, 0x556) || in(r, 0x561, 0x587)) return true;                     // Armenian
    if (in(r, 0x5D0, 0x5EA) || in(r, 0x620, 0x64A) || in(r, 0x671, 0x6D3)) return true;   // Hebrew, Arabic
    if (in(r, 0x904, 0x939) || in(r, 0x958, 0x961)) return true;                     // Devanagari
    if (in(r, 0xE01, 0xE30) || in(r, 0x10A0, 0x10FF) || in(r, 0x1E00, 0x1FFF)) return true;   // Thai, Georgian, Latin Ext. Additional / Greek Ext.
    if (in(r, 0x3041, 0x3096) || in(r, 0x30A1, 0x30FA) || in(r, 0x3400, 0x4DBF) || in(r, 0x4E00, 0x9FFF) || in(r, 0xAC00, 0xD7A3)) return true;   // kana, CJK, Hangul
    if (in(r, 0xFF21, 0xFF3A) || in(r, 0xFF41, 0xFF5A) || in(r, 0x20000, 0x2FA1F)) return true;
    return false;
}

std::string trim_space(const std::string& s) {   // strings.TrimSpace
    size_t a = 0, b = s.size();
    while (a < b) { Rune r = decode_rune(s, a); if (!is_space(r.r)) break; a += (size_t)r.size; }
    while (b > a) {
        size_t k = b - 1;
        while (k > a && (((unsigned char)s[k]) & 0xC0) == 0x80 && b - k < 4) k--;
        Rune r = decode_rune(s, k);
        if (k + (size_t)r.size != b) { k = b - 1; r = {0xFFFD, 1}; }
        if (!is_space(r.r)) break;
        b = k;
    }
    return s.substr(a, b - a);
}

void replace_all(std::string& s, const std::string& from, const std::string& to) {
    size_t pos = 0;
    while ((pos = s.find(from, pos)) != std::string::npos) { s.replace(pos, from.size(), to); pos += to.size(); }
}

}  // namespace

int text_count_words(const std::string& s) {   // len(strings.FieldsFunc(s, unicode.IsSpace)), prepare.go:187-189
    int n = 0;
    bool in_word = false;
    for (size_t i = 0; i < s.size();) {
        Rune r = decode_rune(s, i);
        const bool sp = is_space(r.r);
        if (!sp && !in_word) n++;
        in_word = !sp;
        i += (size_t)r.size;
    }
    return n;
}

int text_estimate_max_frames(int64_t token_count, double frame_rate) {   // prepare.go:38-48
    if (token_count < 0) token_count = 0;
    if (!(frame_rate > 0) || std::isnan(frame_rate) || std::isinf(frame_rate)) frame_rate = 12.5;
    return (int)std::ceil(((double)token_count / 3.0 + 2.0) * frame_rate);
}

int text_frames_after_eos(int64_t num_words) { return num_words <= 4 ? 5 : 3; }   // prepare.go:53-59

std::string text_prepare(const std::string& input) {   // PrepareText, prepare.go:66-100
    std::string s = input;
    replace_all(s, "\r\n", " ");
    replace_all(s, "\r", " ");
    replace_all(s, "\n", " ");
    while (s.find("  ") != std::string::npos) replace_all(s, "  ", " ");
    s = trim_space(s);
    if (!s.empty()) {
        Rune r = decode_rune(s, 0);
        if (r.r != 0xFFFD) {   // utf8.RuneError (also a literal U+FFFD, like the reference's comparison)
            std::string head;
            append_rune(head, to_upper(r.r));
            s = head + s.substr((size_t)r.size);
        }
    }
    if (!s.empty()) {
        size_t k = s.size() - 1;
        while (k > 0 && (((unsigned char)s[k]) & 0xC0) == 0x80 && s.size() - k < 4) k--;
        Rune last = decode_rune(s, k);
        if (k + (size_t)last.size != s.size()) last = {0xFFFD, 1};
        if (is_letter(last.r) || is_digit(last.r)) s += ".";
    }
    if (text_count_words(s) < 5) s = "        " + s;
    return s;
}

std::vector<std::string> text_split_sentences(const std::string& text) {   // chunk.go:49-73
    std::vector<std::string> out;
    size_t start = 0;
    for (size_t i = 0; i < text.size(); i++) {   // '.', '!', '?' are ASCII: a byte scan sees exactly the runes the reference's range loop sees
        const char c = text[i];
        if (c == '.' || c == '!' || c == '?') {
            std::string s = trim_space(text.substr(start, i + 1 - start));
            if (!s.empty()) out.push_back(s);
            start = i + 1;
        }
    }
    if (start < text.size()) {
        std::string s = trim_space(text.substr(start));
        if (!s.empty()) out.push_back(s);
    }
    return out;
}

static std::string join(const std::vector<std::string>& v) {
    std::string s;
    for (size_t i = 0; i < v.size(); i++) { if (i) s += " "; s += v[i]; }
    return s;
}

std::vector<TextChunk> text_chunks(const std::string& input, const TextEncodeFn& encode, int max_tokens, double frame_rate) {   // PrepareChunks, prepare.go:105-184
    if (trim_space(input).empty()) throw Error(PTTS_EINVAL, "input text is empty");
    std::vector<std::string> sentences = text_split_sentences(input);
    if (sentences.empty()) sentences.push_back(input);
    std::vector<TextChunk> chunks;
    std::vector<std::string> pending;
    auto flush = [&] {
        if (pending.empty()) return;
        const std::string joined = join(pending);
        TextChunk c;
        c.text = text_prepare(joined);
        c.token_ids = encode(c.text);
        c.num_words = text_count_words(joined);   // of the raw sentences, not of the padded text
        c.max_frames = text_estimate_max_frames((int64_t)c.token_ids.size(), frame_rate);
        c.frames_after_eos = text_frames_after_eos(c.num_words);
        chunks.push_back(std::move(c));
        pending.clear();
    };
    for (const std::string& sent : sentences) {
        const size_t own = encode(text_prepare(sent)).size();   // the reference encodes every sentence on its own first (an encoder error surfaces here)
        size_t would_be = own;
        if (!pending.empty()) {
            std::vector<std::string> t = pending;
            t.push_back(sent);
            would_be = encode(text_prepare(join(t))).size();
        }
        if (!pending.empty() && (int64_t)would_be > (int64_t)max_tokens) flush();
        pending.push_back(sent);
    }
    flush();
    return chunks;
}

}  // namespace ptts
