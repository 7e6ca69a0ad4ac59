// tokenizer.cpp -- SURVEY.md 8f N2: the SentencePiece unigram encoder of the product.
//
// The reference tokenises through github.com/vikesh-raj/go-sentencepiece-encoder v1.1.1 (go.mod:14, call site
// internal/tokenizer/sentencepiece.go:19-40) and restates that library's algorithm in-tree for js/wasm
// (internal/tokenizer/sentencepiece_bytes_wasm.go): normalise (drop control characters, every White_Space to ' ', NFKC), prepend
// U+2581 unless the text starts with one, White_Space -> U+2581, then a Viterbi search over a trie of the model's NORMAL /
// USER_DEFINED pieces (float32 scores, strict '>' updates, a single-rune UNKNOWN step where nothing matches) and a merge of
// consecutive UNKNOWNs.  That file is what this one follows.  The model file is the SentencePiece ModelProto; only
// `pieces` (field 1: piece, score, type) is read, like the reference.
//
// NFKC comes from generated Unicode tables (nfkc_tables.h, tools/gen_nfkc_tables.py): compatibility decomposition, canonical
// ordering, canonical composition, Hangul by arithmetic.
#include <algorithm>
#include <cfloat>
#include <fstream>
#include <unordered_map>

#include "nfkc_tables.h"
#include "runtime.h"

namespace ptts {

namespace {

// ---- UTF-8 (utf8.DecodeRuneInString semantics: an invalid byte is U+FFFD of width 1) ----
uint32_t decode_at(const std::string& s, size_t& i) {
    const unsigned char c0 = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && (((unsigned char)s[i + k]) & 0xC0) == 0x80; };
    if (c0 < 0x80) { i += 1; return c0; }
    if (c0 >= 0xC2 && c0 <= 0xDF && cont(1)) { uint32_t r = (uint32_t)((c0 & 0x1F) << 6 | ((unsigned char)s[i + 1] & 0x3F)); i += 2; return r; }
    if (c0 >= 0xE0 && c0 <= 0xEF && cont(1) && cont(2)) {
        uint32_t r = (uint32_t)((c0 & 0x0F) << 12 | ((unsigned char)s[i + 1] & 0x3F) << 6 | ((unsigned char)s[i + 2] & 0x3F));
        if (r >= 0x800 && !(r >= 0xD800 && r <= 0xDFFF)) { i += 3; return r; }
    }
    if (c0 >= 0xF0 && c0 <= 0xF4 && cont(1) && cont(2) && cont(3)) {
        uint32_t r = (uint32_t)((c0 & 0x07) << 18 | ((unsigned char)s[i + 1] & 0x3F) << 12 | ((unsigned char)s[i + 2] & 0x3F) << 6 | ((unsigned char)s[i + 3] & 0x3F));
        if (r >= 0x10000 && r <= 0x10FFFF) { i += 4; return r; }
    }
    i += 1;
    return 0xFFFD;
}

std::vector<uint32_t> to_runes(const std::string& s) {
    std::vector<uint32_t> r;
    r.reserve(s.size());
    for (size_t i = 0; i < s.size();) r.push_back(decode_at(s, i));
    return r;
}

bool is_space(uint32_t r) {   // unicode.IsSpace: White_Space property
    switch (r) {
        case '\t': case '\n': case '\v': case '\f': case '\r': case ' ': case 0x85: case 0xA0: case 0x1680: case 0x2028: case 0x2029: case 0x202F: case 0x205F: case 0x3000:
            return true;
        default:
            return r >= 0x2000 && r <= 0x200A;
    }
}

// ---- sentencepiece_bytes_wasm.go:222-260: characters the normaliser drops ----
const uint32_t kControlChars[] = {
    0x007F, 0x00AD, 0x0600, 0x0601, 0x0602, 0x0603, 0x0604, 0x0605, 0x061C, 0x06DD, 0x070F, 0x08E2, 0x180E, 0x200B, 0x200C, 0x200D, 0x200E, 0x200F,
    0x202A, 0x202B, 0x202C, 0x202D, 0x202E, 0x2060, 0x2061, 0x2062, 0x2063, 0x2064, 0x2066, 0x2067, 0x2068, 0x2069, 0x206A, 0x206B, 0x206C, 0x206D,
    0x206E, 0x206F, 0xFEFF, 0xFFF9, 0xFFFA, 0xFFFB, 0x110BD, 0x110CD, 0x13430, 0x13431, 0x13432, 0x13433, 0x13434, 0x13435, 0x13436, 0x13437, 0x13438,
    0x1BCA0, 0x1BCA1, 0x1BCA2, 0x1BCA3, 0x1D173, 0x1D174, 0x1D175, 0x1D176, 0x1D177, 0x1D178, 0x1D179, 0x1D17A, 0xE0001,
};
bool is_control(uint32_t c) {
    if (c == ' ' || c == '\n' || c == '\r' || c == '\t') return false;
    if (c <= 0x1F || (c >= 0x80 && c <= 0x9F) || (c >= 0xE0020 && c <= 0xE007F) || (c >= 0xE000 && c <= 0xF8FF) || (c >= 0xF0000 && c <= 0xFFFFD) ||
        (c >= 0x100000 && c <= 0x10FFFD) || (c >= 0xD800 && c <= 0xDFFF))
        return true;
    for (uint32_t k : kControlChars) if (k == c) return true;
    return false;
}

// ---- NFKC ----
constexpr uint32_t SB = 0xAC00, LB = 0x1100, VB = 0x1161, TB = 0x11A7, LC = 19, VC = 21, TC = 28, NC = VC * TC, SC = LC * NC;

uint8_t ccc_of(uint32_t cp) {
    const nfkc::Ccc* lo = nfkc::kCcc;
    const nfkc::Ccc* hi = lo + sizeof(nfkc::kCcc) / sizeof(nfkc::kCcc[0]);
    const nfkc::Ccc* it = std::lower_bound(lo, hi, cp, [](const nfkc::Ccc& e, uint32_t v) { return e.cp < v; });
    return it != hi && it->cp == cp ? it->cls : 0;
}

void decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp >= SB && cp < SB + SC) {   // Hangul syllable -> L V (T)
        const uint32_t s = cp - SB;
        out.push_back(LB + s / NC);
        out.push_back(VB + (s % NC) / TC);
        if (s % TC) out.push_back(TB + s % TC);
        return;
    }
    const nfkc::Dec* lo = nfkc::kDec;
    const nfkc::Dec* hi = lo + sizeof(nfkc::kDec) / sizeof(nfkc::kDec[0]);
    const nfkc::Dec* it = std::lower_bound(lo, hi, cp, [](const nfkc::Dec& e, uint32_t v) { return e.cp < v; });
    if (it != hi && it->cp == cp) out.insert(out.end(), nfkc::kPool + it->off, nfkc::kPool + it->off + it->len);   // already a full decomposition
    else out.push_back(cp);
}

bool compose_pair(uint32_t a, uint32_t b, uint32_t& c) {
    if (a >= LB && a < LB + LC && b >= VB && b < VB + VC) { c = SB + ((a - LB) * VC + (b - VB)) * TC; return true; }
    if (a >= SB && a < SB + SC && (a - SB) % TC == 0 && b > TB && b < TB + TC) { c = a + (b - TB); return true; }
    const nfkc::Comp* lo = nfkc::kComp;
    const nfkc::Comp* hi = lo + sizeof(nfkc::kComp) / sizeof(nfkc::kComp[0]);
    const nfkc::Comp* it = std::lower_bound(lo, hi, std::make_pair(a, b), [](const nfkc::Comp& e, const std::pair<uint32_t, uint32_t>& v) {
        return e.a < v.first || (e.a == v.first && e.b < v.second);
    });
    if (it != hi && it->a == a && it->b == b) { c = it->c; return true; }
    return false;
}

std::vector<uint32_t> nfkc_runes(const std::vector<uint32_t>& in) {
    std::vector<uint32_t> d;
    d.reserve(in.size() + 8);
    for (uint32_t cp : in) decompose(cp, d);
    // canonical ordering: stable sort of every run of non-starters by combining class
    for (size_t i = 0; i < d.size();) {
        if (ccc_of(d[i]) == 0) { i++; continue; }
        size_t j = i;
        while (j < d.size() && ccc_of(d[j]) != 0) j++;
        std::stable_sort(d.begin() + (long)i, d.begin() + (long)j, [](uint32_t x, uint32_t y) { return ccc_of(x) < ccc_of(y); });
        i = j;
    }
    // canonical composition (UAX #15): a character combines with the last starter unless blocked by an intervening character of
    // the same or a higher class
    std::vector<uint32_t> out;
    out.reserve(d.size());
    long starter = -1;
    int last_ccc = -1;
    for (uint32_t cp : d) {
        const int cc = ccc_of(cp);
        uint32_t c;
        if (starter >= 0 && (last_ccc < cc || last_ccc == -1) && compose_pair(out[(size_t)starter], cp, c)) {
            out[(size_t)starter] = c;
            continue;
        }
        if (cc == 0) { starter = (long)out.size(); last_ccc = -1; }
        else last_ccc = cc;
        out.push_back(cp);
    }
    return out;
}

// ---- minimal protobuf reader for ModelProto.pieces ----
struct Piece { std::string piece; float score = 0.f; int type = 1; };

bool read_varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    v = 0;
    for (int shift = 0; p < end && shift < 64; shift += 7) {
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

bool skip_field(const uint8_t*& p, const uint8_t* end, int wire) {
    uint64_t v;
    switch (wire) {
        case 0: return read_varint(p, end, v);
        case 1: if (end - p < 8) return false; p += 8; return true;
        case 2: if (!read_varint(p, end, v) || (uint64_t)(end - p) < v) return false; p += v; return true;
        case 5: if (end - p < 4) return false; p += 4; return true;
        default: return false;
    }
}

std::vector<Piece> parse_model(const uint8_t* data, size_t len) {
    std::vector<Piece> pieces;
    const uint8_t* p = data;
    const uint8_t* end = data + len;
    while (p < end) {
        uint64_t key;
        if (!read_varint(p, end, key)) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated varint");
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 1 && wire == 2) {   // repeated SentencePiece pieces = 1
            uint64_t n;
            if (!read_varint(p, end, n) || (uint64_t)(end - p) < n) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated piece");
            const uint8_t* q = p;
            const uint8_t* qe = p + n;
            p = qe;
            Piece pc;
            while (q < qe) {
                uint64_t k2;
                if (!read_varint(q, qe, k2)) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated piece field");
                const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 1 && w2 == 2) {
                    uint64_t sl;
                    if (!read_varint(q, qe, sl) || (uint64_t)(qe - q) < sl) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated piece string");
                    pc.piece.assign((const char*)q, (size_t)sl);
                    q += sl;
                } else if (f2 == 2 && w2 == 5) {
                    if (qe - q < 4) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated score");
                    std::memcpy(&pc.score, q, 4);
                    q += 4;
                } else if (f2 == 3 && w2 == 0) {
                    uint64_t t;
                    if (!read_varint(q, qe, t)) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: truncated type");
                    pc.type = (int)t;
                } else if (!skip_field(q, qe, w2)) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: bad piece field");
            }
            pieces.push_back(std::move(pc));
        } else if (!skip_field(p, end, wire)) {
            throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: bad field");
        }
    }
    return pieces;
}

}  // namespace

// ModelProto.SentencePiece.Type
enum { SP_NORMAL = 1, SP_UNKNOWN = 2, SP_CONTROL = 3, SP_USER_DEFINED = 4 };

struct Tokenizer {
    struct Node {
        float score = 0.f;
        int32_t index = 0;
        int level = 0;
        bool end = false;
        std::unordered_map<uint32_t, int> children;   // rune -> node index
    };
    std::vector<Node> nodes;
    int32_t unknown = 0;
    std::map<std::string, int32_t> control_words;
    size_t n_pieces = 0;

    void insert(const std::string& word, float score, int32_t index) {   // :102-122
        const std::vector<uint32_t> rs = to_runes(word);
        int node = 0;
        for (size_t i = 0; i < rs.size(); i++) {
            int child;
            auto it = nodes[(size_t)node].children.find(rs[i]);
            if (it == nodes[(size_t)node].children.end()) {
                child = (int)nodes.size();
                nodes.emplace_back();
                nodes[(size_t)child].level = nodes[(size_t)node].level + 1;
                nodes[(size_t)node].children[rs[i]] = child;
            } else child = it->second;
            if (i + 1 == rs.size()) { nodes[(size_t)child].end = true; nodes[(size_t)child].score = score; nodes[(size_t)child].index = index; }
            node = child;
        }
    }

    std::vector<int64_t> encode(const std::string& text) const {
        std::vector<int64_t> ids;
        if (text.empty()) return ids;                                     // :86-88
        // spNormalize (:262-277)
        std::vector<uint32_t> mapped;
        for (uint32_t r : to_runes(text)) {
            if (is_control(r) || r == 0) continue;
            mapped.push_back(is_space(r) ? (uint32_t)' ' : r);
        }
        std::vector<uint32_t> norm = nfkc_runes(mapped);
        // spToRunes (:279-292): the check looks at the first rune of the normalised text
        std::vector<uint32_t> runes;
        runes.reserve(norm.size() + 1);
        if (norm.empty() || norm[0] != 0x2581) runes.push_back(0x2581);
        runes.insert(runes.end(), norm.begin(), norm.end());
        for (uint32_t& r : runes) if (is_space(r)) r = 0x2581;           // spReplaceWhitespace (:294-300)
        // viterbiForward (:168-197)
        const size_t n = runes.size() + 1;
        const float min_score = -FLT_MAX;
        struct Slice { float score; int32_t idx; long start; long end; };
        std::vector<float> scores(n, min_score);
        std::vector<Slice> slices(n, Slice{0.f, unknown, -1, 0});
        scores[0] = 0.0f;
        for (size_t i = 0; i < runes.size(); i++) {
            int node = 0;
            for (size_t j = i; j < runes.size(); j++) {                   // commonPrefixSearch (:147-166)
                auto it = nodes[(size_t)node].children.find(runes[j]);
                if (it == nodes[(size_t)node].children.end()) break;
                node = it->second;
                const Node& nd = nodes[(size_t)node];
                if (nd.end) {
                    const float local = scores[i] + nd.score;
                    const size_t end = i + (size_t)nd.level;
                    if (local > scores[end]) { slices[end] = Slice{local, nd.index, (long)i, (long)end}; scores[end] = local; }
                }
            }
            if (scores[i + 1] <= min_score) {
                slices[i + 1] = Slice{min_score, unknown, (long)i, (long)i + 1};
                scores[i + 1] = 0.0f;
            }
        }
        // viterbiBackward (:199-217)
        std::vector<int32_t> rev;
        for (long idx = (long)n - 1; idx >= 0;) {
            const Slice& s = slices[(size_t)idx];
            if (s.start == -1) break;
            rev.push_back(s.idx);
            idx = s.start;
        }
        bool prev_unknown = false;                                        // :124-139
        for (auto it = rev.rbegin(); it != rev.rend(); ++it) {
            if (!(prev_unknown && *it == unknown)) ids.push_back(*it);
            prev_unknown = *it == unknown;
        }
        return ids;
    }
};

Tokenizer* tokenizer_from_bytes(const void* data, size_t len) {
    if (!data || len == 0) throw Error(PTTS_EINVAL, "tokenizer model data must not be empty");
    std::vector<Piece> pieces = parse_model((const uint8_t*)data, len);
    if (pieces.empty()) throw Error(PTTS_EFORMAT, "unmarshal sentencepiece model: no pieces");
    std::unique_ptr<Tokenizer> t(new Tokenizer());
    t->nodes.emplace_back();
    t->n_pieces = pieces.size();
    for (size_t i = 0; i < pieces.size(); i++) {   // :42-51
        const Piece& p = pieces[i];
        if (p.type == SP_NORMAL || p.type == SP_USER_DEFINED) t->insert(p.piece, p.score, (int32_t)i);
        else if (p.type == SP_UNKNOWN) t->unknown = (int32_t)i;
        else if (p.type == SP_CONTROL) t->control_words[p.piece] = (int32_t)i;
    }
    return t.release();
}

Tokenizer* tokenizer_from_path(const std::string& path) {
    if (path.empty()) throw Error(PTTS_EINVAL, "tokenizer model path must not be empty");   // sentencepiece.go:20-22
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Error(PTTS_EIO, "load sentencepiece model \"" + path + "\": cannot open file");
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        return tokenizer_from_bytes(buf.data(), buf.size());
    } catch (const Error& e) {
        throw Error(e.code, "load sentencepiece model \"" + path + "\": " + e.what());
    }
}

std::vector<int64_t> tokenizer_encode(const Tokenizer& t, const std::string& text) { return t.encode(text); }
size_t tokenizer_vocab(const Tokenizer& t) { return t.n_pieces; }
void tokenizer_free(Tokenizer* t) { delete t; }

std::string nfkc_utf8(const std::string& s) {   // exposed for the table tests
    std::string out;
    for (uint32_t r : nfkc_runes(to_runes(s))) {
        if (r < 0x80) out.push_back((char)r);
        else if (r < 0x800) { out.push_back((char)(0xC0 | r >> 6)); out.push_back((char)(0x80 | (r & 0x3F))); }
        else if (r < 0x10000) { out.push_back((char)(0xE0 | r >> 12)); out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F))); }
        else { out.push_back((char)(0xF0 | r >> 18)); out.push_back((char)(0x80 | ((r >> 12) & 0x3F))); out.push_back((char)(0x80 | ((r >> 6) & 0x3F))); out.push_back((char)(0x80 | (r & 0x3F))); }
    }
    return out;
}

}  // namespace ptts
