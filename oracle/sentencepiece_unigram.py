"""TEST INFRASTRUCTURE (CPU oracle): restatement of the reference's SentencePiece unigram encoder,
internal/tokenizer/sentencepiece_bytes_wasm.go (= github.com/vikesh-raj/go-sentencepiece-encoder v1.1.1, go.mod:14, which the
native build calls through internal/tokenizer/sentencepiece.go:19-40).  Pure Python; NFKC from unicodedata."""
from __future__ import annotations

import struct
import unicodedata

SEP = 0x2581            # :77
MIN_SCORE = -3.4028234663852886e38   # -math.MaxFloat32 (:76)
NORMAL, UNKNOWN, CONTROL, USER_DEFINED = 1, 2, 3, 4

_CONTROL_CHARS = {   # :222-231
    0x007F, 0x00AD, 0x0600, 0x0601, 0x0602, 0x0603, 0x0604, 0x0605, 0x061C, 0x06DD, 0x070F, 0x08E2, 0x180E, 0x200B, 0x200C, 0x200D, 0x200E, 0x200F,
    0x202A, 0x202B, 0x202C, 0x202D, 0x202E, 0x2060, 0x2061, 0x2062, 0x2063, 0x2064, 0x2066, 0x2067, 0x2068, 0x2069, 0x206A, 0x206B, 0x206C, 0x206D,
    0x206E, 0x206F, 0xFEFF, 0xFFF9, 0xFFFA, 0xFFFB, 0x110BD, 0x110CD, 0x13430, 0x13431, 0x13432, 0x13433, 0x13434, 0x13435, 0x13436, 0x13437, 0x13438,
    0x1BCA0, 0x1BCA1, 0x1BCA2, 0x1BCA3, 0x1D173, 0x1D174, 0x1D175, 0x1D176, 0x1D177, 0x1D178, 0x1D179, 0x1D17A, 0xE0001}

_GO_SPACE = {0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000} | set(range(0x2000, 0x200B))


def is_space(c: int) -> bool:   # unicode.IsSpace
    return c in _GO_SPACE


def is_control(c: int) -> bool:   # :237-260
    if c in (0x20, 0x0A, 0x0D, 0x09):
        return False
    return (c <= 0x1F or 0x80 <= c <= 0x9F or 0xE0020 <= c <= 0xE007F or 0xE000 <= c <= 0xF8FF or 0xF0000 <= c <= 0xFFFFD or
            0x100000 <= c <= 0x10FFFD or 0xD800 <= c <= 0xDFFF or c in _CONTROL_CHARS)


def _f32(x: float) -> float:
    return struct.unpack("<f", struct.pack("<f", x))[0]


def parse_model(data: bytes) -> list[tuple[str, float, int]]:
    """ModelProto.pieces (field 1): (piece, score, type)."""
    def varint(b, i):
        v, s = 0, 0
        while True:
            c = b[i]
            i += 1
            v |= (c & 0x7F) << s
            s += 7
            if not c & 0x80:
                return v, i

    def skip(b, i, wire):
        if wire == 0:
            return varint(b, i)[1]
        if wire == 1:
            return i + 8
        if wire == 2:
            n, i = varint(b, i)
            return i + n
        if wire == 5:
            return i + 4
        raise ValueError("unmarshal sentencepiece model: bad wire type")

    pieces, i = [], 0
    while i < len(data):
        key, i = varint(data, i)
        field, wire = key >> 3, key & 7
        if field == 1 and wire == 2:
            n, i = varint(data, i)
            sub, i = data[i:i + n], i + n
            piece, score, typ, j = "", 0.0, NORMAL, 0
            while j < len(sub):
                k2, j = varint(sub, j)
                f2, w2 = k2 >> 3, k2 & 7
                if f2 == 1 and w2 == 2:
                    ln, j = varint(sub, j)
                    piece, j = sub[j:j + ln].decode("utf-8", "replace"), j + ln
                elif f2 == 2 and w2 == 5:
                    score, j = struct.unpack("<f", sub[j:j + 4])[0], j + 4
                elif f2 == 3 and w2 == 0:
                    typ, j = varint(sub, j)
                else:
                    j = skip(sub, j, w2)
            pieces.append((piece, score, typ))
        else:
            i = skip(data, i, wire)
    return pieces


class Unigram:
    def __init__(self, data: bytes):
        if not data:
            raise ValueError("tokenizer model data must not be empty")
        self.root = {"children": {}, "level": 0, "end": False, "score": 0.0, "index": 0}
        self.unknown = 0
        self.control_words = {}
        self.pieces = parse_model(data)
        for i, (piece, score, typ) in enumerate(self.pieces):   # :42-51
            if typ in (NORMAL, USER_DEFINED):
                self._insert(piece, score, i)
            elif typ == UNKNOWN:
                self.unknown = i
            elif typ == CONTROL:
                self.control_words[piece] = i

    def _insert(self, word, score, index):   # :102-122
        node = self.root
        for i, ch in enumerate(word):
            child = node["children"].get(ch)
            if child is None:
                child = {"children": {}, "level": node["level"] + 1, "end": False, "score": 0.0, "index": 0}
                node["children"][ch] = child
            if i == len(word) - 1:
                child["end"], child["score"], child["index"] = True, score, index
            node = child

    @staticmethod
    def normalize(s: str) -> str:   # :262-277
        mapped = "".join(" " if is_space(ord(c)) else c for c in s if not (is_control(ord(c)) or ord(c) == 0))
        return unicodedata.normalize("NFKC", mapped)

    def encode(self, text: str) -> list[int]:
        if text == "":
            return []
        text = self.normalize(text)
        runes = [ord(c) for c in text]
        if not runes or runes[0] != SEP:   # :279-292
            runes = [SEP] + runes
        runes = [SEP if is_space(r) else r for r in runes]   # :294-300
        n = len(runes) + 1
        scores = [MIN_SCORE] * n
        slices = [(0.0, self.unknown, -1, 0)] * n   # (score, index, start, end)
        scores[0] = 0.0
        for i in range(len(runes)):   # :168-197
            node = self.root
            for j in range(i, len(runes)):
                node = node["children"].get(chr(runes[j]))
                if node is None:
                    break
                if node["end"]:
                    local = _f32(scores[i] + node["score"])
                    end = i + node["level"]
                    if local > scores[end]:
                        slices[end] = (local, node["index"], i, end)
                        scores[end] = local
            if scores[i + 1] <= MIN_SCORE:
                slices[i + 1] = (MIN_SCORE, self.unknown, i, i + 1)
                scores[i + 1] = 0.0
        rev, idx = [], n - 1   # :199-217
        while idx >= 0:
            s = slices[idx]
            if s[2] == -1:
                break
            rev.append(s[1])
            idx = s[2]
        ids, prev_unknown = [], False   # :124-139
        for sp in reversed(rev):
            if not (prev_unknown and sp == self.unknown):
                ids.append(sp)
            prev_unknown = sp == self.unknown
        return ids
