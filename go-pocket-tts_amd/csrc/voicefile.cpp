// voicefile.cpp -- voice files: the two kinds of safetensors file the reference accepts as a voice, read on the host.
// Follows internal/safetensors/reader.go: InspectVoiceFile / classifyVoiceTensorNames (:107-125, :232-271), LoadVoiceEmbedding
// (:69-85, :219-230), LoadVoiceModelState / loadVoiceModelStateFromStore (:127-155, :273-308); and the consumer side of a model
// state, flowTransformer.initStateFromVoiceModelState / layerStateFromVoiceModule / readVoiceStateOffset / splitVoiceKVCache's
// shape checks (internal/native/flow_transformer.go:451-480, :517-566, :568-590).  Integer / byte work only: nothing here touches
// the GPU (ptts_voice_open in capi.cpp uploads what voice_file_state returns).
#include <cmath>

#include "runtime.h"

namespace ptts {

namespace {

// reader.go:258-271
bool is_model_state_name(const std::string& name) {
    const size_t slash = name.rfind('/');
    if (slash == std::string::npos || slash == 0 || slash == name.size() - 1) return false;
    const std::string key = name.substr(slash + 1);
    return key == "cache" || key == "offset" || key == "current_end";
}

// reader.go:232-256 (names are never empty here: st_parse refuses a file without tensors, as OpenStore's callers do)
int classify(const StFile& f) {
    bool has_prompt = false, has_state = false;
    for (const auto& kv : f.entries) {
        if (kv.first == "audio_prompt") { has_prompt = true; continue; }
        if (is_model_state_name(kv.first)) has_state = true;
    }
    if (has_state) return PTTS_VOICE_FILE_MODEL_STATE;
    if (has_prompt || !f.entries.empty()) return PTTS_VOICE_FILE_EMBEDDING;
    return PTTS_VOICE_FILE_UNKNOWN;
}

const char* kind_name(int k) { return k == PTTS_VOICE_FILE_MODEL_STATE ? "model_state" : k == PTTS_VOICE_FILE_EMBEDDING ? "embedding" : "unknown"; }

std::string shape_str(const std::vector<int64_t>& s) {   // Go's %v of a []int64
    std::string r = "[";
    for (size_t i = 0; i < s.size(); i++) r += (i ? " " : "") + std::to_string((long long)s[i]);
    return r + "]";
}

void load(VoiceFile& v) {
    v.kind = classify(v.st);
    if (v.kind == PTTS_VOICE_FILE_MODEL_STATE) {
        // loadVoiceModelStateFromStore (reader.go:273-308): every tensor of the file must be "<module>/<key>"
        for (const auto& kv : v.st.entries) {
            const std::string& name = kv.first;
            const size_t slash = name.rfind('/');
            if (slash == std::string::npos || slash == 0 || slash == name.size() - 1) {
                v.state_error = strfmt("safetensors: invalid model-state tensor name \"%s\"", name.c_str());
                return;
            }
            const std::string mod = name.substr(0, slash);
            std::string key = name.substr(slash + 1);
            VoiceFile::Tensor t;
            t.shape = kv.second.shape;
            if (key == "current_end") {   // legacy files: the LENGTH of current_end is the offset (reader.go:290-297)
                key = "offset";
                t.shape = {1};
                t.data = {(float)(kv.second.shape.empty() ? 0 : kv.second.shape[0])};
            } else {
                t.data.resize((size_t)kv.second.count());
                v.st.decode_f32(name, t.data.data());
            }
            VoiceFile::Module* m = nullptr;
            for (auto& mm : v.modules) if (mm.name == mod) m = &mm;
            if (!m) { v.modules.emplace_back(); m = &v.modules.back(); m->name = mod; }
            m->tensors[key] = std::move(t);   // entries are walked in sorted order, so modules end up sorted too; a later key of the same name wins, as the Go map does
        }
    } else if (v.kind == PTTS_VOICE_FILE_EMBEDDING) {
        // LoadFirstTensor (reader.go:31-46): names[0] of the sorted names; normalizeVoiceEmbeddingShape (:219-230)
        const auto& first = *v.st.entries.begin();
        const auto& sh = first.second.shape;
        if (sh.size() != 2 && sh.size() != 3) {
            v.emb_error = strfmt("safetensors: voice embedding has %zuD shape %s, expected 2D or 3D", sh.size(), shape_str(sh).c_str());
            return;
        }
        v.emb.resize((size_t)first.second.count());
        v.st.decode_f32(first.first, v.emb.data());
        if (sh.size() == 2) v.emb_shape = {1, sh[0], sh[1]};
        else v.emb_shape = sh;
    }
}

}  // namespace

VoiceFile* voice_file_from_path(const std::string& path) {
    std::unique_ptr<VoiceFile> v(new VoiceFile());
    st_open_path(path, v->st);
    load(*v);
    return v.release();
}

VoiceFile* voice_file_from_bytes(const void* data, size_t len) {
    std::unique_ptr<VoiceFile> v(new VoiceFile());
    v->st.owned.assign((const uint8_t*)data, (const uint8_t*)data + len);
    v->st.data = v->st.owned.data();
    v->st.size = v->st.owned.size();
    st_parse(v->st);
    load(*v);
    return v.release();
}

void voice_file_embedding(const VoiceFile& v, const float** data, int64_t shape[3]) {
    if (v.kind == PTTS_VOICE_FILE_MODEL_STATE)   // reader.go:75-77
        throw Error(PTTS_EFORMAT, "safetensors: voice file contains upstream model state, not a legacy audio_prompt embedding");
    if (!v.emb_error.empty()) throw Error(PTTS_EFORMAT, v.emb_error);
    *data = v.emb.data();
    for (int i = 0; i < 3; i++) shape[i] = v.emb_shape[i];
}

void voice_file_require_state(const VoiceFile& v) {
    if (v.kind != PTTS_VOICE_FILE_MODEL_STATE)   // reader.go:134-137
        throw Error(PTTS_EFORMAT, strfmt("safetensors: voice file kind \"%s\" is not upstream model state", kind_name(v.kind)));
    if (!v.state_error.empty()) throw Error(PTTS_EFORMAT, v.state_error);
}

// flow_transformer.go:451-480 with layerStateFromVoiceModule (:517-552), readVoiceStateOffset (:554-566) and the shape checks of
// splitVoiceKVCache (:568-590); the re-layout [2,B,T,H,D] -> [B,H,T,D] itself happens on the device (k_voice_scatter).
void voice_file_state(const VoiceFile& v, int n_layers, int heads, int head_dim, const float** caches, int64_t* steps, int64_t* offsets) {
    voice_file_require_state(v);
    for (int l = 0; l < n_layers; l++) {
        const std::string mod = "transformer.layers." + std::to_string(l) + ".self_attn";   // flowAttentionModuleName :513-515
        const VoiceFile::Module* m = nullptr;
        for (const auto& mm : v.modules) if (mm.name == mod) m = &mm;
        if (!m) throw Error(PTTS_EINVAL, strfmt("native: voice model state missing module \"%s\"", mod.c_str()));
        auto ci = m->tensors.find("cache"), oi = m->tensors.find("offset");
        if (ci == m->tensors.end()) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" missing cache", mod.c_str()));
        if (oi == m->tensors.end()) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" missing offset", mod.c_str()));
        const VoiceFile::Tensor& off = oi->second;
        if (off.data.empty()) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" has empty offset tensor", mod.c_str()));
        const float ov = off.data[0];
        // int64(v) of a NaN or an out-of-range float is implementation-defined in Go; neither is integral, so refuse them here
        if (!(std::fabs(ov) < 9.0e18f) || (float)(int64_t)ov != ov)
            throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" offset %g is not an integer", mod.c_str(), (double)ov));
        const int64_t o = (int64_t)ov;
        const VoiceFile::Tensor& c = ci->second;
        if (c.shape.size() != 5) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" cache shape %s, want [2,B,T,H,D]", mod.c_str(), shape_str(c.shape).c_str()));
        if (c.shape[0] != 2) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" cache first dim %lld, want 2", mod.c_str(), (long long)c.shape[0]));
        const int64_t b = c.shape[1], T = c.shape[2], H = c.shape[3], D = c.shape[4];
        if (b <= 0 || T < 0 || H <= 0 || D <= 0) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" has invalid cache shape %s", mod.c_str(), shape_str(c.shape).c_str()));
        if (heads != 0 && H != heads) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" heads %lld, want %d", mod.c_str(), (long long)H, heads));
        if (head_dim != 0 && D != head_dim) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" head dim %lld, want %d", mod.c_str(), (long long)D, head_dim));
        if (b != 1) throw Error(PTTS_EINVAL, strfmt("ptts-hip: voice model state module \"%s\" batch %lld, want 1 (a voice conditions one utterance)", mod.c_str(), (long long)b));
        if (o < 0) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" has negative offset %lld", mod.c_str(), (long long)o));
        if (o > T) throw Error(PTTS_EINVAL, strfmt("native: voice model state module \"%s\" offset %lld exceeds cache length %lld", mod.c_str(), (long long)o, (long long)T));
        caches[l] = c.data.data();
        steps[l] = T;
        offsets[l] = o;
    }
}

}  // namespace ptts
