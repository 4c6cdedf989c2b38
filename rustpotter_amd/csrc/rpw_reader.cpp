// rpw_reader.cpp -- reader of rustpotter `.rpw` wakeword files (CBOR written by
// ciborium 0.2.1 through serde: src/wakewords/wakeword_file.rs:10-42).  Definite
// lengths only; floats may be f16/f32/f64 (minimal-width encoding); TensorData.bytes
// is an ARRAY of small uints, not a byte string.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "rp_host.h"

namespace rp {
namespace {

struct Cur {
    const uint8_t *p, *end;
    std::string err;
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool need(size_t n) { return (size_t)(end - p) >= n ? true : fail("unexpected end of wakeword file"); }
    bool head(int *major, int *info, uint64_t *val) {
        if (!need(1)) return false;
        uint8_t ib = *p++;
        *major = ib >> 5; *info = ib & 31; *val = 0;
        if (*info < 24) { *val = (uint64_t)*info; return true; }
        int nb = *info == 24 ? 1 : *info == 25 ? 2 : *info == 26 ? 4 : *info == 27 ? 8 : 0;
        if (nb == 0) return fail("unsupported CBOR length encoding");
        if (!need((size_t)nb)) return false;
        for (int i = 0; i < nb; ++i) *val = (*val << 8) | *p++;
        return true;
    }
};

float half_to_float(uint16_t h) {
    int s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = std::ldexp((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = std::ldexp((float)(m + 1024), e - 25);
    return s ? -v : v;
}

// number (float of any width, or integer) -> float; `is_null` set for null/undefined
bool read_number(Cur &c, float *out, bool *is_null) {
    int mj, info; uint64_t v;
    if (!c.head(&mj, &info, &v)) return false;
    if (is_null) *is_null = false;
    if (mj == 0) { *out = (float)v; return true; }
    if (mj == 1) { *out = -1.f - (float)v; return true; }
    if (mj == 7) {
        if (info == 22 || info == 23) { if (is_null) { *is_null = true; *out = 0.f; return true; } return c.fail("unexpected null"); }
        if (info == 25) { *out = half_to_float((uint16_t)v); return true; }
        if (info == 26) { uint32_t u = (uint32_t)v; std::memcpy(out, &u, 4); return true; }
        if (info == 27) { double d; std::memcpy(&d, &v, 8); *out = (float)d; return true; }
    }
    return c.fail("expected a number");
}
bool read_uint(Cur &c, uint64_t *out) {
    int mj, info; uint64_t v;
    if (!c.head(&mj, &info, &v)) return false;
    if (mj != 0) return c.fail("expected an unsigned integer");
    *out = v; return true;
}
// what Rust's String accepts (core::str::from_utf8): no overlong forms, no surrogates, nothing above U+10FFFF
bool valid_utf8(const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint8_t b = p[i];
        size_t extra; uint32_t cp, lo;
        if (b < 0x80) { ++i; continue; }
        else if ((b & 0xe0) == 0xc0) { extra = 1; cp = b & 0x1f; lo = 0x80; }
        else if ((b & 0xf0) == 0xe0) { extra = 2; cp = b & 0x0f; lo = 0x800; }
        else if ((b & 0xf8) == 0xf0) { extra = 3; cp = b & 0x07; lo = 0x10000; }
        else return false;
        if (extra >= n - i) return false;  // truncated sequence
        for (size_t k = 1; k <= extra; ++k) {
            if ((p[i + k] & 0xc0) != 0x80) return false;
            cp = (cp << 6) | (p[i + k] & 0x3f);
        }
        if (cp < lo || cp > 0x10ffff || (cp >= 0xd800 && cp <= 0xdfff)) return false;
        i += extra + 1;
    }
    return true;
}
bool read_text(Cur &c, std::string *out) {
    int mj, info; uint64_t v;
    if (!c.head(&mj, &info, &v)) return false;
    if (mj != 3) return c.fail("expected a string");
    if (!c.need((size_t)v)) return false;
    if (!valid_utf8(c.p, (size_t)v)) return c.fail("invalid UTF-8 in a string");  // ciborium -> String refuses it too
    out->assign(reinterpret_cast<const char *>(c.p), (size_t)v);
    c.p += v; return true;
}
bool skip_item(Cur &c, int depth = 0) {
    if (depth > 64) return c.fail("CBOR nesting too deep");
    int mj, info; uint64_t v;
    if (!c.head(&mj, &info, &v)) return false;
    switch (mj) {
    case 0: case 1: case 7: return true;
    case 2: case 3: if (!c.need((size_t)v)) return false; c.p += v; return true;
    case 4: for (uint64_t i = 0; i < v; ++i) if (!skip_item(c, depth + 1)) return false; return true;
    case 5: for (uint64_t i = 0; i < 2 * v; ++i) if (!skip_item(c, depth + 1)) return false; return true;
    case 6: return skip_item(c, depth + 1);
    }
    return c.fail("unsupported CBOR item");
}
// array[array[f32]] -> flat rows; all rows must have the same width
bool read_matrix(Cur &c, std::vector<float> *out, int *rows, int *cols, bool *is_null) {
    int mj, info; uint64_t v;
    if (!c.head(&mj, &info, &v)) return false;
    if (mj == 7 && (info == 22 || info == 23) && is_null) { *is_null = true; *rows = *cols = 0; return true; }
    if (is_null) *is_null = false;
    if (mj != 4) return c.fail("expected an array of mfcc frames");
    if (v > 0x7fffffffULL) return c.fail("mfcc matrix too large");
    *rows = (int)v; *cols = -1; out->clear();
    for (uint64_t r = 0; r < v; ++r) {
        int mj2, info2; uint64_t n;
        if (!c.head(&mj2, &info2, &n)) return false;
        if (mj2 != 4) return c.fail("expected an array of mfcc coefficients");
        if (n > 0x7fffffffULL) return c.fail("mfcc matrix too large");
        if (*cols < 0) *cols = (int)n; else if (*cols != (int)n) return c.fail("ragged mfcc matrix");
        for (uint64_t i = 0; i < n; ++i) { float f; if (!read_number(c, &f, nullptr)) return false; out->push_back(f); }
    }
    if (*cols < 0) *cols = 0;
    return true;
}
bool read_opt_float(Cur &c, bool *has, float *val) {
    bool is_null = false;
    if (!read_number(c, val, &is_null)) return false;
    *has = !is_null; return true;
}

}  // namespace

bool parse_rpw(const uint8_t *buf, size_t len, RpwKind *kind, WakewordRefData *ref, WakewordModelData *model,
               std::string *err) {
    Cur c{buf, buf + len, {}};
    int mj, info; uint64_t n;
    if (!c.head(&mj, &info, &n) || mj != 5) { *err = "invalid wakeword file: expected a CBOR map"; return false; }
    bool seen_enabled = false, seen_mfcc_size = false, seen_samples = false, seen_labels = false, seen_name = false,
         seen_rms = false, seen_weights = false, seen_train = false, seen_mtype = false;
    for (uint64_t e = 0; e < n; ++e) {
        std::string key;
        if (!read_text(c, &key)) { *err = c.err; return false; }
        bool ok = true;
        if (key == "name") { ok = read_text(c, &ref->name); seen_name = true; }
        else if (key == "avg_features") {
            bool is_null = false; int rows, cols;
            ok = read_matrix(c, &ref->avg, &rows, &cols, &is_null);
            ref->has_avg = ok && !is_null && rows > 0; ref->avg_len = ref->has_avg ? rows : 0;
        } else if (key == "samples_features") {
            int mj2, info2; uint64_t m;
            ok = c.head(&mj2, &info2, &m) && (mj2 == 5 || c.fail("samples_features: expected a map"));
            for (uint64_t i = 0; ok && i < m; ++i) {
                std::string tn; std::vector<float> mat; int rows = 0, cols = 0;
                ok = read_text(c, &tn) && read_matrix(c, &mat, &rows, &cols, nullptr);
                if (ok) { ref->tnames.push_back(tn); ref->lens.push_back(rows); ref->feats.push_back(std::move(mat));
                          if (i == 0) ref->mfcc_size = seen_mfcc_size ? ref->mfcc_size : cols; }
            }
            seen_samples = true;
        } else if (key == "threshold") ok = read_opt_float(c, &ref->has_threshold, &ref->threshold);
        else if (key == "avg_threshold") ok = read_opt_float(c, &ref->has_avg_threshold, &ref->avg_threshold);
        else if (key == "rms_level") { float f; ok = read_number(c, &f, nullptr); ref->rms_level = model->rms_level = f; seen_rms = true; }
        else if (key == "mfcc_size") { uint64_t v; ok = read_uint(c, &v); ref->mfcc_size = model->mfcc_size = (int)v; seen_mfcc_size = true; }
        else if (key == "enabled") { ok = skip_item(c); seen_enabled = true; }
        else if (key == "labels") {
            int mj2, info2; uint64_t m;
            ok = c.head(&mj2, &info2, &m) && (mj2 == 4 || c.fail("labels: expected an array"));
            for (uint64_t i = 0; ok && i < m; ++i) { std::string s; ok = read_text(c, &s); if (ok) model->labels.push_back(s); }
            seen_labels = true;
        } else if (key == "train_size") { uint64_t v; ok = read_uint(c, &v); model->train_size = (size_t)v; seen_train = true; }
        else if (key == "m_type") { ok = read_text(c, &model->m_type); seen_mtype = true; }
        else if (key == "weights") {
            int mj2, info2; uint64_t m;
            ok = c.head(&mj2, &info2, &m) && (mj2 == 5 || c.fail("weights: expected a map"));
            for (uint64_t i = 0; ok && i < m; ++i) {
                std::string wn;
                ok = read_text(c, &wn);
                int mj3, info3; uint64_t nf;
                ok = ok && c.head(&mj3, &info3, &nf) && (mj3 == 5 || c.fail("TensorData: expected a map"));
                std::vector<uint8_t> bytes; std::vector<size_t> dims; std::string dtype;
                for (uint64_t f = 0; ok && f < nf; ++f) {
                    std::string fk;
                    ok = read_text(c, &fk);
                    if (!ok) break;
                    if (fk == "bytes") {
                        int mj4, info4; uint64_t nb;
                        ok = c.head(&mj4, &info4, &nb);
                        if (ok && mj4 == 2) { ok = c.need((size_t)nb); if (ok) { bytes.assign(c.p, c.p + nb); c.p += nb; } }
                        else if (ok && mj4 == 4) { bytes.reserve((size_t)std::min<uint64_t>(nb, (uint64_t)(c.end - c.p))); for (uint64_t b = 0; ok && b < nb; ++b) { uint64_t v; ok = read_uint(c, &v); bytes.push_back((uint8_t)v); } }
                        else if (ok) ok = c.fail("TensorData.bytes: expected an array");
                    } else if (fk == "dims") {
                        int mj4, info4; uint64_t nd;
                        ok = c.head(&mj4, &info4, &nd) && (mj4 == 4 || c.fail("TensorData.dims: expected an array"));
                        for (uint64_t d = 0; ok && d < nd; ++d) {
                            uint64_t v; ok = read_uint(c, &v);
                            if (ok && v > 0x7fffffffULL) ok = c.fail("TensorData.dims: dimension too large");  // the detector keeps dims as int
                            dims.push_back((size_t)v);
                        }
                    } else if (fk == "d_type") ok = read_text(c, &dtype);
                    else ok = skip_item(c);
                }
                if (ok) {
                    if (dtype != "f32") { ok = c.fail("unsupported tensor d_type (only f32)"); break; }
                    // element count with an overflow check (a wrapped product once passed the length test below)
                    size_t cnt = 1; bool ovf = false;
                    for (size_t d : dims) { if (d && cnt > (SIZE_MAX / 4) / d) ovf = true; else cnt *= d; }
                    if (ovf || bytes.size() != cnt * 4) { ok = c.fail("tensor byte length does not match dims"); break; }
                    std::vector<float> data(cnt);
                    std::memcpy(data.data(), bytes.data(), cnt * 4);  // little-endian f32
                    model->weights[wn] = std::make_pair(dims, std::move(data));
                }
            }
            seen_weights = true;
        } else ok = skip_item(c);  // serde ignores unknown fields
        if (!ok) { *err = "invalid wakeword file: " + c.err; return false; }
    }
    // fall-through of src/detector.rs:152-176, decided on the fields present
    const bool ref_common = seen_name && seen_samples && seen_rms;
    if (ref_common && (seen_enabled || seen_mfcc_size)) {
        if (ref->tnames.empty()) { *err = "invalid wakeword file: no sample features"; return false; }
        if (!seen_mfcc_size) ref->mfcc_size = (int)(ref->lens[0] ? ref->feats[0].size() / (size_t)ref->lens[0] : 0);  // wakeword_v2.rs:22
        for (size_t t = 0; t < ref->feats.size(); ++t)
            if (ref->lens[t] == 0 || ref->feats[t].size() != (size_t)ref->lens[t] * (size_t)ref->mfcc_size) {
                *err = "invalid wakeword file: template width differs from mfcc_size"; return false;
            }
        if (ref->has_avg && ref->avg.size() != (size_t)ref->avg_len * (size_t)ref->mfcc_size) {
            *err = "invalid wakeword file: avg_features width differs from mfcc_size"; return false;
        }
        *kind = RpwKind::Ref;
        return true;
    }
    if (seen_labels && seen_train && seen_mfcc_size && seen_mtype && seen_weights && seen_rms) { *kind = RpwKind::Model; return true; }
    *err = "invalid wakeword file: neither a wakeword reference nor a wakeword model";
    return false;
}

}  // namespace rp
