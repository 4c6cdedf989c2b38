// Test infrastructure: mutation fuzzer for the product's .rpw reader (rustpotter_amd/csrc/rpw_reader.cpp), built with
// g++ -fsanitize=address,undefined by tests/test_rpw_fuzz.py.  Seeds are the reference's own .rpw files (tests/golden);
// every mutant must be either parsed or rejected with an error text -- never crash, over-read or allocate without bound.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rp_host.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
    return g_state;
}

int main(int argc, char **argv) {
    if (argc == 3 && std::strcmp(argv[1], "check") == 0) {  // fuzz_rpw check <file>: "parsed" or "rejected: <text>"
        FILE *f = std::fopen(argv[2], "rb");
        if (!f) { std::perror(argv[2]); return 2; }
        std::vector<uint8_t> data;
        uint8_t tmp[65536];
        size_t n;
        while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) data.insert(data.end(), tmp, tmp + n);
        std::fclose(f);
        uint8_t *buf = (uint8_t *)std::malloc(data.size() ? data.size() : 1);  // exact-size copy: ASan sees over-reads
        std::memcpy(buf, data.data(), data.size());
        rp::RpwKind kind; rp::WakewordRefData ref; rp::WakewordModelData model; std::string err;
        if (rp::parse_rpw(buf, data.size(), &kind, &ref, &model, &err)) std::printf("parsed\n");
        else std::printf("rejected: %s\n", err.c_str());
        std::free(buf);
        return 0;
    }
    if (argc < 3) { std::fprintf(stderr, "usage: fuzz_rpw <iterations> <seed.rpw>... | fuzz_rpw check <file>\n"); return 2; }
    const long iters = std::atol(argv[1]);
    long parsed = 0, rejected = 0;
    for (int a = 2; a < argc; ++a) {
        FILE *f = std::fopen(argv[a], "rb");
        if (!f) { std::perror(argv[a]); return 2; }
        std::vector<uint8_t> seed;
        uint8_t tmp[65536];
        size_t n;
        while ((n = std::fread(tmp, 1, sizeof tmp, f)) > 0) seed.insert(seed.end(), tmp, tmp + n);
        std::fclose(f);
        {  // the seed itself must parse
            rp::RpwKind kind; rp::WakewordRefData ref; rp::WakewordModelData model; std::string err;
            if (!rp::parse_rpw(seed.data(), seed.size(), &kind, &ref, &model, &err)) {
                std::fprintf(stderr, "seed %s rejected: %s\n", argv[a], err.c_str());
                return 1;
            }
        }
        {  // a text string that is not UTF-8 is refused (ciborium -> String in the reference)
            const uint8_t key[] = {0x64, 'n', 'a', 'm', 'e'};  // text(4) "name", then the value's text header
            std::vector<uint8_t> m = seed;
            bool hit = false;
            for (size_t i = 0; i + sizeof key + 2 < m.size() && !hit; ++i)
                if (std::memcmp(&m[i], key, sizeof key) == 0 && (m[i + sizeof key] >> 5) == 3 && (m[i + sizeof key] & 31) >= 1 &&
                    (m[i + sizeof key] & 31) < 24) { m[i + sizeof key + 1] = 0xa1; hit = true; }
            if (hit) {
                rp::RpwKind kind; rp::WakewordRefData ref; rp::WakewordModelData model; std::string err;
                if (rp::parse_rpw(m.data(), m.size(), &kind, &ref, &model, &err) || err.find("UTF-8") == std::string::npos) {
                    std::fprintf(stderr, "%s: a name that is not UTF-8 was not refused (%s)\n", argv[a], err.c_str());
                    return 1;
                }
            }
        }
        for (long it = 0; it < iters; ++it) {
            std::vector<uint8_t> m = seed;
            switch (rnd() % 6) {
            case 0: m.resize(rnd() % (m.size() + 1)); break;                                  // truncate
            case 1: for (int k = 0, e = 1 + rnd() % 8; k < e; ++k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 2: for (int k = 0, e = 1 + rnd() % 4; k < e; ++k) m[rnd() % m.size()] = (uint8_t)rnd(); break;
            case 3: {  // corrupt something in the header region, where the map / array / string lengths live
                const size_t span = m.size() < 512 ? m.size() : 512;
                for (int k = 0, e = 1 + rnd() % 4; k < e; ++k) m[rnd() % span] = (uint8_t)rnd();
                break; }
            case 4: {  // a huge length: 0x9b / 0x5b / 0x7b / 0xbb + 8 bytes of 0xff somewhere
                const uint8_t heads[4] = {0x9b, 0x5b, 0x7b, 0xbb};
                const size_t at = rnd() % m.size();
                m[at] = heads[rnd() % 4];
                for (size_t k = 1; k <= 8 && at + k < m.size(); ++k) m[at + k] = 0xff;
                break; }
            default: {  // splice a random slice over another place
                const size_t len = 1 + rnd() % 64, from = rnd() % m.size(), to = rnd() % m.size();
                for (size_t k = 0; k < len && from + k < m.size() && to + k < m.size(); ++k) m[to + k] = m[from + k];
                break; }
            }
            rp::RpwKind kind; rp::WakewordRefData ref; rp::WakewordModelData model; std::string err;
            // an exact-size heap copy so that ASan sees any read past the end
            uint8_t *buf = (uint8_t *)std::malloc(m.size() ? m.size() : 1);
            std::memcpy(buf, m.data(), m.size());
            if (rp::parse_rpw(buf, m.size(), &kind, &ref, &model, &err)) {
                ++parsed;
                // what the detector relies on when it uploads a parsed wakeword
                bool ok = true;
                if (kind == rp::RpwKind::Ref) {
                    ok = ref.mfcc_size > 0 && ref.lens.size() == ref.feats.size() && ref.tnames.size() == ref.feats.size();
                    for (size_t t = 0; ok && t < ref.feats.size(); ++t)
                        ok = ref.lens[t] >= 0 && ref.feats[t].size() == (size_t)ref.lens[t] * (size_t)ref.mfcc_size;
                    if (ok && ref.has_avg) ok = ref.avg_len >= 0 && ref.avg.size() == (size_t)ref.avg_len * (size_t)ref.mfcc_size;
                } else {
                    ok = model.mfcc_size > 0;
                    for (const auto &kv : model.weights) {
                        size_t n = 1;
                        for (size_t d : kv.second.first) {
                            if (d > 0x7fffffffULL || (d && n > (SIZE_MAX / 4) / d)) { ok = false; break; }  // no wrapped products
                            n *= d;
                        }
                        if (n != kv.second.second.size()) ok = false;
                    }
                }
                if (!ok) { std::fprintf(stderr, "accepted an inconsistent file (mutant %ld of %s)\n", it, argv[a]); return 1; }
            }
            else { ++rejected; if (err.empty()) { std::fprintf(stderr, "rejected without an error text\n"); return 1; } }
            std::free(buf);
        }
    }
    std::printf("parsed %ld rejected %ld\n", parsed, rejected);
    return 0;
}
