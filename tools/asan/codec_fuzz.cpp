// AddressSanitizer / UBSan driver for the host-side byte codecs of libecseg_hip (csrc/host_codec.cpp): the TIFF LZW
// decoder parses untrusted files (inputs globbed by get_imgs).  CPU build only (`make asan`); GPU sanitizers are not
// available on the target pool.  Corpus: round trips of structured / random buffers, then truncated, bit-flipped and
// random streams decoded into exact, short and oversized destinations.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" long long ecseg_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap);
extern "C" long long ecseg_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap);
extern "C" int ecseg_tiff_write_gray8(const char* path, const uint8_t* img, int H, int W, int invert);
extern "C" int ecseg_tiff_info(const char* path, int* H, int* W, int* spp, int* bits);
extern "C" int ecseg_tiff_read(const char* path, void* dst, long long dst_bytes);
extern "C" int ecseg_npy_write_i64(const char* path, const uint8_t* labels, int H, int W);
extern "C" int ecseg_npy_label_info(const char* path, int* H, int* W);
extern "C" int ecseg_npy_read_labels_u8(const char* path, uint8_t* dst, int H, int W);
extern "C" int ecseg_png_write_channel(const char* path, const uint8_t* px, int H, int W, int channels, int channel, int invert);

static uint64_t s = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }

int main() {
    long long checked = 0;
    for (int round = 0; round < 400; ++round) {
        const int n = round < 8 ? round : (int)(rnd() % 20000);
        std::vector<uint8_t> raw(n + 1);       // (+1: data() of an empty vector is null, which the ABI rejects)
        const int kind = round % 4;
        for (int i = 0; i < n; ++i)
            raw[i] = kind == 0 ? 0 : kind == 1 ? (uint8_t)rnd() : kind == 2 ? (uint8_t)((i / 37) & 3) : (uint8_t)(i * 7);
        std::vector<uint8_t> enc(2 * n + 64);
        const long long m = ecseg_lzw_encode(raw.data(), n, enc.data(), (long long)enc.size());
        if (m < 0) { std::printf("encode failed at round %d\n", round); return 1; }
        std::vector<uint8_t> dec(n + 1);
        const long long k = ecseg_lzw_decode(enc.data(), m, dec.data(), n);
        if (k != n || std::memcmp(dec.data(), raw.data(), n) != 0) { std::printf("round trip mismatch at round %d\n", round); return 1; }
        // encoder with a destination that is too small must fail cleanly
        if (m > 2) { std::vector<uint8_t> small(m / 2); (void)ecseg_lzw_encode(raw.data(), n, small.data(), (long long)small.size()); }
        for (int t = 0; t < 24; ++t) {
            std::vector<uint8_t> bad(enc.begin(), enc.begin() + (m ? 1 + rnd() % m : 0)); bad.push_back(0); bad.pop_back();
            const int flips = rnd() % 6;
            for (int f = 0; f < flips && !bad.empty(); ++f) bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() % 8));
            const long long cap = t % 3 == 0 ? n : t % 3 == 1 ? (long long)(rnd() % (n + 1)) : n + (long long)(rnd() % 4096);
            std::vector<uint8_t> out((size_t)cap + 1, 0xAB);
            const long long r = ecseg_lzw_decode(bad.data(), (long long)bad.size(), out.data(), cap);
            if (r > cap) { std::printf("decoder reported %lld bytes for a %lld-byte destination\n", r, cap); return 1; }
            if (out[(size_t)cap] != 0xAB) { std::printf("decoder wrote past its destination\n"); return 1; }
            ++checked;
        }
        std::vector<uint8_t> junk(rnd() % 4096);
        for (auto& b : junk) b = (uint8_t)rnd();
        std::vector<uint8_t> out(5000);
        (void)ecseg_lzw_decode(junk.data(), (long long)junk.size(), out.data(), (long long)out.size());
    }
    (void)ecseg_lzw_decode(nullptr, 0, nullptr, 0);
    // ---- hand-made streams the repo's own encoder never writes: literal-only strips without a leading ClearCode,
    //      with two ClearCodes in a row, ClearCodes sprinkled in; fixed 9-bit codes (< 250 codes: the width never changes) ----
    for (int round = 0; round < 200; ++round) {
        const int nlit = 1 + (int)(rnd() % 200);
        std::vector<int> codes; std::vector<uint8_t> want;
        if (round % 3 != 0) codes.push_back(256);
        if (round % 3 == 2) codes.push_back(256);
        for (int i = 0; i < nlit; ++i) {
            if (round % 5 == 4 && rnd() % 16 == 0) { codes.push_back(256); if (rnd() & 1) codes.push_back(256); }
            const uint8_t b = (uint8_t)rnd(); codes.push_back(b); want.push_back(b);
        }
        codes.push_back(257);
        std::vector<uint8_t> bits((codes.size() * 9 + 7) / 8 + (round & 1 ? 16 : 0), 0);
        size_t bp = 0;
        for (int c : codes) for (int b = 8; b >= 0; --b, ++bp) if ((c >> b) & 1) bits[bp >> 3] |= (uint8_t)(0x80u >> (bp & 7));
        std::vector<uint8_t> out(want.size() + 1, 0xAB);
        const long long r = ecseg_lzw_decode(bits.data(), (long long)bits.size(), out.data(), (long long)want.size());
        // (codes < 256 add table entries: with ClearCodes in between fewer than 250 are ever live, the width stays 9)
        if (r != (long long)want.size() || std::memcmp(out.data(), want.data(), want.size()) != 0 || out[want.size()] != 0xAB) {
            std::printf("hand-made literal stream %d: %lld of %zu bytes or wrong bytes\n", round, r, want.size()); return 1;
        }
        ++checked;
    }
    // ---- whole-file TIFF reader (csrc/host_io.cpp): files written by our own writer, then truncated / bit-flipped /
    //      overwritten with random tag values; the reader must return a status and never touch memory outside dst ----
    long long files = 0;
    const char* path = "/tmp/ecseg_codec_fuzz.tif";
    for (int round = 0; round < 300; ++round) {
        const int H = 1 + (int)(rnd() % 40), W = 1 + (int)(rnd() % 300);
        std::vector<uint8_t> img((size_t)H * W);
        for (auto& b : img) b = (uint8_t)(rnd() % (round % 3 ? 4 : 256));
        if (ecseg_tiff_write_gray8(path, img.data(), H, W, round & 1) != 0) { std::printf("tiff write failed\n"); return 1; }
        int h = 0, w = 0, spp = 0, bits = 0;
        if (ecseg_tiff_info(path, &h, &w, &spp, &bits) != 0 || h != H || w != W || spp != 1 || bits != 8) { std::printf("tiff info mismatch\n"); return 1; }
        std::vector<uint8_t> back((size_t)H * W + 1, 0xCD);
        if (ecseg_tiff_read(path, back.data(), (long long)H * W) != 0) { std::printf("tiff read failed\n"); return 1; }
        for (size_t i = 0; i < img.size(); ++i)
            if (back[i] != (uint8_t)((round & 1) ? 255 - img[i] : img[i])) { std::printf("tiff round trip mismatch\n"); return 1; }
        if (back[img.size()] != 0xCD) { std::printf("tiff reader wrote past its destination\n"); return 1; }
        std::vector<uint8_t> file;
        { FILE* f = std::fopen(path, "rb"); std::fseek(f, 0, SEEK_END); file.resize((size_t)std::ftell(f)); std::fseek(f, 0, SEEK_SET); if (std::fread(file.data(), 1, file.size(), f) != file.size()) return 1; std::fclose(f); }
        for (int t = 0; t < 12; ++t) {
            std::vector<uint8_t> bad = file;
            if (t % 3 == 0) bad.resize(rnd() % (bad.size() + 1));
            const int flips = 1 + rnd() % 8;
            for (int k = 0; k < flips && !bad.empty(); ++k) {
                const size_t at = t % 3 == 2 ? bad.size() - 1 - rnd() % (bad.size() < 160 ? bad.size() : 160) : rnd() % bad.size();   // t % 3 == 2: aim at the IFD
                bad[at] = (uint8_t)rnd();
            }
            { FILE* f = std::fopen(path, "wb"); if (!bad.empty() && std::fwrite(bad.data(), 1, bad.size(), f) != bad.size()) return 1; std::fclose(f); }
            int hh = 0, ww = 0, ss = 0, bb = 0;
            if (ecseg_tiff_info(path, &hh, &ww, &ss, &bb) == 0) {
                const size_t need = (size_t)hh * ww * ss * (bb / 8);
                if (need <= (64u << 20)) {
                    std::vector<uint8_t> out(need + 1, 0xEF);
                    (void)ecseg_tiff_read(path, out.data(), (long long)need);
                    if (out[need] != 0xEF) { std::printf("tiff reader wrote past its destination (corrupt file)\n"); return 1; }
                    if (need > 16) (void)ecseg_tiff_read(path, out.data(), (long long)need / 2);     // too small a destination: must refuse
                }
            }
            ++files;
        }
    }
    // .npy label files (round 5: labels/<stem>.npy is parsed natively by `make meta_overlay`): a valid file, then headers with
    // flipped / truncated / random bytes - every outcome but a crash or a write past the destination is acceptable
    long long npy_files = 0;
    {
        const int H = 37, W = 53;
        std::vector<uint8_t> lab((size_t)H * W), back((size_t)H * W + 1, 0xCD);
        for (auto& v : lab) v = (uint8_t)(rnd() & 3);
        if (ecseg_npy_write_i64(path, lab.data(), H, W) != 0) { std::printf("npy write failed\n"); return 1; }
        int h = 0, w = 0;
        if (ecseg_npy_label_info(path, &h, &w) != 0 || h != H || w != W) { std::printf("npy info mismatch\n"); return 1; }
        if (ecseg_npy_read_labels_u8(path, back.data(), H, W) != 0 || std::memcmp(back.data(), lab.data(), lab.size()) != 0) { std::printf("npy round trip mismatch\n"); return 1; }
        std::vector<uint8_t> file;
        { FILE* f = std::fopen(path, "rb"); uint8_t b[4096]; size_t k; while ((k = std::fread(b, 1, sizeof b, f)) > 0) file.insert(file.end(), b, b + k); std::fclose(f); }
        for (int round = 0; round < 3000; ++round) {
            std::vector<uint8_t> bad(file);
            const int kind = round % 3;
            if (kind == 0) for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) bad[rnd() % 140] ^= (uint8_t)(1u << (rnd() % 8));     // header bit flips
            else if (kind == 1) bad.resize(rnd() % bad.size());                                                             // truncation
            else for (int k = 0; k < 24; ++k) bad[6 + rnd() % 120] = (uint8_t)rnd();                                        // random header bytes
            { FILE* f = std::fopen(path, "wb"); std::fwrite(bad.data(), 1, bad.size(), f); std::fclose(f); }
            int hh = 0, ww = 0;
            if (ecseg_npy_label_info(path, &hh, &ww) == 0 && hh > 0 && ww > 0 && (long long)hh * ww <= (1 << 22)) {
                std::vector<uint8_t> out((size_t)hh * ww + 1, 0xEF);
                (void)ecseg_npy_read_labels_u8(path, out.data(), hh, ww);
                if (out[(size_t)hh * ww] != 0xEF) { std::printf("npy reader wrote past its destination\n"); return 1; }
            }
            (void)ecseg_npy_read_labels_u8(path, back.data(), H, W);
            if (back[(size_t)H * W] != 0xCD) { std::printf("npy reader wrote past its destination (fixed shape)\n"); return 1; }
            ++npy_files;
        }
        // the channel PNG coder on random shapes (reads strided pixels, writes a worst-case-sized buffer)
        for (int round = 0; round < 200; ++round) {
            const int hh = 1 + (int)(rnd() % 40), ww = 1 + (int)(rnd() % 600), ch = 1 + (int)(rnd() % 4);
            std::vector<uint8_t> px((size_t)hh * ww * ch);
            const int mode = round % 3;
            for (auto& v : px) v = mode == 0 ? (uint8_t)rnd() : mode == 1 ? (uint8_t)0 : (uint8_t)((rnd() % 50) ? 9 : rnd());
            if (ecseg_png_write_channel(path, px.data(), hh, ww, ch, (int)(rnd() % ch), round & 1) != 0) { std::printf("png channel write failed\n"); return 1; }
        }
    }
    std::remove(path);
    std::printf("npy_fuzz ok: %lld corrupt files\n", npy_files);
    std::printf("tiff_fuzz ok: %lld corrupt files\n", files);
    std::printf("codec_fuzz ok: %lld corrupt streams\n", checked);
    return 0;
}
