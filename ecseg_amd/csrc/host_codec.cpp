// Host-side byte codecs used by the file I/O around the hot path (no GPU work): TIFF-flavoured LZW
// (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, "early change"), as written by OpenCV's imwrite for the
// reference's dapi/<name>.tif outputs and found in typical microscope TIFF inputs (reference src/utils.py:110,123).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ecseg_hip.h"

extern "C" {

// Decodes one LZW strip.  Returns the number of bytes written (<= dst_cap) or -1 on a corrupt stream.
// Every table entry is remembered as (offset, length) of an occurrence of its string in the OUTPUT written so far: the
// string of a new entry (previous string + first byte of the current one) starts where the previous code was emitted,
// so emitting a code is one forward copy from earlier output instead of a backward walk along a prefix chain.
long long ecseg_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0 || dst_cap < 0) return -1;
    struct Entry { long long pos; int32_t len; };      // pos < 0: a literal byte (codes 0..255)
    Entry tab[4096];
    for (int i = 0; i < 256; ++i) tab[i] = Entry{-1, 1};
    int next = 258, width = 9;
    long long out = 0;
    uint64_t acc = 0;
    int nbits = 0;
    long long pos = 0;
    int old = -1;
    long long old_pos = 0;                              // where the previous code's string starts in dst
    for (;;) {
        while (nbits < width) {
            if (pos >= n) return out;                   // stream ended without EOI: accept what we have
            acc = (acc << 8) | src[pos++];
            nbits += 8;
        }
        const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; old = -1; continue; }
        if (old < 0) {
            if (code >= 256) return -1;
            if (out >= dst_cap) return out;             // truncated output buffer: write what fits
            old_pos = out;
            dst[out++] = (uint8_t)code;
            old = code;
            continue;
        }
        const int old_len = tab[old].len;
        long long cpos; int clen;                       // the string of `code`
        if (code < next) {
            if (code >= 256 && code < 258) return -1;
            cpos = tab[code].pos; clen = tab[code].len;
        } else if (code == next && next < 4096) {      // KwKwK: previous string + its own first byte
            cpos = old_pos; clen = old_len + 1;
        } else {
            return -1;
        }
        const long long start = out;
        const long long room = dst_cap - out;
        const int ncopy = (long long)clen <= room ? clen : (int)room;
        if (cpos < 0) { if (ncopy > 0) dst[out] = (uint8_t)code; }
        else for (int k = 0; k < ncopy; ++k) dst[out + k] = dst[cpos + k];     // may overlap forward (KwKwK): byte by byte
        out += ncopy;
        if (next < 4096) { tab[next] = Entry{old_pos, old_len + 1}; ++next; }   // previous string + first byte of this one
        if (ncopy < clen) return out;                   // destination full
        old = code; old_pos = start;
        if (next >= (1 << width) - 1 && width < 12) ++width;   // early change
    }
    return out;
}

// Encodes one strip.  Returns the number of bytes written, or -1 when dst_cap is too small
// (n * 2 + 16 bytes always suffice).
long long ecseg_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0) return -1;
    // hash table: key = (prefix code << 8) | byte -> code
    const int HSIZE = 9001;
    std::vector<int32_t> hkey(HSIZE), hval(HSIZE);
    auto clear = [&]() { std::fill(hkey.begin(), hkey.end(), -1); };
    long long out = 0;
    uint64_t acc = 0;
    int nbits = 0, width = 9, next = 258;
    bool overflow = false;
    auto put = [&](int code) {
        acc = (acc << width) | (uint32_t)code;
        nbits += width;
        while (nbits >= 8) {
            if (out < dst_cap) dst[out] = (uint8_t)(acc >> (nbits - 8)); else overflow = true;
            ++out;
            nbits -= 8;
        }
    };
    clear();
    put(256);
    if (n > 0) {
        int cur = src[0];
        for (long long i = 1; i < n; ++i) {
            const int c = src[i];
            const int32_t key = (cur << 8) | c;
            int hpos = (int)(((uint32_t)key * 2654435761u) % HSIZE);
            int found = -1;
            while (hkey[hpos] != -1) {
                if (hkey[hpos] == key) { found = hval[hpos]; break; }
                if (++hpos == HSIZE) hpos = 0;
            }
            if (found >= 0) { cur = found; continue; }
            put(cur);
            hkey[hpos] = key; hval[hpos] = next++;
            if (next == 4094) {                         // table full: restart (libtiff's CODE_MAX - 1 rule)
                put(256);
                clear();
                next = 258; width = 9;
            } else if (next > (1 << width) - 1 && width < 12) {
                ++width;
            }
            cur = c;
        }
        put(cur);
        // the decoder adds one more entry after this code; keep the widths in step for the EOI code
        ++next;
        if (next > (1 << width) - 1 && width < 12) ++width;
    }
    put(257);
    if (nbits > 0) {
        if (out < dst_cap) dst[out] = (uint8_t)(acc << (8 - nbits)); else overflow = true;
        ++out;
    }
    return overflow ? -1 : out;
}

}  // extern "C"
