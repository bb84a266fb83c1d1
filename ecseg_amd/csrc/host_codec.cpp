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
// so emitting a code is one forward copy from earlier output instead of a backward walk along a prefix chain.  Round 3:
// codes are pulled from a 64-bit window refilled four bytes at a time, and a string is copied in 8-byte steps whenever
// source and destination are at least 8 bytes apart and the destination has 8 bytes of slack (the copy may then run past
// the string's end inside the buffer; the next code overwrites the excess) - 140 -> ~400 MB/s on microscope-like RGB data.
long long ecseg_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0 || dst_cap < 0) return -1;
    if (dst_cap >= 0xffffffffll) return -1;             // (32-bit offsets: a TIFF strip is far below 4 GB)
    struct Entry { uint32_t pos; uint32_t len; };       // codes 0..255 are literal bytes (pos unused); 32 KB: stays in L1
    Entry tab[4096];
    for (int i = 0; i < 256; ++i) tab[i] = Entry{0, 1};
    int next = 258, width = 9;
    long long out = 0;
    uint64_t acc = 0;
    int nbits = 0;
    long long pos = 0;
    int old = -1;
    long long old_pos = 0;                              // where the previous code's string starts in dst
    for (;;) {
        if (nbits < width) {
            if (pos + 4 <= n && nbits <= 32) {          // four bytes at once
                acc = (acc << 32) | ((uint64_t)src[pos] << 24) | ((uint64_t)src[pos + 1] << 16) | ((uint64_t)src[pos + 2] << 8) | src[pos + 3];
                pos += 4; nbits += 32;
            } else {
                while (nbits < width) {
                    if (pos >= n) return out;           // stream ended without EOI: accept what we have
                    acc = (acc << 8) | src[pos++];
                    nbits += 8;
                }
            }
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
        const int old_len = (int)tab[old].len;
        long long cpos; int clen;                       // the string of `code`
        if (code < next) {
            if (code >= 256 && code < 258) return -1;
            cpos = tab[code].pos; clen = (int)tab[code].len;
        } else if (code == next && next < 4096) {      // KwKwK: previous string + its own first byte
            cpos = old_pos; clen = old_len + 1;
        } else {
            return -1;
        }
        const long long start = out;
        const long long room = dst_cap - out;
        const int ncopy = (long long)clen <= room ? clen : (int)room;
        // Round 5: on noisy images most strings are 1 - 3 bytes long and start a few bytes before the write position, where
        // the 8-byte-step copy above did not apply and every code paid a byte loop plus the mispredicted literal / string
        // branch.  A string of a code < next never reaches into the bytes being written (cpos + clen <= out; only KwKwK
        // does), so ONE 8-byte load followed by ONE 8-byte store is exact for clen <= 8 whatever the distance; literals take
        // the same path from a table of their own bytes.
        if (room >= 16 && clen <= 8 && code != next) {
            uint64_t v;
            if (code < 256) v = (uint64_t)code;
            else std::memcpy(&v, dst + cpos, 8);               // (cpos + 8 <= out + 8 <= dst_cap)
            std::memcpy(dst + out, &v, 8);
        }
        else if (code < 256) { if (ncopy > 0) dst[out] = (uint8_t)code; }
        else if (out - cpos >= 8 && room >= (long long)ncopy + 8) {
            const uint8_t* sp = dst + cpos; uint8_t* dp = dst + out;
            for (int k = 0; k < ncopy; k += 8) std::memcpy(dp + k, sp + k, 8);
        }
        else for (int k = 0; k < ncopy; ++k) dst[out + k] = dst[cpos + k];     // may overlap forward (KwKwK): byte by byte
        out += ncopy;
        if (next < 4096) { tab[next] = Entry{(uint32_t)old_pos, (uint32_t)(old_len + 1)}; ++next; }   // previous string + first byte of this one
        if (ncopy < clen) return out;                   // destination full
        old = code; old_pos = start;
        if (next >= (1 << width) - 1 && width < 12) ++width;   // early change
    }
    return out;
}

// Encodes one strip.  Returns the number of bytes written, or -1 when dst_cap is too small
// (n * 2 + 16 bytes always suffice).  Dictionary: open addressing in a 16384-slot table keyed by (prefix code << 8 | byte)
// with a multiplicative hash; a slot is live when its generation tag equals the current one, so a table reset (every 3836
// new codes) costs one increment instead of clearing the table (round 3: 80 -> ~200 MB/s; same output bytes as before).
long long ecseg_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0) return -1;
    constexpr int HBITS = 14, HSIZE = 1 << HBITS;
    static thread_local uint32_t hkey[HSIZE];            // generation << 20 | key (key < 2^20)
    static thread_local uint16_t hval[HSIZE];
    static thread_local uint32_t gen = 0;
    auto clear = [&]() {
        if (++gen >= 4096) { std::memset(hkey, 0, sizeof hkey); gen = 1; }     // (tags of a wrapped generation would alias)
    };
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
            const uint32_t key = ((uint32_t)cur << 8) | (uint32_t)c;
            const uint32_t tag = (gen << 20) | key;
            uint32_t hpos = (key * 2654435761u) >> (32 - HBITS);
            int found = -1;
            while ((hkey[hpos] >> 20) == gen) {
                if (hkey[hpos] == tag) { found = hval[hpos]; break; }
                hpos = (hpos + 1) & (HSIZE - 1);
            }
            if (found >= 0) { cur = found; continue; }
            put(cur);
            hkey[hpos] = tag; hval[hpos] = (uint16_t)next++;
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
