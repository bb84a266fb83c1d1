// Host-side byte codecs used by the file I/O around the hot path (no GPU work): TIFF-flavoured LZW
// (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, "early change"), as written by OpenCV's imwrite for the
// reference's dapi/<name>.tif outputs and found in typical microscope TIFF inputs (reference src/utils.py:110,123).
// These two functions are most of the CPU time of `make metaseg` (an RGB LZW input is 2.3 M codes; DESIGN.md 8: on a
// GPU box with a 16-CPU quota the codecs, not the GPU, bound a narrow model's images/s).
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ecseg_hip.h"

namespace {

inline uint64_t load_be64(const uint8_t* p) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}

inline void copy8(uint8_t* d, const uint8_t* s) {
    uint64_t v;
    std::memcpy(&v, s, 8);
    std::memcpy(d, &v, 8);
}

// State of one LZW encoder (see ecseg_lzw_encode).
struct LzwEncoder {
    static constexpr int HBITS = 14, HSIZE = 1 << HBITS;
    static constexpr uint32_t EMPTY = 0xffffffffu;      // (a live slot is at most 0xfffff << 12 | 4093)
    uint32_t* slot;
    uint8_t* dst;
    long long cap, out = 0;
    uint64_t acc = 0;
    int nbits = 0, width = 9, next = 258, cur = 0;
    bool overflow = false;

    LzwEncoder(uint32_t* table, uint8_t* d, long long dst_cap) : slot(table), dst(d), cap(dst_cap) {
        std::memset(slot, 0xff, sizeof(uint32_t) * HSIZE);
        put(256);
    }
    inline void put(int code) {
        acc = (acc << width) | (uint32_t)code;          // (nbits <= 31 before: at most 43 bits in flight)
        nbits += width;
        if (nbits >= 32) {
            const uint32_t w = (uint32_t)(acc >> (nbits - 32));
            if (out + 4 <= cap) {
                const uint32_t be = __builtin_bswap32(w);
                std::memcpy(dst + out, &be, 4);
            } else {
                for (int k = 0; k < 4; ++k) { if (out + k < cap) dst[out + k] = (uint8_t)(w >> (24 - 8 * k)); else overflow = true; }
            }
            out += 4;
            nbits -= 32;
        }
    }
    inline void first(uint8_t c) { cur = c; }
    inline void step(uint8_t c) {
        const uint32_t key = ((uint32_t)cur << 8) | (uint32_t)c;
        uint32_t hpos = (key * 2654435761u) >> (32 - HBITS);
        uint32_t e = slot[hpos];
        while (e != EMPTY && (e >> 12) != key) {
            hpos = (hpos + 1) & (HSIZE - 1);
            e = slot[hpos];
        }
        if (e != EMPTY) { cur = (int)(e & 0xfffu); return; }
        put(cur);
        slot[hpos] = (key << 12) | (uint32_t)next++;
        if (next == 4094) {                             // table full: restart (libtiff's CODE_MAX - 1 rule)
            put(256);
            std::memset(slot, 0xff, sizeof(uint32_t) * HSIZE);
            next = 258; width = 9;
        } else if (next > (1 << width) - 1 && width < 12) {
            ++width;
        }
        cur = c;
    }
    inline void last() {
        put(cur);
        // the decoder adds one more entry after this code; keep the widths in step for the EOI code
        ++next;
        if (next > (1 << width) - 1 && width < 12) ++width;
    }
    inline long long finish() {
        put(257);
        while (nbits > 0) {                             // the last 1..31 bits, zero-padded to a byte
            const int take = nbits >= 8 ? 8 : nbits;
            const uint8_t b = (uint8_t)(nbits >= 8 ? (acc >> (nbits - 8)) : (acc << (8 - nbits)));
            if (out < cap) dst[out] = b; else overflow = true;
            ++out;
            nbits -= take;
        }
        return overflow ? -1 : out;
    }
};

}  // namespace

extern "C" {

// Decodes one LZW strip.  Returns the number of bytes written (<= dst_cap) or -1 on a corrupt stream.
// Every table entry is remembered as (offset, length) of an occurrence of its string in the OUTPUT written so far: the
// string of a new entry (previous string + first byte of the current one) starts where the previous code was emitted,
// so emitting a code is one forward copy from earlier output instead of a backward walk along a prefix chain.
//   Round 5: the loop has a fast and a careful form.  The fast form runs while 8 more source bytes and (longest string
// + 16) more destination bytes exist: codes come from a left-aligned 64-bit window refilled without a branch (one
// unaligned big-endian load per code), a literal and a string take the SAME path - one 8-byte load from either a
// 256-byte table of the literals or the earlier output (a pointer select, not a branch: on noisy images half of the
// codes are literals and that branch mispredicted every other time), then 8-byte steps for the rare string longer than
// 8 - and the first code after a ClearCode is consumed by the ClearCode's own handler, so the loop carries no "is there
// a previous code" test.  An 8-byte copy may write up to 7 bytes past the string's end inside the buffer; the next code
// overwrites them.  The careful form (byte-exact copies, bounds on every step) finishes the last bytes of a strip and
// handles truncated inputs / destinations exactly as rounds 1-4 did.  236 -> ~400 MB/s on noisy RGB data.
long long ecseg_lzw_decode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0 || dst_cap < 0) return -1;
    if (dst_cap >= 0xffffffffll) return -1;             // (32-bit offsets: a TIFF strip is far below 4 GB)
    struct Entry { uint32_t pos; uint32_t len; };       // codes 0..255: literal bytes (pos unused); 32 KB: stays in L1
    Entry tab[4096];
    uint8_t lit[256 + 8];
    for (int i = 0; i < 256; ++i) { tab[i] = Entry{0, 1}; lit[i] = (uint8_t)i; }
    std::memset(lit + 256, 0, 8);
    int next = 258, width = 9;
    long long out = 0;
    long long pos = 0;                                  // next source byte to enter the window
    uint64_t acc = 0;                                   // window: `nbits` valid bits, left-aligned
    int nbits = 0;
    int old = -1;                                       // previous code (-1: none since the last ClearCode)
    long long old_pos = 0;                              // where the previous code's string starts in dst
    uint32_t old_len = 0;
    uint32_t longest = 1;                               // longest string in the table (bounds the fast form's over-write)

    // ---- fast form ------------------------------------------------------------------------------------------------
    for (;;) {
        if (pos + 8 > n || out + (long long)longest + 17 > dst_cap) break;
        // refill: bytes [pos, pos + 8) hold the bits that follow the window's `nbits` valid ones
        acc |= load_be64(src + pos) >> nbits;
        pos += (63 - nbits) >> 3;
        nbits |= 56;
        int code = (int)(acc >> (64 - width));
        acc <<= width; nbits -= width;
        if (__builtin_expect(code == 257, 0)) return out;
        if (__builtin_expect(code == 256, 0)) {
            next = 258; width = 9; old = -1; longest = 1;
            // the code after a ClearCode is a literal that adds no entry (nbits >= 44 here: no refill needed)
            code = (int)(acc >> (64 - 9));
            acc <<= 9; nbits -= 9;
            if (code == 257) return out;
            if (code == 256) continue;                  // (a second ClearCode: same state)
            if (code > 255) return -1;
            old_pos = out; old_len = 1; old = code;
            dst[out++] = (uint8_t)code;
            continue;
        }
        if (__builtin_expect(old < 0, 0)) {
            // no previous code: a stream that does not begin with a ClearCode (libtiff tolerates it) or the code behind a
            // second ClearCode in a row.  This code is ALREADY consumed: it is handled here as the literal it has to be (round 5
            // left for the careful form at this point, which resumed from the next code and silently dropped this byte)
            if (code > 255) return -1;
            old_pos = out; old_len = 1; old = code;
            dst[out++] = (uint8_t)code;
            continue;
        }
        uint32_t cpos, clen;
        const uint8_t* sp;
        if (__builtin_expect(code < next, 1)) {
            cpos = tab[code].pos; clen = tab[code].len;
            sp = code < 256 ? lit + code : dst + cpos;  // (cpos + clen <= out: the 8-byte load stays inside dst, see the guard)
            copy8(dst + out, sp);
            if (__builtin_expect(clen > 8, 0))
                for (uint32_t k = 8; k < clen; k += 8) copy8(dst + out + k, sp + k);      // (out - cpos >= clen > 8: no overlap inside a step)
        } else if (code == next && next < 4096) {      // KwKwK: the previous string + its own first byte
            cpos = (uint32_t)old_pos; clen = old_len + 1;
            const uint8_t first = old < 256 ? (uint8_t)old : dst[old_pos];
            if (old_len >= 8) {
                for (uint32_t k = 0; k < clen; k += 8) copy8(dst + out + k, dst + cpos + k);   // distance old_len >= 8
            } else {
                const uint8_t* s2 = old < 256 ? lit + old : dst + cpos;
                copy8(dst + out, s2);                   // bytes [0, old_len) are right, byte old_len is patched below
            }
            dst[out + old_len] = first;
        } else {
            return -1;
        }
        if (next < 4096) {
            tab[next] = Entry{(uint32_t)old_pos, old_len + 1};     // previous string + first byte of this one
            if (old_len + 1 > longest) longest = old_len + 1;
            ++next;
        }
        old = code; old_pos = out; old_len = clen;
        out += clen;
        if (next >= (1 << width) - 1 && width < 12) ++width;       // early change
    }

    // ---- careful form (rounds 1-4): the tail of a strip, short inputs, a destination smaller than the stream ---------
    // hand the window over: the careful reader keeps its bits right-aligned
    uint64_t racc = nbits ? acc >> (64 - nbits) : 0;
    for (;;) {
        if (nbits < width) {
            while (nbits < width) {
                if (pos >= n) return out;               // stream ended without EOI: accept what we have
                racc = (racc << 8) | src[pos++];
                nbits += 8;
            }
        }
        const int code = (int)((racc >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 257) break;
        if (code == 256) { next = 258; width = 9; old = -1; continue; }
        if (old < 0) {
            if (code >= 256) return -1;
            if (out >= dst_cap) return out;             // truncated output buffer: write what fits
            old_pos = out; old_len = 1;
            dst[out++] = (uint8_t)code;
            old = code;
            continue;
        }
        long long cpos; int clen;                       // the string of `code`
        if (code < next) {
            if (code >= 256 && code < 258) return -1;
            cpos = tab[code].pos; clen = (int)tab[code].len;
        } else if (code == next && next < 4096) {      // KwKwK: previous string + its own first byte
            cpos = old_pos; clen = (int)old_len + 1;
        } else {
            return -1;
        }
        const long long start = out;
        const long long room = dst_cap - out;
        const int ncopy = (long long)clen <= room ? clen : (int)room;
        if (code < 256) { if (ncopy > 0) dst[out] = (uint8_t)code; }
        else if (code == next && old < 256) {          // KwKwK after a literal: that byte twice
            for (int k = 0; k < ncopy; ++k) dst[out + k] = (uint8_t)old;
        }
        else for (int k = 0; k < ncopy; ++k) dst[out + k] = dst[cpos + k];     // may overlap forward (KwKwK): byte by byte
        out += ncopy;
        if (next < 4096) { tab[next] = Entry{(uint32_t)old_pos, old_len + 1}; ++next; }
        if (ncopy < clen) return out;                   // destination full
        old = code; old_pos = start; old_len = (uint32_t)clen;
        if (next >= (1 << width) - 1 && width < 12) ++width;   // early change
    }
    return out;
}

// Encodes one strip.  Returns the number of bytes written, or -1 when dst_cap is too small
// (n * 2 + 16 bytes always suffice).  Dictionary: open addressing keyed by (prefix code << 8 | byte), ONE 32-bit word per
// slot - key (20 bits) << 12 | code (12 bits), all ones = empty - in a 16384-slot table cleared with one memset at every
// table reset (3836 new codes = at least 7 KB of input; rounds 3-4 probed a tag table and then a value table); codes leave
// through a 64-bit accumulator flushed four bytes at a time.  Greedy LZW with a fixed reset rule has one output: the bytes
// are those of every earlier round.  The loop is one dependency chain per input byte (previous code -> key -> hash -> slot
// -> code) plus one mispredicted hit / miss branch per emitted code: ~7 ns per byte.  (Two strips interleaved in one loop -
// a TIFF strip is its own stream - gained 8 % on the 5-row strips of the dapi/ writer, a 4096-slot table that stays in L1 for
// such strips gained 2 %: neither kept.)
long long ecseg_lzw_encode(const uint8_t* src, long long n, uint8_t* dst, long long dst_cap) {
    if (!src || !dst || n < 0) return -1;
    static thread_local uint32_t table[LzwEncoder::HSIZE];
    LzwEncoder e(table, dst, dst_cap);
    if (n > 0) {
        e.first(src[0]);
        for (long long i = 1; i < n; ++i) e.step(src[i]);
        e.last();
    }
    return e.finish();
}

}  // extern "C"
