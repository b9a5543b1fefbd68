// Dependency-free image decoding for the texture path (pure CPU).
// Stands in for `image::load_from_memory(bytes)?.to_rgba8()` at
// /root/reference/src/texture.rs:104,114-115 (image 0.24.6 with its "png" and
// "jpeg" features = png 0.17.9 + jpeg-decoder 0.3.0; none of them vendored).
//   PNG  : RFC 2083 / RFC 1950-1951 (inflate); lossless, so bit-exact with any decoder.
//   JPEG : baseline / extended-sequential Huffman, 8-bit, 1 or 3 components,
//          sampling factors 1-2; integer "islow"-style IDCT, triangle-filter
//          chroma upsampling and fixed-point BT.601 conversion — the same
//          published algorithms jpeg-decoder descends from.  JPEG decoders are not
//          bit-identical to each other (+-1-2 LSB): parity unpinned for .jpg.
// Also a minimal PNG writer (stored deflate blocks) for the presentation step.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace rwr {
namespace codec {

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;  // width*height*4, row 0 = top row of the file
};

// ----------------------------------------------------------------- inflate --
class BitReader {
public:
    BitReader(const uint8_t *p, size_t n) : p_(p), n_(n) {}
    bool need(int bits)
    {
        while (cnt_ < bits) {
            if (pos_ >= n_) return false;
            buf_ |= (uint64_t)p_[pos_++] << cnt_;
            cnt_ += 8;
        }
        return true;
    }
    bool get(int bits, uint32_t &out)
    {
        if (bits == 0) { out = 0; return true; }
        if (!need(bits)) return false;
        out = (uint32_t)(buf_ & ((1ull << bits) - 1));
        buf_ >>= bits;
        cnt_ -= bits;
        return true;
    }
    void align_byte()
    {
        const int drop = cnt_ & 7;
        buf_ >>= drop;
        cnt_ -= drop;
    }
    // only valid right after align_byte()
    bool read_bytes(uint8_t *dst, size_t len)
    {
        while (len && cnt_ >= 8) {
            *dst++ = (uint8_t)(buf_ & 0xff);
            buf_ >>= 8;
            cnt_ -= 8;
            len--;
        }
        if (pos_ + len > n_) return false;
        std::memcpy(dst, p_ + pos_, len);
        pos_ += len;
        return true;
    }

private:
    const uint8_t *p_;
    size_t n_, pos_ = 0;
    uint64_t buf_ = 0;
    int cnt_ = 0;
};

struct Huffman {
    uint16_t count[16] = {};
    uint16_t symbol[288] = {};
    bool build(const uint8_t *lengths, int n)
    {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; len++) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + count[len];
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; len++) {
            uint32_t bit;
            if (!br.get(1, bit)) return -1;
            code |= (int)bit;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// max_out: the stream may not inflate to more than this (a few bytes of deflate can describe gigabytes)
inline bool inflate_raw(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t max_out)
{
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                          67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                           1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    BitReader br(src, n);
    uint32_t final_block = 0;
    while (!final_block) {
        uint32_t type;
        if (!br.get(1, final_block) || !br.get(2, type)) return false;
        if (type == 0) {
            br.align_byte();
            uint8_t hdr[4];
            if (!br.read_bytes(hdr, 4)) return false;
            const uint32_t len = hdr[0] | (hdr[1] << 8), nlen = hdr[2] | (hdr[3] << 8);
            if ((len ^ 0xffffu) != nlen) return false;
            const size_t at = out.size();
            if (at + len > max_out) return false;
            out.resize(at + len);
            if (!br.read_bytes(out.data() + at, len)) return false;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        if (type == 1) {
            uint8_t l[288];
            for (int i = 0; i < 144; i++) l[i] = 8;
            for (int i = 144; i < 256; i++) l[i] = 9;
            for (int i = 256; i < 280; i++) l[i] = 7;
            for (int i = 280; i < 288; i++) l[i] = 8;
            lit.build(l, 288);
            uint8_t d[30];
            for (int i = 0; i < 30; i++) d[i] = 5;
            dist.build(d, 30);
        } else {
            uint32_t hlit, hdist, hclen;
            if (!br.get(5, hlit) || !br.get(5, hdist) || !br.get(4, hclen)) return false;
            hlit += 257; hdist += 1; hclen += 4;
            if (hlit > 286 || hdist > 30) return false;
            uint8_t cl[19] = {};
            for (uint32_t i = 0; i < hclen; i++) {
                uint32_t v;
                if (!br.get(3, v)) return false;
                cl[order[i]] = (uint8_t)v;
            }
            Huffman clh;
            if (!clh.build(cl, 19)) return false;
            uint8_t lengths[320] = {};
            uint32_t idx = 0;
            while (idx < hlit + hdist) {
                const int sym = clh.decode(br);
                if (sym < 0) return false;
                if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
                uint32_t rep, prev = 0;
                if (sym == 16) {
                    if (idx == 0) return false;
                    prev = lengths[idx - 1];
                    if (!br.get(2, rep)) return false;
                    rep += 3;
                } else if (sym == 17) {
                    if (!br.get(3, rep)) return false;
                    rep += 3;
                } else {
                    if (!br.get(7, rep)) return false;
                    rep += 11;
                }
                if (idx + rep > hlit + hdist) return false;
                while (rep--) lengths[idx++] = (uint8_t)prev;
            }
            if (lengths[256] == 0) return false;
            if (!lit.build(lengths, (int)hlit)) return false;
            // a single distance code of length 1 is "incomplete" but legal
            dist.build(lengths + hlit, (int)hdist);
        }
        for (;;) {
            const int sym = lit.decode(br);
            if (sym < 0) return false;
            if (sym < 256) {
                if (out.size() >= max_out) return false;
                out.push_back((uint8_t)sym);
                continue;
            }
            if (sym == 256) break;
            const int ls = sym - 257;
            if (ls >= 29) return false;
            uint32_t eb;
            if (!br.get(len_extra[ls], eb)) return false;
            const uint32_t len = len_base[ls] + eb;
            const int ds = dist.decode(br);
            if (ds < 0 || ds >= 30) return false;
            if (!br.get(dist_extra[ds], eb)) return false;
            const uint32_t d = dist_base[ds] + eb;
            if (d > out.size()) return false;
            const size_t at = out.size();
            if (at + len > max_out) return false;
            out.resize(at + len);
            for (uint32_t i = 0; i < len; i++) out[at + i] = out[at + i - d];
        }
    }
    return true;
}

inline uint32_t adler32(const uint8_t *p, size_t n)
{
    uint32_t a = 1, b = 0;
    for (size_t i = 0; i < n; i++) {
        a = (a + p[i]) % 65521u;
        b = (b + a) % 65521u;
    }
    return (b << 16) | a;
}

inline bool zlib_decompress(const uint8_t *src, size_t n, std::vector<uint8_t> &out, std::string &err, size_t max_out)
{
    if (n < 6) { err = "zlib stream too short"; return false; }
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) { err = "bad zlib header"; return false; }
    if (!inflate_raw(src + 2, n - 2, out, max_out)) { err = "corrupt deflate stream (or one that inflates beyond the image size)"; return false; }
    return true;
}

// --------------------------------------------------------------------- PNG --
inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

inline uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

inline bool is_png(const uint8_t *b, size_t n)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    return n >= 8 && std::memcmp(b, sig, 8) == 0;
}

inline uint8_t paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

inline bool decode_png(const uint8_t *b, size_t n, Image &img, std::string &err)
{
    if (!is_png(b, n)) { err = "not a PNG"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool seen_iend = false;
    while (pos + 12 <= n && !seen_iend) {
        const uint32_t len = be32(b + pos);
        if (pos + 12 + (size_t)len > n) { err = "truncated PNG chunk"; return false; }
        const uint8_t *type = b + pos + 4, *data = b + pos + 8;
        if (crc32_update(0, type, 4 + (size_t)len) != be32(data + len)) { err = "PNG chunk CRC mismatch"; return false; }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            w = be32(data); h = be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { err = "bad PNG compression/filter method"; return false; }
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_iend = true;
        }
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0) { err = "PNG without IHDR"; return false; }
    if ((uint64_t)w * h > (1ull << 28)) { err = "PNG too large"; return false; }
    if (interlace != 0) { err = "interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "bad PNG colour type"; return false;
    }
    const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
    if (!depth_ok) { err = "bad PNG bit depth"; return false; }
    if (ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }

    std::vector<uint8_t> raw;
    const size_t bits_pp = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bits_pp + 7) / 8;
    if (!zlib_decompress(idat.data(), idat.size(), raw, err, (stride + 1) * (size_t)h)) return false;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    if (raw.size() < (stride + 1) * h) { err = "PNG image data too short"; return false; }

    // unfilter in place (RFC 2083 §6)
    std::vector<uint8_t> prev_row(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *line = raw.data() + (size_t)y * (stride + 1);
        const uint8_t ft = line[0];
        uint8_t *cur = line + 1;
        const uint8_t *up = y ? line - stride : prev_row.data();
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, bb = up[i], c = i >= bpp ? up[i - bpp] : 0;
            switch (ft) {
                case 0: break;
                case 1: cur[i] = (uint8_t)(cur[i] + a); break;
                case 2: cur[i] = (uint8_t)(cur[i] + bb); break;
                case 3: cur[i] = (uint8_t)(cur[i] + ((a + bb) >> 1)); break;
                case 4: cur[i] = (uint8_t)(cur[i] + paeth(a, bb, c)); break;
                default: err = "bad PNG filter type"; return false;
            }
        }
    }

    img.width = w; img.height = h;
    img.rgba.resize((size_t)w * h * 4);
    auto sample = [&](const uint8_t *row, size_t idx) -> uint32_t {  // idx-th sample of the row, native depth
        if (depth == 8) return row[idx];
        if (depth == 16) return ((uint32_t)row[2 * idx] << 8) | row[2 * idx + 1];
        const size_t bit = idx * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    };
    auto to8 = [&](uint32_t v) -> uint8_t {
        if (depth == 8) return (uint8_t)v;
        if (depth == 16) return (uint8_t)((v + 128u) / 257u);  // image-rs u16 -> u8 conversion
        return (uint8_t)(v * 255u / ((1u << depth) - 1));
    };
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = raw.data() + (size_t)y * (stride + 1) + 1;
        uint8_t *dst = img.rgba.data() + (size_t)y * w * 4;
        for (uint32_t x = 0; x < w; x++, dst += 4) {
            switch (ctype) {
                case 0: {
                    const uint32_t g = sample(row, x);
                    dst[0] = dst[1] = dst[2] = to8(g);
                    dst[3] = 255;
                    if (trns.size() >= 2 && g == (((uint32_t)trns[0] << 8) | trns[1])) dst[3] = 0;
                    break;
                }
                case 2: {
                    const uint32_t r = sample(row, 3 * x), g = sample(row, 3 * x + 1), bl = sample(row, 3 * x + 2);
                    dst[0] = to8(r); dst[1] = to8(g); dst[2] = to8(bl); dst[3] = 255;
                    if (trns.size() >= 6 && r == (((uint32_t)trns[0] << 8) | trns[1]) &&
                        g == (((uint32_t)trns[2] << 8) | trns[3]) && bl == (((uint32_t)trns[4] << 8) | trns[5]))
                        dst[3] = 0;
                    break;
                }
                case 3: {
                    const uint32_t i = sample(row, x);
                    if (3 * (size_t)i + 2 >= plte.size()) { err = "PNG palette index out of range"; return false; }
                    dst[0] = plte[3 * i]; dst[1] = plte[3 * i + 1]; dst[2] = plte[3 * i + 2];
                    dst[3] = i < trns.size() ? trns[i] : 255;
                    break;
                }
                case 4: {
                    const uint8_t g = to8(sample(row, 2 * x));
                    dst[0] = dst[1] = dst[2] = g;
                    dst[3] = to8(sample(row, 2 * x + 1));
                    break;
                }
                default: {
                    dst[0] = to8(sample(row, 4 * x)); dst[1] = to8(sample(row, 4 * x + 1));
                    dst[2] = to8(sample(row, 4 * x + 2)); dst[3] = to8(sample(row, 4 * x + 3));
                }
            }
        }
    }
    return true;
}

// PNG writer: RGBA8, filter 0, stored deflate blocks.
inline std::vector<uint8_t> encode_png_rgba8(const uint8_t *rgba, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    size_t off = 0;
    do {
        const size_t chunk = raw.size() - off > 65535 ? 65535 : raw.size() - off;
        z.push_back(off + chunk == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(chunk & 0xff)); z.push_back((uint8_t)(chunk >> 8));
        z.push_back((uint8_t)(~chunk & 0xff)); z.push_back((uint8_t)((~chunk >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + chunk);
        off += chunk;
    } while (off < raw.size());
    const uint32_t ad = adler32(raw.data(), raw.size());
    for (int s = 24; s >= 0; s -= 8) z.push_back((uint8_t)(ad >> s));

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    auto put_chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        const uint32_t len = (uint32_t)data.size();
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(len >> s));
        const size_t at = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        const uint32_t crc = crc32_update(0, out.data() + at, 4 + data.size());
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(crc >> s));
    };
    std::vector<uint8_t> ihdr(13);
    for (int i = 0; i < 4; i++) { ihdr[i] = (uint8_t)(w >> (24 - 8 * i)); ihdr[4 + i] = (uint8_t)(h >> (24 - 8 * i)); }
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    put_chunk("IHDR", ihdr);
    put_chunk("IDAT", z);
    put_chunk("IEND", {});
    return out;
}

// -------------------------------------------------------------------- JPEG --
inline bool is_jpeg(const uint8_t *b, size_t n) { return n >= 3 && b[0] == 0xff && b[1] == 0xd8 && b[2] == 0xff; }

namespace jpeg {

struct HuffTable {
    bool present = false;
    uint8_t bits[17] = {};
    uint8_t vals[256] = {};
    int mincode[17], maxcode[18], valptr[17];
    void finish()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int blocks_w = 0, blocks_h = 0;  // padded to whole MCUs
    int pred = 0;
    std::vector<uint8_t> plane;  // blocks_w*8 x blocks_h*8
};

class EntropyReader {
public:
    EntropyReader(const uint8_t *p, size_t n, size_t pos) : p_(p), n_(n), pos_(pos) {}
    int bit()
    {
        if (cnt_ == 0) {
            uint8_t c = 0;
            if (!hit_marker_ && pos_ < n_) {
                c = p_[pos_++];
                if (c == 0xff) {
                    const uint8_t d = pos_ < n_ ? p_[pos_] : 0xd9;
                    if (d == 0) pos_++;
                    else { hit_marker_ = true; pos_--; c = 0; }
                }
            }
            buf_ = c;
            cnt_ = 8;
        }
        cnt_--;
        return (buf_ >> cnt_) & 1;
    }
    int bits(int n)
    {
        int v = 0;
        while (n--) v = (v << 1) | bit();
        return v;
    }
    int decode(const HuffTable &t)
    {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | bit();
            if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
        }
        return -1;
    }
    static int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }
    // skip to and consume an RSTn marker
    bool restart()
    {
        cnt_ = 0;
        hit_marker_ = false;
        while (pos_ + 1 < n_) {
            if (p_[pos_] == 0xff && p_[pos_ + 1] >= 0xd0 && p_[pos_ + 1] <= 0xd7) { pos_ += 2; return true; }
            if (p_[pos_] == 0xff && p_[pos_ + 1] != 0 && p_[pos_ + 1] != 0xff) return false;
            pos_++;
        }
        return false;
    }

private:
    const uint8_t *p_;
    size_t n_, pos_;
    uint32_t buf_ = 0;
    int cnt_ = 0;
    bool hit_marker_ = false;
};

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// 8x8 inverse DCT, 12-bit fixed-point two-pass butterfly (the public-domain
// stb_image / jpeg-decoder formulation of the LL&M algorithm).
inline void idct_block(const int *d, uint8_t *out, int out_stride)
{
    auto f2f = [](double x) { return (int)(x * 4096 + 0.5); };
    static const int c0541 = f2f(0.5411961), cm1847 = -f2f(1.847759065), c0765 = f2f(0.765366865), c1175 = f2f(1.175875602),
                     c0298 = f2f(0.298631336), c2053 = f2f(2.053119869), c3072 = f2f(3.072711026), c1501 = f2f(1.501321110),
                     cm0899 = -f2f(0.899976223), cm2562 = -f2f(2.562915447), cm1961 = -f2f(1.961570560), cm0390 = -f2f(0.390180644);
    int val[64];
    auto pass = [&](int s0, int s1, int s2, int s3, int s4, int s5, int s6, int s7, int &x0, int &x1, int &x2, int &x3,
                    int &t0, int &t1, int &t2, int &t3) {
        int p2 = s2, p3 = s6;
        int p1 = (p2 + p3) * c0541;
        t2 = p1 + p3 * cm1847;
        t3 = p1 + p2 * c0765;
        p2 = s0; p3 = s4;
        t0 = (p2 + p3) * 4096;
        t1 = (p2 - p3) * 4096;
        x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
        t0 = s7; t1 = s5; t2 = s3; t3 = s1;
        p3 = t0 + t2;
        int p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        const int p5 = (p3 + p4) * c1175;
        t0 = t0 * c0298; t1 = t1 * c2053; t2 = t2 * c3072; t3 = t3 * c1501;
        p1 = p5 + p1 * cm0899;
        p2 = p5 + p2 * cm2562;
        p3 = p3 * cm1961;
        p4 = p4 * cm0390;
        t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    };
    for (int i = 0; i < 8; i++) {
        const int *c = d + i;
        int *v = val + i;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            const int dc = c[0] * 4;
            v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
        } else {
            int x0, x1, x2, x3, t0, t1, t2, t3;
            pass(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56], x0, x1, x2, x3, t0, t1, t2, t3);
            x0 += 512; x1 += 512; x2 += 512; x3 += 512;
            v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
            v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
            v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
            v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
        }
    }
    for (int i = 0; i < 8; i++) {
        const int *v = val + 8 * i;
        uint8_t *o = out + i * out_stride;
        int x0, x1, x2, x3, t0, t1, t2, t3;
        pass(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], x0, x1, x2, x3, t0, t1, t2, t3);
        const int bias = 65536 + (128 << 17);
        x0 += bias; x1 += bias; x2 += bias; x3 += bias;
        o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
        o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
        o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
        o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
    }
}

static const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                    15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// Upsample one component plane (pw x ph valid samples, row stride `stride`) by
// integer factors (fx, fy) in {1,2} to out (ow x oh), triangle filter
// ("fancy upsampling") where a factor is 2.
inline void upsample(const uint8_t *src, int stride, int pw, int ph, int fx, int fy, int ow, int oh, std::vector<uint8_t> &out)
{
    out.resize((size_t)ow * oh);
    std::vector<int> rowbuf((size_t)pw);
    for (int oy = 0; oy < oh; oy++) {
        // vertical: blend the two nearest source rows 3:1 (scaled by 4), or copy (scaled by 4)
        int sy = fy == 2 ? oy >> 1 : oy;
        if (sy >= ph) sy = ph - 1;
        int sy2 = sy;
        if (fy == 2) sy2 = (oy & 1) ? (sy + 1 < ph ? sy + 1 : sy) : (sy > 0 ? sy - 1 : sy);
        const uint8_t *near = src + (size_t)sy * stride, *far = src + (size_t)sy2 * stride;
        for (int x = 0; x < pw; x++) rowbuf[x] = fy == 2 ? 3 * near[x] + far[x] : 4 * near[x];
        uint8_t *o = out.data() + (size_t)oy * ow;
        if (fx == 1) {
            for (int x = 0; x < ow; x++) o[x] = (uint8_t)((rowbuf[x < pw ? x : pw - 1] + 2) >> 2);
        } else {
            for (int x = 0; x < ow; x++) {
                int sx = x >> 1;
                if (sx >= pw) sx = pw - 1;
                const int nx = (x & 1) ? (sx + 1 < pw ? sx + 1 : sx) : (sx > 0 ? sx - 1 : sx);
                const int bias = (x & 1) ? 7 : 8;
                o[x] = (uint8_t)((3 * rowbuf[sx] + rowbuf[nx] + bias) >> 4);
            }
        }
    }
}

}  // namespace jpeg

inline bool decode_jpeg(const uint8_t *b, size_t n, Image &img, std::string &err)
{
    using namespace jpeg;
    if (!is_jpeg(b, n)) { err = "not a JPEG"; return false; }
    uint16_t qt[4][64] = {};
    bool qt_present[4] = {};
    HuffTable dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, restart_interval = 0;
    bool have_sof = false, decoded = false;
    int adobe_transform = -1;
    size_t pos = 2;
    while (pos + 4 <= n && !decoded) {
        if (b[pos] != 0xff) { pos++; continue; }
        const uint8_t m = b[pos + 1];
        if (m == 0xff) { pos++; continue; }
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) { pos += 2; continue; }
        if (m == 0xd9) break;
        const size_t len = ((size_t)b[pos + 2] << 8) | b[pos + 3];
        if (len < 2 || pos + 2 + len > n) { err = "truncated JPEG segment"; return false; }
        const uint8_t *s = b + pos + 4;
        const size_t sl = len - 2;
        switch (m) {
            case 0xdb: {  // DQT
                size_t i = 0;
                while (i < sl) {
                    const int pq = s[i] >> 4, tq = s[i] & 15;
                    i++;
                    if (tq > 3 || i + (pq ? 128 : 64) > sl) { err = "bad DQT"; return false; }
                    for (int k = 0; k < 64; k++) {
                        qt[tq][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
                        i += pq ? 2 : 1;
                    }
                    qt_present[tq] = true;
                }
                break;
            }
            case 0xc4: {  // DHT
                size_t i = 0;
                while (i + 17 <= sl) {
                    const int tc = s[i] >> 4, th = s[i] & 15;
                    if (tc > 1 || th > 3) { err = "bad DHT"; return false; }
                    HuffTable &t = tc ? ac[th] : dc[th];
                    int total = 0;
                    for (int l = 1; l <= 16; l++) { t.bits[l] = s[i + l]; total += t.bits[l]; }
                    i += 17;
                    if (total > 256 || i + total > sl) { err = "bad DHT"; return false; }
                    std::memcpy(t.vals, s + i, total);
                    i += total;
                    t.present = true;
                    t.finish();
                }
                break;
            }
            case 0xc0: case 0xc1: {  // SOF0 / SOF1
                if (sl < 6) { err = "bad SOF"; return false; }
                if (s[0] != 8) { err = "only 8-bit JPEG is supported"; return false; }
                height = (s[1] << 8) | s[2];
                width = (s[3] << 8) | s[4];
                const int nc = s[5];
                if ((nc != 1 && nc != 3) || sl < (size_t)(6 + 3 * nc) || width == 0 || height == 0) { err = "unsupported JPEG component count/size"; return false; }
                // bounded before anything is allocated: textures larger than 16384 on a side are refused at upload anyway
                if (width > 16384 || height > 16384 || (uint64_t)width * (uint64_t)height > (1ull << 28)) { err = "JPEG too large"; return false; }
                comps.resize(nc);
                for (int c = 0; c < nc; c++) {
                    comps[c].id = s[6 + 3 * c];
                    comps[c].h = s[7 + 3 * c] >> 4;
                    comps[c].v = s[7 + 3 * c] & 15;
                    comps[c].tq = s[8 + 3 * c];
                    if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2 || comps[c].tq > 3) { err = "unsupported JPEG sampling factors"; return false; }
                }
                have_sof = true;
                break;
            }
            case 0xc2: case 0xc3: case 0xc5: case 0xc6: case 0xc7: case 0xc9: case 0xca: case 0xcb: case 0xcd: case 0xce: case 0xcf:
                err = "progressive / lossless / arithmetic JPEG is not supported";
                return false;
            case 0xdd:
                if (sl >= 2) restart_interval = (s[0] << 8) | s[1];
                break;
            case 0xee:
                if (sl >= 12 && !std::memcmp(s, "Adobe", 5)) adobe_transform = s[11];
                break;
            case 0xda: {  // SOS
                if (!have_sof) { err = "SOS before SOF"; return false; }
                if (sl < 1) { err = "truncated SOS"; return false; }
                const int ns = s[0];
                if (ns != (int)comps.size() || sl < (size_t)(1 + 2 * ns + 3)) { err = "non-interleaved JPEG scans are not supported"; return false; }
                for (int k = 0; k < ns; k++) {
                    const int cid = s[1 + 2 * k];
                    bool found = false;
                    for (auto &c : comps)
                        if (c.id == cid) { c.td = s[2 + 2 * k] >> 4; c.ta = s[2 + 2 * k] & 15; found = true; }
                    if (!found) { err = "SOS references unknown component"; return false; }
                }
                int hmax = 1, vmax = 1;
                for (auto &c : comps) { hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax; }
                if (comps.size() == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
                const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
                const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
                for (auto &c : comps) {
                    if (!qt_present[c.tq] || c.td > 3 || c.ta > 3 || !dc[c.td].present || !ac[c.ta].present) { err = "JPEG scan uses an undefined table"; return false; }
                    c.blocks_w = mcus_x * c.h;
                    c.blocks_h = mcus_y * c.v;
                    c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
                    c.pred = 0;
                }
                EntropyReader er(b, n, pos + 2 + len);
                int until_restart = restart_interval;
                for (int my = 0; my < mcus_y; my++) {
                    for (int mx = 0; mx < mcus_x; mx++) {
                        if (restart_interval && until_restart == 0) {
                            if (!er.restart()) { err = "missing JPEG restart marker"; return false; }
                            for (auto &c : comps) c.pred = 0;
                            until_restart = restart_interval;
                        }
                        for (auto &c : comps) {
                            for (int by = 0; by < c.v; by++) {
                                for (int bx = 0; bx < c.h; bx++) {
                                    int coef[64] = {};
                                    const int t = er.decode(dc[c.td]);
                                    if (t < 0 || t > 16) { err = "corrupt JPEG (DC)"; return false; }
                                    const int diff = t ? EntropyReader::extend(er.bits(t), t) : 0;
                                    c.pred += diff;
                                    coef[0] = c.pred * qt[c.tq][0];
                                    for (int k = 1; k < 64;) {
                                        const int rs = er.decode(ac[c.ta]);
                                        if (rs < 0) { err = "corrupt JPEG (AC)"; return false; }
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) {
                                            if (r != 15) break;
                                            k += 16;
                                            continue;
                                        }
                                        k += r;
                                        if (k > 63) { err = "corrupt JPEG (AC run)"; return false; }
                                        coef[kZigzag[k]] = EntropyReader::extend(er.bits(sz), sz) * qt[c.tq][kZigzag[k]];
                                        k++;
                                    }
                                    const int px = (mx * c.h + bx) * 8, py = (my * c.v + by) * 8;
                                    idct_block(coef, c.plane.data() + (size_t)py * c.blocks_w * 8 + px, c.blocks_w * 8);
                                }
                            }
                        }
                        until_restart--;
                    }
                }
                // upsample + colour convert
                img.width = (uint32_t)width; img.height = (uint32_t)height;
                img.rgba.resize((size_t)width * height * 4);
                std::vector<std::vector<uint8_t>> full(comps.size());
                for (size_t ci = 0; ci < comps.size(); ci++) {
                    const Component &c = comps[ci];
                    const int fx = hmax / c.h, fy = vmax / c.v;
                    const int pw = (width * c.h + hmax - 1) / hmax, ph = (height * c.v + vmax - 1) / vmax;
                    upsample(c.plane.data(), c.blocks_w * 8, pw, ph, fx, fy, width, height, full[ci]);
                }
                const bool ycc = comps.size() == 3 && adobe_transform != 0;
                for (size_t i = 0; i < (size_t)width * height; i++) {
                    uint8_t *d = img.rgba.data() + 4 * i;
                    if (comps.size() == 1) {
                        d[0] = d[1] = d[2] = full[0][i];
                    } else if (!ycc) {
                        d[0] = full[0][i]; d[1] = full[1][i]; d[2] = full[2][i];
                    } else {
                        // BT.601 full range, 20-bit fixed point
                        const int y = ((int)full[0][i] << 20) + (1 << 19);
                        const int cb = (int)full[1][i] - 128, cr = (int)full[2][i] - 128;
                        const int r = y + cr * 1470208;                                      // 1.40200 * 2^20 (rounded to 2^12 steps)
                        const int g = y + cr * -748800 + ((cb * -360960) & (int)0xffff0000);  // 0.71414, 0.34414
                        const int bl = y + cb * 1858048;                                     // 1.77200
                        d[0] = clamp8(r >> 20); d[1] = clamp8(g >> 20); d[2] = clamp8(bl >> 20);
                    }
                    d[3] = 255;
                }
                decoded = true;
                break;
            }
            default: break;
        }
        pos += 2 + len;
    }
    if (!decoded) { if (err.empty()) err = "JPEG has no image scan"; return false; }
    return true;
}

inline bool decode_image(const uint8_t *bytes, size_t n, Image &img, std::string &err)
{
    if (is_png(bytes, n)) return decode_png(bytes, n, img, err);
    if (is_jpeg(bytes, n)) return decode_jpeg(bytes, n, img, err);
    err = "unrecognised image format (PNG and JPEG are supported)";
    return false;
}

}  // namespace codec
}  // namespace rwr
