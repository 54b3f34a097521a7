// FLAC container -> interleaved integer PCM, on the HOST (row N1 of SURVEY.md §8(f): sylber/model/sylber.py:83 reads files with torchaudio.load, which decodes
// FLAC through its codec backend; the reference's data code falls back from .wav to .flac, sylber/dataset/collective_audio_segment.py:61-67).
//
// A FLAC stream is a serial, bit-granular entropy code (unary + Rice), so decoding it is host work like the RIFF header parse; everything after it -- integer -> float,
// resampling to 16 kHz, per-file normalisation -- stays on the device (csrc/ingest.hip: the decoded samples are handed over as 16- or 32-bit PCM scaled to full range,
// which is exactly what torchaudio's int -> float conversion of a FLAC file produces).  No codec library is in the image and none is linked: this file restates the
// published format (xiph.org "FLAC format", RFC 9639): STREAMINFO, frame header (fixed / variable block size, UTF-8 coded number, CRC-8), subframes CONSTANT / VERBATIM /
// FIXED (orders 0-4) / LPC (orders 1-32) with wasted bits, residual coding methods 0 / 1 (4- / 5-bit Rice parameters, escape partitions), the four stereo channel
// assignments, 4-32 bits per sample, frame CRC-16.  Parity is UNPINNED against torchaudio (absent from the image; no FLAC file or encoder exists in it either): the
// tests encode with an independent Python encoder (tests/flac_enc.py) and decode here.  What protects a user's real files: every frame's CRC-16 is checked, and the MD5
// of the decoded PCM is compared with the one the ENCODER of the file wrote into STREAMINFO -- a wrong decode of a real file cannot pass silently.
#include "kernels.h"
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct BitReader {
    const uint8_t* p; size_t n, pos = 0; uint64_t acc = 0; int bits = 0; bool eof = false;
    BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    void fill() { while (bits <= 56 && pos < n) { acc = (acc << 8) | p[pos++]; bits += 8; } }
    uint32_t read(int k) {                                   // k <= 32
        if (k == 0) return 0;
        if (bits < k) { fill(); if (bits < k) { eof = true; return 0; } }
        const uint32_t v = (uint32_t)((acc >> (bits - k)) & (k == 32 ? 0xffffffffull : ((1ull << k) - 1)));
        bits -= k;
        return v;
    }
    int32_t read_signed(int k) { const uint32_t v = read(k); return k == 32 ? (int32_t)v : (int32_t)(v << (32 - k)) >> (32 - k); }
    uint32_t unary() {                                       // zeros up to the next one bit
        uint32_t z = 0;
        for (;;) {
            if (bits == 0) { fill(); if (bits == 0) { eof = true; return z; } }
            const uint64_t window = acc & (bits == 64 ? ~0ull : ((1ull << bits) - 1));
            if (window == 0) { z += bits; bits = 0; continue; }
            const int lead = __builtin_clzll(window) - (64 - bits);
            z += lead; bits -= lead + 1;
            return z;
        }
    }
    void align() { bits -= bits & 7; }
    size_t byte_pos() const { return pos - (size_t)(bits >> 3); }      // only meaningful when aligned
};

uint8_t crc8(const uint8_t* d, size_t n) {
    uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) { c ^= d[i]; for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1)); }
    return c;
}
uint16_t crc16(const uint8_t* d, size_t n) {
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) { c ^= (uint16_t)(d[i] << 8); for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : (c << 1)); }
    return c;
}

// ---- MD5 (RFC 1321) of the decoded PCM, little-endian samples of ceil(bps / 8) bytes, interleaved: what every FLAC encoder stores in STREAMINFO
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u; uint64_t len = 0; uint8_t buf[64]; int fill = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* m) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122,
            0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6,
            0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60,
            0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039,
            0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t w[16];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)m[4 * i] | ((uint32_t)m[4 * i + 1] << 8) | ((uint32_t)m[4 * i + 2] << 16) | ((uint32_t)m[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; ++i) {
            uint32_t f; int g;
            if (i < 16) { f = (B & C) | (~B & D); g = i; }
            else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
            else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
            else { f = C ^ (B | ~D); g = (7 * i) & 15; }
            const uint32_t t = D; D = C; C = B; B = B + rol(A + f + K[i] + w[g], S[i]); A = t;
        }
        a += A; b += B; c += C; d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        while (n) {
            const size_t k = (size_t)(64 - fill) < n ? (size_t)(64 - fill) : n;
            memcpy(buf + fill, p, k); fill += (int)k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    void final(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t l[8];
        for (int i = 0; i < 8; ++i) l[i] = (uint8_t)(bits >> (8 * i));
        update(l, 8);
        const uint32_t v[4] = {a, b, c, d};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(v[i] >> (8 * j));
    }
};

struct StreamInfo { int sr = 0, channels = 0, bps = 0; int64_t total = 0; uint8_t md5[16]; bool has_md5 = false; size_t first_frame = 0; int min_bs = 0, max_bs = 0; };

// returns "" or an error; leaves si.first_frame at the first audio frame
const char* parse_metadata(const uint8_t* d, size_t n, StreamInfo& si) {
    size_t p = 0;
    if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3')     // an ID3v2 tag in front of the stream (tolerated by every decoder)
        p = 10 + (((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (size_t)(d[9] & 0x7f));
    if (p + 4 > n || memcmp(d + p, "fLaC", 4) != 0) return "not a FLAC stream (no fLaC marker)";
    p += 4;
    bool last = false, have = false;
    while (!last) {
        if (p + 4 > n) return "truncated metadata";
        last = (d[p] & 0x80) != 0;
        const int type = d[p] & 0x7f;
        const size_t len = ((size_t)d[p + 1] << 16) | ((size_t)d[p + 2] << 8) | d[p + 3];
        p += 4;
        if (p + len > n) return "truncated metadata block";
        if (type == 0) {
            if (len < 34) return "STREAMINFO too short";
            const uint8_t* s = d + p;
            si.min_bs = (s[0] << 8) | s[1]; si.max_bs = (s[2] << 8) | s[3];
            si.sr = ((int)s[10] << 12) | ((int)s[11] << 4) | (s[12] >> 4);
            si.channels = ((s[12] >> 1) & 7) + 1;
            si.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            si.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
            memcpy(si.md5, s + 18, 16);
            si.has_md5 = false;
            for (int i = 0; i < 16; ++i) if (si.md5[i]) si.has_md5 = true;
            have = true;
        }
        p += len;
    }
    if (!have) return "no STREAMINFO block";
    if (si.sr <= 0 || si.bps < 4 || si.bps > 32) return "unsupported STREAMINFO (sample rate / bits per sample)";
    si.first_frame = p;
    return "";
}

const char* decode_residual(BitReader& br, int order, int bs, int32_t* out) {     // out[order .. bs)
    const int method = (int)br.read(2);
    if (method > 1) return "reserved residual coding method";
    const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
    const int porder = (int)br.read(4);
    const int parts = 1 << porder;
    if ((bs >> porder) << porder != bs && porder > 0) return "block size not divisible by the partition count";
    int i = order;
    for (int pt = 0; pt < parts; ++pt) {
        const int cnt = (pt == 0 ? (bs >> porder) - order : (bs >> porder));
        if (cnt < 0) return "partition shorter than the predictor order";
        const int k = (int)br.read(pbits);
        if (k == esc) {
            const int nb = (int)br.read(5);
            for (int j = 0; j < cnt; ++j) out[i++] = nb ? br.read_signed(nb) : 0;
        } else {
            for (int j = 0; j < cnt; ++j) {
                const uint32_t q = br.unary();
                const uint32_t u = (q << k) | br.read(k);
                out[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
            }
        }
        if (br.eof) return "truncated residual";
    }
    return "";
}

const char* decode_subframe(BitReader& br, int bs, int bps, int64_t* out) {
    if (br.read(1)) return "subframe padding bit set";
    const int type = (int)br.read(6);
    int wasted = 0;
    if (br.read(1)) wasted = (int)br.unary() + 1;
    bps -= wasted;
    if (bps < 1) return "wasted bits exceed the sample size";
    std::vector<int32_t> res(bs);
    auto rs = [&](int k) -> int64_t {                        // a k-bit signed sample, k up to 33 (side channel of 32-bit audio)
        if (k <= 32) return br.read_signed(k);
        const int64_t hi = br.read_signed(k - 32);
        return (hi << 32) | br.read(32);
    };
    if (type == 0) {
        const int64_t v = rs(bps);
        for (int i = 0; i < bs; ++i) out[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < bs; ++i) out[i] = rs(bps);
    } else if (type >= 8 && type <= 12) {
        const int order = type - 8;
        if (order > bs) return "fixed predictor order exceeds the block";
        for (int i = 0; i < order; ++i) out[i] = rs(bps);
        if (const char* e = decode_residual(br, order, bs, res.data()); *e) return e;
        for (int i = order; i < bs; ++i) {
            int64_t p = 0;
            switch (order) {
                case 1: p = out[i - 1]; break;
                case 2: p = 2 * out[i - 1] - out[i - 2]; break;
                case 3: p = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
                case 4: p = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
                default: break;
            }
            out[i] = p + res[i];
        }
    } else if (type >= 32) {
        const int order = type - 31;
        if (order > bs) return "LPC order exceeds the block";
        for (int i = 0; i < order; ++i) out[i] = rs(bps);
        const int prec = (int)br.read(4) + 1;
        if (prec == 16) return "reserved LPC precision";
        const int shift = br.read_signed(5);
        if (shift < 0) return "negative LPC shift";
        int32_t coef[32];
        for (int j = 0; j < order; ++j) coef[j] = br.read_signed(prec);
        if (const char* e = decode_residual(br, order, bs, res.data()); *e) return e;
        for (int i = order; i < bs; ++i) {
            int64_t p = 0;
            for (int j = 0; j < order; ++j) p += (int64_t)coef[j] * out[i - 1 - j];
            out[i] = (p >> shift) + res[i];
        }
    } else {
        return "reserved subframe type";
    }
    if (br.eof) return "truncated subframe";
    if (wasted) for (int i = 0; i < bs; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
    return "";
}

}  // namespace

// sample rate, channels, bits per sample and total frames (0 = unknown to the encoder) of a FLAC stream held in memory; 0 = ok
extern "C" int sylber_flac_info(const uint8_t* data, int64_t size, int32_t* sr, int32_t* channels, int32_t* bps, int64_t* frames) {
    StreamInfo si;
    const char* e = parse_metadata(data, (size_t)size, si);
    if (*e) { syl_set_error("sylber_flac_info", e); return 1; }
    *sr = si.sr; *channels = si.channels; *bps = si.bps; *frames = si.total;
    return 0;
}

// decodes the whole stream into out[capacity_frames][channels] (int32, the samples' own scale); *frames_out = frames decoded.  out == NULL: decode, verify and count only
// (a stream whose encoder did not know its length carries 0 frames in STREAMINFO).  Every frame's header CRC-8 and
// frame CRC-16 are checked; when the stream carries an MD5 of its unencoded audio (every mainstream encoder writes one) the decoded PCM is checked against it.
extern "C" int sylber_flac_decode(const uint8_t* data, int64_t size, int32_t* out, int64_t capacity_frames, int64_t* frames_out) {
    StreamInfo si;
    const char* e = parse_metadata(data, (size_t)size, si);
    if (*e) { syl_set_error("sylber_flac_decode", e); return 1; }
    const int nch = si.channels;
    size_t p = si.first_frame;
    int64_t done = 0;
    std::vector<int64_t> ch[8];
    Md5 md5;
    const int bytes_ps = (si.bps + 7) / 8;
    std::vector<uint8_t> pcm;
    while (p + 2 <= (size_t)size) {
        if (!(data[p] == 0xff && (data[p + 1] & 0xfe) == 0xf8)) {        // sync code 11111111 111110 + reserved 0
            bool pad = true;                                           // trailing padding / tags behind the last frame end the stream
            for (size_t q = p; q < (size_t)size && q < p + 16; ++q) if (data[q]) pad = false;
            if (pad || (si.total && done >= si.total)) break;
            syl_set_error("sylber_flac_decode", "lost frame sync"); return 1;
        }
        BitReader br(data + p, (size_t)size - p);
        br.read(15);
        br.read(1);                                                    // blocking strategy (the coded number is a frame or a sample number: not needed)
        const int bs_code = (int)br.read(4), sr_code = (int)br.read(4), ch_code = (int)br.read(4), ss_code = (int)br.read(3);
        if (br.read(1)) { syl_set_error("sylber_flac_decode", "reserved frame header bit set"); return 1; }
        {   // UTF-8-like coded number: 1..7 bytes
            const uint32_t b0 = br.read(8);
            int extra = 0;
            if (b0 & 0x80) { for (uint32_t m = 0x40; m && (b0 & m); m >>= 1) ++extra; if (extra == 0 || extra > 6) { syl_set_error("sylber_flac_decode", "bad coded frame number"); return 1; } }
            for (int i = 0; i < extra; ++i) br.read(8);
        }
        int bs;
        if (bs_code == 0) { syl_set_error("sylber_flac_decode", "reserved block size code"); return 1; }
        else if (bs_code == 1) bs = 192;
        else if (bs_code <= 5) bs = 576 << (bs_code - 2);
        else if (bs_code == 6) bs = (int)br.read(8) + 1;
        else if (bs_code == 7) bs = (int)br.read(16) + 1;
        else bs = 256 << (bs_code - 8);
        if (sr_code == 12) br.read(8); else if (sr_code == 13 || sr_code == 14) br.read(16); else if (sr_code == 15) { syl_set_error("sylber_flac_decode", "invalid sample rate code"); return 1; }
        static const int SS[8] = {0, 8, 12, 0, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? si.bps : SS[ss_code];
        if (bps == 0 || bps != si.bps) { syl_set_error("sylber_flac_decode", "frame sample size differs from STREAMINFO (not supported)"); return 1; }
        const size_t hdr_len = br.byte_pos();
        const uint32_t hcrc = br.read(8);
        if (br.eof || crc8(data + p, hdr_len) != hcrc) { syl_set_error("sylber_flac_decode", "frame header CRC-8 mismatch"); return 1; }
        int fch;
        if (ch_code < 8) fch = ch_code + 1; else if (ch_code <= 10) fch = 2; else { syl_set_error("sylber_flac_decode", "reserved channel assignment"); return 1; }
        if (fch != nch) { syl_set_error("sylber_flac_decode", "frame channel count differs from STREAMINFO"); return 1; }
        for (int c = 0; c < nch; ++c) {
            ch[c].assign((size_t)bs, 0);
            const int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
            const char* se = decode_subframe(br, bs, bps + side, ch[c].data());
            if (*se) { syl_set_error("sylber_flac_decode", se); return 1; }
        }
        br.align();
        const size_t body_len = br.byte_pos();
        const uint32_t fcrc = br.read(16);
        if (br.eof || crc16(data + p, body_len) != fcrc) { syl_set_error("sylber_flac_decode", "frame CRC-16 mismatch"); return 1; }
        if (ch_code == 8) for (int i = 0; i < bs; ++i) ch[1][i] = ch[0][i] - ch[1][i];                    // left, side -> right = left - side
        else if (ch_code == 9) for (int i = 0; i < bs; ++i) ch[0][i] = ch[0][i] + ch[1][i];               // side, right -> left = side + right
        else if (ch_code == 10) for (int i = 0; i < bs; ++i) {                                            // mid, side
            const int64_t s = ch[1][i], m = ((int64_t)((uint64_t)ch[0][i] << 1)) | (s & 1);
            ch[0][i] = (m + s) >> 1; ch[1][i] = (m - s) >> 1;
        }
        int take = bs;
        if (si.total && done + take > si.total) take = (int)(si.total - done);
        if (out && done + take > capacity_frames) { syl_set_error("sylber_flac_decode", "output buffer too small"); return 1; }
        pcm.resize((size_t)take * nch * bytes_ps);
        size_t w = 0;
        for (int i = 0; i < take; ++i)
            for (int c = 0; c < nch; ++c) {
                const int32_t v = (int32_t)ch[c][i];
                if (out) out[(done + i) * nch + c] = v;
                for (int b = 0; b < bytes_ps; ++b) pcm[w++] = (uint8_t)((uint32_t)v >> (8 * b));
            }
        md5.update(pcm.data(), pcm.size());
        done += take;
        p += body_len + 2;
    }
    if (si.total && done != si.total) { syl_set_error("sylber_flac_decode", "stream ends before the frame count of STREAMINFO"); return 1; }
    if (si.has_md5) {
        uint8_t dg[16];
        md5.final(dg);
        if (memcmp(dg, si.md5, 16) != 0) { syl_set_error("sylber_flac_decode", "MD5 of the decoded audio differs from the one the encoder stored in STREAMINFO"); return 1; }
    }
    *frames_out = done;
    return 0;
}
