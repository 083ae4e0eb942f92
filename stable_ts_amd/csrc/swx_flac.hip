// swx_flac.hip -- host-side FLAC stream decoder (no device code): the offline replacement of the reference's ffmpeg child
// process for .flac sources (stable_whisper/audio/utils.py:63-125 pipes every container through `ffmpeg -f s16le`; the
// reference's only real-speech fixture is test/jfk.flac and the GPU boxes have no ffmpeg).
//
// Written from the published format description (xiph.org "FLAC format" / RFC 9639): fLaC marker, metadata blocks (only
// STREAMINFO is read, the others are skipped by length), frames = header (sync, blocking strategy, block size, sample rate,
// channel assignment, sample size, UTF-8 coded number, CRC-8) + one subframe per channel (CONSTANT / VERBATIM / FIXED order
// 0-4 / LPC order 1-32, wasted bits, Rice / Rice2 residual partitions with escape) + padding + CRC-16; left/side, side/right
// and mid/side decorrelation.  Every read is bounds-checked; both CRCs are verified; a stream that ends or fails a check
// yields an error code, never a partial result presented as complete.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <vector>
#include "../../include/swx.h"

namespace {

struct BitReader {
    const uint8_t *p; size_t n; size_t byte = 0; uint64_t acc = 0; int have = 0; bool bad = false;
    BitReader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    inline void fill() { while (have <= 56 && byte < n) { acc |= (uint64_t)p[byte++] << (56 - have); have += 8; } }
    inline uint32_t bits(int k)            // k <= 32
    {
        if (k == 0) return 0;
        if (have < k) { fill(); if (have < k) { bad = true; return 0; } }
        const uint32_t v = (uint32_t)(acc >> (64 - k));
        acc <<= k; have -= k;
        return v;
    }
    inline int32_t sbits(int k)
    {
        if (k == 0) return 0;
        const uint32_t v = bits(k);
        return (int32_t)(v << (32 - k)) >> (32 - k);
    }
    inline int64_t sbits64(int k)          // k <= 33 (the side channel of a 32-bit stream)
    {
        if (k <= 32) return sbits(k);
        const uint64_t hi = bits(k - 32), lo = bits(32);
        const uint64_t v = (hi << 32) | lo;
        return (int64_t)(v << (64 - k)) >> (64 - k);
    }
    inline uint32_t unary()                // number of 0 bits before the next 1 bit
    {
        uint32_t q = 0;
        for (;;) {
            if (have == 0) { fill(); if (have == 0) { bad = true; return 0; } }
            if (acc == 0) { q += have; have = 0; continue; }      // (bits below `have` are zero by construction)
            const int z = __builtin_clzll(acc);
            if (z >= have) { q += have; acc = 0; have = 0; continue; }
            q += z;
            acc = (z == 63) ? 0 : acc << (z + 1);
            have -= (z + 1);
            return q;
        }
    }
    inline void align() { const int r = have & 7; acc <<= r; have -= r; }
    inline size_t pos_bytes() const { return byte - (size_t)(have >> 3); }     // only meaningful when byte-aligned
};

uint8_t g_crc8[256]; uint16_t g_crc16[256]; bool g_crc_ready = false;
void crc_init()
{
    if (g_crc_ready) return;
    for (int i = 0; i < 256; ++i) {
        uint8_t c = (uint8_t)i;
        for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : (c << 1));
        g_crc8[i] = c;
        uint16_t d = (uint16_t)(i << 8);
        for (int b = 0; b < 8; ++b) d = (uint16_t)((d & 0x8000) ? (d << 1) ^ 0x8005 : (d << 1));
        g_crc16[i] = d;
    }
    g_crc_ready = true;
}
uint8_t crc8(const uint8_t *p, size_t n) { uint8_t c = 0; for (size_t i = 0; i < n; ++i) c = g_crc8[c ^ p[i]]; return c; }
uint16_t crc16(const uint8_t *p, size_t n)
{
    uint16_t c = 0;
    for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ g_crc16[(c >> 8) ^ p[i]]);
    return c;
}

constexpr int E_ARG = -1, E_NOT_FLAC = -20, E_CORRUPT = -21, E_UNSUPPORTED = -22, E_TRUNC = -23;

int parse_streaminfo(const uint8_t *d, size_t n, swx_flac_info *info, size_t *first_frame)
{
    if (!d || n < 4 + 4 + 34 || memcmp(d, "fLaC", 4) != 0) return E_NOT_FLAC;
    size_t pos = 4;
    bool got = false;
    for (;;) {
        if (pos + 4 > n) return E_TRUNC;
        const bool last = d[pos] & 0x80;
        const int type = d[pos] & 0x7F;
        const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
        pos += 4;
        if (pos + len > n) return E_TRUNC;
        if (type == 0) {
            if (len < 34) return E_CORRUPT;
            const uint8_t *s = d + pos;
            info->min_block = (s[0] << 8) | s[1];
            info->max_block = (s[2] << 8) | s[3];
            info->sample_rate = ((int32_t)s[10] << 12) | ((int32_t)s[11] << 4) | (s[12] >> 4);
            info->channels = ((s[12] >> 1) & 7) + 1;
            info->bits_per_sample = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
            info->total_samples = ((int64_t)(s[13] & 0x0F) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) |
                                  ((int64_t)s[16] << 8) | s[17];
            memcpy(info->md5, s + 18, 16);
            got = true;
        }
        pos += len;
        if (last) break;
    }
    if (!got) return E_CORRUPT;
    if (info->sample_rate <= 0 || info->bits_per_sample < 4 || info->bits_per_sample > 32) return E_UNSUPPORTED;
    *first_frame = pos;
    return 0;
}

// residual of one subframe into res[order .. bs)
int read_residual(BitReader &br, int bs, int order, int64_t *res)
{
    const int method = br.bits(2);
    if (method > 1) return E_UNSUPPORTED;
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = br.bits(4);
    const int parts = 1 << porder;
    if ((bs >> porder) << porder != bs && porder > 0) return E_CORRUPT;
    if ((bs >> porder) < order && porder > 0) return E_CORRUPT;
    int i = order;
    for (int p = 0; p < parts; ++p) {
        int cnt = (bs >> porder) - (p == 0 ? order : 0);
        if (porder == 0) cnt = bs - order;
        if (cnt < 0 || i + cnt > bs) return E_CORRUPT;
        const int k = br.bits(pbits);
        if (k == esc) {
            const int raw = br.bits(5);
            for (int j = 0; j < cnt; ++j) res[i++] = br.sbits(raw);
        } else {
            for (int j = 0; j < cnt; ++j) {
                const uint32_t q = br.unary();
                const uint32_t r = br.bits(k);
                const uint64_t v = ((uint64_t)q << k) | r;
                res[i++] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
            }
        }
        if (br.bad) return E_TRUNC;
    }
    return 0;
}

int read_subframe(BitReader &br, int bs, int bps, int64_t *out)
{
    if (br.bits(1)) return E_CORRUPT;                    // padding bit
    const int type = br.bits(6);
    int wasted = 0;
    if (br.bits(1)) wasted = (int)br.unary() + 1;
    if (br.bad) return E_TRUNC;
    bps -= wasted;
    if (bps <= 0) return E_CORRUPT;
    if (type == 0) {                                      // CONSTANT
        const int64_t v = br.sbits64(bps);
        for (int i = 0; i < bs; ++i) out[i] = v;
    } else if (type == 1) {                               // VERBATIM
        for (int i = 0; i < bs; ++i) out[i] = br.sbits64(bps);
    } else if (type >= 8 && type <= 12) {                 // FIXED, order 0..4
        const int order = type - 8;
        if (order > bs) return E_CORRUPT;
        for (int i = 0; i < order; ++i) out[i] = br.sbits64(bps);
        const int rc = read_residual(br, bs, order, out);
        if (rc < 0) return rc;
        switch (order) {
            case 0: break;
            case 1: for (int i = 1; i < bs; ++i) out[i] += out[i - 1]; break;
            case 2: for (int i = 2; i < bs; ++i) out[i] += 2 * out[i - 1] - out[i - 2]; break;
            case 3: for (int i = 3; i < bs; ++i) out[i] += 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
            case 4: for (int i = 4; i < bs; ++i) out[i] += 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        }
    } else if (type >= 32) {                              // LPC, order 1..32
        const int order = (type & 31) + 1;
        if (order > bs) return E_CORRUPT;
        for (int i = 0; i < order; ++i) out[i] = br.sbits64(bps);
        const int prec = br.bits(4) + 1;
        if (prec == 16) return E_CORRUPT;
        const int shift = br.sbits(5);
        if (shift < 0) return E_UNSUPPORTED;
        int32_t coef[32];
        for (int i = 0; i < order; ++i) coef[i] = br.sbits(prec);
        const int rc = read_residual(br, bs, order, out);
        if (rc < 0) return rc;
        for (int i = order; i < bs; ++i) {
            int64_t acc = 0;
            for (int j = 0; j < order; ++j) acc += (int64_t)coef[j] * out[i - 1 - j];
            out[i] += acc >> shift;
        }
    } else {
        return E_CORRUPT;                                 // reserved subframe type
    }
    if (br.bad) return E_TRUNC;
    if (wasted) for (int i = 0; i < bs; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
    return 0;
}

const int kBlockSizes[16] = {0, 192, 576, 1152, 2304, 4608, 0, 0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768};
const int kSampleBits[8] = {0, 8, 12, 0, 16, 20, 24, 32};

}  // namespace

extern "C" {

int swx_flac_probe(const uint8_t *h_data, size_t n_bytes, swx_flac_info *info)
{
    if (!info) return E_ARG;
    size_t first = 0;
    return parse_streaminfo(h_data, n_bytes, info, &first);
}

int64_t swx_flac_decode(const uint8_t *h_data, size_t n_bytes, int32_t *h_out, int64_t capacity_frames, swx_flac_info *info)
{
    swx_flac_info si{};
    size_t pos = 0;
    int rc = parse_streaminfo(h_data, n_bytes, &si, &pos);
    if (rc < 0) return rc;
    if (info) *info = si;
    crc_init();
    const int C = si.channels;
    std::vector<int64_t> ch[8];
    int64_t n_out = 0;
    while (pos + 2 <= n_bytes) {
        // frame sync: 0xFF, 0b111110xx (14 sync bits, reserved 0); anything else here is trailing data or corruption
        if (h_data[pos] != 0xFF || (h_data[pos + 1] & 0xFE) != 0xF8) {
            if (si.total_samples && n_out >= si.total_samples) break;       // bytes after the last frame (ID3v1 tags etc.)
            return E_CORRUPT;
        }
        BitReader br(h_data + pos, n_bytes - pos);
        br.bits(15);
        br.bits(1);                                      // blocking strategy (the number below is a frame or a sample index)
        const int bs_code = br.bits(4), sr_code = br.bits(4), ch_code = br.bits(4), sz_code = br.bits(3);
        if (br.bits(1)) return E_CORRUPT;
        // UTF-8 style coded number: 1-7 bytes
        const uint32_t b0 = br.bits(8);
        int lead = 0;                                     // leading 1 bits of the first byte: 0 (one byte) or 2..7
        while (lead < 8 && (b0 & (0x80u >> lead))) ++lead;
        if (lead == 1 || lead == 8) return E_CORRUPT;
        for (int i = 1; i < lead; ++i) if ((br.bits(8) & 0xC0) != 0x80) return E_CORRUPT;
        int bs = kBlockSizes[bs_code];
        if (bs_code == 6) bs = (int)br.bits(8) + 1;
        else if (bs_code == 7) bs = (int)br.bits(16) + 1;
        if (bs <= 0) return E_CORRUPT;
        if (sr_code == 12) br.bits(8);
        else if (sr_code == 13 || sr_code == 14) br.bits(16);
        else if (sr_code == 15) return E_CORRUPT;
        if (br.bad) return E_TRUNC;
        const size_t hdr_len = br.pos_bytes();
        const uint32_t c8 = br.bits(8);
        if (br.bad) return E_TRUNC;
        if (crc8(h_data + pos, hdr_len) != c8) return E_CORRUPT;
        int bps = kSampleBits[sz_code];
        if (sz_code == 0) bps = si.bits_per_sample;
        if (sz_code == 3 || bps != si.bits_per_sample) return E_UNSUPPORTED;      // a stream whose sample size changes
        int nch = ch_code < 8 ? ch_code + 1 : 2;
        if (ch_code > 10 || nch != C) return E_UNSUPPORTED;
        for (int c = 0; c < nch; ++c) {
            ch[c].resize(bs);
            int cb = bps;
            if ((ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1)) cb = bps + 1;   // the side channel
            rc = read_subframe(br, bs, cb, ch[c].data());
            if (rc < 0) return rc;
        }
        br.align();
        const size_t body_len = br.pos_bytes();
        const uint32_t c16 = br.bits(16);
        if (br.bad) return E_TRUNC;
        if (crc16(h_data + pos, body_len) != c16) return E_CORRUPT;
        if (ch_code == 8) { for (int i = 0; i < bs; ++i) ch[1][i] = ch[0][i] - ch[1][i]; }                 // left, side
        else if (ch_code == 9) { for (int i = 0; i < bs; ++i) ch[0][i] = ch[0][i] + ch[1][i]; }            // side, right
        else if (ch_code == 10) {                                                                            // mid, side
            for (int i = 0; i < bs; ++i) {
                const int64_t side = ch[1][i];
                const int64_t mid = (int64_t)(((uint64_t)ch[0][i] << 1) | (uint64_t)(side & 1));
                ch[0][i] = (mid + side) >> 1;
                ch[1][i] = (mid - side) >> 1;
            }
        }
        int take = bs;
        if (si.total_samples && n_out + take > si.total_samples) take = (int)(si.total_samples - n_out);
        if (h_out) {
            if (n_out + take > capacity_frames) return E_ARG;
            for (int c = 0; c < nch; ++c) {
                int32_t *o = h_out + n_out * C + c;
                const int64_t *s = ch[c].data();
                for (int i = 0; i < take; ++i) o[(size_t)i * C] = (int32_t)s[i];
            }
        }
        n_out += take;
        pos += body_len + 2;
    }
    if (si.total_samples && n_out < si.total_samples) return E_TRUNC;
    return n_out;
}

}  // extern "C"
