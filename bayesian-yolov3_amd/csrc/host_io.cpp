// host_io.cpp -- the host side of the drop-in entry points, native so that it keeps up with the device:
//
//   byolo_png_decode_batch   the `tf.image.decode_png` of `decode_img` (lib_yolo/dataset_utils.py:6-11) for a batch of
//                            records on a pool of threads -- the `map(..., num_parallel_calls=cpu_thread_cnt)` stage of
//                            `TestingDataset` (lib_yolo/dataset_utils.py:196);
//   byolo_format_ecp_json    `json.dump({'children': [bbox_to_ecp_format(b) ...]})` of the three inference scripts
//                            (inference_epistemic.py:131-170, inference_aleatoric.py:139-178,
//                            inference_standard_yolov3.py:128-150, writer :84-92), BYTE-identical to what CPython's
//                            json module writes for the same rows (tests/test_host_io.py compares the two).
//
// Pure host code (no HIP): called through ctypes, which drops the GIL for the duration of the call.
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/byolo.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// PNG (ISO/IEC 15948): 8-bit greyscale / RGB / grey+alpha / RGBA, non-interlaced.  Anything else -> BYOLO_PNG_UNSUPPORTED
// and the caller decodes that record with its general decoder.
// ---------------------------------------------------------------------------------------------------------------------
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the scanline filter of `cur` (row bytes, without the filter-type byte) in place; `prev` = the row above (nullptr: zeros)
template <int BPP>
bool unfilter_row(int ft, uint8_t* cur, const uint8_t* prev, size_t n) {
    switch (ft) {
    case 0: return true;
    case 1:
        for (size_t i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);
        return true;
    case 2:
        if (prev) for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
        return true;
    case 3:
        for (size_t i = 0; i < n; ++i) {
            const int a = i >= BPP ? cur[i - BPP] : 0, b = prev ? prev[i] : 0;
            cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
        }
        return true;
    case 4:
        for (size_t i = 0; i < n; ++i) {
            const int a = i >= BPP ? cur[i - BPP] : 0, b = prev ? prev[i] : 0, c = (prev && i >= BPP) ? prev[i - BPP] : 0;
            cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
        }
        return true;
    default: return false;
    }
}

// Scratch buffers (a record's payload, a frame's filtered scanlines: about a megabyte each) are kept and reused across calls: a
// fresh one per frame means an mmap, a few hundred page faults and a munmap each time, and those serialise the threads of a
// process on its address-space lock (measured: 8 decoding threads gave 1.2x of one).
struct BufPool {
    std::mutex m;
    std::vector<std::vector<uint8_t>> free;
    size_t held = 0;                                   // bytes parked in `free`
    // at most one scratch buffer per hardware thread and 256 MB in total stay parked (a 1024 x 1920 x 3 frame is 6 MB; round 4
    // kept up to 128 buffers of the largest size ever asked for: ~0.75 GB for the life of the process -- ADVICE r4)
    const size_t max_bufs = std::max(4u, std::min(128u, std::thread::hardware_concurrency()));
    static constexpr size_t MAX_HELD = (size_t)256 << 20;
    std::vector<uint8_t> get(size_t n) {
        std::vector<uint8_t> b;
        {
            std::lock_guard<std::mutex> g(m);
            if (!free.empty()) { b = std::move(free.back()); free.pop_back(); held -= b.capacity(); }
        }
        if (b.capacity() < n) b.clear();               // growth must not copy the old contents
        if (b.size() < n) b.resize(n);
        return b;
    }
    void put(std::vector<uint8_t>&& b) {
        std::lock_guard<std::mutex> g(m);
        if (free.size() < max_bufs && held + b.capacity() <= MAX_HELD) { held += b.capacity(); free.push_back(std::move(b)); }
    }
};
BufPool g_pool;
struct Scratch {
    std::vector<uint8_t> v;
    explicit Scratch(size_t n) : v(g_pool.get(n)) {}
    ~Scratch() { g_pool.put(std::move(v)); }
    uint8_t* data() { return v.data(); }
};

int32_t png_decode_one(const uint8_t* png, size_t n, int32_t H, int32_t W, int32_t C, uint8_t* out, int32_t found[3]) {
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    found[0] = found[1] = found[2] = 0;
    if (n < 8 + 25 || memcmp(png, SIG, 8) != 0) return BYOLO_PNG_CORRUPT;
    size_t pos = 8;
    bool have_ihdr = false, done = false, stream_open = false;
    int channels = 0;
    size_t stride = 0, total = 0;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    std::unique_ptr<Scratch> raw;                      // filter byte + row bytes, all rows
    int32_t rc = BYOLO_PNG_OK;
    while (pos + 12 <= n) {
        const uint32_t len = be32(png + pos);
        const uint8_t* type = png + pos + 4;
        if ((size_t)len > n - pos - 12) { rc = BYOLO_PNG_CORRUPT; break; }
        const uint8_t* data = png + pos + 8;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(data + len)) { rc = BYOLO_PNG_CORRUPT; break; }
        pos += 12 + (size_t)len;
        if (!have_ihdr) {
            if (memcmp(type, "IHDR", 4) != 0 || len != 13) { rc = BYOLO_PNG_CORRUPT; break; }
            const uint32_t w = be32(data), h = be32(data + 4);
            const int depth = data[8], color = data[9], comp = data[10], filt = data[11], interlace = data[12];
            if (w == 0 || h == 0 || w > (1u << 20) || h > (1u << 20) || comp != 0 || filt != 0) { rc = BYOLO_PNG_CORRUPT; break; }
            channels = color == 0 ? 1 : color == 2 ? 3 : color == 4 ? 2 : color == 6 ? 4 : 0;
            if (depth != 8 || channels == 0 || interlace != 0) { rc = BYOLO_PNG_UNSUPPORTED; break; }
            found[0] = (int32_t)h; found[1] = (int32_t)w; found[2] = channels;
            if ((int32_t)h != H || (int32_t)w != W || channels != C) { rc = BYOLO_PNG_SHAPE; break; }
            stride = (size_t)W * C;
            total = (size_t)H * (stride + 1);
            raw.reset(new Scratch(total));
            if (inflateInit(&zs) != Z_OK) { rc = BYOLO_PNG_CORRUPT; break; }
            stream_open = true;
            zs.next_out = raw->data();
            zs.avail_out = (uInt)total;                  // H, W <= 2^20 and C <= 4 were checked; a frame beyond 4 GiB is refused below
            if (total > 0xFFFFFFF0ull) { rc = BYOLO_PNG_UNSUPPORTED; break; }
            have_ihdr = true;
            continue;
        }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (done) continue;                          // data after the end of the zlib stream: ignored, like libpng
            zs.next_in = const_cast<Bytef*>(data);
            zs.avail_in = len;
            while (zs.avail_in > 0) {
                // with the frame complete (avail_out == 0) the stream may still hold its end-of-block code and checksum,
                // which need no output space; Z_BUF_ERROR then means more image data than the header announces
                const int z = inflate(&zs, Z_NO_FLUSH);
                if (z == Z_STREAM_END) { done = true; break; }
                if (z != Z_OK) { rc = BYOLO_PNG_CORRUPT; break; }
            }
            if (rc != BYOLO_PNG_OK) break;
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }                                                // ancillary chunks: skipped
    }
    if (stream_open) {
        if (rc == BYOLO_PNG_OK && zs.total_out != total) rc = BYOLO_PNG_CORRUPT;
        inflateEnd(&zs);
    } else if (rc == BYOLO_PNG_OK) {
        rc = BYOLO_PNG_CORRUPT;                          // no IHDR / no data
    }
    if (rc != BYOLO_PNG_OK) return rc;
    const uint8_t* prev = nullptr;
    for (int32_t y = 0; y < H; ++y) {
        uint8_t* row = raw->data() + (size_t)y * (stride + 1);
        uint8_t* dst = out + (size_t)y * stride;
        const int ft = row[0];
        memcpy(dst, row + 1, stride);
        bool ok;
        switch (C) {
        case 1: ok = unfilter_row<1>(ft, dst, prev, stride); break;
        case 2: ok = unfilter_row<2>(ft, dst, prev, stride); break;
        case 3: ok = unfilter_row<3>(ft, dst, prev, stride); break;
        default: ok = unfilter_row<4>(ft, dst, prev, stride); break;
        }
        if (!ok) return BYOLO_PNG_CORRUPT;
        prev = dst;
    }
    return BYOLO_PNG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// ECP JSON
// ---------------------------------------------------------------------------------------------------------------------
struct Out {
    char* p; char* end; size_t need = 0;                 // `need` keeps counting when the buffer is full
    void put(const char* s, size_t n) {
        need += n;
        if ((size_t)(end - p) >= n) { memcpy(p, s, n); p += n; } else { p = end; }
    }
    void lit(const char* s) { put(s, strlen(s)); }
};

// float.__repr__ of CPython (Python/pystrtod.c format_float_short, format code 'r'): the shortest digit string that
// round-trips, in positional notation for -4 <= exponent10 < 16 (with ".0" appended to an integer), else d.ddde+XX with at
// least two exponent digits; json.dumps(allow_nan=True) spells the non-finite values NaN / Infinity / -Infinity.
void put_double(Out& o, double v) {
    if (std::isnan(v)) { o.lit("NaN"); return; }
    if (std::isinf(v)) { o.lit(v > 0 ? "Infinity" : "-Infinity"); return; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX, shortest
    char* s = sci;
    char buf[48];
    char* b = buf;
    if (*s == '-') { *b++ = '-'; ++s; }
    char digits[24];
    int nd = 0;
    char* e = s;
    for (; e < r.ptr && *e != 'e'; ++e) if (*e != '.') digits[nd++] = *e;
    int x = 0;
    {
        const char* q = e + 1;
        const bool neg = *q == '-';
        if (*q == '-' || *q == '+') ++q;
        for (; q < r.ptr; ++q) x = 10 * x + (*q - '0');
        if (neg) x = -x;
    }
    const int decpt = x + 1;                              // digits * 10^(decpt - nd), 0.d1d2... * 10^decpt
    if (decpt > 16 || decpt <= -4) {
        *b++ = digits[0];
        if (nd > 1) { *b++ = '.'; memcpy(b, digits + 1, nd - 1); b += nd - 1; }
        *b++ = 'e';
        int ex = decpt - 1;
        if (ex < 0) { *b++ = '-'; ex = -ex; } else { *b++ = '+'; }
        char t[8]; int nt = 0;
        do { t[nt++] = (char)('0' + ex % 10); ex /= 10; } while (ex);
        if (nt < 2) t[nt++] = '0';
        while (nt) *b++ = t[--nt];
    } else if (decpt <= 0) {
        *b++ = '0'; *b++ = '.';
        for (int i = 0; i < -decpt; ++i) *b++ = '0';
        memcpy(b, digits, nd); b += nd;
    } else if (decpt >= nd) {
        memcpy(b, digits, nd); b += nd;
        for (int i = 0; i < decpt - nd; ++i) *b++ = '0';
        *b++ = '.'; *b++ = '0';
    } else {
        memcpy(b, digits, decpt); b += decpt;
        *b++ = '.';
        memcpy(b, digits + decpt, nd - decpt); b += nd - decpt;
    }
    o.put(buf, (size_t)(b - buf));
}

inline void kv(Out& o, const char* key, double v, bool first = false) {
    if (!first) o.lit(", ");
    o.lit("\""); o.lit(key); o.lit("\": ");
    put_double(o, v);
}

}  // namespace

namespace {
template <class F>
void run_pool(int32_t n, int32_t threads, F&& one) {
    std::atomic<int32_t> next{0};
    auto work = [&] {
        for (;;) {
            const int32_t i = next.fetch_add(1);
            if (i >= n) return;
            one(i);
        }
    };
    int nt = threads < 1 ? 1 : threads;
    if (nt > n) nt = n;
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    try { for (int t = 0; t < nt - 1; ++t) pool.emplace_back(work); } catch (...) {}          // fewer threads is still correct
    work();
    for (auto& t : pool) t.join();
}

// ---- tf.train.Example: Example{1: Features{1: map<string, Feature>}}, Feature{1: BytesList{1: bytes}} -- the first bytes value
// of the features `image/encoded` and `image/filename` (create_tf_records_citypersons.py:132-147), in place
struct Span { const uint8_t* p = nullptr; size_t n = 0; };
bool varint(const uint8_t*& p, const uint8_t* e, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
        if (p >= e) return false;
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}
// next field of the message [p, e): number, wire type, and for length-delimited fields the payload; false = end or malformed (ok tells)
bool next_field(const uint8_t*& p, const uint8_t* e, uint32_t& num, uint32_t& wt, Span& val, bool& ok) {
    if (p >= e) return false;
    uint64_t key, len;
    if (!varint(p, e, key)) { ok = false; return false; }
    num = (uint32_t)(key >> 3); wt = (uint32_t)(key & 7);
    val = Span{};
    switch (wt) {
    case 0: if (!varint(p, e, len)) { ok = false; return false; } return true;
    case 1: if (e - p < 8) { ok = false; return false; } p += 8; return true;
    case 5: if (e - p < 4) { ok = false; return false; } p += 4; return true;
    case 2:
        if (!varint(p, e, len) || len > (uint64_t)(e - p)) { ok = false; return false; }
        val = Span{p, (size_t)len}; p += len; return true;
    default: ok = false; return false;
    }
}
bool example_find(const uint8_t* rec, size_t n, Span& encoded, Span& filename) {
    bool ok = true;
    uint32_t num, wt;
    Span features, entry, f;
    const uint8_t* p = rec;
    while (next_field(p, rec + n, num, wt, features, ok)) {
        if (num != 1 || wt != 2) continue;
        const uint8_t* q = features.p;
        while (next_field(q, features.p + features.n, num, wt, entry, ok)) {
            if (num != 1 || wt != 2) continue;
            Span key, feature;
            const uint8_t* r = entry.p;
            while (next_field(r, entry.p + entry.n, num, wt, f, ok)) {
                if (wt != 2) continue;
                if (num == 1) key = f; else if (num == 2) feature = f;
            }
            Span* dst = nullptr;
            if (key.n == 13 && memcmp(key.p, "image/encoded", 13) == 0) dst = &encoded;
            else if (key.n == 14 && memcmp(key.p, "image/filename", 14) == 0) dst = &filename;
            if (!dst || !ok) continue;
            Span lst, v;
            const uint8_t* t = feature.p;
            bool have = false;
            while (!have && next_field(t, feature.p + feature.n, num, wt, lst, ok)) {
                if (num != 1 || wt != 2) continue;                       // bytes_list
                const uint8_t* u = lst.p;
                while (next_field(u, lst.p + lst.n, num, wt, v, ok)) {
                    if (num == 1 && wt == 2) { *dst = v; have = true; break; }
                }
            }
        }
    }
    return ok;
}
}  // namespace

extern "C" int32_t byolo_png_decode_batch(const uint8_t* const* h_png, const size_t* png_bytes, int32_t n, int32_t img_h,
                                          int32_t img_w, int32_t img_c, uint8_t* h_out, int32_t threads, int32_t* status,
                                          int32_t* found_shape) {
    if (n < 0 || img_h <= 0 || img_w <= 0 || img_c <= 0 || img_c > 4 || (n > 0 && (!h_png || !png_bytes || !h_out || !status)))
        return BYOLO_ERR_ARG;
    const size_t frame = (size_t)img_h * img_w * img_c;
    run_pool(n, threads, [&](int32_t i) {
        int32_t found[3] = {0, 0, 0};
        int32_t rc;
        try {
            rc = h_png[i] ? png_decode_one(h_png[i], png_bytes[i], img_h, img_w, img_c, h_out + (size_t)i * frame, found)
                          : (int32_t)BYOLO_PNG_CORRUPT;
        } catch (...) { rc = BYOLO_PNG_CORRUPT; }                          // bad_alloc on a hostile header
        status[i] = rc;
        if (found_shape) memcpy(found_shape + 3 * (size_t)i, found, sizeof found);
    });
    int32_t bad = 0;
    for (int32_t i = 0; i < n; ++i) bad += status[i] != BYOLO_PNG_OK;
    return bad;
}

extern "C" int32_t byolo_feed_records(const int32_t* fds, const int64_t* offsets, const int64_t* lengths, int32_t n, int32_t verify_crc,
                                      int32_t img_h, int32_t img_w, int32_t img_c, uint8_t* h_out, int32_t threads, char* h_names,
                                      int32_t name_cap, int32_t* status, int32_t* found_shape) {
    if (n < 0 || img_h <= 0 || img_w <= 0 || img_c <= 0 || img_c > 4 || name_cap < 1 ||
        (n > 0 && (!fds || !offsets || !lengths || !h_out || !h_names || !status)))
        return BYOLO_ERR_ARG;
    const size_t frame = (size_t)img_h * img_w * img_c;
    run_pool(n, threads, [&](int32_t i) {
        int32_t found[3] = {0, 0, 0};
        int32_t rc = BYOLO_FEED_IO;
        h_names[(size_t)i * name_cap] = 0;
        try {
            do {
                if (lengths[i] < 0 || offsets[i] < 0) break;
                const size_t len = (size_t)lengths[i];
                Scratch rec(len + 4);
                size_t got = 0;
                while (got < len + 4) {                                    // payload + its masked CRC-32C
                    const ssize_t r = pread(fds[i], rec.data() + got, len + 4 - got, (off_t)(offsets[i] + (int64_t)got));
                    if (r <= 0) break;
                    got += (size_t)r;
                }
                if (got < len + 4) break;
                if (verify_crc) {
                    const uint32_t crc = byolo_crc32c(rec.data(), len);
                    const uint32_t masked = (uint32_t)(((crc >> 15) | (crc << 17)) + 0xA282EAD8u);
                    uint32_t want;
                    memcpy(&want, rec.data() + len, 4);                    // little-endian on disk, as this host
                    if (masked != want) { rc = BYOLO_FEED_CRC; break; }
                }
                Span enc, name;
                if (!example_find(rec.data(), len, enc, name) || !enc.p) { rc = BYOLO_FEED_PROTO; break; }
                if (name.n >= (size_t)name_cap) { rc = BYOLO_FEED_PROTO; break; }
                if (name.n) memcpy(h_names + (size_t)i * name_cap, name.p, name.n);
                h_names[(size_t)i * name_cap + name.n] = 0;
                rc = png_decode_one(enc.p, enc.n, img_h, img_w, img_c, h_out + (size_t)i * frame, found);
            } while (false);
        } catch (...) { rc = BYOLO_PNG_CORRUPT; }
        status[i] = rc;
        if (found_shape) memcpy(found_shape + 3 * (size_t)i, found, sizeof found);
    });
    int32_t bad = 0;
    for (int32_t i = 0; i < n; ++i) bad += status[i] != BYOLO_PNG_OK;
    return bad;
}

extern "C" int64_t byolo_format_ecp_json(int32_t kind, const float* h_rows, int32_t n_rows, int32_t row_len, int32_t img_h,
                                         int32_t img_w, int32_t cls_cnt, int32_t obj_idx, int32_t cls_start,
                                         int32_t implicit_background, const char* const* labels, int32_t n_labels,
                                         char* h_out, size_t cap) {
    const int C = cls_cnt;
    const int need_cols = kind == BYOLO_DET_STANDARD ? cls_start + C : kind == BYOLO_DET_ALEATORIC ? cls_start + C + 1 : cls_start + C + 4;
    if (kind < 0 || kind > 2 || n_rows < 0 || C < 1 || obj_idx < 0 || cls_start < 0 || row_len < need_cols || (n_rows && !h_rows) ||
        (kind == BYOLO_DET_EPISTEMIC && row_len < 19) || (cap && !h_out))
        return BYOLO_ERR_ARG;
    Out o{h_out, h_out + cap};
    const float fh = (float)img_h, fw = (float)img_w;
    o.lit("{\"children\": [");
    for (int32_t r = 0; r < n_rows; ++r) {
        const float* b = h_rows + (size_t)r * row_len;
        if (r) o.lit(", ");
        o.lit("{");
        // float(bbox[0] * img_height): the product is a float32 (numpy scalar * Python int), then widened
        kv(o, "y0", (double)(float)(b[0] * fh), true);
        kv(o, "x0", (double)(float)(b[1] * fw));
        kv(o, "y1", (double)(float)(b[2] * fh));
        kv(o, "x1", (double)(float)(b[3] * fw));
        // np.argmax: the first maximum; a NaN is the maximum
        int cls = 0;
        for (int k = 0; k < C; ++k) {
            const float v = b[cls_start + k];
            if (std::isnan(v)) { cls = k; break; }
            if (v > b[cls_start + cls]) cls = k;
        }
        const double score = (double)b[obj_idx] * (double)b[cls_start + cls];
        auto cls_scores = [&] {
            o.lit(", \"cls_scores\": [");
            for (int k = 0; k < C; ++k) { if (k) o.lit(", "); put_double(o, (double)b[cls_start + k]); }
            o.lit("]");
        };
        if (kind == BYOLO_DET_STANDARD) {
            kv(o, "score", score);
            cls_scores();
        } else if (kind == BYOLO_DET_ALEATORIC) {
            static const char* K[5] = {"x_var", "y_var", "w_var", "h_var", "total_var"};
            for (int i = 0; i < 5; ++i) kv(o, K[i], (double)b[4 + i]);
            kv(o, "score", score);
            kv(o, "obj_entropy", (double)b[obj_idx + 1]);
            cls_scores();
            kv(o, "cls_entropy", (double)b[cls_start + C]);     // the reference reads all three from this column
            kv(o, "layer_id", (double)b[cls_start + C]);        // (inference_aleatoric.py:174-176)
            kv(o, "prior_id", (double)b[cls_start + C]);
        } else {
            static const char* K[10] = {"x_var_epi", "y_var_epi", "w_var_epi", "h_var_epi", "x_var_ale", "y_var_ale", "w_var_ale",
                                        "h_var_ale", "total_var_epi", "total_var_ale"};
            for (int i = 0; i < 10; ++i) kv(o, K[i], (double)b[4 + i]);
            kv(o, "score", score);
            kv(o, "obj_mutual_info", (double)b[obj_idx + 1]);
            kv(o, "obj_entropy", (double)b[obj_idx + 2]);
            cls_scores();
            kv(o, "ped_score", (double)b[17]);                  // hard-coded columns (inference_epistemic.py:163-164)
            kv(o, "rider_score", (double)b[18]);
            kv(o, "cls_mutual_info", (double)b[cls_start + C]);
            kv(o, "cls_entropy", (double)b[cls_start + C + 1]);
            kv(o, "layer_id", (double)b[cls_start + C + 2]);
            kv(o, "prior_id", (double)b[cls_start + C + 3]);
        }
        const int ident = cls + (implicit_background ? 1 : 0);
        o.lit(", \"identity\": ");
        if (labels && ident >= 0 && ident < n_labels && labels[ident]) {
            o.lit("\""); o.lit(labels[ident]); o.lit("\"");       // the caller passes JSON-safe ASCII names only
        } else {
            char t[16];
            const auto rr = std::to_chars(t, t + sizeof t, ident);
            o.put(t, (size_t)(rr.ptr - t));
        }
        o.lit("}");
    }
    o.lit("]}");
    return o.need <= cap ? (int64_t)o.need : -(int64_t)o.need - 16;     // too small: -(needed) - 16 (clear of the error codes)
}
