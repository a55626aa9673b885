// tfrecord_pipeline.cpp -- host-side input pipeline (libsavp_io.so).  See include/savp_io.h for the reference lines each
// entry point replaces.  Plain C++17 + pthreads; no TensorFlow, no protobuf library: the two wire formats involved
// (TFRecord framing, tf.train.Example) are restated from their published definitions.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "savp_io.h"

// ------------------------------------------------------------------------------------------------------------------------
// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slice-by-8
// ------------------------------------------------------------------------------------------------------------------------
namespace {
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
const CrcTables& tables() { static CrcTables T; return T; }
}  // namespace

extern "C" uint32_t savp_io_crc32c(const void* data, uint64_t n) {
    const CrcTables& T = tables();
    const uint8_t* p = (const uint8_t*)data;
    uint32_t c = 0xffffffffu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
            T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff];
    return c ^ 0xffffffffu;
}

extern "C" uint32_t savp_io_masked_crc32c(const void* data, uint64_t n) {
    const uint32_t c = savp_io_crc32c(data, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// ------------------------------------------------------------------------------------------------------------------------
// TFRecord framing: uint64 length | uint32 masked_crc(length) | data | uint32 masked_crc(data)   (little endian)
// ------------------------------------------------------------------------------------------------------------------------
struct SavpTfrFile {
    FILE* fp = nullptr;
    std::vector<char> iobuf;
    std::vector<uint8_t> rec;
};

extern "C" int savp_tfr_open(const char* path, int64_t buffer_bytes, SavpTfrFile** out) {
    if (!path || !out) return SAVP_IO_EINVAL;
    FILE* fp = fopen(path, "rb");
    if (!fp) return SAVP_IO_EIO;
    SavpTfrFile* f = new SavpTfrFile();
    f->fp = fp;
    if (buffer_bytes > 0) {
        f->iobuf.resize((size_t)buffer_bytes);
        setvbuf(fp, f->iobuf.data(), _IOFBF, f->iobuf.size());
    }
    *out = f;
    return SAVP_IO_OK;
}

extern "C" int savp_tfr_next(SavpTfrFile* f, const uint8_t** data, uint64_t* len) {
    if (!f || !data || !len) return SAVP_IO_EINVAL;
    uint8_t hdr[12];
    const size_t got = fread(hdr, 1, 12, f->fp);
    if (got == 0) return SAVP_IO_EOF;
    if (got != 12) return SAVP_IO_ECORRUPT;
    uint64_t n; uint32_t crc;
    memcpy(&n, hdr, 8); memcpy(&crc, hdr + 8, 4);
    if (savp_io_masked_crc32c(hdr, 8) != crc) return SAVP_IO_ECORRUPT;
    if (n > (1ull << 32)) return SAVP_IO_ECORRUPT;
    f->rec.resize((size_t)n + 4);
    if (fread(f->rec.data(), 1, (size_t)n + 4, f->fp) != (size_t)n + 4) return SAVP_IO_ECORRUPT;
    memcpy(&crc, f->rec.data() + n, 4);
    if (savp_io_masked_crc32c(f->rec.data(), n) != crc) return SAVP_IO_ECORRUPT;
    *data = f->rec.data(); *len = n;
    return SAVP_IO_OK;
}

extern "C" void savp_tfr_close(SavpTfrFile* f) {
    if (!f) return;
    if (f->fp) fclose(f->fp);
    delete f;
}

// ------------------------------------------------------------------------------------------------------------------------
// tf.train.Example wire format (proto3):
//   Example   { Features features = 1; }
//   Features  { map<string, Feature> feature = 1; }       -- repeated entry { string key = 1; Feature value = 2; }
//   Feature   { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
//   BytesList { repeated bytes value = 1; }   FloatList { repeated float value = 1 [packed]; }   Int64List { repeated int64 value = 1 [packed]; }
// ------------------------------------------------------------------------------------------------------------------------
namespace {
struct Span { const uint8_t* p; uint64_t n; };

bool varint(const uint8_t*& p, const uint8_t* end, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && p < end; shift += 7) {
        const uint8_t b = *p++;
        v |= (uint64_t)(b & 0x7f) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// next field of a message; returns false at the end or on malformed input (ok tells which)
bool next_field(const uint8_t*& p, const uint8_t* end, uint32_t& field, uint32_t& wire, Span& payload, uint64_t& value, bool& ok) {
    ok = true;
    if (p >= end) return false;
    uint64_t key;
    if (!varint(p, end, key)) { ok = false; return false; }
    field = (uint32_t)(key >> 3); wire = (uint32_t)(key & 7);
    payload = Span{nullptr, 0}; value = 0;
    switch (wire) {
        case 0: if (!varint(p, end, value)) { ok = false; return false; } return true;
        case 1: if (end - p < 8) { ok = false; return false; } payload = Span{p, 8}; p += 8; return true;
        case 2: {
            uint64_t n;
            if (!varint(p, end, n) || (uint64_t)(end - p) < n) { ok = false; return false; }
            payload = Span{p, n}; p += n; return true;
        }
        case 5: if (end - p < 4) { ok = false; return false; } payload = Span{p, 4}; p += 4; return true;
        default: ok = false; return false;
    }
}

// locate the Feature message of `name`; returns 0 / EMISSING / ECORRUPT
int find_feature(const uint8_t* ex, uint64_t ex_len, const char* name, size_t name_len, Span& feat) {
    const uint8_t *p = ex, *end = ex + ex_len;
    uint32_t f, w; Span s; uint64_t v; bool ok;
    while (next_field(p, end, f, w, s, v, ok)) {
        if (f != 1 || w != 2) continue;                                  // Example.features
        const uint8_t *q = s.p, *qend = s.p + s.n;
        uint32_t f2, w2; Span s2; uint64_t v2; bool ok2;
        while (next_field(q, qend, f2, w2, s2, v2, ok2)) {
            if (f2 != 1 || w2 != 2) continue;                            // Features.feature entry
            const uint8_t *r = s2.p, *rend = s2.p + s2.n;
            uint32_t f3, w3; Span s3; uint64_t v3; bool ok3;
            Span key{nullptr, 0}, val{nullptr, 0};
            while (next_field(r, rend, f3, w3, s3, v3, ok3)) {
                if (f3 == 1 && w3 == 2) key = s3;
                else if (f3 == 2 && w3 == 2) val = s3;
            }
            if (!ok3) return SAVP_IO_ECORRUPT;
            if (key.n == name_len && memcmp(key.p, name, name_len) == 0) { feat = val; return SAVP_IO_OK; }
        }
        if (!ok2) return SAVP_IO_ECORRUPT;
    }
    return ok ? SAVP_IO_EMISSING : SAVP_IO_ECORRUPT;
}
}  // namespace

extern "C" int savp_example_feature(const uint8_t* ex, uint64_t ex_len, const char* name, int32_t index, int32_t* kind,
                                    const uint8_t** ptr, uint64_t* len) {
    if (!ex || !name || !kind || !ptr || !len || index < 0) return SAVP_IO_EINVAL;
    Span feat;
    int rc = find_feature(ex, ex_len, name, strlen(name), feat);
    if (rc) return rc;
    const uint8_t *p = feat.p, *end = feat.p + feat.n;
    uint32_t f, w; Span s; uint64_t v; bool ok;
    *kind = 0; *ptr = nullptr; *len = 0;
    while (next_field(p, end, f, w, s, v, ok)) {
        if (w != 2 || f < 1 || f > 3) continue;
        *kind = (int32_t)f;
        const uint8_t *q = s.p, *qend = s.p + s.n;
        uint32_t f2, w2; Span s2; uint64_t v2; bool ok2;
        if (f == 1) {                                                     // BytesList: the index-th value
            int32_t i = 0;
            while (next_field(q, qend, f2, w2, s2, v2, ok2)) {
                if (f2 == 1 && w2 == 2) { if (i == index) { *ptr = s2.p; *len = s2.n; return SAVP_IO_OK; } ++i; }
            }
            return ok2 ? SAVP_IO_EMISSING : SAVP_IO_ECORRUPT;
        }
        // FloatList / Int64List: packed payload (the only encoding TensorFlow writes) -> pointer + element count
        while (next_field(q, qend, f2, w2, s2, v2, ok2)) {
            if (f2 == 1 && w2 == 2) {
                *ptr = s2.p;
                if (f == 2) { *len = s2.n / 4; return SAVP_IO_OK; }
                uint64_t cnt = 0; const uint8_t* r = s2.p; uint64_t tmp;
                while (r < s2.p + s2.n) { if (!varint(r, s2.p + s2.n, tmp)) return SAVP_IO_ECORRUPT; ++cnt; }
                *len = cnt; return SAVP_IO_OK;
            }
        }
        return ok2 ? SAVP_IO_OK : SAVP_IO_ECORRUPT;                       // empty list
    }
    return ok ? SAVP_IO_EMISSING : SAVP_IO_ECORRUPT;
}

extern "C" int savp_example_int64(const uint8_t* ex, uint64_t ex_len, const char* name, int32_t index, int64_t* out) {
    if (!ex || !name || !out || index < 0) return SAVP_IO_EINVAL;
    int32_t kind; const uint8_t* p; uint64_t n;
    int rc = savp_example_feature(ex, ex_len, name, 0, &kind, &p, &n);
    if (rc) return rc;
    if (kind != 3 || (uint64_t)index >= n) return SAVP_IO_EINVAL;
    const uint8_t* end = p + 10 * n;                                         // a varint is at most 10 bytes; the count was validated above
    uint64_t v = 0;
    for (int32_t i = 0; i <= index; ++i)
        if (!varint(p, end, v)) return SAVP_IO_ECORRUPT;
    *out = (int64_t)v;
    return SAVP_IO_OK;
}

extern "C" int savp_example_floats(const uint8_t* ex, uint64_t ex_len, const char* name, float* out, int64_t n) {
    if (!ex || !name || !out || n < 0) return SAVP_IO_EINVAL;
    Span feat;
    int rc = find_feature(ex, ex_len, name, strlen(name), feat);
    if (rc) return rc;
    const uint8_t *p = feat.p, *end = feat.p + feat.n;
    uint32_t f, w; Span s; uint64_t v; bool ok;
    int64_t got = 0;
    while (next_field(p, end, f, w, s, v, ok)) {
        if (f != 2 || w != 2) continue;                                   // Feature.float_list
        const uint8_t *q = s.p, *qend = s.p + s.n;
        uint32_t f2, w2; Span s2; uint64_t v2; bool ok2;
        while (next_field(q, qend, f2, w2, s2, v2, ok2)) {
            if (f2 != 1) continue;
            if (w2 == 2) {                                                // packed
                const int64_t cnt = (int64_t)(s2.n / 4);
                if (got + cnt > n) return SAVP_IO_EINVAL;
                memcpy(out + got, s2.p, (size_t)cnt * 4); got += cnt;
            } else if (w2 == 5) {                                         // unpacked fixed32
                if (got + 1 > n) return SAVP_IO_EINVAL;
                memcpy(out + got, s2.p, 4); ++got;
            }
        }
        if (!ok2) return SAVP_IO_ECORRUPT;
    }
    if (!ok) return SAVP_IO_ECORRUPT;
    return got == n ? SAVP_IO_OK : SAVP_IO_EINVAL;
}

// ------------------------------------------------------------------------------------------------------------------------
// batched video pipeline: reader thread -> bounded queue of ready batches
// ------------------------------------------------------------------------------------------------------------------------
namespace {
struct Rng {                                                               // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
                      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
};

struct FloatKey { std::string fmt; int dim; int minus; };

struct Batch { std::vector<uint8_t> images; std::vector<std::vector<float>> floats; };
}  // namespace

struct SavpVideoPipeline {
    std::vector<std::string> files;
    std::string image_fmt;
    std::vector<FloatKey> fkeys;
    int example_frames, H, W, C, seq, frame_skip, time_shift, batch, shuffle, shuffle_buffer, num_epochs, prefetch;
    int var_len = 0;                  // 1: one bytes_list feature holds all frames of a sequence + int64 'sequence_length' (KTH)
    uint64_t seed;

    std::thread th;
    std::mutex mu;
    std::condition_variable cv_ready, cv_space;
    std::deque<Batch> ready;
    bool done = false, stop = false;
    int err = 0;
    std::string errmsg;

    void fail(int code, const std::string& msg) {
        std::lock_guard<std::mutex> l(mu);
        err = code; errmsg = msg; done = true;
        cv_ready.notify_all();
    }

    // decode one serialized Example into the slot `b` of the batch under construction; SKIP = filtered out (too short, var_len only)
    enum { SKIP = 1 };
    int decode(const std::vector<uint8_t>& ex, Rng& rng, Batch& out, int b) {
        const int fs1 = frame_skip + 1;
        int frames = example_frames;
        if (var_len) {                                                     // VarLenFeatureVideoDataset.filter / parser (base_dataset.py:401-429)
            int64_t n64 = 0;
            int rc = savp_example_int64(ex.data(), ex.size(), "sequence_length", 0, &n64);
            if (rc) { errmsg = "feature sequence_length missing or not an int64"; return rc; }
            if (n64 < (int64_t)seq) return SKIP;                           // tf.greater_equal(example_sequence_length, sequence_length)
            frames = (int)n64;
        }
        int t_start = 0;
        if (time_shift > 0) {                                              // base_dataset.py:198-211
            const int num_shifts = ((frames - 1) - (seq - 1) * fs1) / time_shift;
            if ((frames - 1) - (seq - 1) * fs1 < 0) { errmsg = "example_sequence_length too short for sequence_length / frame_skip"; return SAVP_IO_EINVAL; }
            t_start = (int)rng.below((uint64_t)num_shifts + 1) * time_shift;
        } else if ((seq - 1) * fs1 + 1 > frames) {
            errmsg = "example_sequence_length too short for sequence_length / frame_skip"; return SAVP_IO_EINVAL;
        }
        const size_t frame = (size_t)H * W * C;
        char name[256];
        for (int t = 0; t < seq; ++t) {                                    // state-like slice (:213)
            const int src_t = t_start + t * fs1;
            if (var_len) snprintf(name, sizeof(name), "%s", image_fmt.c_str());
            else snprintf(name, sizeof(name), image_fmt.c_str(), src_t);
            int32_t kind; const uint8_t* p; uint64_t n;
            int rc = savp_example_feature(ex.data(), ex.size(), name, var_len ? src_t : 0, &kind, &p, &n);
            if (rc) { errmsg = std::string("feature ") + name + (rc == SAVP_IO_EMISSING ? " not found in tfrecord" : " is corrupt"); return rc; }
            if (kind != 1 || n != frame) { errmsg = std::string("feature ") + name + ": expected one raw uint8 image of H*W*C bytes"; return SAVP_IO_EINVAL; }
            memcpy(out.images.data() + ((size_t)b * seq + t) * frame, p, frame);
        }
        for (size_t k = 0; k < fkeys.size(); ++k) {
            const FloatKey& fk = fkeys[k];
            if (fk.minus == 0) {                                           // state-like: one vector per selected frame
                for (int t = 0; t < seq; ++t) {
                    snprintf(name, sizeof(name), fk.fmt.c_str(), t_start + t * fs1);
                    int rc = savp_example_floats(ex.data(), ex.size(), name, out.floats[k].data() + ((size_t)b * seq + t) * fk.dim, fk.dim);
                    if (rc) { errmsg = std::string("float feature ") + name + " missing or of the wrong size"; return rc; }
                }
            } else {                                                       // action-like: all (seq-1)*(fs+1) steps, grouped (:214,223-226)
                const int steps = (seq - 1) * fs1;
                for (int s = 0; s < steps; ++s) {
                    snprintf(name, sizeof(name), fk.fmt.c_str(), t_start + s);
                    int rc = savp_example_floats(ex.data(), ex.size(), name, out.floats[k].data() + ((size_t)b * steps + s) * fk.dim, fk.dim);
                    if (rc) { errmsg = std::string("float feature ") + name + " missing or of the wrong size"; return rc; }
                }
            }
        }
        return SAVP_IO_OK;
    }

    Batch new_batch() const {
        Batch bt;
        bt.images.resize((size_t)batch * seq * H * W * C);
        bt.floats.resize(fkeys.size());
        for (size_t k = 0; k < fkeys.size(); ++k)
            bt.floats[k].resize(fkeys[k].minus == 0 ? (size_t)batch * seq * fkeys[k].dim
                                                    : (size_t)batch * (seq - 1) * (frame_skip + 1) * fkeys[k].dim);
        return bt;
    }

    void run() {
        Rng rng(seed ? seed : 0x5eedull);
        std::vector<std::string> order = files;
        if (shuffle)                                                       // random.shuffle(filenames), base_dataset.py:132-133
            for (size_t i = order.size(); i > 1; --i) std::swap(order[i - 1], order[rng.below(i)]);
        std::vector<std::vector<uint8_t>> pool;                            // shuffle buffer (:137-138)
        const size_t cap = shuffle ? (size_t)(shuffle_buffer > 0 ? shuffle_buffer : 1024) : 1;
        Batch cur = new_batch();
        int filled = 0;
        auto emit = [&](const std::vector<uint8_t>& ex) -> bool {
            std::string msg;
            int rc = decode(ex, rng, cur, filled);
            if (rc == SKIP) return true;
            if (rc) { fail(rc, errmsg); return false; }
            if (++filled == batch) {
                std::unique_lock<std::mutex> l(mu);
                cv_space.wait(l, [&] { return stop || (int)ready.size() < prefetch; });
                if (stop) return false;
                ready.push_back(std::move(cur));
                cv_ready.notify_one();
                l.unlock();
                cur = new_batch(); filled = 0;
            }
            return true;
        };
        for (int epoch = 0; num_epochs <= 0 || epoch < num_epochs; ++epoch) {
            for (const std::string& path : order) {
                SavpTfrFile* f = nullptr;
                if (savp_tfr_open(path.c_str(), 8 << 20, &f)) { fail(SAVP_IO_EIO, "cannot open " + path); return; }
                for (;;) {
                    const uint8_t* d; uint64_t n;
                    int rc = savp_tfr_next(f, &d, &n);
                    if (rc == SAVP_IO_EOF) break;
                    if (rc) { savp_tfr_close(f); fail(rc, "corrupt record in " + path); return; }
                    { std::lock_guard<std::mutex> l(mu); if (stop) { savp_tfr_close(f); return; } }
                    if (pool.size() < cap) { pool.emplace_back(d, d + n); if (pool.size() < cap) continue; }
                    else {
                        // buffer full: emit a random element and put the new record in its place
                        const size_t i = shuffle ? (size_t)rng.below(pool.size()) : 0;
                        std::vector<uint8_t> ex(d, d + n);
                        std::swap(ex, pool[i]);
                        if (!emit(ex)) { savp_tfr_close(f); return; }
                        continue;
                    }
                    if (!shuffle) {                                         // cap == 1: emit in file order
                        std::vector<uint8_t> ex; std::swap(ex, pool[0]); pool.clear();
                        if (!emit(ex)) { savp_tfr_close(f); return; }
                    }
                }
                savp_tfr_close(f);
            }
        }
        while (!pool.empty()) {                                            // drain the shuffle buffer at the end of the last epoch
            const size_t i = shuffle ? (size_t)rng.below(pool.size()) : 0;
            std::vector<uint8_t> ex; std::swap(ex, pool[i]);
            pool[i] = std::move(pool.back()); pool.pop_back();
            if (!emit(ex)) return;
        }
        std::lock_guard<std::mutex> l(mu);                                 // the incomplete last batch is dropped (drop_remainder)
        done = true;
        cv_ready.notify_all();
    }
};

extern "C" int savp_pipeline_create(const SavpVideoPipelineArgs* a, SavpVideoPipeline** out) {
    if (!a || !out || a->num_files < 1 || !a->filenames || !a->image_key_fmt || (a->example_frames < 1 && !a->var_len) || a->height < 1 ||
        a->width < 1 || a->channels < 1 || a->sequence_length < 1 || a->frame_skip < 0 || a->time_shift < 0 || a->batch_size < 1)
        return SAVP_IO_EINVAL;
    if (!a->var_len && (a->sequence_length - 1) * (a->frame_skip + 1) + 1 > a->example_frames) return SAVP_IO_EINVAL;
    SavpVideoPipeline* p = new SavpVideoPipeline();
    for (int i = 0; i < a->num_files; ++i) p->files.emplace_back(a->filenames[i]);
    p->image_fmt = a->image_key_fmt;
    for (int k = 0; k < a->num_float_keys; ++k)
        p->fkeys.push_back(FloatKey{a->float_keys_fmt[k], a->float_dims[k], a->float_per_frame_minus[k]});
    p->example_frames = a->example_frames; p->H = a->height; p->W = a->width; p->C = a->channels;
    p->seq = a->sequence_length; p->frame_skip = a->frame_skip; p->time_shift = a->time_shift; p->batch = a->batch_size;
    p->shuffle = a->shuffle; p->shuffle_buffer = a->shuffle_buffer; p->num_epochs = a->num_epochs; p->seed = a->seed;
    p->prefetch = a->prefetch_batches > 0 ? a->prefetch_batches : 2;
    p->var_len = a->var_len ? 1 : 0;
    p->th = std::thread([p] { p->run(); });
    *out = p;
    return SAVP_IO_OK;
}

extern "C" int savp_pipeline_next(SavpVideoPipeline* p, uint8_t* images, float* const* floats) {
    if (!p || !images) return SAVP_IO_EINVAL;
    Batch bt;
    {
        std::unique_lock<std::mutex> l(p->mu);
        p->cv_ready.wait(l, [&] { return !p->ready.empty() || p->done; });
        if (p->ready.empty()) return p->err ? p->err : SAVP_IO_EOF;
        bt = std::move(p->ready.front());
        p->ready.pop_front();
        p->cv_space.notify_one();
    }
    memcpy(images, bt.images.data(), bt.images.size());
    for (size_t k = 0; k < bt.floats.size(); ++k)
        if (floats && floats[k]) memcpy(floats[k], bt.floats[k].data(), bt.floats[k].size() * sizeof(float));
    return SAVP_IO_OK;
}

extern "C" const char* savp_pipeline_error(SavpVideoPipeline* p) {
    if (!p) return "";
    std::lock_guard<std::mutex> l(p->mu);
    return p->errmsg.c_str();
}

extern "C" void savp_pipeline_destroy(SavpVideoPipeline* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> l(p->mu);
        p->stop = true;
        p->cv_space.notify_all();
    }
    if (p->th.joinable()) p->th.join();
    delete p;
}
