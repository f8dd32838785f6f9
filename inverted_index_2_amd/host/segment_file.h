// segment_file.h — the host mirror's file layer: the reference's `file` package (file/writer.go, file/reader.go)
// and the removed-list persistence (removed_list.go:26-33,73-80) over this repo's own on-disk format.
//
// The reference stores a segment as <key>_fst (vellum FST: term -> offset, or term -> value in direct mode) and
// <key>_val (intcomp-compressed runs).  Neither format can be reproduced here: vellum v1.0.10 and intcomp v1.1.0 are
// third-party modules that are not part of the reference tree and the reference's tests hold no golden bytes
// (SURVEY.md §8 c: byte-level parity unpinned).  What IS pinned — and kept — is the behaviour around the files:
//   * two files per segment, written under *_tmp names and renamed on Close (file/writer.go:61-90);
//   * direct segments (one value per term, Shard.Put) have no value file (file/writer.go:34-40, :95-120);
//   * the key is the creation time in unix nanoseconds (file/writer.go:97);
//   * RemoveSegment unlinks both files (file/writer.go:138-147);
//   * lists round-trip verbatim, unsorted and empty ones included (file/writer_test.go:14-16);
//   * removed.list holds the timestamped batches (shard.go:107-120).
// Formats (little-endian, every file closed by a 64-bit FNV-1a checksum, taken over
// 8-byte words, of all bytes before it):
//   <key>_tdx  "II2TDX1\0" | u64 n_terms | u64 direct (0/1) | u64 term_bytes | u64 term_off[n_terms+1] | bytes
//              | direct only: u32 value[n_terms]
//   <key>_dv1  "II2DV1F\0" | u64 n_lists | u64 n_postings | u64 n_blocks | u64 n_bytes | u32 blk_off[n_lists+1]
//              | {u32 first_doc, u32 byte_off} skip[n_blocks+1] | u8 payload[n_bytes]          (include/ii2.h DV1 arrays)
//   removed.list  "II2RML1\0" | u64 n_batches | per batch: i64 timestamp | u64 n | u32 values[n]
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ii2.h"

namespace ii2h {
namespace file {

namespace fs = std::filesystem;
using Term = std::string;

struct FileError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// FNV-1a over 8-byte little-endian words (the tail byte by byte): the byte-wise form costs a multiply per byte, which
// was a third of a 160 MB segment's write time
inline uint64_t fnv1a(const uint8_t *p, size_t n, uint64_t h = 1469598103934665603ull) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        std::memcpy(&w, p + i, 8);
        h ^= w;
        h *= 1099511628211ull;
    }
    for (; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

class Blob {                         // append-only byte buffer with typed helpers
   public:
    std::vector<uint8_t> b;
    template <class T> void put(const T &v) { const uint8_t *p = (const uint8_t *)&v; b.insert(b.end(), p, p + sizeof(T)); }
    void put_bytes(const void *p, size_t n) { if (n) b.insert(b.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
};

class Cursor {                       // bounds-checked reader over a loaded file
   public:
    Cursor(const std::vector<uint8_t> &buf, const std::string &what) : b_(buf), what_(what) {}
    template <class T> T get() { T v; need(sizeof(T)); std::memcpy(&v, b_.data() + at_, sizeof(T)); at_ += sizeof(T); return v; }
    const uint8_t *bytes(size_t n) { need(n); const uint8_t *p = b_.data() + at_; at_ += n; return p; }
    size_t at() const { return at_; }
    size_t left() const { return b_.size() - at_; }
   private:
    void need(size_t n) const { if (n > b_.size() - at_) throw FileError(what_ + ": truncated file"); }
    const std::vector<uint8_t> &b_;
    std::string what_;
    size_t at_ = 0;
};

inline std::vector<uint8_t> read_whole(const fs::path &p) {
    FILE *f = std::fopen(p.c_str(), "rb");
    if (!f) throw FileError(p.string() + ": " + std::strerror(errno));
    std::vector<uint8_t> buf;
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (sz > 0) {
        buf.resize((size_t)sz);
        if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); throw FileError(p.string() + ": short read"); }
    }
    std::fclose(f);
    return buf;
}

// checksum appended, written under `<name>_tmp`; commit() renames (file/writer.go:75-88: readers never see a half-written file)
inline void write_tmp(const fs::path &final_path, Blob &blob) {
    blob.put<uint64_t>(fnv1a(blob.b.data(), blob.b.size()));
    const fs::path tmp = final_path.string() + "_tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw FileError(tmp.string() + ": " + std::strerror(errno));
    const bool ok = std::fwrite(blob.b.data(), 1, blob.b.size(), f) == blob.b.size();
    if (std::fclose(f) != 0 || !ok) throw FileError(tmp.string() + ": write failed");
}
inline void commit(const fs::path &final_path) {
    std::error_code ec;
    fs::rename(final_path.string() + "_tmp", final_path, ec);
    if (ec) throw FileError(final_path.string() + ": rename: " + ec.message());
}

inline std::vector<uint8_t> read_checked(const fs::path &p, const char magic[8]) {
    std::vector<uint8_t> buf = read_whole(p);
    if (buf.size() < 16 || std::memcmp(buf.data(), magic, 8) != 0) throw FileError(p.string() + ": not a " + std::string(magic, 7) + " file");
    uint64_t want;
    std::memcpy(&want, buf.data() + buf.size() - 8, 8);
    if (fnv1a(buf.data(), buf.size() - 8) != want) throw FileError(p.string() + ": checksum mismatch (corrupt file)");
    buf.resize(buf.size() - 8);
    return buf;
}

inline fs::path tdx_path(const std::string &dir, const std::string &key) { return fs::path(dir) / (key + "_tdx"); }
inline fs::path dv1_path(const std::string &dir, const std::string &key) { return fs::path(dir) / (key + "_dv1"); }

// ---- term dictionary file ----
struct TermFile {
    std::vector<Term> terms;             // ascending (bytes.Compare) — the writer's caller guarantees it, like fst.Insert
    bool direct = false;
    std::vector<uint32_t> direct_vals;   // direct: one value per term (file/writer.go:34-40)
};

inline void write_terms(const std::string &dir, const std::string &key, const TermFile &t) {
    Blob o;
    o.put_bytes("II2TDX1\0", 8);
    o.put<uint64_t>(t.terms.size());
    o.put<uint64_t>(t.direct ? 1 : 0);
    uint64_t total = 0;
    for (auto &s : t.terms) total += s.size();
    o.put<uint64_t>(total);
    uint64_t off = 0;
    o.put<uint64_t>(0);
    for (auto &s : t.terms) { off += s.size(); o.put<uint64_t>(off); }
    for (auto &s : t.terms) o.put_bytes(s.data(), s.size());
    if (t.direct) o.put_bytes(t.direct_vals.data(), t.direct_vals.size() * 4);
    write_tmp(tdx_path(dir, key), o);
}

inline TermFile read_terms(const std::string &dir, const std::string &key) {
    const fs::path p = tdx_path(dir, key);
    const std::vector<uint8_t> buf = read_checked(p, "II2TDX1\0");
    Cursor c(buf, p.string());
    c.bytes(8);
    TermFile t;
    const uint64_t n = c.get<uint64_t>();
    t.direct = c.get<uint64_t>() != 0;
    const uint64_t total = c.get<uint64_t>();
    if (n > c.left() / 8) throw FileError(p.string() + ": term count exceeds the file");
    std::vector<uint64_t> off(n + 1);
    for (auto &o : off) o = c.get<uint64_t>();
    if (off[0] != 0 || off[n] != total) throw FileError(p.string() + ": malformed term offsets");
    for (uint64_t i = 0; i < n; i++) if (off[i + 1] < off[i]) throw FileError(p.string() + ": malformed term offsets");
    const char *bytes = (const char *)c.bytes(total);
    t.terms.reserve(n);
    for (uint64_t i = 0; i < n; i++) t.terms.emplace_back(bytes + off[i], off[i + 1] - off[i]);
    if (t.direct) {
        t.direct_vals.resize(n);
        if (n) std::memcpy(t.direct_vals.data(), c.bytes(n * 4), n * 4);
    }
    if (c.left() != 0) throw FileError(p.string() + ": trailing bytes");
    return t;
}

// ---- DV1 value file ----
struct Dv1File {
    uint64_t n_lists = 0, n_postings = 0;
    std::vector<uint32_t> blk_off;       // n_lists + 1
    std::vector<ii2_skip> skip;          // n_blocks + 1
    std::vector<uint8_t> payload;        // n_bytes
};

inline void write_dv1(const std::string &dir, const std::string &key, const Dv1File &d) {
    Blob o;
    o.put_bytes("II2DV1F\0", 8);
    o.put<uint64_t>(d.n_lists);
    o.put<uint64_t>(d.n_postings);
    o.put<uint64_t>(d.skip.size() - 1);
    o.put<uint64_t>(d.payload.size());
    o.put_bytes(d.blk_off.data(), d.blk_off.size() * 4);
    o.put_bytes(d.skip.data(), d.skip.size() * sizeof(ii2_skip));
    o.put_bytes(d.payload.data(), d.payload.size());
    write_tmp(dv1_path(dir, key), o);
}

inline Dv1File read_dv1(const std::string &dir, const std::string &key) {
    const fs::path p = dv1_path(dir, key);
    const std::vector<uint8_t> buf = read_checked(p, "II2DV1F\0");
    Cursor c(buf, p.string());
    c.bytes(8);
    Dv1File d;
    d.n_lists = c.get<uint64_t>();
    d.n_postings = c.get<uint64_t>();
    const uint64_t nb = c.get<uint64_t>(), nbytes = c.get<uint64_t>();
    if (d.n_lists > c.left() / 4 || nb > c.left() / 8 || nbytes > c.left()) throw FileError(p.string() + ": header exceeds the file");
    d.blk_off.resize(d.n_lists + 1);
    std::memcpy(d.blk_off.data(), c.bytes((d.n_lists + 1) * 4), (d.n_lists + 1) * 4);
    d.skip.resize(nb + 1);
    std::memcpy(d.skip.data(), c.bytes((nb + 1) * sizeof(ii2_skip)), (nb + 1) * sizeof(ii2_skip));
    d.payload.resize(nbytes);
    if (nbytes) std::memcpy(d.payload.data(), c.bytes(nbytes), nbytes);
    if (c.left() != 0) throw FileError(p.string() + ": trailing bytes");
    return d;                            // structure is validated by ii2_seg_import when it goes to the device
}

inline bool has_dv1(const std::string &dir, const std::string &key) { return fs::exists(dv1_path(dir, key)); }

// file/writer.go:138-147
inline void remove_segment(const std::string &dir, const std::string &key) {
    std::error_code e1, e2;
    fs::remove(tdx_path(dir, key), e1);
    fs::remove(dv1_path(dir, key), e2);
    if (e1) throw FileError("remove segment " + key + ": " + e1.message());
    if (e2) throw FileError("remove segment " + key + ": " + e2.message());
}

// ---- removed.list (removed_list.go:26-33, 73-80; shard.go:107-120) ----
using RemovedBatches = std::map<int64_t, std::vector<uint32_t>>;

inline void write_removed(const std::string &dir, const RemovedBatches &lists) {
    Blob o;
    o.put_bytes("II2RML1\0", 8);
    o.put<uint64_t>(lists.size());
    for (auto &kv : lists) {
        o.put<int64_t>(kv.first);
        o.put<uint64_t>(kv.second.size());
        o.put_bytes(kv.second.data(), kv.second.size() * 4);
    }
    const fs::path p = fs::path(dir) / "removed.list";
    write_tmp(p, o);
    commit(p);
}

inline bool read_removed(const std::string &dir, RemovedBatches *out) {       // false: no file (shard.go:338-342)
    const fs::path p = fs::path(dir) / "removed.list";
    if (!fs::exists(p)) return false;
    const std::vector<uint8_t> buf = read_checked(p, "II2RML1\0");
    Cursor c(buf, p.string());
    c.bytes(8);
    const uint64_t n = c.get<uint64_t>();
    for (uint64_t i = 0; i < n; i++) {
        const int64_t ts = c.get<int64_t>();
        const uint64_t m = c.get<uint64_t>();
        if (m > c.left() / 4) throw FileError(p.string() + ": batch exceeds the file");
        std::vector<uint32_t> v(m);
        if (m) std::memcpy(v.data(), c.bytes(m * 4), m * 4);
        (*out)[ts] = std::move(v);
    }
    if (c.left() != 0) throw FileError(p.string() + ": trailing bytes");
    return true;
}

}  // namespace file
}  // namespace ii2h
