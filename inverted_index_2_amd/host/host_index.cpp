// host_index.cpp — host-side mirror of the reference's Shard / InvertedIndex operations above
// the C ABI (include/ii2.h).  The reference is Go; this image has no Go toolchain, so the host
// layer is C++ with the reference's names, argument meaning and error behaviour, and the posting
// work of every operation goes to the GPU through the same entry points a cgo binding would
// use.  Segments live in HBM (DV1); term dictionaries are plain sorted byte strings here (the
// reference's vellum FST, locks and pools are out of scope — SURVEY.md §8).  A shard opened on a
// directory keeps its segments and its removed list on disk as well (segment_file.h: the `file`
// package's Writer / Reader / RemoveSegment and the removed.list persistence).
//
//   Shard.Put / Read / Remove / Merge / MinMax      shard.go:33-298
//   NewShard (load existing files, removed.list)     shard.go:300-358
//   file.Writer / Reader / RemoveSegment             file/writer.go, file/reader.go
//   Segments.add ordering                            segments.go:56-64
//   RemovedLists.Put / Values / Sync                 removed_list.go:36-71
//   InvertedIndex.Put / Read / Merge / PutRemoved / PrefixSearch   inverted_index.go:41-340
//   shardKey                                         shard.go:362-378
//   Intersect(terms)                                 additive
//
// Built as its own library (libii2_host.so) that only sees include/ii2.h and links libii2_hip.so:
// the product library exports the C ABI and nothing else.  A small C facade (ii2h_*) at the bottom
// lets the Python tests replay the reference's test scripts against this layer.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ii2.h"
#include "segment_file.h"

namespace ii2h {

using Term = std::string;      // raw bytes; std::string compares like bytes.Compare (unsigned char order)

struct TermValues {            // file/types.go:9-12
    Term term;
    std::vector<uint32_t> values;
};

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

static void ck(ii2_ctx *ctx, int rc, const char *what) {
    if (rc) throw Error(std::string(what) + ": " + ii2_last_error(ctx));     // fmt.Errorf("…: %w", err)
}

static bool term_less(const Term &a, const Term &b) {
    const size_t m = std::min(a.size(), b.size());
    const int c = m ? std::memcmp(a.data(), b.data(), m) : 0;
    return c ? c < 0 : a.size() < b.size();
}

struct SegHandle {
    ii2_seg *h = nullptr;
    explicit SegHandle(ii2_seg *s) : h(s) {}
    ~SegHandle() { ii2_seg_free(h); }
    SegHandle(const SegHandle &) = delete;
};

struct DevMem {                // device buffer freed at scope exit
    ii2_ctx *ctx;
    void *p = nullptr;
    explicit DevMem(ii2_ctx *c) : ctx(c) {}
    ~DevMem() { if (p) ii2_dev_free(ctx, p); }
    DevMem(const DevMem &) = delete;
};
struct TombHandle {
    ii2_tomb *h = nullptr;
    ~TombHandle() { ii2_tomb_free(h); }
};

struct Segment {               // segments.go:16-24
    int64_t key;               // unix-ns key
    std::vector<Term> terms;   // sorted
    std::shared_ptr<SegHandle> seg;   // terms.size() lists
    bool merging = false;
};

static int64_t now_ns() {      // strictly increasing across threads (segment keys order the tombstone batches)
    static std::mutex mu;
    static int64_t last = 0;
    std::lock_guard<std::mutex> g(mu);
    int64_t t = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    if (t <= last) t = last + 1;
    last = t;
    return t;
}

// ---- file.Writer / file.Reader over segment_file.h (file/writer.go, file/reader.go) -----------------------------
// The encode and decode steps run on the device through the C ABI (ii2_seg_encode / ii2_seg_import + ii2_seg_decode):
// there is no host codec in the product.
class Writer {
   public:
    // file/writer.go:122-136 (NewWriter) and :95-120 (NewDirectWriter)
    Writer(ii2_ctx *ctx, std::string dir, bool direct) : ctx_(ctx), dir_(std::move(dir)), direct_(direct), key_(std::to_string(now_ns())) {}
    // terms must ascend (file/writer.go:30); direct mode keeps Values[0] only (:34-40)
    void Append(const TermValues &tv) {
        if (closed_) throw Error("writer: append after close");
        if (direct_ && tv.values.empty()) throw Error("writer: fst insert: direct mode needs one value per term");
        terms_.push_back(tv.term);
        if (direct_) vals_.push_back(tv.values[0]);
        else vals_.insert(vals_.end(), tv.values.begin(), tv.values.end());
        off_.push_back(vals_.size());
    }
    // file/writer.go:61-90: both files appear under their final names only here
    void Close() {
        if (closed_) return;
        closed_ = true;
        file::TermFile tf;
        tf.terms = std::move(terms_);
        tf.direct = direct_;
        if (direct_) tf.direct_vals = vals_;
        file::write_terms(dir_, key_, tf);
        const uint64_t n_terms = tf.terms.size();
        if (!direct_) {
            ii2_seg *s = nullptr;
            ck(ctx_, ii2_seg_encode(ctx_, n_terms, off_.data(), vals_.data(), II2_HOST, &s), "writer: encode");
            SegHandle seg(s);
            file::write_dv1(dir_, key_, export_dv1(ctx_, s));
            file::commit(file::dv1_path(dir_, key_));
        }
        file::commit(file::tdx_path(dir_, key_));
    }
    const std::string &GetKey() const { return key_; }

    static file::Dv1File export_dv1(ii2_ctx *ctx, const ii2_seg *s) {
        ii2_seg_info info;
        ii2_seg_get_info(s, &info);
        file::Dv1File d;
        d.n_lists = info.n_lists;
        d.n_postings = info.n_postings;
        d.blk_off.resize(info.n_lists + 1);
        d.skip.resize(info.n_blocks + 1);
        d.payload.resize(info.n_bytes);
        ck(ctx, ii2_seg_export(ctx, s, d.blk_off.data(), d.skip.data(), d.payload.empty() ? nullptr : d.payload.data()), "writer: export");
        return d;
    }

   private:
    ii2_ctx *ctx_;
    std::string dir_;
    bool direct_;
    std::string key_;
    bool closed_ = false;
    std::vector<Term> terms_;
    std::vector<uint32_t> vals_;
    std::vector<uint64_t> off_{0};
};

// a segment file pair brought to the device: the term dictionary on the host, the lists as an ii2_seg
struct LoadedSegment {
    std::vector<Term> terms;
    ii2_seg *seg = nullptr;        // caller owns
};
static LoadedSegment load_segment(ii2_ctx *ctx, const std::string &dir, const std::string &key) {
    LoadedSegment out;
    file::TermFile tf = file::read_terms(dir, key);
    const uint64_t n = tf.terms.size();
    if (tf.direct) {
        std::vector<uint64_t> off(n + 1);
        for (uint64_t i = 0; i <= n; i++) off[i] = i;
        ck(ctx, ii2_seg_encode(ctx, n, off.data(), tf.direct_vals.data(), II2_HOST, &out.seg), "reader: encode");
    } else {
        file::Dv1File d = file::read_dv1(dir, key);
        if (d.n_lists != n) throw Error("reader: " + key + ": term file and value file disagree on the term count");
        ck(ctx, ii2_seg_import(ctx, d.n_lists, d.n_postings, d.skip.size() - 1, d.payload.size(), d.blk_off.data(), d.skip.data(),
                               d.payload.empty() ? nullptr : d.payload.data(), II2_HOST, &out.seg), "reader: values file");
    }
    out.terms = std::move(tf.terms);
    return out;
}

class Reader {
   public:
    struct NothingInRange {};      // vellum.ErrIteratorDone from NewReader: the shard skips the segment (shard.go:257-261)
    // file/reader.go:136-199: min / max inclusive, nullptr = open
    Reader(ii2_ctx *ctx, const std::string &dir, const std::string &key, const Term *min, const Term *max) {
        LoadedSegment ls = load_segment(ctx, dir, key);
        SegHandle seg(ls.seg);
        terms_ = std::move(ls.terms);
        at_ = min ? std::lower_bound(terms_.begin(), terms_.end(), *min, term_less) - terms_.begin() : 0;
        end_ = max ? std::upper_bound(terms_.begin(), terms_.end(), *max, term_less) - terms_.begin() : terms_.size();
        if (at_ >= end_) throw NothingInRange{};
        ii2_seg_info info;
        ii2_seg_get_info(seg.h, &info);
        off_.resize(terms_.size() + 1);
        vals_.resize(info.n_postings);
        ck(ctx, ii2_seg_decode(ctx, seg.h, off_.data(), vals_.empty() ? nullptr : vals_.data(), II2_HOST), "reader: values file: decompress");
    }
    // false = go_iterators.EmptyIterator
    bool Next(TermValues *tv) {
        if (closed_ || at_ >= end_) return false;
        tv->term = terms_[at_];
        tv->values.assign(vals_.begin() + off_[at_], vals_.begin() + off_[at_ + 1]);
        at_++;
        return true;
    }
    void Close() { closed_ = true; }

   private:
    std::vector<Term> terms_;
    std::vector<uint64_t> off_;
    std::vector<uint32_t> vals_;
    size_t at_ = 0, end_ = 0;
    bool closed_ = false;
};

class Shard {
   public:
    // basedir empty: segments live in HBM only (the replayed reference scripts); else shard.go:300-358 — load every
    // segment file pair and removed.list found there, and keep the directory in step with every later operation
    explicit Shard(ii2_ctx *ctx, std::string basedir = "") : ctx_(ctx), basedir_(std::move(basedir)) {
        if (basedir_.empty()) return;
        std::error_code ec;
        if (!file::fs::is_directory(basedir_, ec)) throw Error("load inverted index shard: " + basedir_ + " is not a directory");
        std::vector<std::string> keys;
        for (auto &e : file::fs::directory_iterator(basedir_)) {
            if (e.is_directory()) continue;
            const std::string name = e.path().filename().string();
            if (name.size() <= 4 || name.compare(name.size() - 4, 4, "_tdx") != 0) continue;      // (*_tmp files are not segments)
            keys.push_back(name.substr(0, name.size() - 4));
        }
        std::sort(keys.begin(), keys.end());
        for (auto &key : keys) {
            char *endp = nullptr;
            errno = 0;
            const long long k = std::strtoll(key.c_str(), &endp, 10);
            if (errno || !endp || *endp || key.empty()) throw Error("load inverted index: key to int conversion: " + key);
            LoadedSegment ls = load_segment(ctx_, basedir_, key);
            add(Segment{(int64_t)k, std::move(ls.terms), std::make_shared<SegHandle>(ls.seg)});
        }
        try { file::read_removed(basedir_, &removed_); }
        catch (const std::exception &e) { throw Error(std::string("rem list: ") + e.what()); }
    }
    const std::string &basedir() const { return basedir_; }
    void Close() {}                // shard.go:247-249

    // shard.go:33-67 — one direct segment, every term -> [val]
    void Put(std::vector<Term> terms, uint32_t val) {
        std::sort(terms.begin(), terms.end(), term_less);
        terms.erase(std::unique(terms.begin(), terms.end()), terms.end());
        std::vector<uint64_t> off(terms.size() + 1);
        for (size_t i = 0; i <= terms.size(); i++) off[i] = i;
        std::vector<uint32_t> vals(terms.size(), val);
        ii2_seg *s = nullptr;
        ck(ctx_, ii2_seg_encode(ctx_, terms.size(), off.data(), vals.data(), II2_HOST, &s), "s: put");
        Segment seg{now_ns(), std::move(terms), std::make_shared<SegHandle>(s)};
        if (!basedir_.empty()) {   // file.NewDirectWriter: term file only, the value rides in it (file/writer.go:95-120)
            file::TermFile tf;
            tf.terms = seg.terms;
            tf.direct = true;
            tf.direct_vals = vals;
            const std::string key = std::to_string(seg.key);
            try { file::write_terms(basedir_, key, tf); file::commit(file::tdx_path(basedir_, key)); }
            catch (const std::exception &e) { throw Error(std::string("index put: ") + e.what()); }
        }
        std::lock_guard<std::mutex> g(mu_);
        add(std::move(seg));       // make the new segment visible (shard.go:64)
    }

    // shard.go:72-75 + makeIterator :253-278 — merged view of all segments, [min,max] inclusive, no tombstones
    // The snapshot of shared pointers plays the part of the reference's per-segment read locks (segments.go:32-46):
    // a merge may detach and unlink these segments meanwhile, their device arrays live until the last reader lets go.
    std::vector<TermValues> Read(const Term *min, const Term *max) const {
        const std::vector<std::shared_ptr<Segment>> snap = snapshot();
        std::vector<const Segment *> segs;
        for (auto &s : snap) segs.push_back(s.get());
        return merged(segs, min, max, nullptr);
    }

    // shard.go:78-105
    void Remove(const std::vector<uint32_t> &values) {
        if (values.empty()) return;
        std::lock_guard<std::mutex> g(mu_);
        std::vector<int64_t> ts{now_ns()};
        for (auto &s : segments_) ts.push_back(s->key);
        removed_sync(ts);
        removed_[now_ns()] = values;
        WriteRemovedList();
    }

    // shard.go:107-120 (called with mu_ held)
    void WriteRemovedList() const {
        if (basedir_.empty()) return;
        try { file::write_removed(basedir_, removed_); }
        catch (const std::exception &e) { throw Error(std::string("write rem list: ") + e.what()); }
    }

    // removed_list.go:44-54
    std::vector<uint32_t> RemovedValues() const {
        std::lock_guard<std::mutex> g(mu_);
        std::vector<uint32_t> r;
        for (auto &kv : removed_) r.insert(r.end(), kv.second.begin(), kv.second.end());
        std::sort(r.begin(), r.end());
        return r;
    }

    // shard.go:127-245.  `worker`: the context (= HIP stream) of the calling worker thread; segments are shared
    // between contexts of one device, so InvertedIndex.Merge's workers each bring their own (inverted_index.go:83-103)
    int Merge(int reqCount, int mCount, ii2_ctx *worker = nullptr) {
        ii2_ctx *ctx = worker ? worker : ctx_;
        std::vector<std::shared_ptr<Segment>> picked;
        {   // pick under the list lock; the `merging` flag keeps concurrent merges off the same segments (shard.go:134-146)
            std::lock_guard<std::mutex> g(mu_);
            if ((int)segments_.size() < reqCount) return 0;
            for (auto &s : segments_) {
                if ((int)picked.size() == mCount) break;
                if (!s->merging) { s->merging = true; picked.push_back(s); }
            }
        }
        if (picked.size() < 2) return 0;           // NB: a lone picked flag stays set (shard.go:149-151)
        std::vector<const Segment *> segs;
        for (auto &s : picked) segs.push_back(s.get());
        const std::vector<uint32_t> removed = RemovedValues();
        Segment out;
        const bool any = merged_segment(ctx, segs, removed, &out);
        if (any && !basedir_.empty()) {            // file.NewWriter + Close: both files, renamed when complete
            const std::string key = std::to_string(out.key);
            try {
                file::TermFile tf;
                tf.terms = out.terms;
                file::write_terms(basedir_, key, tf);
                file::write_dv1(basedir_, key, Writer::export_dv1(ctx, out.seg->h));
                file::commit(file::dv1_path(basedir_, key));
                file::commit(file::tdx_path(basedir_, key));
            } catch (const std::exception &e) { throw Error(std::string("s: merge: writer close: ") + e.what()); }
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            if (any) add(std::move(out));          // lazy writer: nothing survives -> no segment (shard.go:219-225)
            segments_.erase(std::remove_if(segments_.begin(), segments_.end(),
                                           [&](const std::shared_ptr<Segment> &s) {
                                               return std::find(picked.begin(), picked.end(), s) != picked.end();
                                           }),
                            segments_.end());      // Segments.detach: invisible for new reads (shard.go:228-230)
        }
        if (!basedir_.empty()) {                   // file.RemoveSegment for every merged segment; the last error is reported (shard.go:232-242)
            std::string err;
            for (auto &sg : picked) {
                try { file::remove_segment(basedir_, std::to_string(sg->key)); }
                catch (const std::exception &e) { err = e.what(); }
            }
            if (!err.empty()) throw Error(err);
        }
        return (int)picked.size();
    }

    // shard.go:280-298
    bool MinMax(Term *mn, Term *mx) const {
        std::lock_guard<std::mutex> g(mu_);
        bool any = false;
        for (auto &s : segments_) {
            if (s->terms.empty()) continue;
            if (!any || term_less(s->terms.front(), *mn)) *mn = s->terms.front();
            if (!any || term_less(*mx, s->terms.back())) *mx = s->terms.back();
            any = true;
        }
        return any;
    }

    size_t SegmentCount() const { std::lock_guard<std::mutex> g(mu_); return segments_.size(); }

   private:
    std::vector<std::shared_ptr<Segment>> snapshot() const { std::lock_guard<std::mutex> g(mu_); return segments_; }

    // segments.go:56-64 — insert before the first segment with terms >= new.terms (mu_ held, or the constructor)
    void add(Segment s) {
        auto sp = std::make_shared<Segment>(std::move(s));
        size_t pos = 0;
        while (pos < segments_.size() && segments_[pos]->terms.size() < sp->terms.size()) pos++;
        segments_.insert(segments_.begin() + pos, sp);
    }

    void removed_sync(const std::vector<int64_t> &ts) {   // removed_list.go:57-71
        if (ts.empty()) return;
        const int64_t oldest = *std::min_element(ts.begin(), ts.end());
        for (auto it = removed_.begin(); it != removed_.end();) it = it->first < oldest ? removed_.erase(it) : ++it;
    }

    // Term alignment (host; file.CompareTermValues order): union of the segments' terms inside [min,max]
    // and, per segment, an aligned view with one slot per union term.
    struct Aligned {
        std::vector<Term> terms;
        std::vector<std::shared_ptr<SegHandle>> views;
    };
    Aligned align(ii2_ctx *ctx, const std::vector<const Segment *> &segs, const Term *min, const Term *max) const {
        // every segment's dictionary restricted to [min, max] is a run of consecutive terms (the dictionaries are sorted);
        // the k-way dictionary merge itself runs on the device (ii2_align_terms) and so does the construction of the
        // aligned views (ii2_seg_select_aligned): the host only flattens the byte strings
        Aligned a;
        struct Part { const Segment *seg; size_t j0, j1; };
        std::vector<Part> parts;
        for (auto *s : segs) {
            size_t j0 = 0, j1 = s->terms.size();
            if (min) j0 = std::lower_bound(s->terms.begin(), s->terms.end(), *min, term_less) - s->terms.begin();
            if (max) j1 = std::upper_bound(s->terms.begin(), s->terms.end(), *max, term_less) - s->terms.begin();
            if (j0 >= j1) continue;                // segment has nothing in range: skipped (shard.go:257-261)
            parts.push_back(Part{s, j0, j1});
        }
        if (parts.empty()) return a;
        std::vector<std::shared_ptr<SegHandle>> views;
        // the library aligns at most II2_MAX_LISTS dictionaries per call: more segments are aligned in groups against the
        // union of the previous groups (kept as an extra dictionary)... the host mirror keeps it simple and aligns in one
        // call when it can, else falls back to groups of II2_MAX_LISTS merged pairwise by fold_to_limit's rounds
        const size_t kmax = II2_MAX_LISTS;
        if (parts.size() <= kmax) {
            std::string bytes;
            std::vector<uint64_t> off{0}, first{0};
            for (auto &pt : parts) {
                for (size_t j = pt.j0; j < pt.j1; j++) { bytes += pt.seg->terms[j]; off.push_back(bytes.size()); }
                first.push_back(off.size() - 1);
            }
            ii2_align *al = nullptr;
            ck(ctx, ii2_align_terms(ctx, (uint32_t)parts.size(), (const uint8_t *)bytes.data(), off.data(), first.data(), &al), "index read");
            struct AlignGuard { ii2_align *h; ~AlignGuard() { ii2_align_free(h); } } guard{al};
            uint64_t nu = 0;
            ii2_align_info(al, &nu, nullptr);
            std::vector<uint64_t> rep(nu);
            ck(ctx, ii2_align_export(ctx, al, rep.data(), nullptr), "index read");
            a.terms.reserve(nu);
            for (uint64_t u = 0; u < nu; u++) a.terms.emplace_back(bytes.data() + off[rep[u]], off[rep[u] + 1] - off[rep[u]]);
            std::vector<const ii2_seg *> srcs;
            std::vector<uint64_t> fl;
            for (auto &pt : parts) { srcs.push_back(pt.seg->seg->h); fl.push_back(pt.j0); }
            std::vector<ii2_seg *> vs(parts.size(), nullptr);
            ck(ctx, ii2_seg_select_aligned_all(ctx, srcs.data(), al, fl.data(), vs.data()), "index read");     // (one wait for all the views)
            for (ii2_seg *v : vs) a.views.push_back(std::make_shared<SegHandle>(v));
            return a;
        }
        // more than II2_MAX_LISTS segments: union dictionary on the host, views through ii2_seg_select
        for (auto &pt : parts) a.terms.insert(a.terms.end(), pt.seg->terms.begin() + pt.j0, pt.seg->terms.begin() + pt.j1);
        std::sort(a.terms.begin(), a.terms.end(), term_less);
        a.terms.erase(std::unique(a.terms.begin(), a.terms.end()), a.terms.end());
        for (auto &pt : parts) {
            std::vector<int64_t> src(a.terms.size(), -1);
            size_t j = pt.j0;
            for (size_t i = 0; i < a.terms.size(); i++) {
                while (j < pt.j1 && term_less(pt.seg->terms[j], a.terms[i])) j++;
                if (j < pt.j1 && pt.seg->terms[j] == a.terms[i]) src[i] = (int64_t)j;
            }
            ii2_seg *v = nullptr;
            ck(ctx, ii2_seg_select(ctx, pt.seg->seg->h, a.terms.size(), src.data(), &v), "index read");
            a.views.push_back(std::make_shared<SegHandle>(v));
        }
        return a;
    }

    // The library merges at most II2_MAX_LISTS segments per call; the reference has no such bound (mCount is the
    // caller's), so more are folded in rounds: the union is associative, and the tombstone filter and the empty-term
    // drop only have to happen in the last round.
    void fold_to_limit(ii2_ctx *ctx, std::vector<std::shared_ptr<SegHandle>> &views) const {
        while (views.size() > II2_MAX_LISTS) {
            std::vector<const ii2_seg *> hs;
            for (size_t i = 0; i < II2_MAX_LISTS; i++) hs.push_back(views[i]->h);
            ii2_seg *m = nullptr;
            ii2_merge_stats st;
            std::memset(&st, 0, sizeof st);
            ck(ctx, ii2_merge_segments_to_seg(ctx, II2_MAX_LISTS, hs.data(), nullptr, &m, &st), "s: merge");
            views.erase(views.begin(), views.begin() + II2_MAX_LISTS);
            if (m) views.insert(views.begin(), std::make_shared<SegHandle>(m));
        }
    }

    // The small read in one launch (ii2_read_small): a few small segments, the slice of each one's terms that lies in
    // [min, max].  false: over the limits, take the general path.
    bool read_small(const std::vector<const Segment *> &segs, const Term *min, const Term *max, std::vector<TermValues> *out) const {
        if (segs.empty() || segs.size() > II2_MAX_LISTS) return false;
        std::string bytes;
        std::vector<uint64_t> off{0}, first{0}, lfirst;
        std::vector<const ii2_seg *> hs;
        uint64_t n_post = 0;
        for (auto *sg : segs) {
            size_t j0 = 0, j1 = sg->terms.size();
            if (min) j0 = std::lower_bound(sg->terms.begin(), sg->terms.end(), *min, term_less) - sg->terms.begin();
            if (max) j1 = std::upper_bound(sg->terms.begin(), sg->terms.end(), *max, term_less) - sg->terms.begin();
            if (j0 >= j1) continue;                // segment has nothing in range: skipped (shard.go:257-261)
            ii2_seg_info info;
            ii2_seg_get_info(sg->seg->h, &info);
            n_post += info.n_postings;
            if (off.size() - 1 + (j1 - j0) > II2_SMALL_MERGE_TERMS || n_post > II2_SMALL_MERGE_POSTINGS) return false;
            for (size_t j = j0; j < j1; j++) { bytes += sg->terms[j]; off.push_back(bytes.size()); }
            if (bytes.size() > 16384) return false;
            first.push_back(off.size() - 1);
            lfirst.push_back(j0);
            hs.push_back(sg->seg->h);
        }
        if (hs.empty()) return true;                // nothing in range: an empty read
        const uint64_t n = off.size() - 1;
        std::vector<uint64_t> rep(n), po(n + 1);
        std::vector<uint32_t> vals(n_post + 1);
        uint64_t nu = 0;
        const int rc = ii2_read_small(ctx_, (uint32_t)hs.size(), hs.data(), (const uint8_t *)bytes.data(), off.data(), first.data(), lfirst.data(),
                                      rep.data(), po.data(), vals.data(), n_post, &nu);
        if (rc == II2_ERANGE) return false;
        ck(ctx_, rc, "index read");
        out->reserve(nu);
        for (uint64_t u = 0; u < nu; u++)
            out->push_back(TermValues{Term(bytes.data() + off[rep[u]], off[rep[u] + 1] - off[rep[u]]),
                                      std::vector<uint32_t>(vals.begin() + po[u], vals.begin() + po[u + 1])});
        return true;
    }

    std::vector<TermValues> merged(const std::vector<const Segment *> &segs, const Term *min, const Term *max,
                                   const std::vector<uint32_t> *removed) const {
        std::vector<TermValues> out;
        if (!removed && read_small(segs, min, max, &out)) return out;
        Aligned a = align(ctx_, segs, min, max);
        if (a.views.empty()) return out;
        fold_to_limit(ctx_, a.views);
        TombHandle tomb;
        if (removed && !removed->empty()) ck(ctx_, ii2_tomb_create(ctx_, removed->data(), removed->size(), II2_HOST, &tomb.h), "s: merge");
        std::vector<const ii2_seg *> hs;
        uint64_t cap = 0;
        for (auto &v : a.views) {
            hs.push_back(v->h);
            ii2_seg_info info;
            ii2_seg_get_info(v->h, &info);
            cap += info.n_postings;
        }
        const uint64_t T = a.terms.size();
        DevMem d_off(ctx_), d_vals(ctx_);
        ck(ctx_, ii2_dev_alloc(ctx_, (T + 1) * 8, &d_off.p), "s: merge");
        ck(ctx_, ii2_dev_alloc(ctx_, (cap + 1) * 4, &d_vals.p), "s: merge");
        ii2_merge_stats st;
        std::memset(&st, 0, sizeof st);
        ck(ctx_, ii2_merge_segments(ctx_, (uint32_t)hs.size(), hs.data(), tomb.h, (uint64_t *)d_off.p, (uint32_t *)d_vals.p, cap + 1, &st), "s: merge");
        std::vector<uint64_t> off(T + 1);
        std::vector<uint32_t> vals(st.n_out);
        ck(ctx_, ii2_copy_d2h(ctx_, off.data(), d_off.p, (T + 1) * 8), "s: merge");
        if (st.n_out) ck(ctx_, ii2_copy_d2h(ctx_, vals.data(), d_vals.p, st.n_out * 4), "s: merge");
        for (uint64_t t = 0; t < T; t++) {
            TermValues tv{a.terms[t], std::vector<uint32_t>(vals.begin() + off[t], vals.begin() + off[t + 1])};
            if (removed && tv.values.empty()) continue;      // merge drops emptied terms (shard.go:192-194)
            out.push_back(std::move(tv));
        }
        return out;
    }

    // Merge into a new device-resident segment; false when no term survives.
    bool merged_segment(ii2_ctx *ctx, const std::vector<const Segment *> &segs, const std::vector<uint32_t> &removed, Segment *out) const {
        // The common case — a few small segments (every Put writes a direct segment of one posting per term, shard.go:33-67,
        // and the smallest segments are picked first, shard.go:135-146) — is one launch: alignment, union, removed-list
        // filter, empty-term drop and encode in ii2_merge_small.  Anything larger takes the general path below.
        {
            size_t n_terms = 0, n_bytes = 0;
            uint64_t n_post = 0;
            for (auto *sg : segs) {
                n_terms += sg->terms.size();
                for (auto &t : sg->terms) n_bytes += t.size();
                ii2_seg_info inf;
                if (ii2_seg_get_info(sg->seg->h, &inf) == II2_OK) n_post += inf.n_postings;
            }
            if (!segs.empty() && segs.size() <= II2_MAX_LISTS && n_terms <= II2_SMALL_MERGE_TERMS && n_post <= II2_SMALL_MERGE_POSTINGS &&
                removed.size() <= II2_SMALL_MERGE_REMOVED && n_bytes <= 16384) {
                std::string bytes;
                std::vector<uint64_t> off{0}, first{0};
                std::vector<const ii2_seg *> hs;
                for (auto *sg : segs) {
                    for (auto &t : sg->terms) { bytes += t; off.push_back(bytes.size()); }
                    first.push_back(off.size() - 1);
                    hs.push_back(sg->seg->h);
                }
                std::vector<uint64_t> kept(n_terms + 1);
                uint64_t n_kept = 0;
                ii2_seg *m = nullptr;
                ii2_merge_stats st;
                std::memset(&st, 0, sizeof st);
                const int rc = ii2_merge_small(ctx, (uint32_t)hs.size(), hs.data(), (const uint8_t *)bytes.data(), off.data(), first.data(),
                                               removed.empty() ? nullptr : removed.data(), removed.size(), &m, kept.data(), &n_kept, &st);
                if (rc == II2_OK) {
                    if (!m) return false;
                    out->terms.clear();
                    for (uint64_t j = 0; j < n_kept; j++) out->terms.emplace_back(bytes.data() + off[kept[j]], off[kept[j] + 1] - off[kept[j]]);
                    out->seg = std::make_shared<SegHandle>(m);
                    out->key = now_ns();
                    return true;
                }
                if (rc != II2_ERANGE) ck(ctx, rc, "s: merge");      // (a view's posting count is an upper bound: the kernel may still refuse)
            }
        }
        Aligned a = align(ctx, segs, nullptr, nullptr);
        if (a.views.empty()) return false;
        fold_to_limit(ctx, a.views);
        TombHandle tomb;
        if (!removed.empty()) ck(ctx, ii2_tomb_create(ctx, removed.data(), removed.size(), II2_HOST, &tomb.h), "s: merge");
        std::vector<const ii2_seg *> hs;
        for (auto &v : a.views) hs.push_back(v->h);
        ii2_seg *m = nullptr;
        ii2_merge_stats st;
        std::memset(&st, 0, sizeof st);
        ck(ctx, ii2_merge_segments_to_seg(ctx, (uint32_t)hs.size(), hs.data(), tomb.h, &m, &st), "s: merge");
        if (!m) return false;
        SegHandle full(m);
        // drop the terms that lost every posting: compacted view over the merged segment
        const uint64_t T = a.terms.size();
        std::vector<uint64_t> off(T + 1);
        ck(ctx, ii2_seg_decode(ctx, m, off.data(), nullptr, II2_HOST), "s: merge");
        std::vector<int64_t> src;
        for (uint64_t t = 0; t < T; t++)
            if (off[t + 1] > off[t]) { src.push_back((int64_t)t); out->terms.push_back(a.terms[t]); }
        ii2_seg *c = nullptr;
        ck(ctx, ii2_seg_select(ctx, m, src.size(), src.data(), &c), "s: merge");
        out->seg = std::make_shared<SegHandle>(c);
        out->key = now_ns();
        return true;
    }

    ii2_ctx *ctx_;
    std::string basedir_;                                  // empty: no files
    mutable std::mutex mu_;                                // Segments.m and RemovedLists.m in one: guards the two members below
    std::vector<std::shared_ptr<Segment>> segments_;       // sorted by term count
    std::map<int64_t, std::vector<uint32_t>> removed_;     // RemovedLists.lists
};

// shard.go:362-378
static uint32_t shard_key(const Term &t) {
    uint8_t a = 0, b = 0;
    if (t.size() >= 2) { a = (uint8_t)t[0]; b = (uint8_t)t[1]; }
    return (uint32_t)((uint16_t)((uint16_t)(a << 8) + b) >> 6);
}

class InvertedIndex {
   public:
    // inverted_index.go:342-403: every sub-directory of basedir is a shard named by its key ("%04d")
    explicit InvertedIndex(ii2_ctx *ctx, std::string basedir = "") : ctx_(ctx), basedir_(std::move(basedir)) {
        if (basedir_.empty()) return;
        std::error_code ec;
        if (!file::fs::is_directory(basedir_, ec)) throw Error("shards read: " + basedir_ + " is not a directory");
        for (auto &e : file::fs::directory_iterator(basedir_)) {
            if (!e.is_directory()) continue;
            const std::string name = e.path().filename().string();
            char *endp = nullptr;
            const unsigned long k = std::strtoul(name.c_str(), &endp, 10);
            if (name.empty() || !endp || *endp || k >= 1024) continue;          // not a shard directory
            try { shards_.emplace((uint32_t)k, std::make_unique<Shard>(ctx_, e.path().string())); }
            catch (const std::exception &ex) { throw Error(std::string("shard init: ") + ex.what()); }
        }
    }

    void Put(const std::vector<Term> &terms, uint32_t val) {                  // inverted_index.go:113-145
        std::map<uint32_t, std::vector<Term>> groups;
        for (auto &t : terms) groups[shard_key(t)].push_back(t);
        for (auto &g : groups) shard(g.first).Put(g.second, val);
    }
    void PutRemoved(const std::vector<uint32_t> &values) {                    // inverted_index.go:41-55
        for (auto &s : shard_list()) s.second->Remove(values);
    }
    // inverted_index.go:62-109: `concurrency` workers pull shards off one queue; every worker owns a context (a HIP
    // stream with its scratch) on the index's device, the segments are shared.  A worker that fails records the
    // error and stops pulling, the others drain the queue (the reference's goroutines do the same); concurrency <= 0
    // starts no worker and merges nothing.
    int64_t Merge(int reqCount, int mCount, int concurrency) {
        std::vector<Shard *> shards;
        for (auto &s : shard_list()) shards.push_back(s.second);
        if (concurrency <= 0 || shards.empty()) return 0;
        const size_t nw = std::min<size_t>((size_t)concurrency, shards.size());
        std::vector<ii2_ctx *> workers;                          // worker 0 uses the index's own context
        {
            std::lock_guard<std::mutex> g(mu_);
            while (workers_.size() + 1 < nw) {
                ii2_ctx *c = nullptr;
                if (ii2_ctx_create(ii2_ctx_device(ctx_), 0, &c)) throw Error(std::string("merge: worker context: ") + ii2_last_error(nullptr));
                workers_.push_back(c);
            }
            workers.assign(workers_.begin(), workers_.begin() + (nw - 1));
        }
        std::atomic<size_t> next{0};
        std::atomic<int64_t> merged{0};
        std::mutex err_mu;
        std::string err;
        auto work = [&](ii2_ctx *c) {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= shards.size()) return;
                try { merged += shards[i]->Merge(reqCount, mCount, c); }
                catch (const std::exception &e) { std::lock_guard<std::mutex> g(err_mu); err = e.what(); return; }
            }
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < nw; w++) pool.emplace_back(work, workers[w - 1]);
        work(ctx_);
        for (auto &t : pool) t.join();
        if (!err.empty()) throw Error(err);
        return merged.load();
    }
    ~InvertedIndex() { for (ii2_ctx *c : workers_) ii2_ctx_destroy(c); }
    std::vector<TermValues> Read(const Term *min, const Term *max) const {    // inverted_index.go:300-340
        std::vector<TermValues> out;
        for (auto &s : shard_list()) {                                        // ascending shard key
            Term mn, mx;
            if (!s.second->MinMax(&mn, &mx)) continue;
            if (min && term_less(mx, *min)) continue;
            if (max && term_less(*max, mn)) continue;
            auto part = s.second->Read(min, max);
            out.insert(out.end(), std::make_move_iterator(part.begin()), std::make_move_iterator(part.end()));
        }
        return out;
    }
    std::map<Term, std::vector<uint32_t>> PrefixSearch(std::vector<Term> prefixes) const {   // inverted_index.go:192-295
        std::sort(prefixes.begin(), prefixes.end(), term_less);
        std::map<Term, std::vector<std::vector<uint32_t>>> found;
        for (auto &s : shard_list()) {
            Term mn, mx;
            if (!s.second->MinMax(&mn, &mx)) continue;
            std::vector<Term> mine;
            for (auto &p : prefixes) {
                size_t l = std::min(p.size(), mn.size());
                if (p.compare(0, l, mn, 0, l) < 0) continue;
                l = std::min(p.size(), mx.size());
                if (p.compare(0, l, mx, 0, l) > 0) continue;
                mine.push_back(p);
            }
            if (mine.empty()) continue;
            const Term &greatest = mine.back();
            for (auto &tv : s.second->Read(&mine.front(), nullptr)) {
                const Term tp = tv.term.substr(0, std::min(tv.term.size(), greatest.size()));
                if (term_less(greatest, tp)) break;
                for (auto &p : mine)
                    if (tv.term.compare(0, p.size(), p) == 0 && tv.term.size() >= p.size()) found[p].push_back(tv.values);
            }
        }
        std::map<Term, std::vector<uint32_t>> out;
        for (auto &f : found) out[f.first] = lists_op(true, f.second);        // :288-292 sort + compact, on the GPU
        return out;
    }
    // additive: ids present under every term
    std::vector<uint32_t> Intersect(const std::vector<Term> &terms) const {
        std::vector<std::vector<uint32_t>> lists;
        for (auto &t : terms) {
            std::vector<uint32_t> v;
            if (Shard *sh = find_shard(shard_key(t)))
                for (auto &tv : sh->Read(&t, &t)) v = tv.values;
            lists.push_back(std::move(v));
        }
        return lists_op(false, lists);
    }
    size_t ShardCount() const { std::lock_guard<std::mutex> g(mu_); return shards_.size(); }
    Shard *OnlyShard() { std::lock_guard<std::mutex> g(mu_); return shards_.empty() ? nullptr : shards_.begin()->second.get(); }

   private:
    // shards are never removed: a snapshot of (key, pointer) pairs in key order is all a reader needs (ii.shardsM)
    std::vector<std::pair<uint32_t, Shard *>> shard_list() const {
        std::lock_guard<std::mutex> g(mu_);
        std::vector<std::pair<uint32_t, Shard *>> v;
        for (auto &s : shards_) v.emplace_back(s.first, s.second.get());
        return v;
    }
    Shard *find_shard(uint32_t key) const {
        std::lock_guard<std::mutex> g(mu_);
        auto it = shards_.find(key);
        return it == shards_.end() ? nullptr : it->second.get();
    }
    Shard &shard(uint32_t key) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = shards_.find(key);
        if (it == shards_.end()) {                 // inverted_index.go:163-190 newShard: mkdir <basedir>/<key>
            std::string dir;
            if (!basedir_.empty()) {
                char name[8];
                std::snprintf(name, sizeof name, "%04u", key);
                dir = (file::fs::path(basedir_) / name).string();
                std::error_code ec;
                if (!file::fs::create_directory(dir, ec) && ec) throw Error("new shard: " + dir + ": " + ec.message());
            }
            it = shards_.emplace(key, std::make_unique<Shard>(ctx_, dir)).first;
        }
        return *it->second;
    }
    std::vector<uint32_t> lists_op(bool is_union, const std::vector<std::vector<uint32_t>> &lists) const {
        std::vector<uint64_t> off(lists.size() + 1, 0);
        std::vector<uint32_t> flat;
        for (size_t i = 0; i < lists.size(); i++) {
            flat.insert(flat.end(), lists[i].begin(), lists[i].end());
            off[i + 1] = flat.size();
        }
        std::vector<uint32_t> out(flat.size() + 1);
        uint64_t n = 0;
        if (lists.empty()) return {};
        int rc = is_union ? ii2_union_host(ctx_, (uint32_t)lists.size(), off.data(), flat.data(), nullptr, 0, out.data(), out.size(), &n)
                          : ii2_intersect_host(ctx_, (uint32_t)lists.size(), off.data(), flat.data(), nullptr, 0, out.data(), out.size(), &n);
        ck(ctx_, rc, is_union ? "prefix search" : "intersect");
        out.resize(n);
        return out;
    }
    ii2_ctx *ctx_;
    std::string basedir_;
    mutable std::mutex mu_;                                // guards shards_ (the map, not the shards) and workers_
    std::vector<ii2_ctx *> workers_;                       // contexts of Merge's workers 1..n-1 (created on demand)
    std::map<uint32_t, std::unique_ptr<Shard>> shards_;    // sorted by key, like ii.shards
};

}  // namespace ii2h

// ---- C facade for the Python tests ----------------------------------------------------------
using namespace ii2h;

struct ii2h_target {
    std::shared_ptr<Shard> shard;          // shared by every session attached to the same shard / index:
    std::shared_ptr<InvertedIndex> index;  // the error text and the last results below are per session
    std::string err;
    std::vector<TermValues> result;     // last Read / PrefixSearch result
    std::vector<uint32_t> ids;          // last Intersect / RemovedValues result
};

static std::vector<Term> unpack_terms(const uint8_t *bytes, const uint64_t *off, uint64_t n) {
    std::vector<Term> t(n);
    for (uint64_t i = 0; i < n; i++) t[i].assign((const char *)bytes + off[i], off[i + 1] - off[i]);
    return t;
}

#define H_TRY(t, ...)                                   \
    try { __VA_ARGS__; return 0; }                      \
    catch (const std::exception &e) { (t)->err = e.what(); return -1; }

extern "C" {

ii2h_target *ii2h_create(ii2_ctx *ctx, int is_index) {
    auto *t = new ii2h_target();
    if (is_index) t->index = std::make_shared<InvertedIndex>(ctx);
    else t->shard = std::make_shared<Shard>(ctx);
    return t;
}
// another handle on the same Shard / InvertedIndex with its own result and error slots: one per calling thread
// (the operations themselves are thread-safe like the reference's; a handle's result buffers are not shared state)
ii2h_target *ii2h_attach(const ii2h_target *owner) {
    auto *t = new ii2h_target();
    t->shard = owner->shard;
    t->index = owner->index;
    return t;
}
// NewShard(basedir) / NewInvertedIndex(basedir): loads what the directory holds.  NULL + message in err[] on failure.
ii2h_target *ii2h_open(ii2_ctx *ctx, int is_index, const char *basedir, char *err, uint64_t err_cap) {
    auto t = std::make_unique<ii2h_target>();
    try {
        if (is_index) t->index = std::make_shared<InvertedIndex>(ctx, basedir);
        else t->shard = std::make_shared<Shard>(ctx, basedir);
    } catch (const std::exception &e) {
        if (err && err_cap) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
        return nullptr;
    }
    return t.release();
}
// file.NewWriter / NewDirectWriter + Append* + Close in one call; key_out receives GetKey() (NUL-terminated, <= 31 chars)
int ii2h_file_write(ii2h_target *t, ii2_ctx *ctx, const char *dir, int direct, const uint8_t *term_bytes, const uint64_t *term_off, uint64_t n_terms,
                    const uint64_t *post_off, const uint32_t *values, char *key_out) {
    H_TRY(t, {
        Writer w(ctx, dir, direct != 0);
        auto terms = unpack_terms(term_bytes, term_off, n_terms);
        for (uint64_t i = 0; i < n_terms; i++)
            w.Append(TermValues{terms[i], std::vector<uint32_t>(values + post_off[i], values + post_off[i + 1])});
        w.Close();
        std::strncpy(key_out, w.GetKey().c_str(), 31);
        key_out[31] = 0;
    })
}
// file.NewReader(dir, key, min, max) drained with Next() into the target's result; *n_terms = 0 and rc 1 when the
// segment has nothing in range (vellum.ErrIteratorDone)
int ii2h_file_read(ii2h_target *t, ii2_ctx *ctx, const char *dir, const char *key, const uint8_t *mn, uint64_t mnl, int has_min,
                   const uint8_t *mx, uint64_t mxl, int has_max, uint64_t *n_terms) {
    try {
        Term a((const char *)mn, has_min ? mnl : 0), b((const char *)mx, has_max ? mxl : 0);
        t->result.clear();
        *n_terms = 0;
        Reader r(ctx, dir, key, has_min ? &a : nullptr, has_max ? &b : nullptr);
        TermValues tv;
        while (r.Next(&tv)) t->result.push_back(tv);
        r.Close();
        *n_terms = t->result.size();
        return 0;
    } catch (const Reader::NothingInRange &) {
        return 1;
    } catch (const std::exception &e) {
        t->err = e.what();
        return -1;
    }
}
int ii2h_remove_segment(ii2h_target *t, const char *dir, const char *key) {
    H_TRY(t, { file::remove_segment(dir, key); })
}
// the term dictionary file alone (no device involved): terms into the target's result, a direct segment's value as each
// term's one-element list; *direct = the file's mode
int ii2h_terms_read(ii2h_target *t, const char *dir, const char *key, int *direct, uint64_t *n_terms) {
    H_TRY(t, {
        file::TermFile tf = file::read_terms(dir, key);
        t->result.clear();
        for (size_t i = 0; i < tf.terms.size(); i++)
            t->result.push_back(TermValues{tf.terms[i], tf.direct ? std::vector<uint32_t>{tf.direct_vals[i]} : std::vector<uint32_t>{}});
        *direct = tf.direct ? 1 : 0;
        *n_terms = t->result.size();
    })
}
// removed.list alone: write batches (timestamp i, values vals[off[i] .. off[i+1])); read them back as RemovedLists.Values()
int ii2h_removed_write(ii2h_target *t, const char *dir, uint64_t n, const int64_t *ts, const uint64_t *off, const uint32_t *vals) {
    H_TRY(t, {
        file::RemovedBatches b;
        for (uint64_t i = 0; i < n; i++) b[ts[i]] = std::vector<uint32_t>(vals + off[i], vals + off[i + 1]);
        file::write_removed(dir, b);
    })
}
int ii2h_removed_read(ii2h_target *t, const char *dir, uint64_t *n_batches, uint64_t *n_ids) {
    H_TRY(t, {
        file::RemovedBatches b;
        t->ids.clear();
        *n_batches = 0;
        if (file::read_removed(dir, &b)) {
            *n_batches = b.size();
            for (auto &kv : b) t->ids.insert(t->ids.end(), kv.second.begin(), kv.second.end());
            std::sort(t->ids.begin(), t->ids.end());
        }
        *n_ids = t->ids.size();
    })
}
void ii2h_destroy(ii2h_target *t) { delete t; }
const char *ii2h_last_error(const ii2h_target *t) { return t->err.c_str(); }

int ii2h_put(ii2h_target *t, const uint8_t *bytes, const uint64_t *off, uint64_t n, uint32_t val) {
    H_TRY(t, { auto terms = unpack_terms(bytes, off, n); if (t->index) t->index->Put(terms, val); else t->shard->Put(terms, val); })
}
int ii2h_remove(ii2h_target *t, const uint32_t *vals, uint64_t n) {
    H_TRY(t, { std::vector<uint32_t> v(vals, vals + n); if (t->index) t->index->PutRemoved(v); else t->shard->Remove(v); })
}
int ii2h_merge(ii2h_target *t, int req, int m, int concurrency, int64_t *merged) {
    H_TRY(t, { *merged = t->index ? t->index->Merge(req, m, concurrency) : t->shard->Merge(req, m); })
}
// min/max: NULL pointer = nil
int ii2h_read(ii2h_target *t, const uint8_t *mn, uint64_t mnl, int has_min, const uint8_t *mx, uint64_t mxl, int has_max, uint64_t *n_terms) {
    H_TRY(t, {
        Term a((const char *)mn, has_min ? mnl : 0), b((const char *)mx, has_max ? mxl : 0);
        t->result = t->index ? t->index->Read(has_min ? &a : nullptr, has_max ? &b : nullptr)
                             : t->shard->Read(has_min ? &a : nullptr, has_max ? &b : nullptr);
        *n_terms = t->result.size();
    })
}
int ii2h_prefix_search(ii2h_target *t, const uint8_t *bytes, const uint64_t *off, uint64_t n, uint64_t *n_found) {
    H_TRY(t, {
        t->result.clear();
        for (auto &kv : t->index->PrefixSearch(unpack_terms(bytes, off, n))) t->result.push_back(TermValues{kv.first, kv.second});
        *n_found = t->result.size();
    })
}
int ii2h_intersect(ii2h_target *t, const uint8_t *bytes, const uint64_t *off, uint64_t n, uint64_t *n_ids) {
    H_TRY(t, { t->ids = t->index->Intersect(unpack_terms(bytes, off, n)); *n_ids = t->ids.size(); })
}
int ii2h_removed_values(ii2h_target *t, uint64_t *n_ids) {
    H_TRY(t, { Shard *s = t->shard ? t->shard.get() : t->index->OnlyShard(); t->ids = s ? s->RemovedValues() : std::vector<uint32_t>(); *n_ids = t->ids.size(); })
}
uint64_t ii2h_result_term_len(const ii2h_target *t, uint64_t i) { return t->result[i].term.size(); }
uint64_t ii2h_result_values_len(const ii2h_target *t, uint64_t i) { return t->result[i].values.size(); }
void ii2h_result_copy(const ii2h_target *t, uint64_t i, uint8_t *term, uint32_t *values) {
    std::memcpy(term, t->result[i].term.data(), t->result[i].term.size());
    if (!t->result[i].values.empty()) std::memcpy(values, t->result[i].values.data(), t->result[i].values.size() * 4);
}
void ii2h_ids_copy(const ii2h_target *t, uint32_t *out) {
    if (!t->ids.empty()) std::memcpy(out, t->ids.data(), t->ids.size() * 4);
}
uint64_t ii2h_segment_count(const ii2h_target *t) { return t->shard ? t->shard->SegmentCount() : 0; }
uint64_t ii2h_shard_count(const ii2h_target *t) { return t->index ? t->index->ShardCount() : 1; }

}  // extern "C"
