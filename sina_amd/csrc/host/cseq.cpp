// Host-side cseq container: behaviour of src/cseq.cpp / src/aligned_base.cpp, written from the specs
// in SURVEY.md (A.4 "container rule", A.5 NAST fix-up) as flat, index-based code.  The log and
// exception texts are the reference's own (they are part of the result: align_log_slv); the
// structure is not.
#include "cseq.h"

#include <cstring>
#include <mutex>
#include <unordered_map>

#include <algorithm>

#include <atomic>
#include <immintrin.h>
#include <sys/mman.h>
#include <vector>

namespace sina {

__attribute__((target("avx2"))) static void stream_copy_avx2(unsigned char *d, const unsigned char *s, size_t n) {
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;  // (bytes up to the destination's next 32-byte boundary)
    if (head) {
        memcpy(d, s, head);
        d += head, s += head, n -= head;
    }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 96), e);
    }
    _mm_sfence();
    if (i < n) memcpy(d + i, s + i, n - i);
}
void stream_copy(void *dst, const void *src, size_t bytes) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (!avx2 || bytes < 1024) {
        memcpy(dst, src, bytes);
        return;
    }
    stream_copy_avx2(static_cast<unsigned char *>(dst), static_cast<const unsigned char *>(src), bytes);
}

// ---- the pool behind base_block_allocator (cseq.h)
namespace {
constexpr size_t kMinClass = 12, kMaxClass = 16;  // 4 KB .. 64 KB
constexpr size_t kRegion = (size_t)2 << 20;
struct block_pool {
    struct depot_t {
        std::mutex mu;
        std::vector<void *> free[kMaxClass - kMinClass + 1];
        std::atomic<size_t> n_free[kMaxClass - kMinClass + 1] = {};  // (sizes of the lists above, readable without the lock)
        unsigned char *region = nullptr;
        size_t left = 0;
    };
    static depot_t &depot() {
        static depot_t *d = new depot_t();  // (never destroyed: blocks outlive static destruction order)
        return *d;
    }
    struct local_t {
        std::vector<void *> free[kMaxClass - kMinClass + 1];
        unsigned char *region = nullptr;  // what is left of the 2 MB region this thread is carving
        size_t left = 0;
    };
    // A thread's free lists.  Other thread-local objects (the stages' object caches) free base lists from THEIR
    // destructors, possibly after this one has run: from then on the thread goes to the depot directly.
    struct holder {
        local_t *p = nullptr;
        bool dead = false;
        ~holder() {
            dead = true;
            if (!p) return;
            depot_t &d = depot();
            {
                std::lock_guard<std::mutex> lk(d.mu);
                for (size_t c = 0; c <= kMaxClass - kMinClass; c++) {
                    for (void *b : p->free[c]) d.free[c].push_back(b);
                    d.n_free[c].store(d.free[c].size(), std::memory_order_relaxed);
                }
            }
            delete p;
            p = nullptr;
        }
    };
    static local_t *mine() {
        thread_local holder h;
        if (h.dead) return nullptr;
        if (!h.p) h.p = new local_t();
        return h.p;
    }
    static size_t class_of(size_t bytes) {
        size_t c = kMinClass;
        while (((size_t)1 << c) < bytes) c++;
        return c;
    }
    static void *take(size_t c) {
        local_t *l = mine();
        if (l) {
            auto &f = l->free[c - kMinClass];
            if (!f.empty()) {
                void *p = f.back();
                f.pop_back();
                return p;
            }
        }
        depot_t &d = depot();
        if (d.n_free[c - kMinClass].load(std::memory_order_relaxed) != 0) {
            std::lock_guard<std::mutex> lk(d.mu);
            auto &df = d.free[c - kMinClass];
            if (!df.empty()) {
                void *p = df.back();
                df.pop_back();
                // (a batch at a time: the thread that frees blocks and the one that takes them are rarely the same)
                if (l)
                    for (int i = 0; i < 63 && !df.empty(); i++) {
                        l->free[c - kMinClass].push_back(df.back());
                        df.pop_back();
                    }
                d.n_free[c - kMinClass].store(df.size(), std::memory_order_relaxed);
                return p;
            }
        }
        // fresh memory: every thread carves its own 2 MB regions (no lock, and a region's pages are first touched by
        // the thread that asked for it)
        const size_t sz = (size_t)1 << c;
        unsigned char *&region = l ? l->region : d.region;
        size_t &left = l ? l->left : d.left;
        std::unique_lock<std::mutex> lk(d.mu, std::defer_lock);
        if (!l) lk.lock();
        if (left < sz) {
            void *m = nullptr;
            if (posix_memalign(&m, kRegion, kRegion) != 0) throw std::bad_alloc();
            (void)madvise(m, kRegion, MADV_HUGEPAGE);
            region = static_cast<unsigned char *>(m);
            left = kRegion;
        }
        void *p = region;
        region += sz;
        left -= sz;
        return p;
    }
    static void give(void *p, size_t c) {
        local_t *l = mine();
        if (!l) {
            depot_t &d = depot();
            std::lock_guard<std::mutex> lk(d.mu);
            d.free[c - kMinClass].push_back(p);
            d.n_free[c - kMinClass].store(d.free[c - kMinClass].size(), std::memory_order_relaxed);
            return;
        }
        auto &f = l->free[c - kMinClass];
        f.push_back(p);
        if (f.size() >= 512) {  // (a thread that only frees: half of its list goes to the depot)
            depot_t &d = depot();
            std::lock_guard<std::mutex> lk(d.mu);
            for (int i = 0; i < 256; i++) {
                d.free[c - kMinClass].push_back(f.back());
                f.pop_back();
            }
            d.n_free[c - kMinClass].store(d.free[c - kMinClass].size(), std::memory_order_relaxed);
        }
    }
};
}  // namespace
void *base_block_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes > ((size_t)1 << kMaxClass)) {
        void *p = malloc(bytes);
        if (!p) throw std::bad_alloc();
        return p;
    }
    return block_pool::take(block_pool::class_of(bytes));
}
void base_block_free(void *p, size_t bytes) {
    if (p == nullptr) return;
    if (bytes == 0) bytes = 1;
    if (bytes > ((size_t)1 << kMaxClass)) {
        free(p);
        return;
    }
    block_pool::give(p, block_pool::class_of(bytes));
}

namespace fn {
const char *turn = "turn";
const char *acc = "acc";
const char *start = "start";
const char *used_rels = "used_rels";
const char *fullname = "full_name";
const char *qual = "align_quality_slv";
const char *head = "align_cutoff_head_slv";
const char *tail = "align_cutoff_tail_slv";
const char *date = "aligned_slv";
const char *idty = "align_ident_slv";
const char *family = "align_family_slv";
const char *align_log = "align_log_slv";
const char *filter = "align_filter_slv";
const char *cutoff_head = "align_cutoff_head_slv";
const char *cutoff_tail = "align_cutoff_tail_slv";
}  // namespace fn

// ---- IUPAC (src/aligned_base.cpp:70-114): letter -> set of {A,G,C,T/U}

namespace {
struct iupac_tables {
    unsigned char to_mask[256];
    unsigned char rna[32], dna[32];
    iupac_tables() {
        for (auto &m : to_mask) m = 0;
        const char *letters = "AGCTURYKMSWBDHVN";
        const unsigned char masks[] = {1, 2, 4, 8, 8, 3, 12, 10, 5, 6, 9, 14, 11, 13, 7, 15};
        for (int i = 0; letters[i]; i++) {
            to_mask[(unsigned char)letters[i]] = masks[i];
            to_mask[(unsigned char)(letters[i] - 'A' + 'a')] = masks[i] | 16;
        }
        // mask -> letter: index = A|G<<1|C<<2|T<<3
        const char *order = ".AGRCMSVUWKDYHBN";
        for (int i = 0; i < 16; i++) {
            rna[i] = (unsigned char)order[i];
            rna[i + 16] = (unsigned char)(i ? order[i] - 'A' + 'a' : '.');
            dna[i] = rna[i] == 'U' ? 'T' : rna[i];
            dna[i + 16] = rna[i + 16] == 'u' ? 't' : rna[i + 16];
        }
    }
};
const iupac_tables &tables() {
    static const iupac_tables t;
    return t;
}
}  // namespace

base_iupac::value_type base_iupac::from_char(unsigned char c) {
    const value_type d = tables().to_mask[c];
    if (d == 0 && c != '-' && c != '.') throw bad_character_exception(c);
    return d;
}
unsigned char base_iupac::iupac_rna() const { return tables().rna[_data & 31]; }
unsigned char base_iupac::iupac_dna() const { return tables().dna[_data & 31]; }
base_iupac &base_iupac::complement() {  // A<->T/U, G<->C, case kept
    const value_type d = _data;
    _data = (value_type)(((d & 2) << 1) | ((d & 4) >> 1) | ((d & 1) << 3) | ((d & 8) >> 3) | (d & 16));
    return *this;
}

// ---- cseq_base

cseq_base::cseq_base(const char *_name, const char *_data) : name(_name) {
    if (_data != nullptr) append(_data);
}

void cseq_base::clearSequence() {
    drop_masks();
    bases.clear();
    alignment_width = 0;
}

cseq_base &cseq_base::append(const char *str) {
    touch();
    // (an aligned line is 97 % gap characters: runs of them are skipped with strspn, which the C
    // library vectorises, instead of a branch per character)
    for (;;) {
        const size_t gaps = strspn(str, "-.");
        alignment_width += (vidx_type)gaps;
        str += gaps;
        const char c = *str;
        if (c == 0) break;
        ++str;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
        bases.emplace_back(alignment_width, (unsigned char)c);
        alignment_width++;
    }
    return *this;
}

// Container rule: a base may share the column of its predecessor
// (insertions are resolved later) but never precede it.
cseq_base &cseq_base::append(const aligned_base &ab) {
    touch();
    if (ab.getPosition() >= alignment_width) {
        bases.push_back(ab);
        alignment_width = ab.getPosition();
    } else {
        bases.emplace_back(alignment_width, ab.getBase());
    }
    return *this;
}

void cseq_base::setWidth(vidx_type newWidth) {
    // (a dense sequence's last base sits in column size() - 1; a width that leaves every base where it is
    // leaves the mask bytes valid too)
    if (size() == 0 || newWidth >= (unpacked ? (vidx_type)dmask.size() : bases.back().getPosition() + 1)) {
        alignment_width = newWidth;
        return;
    }
    touch();
    if (newWidth < size()) throw std::runtime_error("Attempted to shrink alignment width below base count");
    // pack the right-most bases against the new right edge
    const unsigned int n = size();
    unsigned int skip = 0;
    while (skip < n && bases[n - skip - 1].getPosition() + skip >= newWidth) skip++;
    for (unsigned int i = skip; i > 0; --i) bases[n - i].setPosition(newWidth - i);
    alignment_width = newWidth;
}

void cseq_base::reverse() {
    touch();
    std::reverse(bases.begin(), bases.end());
    for (auto &b : bases) b.setPosition(alignment_width - 1 - b.getPosition());
}
void cseq_base::complement() {
    touch();
    for (auto &b : bases) b.complement();
}
void cseq_base::upperCaseAll() {
    touch();
    for (auto &b : bases) b.setUpperCase();
}

std::string cseq_base::getAligned(bool nodots, bool dna) const {
    unpack();
    std::string out;
    out.reserve(alignment_width);
    char gap = nodots ? '-' : '.';  // leading and trailing gaps are dots unless nodots
    unsigned int cursor = 0;
    for (const auto &b : bases) {
        const unsigned int pos = b.getPosition();
        if (pos > cursor) out.append(pos - cursor, gap);
        gap = '-';
        out.push_back((char)(dna ? b.getBase().iupac_dna() : b.getBase().iupac_rna()));
        cursor = pos + 1;
    }
    if (cursor < alignment_width) out.append(alignment_width - cursor, nodots ? '-' : '.');
    return out;
}

std::string cseq_base::getBases() const {
    std::string s;
    if (!dmask.empty()) {  // (dense: the letters of the mask bytes)
        s.reserve(dmask.size());
        for (const uint8_t m : dmask) s.push_back((char)base_iupac::from_mask(m).iupac_rna());
        return s;
    }
    s.reserve(bases.size());
    for (const auto &b : bases) s.push_back((char)b.getBase().iupac_rna());
    return s;
}

char cseq_base::operator[](vidx_type i) const {
    unpack();
    auto it = std::lower_bound(bases.begin(), bases.end(), aligned_base(i, '.'));
    if (it != bases.end() && it->getPosition() == i) return (char)it->getBase().iupac_rna();
    return '-';
}

std::ostream &operator<<(std::ostream &out, const cseq_base &c) { return out << c.getName(); }

// ---- attribute names, interned.  A per-thread table in front of the process-wide one: the few dozen
// names are looked up millions of times and added a few dozen times.
namespace {
const std::string *intern_attr_name(std::string_view key) {
    struct sv_hash {
        size_t operator()(std::string_view v) const { return std::hash<std::string_view>()(v); }
    };
    static std::mutex mu;
    static std::unordered_map<std::string_view, const std::string *, sv_hash> all;  // (keys view the interned strings)
    thread_local std::unordered_map<std::string_view, const std::string *, sv_hash> mine;
    auto it = mine.find(key);
    if (it != mine.end()) return it->second;
    const std::string *name;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto jt = all.find(key);
        if (jt == all.end()) {
            const std::string *fresh = new std::string(key);  // (never freed: the table lives as long as the process)
            jt = all.emplace(std::string_view(*fresh), fresh).first;
        }
        name = jt->second;
    }
    mine.emplace(std::string_view(*name), name);
    return name;
}
}  // namespace

const std::string *annotated_cseq::interned(std::string_view key) { return intern_attr_name(key); }

annotated_cseq::variant &annotated_cseq::slot(std::string_view key) {
    if (lazy.name && *lazy.name == key) lazy.name = nullptr;  // (about to be overwritten: the pending text is moot)
    size_t at = 0;
    for (; at < attributes.size(); at++) {
        const int c = attributes[at].name->compare(key);
        if (c == 0) return attributes[at].second;
        if (c > 0) break;  // (kept in key order: what iterating the reference's std::map gives)
    }
    attributes.insert(attributes.begin() + (std::ptrdiff_t)at, attr{intern_attr_name(key), variant()});
    return attributes[at].second;
}

// NAST insertion fix-up (reference src/cseq.cpp:456-594).
//
// After backtracking, several bases may sit in one column (insertions relative
// to the reference).  Walk the bases; `anchor` is the last base whose column is
// final.  A run of bases sharing the anchor's column is spread over the free
// columns up to the next differently placed base (right-aligned).  If the free
// range is too small, the run swallows neighbouring bases, always towards the
// nearer free column, until it fits.
void cseq_base::fix_duplicate_positions(std::ostream &log, bool lowercase, bool remove) {
    touch();
    if (remove) log << "insertion=remove not implemented, using shift; ";
    const long n = (long)bases.size();
    auto col = [&](long i) { return bases[(size_t)i].getPosition(); };
    // A placement problem: the bases [first, last] must go into the free columns [lo, hi).
    struct span {
        long first, last;
        idx_type lo, hi;
        idx_type need() const { return (idx_type)(last - first + 1); }
        idx_type room() const { return hi - lo; }
    };
    // Nearest free column left of the span, and the base from which everything up to the span
    // would have to move along with it: bases that sit column to column with their right neighbour
    // are walked over.  -1: the alignment's left edge is reached without finding one.
    auto free_left = [&](const span &s, long *from) -> int {
        *from = s.first;
        if (s.first == 0) return s.lo > 0 ? (int)(s.lo - 1) : -1;
        if (col(s.first - 1) + 1 < s.lo) return (int)(s.lo - 1);
        long l = s.first - 1;
        while (l != 0 && col(l - 1) + 1 >= col(l)) --l;
        *from = l;
        return (int)(col(l) - 1u);
    };
    // ... and the same to the right (-1: the alignment's right edge)
    auto free_right = [&](const span &s, long *upto) -> int {
        *upto = s.last;
        if (s.last + 1 == n) return s.hi < alignment_width ? (int)s.hi : -1;
        if (col(s.last + 1) > s.hi) return (int)s.hi;
        long r = s.last + 1;
        while (r + 1 != n && col(r) + 1 >= col(r + 1)) ++r;
        *upto = r;
        return (int)(col(r) + 1);
    };

    idx_type placed_total = 0, placed_longest = 0, last_run_before_shifting = 0;
    for (long anchor = 0; anchor < n;) {
        // the bases after `anchor` that share its column are an insertion
        long after = anchor + 1;
        while (after < n && col(after) == col(anchor)) ++after;
        if (after == anchor + 1) {
            anchor = after;
            continue;
        }
        span s{anchor + 1, after - 1, col(anchor) + 1, after == n ? alignment_width : col(after)};
        last_run_before_shifting = s.need();
        if (s.room() >= s.need()) {
            s.lo = s.hi - s.need();  // enough room: right-aligned in the free range
        } else {
            log << "shifting bases to fit in " << s.need() << " bases at pos " << s.lo << " to " << s.hi << ";";
            while (s.room() < s.need()) {  // grow towards the nearer free column, taking the bases in between along
                long from, upto;
                const int lgap = free_left(s, &from), rgap = free_right(s, &upto);
                const bool to_left = rgap == -1 || (lgap != -1 && (idx_type)(s.lo - (idx_type)lgap) <=
                                                                      (idx_type)((idx_type)rgap - (s.hi - 1)));
                if (to_left && lgap == -1)
                    throw std::runtime_error("ERROR: no space to left and right?? sequence longe than alignment?!");
                if (to_left) {
                    s.first = from;
                    s.lo = (idx_type)lgap;
                } else {
                    s.last = upto;
                    s.hi = (idx_type)rgap + 1;
                }
            }
        }
        idx_type c = s.lo;
        for (long i = s.first; i <= s.last; ++i) {
            bases[(size_t)i].setPosition(c++);
            if (lowercase) bases[(size_t)i].setLowerCase();
        }
        placed_total += s.need();
        placed_longest = std::max(placed_longest, s.need());
        anchor = s.last + 1;
    }
    if (placed_total > 0) {
        log << "total inserted bases=" << placed_total << ";"
            << "longest insertion=" << placed_longest << ";"
            << "total inserted bases before shifting=" << last_run_before_shifting << ";";
    }
}

}  // namespace sina
