// Host side of the drop-in boundary: SINA's stage surface for the hot path
// (tray in, tray out) on top of the C ABI in include/sina_hip.h.
//
//   reference interface                       here
//   ------------------------------------      -------------------------------------
//   class tray           src/tray.h:41-57     sina::tray            (same fields)
//   class search         src/search.h:51-106  sina::search          (same virtuals)
//   class kmer_search    src/kmer_search.h    sina::kmer_search     (+ find_batch)
//   class famfinder      src/famfinder.h:51   sina::famfinder       (+ batch call)
//   class aligner        src/align.h:70-84    sina::aligner         (+ batch call)
//   query_arb (ARB DB)   src/query_arb.h      sina::reference_store (aligned FASTA /
//                                              packed arrays; ARB itself is out of scope)
//
// boost::program_options is not available here, so get_options_description /
// validate_vm become set_option(name, value) / validate_options() with the SAME
// option names and defaults (src/famfinder.cpp:144-211, src/align.cpp:231-274).
#pragma once

#include <condition_variable>
#include <atomic>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "cseq.h"
#include "sina_hip.h"

namespace sina {

// ---------------------------------------------------------------- search iface
enum ENGINE_TYPE { ENGINE_ARB_PT = 0, ENGINE_SINA_KMER = 1 };

class reference_store;

class search {
protected:
    search() = default;

public:
    search(const search &) = delete;
    search &operator=(const search &) = delete;
    virtual ~search() = default;
    struct result_item {
        result_item(float sc, const cseq *seq) : score(sc), sequence(seq) {}
        float score;
        const cseq *sequence;
        bool operator<(const result_item &o) const {
            if (score < o.score) return true;
            if (score > o.score) return false;
            return *sequence < *o.sequence;
        }
        bool operator>(const result_item &o) const { return !operator<(o); }
    };
    using result_vector = std::vector<result_item>;

    virtual double match(result_vector &family, const cseq &query, int min_match, int max_match, float min_score,
                         float max_score, reference_store *arb, bool noid, int minlen, int num_full,
                         int minlen_full, int range_cover, bool leave_query_out) = 0;
    virtual void find(const cseq &query, result_vector &results, unsigned int max) = 0;
    virtual unsigned int size() const = 0;
};

// ---------------------------------------------------------------- alignment_stats
// Only what the aligner reads (src/alignment_stats.h): name, width, per-column
// weights.  Default-constructed (width 0) selects scoring_scheme_simple
// (src/align.cpp:405-408).  Computing weights from ARB SAI data is out of scope.
class alignment_stats {
public:
    alignment_stats() = default;
    alignment_stats(std::string name_, std::vector<float> weights_)
        : name(std::move(name_)), weights(std::move(weights_)) {}
    const std::string &getName() const { return name; }
    unsigned int getWidth() const { return (unsigned int)weights.size(); }
    const std::vector<float> &getWeights() const { return weights; }
    // the default-constructed statistics ("no filter": simple scoring scheme).  One immutable
    // instance shared by every tray that needs nothing else; tray::destroy() leaves it alone.
    static alignment_stats *shared_default();

private:
    std::string name;
    std::vector<float> weights;
};

// ---------------------------------------------------------------- recycling of per-query heap objects
// A query's way through the stages creates and destroys two sequences, a result vector and their
// innards -- two dozen heap blocks, allocated by one pool thread and freed by another, at 100 000+
// queries a second: the allocator's locks were a third of the host's CPU time.  Objects that trays
// own go back to a per-thread free list instead (tray::destroy) and are handed out again in their
// default state WITH their heap blocks (the 6 KB base list above all), by the stages and the driver.
// (Role: sequences that arrive as queries and sequences that leave as alignments keep different heap blocks -- mask
// bytes there, packed words here -- and each kind goes back to a cache of its own, so that an object's next life
// finds the blocks it needs)
enum { cache_any = 0, cache_query_seq = 1, cache_aligned_seq = 2 };
template <typename T, int Role = cache_any> struct object_cache {
    static T *take() {
        auto &f = mine().free;
        if (f.empty() && !from_depot(f)) return new T();
        T *o = f.back();
        f.pop_back();
        return o;
    }
    static void give(T *o) {
        if (o == nullptr) return;
        auto &f = mine().free;
        reset(*o);
        f.push_back(o);
        if (f.size() >= kLocalMax) to_depot(f);
    }

private:
    // Threads that only hand out (the finder's driver thread building trays) and threads that only take back
    // (the sink's) exchange whole chunks through a depot: one lock per kChunk objects.
    static constexpr size_t kChunk = 128, kLocalMax = 4 * kChunk, kDepotChunks = 1024;
    struct list {
        std::vector<T *> free;
        ~list() {
            for (T *o : free) delete o;
        }
    };
    struct depot_t {
        std::mutex mu;
        std::vector<std::vector<T *>> chunks;
        ~depot_t() {
            for (auto &c : chunks)
                for (T *o : c) delete o;
        }
    };
    static list &mine() {
        thread_local list l;
        return l;
    }
    static depot_t &depot() {
        static depot_t d;
        return d;
    }
    static bool from_depot(std::vector<T *> &f) {
        depot_t &d = depot();
        std::lock_guard<std::mutex> lk(d.mu);
        if (d.chunks.empty()) return false;
        f.swap(d.chunks.back());
        d.chunks.pop_back();
        return !f.empty();
    }
    static void to_depot(std::vector<T *> &f) {
        std::vector<T *> chunk(f.end() - (std::ptrdiff_t)kChunk, f.end());
        f.resize(f.size() - kChunk);
        depot_t &d = depot();
        {
            std::lock_guard<std::mutex> lk(d.mu);
            if (d.chunks.size() < kDepotChunks) {
                d.chunks.push_back(std::move(chunk));
                return;
            }
        }
        for (T *o : chunk) delete o;
    }
    static void reset(cseq &c) { c.clear_all(); }
    template <typename V> static void reset(std::vector<V> &v) { v.clear(); }
};

// ---------------------------------------------------------------- tray
class tray {
public:
    unsigned int seqno{0};
    cseq *input_sequence{nullptr};
    cseq *aligned_sequence{nullptr};
    search::result_vector *alignment_reference{nullptr};
    search::result_vector *search_result{nullptr};
    std::stringstream log;
    alignment_stats *astats{nullptr};
    // set by famfinder when the scores of alignment_reference are raw k-mer counts of its internal
    // engine (value: k, negative with --fs-kmer-no-fast); 0 = unknown.  The aligner's containment
    // pre-filter (a member with fewer k-mers than the query cannot contain it) is only valid then.
    int family_scores_kmer_k{0};
    // with it: the query's own k-mer count under that k / filter, or -1 = not counted
    int query_kmer_count{-1};
    // The aligner's "scoring: raw=..., score=...;" line (src/align.cpp:455-457), kept as the five numbers it is made
    // of and the place in `log` where it belongs: three floats to text and a stream insert were 0.6 us per query, and
    // the text is only read when a sink asks for the log.  Readers take the log through log_text().
    struct score_note {
        float raw = 0.f, weight = 0.f, score = 0.f;
        uint32_t len = 0;
        int32_t aligned = 0;
        uint32_t at = 0;
        bool set = false;
        // the line as text (at most 160 characters); returns its length
        size_t render(char *line, size_t cap) const;
        // `log` with the line put in at `at`
        void merge_into(std::string &log_text) const;
    };
    score_note pending_score;
    std::string log_text() const;   // log, with a pending score line rendered in its place
    void clear_log() {
        log.str(std::string());
        log.clear();
        pending_score.set = false;
    }

    tray() = default;
    tray(const tray &o);
    tray &operator=(const tray &o);
    ~tray() = default;
    void destroy();
};

// ---------------------------------------------------------------- reference store
// Holds the aligned reference sequences on the host (cseq objects the result
// items point into, like query_arb's sequence cache) and owns the device context
// with the packed copy + k-mer index in HBM.
class reference_store {
public:
    // register / look up by "database path" (famfinder --db)
    // aligned FASTA file.  arb_id_order: number the sequences as the reference numbers an ARB database's
    // (host/id_order.h) instead of in file order -- tie-exact family order against a stock SINA binary,
    // and the order of names a .sidx written by it holds
    static std::shared_ptr<reference_store> open(const std::string &path, bool arb_id_order = false);
    static std::shared_ptr<reference_store> from_packed(const std::string &path_key, const uint32_t *ab,
                                                         const uint64_t *off, uint32_t n, uint32_t width,
                                                         const char *const *names = nullptr);
    static std::shared_ptr<reference_store> get(const std::string &path);
    static void close(const std::string &path);

    ~reference_store();
    const std::string &getFileName() const { return path; }
    unsigned int size() const { return (unsigned int)seqs.size(); }
    unsigned int getAlignmentWidth() const { return width; }
    const cseq &getCseq(unsigned int id) const { return seqs[id]; }
    const cseq &getCseq(const std::string &name) const;
    unsigned int id_of(const cseq *c) const { return (unsigned int)(c - seqs.data()); }
    bool owns(const cseq *c) const { return c >= seqs.data() && c < seqs.data() + seqs.size(); }
    // getBases() of a reference in upper case, computed once for the whole store (the aligner asks
    // for it for every family member of every query: icontains(), src/align.cpp:329-333)
    const std::string &upper_bases(unsigned int id);
    // "<acc>.<start>" of reference `id` as famfinder's family attribute lists it (computed once; a
    // later set_attr of acc / start on a reference is not picked up)
    const std::string &family_label(unsigned int id);
    // what famfinder's cascade reads of a candidate (src/famfinder.cpp:527-533,474-480) -- number of bases, first and
    // last column -- in one 12-byte record per reference: forty candidates per query out of a 100 000-sequence store
    // were forty sequence objects and their first and last bases, three cache misses each
    struct ref_meta {
        uint32_t size, first_pos, last_pos;
    };
    const ref_meta &meta(unsigned int id) const { return metas[id]; }
    std::vector<std::string> getSequenceNames() const;
    void loadKey(const cseq &c, const std::string &key) const;  // acc := name, start := "0" if absent
    std::vector<alignment_stats> &getAlignmentStats() { return vastats; }
    // field of a reference sequence (what loadKey would fetch from the ARB database: version, start,
    // stop, taxonomy paths ...); set before the stages run
    // (database fields of a reference; not while a pipeline is running)
    void set_attr(unsigned int id, const std::string &key, const std::string &value) {
        seqs.at(id).set_attr(key, value);
        std::lock_guard<std::mutex> lk(labels_mu);
        labels_ready.store(false, std::memory_order_release);  // (rebuilt by the next family_label)
    }

    // device side
    void set_device(int device) { device_id = device; }
    // ranks other than 0 of a multi-GPU run: the device copy of the references (and the index) arrives by
    // broadcast into buffers made by sina_hip_store_alloc_like -- device() then uploads nothing
    void expect_broadcast() { refs_by_broadcast = true; }
    sina_hip_ctx *device();                    // lazily creates the context and uploads the references
    void ensure_index(unsigned k, bool nofast);  // builds the k-mer index on the GPU once per (k, nofast)
    void adopt_index(unsigned k, bool nofast) {  // index already in HBM (broadcast from another rank)
        std::lock_guard<std::mutex> lk(gpu_mu);
        idx_k = (int)k;
        idx_nofast = nofast;
    }
    std::mutex &gpu_mutex() { return gpu_mu; }
    // how the current index came to be: "built" (on the GPU), "loaded <file>", "adopted" (broadcast)
    const std::string &index_origin() const { return idx_origin; }

    // A forked context (own stream + scratch, shared store/index: sina_hip_fork) for the duration
    // of one GPU call, so that batches in flight on different host threads overlap on the GPU.
    // Contexts are pooled per KIND of call: a context's scratch buffers grow to what its calls need
    // (the aligner's trace-back plane is tens of GB), growing means hipMalloc, and hipMalloc in
    // steady state stalls the device -- a context that only ever searches never grows aligner scratch.
    enum device_role { dev_search = 0, dev_align = 1, dev_compare = 2, dev_roles = 3 };
    class lease {
    public:
        lease(reference_store *s, sina_hip_ctx *c, int role) : st(s), c(c), role(role) {}
        lease(lease &&o) : st(o.st), c(o.c), role(o.role) { o.c = nullptr; }
        lease(const lease &) = delete;
        ~lease();
        sina_hip_ctx *get() const { return c; }

    private:
        reference_store *st;
        sina_hip_ctx *c;
        int role;
    };
    lease worker_device(device_role role);
    // at least `n` contexts of that kind exist and are warm (sina_hip_prewarm) -- called by a driver BEFORE it starts
    // the threads that will lease them: a context made, or grown, in the middle of a run stalls the device
    void reserve_workers(device_role role, unsigned n);

private:
    reference_store() = default;
    std::string path;
    std::vector<cseq> seqs;
    std::vector<ref_meta> metas;
    void fill_metas();
    unsigned int width{0};
    std::vector<alignment_stats> vastats;
    int device_id{0};
    bool refs_by_broadcast{false};
    sina_hip_ctx *ctx{nullptr};
    std::vector<sina_hip_ctx *> idle_forks[dev_roles];  // guarded by gpu_mu
    int idx_k{-1};
    bool idx_nofast{false};
    std::string idx_origin;
    std::vector<std::string> ubases;
    std::once_flag ubases_once;
    std::vector<std::string> labels;
    std::mutex labels_mu;
    std::atomic<bool> labels_ready{false};
    std::mutex gpu_mu;
};

// ---------------------------------------------------------------- .sidx index cache
// The reference's on-disk k-mer index (src/kmer_search.cpp:279-351, src/idset.h:386-410) to and from
// the CSR form the GPU uses (host/sidx.cpp).  sidx_load returns false (and why) when the file does
// not fit, as try_load does.
void sidx_store(const std::string &path, unsigned k, bool nofast, const std::vector<std::string> &names,
                const std::vector<uint32_t> &offsets, const std::vector<uint32_t> &ids);
bool sidx_load(const std::string &path, unsigned k, bool nofast, std::vector<std::string> *names,
               std::vector<uint32_t> *offsets, std::vector<uint32_t> *ids, std::string *why = nullptr);

// ---------------------------------------------------------------- kmer_search
class kmer_search : public search {
public:
    static kmer_search *get_kmer_search(const std::string &filename, int k = 10, bool nofast = false);
    static void release_kmer_search(const std::string &filename, int k = 10, bool nofast = false);

    double match(result_vector &, const cseq &, int, int, float, float, reference_store *, bool, int, int, int, int,
                 bool) override;
    void find(const cseq &query, result_vector &results, unsigned int max) override;
    // one launch for many queries; results[i] gets min(max, size()) items
    // (kmer_counts, optional: the number of k-mers of every query, with multiplicity -- what a reference
    // holding all of them scores; counted in the pass that packs the queries for the device)
    void find_batch(const std::vector<const cseq *> &queries, std::vector<result_vector> &results,
                    unsigned int max, std::vector<uint32_t> *kmer_counts = nullptr);
    unsigned int size() const override;
    ~kmer_search() override;

    class impl;

private:
    explicit kmer_search(std::shared_ptr<impl> pimpl_);
    std::shared_ptr<impl> pimpl;
};

// ---------------------------------------------------------------- famfinder
enum TURN_TYPE { TURN_NONE = 0, TURN_REVCOMP = 1, TURN_ALL = 2 };

class famfinder {
    class impl;
    std::shared_ptr<impl> pimpl;

public:
    famfinder();
    famfinder(const famfinder &o);
    famfinder &operator=(const famfinder &o);
    ~famfinder();
    tray operator()(const tray &t);
    void operator()(std::vector<tray> &batch);  // batched form of the same stage
    int turn_check(const cseq &query, bool all);

    // option names as on the SINA command line: "db", "turn", "fs-kmer-len", "fs-req", "fs-min",
    // "fs-max", "fs-msc", "fs-req-full", "fs-full-len", "fs-req-gaps", "fs-min-len",
    // "fs-kmer-no-fast", "fs-msc-max", "fs-leave-query-out", "fs-cover-gene", "filter"
    static void set_option(const std::string &name, const std::string &value);
    static void reset_options();
    static void validate_options();
    static ENGINE_TYPE get_engine();
};

// ---------------------------------------------------------------- aligner
enum OVERHANG_TYPE { OVERHANG_ATTACH, OVERHANG_REMOVE, OVERHANG_EDGE };
enum LOWERCASE_TYPE { LOWERCASE_NONE, LOWERCASE_ORIGINAL, LOWERCASE_UNALIGNED };
enum INSERTION_TYPE { INSERTION_SHIFT, INSERTION_FORBID, INSERTION_REMOVE };

class aligner {
public:
    struct options {
        bool realign;
        OVERHANG_TYPE overhang;
        LOWERCASE_TYPE lowercase;
        INSERTION_TYPE insertion;
        bool calc_idty;
        bool fs_no_graph;
        float fs_weight;
        float match_score, mismatch_score, gap_penalty, gap_ext_penalty;
        bool debug_graph, write_used_rels, use_subst_matrix;
        bool device_graph;  // build the family DAG on the GPU (default) or on the host
        std::string database;  // reference store the DAGs are built from (same as famfinder "db")
    };
    static options *opts;
    // The "scoring: ..." line of a tray's log as a note to be rendered by tray::log_text() instead of text in `log`
    // (tray::score_note).  Off by default: a tray handed back to code that reads `log` itself -- SINA's own stages
    // behind batched<aligner> -- carries the whole text; the pipeline driver of host/capi.cpp, whose sink reads the
    // log through log_text(), switches it on.
    bool defer_score_line = false;
    aligner();
    aligner(const aligner &rhs);
    ~aligner();
    aligner &operator=(const aligner &rhs);
    tray operator()(tray t);
    void operator()(std::vector<tray> &batch);

    // "realign", "overhang", "lowercase", "insertion", "fs-weight", "match-score",
    // "mismatch-score", "pen-gap", "pen-gapext", "write-used-rels", "calc-idty", "db"
    static void set_option(const std::string &name, const std::string &value);
    static void reset_options();
    static void validate_options();
};

// ---------------------------------------------------------------- search_filter
// src/search_filter.{h,cpp}: for an ALIGNED query, 1000 k-mer candidates -> identity against each
// (cseq_comparator) -> best 10 above min-sim -> nearest_slv / lca_<field> / copy_* attributes.
enum CMP_IUPAC_TYPE { CMP_IUPAC_OPTIMISTIC, CMP_IUPAC_PESSIMISTIC, CMP_IUPAC_EXACT };
enum CMP_DIST_TYPE { CMP_DIST_NONE, CMP_DIST_JC };
enum CMP_COVER_TYPE {
    CMP_COVER_ABS, CMP_COVER_QUERY, CMP_COVER_TARGET, CMP_COVER_OVERLAP, CMP_COVER_ALL, CMP_COVER_AVERAGE,
    CMP_COVER_MIN, CMP_COVER_MAX, CMP_COVER_NOGAP
};
// src/cseq_comparator.{h,cpp}: the score from the six counters (the counting itself is
// sina_hip_compare on the GPU; counts() is the host restatement used for single pairs)
class cseq_comparator {
public:
    cseq_comparator() = default;
    cseq_comparator(CMP_IUPAC_TYPE iupac, CMP_DIST_TYPE dist, CMP_COVER_TYPE cover, bool filter_lc)
        : iupac_rule(iupac), dist_rule(dist), cover_rule(cover), filter_lc_rule(filter_lc) {}
    float operator()(const cseq &query, const cseq &target) const;
    float score(const sina_hip_match_counts &m) const;
    static void counts(const cseq &query, const cseq &target, CMP_IUPAC_TYPE iupac, bool filter_lc,
                       sina_hip_match_counts *m);
    CMP_IUPAC_TYPE iupac_rule{CMP_IUPAC_OPTIMISTIC};
    CMP_DIST_TYPE dist_rule{CMP_DIST_NONE};
    CMP_COVER_TYPE cover_rule{CMP_COVER_QUERY};
    bool filter_lc_rule{false};
};

class search_filter {
    struct priv_data;
    std::shared_ptr<priv_data> data;

public:
    struct options;
    static options *opts;
    search_filter();
    search_filter(const search_filter &);
    search_filter &operator=(const search_filter &);
    ~search_filter();
    tray operator()(tray t);
    void operator()(std::vector<tray> &batch);

    // "search-db", "search-min-sim", "search-max-result", "lca-fields", "lca-quorum", "search-all",
    // "search-no-fast", "search-kmer-candidates", "search-kmer-len", "search-ignore-super",
    // "search-copy-fields", "search-iupac", "search-correction", "search-cover",
    // "search-filter-lowercase", "db" (fallback for search-db)
    static void set_option(const std::string &name, const std::string &value);
    static void reset_options();
    static void validate_options();
    static const char *fn_nearest;  // "nearest_slv" (query_arb.cpp:126)
};
std::string search_filter_database();  // the store the search stage was configured for

// ---------------------------------------------------------------- FASTA I/O + accuracy metrics
enum FASTA_META_TYPE { FASTA_META_NONE = 0, FASTA_META_HEADER = 1, FASTA_META_COMMENT = 2, FASTA_META_CSV = 3 };

// src/rw_fasta.{h,cpp}: source and sink of the pipeline for FASTA files (host/rw_fasta.cpp)
class rw_fasta {
public:
    struct options;
    static options *opts;
    // "meta-fmt", "line-length", "min-idty", "fasta-write-dna", "fasta-write-dots", "fasta-idx", "fasta-block"
    static void set_option(const std::string &name, const std::string &value);
    static void reset_options();

    class reader {
        struct priv_data;
        std::shared_ptr<priv_data> data;

    public:
        explicit reader(const std::string &infile);
        reader(const reader &);
        reader &operator=(const reader &);
        ~reader();
        bool operator()(tray &t);  // false at end of input; fills seqno and input_sequence
        int skipped() const;       // sequences dropped for characters outside the IUPAC alphabet
    };
    class writer {
        struct priv_data;
        std::shared_ptr<priv_data> data;

    public:
        explicit writer(const std::string &outfile, unsigned int copy_relatives = 0);
        writer(const writer &);
        writer &operator=(const writer &);
        ~writer();
        tray operator()(tray t);
        // composes the records of a batch's aligned sequences on the loop pool, ahead of the ordered operator()
        // calls for those trays (which then only hand the text to the sink); same bytes either way
        void precompose(const std::vector<tray> &batch);
        int written() const;
        int excluded() const;
        void flush();
    };
};

// src/log.cpp Log::printer: per-sequence report and the --show-dist accuracy metrics (sps: identity
// of the new alignment with the input's own alignment; cpm: loss of identity with the closest
// reference; idty: identity of the input with that reference)
class log_printer {
    struct priv_data;
    std::shared_ptr<priv_data> data;

public:
    struct summary {
        int sequences = 0;
        double avg_sps = 0, avg_cpm = 0, avg_idty = 0, avg_bps = 0;
    };
    // one tray's report: its text, and what the running totals take from it
    struct report {
        std::string text;
        bool aligned = false, has_sps = false, has_closest = false;
        float sps = 0, idty = 0, cpm = 0;
    };
    explicit log_printer(bool show_dist);
    log_printer(const log_printer &);
    log_printer &operator=(const log_printer &);
    ~log_printer();
    tray operator()(tray t, std::ostream &log);  // serial stage: call in sequence order (= render + commit)
    void render(tray &t, report &out) const;      // any thread, any order (sets the aligned sequence's attributes)
    void commit(const report &r, std::ostream &log);  // in sequence order
    summary totals() const;
};

// text of a family attribute kept as a list (annotated_cseq::lazy_text; famfinder sets it, sinks may keep the list)
void render_family_list(const void *store, const uint64_t *items, size_t n, std::string &out);

// ---------------------------------------------------------------- batching shim
// SINA calls a stage once per tray from many TBB workers (function_node with
// unlimited concurrency, src/sina.cpp:497-519); the GPU wants thousands of
// queries per launch.  batched<Stage> has the single-tray call signature of the
// reference stage, is re-entrant, and groups concurrent callers into one batch.
template <class Stage> class batched {
public:
    explicit batched(Stage s, size_t max_batch = 1024, unsigned linger_us = 300)
        : st(std::make_shared<state>(std::move(s), max_batch, linger_us)) {}
    tray operator()(tray t) {
        std::future<tray> f;
        {
            std::unique_lock<std::mutex> lk(st->mu);
            st->pending.emplace_back(std::move(t), std::promise<tray>());
            f = st->pending.back().second.get_future();
            st->cv.notify_all();
        }
        return f.get();
    }

private:
    struct state {
        Stage stage;
        size_t max_batch;
        unsigned linger_us;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<std::pair<tray, std::promise<tray>>> pending;
        bool stop = false;
        std::thread worker;
        state(Stage s, size_t mb, unsigned lu) : stage(std::move(s)), max_batch(mb), linger_us(lu) {
            worker = std::thread([this] { run(); });
        }
        ~state() {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            worker.join();
        }
        void run() {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return stop || !pending.empty(); });
                if (stop && pending.empty()) return;
                // linger briefly so that concurrent callers end up in the same launch
                cv.wait_for(lk, std::chrono::microseconds(linger_us),
                            [this] { return stop || pending.size() >= max_batch; });
                // (at most max_batch callers per launch, first come first served: the rest is the next batch)
                std::vector<std::pair<tray, std::promise<tray>>> work;
                if (pending.size() <= max_batch) {
                    work.swap(pending);
                } else {
                    work.assign(std::make_move_iterator(pending.begin()), std::make_move_iterator(pending.begin() + (std::ptrdiff_t)max_batch));
                    pending.erase(pending.begin(), pending.begin() + (std::ptrdiff_t)max_batch);
                }
                lk.unlock();
                std::vector<tray> batch;
                batch.reserve(work.size());
                for (auto &w : work) batch.push_back(w.first);
                try {
                    stage(batch);
                    for (size_t i = 0; i < work.size(); i++) work[i].second.set_value(batch[i]);
                } catch (...) {
                    for (auto &w : work) w.second.set_exception(std::current_exception());
                }
                lk.lock();
            }
        }
    };
    std::shared_ptr<state> st;
};

// ---------------------------------------------------------------- helpers
// parallel for over [0, n) on a process-wide pool (stands in for TBB's workers)
// first occurrence of `needle` in `hay`, like std::string::find (the aligner's exact-relative test; stages.cpp)
size_t find_bases(const std::string &hay, const std::string &needle, int force_scalar = 0);
void parallel_for(size_t n, const std::function<void(size_t)> &fn);
void set_host_threads(unsigned n);
// identical queries of a batch go to the device once (k-mer search; DAG + DP when the family is the same too):
// on by default; off = every tray's query is searched and aligned on its own
void set_batch_dedup(bool on);
unsigned host_threads();
std::string host_profile_dump(bool reset);  // per-phase wall time when SINA_HOST_PROFILE is set
void host_profile_mark(const char *what);  // (`what` must outlive the process: a literal)
void host_profile_add_cpu(const char *what, double seconds);  // (no-op unless SINA_HOST_PROFILE is set)
// Fine-grained timers for per-query code (SINA_HOST_PROFILE=1): time-stamp-counter deltas summed in
// per-slot atomics -- no system call and no lock per sample, unlike host_profile_add_cpu.
//   uint64_t t = host_tsc();  ...  t = host_tick("al.finish: assemble", t);   // adds, returns "now"
uint64_t host_tsc();
uint64_t host_tick(const char *what, uint64_t since);
double host_thread_cpu_seconds();
// a stage driver's thread ends: its CPU seconds, kernel-mode seconds, page faults and context switches
// (thousands) go into the profile under "<who> ..." (no-op unless SINA_HOST_PROFILE is set)
void host_profile_thread_exit(const char *who);
class host_phase {                           // a named phase of the calling thread (profile / SINA_HOST_TRACE)
public:
    explicit host_phase(const char *name);
    ~host_phase();
    host_phase(const host_phase &) = delete;

private:
    void *impl;
};

// Family DAG on the host (flat CSR) -- used when aligner option device_graph is off
// and by tests; see src/mseq.cpp:47-118 for the behaviour it reproduces.
struct host_graph {
    std::vector<uint32_t> pos;
    std::vector<uint8_t> mask;
    std::vector<float> weight;
    std::vector<uint32_t> pred_off, pred, succ_minpos;
    uint32_t width = 0;
    std::vector<float> score16;  // --fs-no-graph only: match term per node and query iupac mask (16 per node)
};
void build_family_graph(const std::vector<const cseq *> &family, float fs_weight, host_graph *g);
// --fs-no-graph (src/pseq.cpp:41-112, src/pseq.h:42-117): the family as a profile -- one node per column, a
// chain -- with base_profile::comp() of every node against the fifteen iupac codes tabulated for the
// scoring_scheme_profile(match, mismatch, gap, gap_ext) the aligner builds (src/align.cpp:429-433)
void build_family_profile(const std::vector<const cseq *> &family, float match, float mismatch, float gap,
                          float gap_ext, host_graph *g);
// comp() of a base's own profile with itself, per iupac mask (16 entries; the sum_weight term of backtrack())
void profile_self_scores(float match, float mismatch, float gap, float gap_ext, float *out16);

}  // namespace sina
