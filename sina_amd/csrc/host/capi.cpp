// Plain-C driver over the C++ stage mirror, for pytest / bench.py (ctypes).
// Not part of the drop-in boundary (that is include/sina_hip.h below the stages
// and the famfinder/aligner classes above it); this only lets Python feed trays
// through the stages and read the results back.
#include <malloc.h>
#include <sys/mman.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <charconv>
#include <chrono>
#include <cstring>

#include "stages.h"
#include "id_order.h"

#include <fstream>
#include <map>
#include <sstream>
#include <thread>
#include <memory>

using namespace sina;

namespace {
thread_local std::string g_err;
int fail(const std::exception &e) {
    g_err = e.what();
    return 1;
}

struct result {
    // (between runs a result keeps its heap blocks: fresh memory per query is a page fault per query)
    void reset() {
        status = 2;
        head = tail = qual = 0;
        width = 0;
        ab = nullptr;
        n_ab = 0;
        own.clear();
        log.clear();
        note.set = false;
        family.clear();
        family_items.clear();
        family_render = nullptr;
        idty = -1.f;
        searched = false;
        sr_ids.clear();
        sr_scores.clear();
        attrs.clear();
    }
    int status = 2;  // 0 aligned by DP, 1 alignment copied from a reference, 2 not aligned
    int head = 0, tail = 0, qual = 0;
    uint32_t width = 0;
    const aligned_base *ab = nullptr;  // the aligned bases: own.data()
    uint32_t n_ab = 0;
    // The aligned bases, TAKEN from the tray's aligned sequence (the two exchange their base lists: the sequence
    // goes back to its cache with this record's previous block, so neither allocates in steady state).  Until
    // round 5 the sink copied them into an arena of the pipeline's -- the fifth time a query's 6 KB went through
    // a core; what a real sink does with them (format, write) it does from the sequence itself.
    base_vector own;
    mutable std::string log;          // (the tray's log; its score line is put in when the log is asked for)
    mutable tray::score_note note;
    // align_family_slv: as text, or as the list famfinder left (rendered by sina_host_result_family; the store is
    // kept alive by the pipeline's stages)
    mutable std::string family;
    std::vector<uint64_t> family_items;
    const void *family_owner = nullptr;
    void (*family_render)(const void *, const uint64_t *, size_t, std::string &) = nullptr;
    float idty = -1.f;              // align_ident_slv (--calc-idty), -1 if not computed
    bool searched = false;          // search stage ran and produced a result vector
    std::vector<uint32_t> sr_ids;   // search results, best first
    std::vector<float> sr_scores;
    // turn, nearest_slv, lca_*, copy_*: a handful per query.  A flat list whose strings keep their blocks from
    // one query to the next (a std::map was two heap blocks per attribute and query, freed by reset())
    struct attr_list {
        std::vector<std::pair<std::string, std::string>> v;
        size_t n = 0;
        void clear() { n = 0; }
        std::string &operator[](std::string_view key) {
            for (size_t i = 0; i < n; i++)
                if (v[i].first == key) return v[i].second;
            if (n == v.size()) v.emplace_back();
            v[n].first.assign(key);
            v[n].second.clear();
            return v[n++].second;
        }
        const std::string *find(std::string_view key) const {
            for (size_t i = 0; i < n; i++)
                if (v[i].first == key) return &v[i].second;
            return nullptr;
        }
    } attrs;
};

struct pipeline {
    famfinder ff;
    aligner al;
    pipeline() { al.defer_score_line = true; }  // (this driver's sink reads a tray's log through log_text())
    std::unique_ptr<search_filter> sf;  // only with sina_host_pipeline_create_search()
    std::shared_ptr<reference_store> search_store;
    double sf_s = 0;
    // The results of a run, batch by batch: chunk b holds the results of queries [b * chunk_q, (b + 1) * chunk_q)
    // (each with its aligned bases).  A chunk is sized and reset by the sink when its batch arrives
    // (and kept between runs) -- sizing everything before the first batch was 17 ms of every run's start.
    struct result_chunk {
        std::vector<result> results;
        uint64_t run = 0;  // the run (pipeline::run_no) whose sink filled it: an older one's results are not this run's
    };
    uint64_t run_no = 0;
    std::vector<result_chunk> chunks;
    uint32_t chunk_q = 1, n_results = 0;
    result &result_of(uint32_t q) { return chunks[q / chunk_q].results[q % chunk_q]; }
    double wall_s = 0, ff_s = 0, al_s = 0;
    // A batch's trays (6144 objects with a stringstream each) travel from the source to the sink and come
    // back here, emptied, for the next batch -- and the next run: constructing and destroying them per batch
    // was 1.5 us per query on the two driver threads that did it, and 6 ms before a run's first GPU call.
    std::mutex spare_mu;
    std::vector<std::vector<tray>> spare_trays;
    std::vector<tray> take_trays() {
        std::lock_guard<std::mutex> lk(spare_mu);
        if (spare_trays.empty()) return {};
        std::vector<tray> v = std::move(spare_trays.back());
        spare_trays.pop_back();
        return v;
    }
    void give_trays(std::vector<tray> &&v) {
        std::lock_guard<std::mutex> lk(spare_mu);
        if (spare_trays.size() < 16) spare_trays.push_back(std::move(v));
    }
};
// The sink's work for one tray (what SINA's writer stage does per sequence): log, family, attributes and the
// aligned bases go into the result record (a DP alignment has exactly the query's bases, a copied one its
// reference's: never more than the query, which the copy short-cut requires to be contained in it) -- and the
// tray gives its objects back.
void extract_tray(pipeline *p, tray &t, result &r, uint32_t query_bases) {
    r.reset();
    uint64_t tk = host_tsc();
    r.log.assign(t.log.view());  // (no temporary: the result's string keeps its block between runs)
    r.note = t.pending_score;
    if (const cseq::lazy_text *lz = t.input_sequence->lazy_attr(fn::family)) {  // (the list; the text when it is asked for)
        r.family_items.assign(lz->items.begin(), lz->items.end());
        r.family_owner = lz->owner;
        r.family_render = lz->render;
    } else if (const std::string *fam = t.input_sequence->string_attr(fn::family)) r.family.assign(*fam);
    else r.family = t.input_sequence->get_attr<std::string>(fn::family);
    if (const std::string *turn = t.input_sequence->string_attr(fn::turn)) r.attrs[fn::turn].assign(*turn);
    else if (t.input_sequence->has_attr(fn::turn)) r.attrs[fn::turn] = t.input_sequence->get_attr<std::string>(fn::turn);
    tk = host_tick("extract: log + family + turn", tk);
    if (t.aligned_sequence) {
        cseq &c = *t.aligned_sequence;
        r.width = c.getWidth();
        r.status = (r.log.find("copied alignment from") != std::string::npos) ? 1 : 0;
        // one walk over the attributes, the stages' own keys told by their interned addresses
        struct keys_t {
            const std::string *qual = cseq::attr_key(fn::qual), *head = cseq::attr_key(fn::head), *tail = cseq::attr_key(fn::tail),
                              *idty = cseq::attr_key(fn::idty), *nearest = cseq::attr_key(search_filter::fn_nearest);
        };
        static const keys_t keys;
        r.idty = -1.f;
        for (const auto &kv : c.attrs_unsettled()) {
            const int *iv = std::get_if<int>(&kv.second);
            if (kv.name == keys.qual) r.qual = iv ? *iv : c.get_attr<int>(fn::qual);
            else if (kv.name == keys.head) r.head = iv ? *iv : c.get_attr<int>(fn::head);
            else if (kv.name == keys.tail) r.tail = iv ? *iv : c.get_attr<int>(fn::tail);
            else if (kv.name == keys.idty) r.idty = c.get_attr<float>(fn::idty);
            else {
                const std::string &k = kv.key();
                if (kv.name == keys.nearest || k.compare(0, 4, "lca_") == 0 || k.compare(0, 5, "copy_") == 0)
                    r.attrs[k] = c.get_attr<std::string>(k);
            }
        }
        (void)query_bases;
        r.own.swap(c.mutableAlignedBases());
        r.n_ab = (uint32_t)r.own.size();
        r.ab = r.own.data();
    }
    if (t.search_result) {
        r.searched = true;
        for (const auto &sr : *t.search_result) {
            r.sr_ids.push_back(p->search_store->id_of(sr.sequence));
            r.sr_scores.push_back(sr.score);
        }
    }
    tk = host_tick("extract: aligned attrs + bases", tk);
    t.destroy();
    host_tick("extract: tray.destroy", tk);
}

// An unaligned query as the reader stage would hand it over: base i in column i.
void fill_query_tray(tray &t, uint32_t q, const uint8_t *qmask, const uint64_t *qoff) {
    uint64_t tk = host_tsc();
    t.seqno = q;
    // (trays are reused from batch to batch: a fresh log -- destroy() frees the sequences but leaves the
    // text, and the previous occupant's would lead this one's)
    t.clear_log();
    t.input_sequence = object_cache<cseq, cache_query_seq>::take();
    {
        char name[24] = "query";
        char *e = std::to_chars(name + 5, name + sizeof name, q).ptr;
        t.input_sequence->setName(std::string_view(name, (size_t)(e - name)));
    }
    tk = host_tick("build: log reset + new cseq", tk);
    // (base i in column i: the sequence keeps the mask bytes, cseq.h "dense" -- the packed words are made if a
    // stage asks for them)
    const uint32_t n_bases = (uint32_t)(qoff[q + 1] - qoff[q]);
    t.input_sequence->setDenseMasks(qmask + qoff[q], n_bases);
    host_tick("build: bases", tk);
}
}  // namespace

extern "C" {
static void tune_allocator();

const char *sina_host_last_error(void) { return g_err.c_str(); }

int sina_host_store_from_packed(const char *key, const uint32_t *ab, const uint64_t *off, uint32_t n, uint32_t width,
                                int device) {
    try {
        auto s = reference_store::from_packed(key, ab, off, n, width);
        s->set_device(device);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// A store of a rank other than 0: its device buffers are allocated like rank 0's
// (sina_hip_store_alloc_like) and filled by the start-up broadcast instead of an upload of its own.
int sina_host_store_expect_broadcast(const char *key) {
    try {
        reference_store::get(key)->expect_broadcast();
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int sina_host_store_close(const char *key) {
    try {
        for (int k = 1; k <= 12; k++)
            for (int nf = 0; nf < 2; nf++) kmer_search::release_kmer_search(key, k, nf != 0);
        reference_store::close(key);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int sina_host_add_filter(const char *key, const char *name, const float *weights, uint32_t n) {
    try {
        reference_store::get(key)->getAlignmentStats().emplace_back(std::string(name),
                                                                    std::vector<float>(weights, weights + n));
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// Device context of a store (creates it and uploads the references on first use): lets a
// launcher reach sina_hip_store_view_get / sina_hip_store_alloc_like for the start-up
// broadcast of the index over RCCL.
void *sina_host_store_ctx(const char *key) {
    try {
        return reference_store::get(key)->device();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
int sina_host_store_build_index(const char *key, unsigned k, int nofast) {
    try {
        reference_store::get(key)->ensure_index(k, nofast != 0);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}
// sina -i in.fasta -o out.fasta --db <key> [--search] [--show-dist]: FASTA reader -> famfinder -> aligner
// [-> search_filter] -> log printer -> FASTA writer, in batches of `batch` sequences.  The stage
// options are whatever sina_host_set_option set.  summary: n_read, n_aligned, n_written, n_skipped,
// then avg_sps, avg_cpm, avg_idty (as Log::printer prints them at exit).
//
// sina_host_run_fasta_serial: one batch at a time, stage after stage, tray after tray (rounds 1-3: parity runs).
// sina_host_run_fasta: the same stages as CONCURRENT nodes, as SINA's flow graph has them (src/sina.cpp:452-586:
// a serial source, the stages, serial ordered sinks): a reader thread cuts and parses the next batch while the
// batch before it is searched (famfinder thread) and the one before that aligned (aligner thread); the sink
// takes the batches in input order, renders the trays' log reports and composes their FASTA records on the loop
// pool, and only the hand-over of the finished text -- running totals, csv rows, the write itself -- stays
// serial.  Output, log and summary are byte for byte those of the serial driver.
namespace {
struct fasta_run {
    famfinder ff;
    aligner al;
    std::unique_ptr<search_filter> sf;
    std::unique_ptr<rw_fasta::reader> rd;
    std::unique_ptr<rw_fasta::writer> wr;
    std::unique_ptr<log_printer> lp;
    std::ofstream logf;
    std::ostringstream devnull;
    bool log_to_file = false;
    int n_read = 0, n_aligned = 0;
    std::ostream &log() { return log_to_file ? static_cast<std::ostream &>(logf) : devnull; }
    fasta_run(const char *in_path, const char *out_path, const char *log_path, int do_search, int show_dist) {
        famfinder::validate_options();
        aligner::validate_options();
        if (do_search) search_filter::validate_options();
        if (do_search) sf.reset(new search_filter());
        rd.reset(new rw_fasta::reader(in_path));
        wr.reset(new rw_fasta::writer(out_path));
        lp.reset(new log_printer(show_dist != 0));
        if (log_path && *log_path) {
            logf.open(log_path);
            log_to_file = logf.is_open();
        }
    }
    bool read_batch(std::vector<tray> &trays, uint32_t batch) {
        trays.clear();
        while (trays.size() < batch) {
            tray t;
            if (!(*rd)(t)) break;
            trays.push_back(t);
        }
        return !trays.empty();
    }
    void finish(double *summary7) {
        wr->flush();
        const log_printer::summary sm = lp->totals();
        summary7[0] = n_read;
        summary7[1] = n_aligned;
        summary7[2] = wr->written();
        summary7[3] = rd->skipped();
        summary7[4] = sm.avg_sps;
        summary7[5] = sm.avg_cpm;
        summary7[6] = sm.avg_idty;
    }
};
}  // namespace

int sina_host_run_fasta_serial(const char *in_path, const char *out_path, const char *log_path, int do_search,
                               int show_dist, uint32_t batch, double *summary7) {
    try {
        fasta_run r(in_path, out_path, log_path, do_search, show_dist);
        if (batch == 0) batch = 1024;
        std::vector<tray> trays;
        while (r.read_batch(trays, batch)) {
            r.n_read += (int)trays.size();
            r.ff(trays);
            r.al(trays);
            if (r.sf) (*r.sf)(trays);
            for (auto &t : trays) {  // serial, in input order (Log::printer and the writer are serial nodes)
                if (t.aligned_sequence) r.n_aligned++;
                t = (*r.lp)(t, r.log());
                t = (*r.wr)(t);
                t.destroy();
                r.devnull.str("");
            }
        }
        r.finish(summary7);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int sina_host_run_fasta(const char *in_path, const char *out_path, const char *log_path, int do_search,
                        int show_dist, uint32_t batch, double *summary7) {
    try {
        fasta_run r(in_path, out_path, log_path, do_search, show_dist);
        if (batch == 0) batch = 1024;
        tune_allocator();
        try {  // the contexts the node threads below will lease, made and warmed before any of them starts
            const auto store = reference_store::get(aligner::opts->database);
            const auto sstore = do_search ? reference_store::get(search_filter_database()) : nullptr;
            store->reserve_workers(reference_store::dev_search, 1 + (sstore == store ? 1 : 0));
            store->reserve_workers(reference_store::dev_align, 1);
            if (sstore) {
                if (sstore != store) sstore->reserve_workers(reference_store::dev_search, 1);
                sstore->reserve_workers(reference_store::dev_compare, 1);
            }
        } catch (const std::exception &) {
            // (a store that cannot be opened yet is reported by the stage that needs it)
        }
        // bounded hand-over between the nodes; batches stay in input order (one thread per node: FIFO)
        struct chan {
            std::mutex mu;
            std::condition_variable cv;
            std::deque<std::vector<tray>> q;
            bool closed = false, abort = false;
            void push(std::vector<tray> &&b) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return q.size() < 2 || abort; });
                if (abort) {
                    for (auto &t : b) t.destroy();
                    return;
                }
                q.push_back(std::move(b));
                cv.notify_all();
            }
            bool pop(std::vector<tray> &b) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty() || closed || abort; });
                if (abort || q.empty()) return false;
                b = std::move(q.front());
                q.pop_front();
                cv.notify_all();
                return true;
            }
            void close() {
                std::lock_guard<std::mutex> lk(mu);
                closed = true;
                cv.notify_all();
            }
            void stop() {
                std::lock_guard<std::mutex> lk(mu);
                abort = true;
                for (auto &b : q)
                    for (auto &t : b) t.destroy();
                q.clear();
                cv.notify_all();
            }
        };
        chan read_q, found_q, aligned_q;
        std::exception_ptr err;
        std::mutex err_mu;
        auto on_error = [&] {
            { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
            read_q.stop();
            found_q.stop();
            aligned_q.stop();
        };
        std::thread reader([&] {
            try {
                std::vector<tray> trays;
                while (r.read_batch(trays, batch)) {
                    r.n_read += (int)trays.size();
                    read_q.push(std::move(trays));
                    trays = std::vector<tray>();
                }
            } catch (...) { on_error(); }
            read_q.close();
        });
        std::thread finder([&] {
            try {
                std::vector<tray> trays;
                while (read_q.pop(trays)) {
                    r.ff(trays);
                    found_q.push(std::move(trays));
                    trays = std::vector<tray>();
                }
            } catch (...) { on_error(); }
            found_q.close();
        });
        std::thread align([&] {
            try {
                std::vector<tray> trays;
                while (found_q.pop(trays)) {
                    r.al(trays);
                    if (r.sf) (*r.sf)(trays);
                    aligned_q.push(std::move(trays));
                    trays = std::vector<tray>();
                }
            } catch (...) { on_error(); }
            aligned_q.close();
        });
        try {  // the sink: this thread
            std::vector<tray> trays;
            std::vector<log_printer::report> reports;
            while (aligned_q.pop(trays)) {
                reports.resize(trays.size());
                parallel_for(trays.size(), [&](size_t i) { r.lp->render(trays[i], reports[i]); });
                r.wr->precompose(trays);  // (after render: the reports' attributes are part of the records' meta data)
                for (size_t i = 0; i < trays.size(); i++) {
                    tray &t = trays[i];
                    if (t.aligned_sequence) r.n_aligned++;
                    r.lp->commit(reports[i], r.log());
                    t = (*r.wr)(t);
                    r.devnull.str("");
                }
                parallel_for(trays.size(), [&](size_t i) { trays[i].destroy(); });
            }
        } catch (...) { on_error(); }
        reader.join();
        finder.join();
        align.join();
        if (err) std::rethrow_exception(err);
        r.finish(summary7);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// FASTA reader + writer alone (no GPU): reads every sequence of in_path and writes it back, the
// input sequence standing in for the aligned one; returns the counts.  For CPU tests of the I/O rules.
int sina_host_fasta_roundtrip(const char *in_path, const char *out_path, int *n_read, int *n_skipped) {
    try {
        rw_fasta::reader rd(in_path);
        rw_fasta::writer wr(out_path);
        int n = 0;
        for (;;) {
            tray t;
            if (!rd(t)) break;
            n++;
            t.aligned_sequence = new cseq(*t.input_sequence);
            t = wr(t);
            t.destroy();
        }
        wr.flush();
        *n_read = n;
        *n_skipped = rd.skipped();
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// .sidx <-> CSR without a GPU (tests): load into caller buffers / store from caller arrays
int sina_host_sidx_load(const char *path, unsigned k, int nofast, uint32_t *n_sequences, uint32_t *offsets,
                        uint32_t *ids, uint64_t ids_cap, uint64_t *n_ids) {
    try {
        std::vector<std::string> names;
        std::vector<uint32_t> off, id;
        std::string why;
        if (!sidx_load(path, k, nofast != 0, &names, &off, &id, &why)) throw std::runtime_error("sidx_load: " + why);
        *n_sequences = (uint32_t)names.size();
        *n_ids = id.size();
        if (id.size() > ids_cap) throw std::runtime_error("sidx_load: ids buffer too small");
        memcpy(offsets, off.data(), 4 * off.size());
        if (!id.empty()) memcpy(ids, id.data(), 4 * id.size());
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}
int sina_host_sidx_store(const char *path, unsigned k, int nofast, uint32_t n_sequences, const uint32_t *offsets,
                         const uint32_t *ids, uint64_t n_ids) {
    try {
        std::vector<std::string> names;
        for (uint32_t i = 0; i < n_sequences; i++) names.push_back("ref" + std::to_string(i));
        sidx_store(path, k, nofast != 0, names, std::vector<uint32_t>(offsets, offsets + ((size_t)1 << (2 * k)) + 1),
                   std::vector<uint32_t>(ids, ids + n_ids));
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}
// opens an aligned-FASTA database by path (its index is cached in <path minus extension>.sidx)
int sina_host_store_open(const char *path, int device) {
    try {
        auto s = reference_store::get(path);
        s->set_device(device);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}
// the same with the sequences numbered as the reference numbers an ARB database's (host/id_order.h)
int sina_host_store_open_arb_order(const char *path, int device) {
    try {
        auto s = reference_store::open(path, true);
        s->set_device(device);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}
// order[i] = database position of the sequence the reference gives id i (query_arb.cpp:160,470-474,732-739);
// hashes[j] (optional) = boost::hash<std::string> of names[j]; returns the number of ids (distinct names)
uint32_t sina_host_reference_order(const char *const *names, uint32_t n, uint32_t *order, uint64_t *hashes,
                                   uint64_t *bucket_count) {
    std::vector<std::string> v(names, names + n);
    std::size_t buckets = 0;
    const std::vector<uint32_t> o = arb_name_order(v, &buckets);
    std::copy(o.begin(), o.end(), order);
    if (hashes)
        for (uint32_t i = 0; i < n; i++) hashes[i] = (uint64_t)arb_name_hash(v[i]);
    if (bucket_count) *bucket_count = buckets;
    return (uint32_t)o.size();
}
// name of reference `id` of a store (empty string: no such id)
const char *sina_host_store_name(const char *key, uint32_t id) {
    static thread_local std::string s;
    try {
        auto st = reference_store::get(key);
        s = id < st->size() ? st->getCseq(id).getName() : std::string();
    } catch (const std::exception &) {
        s.clear();
    }
    return s.c_str();
}
const char *sina_host_store_index_origin(const char *key) {
    static thread_local std::string s;
    try {
        s = reference_store::get(key)->index_origin();
    } catch (const std::exception &e) {
        fail(e);
        s.clear();
    }
    return s.c_str();
}
// After the index arrived by broadcast (sina_hip_store_alloc_like + RCCL): tell the store
// not to rebuild it.
int sina_host_store_index_ready(const char *key, unsigned k, int nofast) {
    try {
        reference_store::get(key)->adopt_index(k, nofast != 0);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int sina_host_reset_options(void) {
    famfinder::reset_options();
    aligner::reset_options();
    search_filter::reset_options();
    rw_fasta::reset_options();
    return 0;
}

// field of a reference sequence (stands in for the ARB database fields the search stage loads)
int sina_host_store_set_attr(const char *key, uint32_t id, const char *field, const char *value) {
    try {
        reference_store::get(key)->set_attr(id, field, value);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

int sina_host_set_option(const char *stage, const char *name, const char *value) {
    try {
        if (!strcmp(stage, "famfinder")) famfinder::set_option(name, value);
        else if (!strcmp(stage, "aligner")) aligner::set_option(name, value);
        else if (!strcmp(stage, "search")) search_filter::set_option(name, value);
        else if (!strcmp(stage, "fasta")) rw_fasta::set_option(name, value);
        else if (!strcmp(stage, "host") && !strcmp(name, "threads")) set_host_threads((unsigned)atoi(value));
        else if (!strcmp(stage, "host") && !strcmp(name, "dedup")) set_batch_dedup(atoi(value) != 0);
        else throw std::logic_error(std::string("unknown stage ") + stage);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// The driver allocates and frees tens of MB of trays per batch: keep what the allocator got from the
// kernel (no trimming, no per-vector mmap) so that steady-state batches do not page-fault their
// memory in again.  Process-wide; a host that sets SINA_HIP_NO_RUNTIME_DEFAULTS (the library's "leave my process
// alone" switch, include/sina_hip.h) keeps its allocator settings too.
static void tune_allocator() {
    static const bool once = [] {
        const char *v = getenv("SINA_HIP_NO_RUNTIME_DEFAULTS");
        if (v && *v && !(v[0] == '0' && v[1] == 0)) return false;
        mallopt(M_TRIM_THRESHOLD, 1 << 30);
        mallopt(M_TOP_PAD, 64 << 20);
        mallopt(M_MMAP_THRESHOLD, 256 << 20);
        return true;
    }();
    (void)once;
}

void *sina_host_pipeline_create(void) {
    try {
        tune_allocator();
        famfinder::validate_options();
        aligner::validate_options();
        return new pipeline();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
// the same with the search stage behind the aligner (sina --search)
void *sina_host_pipeline_create_search(void) {
    try {
        tune_allocator();
        famfinder::validate_options();
        aligner::validate_options();
        search_filter::validate_options();
        std::unique_ptr<pipeline> p(new pipeline());
        p->sf.reset(new search_filter());
        p->search_store = reference_store::get(search_filter_database());
        return p.release();
    } catch (const std::exception &e) {
        fail(e);
        return nullptr;
    }
}
void sina_host_pipeline_destroy(void *p) { delete (pipeline *)p; }

// Feeds nq unaligned queries (iupac masks) through famfinder -> aligner in batches
// of `batch`, with `inflight` batches being worked on concurrently (host stages of
// one batch overlap GPU work of another).
// marks of the bench in the SINA_HOST_TRACE file: 0 = timed region starts, 1 = ends
void sina_host_profile_mark(int which) { host_profile_mark(which == 0 ? "MARK:timed-start" : "MARK:timed-end"); }

int sina_host_pipeline_run(void *pp, const uint8_t *qmask, const uint64_t *qoff, uint32_t nq, uint32_t batch,
                           uint32_t inflight) {
    pipeline *p = (pipeline *)pp;
    try {
        host_profile_mark("MARK:run-entered");
        if (batch == 0) batch = nq ? nq : 1;
        if (p->chunk_q != batch) p->chunks.clear();  // (chunks follow the batch size)
        p->chunk_q = batch;
        p->n_results = nq;
        const size_t n_chunks = ((size_t)nq + batch - 1) / batch;
        if (p->chunks.size() < n_chunks) p->chunks.resize(n_chunks);
        // (a chunk keeps its records -- their heap blocks -- between runs, but not their meaning: only a chunk
        // the sink of THIS run has filled is served by result_at(); a batch that fails early leaves "not aligned")
        ++p->run_no;
        if (inflight == 0) inflight = 1;
        std::atomic<uint32_t> next{0};
        std::atomic<uint64_t> ff_ns{0}, al_ns{0}, sf_ns{0};
        std::exception_ptr err;
        std::mutex err_mu;
        const auto t0 = std::chrono::steady_clock::now();
        host_profile_mark("MARK:run-threads-start");
        // One batch travelling through the stages (what a tray is in SINA's TBB flow graph, times `batch`).
        struct item {
            uint32_t b0 = 0, b1 = 0;
            std::vector<tray> trays;
        };
        auto build_and_find = [&](item &it) {  // source + famfinder node
            it.trays = p->take_trays();
            it.trays.resize(it.b1 - it.b0);
            {
                host_phase hp("drv.build_trays");
                parallel_for(it.b1 - it.b0, [&](size_t i) {  // (what SINA's reader stage does per sequence)
                    fill_query_tray(it.trays[i], it.b0 + (uint32_t)i, qmask, qoff);
                });
            }
            const auto a = std::chrono::steady_clock::now();
            p->ff(it.trays);
            ff_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a)
                         .count();
        };
        auto align_and_search = [&](item &it) {  // aligner node (+ search node)
            const auto b = std::chrono::steady_clock::now();
            p->al(it.trays);
            const auto c = std::chrono::steady_clock::now();
            al_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(c - b).count();
            if (p->sf) {
                (*p->sf)(it.trays);
                sf_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                             std::chrono::steady_clock::now() - c)
                             .count();
            }
        };
        auto extract = [&](item &it) {  // sink
            host_phase hp("drv.extract");
            const uint32_t b0 = it.b0;
            std::vector<tray> &trays = it.trays;
            pipeline::result_chunk &chunk = p->chunks[b0 / batch];
            chunk.results.resize(it.b1 - it.b0);
            chunk.run = p->run_no;
            parallel_for(it.b1 - it.b0, [&](size_t i) {  // (what SINA's writer stage does per sequence)
                const uint32_t q = b0 + (uint32_t)i;
                extract_tray(p, trays[i], chunk.results[i], (uint32_t)(qoff[q + 1] - qoff[q]));
            });
        };
        auto take = [&](item &it) -> bool {
            it.b0 = next.fetch_add(batch);
            if (it.b0 >= nq) return false;
            it.b1 = std::min(nq, it.b0 + batch);
            return true;
        };
        auto guarded = [&](auto &&body) {
            try {
                body();
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mu);
                if (!err) err = std::current_exception();
            }
        };
        if (inflight == 1) {
            // one batch at a time, stage after stage (kernel timings taken this way are undisturbed)
            guarded([&] {
                item it;
                while (take(it)) {
                    build_and_find(it);
                    align_and_search(it);
                    extract(it);
                    p->give_trays(std::move(it.trays));
                }
            });
        } else {
            // The stages as nodes with their own threads and bounded hand-over queues, as SINA wires them
            // (src/sina.cpp:452-586: source -> famfinder -> aligner -> search -> sink).  `inflight`
            // aligner threads keep the GPU's DP slot busy (graph build of one batch beside the DP of
            // another); the famfinder threads run ahead by at most two finished batches, so an aligner
            // thread never waits for a k-mer search to start.
            struct handover {
                std::mutex mu;
                std::condition_variable cv;
                std::deque<item> q;
                size_t cap = 2;
                int producers = 0;
                bool abort = false;
                void push(item &&it) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return q.size() < cap || abort; });
                    if (abort) {
                        for (auto &t : it.trays) t.destroy();
                        return;
                    }
                    q.push_back(std::move(it));
                    cv.notify_all();
                }
                bool pop(item &it) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !q.empty() || producers == 0 || abort; });
                    if (q.empty()) return false;
                    it = std::move(q.front());
                    q.pop_front();
                    cv.notify_all();
                    return true;
                }
                void producer_done() {
                    std::lock_guard<std::mutex> lk(mu);
                    if (--producers == 0) cv.notify_all();
                }
                void stop() {
                    std::lock_guard<std::mutex> lk(mu);
                    abort = true;
                    for (auto &it : q)
                        for (auto &t : it.trays) t.destroy();
                    q.clear();
                    cv.notify_all();
                }
            };
            handover found, aligned;
            // (famfinder threads; they run ahead by `found.cap` finished batches)
            const uint32_t n_find = inflight >= 4 ? 3 : (inflight >= 3 ? 2 : 1);
            const uint32_t n_align = inflight, n_sink = 1;
            try {  // the contexts the threads below will lease, made and warmed before any of them starts
                const auto store = reference_store::get(aligner::opts->database);
                store->reserve_workers(reference_store::dev_search, n_find + (p->sf && p->search_store == store ? n_align : 0));
                store->reserve_workers(reference_store::dev_align, n_align);
                if (p->sf && p->search_store) {
                    if (p->search_store != store) p->search_store->reserve_workers(reference_store::dev_search, n_align);
                    p->search_store->reserve_workers(reference_store::dev_compare, n_align);
                }
            } catch (const std::exception &) {
                // (no store registered under the aligner's name, or no room to warm: the stages will say so themselves)
            }
            found.cap = n_find;
            found.producers = (int)n_find;
            aligned.producers = (int)n_align;
            auto on_error = [&] {
                found.stop();
                aligned.stop();
                next.store(nq);
            };
            std::vector<std::thread> th;
            for (uint32_t i = 0; i < n_find; i++)
                th.emplace_back([&] {
                    struct cpu_report { ~cpu_report() { host_profile_thread_exit("thread: finder (whole run)"); } } rep;
                    try {
                        item it;
                        while (take(it)) {
                            build_and_find(it);
                            found.push(std::move(it));
                            it = item();
                        }
                    } catch (...) {
                        { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
                        on_error();
                    }
                    found.producer_done();
                });
            for (uint32_t i = 0; i < n_align; i++)
                th.emplace_back([&] {
                    struct cpu_report { ~cpu_report() { host_profile_thread_exit("thread: aligner (whole run)"); } } rep;
                    try {
                        item it;
                        while (found.pop(it)) {
                            align_and_search(it);
                            aligned.push(std::move(it));
                            it = item();
                        }
                    } catch (...) {
                        { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
                        on_error();
                    }
                    aligned.producer_done();
                });
            for (uint32_t i = 0; i < n_sink; i++)
                th.emplace_back([&] {
                    struct cpu_report { ~cpu_report() { host_profile_thread_exit("thread: sink (whole run)"); } } rep;
                    try {
                        item it;
                        while (aligned.pop(it)) {
                            extract(it);
                            p->give_trays(std::move(it.trays));
                            it = item();
                        }
                    } catch (...) {
                        { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
                        on_error();
                    }
                });
            for (auto &t : th) t.join();
        }
        p->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        p->ff_s = ff_ns.load() * 1e-9;
        p->al_s = al_ns.load() * 1e-9;
        p->sf_s = sf_ns.load() * 1e-9;
        if (err) std::rethrow_exception(err);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// The boundary as INTEGRATION.md binds it (src/sina.cpp:497-519): `n_threads` callers -- TBB workers of
// unlimited-concurrency function_nodes in SINA -- each push ONE tray at a time through
// batched<famfinder> -> batched<aligner> (-> batched<search_filter> for a search pipeline), and the shim
// groups whoever is waiting into one GPU batch.  Results land where sina_host_pipeline_run puts them.
// poison_q >= 0: that query's family gets a member that is not of the store before the aligner sees it -- the
// device rejects the batch it travels in, the stage throws (std::runtime_error) and every
// caller of that batch gets the exception: failed[q] = 1 for those, the first message in err / err_cap.
int sina_host_pipeline_run_single_trays(void *pp, const uint8_t *qmask, const uint64_t *qoff, uint32_t nq,
                                        uint32_t n_threads, uint32_t max_batch, uint32_t linger_us, int32_t poison_q,
                                        uint8_t *failed, char *err, uint32_t err_cap) {
    pipeline *p = (pipeline *)pp;
    try {
        if (n_threads == 0) n_threads = 1;
        ++p->run_no;
        p->chunk_q = nq ? nq : 1;
        p->n_results = nq;
        p->chunks.clear();
        p->chunks.resize(1);
        pipeline::result_chunk &chunk = p->chunks[0];
        chunk.results.resize(nq);
        chunk.run = p->run_no;
        if (failed) memset(failed, 0, nq);
        if (err && err_cap) err[0] = 0;
        batched<famfinder> bff(p->ff, max_batch, linger_us);
        aligner whole_log = p->al;  // (the callers of the shim read their trays' logs themselves)
        whole_log.defer_score_line = false;
        batched<aligner> bal(whole_log, max_batch, linger_us);
        std::unique_ptr<batched<search_filter>> bsf;
        if (p->sf) bsf.reset(new batched<search_filter>(*p->sf, max_batch, linger_us));
        // (a sequence of the store's width that is not one of the store's: what a stale tray would carry)
        cseq foreign("not-in-this-store");
        if (poison_q >= 0) {
            const auto store = reference_store::get(aligner::opts->database);
            foreign.append("ACGU");
            foreign.setWidth(store->getAlignmentWidth());
        }
        std::atomic<uint32_t> next{0};
        std::mutex err_mu;
        std::exception_ptr fatal;
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (uint32_t w = 0; w < n_threads; w++)
            th.emplace_back([&] {
                for (;;) {
                    const uint32_t q = next.fetch_add(1);
                    if (q >= nq) return;
                    try {
                        tray t;
                        fill_query_tray(t, q, qmask, qoff);
                        t = bff(std::move(t));
                        if ((int32_t)q == poison_q && t.alignment_reference && !t.alignment_reference->empty())
                            t.alignment_reference->back().sequence = &foreign;
                        tray keep = t;  // (the pointers: a stage that throws hands nothing back)
                        try {
                            t = bal(std::move(t));
                            if (bsf) t = (*bsf)(std::move(t));
                        } catch (const std::exception &e) {
                            if (failed) failed[q] = 1;
                            {
                                std::lock_guard<std::mutex> lk(err_mu);
                                if (err && err_cap && err[0] == 0) snprintf(err, err_cap, "%s", e.what());
                            }
                            keep.destroy();
                            continue;
                        }
                        extract_tray(p, t, chunk.results[q], (uint32_t)(qoff[q + 1] - qoff[q]));
                    } catch (...) {
                        std::lock_guard<std::mutex> lk(err_mu);
                        if (!fatal) fatal = std::current_exception();
                        next.store(nq);
                        return;
                    }
                }
            });
        for (auto &t : th) t.join();
        p->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (fatal) std::rethrow_exception(fatal);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

namespace {
const result &result_at(void *pp, uint32_t q) {
    static const result none;  // (out of range, or a batch that never reached the sink: "not aligned")
    pipeline *p = (pipeline *)pp;
    if (q >= p->n_results) return none;
    const auto &chunk = p->chunks[q / p->chunk_q];
    if (chunk.run != p->run_no) return none;
    return (q % p->chunk_q) < chunk.results.size() ? chunk.results[q % p->chunk_q] : none;
}
}  // namespace

int sina_host_result(void *pp, uint32_t q, int *status, int *head, int *tail, int *qual, uint32_t *width,
                     uint32_t *n_bases) {
    pipeline *p = (pipeline *)pp;
    if (q >= p->n_results) return 1;
    const result &r = result_at(pp, q);
    *status = r.status;
    *head = r.head;
    *tail = r.tail;
    *qual = r.qual;
    *width = r.width;
    *n_bases = r.n_ab;
    return 0;
}
const uint32_t *sina_host_result_bases(void *pp, uint32_t q) {
    return reinterpret_cast<const uint32_t *>(result_at(pp, q).ab);
}
const char *sina_host_result_log(void *pp, uint32_t q) {
    const result &r = result_at(pp, q);
    if (r.note.set) {  // (rendered once, in place)
        r.note.merge_into(r.log);
        r.note.set = false;
    }
    return r.log.c_str();
}
const char *sina_host_result_family(void *pp, uint32_t q) {
    const result &r = result_at(pp, q);
    if (r.family_render && r.family.empty()) r.family_render(r.family_owner, r.family_items.data(), r.family_items.size(), r.family);
    return r.family.c_str();
}
// search stage: number of results (-1: stage did not run for this query), ids/scores best first,
// string attributes it set on the sequence (nearest_slv, lca_<field>, copy_<acc>_<field>; "" if absent)
int sina_host_result_search(void *pp, uint32_t q, uint32_t *ids, float *scores, uint32_t cap) {
    const result &r = result_at(pp, q);
    if (!r.searched) return -1;
    for (size_t i = 0; i < r.sr_ids.size() && i < cap; i++) {
        ids[i] = r.sr_ids[i];
        scores[i] = r.sr_scores[i];
    }
    return (int)r.sr_ids.size();
}
const char *sina_host_result_attr(void *pp, uint32_t q, const char *name) {
    const result &r = result_at(pp, q);
    const std::string *v = r.attrs.find(name);
    return v ? v->c_str() : "";
}
double sina_host_search_seconds(void *pp) { return ((pipeline *)pp)->sf_s; }
float sina_host_result_idty(void *pp, uint32_t q) { return result_at(pp, q).idty; }
void sina_host_timings(void *pp, double *wall_s, double *famfinder_s, double *aligner_s) {
    pipeline *p = (pipeline *)pp;
    *wall_s = p->wall_s;
    *famfinder_s = p->ff_s;
    *aligner_s = p->al_s;
}

// per-phase host timings (SINA_HOST_PROFILE=1); returned string is valid until the next call
const char *sina_host_profile(int reset) {
    static thread_local std::string s;
    s = host_profile_dump(reset != 0);
    return s.c_str();
}

// host DAG build exposed for tests (compares with the oracle's mseq)
int sina_host_build_graph(const char *key, const uint32_t *ids, uint32_t F, float fs_weight, uint32_t *n_nodes,
                          uint32_t *n_edges, uint32_t *pos, uint8_t *mask, float *weight, uint32_t *pred_off,
                          uint32_t *pred, uint32_t *succ_minpos, uint32_t cap_nodes, uint32_t cap_edges) {
    try {
        auto st = reference_store::get(key);
        std::vector<const cseq *> fam;
        for (uint32_t i = 0; i < F; i++) fam.push_back(&st->getCseq(ids[i]));
        host_graph g;
        build_family_graph(fam, fs_weight, &g);
        *n_nodes = (uint32_t)g.pos.size();
        *n_edges = (uint32_t)g.pred.size();
        if (g.pos.size() > cap_nodes || g.pred.size() > cap_edges) throw std::runtime_error("graph buffers too small");
        std::copy(g.pos.begin(), g.pos.end(), pos);
        std::copy(g.mask.begin(), g.mask.end(), mask);
        std::copy(g.weight.begin(), g.weight.end(), weight);
        std::copy(g.pred_off.begin(), g.pred_off.end(), pred_off);
        std::copy(g.pred.begin(), g.pred.end(), pred);
        std::copy(g.succ_minpos.begin(), g.succ_minpos.end(), succ_minpos);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// the family as a profile (--fs-no-graph): columns and the match-term table the aligner hands to the device
// (build_family_profile, stages.cpp) -- for CPU-only tests against the oracle's pseq restatement
int sina_host_build_profile(const char *key, const uint32_t *ids, uint32_t F, float match, float mismatch, float gap,
                            float gap_ext, uint32_t *n_nodes, uint32_t *pos, float *score16, float *self16,
                            uint32_t cap_nodes) {
    try {
        auto st = reference_store::get(key);
        std::vector<const cseq *> fam;
        for (uint32_t i = 0; i < F; i++) fam.push_back(&st->getCseq(ids[i]));
        host_graph g;
        build_family_profile(fam, match, mismatch, gap, gap_ext, &g);
        *n_nodes = (uint32_t)g.pos.size();
        if (g.pos.size() > cap_nodes) throw std::runtime_error("profile buffers too small");
        std::copy(g.pos.begin(), g.pos.end(), pos);
        std::copy(g.score16.begin(), g.score16.end(), score16);
        profile_self_scores(match, mismatch, gap, gap_ext, self16);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// cseq_comparator on two aligned strings, for CPU-only tests against the reference's KAT table
// (src/unit_tests/cseq_comparator_test.cpp); also returns the six counters
int sina_host_compare(const char *a_aligned, const char *b_aligned, int iupac, int dist, int cover, int filter_lc,
                      float *score, int32_t *counts6) {
    try {
        cseq a("", a_aligned), b("", b_aligned);
        cseq_comparator cmp((CMP_IUPAC_TYPE)iupac, (CMP_DIST_TYPE)dist, (CMP_COVER_TYPE)cover, filter_lc != 0);
        sina_hip_match_counts m;
        cseq_comparator::counts(a, b, (CMP_IUPAC_TYPE)iupac, filter_lc != 0, &m);
        *score = cmp.score(m);
        const int32_t v[6] = {m.only_a_overhang, m.only_b_overhang, m.only_a, m.only_b, m.match, m.mismatch};
        memcpy(counts6, v, sizeof v);
        return 0;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// Self-checks of the host containers' round-6 machinery, for the CPU suite (no GPU, no store): a "dense" sequence
// (mask bytes only) against the same sequence built base by base, under every operation that must make its packed
// words first; a lazily rendered attribute against the eagerly set one, through copies; base lists out of the block
// pool across threads.  Returns 0, or 1 with the first failed check in err.
int sina_host_selftest(char *err, uint32_t err_cap) {
    auto fail = [&](const char *what) {
        if (err && err_cap) snprintf(err, err_cap, "%s", what);
        return 1;
    };
    try {
        // ---- dense == packed
        const char *bases = "ACGUNRYacguKMSWBDHV";
        const size_t n = strlen(bases);
        std::vector<uint8_t> masks;
        cseq eager("x");
        for (size_t i = 0; i < n; i++) {
            masks.push_back(base_iupac::from_char((unsigned char)bases[i]));
            eager.append(aligned_base((uint32_t)i, (unsigned char)bases[i]));
        }
        eager.setWidth((uint32_t)n);
        auto fresh = [&] {
            cseq d("x");
            d.setDenseMasks(masks.data(), masks.size());
            return d;
        };
        {
            cseq d = fresh();
            if (d.size() != n || d.getWidth() != n || d.denseMasks() == nullptr) return fail("dense: size / width / masks");
            if (d.getBases() != eager.getBases()) return fail("dense: getBases");
            if (d.denseMasks() == nullptr) return fail("dense: getBases must not drop the masks");
            if (!(d == eager)) return fail("dense: operator==");
            if (memcmp(d.packed(), eager.packed(), 4 * n) != 0) return fail("dense: packed words");
            if (d.denseMasks() == nullptr) return fail("dense: a const read must keep the masks");
            if (d.getAligned() != eager.getAligned()) return fail("dense: getAligned");
        }
        for (int op = 0; op < 5; op++) {
            cseq d = fresh(), e = eager;
            switch (op) {
            case 0: d.setWidth(40), e.setWidth(40); break;
            case 1: d.reverse(), e.reverse(); break;
            case 2: d.complement(), e.complement(); break;
            case 3: d.upperCaseAll(), e.upperCaseAll(); break;
            default: d.append(aligned_base(30, 'A')), e.append(aligned_base(30, 'A')); break;
            }
            if (!(d == e) || d.getWidth() != e.getWidth()) return fail("dense: a changing operation differs from the packed sequence");
            if (op >= 1 && d.denseMasks() != nullptr) return fail("dense: a changed sequence still offers its old masks");
            if (op == 0 && d.denseMasks() == nullptr) return fail("dense: a wider alignment must keep the masks");
        }
        {
            cseq d = fresh(), copy(d);
            copy.complement();
            if (!(d == eager)) return fail("dense: a copy's change reached the original");
            d.clearSequence();
            if (d.size() != 0 || d.denseMasks() != nullptr) return fail("dense: clearSequence");
        }
        // ---- lazy attribute == eager attribute
        {
            auto render = [](const void *owner, const uint64_t *items, size_t cnt, std::string &out) {
                out = *static_cast<const std::string *>(owner);
                for (size_t i = 0; i < cnt; i++) out += std::to_string(items[i]) + " ";
            };
            const std::string prefix = "p:";
            cseq c("y");
            c.set_attr("aaa", 1);
            std::vector<uint64_t> &it = c.set_lazy_attr(fn::family, &prefix, +render);
            it = {3, 1, 4};
            c.set_attr("zzz", std::string("z"));
            if (c.lazy_attr(fn::family) == nullptr) return fail("lazy: not pending after set");
            cseq copy;
            copy.copy_meta(c);
            const cseq::attr_init extra[1] = {cseq::attr_init::of(cseq::attr_key(fn::qual), 77)};
            cseq merged;
            merged.copy_meta_with(c, extra, 1);
            if (copy.lazy_attr(fn::family) == nullptr || merged.lazy_attr(fn::family) == nullptr) return fail("lazy: not carried by copy_meta");
            if (c.get_attr<std::string>(fn::family) != "p:3 1 4 ") return fail("lazy: rendered text");
            if (c.lazy_attr(fn::family) != nullptr) return fail("lazy: still pending after a read");
            if (copy.get_attr<std::string>(fn::family) != "p:3 1 4 " || merged.get_attr<std::string>(fn::family) != "p:3 1 4 ")
                return fail("lazy: a copy renders differently");
            if (merged.get_attr<int>(fn::qual) != 77 || merged.get_attr<int>("aaa") != 1 || merged.get_attr<std::string>("zzz") != "z")
                return fail("copy_meta_with: merged attributes");
            std::string keys;
            for (const auto &a : merged.get_attrs()) keys += a.key() + ",";
            if (keys != std::string("aaa,") + fn::family + "," + fn::qual + ",zzz,") return fail("copy_meta_with: key order");
            c.set_attr(fn::family, std::string("plain"));
            if (c.get_attr<std::string>(fn::family) != "plain") return fail("lazy: overwritten by set_attr");
        }
        // ---- find_bases == std::string::find
        {
            uint64_t x = 88172645463325252ull;
            auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
            for (int round = 0; round < 4000; round++) {
                const size_t h = 1 + rnd() % 200, n = 1 + rnd() % 40;
                std::string hay(h, 'A'), needle;
                for (auto &ch : hay) ch = "ACGU"[rnd() % (round % 3 ? 4 : 2)];
                if (rnd() % 3 && n <= h) needle = hay.substr(rnd() % (h - n + 1), n);  // (present: anywhere, the very end included)
                else { needle.assign(n, 'A'); for (auto &ch : needle) ch = "ACGU"[rnd() % 4]; }
                if (round % 7 == 0 && n <= h) needle = hay.substr(h - n, n);
                const size_t want = hay.find(needle);
                if (find_bases(hay, needle, 0) != want || find_bases(hay, needle, 1) != want) return fail("find_bases differs from std::string::find");
            }
        }
        // ---- the block pool across threads
        {
            std::atomic<int> bad{0};
            std::vector<base_vector> handed(64);
            std::vector<std::thread> th;
            for (int t = 0; t < 4; t++)
                th.emplace_back([&, t] {
                    for (int round = 0; round < 200; round++)
                        for (int i = t; i < 64; i += 4) {
                            base_vector v;
                            v.resize((size_t)(500 + 37 * i + round));
                            for (size_t k = 0; k < v.size(); k++) v[k] = aligned_base::from_raw((uint32_t)(k * 2654435761u + (uint32_t)i));
                            handed[i].swap(v);  // (the old block is freed by this thread, the new one possibly by another)
                        }
                });
            for (auto &x : th) x.join();
            for (int i = 0; i < 64; i++)
                for (size_t k = 0; k < handed[i].size(); k++)
                    if (handed[i][k].raw != (uint32_t)(k * 2654435761u + (uint32_t)i)) bad++;
            if (bad.load()) return fail("block pool: a block was handed out twice");
        }
        return 0;
    } catch (const std::exception &e) {
        if (err && err_cap) snprintf(err, err_cap, "exception: %s", e.what());
        return 1;
    }
}

// cseq container ops exposed for CPU-only tests against the reference's cseq_test KATs.
// op: 0 none, 1 setWidth(arg), 2 reverse, 3 complement, 4 upperCaseAll
// what: 0 getAligned(nodots), 1 getAligned(dots), 2 getAligned(nodots, dna), 3 getBases
int sina_host_cseq_op(const char *aligned, int op, uint32_t arg, int what, char *out, uint32_t cap,
                      uint32_t *size, uint32_t *width) {
    try {
        cseq c("", aligned);
        switch (op) {
        case 1: c.setWidth(arg); break;
        case 2: c.reverse(); break;
        case 3: c.complement(); break;
        case 4: c.upperCaseAll(); break;
        default: break;
        }
        std::string s;
        switch (what) {
        case 0: s = c.getAligned(true, false); break;
        case 1: s = c.getAligned(false, false); break;
        case 2: s = c.getAligned(true, true); break;
        default: s = c.getBases(); break;
        }
        if (s.size() + 1 > cap) throw std::runtime_error("buffer too small");
        memcpy(out, s.c_str(), s.size() + 1);
        *size = c.size();
        *width = c.getWidth();
        return 0;
    } catch (const base_iupac::bad_character_exception &e) {
        g_err = e.what();
        return 3;
    } catch (const std::runtime_error &e) {
        g_err = e.what();
        return 2;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

// NAST fix-up exposed for CPU-only property tests against the oracle
int sina_host_fix_duplicates(uint32_t *ab, uint32_t n, uint32_t width, int lowercase, int remove, char *log,
                             uint32_t log_cap) {
    try {
        cseq c("");
        base_vector v;
        for (uint32_t i = 0; i < n; i++) v.push_back(aligned_base::from_raw(ab[i]));
        c.setAlignedBases(v);
        c.setWidth(width);
        std::stringstream ss;
        int rc = 0;
        try {
            c.fix_duplicate_positions(ss, lowercase != 0, remove != 0);
        } catch (const std::runtime_error &) {
            rc = 2;  // "no space to left and right"
        }
        memcpy(ab, c.packed(), 4 * (size_t)n);
        const std::string s = ss.str();
        if (log && log_cap) {
            strncpy(log, s.c_str(), log_cap - 1);
            log[log_cap - 1] = 0;
        }
        return rc;
    } catch (const std::exception &e) {
        return fail(e);
    }
}

}  // extern "C"
