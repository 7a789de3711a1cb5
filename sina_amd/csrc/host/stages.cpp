// Host side of the boundary: reference_store, kmer_search, famfinder, aligner.
// Behaviour follows the reference stage by stage (citations inline); the heavy
// lifting goes through the C ABI (include/sina_hip.h).  No CPU fallback: if the
// HIP library reports an error, the stage throws.
#include "stages.h"
#include <charconv>
#include <sys/resource.h>
#include <immintrin.h>
#include "id_order.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sys/stat.h>
#include <chrono>
#include <map>
#include <unordered_map>

namespace sina {

static void hip_check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + sina_hip_last_error());
}

// The big per-batch arrays of a stage call (packed queries, aligned columns coming back: tens of MB): a
// std::vector of that size is a fresh mmap every batch -- zero-filled by the kernel page by page, then
// zero-filled again by the constructor -- 46 MB and 11 000 page faults per 6144-query batch.  One grow-only,
// uninitialised block per calling thread and use instead.
template <typename T> struct batch_scratch {
    T *p = nullptr;
    size_t cap = 0;
    T *get(size_t n) {
        if (n > cap) {
            free(p);
            cap = n + n / 4 + 1024;
            p = static_cast<T *>(malloc(cap * sizeof(T)));
            if (!p) {
                cap = 0;
                throw std::bad_alloc();
            }
        }
        return p;
    }
    ~batch_scratch() { free(p); }
    batch_scratch() = default;
    batch_scratch(const batch_scratch &) = delete;
    batch_scratch &operator=(const batch_scratch &) = delete;
};

// ================================================================ phase profiler (SINA_HOST_PROFILE=1)

namespace {
struct prof_state {
    std::mutex mu;
    std::map<std::string, std::pair<double, uint64_t>> acc;
    bool on = getenv("SINA_HOST_PROFILE") != nullptr;
    // SINA_HOST_TRACE=<file>: every phase as "thread name t0 t1" (seconds), to see what the batches
    // in flight are doing while the GPU idles
    const char *trace_path = getenv("SINA_HOST_TRACE");
    struct ev {
        size_t tid;
        const char *name;
        double t0, t1;
    };
    std::vector<ev> events;
    std::chrono::steady_clock::time_point origin = std::chrono::steady_clock::now();
};
prof_state &prof() {
    static prof_state p;
    return p;
}
thread_local const char *tl_phase = "(no phase)";  // innermost phase of this thread: names its pool jobs
double thread_cpu_s() {
    if (!prof().on) return 0.0;  // (a system call: only worth it when somebody reads the numbers)
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
// user + kernel CPU seconds, kernel seconds alone and minor page faults of the calling thread so far
struct thread_use {
    double cpu = 0, sys = 0, faults = 0;
};
thread_use thread_use_now() {
    thread_use u;
    if (!prof().on) return u;
    rusage ru;
    if (getrusage(RUSAGE_THREAD, &ru) != 0) return u;
    u.sys = (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec;
    u.cpu = thread_cpu_s();
    u.faults = (double)ru.ru_minflt;
    return u;
}
void add_cpu(const char *phase, double s);
void add_use(const char *phase, const thread_use &since) {
    const thread_use now = thread_use_now();
    add_cpu(phase, now.cpu - since.cpu);
    const std::string ph(phase);
    add_cpu((ph + " (kernel mode)").c_str(), now.sys - since.sys);
    add_cpu((ph + " (k page faults)").c_str(), 1e-3 * (now.faults - since.faults));
}
void add_cpu(const char *phase, double s) {  // CPU seconds spent under a phase, by whichever thread
    prof_state &p = prof();
    if (!p.on) return;
    std::lock_guard<std::mutex> lk(p.mu);
    auto &e = p.acc[std::string(phase) + " [cpu]"];
    e.first += s;
    e.second++;
}
struct scoped_phase {
    const char *name;
    const char *outer;
    thread_use use0;
    std::chrono::steady_clock::time_point t0;
    explicit scoped_phase(const char *n) : name(n), outer(tl_phase), use0(thread_use_now()), t0(std::chrono::steady_clock::now()) {
        tl_phase = n;
    }
    ~scoped_phase() {
        tl_phase = outer;
        prof_state &p = prof();
        if (p.on) add_use(name, use0);
        if (!p.on && !p.trace_path) return;
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        std::lock_guard<std::mutex> lk(p.mu);
        if (p.trace_path)
            p.events.push_back({std::hash<std::thread::id>()(std::this_thread::get_id()) % 9973, name,
                                std::chrono::duration<double>(t0 - p.origin).count(),
                                std::chrono::duration<double>(t1 - p.origin).count()});
        if (!p.on) return;
        auto &e = p.acc[name];
        e.first += s;
        e.second++;
    }
};
}  // namespace

void host_profile_add_cpu(const char *what, double seconds) { add_cpu(what, seconds); }
void host_profile_mark(const char *what) {  // a zero-length event in the SINA_HOST_TRACE file
    prof_state &p = prof();
    if (!p.trace_path) return;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - p.origin).count();
    std::lock_guard<std::mutex> lk(p.mu);
    p.events.push_back({0, what, t, t});
}

namespace {
struct tick_slot {
    std::atomic<const char *> name{nullptr};
    std::atomic<uint64_t> cycles{0}, samples{0};
};
tick_slot g_ticks[64];
}  // namespace
uint64_t host_tsc() { return prof().on ? __builtin_ia32_rdtsc() : 0; }
uint64_t host_tick(const char *what, uint64_t since) {
    if (!prof().on) return 0;
    const uint64_t now = __builtin_ia32_rdtsc();
    // (slots are keyed by the literal's address: a short probe sequence, names are few)
    size_t h = (reinterpret_cast<uintptr_t>(what) >> 3) % 64;
    for (int probe = 0; probe < 64; probe++, h = (h + 1) % 64) {
        const char *cur = g_ticks[h].name.load(std::memory_order_acquire);
        if (cur == nullptr) {
            const char *expect = nullptr;
            if (g_ticks[h].name.compare_exchange_strong(expect, what)) cur = what;
            else cur = expect;
        }
        if (cur == what) {
            g_ticks[h].cycles.fetch_add(now - since, std::memory_order_relaxed);
            g_ticks[h].samples.fetch_add(1, std::memory_order_relaxed);
            break;
        }
    }
    return now;
}
double host_thread_cpu_seconds() { return thread_cpu_s(); }
void host_profile_thread_exit(const char *who) {
    if (!prof().on) return;
    rusage ru;
    if (getrusage(RUSAGE_THREAD, &ru) != 0) return;
    const std::string w(who);
    add_cpu(w.c_str(), thread_cpu_s());
    add_cpu((w + " kernel-mode").c_str(), (double)ru.ru_stime.tv_sec + 1e-6 * (double)ru.ru_stime.tv_usec);
    add_cpu((w + " k-minor-faults").c_str(), 1e-3 * (double)ru.ru_minflt);
    add_cpu((w + " k-voluntary-switches").c_str(), 1e-3 * (double)ru.ru_nvcsw);
    add_cpu((w + " k-involuntary-switches").c_str(), 1e-3 * (double)ru.ru_nivcsw);
}
host_phase::host_phase(const char *n) : impl(new scoped_phase(n)) {}
host_phase::~host_phase() { delete static_cast<scoped_phase *>(impl); }

std::string host_profile_dump(bool reset) {
    prof_state &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    std::string out;
    char buf[160];
    for (auto &kv : p.acc) {
        snprintf(buf, sizeof(buf), "%-28s %9.3f s  %6llu calls\n", kv.first.c_str(), kv.second.first,
                 (unsigned long long)kv.second.second);
        out += buf;
    }
    {   // time-stamp-counter slots: cycles -> seconds by a one-off calibration against the steady clock
        static const double tsc_hz = [] {
            const auto t0 = std::chrono::steady_clock::now();
            const uint64_t c0 = __builtin_ia32_rdtsc();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.02) {
            }
            return (double)(__builtin_ia32_rdtsc() - c0) / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }();
        std::vector<std::pair<std::string, std::pair<double, unsigned long long>>> rows;
        for (auto &t : g_ticks) {
            const char *n = t.name.load();
            if (!n) continue;
            rows.push_back({n, {(double)t.cycles.load() / tsc_hz, (unsigned long long)t.samples.load()}});
            if (reset) {
                t.cycles.store(0);
                t.samples.store(0);
            }
        }
        std::sort(rows.begin(), rows.end());
        for (auto &r : rows) {
            snprintf(buf, sizeof(buf), "%-36s %9.3f s  %8llu samples  %7.2f us each [tsc]\n", r.first.c_str(), r.second.first,
                     r.second.second, r.second.second ? 1e6 * r.second.first / (double)r.second.second : 0.0);
            out += buf;
        }
    }
    if (reset) p.acc.clear();
    if (p.trace_path && !p.events.empty()) {
        if (FILE *f = fopen(p.trace_path, "w")) {
            for (auto &e : p.events) fprintf(f, "%zu %s %.6f %.6f\n", e.tid, e.name, e.t0, e.t1);
            fclose(f);
        }
    }
    if (reset) p.events.clear();
    return out;
}

// ================================================================ thread pool

namespace {
// Worker threads shared by every parallel_for in flight: each call posts a job (index range +
// function), the workers and the caller itself pull indices from the posted jobs until they are
// exhausted.  Several host threads (one per batch in flight) can therefore run their per-query loops
// concurrently on the same pool.
class pool {
    struct job {
        const std::function<void(size_t)> *fn;
        const char *phase;
        size_t n;
        size_t chunk = 1;  // indices taken per grab (one shared counter: a grab per query was a cache-line fight)
        std::atomic<size_t> next{0}, done{0};
        std::mutex err_mu;
        std::exception_ptr error;
    };

public:
    static pool &get() {
        static pool p;
        return p;
    }
    void resize(unsigned n) {
        shutdown();
        start(n);
    }
    unsigned size() const { return (unsigned)workers.size() + 1; }
    void run(size_t n, const std::function<void(size_t)> &fn) {
        if (n == 0) return;
        if (workers.empty() || n == 1) {
            for (size_t i = 0; i < n; i++) fn(i);
            return;
        }
        auto j = std::make_shared<job>();
        j->fn = &fn;
        j->phase = tl_phase;
        j->n = n;
        j->chunk = std::max<size_t>(1, n / ((workers.size() + 1) * 8));
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs.push_back(j);
        }
        cv.notify_all();
        work_on(*j);  // the caller helps with its own job
        {
            std::unique_lock<std::mutex> lk(mu);
            done_cv.wait(lk, [&] { return j->done.load() == j->n; });
            jobs.erase(std::find(jobs.begin(), jobs.end(), j));
        }
        if (j->error) std::rethrow_exception(j->error);
    }
    ~pool() { shutdown(); }

private:
    pool() {
        unsigned hw = std::thread::hardware_concurrency();
        // (measured on a 256-thread host, round 2: 6 threads 104 k seq/s at 8.3 busy cores, 8: 111 k / 8.9,
        // 10: 112 k / 8.7, 16: 113 k / 10.7, 32: 111 k / 15 -- more threads only fight over the allocator;
        // with the rank's threads pinned to 16 neighbouring cores (sina_amd/affinity.py) 16S is the same
        // at 10 and 12 (112.7 k / 7.3 and 7.6 cores) and the host-bound V4 amplicons gain: 273 k -> 292 k)
        start(hw > 1 ? std::min(hw, 12u) : 1);
    }
    void start(unsigned n) {
        stop = false;
        for (unsigned i = 1; i < n; i++) workers.emplace_back([this] { loop(); });
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_all();
        for (auto &w : workers) w.join();
        workers.clear();
    }
    void work_on(job &j) {
        const bool helper = tl_phase != j.phase;  // (the posting thread's own time is in its scoped_phase)
        const thread_use c0 = (helper && prof().on) ? thread_use_now() : thread_use();
        struct at_exit {
            bool on;
            const char *ph;
            thread_use c0;
            ~at_exit() {
                if (on) add_use(ph, c0);
            }
        } acc{helper && prof().on, j.phase, c0};
        for (;;) {
            const size_t i0 = j.next.fetch_add(j.chunk);
            if (i0 >= j.n) break;
            const size_t i1 = std::min(j.n, i0 + j.chunk);
            for (size_t i = i0; i < i1; i++) {
                try {
                    (*j.fn)(i);
                } catch (...) {
                    std::lock_guard<std::mutex> lk(j.err_mu);
                    if (!j.error) j.error = std::current_exception();
                }
            }
            if (j.done.fetch_add(i1 - i0) + (i1 - i0) == j.n) {
                std::lock_guard<std::mutex> lk(mu);  // (pairs with the waiter's predicate check)
                done_cv.notify_all();
            }
        }
    }
    void loop() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            std::shared_ptr<job> j;
            cv.wait(lk, [&] {
                if (stop) return true;
                for (auto &x : jobs)
                    if (x->next.load() < x->n) {
                        j = x;
                        return true;
                    }
                return false;
            });
            if (stop) return;
            lk.unlock();
            work_on(*j);
            j.reset();
            lk.lock();
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    std::vector<std::shared_ptr<job>> jobs;
    bool stop = false;
};
std::mutex pool_resize_mu;
}  // namespace

void parallel_for(size_t n, const std::function<void(size_t)> &fn) { pool::get().run(n, fn); }
void set_host_threads(unsigned n) {  // (not while loops are running)
    std::lock_guard<std::mutex> lk(pool_resize_mu);
    pool::get().resize(n < 1 ? 1 : n);
}
unsigned host_threads() { return pool::get().size(); }

// ================================================================ tray (src/tray.cpp)

tray::tray(const tray &o)
    : seqno(o.seqno), input_sequence(o.input_sequence), aligned_sequence(o.aligned_sequence),
      alignment_reference(o.alignment_reference), search_result(o.search_result), astats(o.astats),
      family_scores_kmer_k(o.family_scores_kmer_k), query_kmer_count(o.query_kmer_count) {
    log.str(o.log.str());
    log.seekp(0, std::ios_base::end);
    pending_score = o.pending_score;
}
// (the stream's default float format is printf's %g = to_chars(general, 6) -- checked on 2e7 floats)
size_t tray::score_note::render(char *line, size_t cap) const {
    char *w = line, *const end = line + cap;
    auto lit = [&](const char *txt) {
        const size_t n = std::min(strlen(txt), (size_t)(end - w));  // (the line is 130 characters at most)
        memcpy(w, txt, n);
        w += n;
    };
    auto flt = [&](float v) { w = std::to_chars(w, end, v, std::chars_format::general, 6).ptr; };
    lit("scoring: raw=");
    flt(raw);
    lit(", weight=");
    flt(weight);
    lit(", query-len=");
    w = std::to_chars(w, end, len).ptr;
    lit(", aligned-bases=");
    w = std::to_chars(w, end, aligned).ptr;
    lit(", score=");
    flt(score);
    lit("; ");
    return (size_t)(w - line);
}
void tray::score_note::merge_into(std::string &text) const {
    if (!set) return;
    char line[192];
    const size_t n = render(line, sizeof line);
    text.insert(std::min<size_t>(at, text.size()), line, n);
}
std::string tray::log_text() const {
    std::string s(log.view());
    pending_score.merge_into(s);
    return s;
}
tray &tray::operator=(const tray &o) {
    seqno = o.seqno;
    input_sequence = o.input_sequence;
    aligned_sequence = o.aligned_sequence;
    alignment_reference = o.alignment_reference;
    search_result = o.search_result;
    log.str(o.log.str());
    log.seekp(0, std::ios_base::end);
    pending_score = o.pending_score;
    astats = o.astats;
    family_scores_kmer_k = o.family_scores_kmer_k;
    query_kmer_count = o.query_kmer_count;
    return *this;
}
alignment_stats *alignment_stats::shared_default() {
    static alignment_stats none;
    return &none;
}
void tray::destroy() {  // (src/tray.cpp:77-86; the objects go back to their caches, see object_cache)
    object_cache<cseq, cache_query_seq>::give(input_sequence);
    object_cache<cseq, cache_aligned_seq>::give(aligned_sequence);
    object_cache<search::result_vector>::give(alignment_reference);
    object_cache<search::result_vector>::give(search_result);
    if (astats != alignment_stats::shared_default()) delete astats;
    input_sequence = aligned_sequence = nullptr;
    alignment_reference = search_result = nullptr;
    astats = nullptr;
}

// ================================================================ reference_store

namespace {
std::mutex stores_mu;
std::map<std::string, std::shared_ptr<reference_store>> stores;
}  // namespace

std::shared_ptr<reference_store> reference_store::get(const std::string &path) {
    {
        std::lock_guard<std::mutex> lk(stores_mu);
        auto it = stores.find(path);
        if (it != stores.end()) return it->second;
    }
    return open(path);
}
void reference_store::close(const std::string &path) {
    std::lock_guard<std::mutex> lk(stores_mu);
    stores.erase(path);
}

std::shared_ptr<reference_store> reference_store::from_packed(const std::string &key, const uint32_t *ab,
                                                              const uint64_t *off, uint32_t n, uint32_t width,
                                                              const char *const *names) {
    std::shared_ptr<reference_store> s(new reference_store());
    s->path = key;
    s->width = width;
    s->seqs.reserve(n);
    base_vector tmp;
    for (uint32_t i = 0; i < n; i++) {
        std::string nm = names ? std::string(names[i]) : ("ref" + std::to_string(i));
        s->seqs.emplace_back(nm.c_str());
        cseq &c = s->seqs.back();
        tmp.clear();
        for (uint64_t x = off[i]; x < off[i + 1]; x++) tmp.push_back(aligned_base::from_raw(ab[x]));
        c.setAlignedBases(tmp);
        c.setWidth(width);
        c.set_attr(fn::acc, nm);
    }
    s->fill_metas();
    std::lock_guard<std::mutex> lk(stores_mu);
    stores[key] = s;
    return s;
}
void reference_store::fill_metas() {
    metas.resize(seqs.size());
    for (size_t i = 0; i < seqs.size(); i++) {
        const cseq &c = seqs[i];
        metas[i] = ref_meta{(uint32_t)c.size(), c.size() ? c.getById(0).getPosition() : 0u,
                            c.size() ? c.getById(c.size() - 1).getPosition() : 0u};
    }
}

// Minimal aligned-FASTA reader ('>' name [description], sequence lines; '-'/'.' are gaps).
std::shared_ptr<reference_store> reference_store::open(const std::string &path, bool arb_id_order) {
    std::ifstream in(path);
    if (!in) throw std::logic_error("Reference database file " + path + " does not exist");
    std::shared_ptr<reference_store> s(new reference_store());
    s->path = path;
    std::string line, name, data;
    auto flush = [&]() {
        if (name.empty()) return;
        s->seqs.emplace_back(name.c_str(), data.c_str());
        s->seqs.back().set_attr(fn::acc, name);
        data.clear();
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            flush();
            size_t e = line.find_first_of(" \t", 1);
            name = line.substr(1, e == std::string::npos ? std::string::npos : e - 1);
        } else {
            data += line;
        }
    }
    flush();
    if (arb_id_order) {
        std::vector<std::string> names;
        names.reserve(s->seqs.size());
        for (const auto &c : s->seqs) names.push_back(c.getName());
        const std::vector<uint32_t> order = arb_name_order(names);
        std::vector<cseq> sorted;
        sorted.reserve(order.size());
        for (uint32_t from : order) sorted.push_back(std::move(s->seqs[from]));
        s->seqs.swap(sorted);
    }
    for (const auto &c : s->seqs) s->width = std::max(s->width, c.getWidth());
    for (auto &c : s->seqs) c.setWidth(s->width);
    s->fill_metas();
    std::lock_guard<std::mutex> lk(stores_mu);
    stores[path] = s;
    return s;
}

reference_store::~reference_store() {
    for (auto &pool : idle_forks)
        for (auto *f : pool) sina_hip_destroy(f);
    if (ctx) sina_hip_destroy(ctx);
}

reference_store::lease reference_store::worker_device(device_role role) {
    sina_hip_ctx *root = device();
    std::lock_guard<std::mutex> lk(gpu_mu);
    sina_hip_ctx *c = nullptr;
    auto &pool = idle_forks[role];
    if (!pool.empty()) {
        c = pool.front();  // (round robin: every context of the kind sees batches -- and grows -- early)
        pool.erase(pool.begin());
    } else {
        hip_check(sina_hip_fork(root, &c), "sina_hip_fork");
    }
    return lease(this, c, role);
}
void reference_store::reserve_workers(device_role role, unsigned n) {
    sina_hip_ctx *root = device();
    std::lock_guard<std::mutex> lk(gpu_mu);
    auto &pool = idle_forks[role];
    while (pool.size() < n) {
        sina_hip_ctx *c = nullptr;
        hip_check(sina_hip_fork(root, &c), "sina_hip_fork");
        pool.push_back(c);
    }
    for (sina_hip_ctx *c : pool) hip_check(sina_hip_prewarm(c, (int)role), "sina_hip_prewarm");
}
reference_store::lease::~lease() {
    if (!c) return;
    std::lock_guard<std::mutex> lk(st->gpu_mu);
    st->idle_forks[role].push_back(c);
}

const cseq &reference_store::getCseq(const std::string &name) const {
    for (const auto &c : seqs)
        if (c.getName() == name) return c;
    throw std::runtime_error("no such sequence: " + name);
}
const std::string &reference_store::upper_bases(unsigned int id) {
    std::call_once(ubases_once, [this] {
        ubases.resize(seqs.size());
        parallel_for(seqs.size(), [this](size_t i) {
            std::string b = seqs[i].getBases();
            for (auto &ch : b) ch = (char)toupper((unsigned char)ch);
            ubases[i].swap(b);
        });
    });
    return ubases[id];
}
const std::string &reference_store::family_label(unsigned int id) {
    if (!labels_ready.load(std::memory_order_acquire)) {  // (set_attr resets it: stores are not modified while pipelines run)
        std::lock_guard<std::mutex> lk(labels_mu);
        if (!labels_ready.load(std::memory_order_relaxed)) {
            labels.assign(seqs.size(), std::string());
            for (size_t i = 0; i < seqs.size(); i++) {
                loadKey(seqs[i], fn::acc);
                loadKey(seqs[i], fn::start);
                labels[i] = seqs[i].get_attr<std::string>(fn::acc) + "." + seqs[i].get_attr<std::string>(fn::start, "0");
            }
            labels_ready.store(true, std::memory_order_release);
        }
    }
    return labels[id];
}
std::vector<std::string> reference_store::getSequenceNames() const {
    std::vector<std::string> v;
    v.reserve(seqs.size());
    for (const auto &c : seqs) v.push_back(c.getName());
    return v;
}
void reference_store::loadKey(const cseq &, const std::string &) const {
    // attributes of FASTA-backed references are loaded eagerly (acc := name); nothing to fetch
}

sina_hip_ctx *reference_store::device() {
    std::lock_guard<std::mutex> lk(gpu_mu);
    if (ctx) return ctx;
    hip_check(sina_hip_init(device_id, &ctx), "sina_hip_init");
    if (refs_by_broadcast) return ctx;  // (filled in place by the start-up broadcast, sina_amd/dist.py)
    std::vector<uint32_t> ab;
    std::vector<uint64_t> off(seqs.size() + 1, 0);
    size_t total = 0;
    for (const auto &c : seqs) total += c.size();
    ab.reserve(total);
    for (size_t i = 0; i < seqs.size(); i++) {
        const uint32_t *p = seqs[i].packed();
        ab.insert(ab.end(), p, p + seqs[i].size());
        off[i + 1] = ab.size();
    }
    hip_check(sina_hip_upload_refs(ctx, ab.data(), off.data(), (uint32_t)seqs.size(), width), "upload_refs");
    return ctx;
}

void reference_store::ensure_index(unsigned k, bool nofast) {
    sina_hip_ctx *c = device();
    std::lock_guard<std::mutex> lk(gpu_mu);
    if (idx_k == (int)k && idx_nofast == nofast) return;
    // One index per store on the device: a second (k, fast) combination would replace the index
    // under the feet of the batches in flight (the reference keeps one kmer_search object per
    // combination; use a second store for that).
    if (idx_k != -1)
        throw std::logic_error("reference store " + path + " already has a k-mer index for k=" + std::to_string(idx_k) +
                               (idx_nofast ? " (no-fast)" : " (fast)") +
                               "; famfinder and search must use the same --fs-kmer-len / --fs-kmer-no-fast");
    // kmer_search::impl::impl (src/kmer_search.cpp:213-243): a file-backed database keeps its index
    // in <db>.sidx -- load it if it is not older than the database, else build and store it.
    // (":..." names are in-memory stores: always built.)
    std::string idxpath;
    if (!path.empty() && path[0] != ':') {
        const size_t dot = path.find_last_of('.'), slash = path.find_last_of('/');
        idxpath = (dot != std::string::npos && (slash == std::string::npos || dot > slash) ? path.substr(0, dot) : path) +
                  ".sidx";
    }
    bool loaded = false;
    if (!idxpath.empty()) {
        struct stat si, sd;
        if (stat(idxpath.c_str(), &si) == 0 && stat(path.c_str(), &sd) == 0 &&
            (si.st_mtim.tv_sec > sd.st_mtim.tv_sec ||
             (si.st_mtim.tv_sec == sd.st_mtim.tv_sec && si.st_mtim.tv_nsec >= sd.st_mtim.tv_nsec))) {
            std::vector<std::string> names;
            std::vector<uint32_t> offsets, ids;
            if (sidx_load(idxpath, k, nofast, &names, &offsets, &ids) && names.size() == seqs.size()) {
                // The file's posting ids count ITS name list (the reference resolves ids through the names it
                // stored: kmer_search.cpp try_load + getCseq(sequence_names[id])).  This store may number the
                // same database differently -- file order vs the ARB walk order of a stock-SINA .sidx
                // (reference_store::open(path, arb_id_order)) -- so: same names in the same order -> take the
                // lists as they are; the same names in another order -> renumber every list through a
                // name -> id table and re-sort it; anything else -> not this database's index, rebuild.
                bool same_order = true;
                for (size_t i = 0; i < seqs.size() && same_order; i++) same_order = names[i] == seqs[i].getName();
                bool usable = same_order;
                if (!same_order) {
                    std::unordered_map<std::string, uint32_t> id_of;
                    id_of.reserve(seqs.size() * 2);
                    for (size_t i = 0; i < seqs.size(); i++) id_of.emplace(seqs[i].getName(), (uint32_t)i);
                    std::vector<uint32_t> remap(names.size());
                    std::vector<char> hit(seqs.size(), 0);
                    usable = id_of.size() == seqs.size();
                    for (size_t i = 0; i < names.size() && usable; i++) {
                        auto it = id_of.find(names[i]);
                        usable = it != id_of.end() && !hit[it->second];
                        if (usable) {
                            remap[i] = it->second;
                            hit[it->second] = 1;
                        }
                    }
                    // (a corrupt or foreign cache may name ids beyond its own name list: rebuilt, not followed)
                    for (size_t i = 0; i < ids.size() && usable; i++) usable = ids[i] < remap.size();
                    if (usable) {
                        for (auto &id : ids) id = remap[id];
                        parallel_for(offsets.size() - 1, [&](size_t km) {
                            if (offsets[km + 1] - offsets[km] > 1) std::sort(ids.begin() + offsets[km], ids.begin() + offsets[km + 1]);
                        });
                    }
                }
                if (usable) {
                    hip_check(sina_hip_upload_index(c, k, nofast ? 1 : 0, offsets.data(), ids.data(), ids.size()),
                              "upload_index");
                    loaded = true;
                    idx_origin = std::string(same_order ? "loaded " : "loaded (ids renumbered by name) ") + idxpath;
                }
            }
        }
    }
    if (!loaded) {
        hip_check(sina_hip_build_index(c, k, nofast ? 1 : 0), "build_index");
        idx_origin = "built";
        if (!idxpath.empty()) {
            sina_hip_store_view v;
            hip_check(sina_hip_store_view_get(c, &v), "store_view_get");
            std::vector<uint32_t> offsets(((size_t)1 << (2 * k)) + 1), ids(v.n_postings ? v.n_postings : 1);
            hip_check(sina_hip_download_index(c, offsets.data(), ids.data()), "download_index");
            ids.resize(v.n_postings);
            std::vector<std::string> names;
            for (const auto &sq : seqs) names.push_back(sq.getName());
            try {
                sidx_store(idxpath, k, nofast, names, offsets, ids);
            } catch (const std::exception &) {
                // (a read-only database directory: keep the in-memory index, like a failed ofstream)
            }
        }
    }
    idx_k = (int)k;
    idx_nofast = nofast;
}

// ================================================================ kmer_search

class kmer_search::impl {
public:
    std::shared_ptr<reference_store> store;
    unsigned k;
    bool nofast;
    impl(std::shared_ptr<reference_store> s, unsigned k_, bool nofast_) : store(std::move(s)), k(k_), nofast(nofast_) {
        store->ensure_index(k, nofast);
    }
};

namespace {
struct idx_key {
    std::string path;
    int k;
    bool nofast;
    bool operator<(const idx_key &o) const { return std::tie(path, k, nofast) < std::tie(o.path, o.k, o.nofast); }
};
std::mutex indices_mu;
std::map<idx_key, std::shared_ptr<kmer_search::impl>> indices;
}  // namespace

// src/kmer_search.cpp:118-144
kmer_search *kmer_search::get_kmer_search(const std::string &filename, int k, bool nofast) {
    std::lock_guard<std::mutex> lk(indices_mu);
    idx_key key{filename, k, nofast};
    auto it = indices.find(key);
    if (it == indices.end())
        it = indices.emplace(key, std::make_shared<impl>(reference_store::get(filename), (unsigned)k, nofast)).first;
    return new kmer_search(it->second);
}
void kmer_search::release_kmer_search(const std::string &filename, int k, bool nofast) {
    std::lock_guard<std::mutex> lk(indices_mu);
    indices.erase(idx_key{filename, k, nofast});
}
kmer_search::kmer_search(std::shared_ptr<impl> p) : pimpl(std::move(p)) {}
kmer_search::~kmer_search() = default;
unsigned int kmer_search::size() const { return pimpl->store->size(); }

double kmer_search::match(result_vector &, const cseq &, int, int, float, float, reference_store *, bool, int, int,
                          int, int, bool) {
    throw std::runtime_error("Legacy family composition not implemented for internal search");
}

void kmer_search::find(const cseq &query, result_vector &results, unsigned int max) {
    std::vector<const cseq *> q{&query};
    std::vector<result_vector> r;
    find_batch(q, r, max);
    results = std::move(r[0]);
}

// src/kmer_search.cpp:366-420 for a batch of queries
// Number of k-mers of a query with multiplicity (kmer.h:188-201): windows of k bases with exactly one base bit
// each, ending before the last base; with the "fast" prefix filter only those starting with A.  Counted as
// (candidate starts) - (starts whose window holds an ambiguous base): both loops over the bytes vectorise, the
// per-window work is done for the few ambiguous bases only (a running-length loop was 2.7 us per 16S query).
static unsigned count_query_kmers(const uint8_t *m, size_t n, unsigned k, bool count_all) {
    if (n < (size_t)k + 1) return 0;
    const size_t last_start = n - 1 - k;  // window [i, i + k), i + k - 1 <= n - 2
    unsigned total = 0;
    if (count_all) total = (unsigned)(last_start + 1);
    else
        for (size_t i = 0; i <= last_start; i++) total += (m[i] & 0xfu) == 1u;
    long covered_to = -1;  // starts up to here are already taken off
    for (size_t blk = 0; blk + 1 < n; blk += 64) {
        const size_t end = std::min(n - 1, blk + 64);
        unsigned bad = 0;
        for (size_t p = blk; p < end; p++) {
            const unsigned x = m[p] & 0xfu;
            bad += (x == 0u) | ((x & (x - 1u)) != 0u);
        }
        if (bad == 0) continue;
        for (size_t p = blk; p < end; p++) {
            const unsigned x = m[p] & 0xfu;
            if (x != 0u && (x & (x - 1u)) == 0u) continue;
            const long lo = std::max<long>(std::max<long>((long)p - (long)k + 1, covered_to + 1), 0);
            const long hi = std::min<long>((long)p, (long)last_start);
            for (long i = lo; i <= hi; i++) total -= count_all ? 1u : (unsigned)((m[i] & 0xfu) == 1u);
            if (hi > covered_to) covered_to = hi;
        }
    }
    return total;
}

// Batch-level memoisation of repeated queries.  The reference's kmer_search::find keeps the scored list of the
// base strings it has just seen (src/kmer_search.cpp:105,377-378,419: a cache of 32 keyed by getBases()) -- real
// amplicon runs are dominated by repeats.  A GPU batch is thousands of queries wide, so the analogue is inside the
// batch: items with the same key bytes (and, for the aligner, the same family) go to the device ONCE, and every
// item reads the slot of its first occurrence.  rep[i] = index of the first item equal to item i; returns the
// number of distinct items.  set_batch_dedup(false) switches it off (tests compare both ways).
static std::atomic<int> g_dedup{1};  // (set_batch_dedup(false): tests compare against the un-deduplicated run)
void set_batch_dedup(bool on) { g_dedup.store(on ? 1 : 0); }
static bool dedup_enabled() { return g_dedup.load(std::memory_order_relaxed) != 0; }
template <class Hash, class Equal>
static size_t group_equal_items(size_t n, Hash &&hash_of, Equal &&equal, std::vector<uint32_t> &rep) {
    rep.resize(n);
    if (!dedup_enabled() || n < 2) {
        for (size_t i = 0; i < n; i++) rep[i] = (uint32_t)i;
        return n;
    }
    std::vector<uint64_t> h(n);
    parallel_for(n, [&](size_t i) { h[i] = hash_of(i); });
    // open addressing over the first occurrences (a batch is a few thousand items)
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    std::vector<uint32_t> slot(cap, 0xFFFFFFFFu);
    size_t distinct = 0;
    for (size_t i = 0; i < n; i++) {
        size_t at = (size_t)(h[i] * 0x9E3779B97F4A7C15ull >> 20) & (cap - 1);
        for (;;) {
            const uint32_t j = slot[at];
            if (j == 0xFFFFFFFFu) {
                slot[at] = (uint32_t)i;
                rep[i] = (uint32_t)i;
                distinct++;
                break;
            }
            if (h[j] == h[i] && equal(j, i)) {
                rep[i] = j;
                break;
            }
            at = (at + 1) & (cap - 1);
        }
    }
    return distinct;
}
// hash of a query's mask bytes for the grouping of repeats: its two ends (64 bytes each) and its length.  Equal
// queries hash equal; unequal ones that agree there are told apart by the byte comparison that follows a hash match
// -- hashing all 1500 bytes, in famfinder and again in the aligner, was 0.4 us per query.
static uint64_t hash_bytes(const void *p, size_t n, uint64_t seed);
static uint64_t hash_ends(const void *p, size_t n, uint64_t seed) {
    if (n <= 160) return hash_bytes(p, n, seed);
    const unsigned char *b = static_cast<const unsigned char *>(p);
    return hash_bytes(b + n - 64, 64, hash_bytes(b, 64, seed ^ n));
}
static uint64_t hash_bytes(const void *p, size_t n, uint64_t seed) {  // (FNV-1a over 8-byte words + tail)
    const unsigned char *b = static_cast<const unsigned char *>(p);
    uint64_t h = 0xcbf29ce484222325ull ^ seed;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
        h ^= h >> 29;
    }
    for (; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
    return h ^ (h >> 32);
}

void kmer_search::find_batch(const std::vector<const cseq *> &queries, std::vector<result_vector> &results,
                             unsigned int max, std::vector<uint32_t> *kmer_counts) {
    reference_store &st = *pimpl->store;
    const unsigned n = st.size();
    // (the callers' vectors are kept -- famfinder hands in recycled ones -- and only emptied)
    results.resize(queries.size());
    for (auto &r : results) r.clear();
    if (max > n) max = n;
    if (max == 0 || queries.empty()) return;
    st.ensure_index(pimpl->k, pimpl->nofast);
    std::vector<uint64_t> qoff(queries.size() + 1, 0);
    for (size_t i = 0; i < queries.size(); i++) qoff[i + 1] = qoff[i] + queries[i]->size();
    thread_local batch_scratch<uint8_t> qmask_buf;
    uint8_t *const qmask = qmask_buf.get(qoff.back() + 1);
    if (kmer_counts) kmer_counts->assign(queries.size(), 0);
    const unsigned kk = pimpl->k;
    const bool count_all = pimpl->nofast;
    parallel_for(queries.size(), [&](size_t i) {
        uint8_t *dst = qmask + qoff[i];
        const size_t nb = queries[i]->size();
        static_assert(sizeof(aligned_base) == 4, "packed words");
        if (const uint8_t *dense = queries[i]->denseMasks()) memcpy(dst, dense, nb);  // (an unaligned query as its reader left it)
        else masks_of_packed(dst, queries[i]->packed(), nb);
        if (kmer_counts) (*kmer_counts)[i] = count_query_kmers(dst, nb, kk, count_all);
    });
    auto dev = st.worker_device(reference_store::dev_search);
    sina_hip_ctx *ctx = dev.get();
    if (max <= 4096) {
        // repeated queries (same bases, same case: the packed mask bytes) are searched once
        std::vector<uint32_t> rep;
        const size_t nu = group_equal_items(
            queries.size(),
            [&](size_t i) { return hash_ends(qmask + qoff[i], qoff[i + 1] - qoff[i], qoff[i + 1] - qoff[i]); },
            [&](size_t a, size_t b) {
                return qoff[a + 1] - qoff[a] == qoff[b + 1] - qoff[b] &&
                       memcmp(qmask + qoff[a], qmask + qoff[b], qoff[a + 1] - qoff[a]) == 0;
            },
            rep);
        std::vector<uint32_t> slot_of(queries.size());
        const uint8_t *dev_mask = qmask;
        const uint64_t *dev_off = qoff.data();
        std::vector<uint64_t> uoff;
        thread_local batch_scratch<uint8_t> umask_buf;
        if (nu == queries.size()) {
            for (size_t i = 0; i < queries.size(); i++) slot_of[i] = (uint32_t)i;
        } else {  // the distinct queries, packed again for the device
            uoff.assign(nu + 1, 0);
            std::vector<uint32_t> firsts;
            firsts.reserve(nu);
            for (size_t i = 0; i < queries.size(); i++) {
                if (rep[i] == i) {
                    slot_of[i] = (uint32_t)firsts.size();
                    uoff[firsts.size() + 1] = uoff[firsts.size()] + (qoff[i + 1] - qoff[i]);
                    firsts.push_back((uint32_t)i);
                } else {
                    slot_of[i] = slot_of[rep[i]];
                }
            }
            uint8_t *const um = umask_buf.get(uoff.back() + 1);
            parallel_for(nu, [&](size_t u) { memcpy(um + uoff[u], qmask + qoff[firsts[u]], uoff[u + 1] - uoff[u]); });
            dev_mask = um;
            dev_off = uoff.data();
        }
        thread_local batch_scratch<uint32_t> ids_buf;
        thread_local batch_scratch<float> sc_buf;
        uint32_t *const ids = ids_buf.get(nu * max);
        float *const sc = sc_buf.get(nu * max);
        std::vector<uint32_t> cnt(nu);
        {
            scoped_phase ph("ff.kmer_topk(C-ABI)");
            hip_check(sina_hip_kmer_topk(ctx, dev_mask, dev_off, (uint32_t)nu, max, ids, sc, cnt.data()), "kmer_topk");
        }
        parallel_for(queries.size(), [&](size_t i) {
            const size_t u = slot_of[i];
            results[i].reserve(cnt[u]);
            for (uint32_t x = 0; x < cnt[u]; x++)
                results[i].emplace_back(sc[u * max + x], &st.getCseq(ids[u * max + x]));
        });
    } else {
        // rare escalation (famfinder asks for >4096 candidates): the GPU still does the
        // counting; ranking the full score vector is the reference's own partial_sort
        // (src/kmer_search.cpp:405-418).
        std::vector<int16_t> scores(n);
        using pair = std::pair<int16_t, int>;
        std::vector<pair> ranks(n);
        for (size_t i = 0; i < queries.size(); i++) {
            hip_check(sina_hip_kmer_scores(ctx, qmask + qoff[i], (uint32_t)(qoff[i + 1] - qoff[i]),
                                           scores.data()),
                      "kmer_scores");
            for (unsigned r = 0; r < n; r++) ranks[r] = pair(scores[r], (int)r);
            std::partial_sort(ranks.begin(), ranks.begin() + max, ranks.end(), std::greater<pair>());
            results[i].reserve(max);
            for (unsigned x = 0; x < max; x++) results[i].emplace_back(ranks[x].first, &st.getCseq(ranks[x].second));
        }
    }
}

// ================================================================ famfinder

namespace {
struct ff_options {
    TURN_TYPE turn_which;
    ENGINE_TYPE engine;
    std::string posvar_filter;
    unsigned int fs_min, fs_max;
    float fs_msc, fs_msc_max;
    bool fs_leave_query_out;
    unsigned int fs_req, fs_req_full, fs_full_len, fs_req_gaps;
    bool fs_no_fast;
    unsigned int fs_kmer_len, fs_min_len, fs_cover_gene;
    std::string database;
};
ff_options ff_defaults() {  // src/famfinder.cpp:144-203
    ff_options o;
    o.turn_which = TURN_NONE;
    o.engine = ENGINE_SINA_KMER;
    o.fs_min = 40;
    o.fs_max = 40;
    o.fs_msc = .7f;
    o.fs_msc_max = 2;
    o.fs_leave_query_out = false;
    o.fs_req = 1;
    o.fs_req_full = 1;
    o.fs_full_len = 1400;
    o.fs_req_gaps = 10;
    o.fs_no_fast = false;
    o.fs_kmer_len = 10;
    o.fs_min_len = 150;
    o.fs_cover_gene = 0;
    return o;
}
ff_options ff_opts = ff_defaults();

bool to_bool(const std::string &v) { return !(v.empty() || v == "0" || v == "false" || v == "no" || v == "off"); }
std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

// cseq_comparator(CMP_IUPAC_OPTIMISTIC, CMP_DIST_NONE, CMP_COVER_QUERY, false)
// (src/cseq_comparator.cpp:59-117,211-295): column-wise merge walk
float identity_cover_query(const cseq &A, const cseq &B) {
    const auto &a = A.getAlignedBases();
    const auto &b = B.getAlignedBases();
    size_t i = 0, j = 0;
    const size_t ae = a.size(), be = b.size();
    if (ae == 0 || be == 0) return 0.f;
    int match = 0, mismatch = 0, only_a = 0, only_a_over = 0;
    if (a[0].getPosition() < b[0].getPosition()) {
        while (i != ae && a[i].getPosition() < b[j].getPosition()) { ++only_a_over; ++i; }
    } else {
        while (j != be && a[i].getPosition() > b[j].getPosition()) ++j;
    }
    while (i != ae && j != be) {
        const int diff = (int)a[i].getPosition() - (int)b[j].getPosition();
        if (diff > 0) ++j;
        else if (diff < 0) { ++only_a; ++i; }
        else {
            if (a[i].getBase().comp(b[j].getBase())) ++match; else ++mismatch;
            ++i; ++j;
        }
    }
    only_a_over += (int)(ae - i);
    const int base = match + mismatch + only_a + only_a_over;
    return (float)match / base;
}
}  // namespace

void famfinder::reset_options() { ff_opts = ff_defaults(); }

void famfinder::set_option(const std::string &name, const std::string &value) {
    ff_options &o = ff_opts;
    if (name == "db" || name == "ptdb") o.database = value;
    else if (name == "turn") {
        const std::string v = lower(value);
        if (v == "none") o.turn_which = TURN_NONE;
        else if (v == "revcomp" || v.empty()) o.turn_which = TURN_REVCOMP;
        else if (v == "all") o.turn_which = TURN_ALL;
        else throw std::logic_error("invalid value for --turn: " + value);
    } else if (name == "fs-engine") {
        if (lower(value) == "internal") o.engine = ENGINE_SINA_KMER;
        else throw std::logic_error("only the internal k-mer engine is available");
    } else if (name == "fs-kmer-len") o.fs_kmer_len = (unsigned)std::stoul(value);
    else if (name == "fs-req") o.fs_req = (unsigned)std::stoul(value);
    else if (name == "fs-min") o.fs_min = (unsigned)std::stoul(value);
    else if (name == "fs-max") o.fs_max = (unsigned)std::stoul(value);
    else if (name == "fs-msc") o.fs_msc = std::stof(value);
    else if (name == "fs-req-full") o.fs_req_full = (unsigned)std::stoul(value);
    else if (name == "fs-full-len") o.fs_full_len = (unsigned)std::stoul(value);
    else if (name == "fs-req-gaps") o.fs_req_gaps = (unsigned)std::stoul(value);
    else if (name == "fs-min-len") o.fs_min_len = (unsigned)std::stoul(value);
    else if (name == "fs-kmer-no-fast") o.fs_no_fast = to_bool(value);
    else if (name == "fs-msc-max") o.fs_msc_max = std::stof(value);
    else if (name == "fs-leave-query-out") o.fs_leave_query_out = to_bool(value);
    else if (name == "fs-cover-gene") o.fs_cover_gene = (unsigned)std::stoul(value);
    else if (name == "filter") o.posvar_filter = value;
    else throw std::logic_error("famfinder: unknown option " + name);
}

void famfinder::validate_options() {  // src/famfinder.cpp:213-239
    if (ff_opts.database.empty()) throw std::logic_error("Family Finder: Must have reference database (--db/-r)");
    if (ff_opts.fs_req < 1) throw std::logic_error("Family Finder: fs-req must be >= 1");
    if (ff_opts.fs_kmer_len < 1 || ff_opts.fs_kmer_len > 12)
        throw std::logic_error("Family Finder: fs-kmer-len must be in 1..12 on this engine");
}
ENGINE_TYPE famfinder::get_engine() { return ff_opts.engine; }

class famfinder::impl {
public:
    kmer_search *index{nullptr};
    std::shared_ptr<reference_store> arb;
    impl() : arb(reference_store::get(ff_opts.database)) {
        index = kmer_search::get_kmer_search(ff_opts.database, (int)ff_opts.fs_kmer_len, ff_opts.fs_no_fast);
    }
    ~impl() { delete index; }
    int turn_check(const cseq &query, bool all);
    std::vector<int> best_orientations(const std::vector<const cseq *> &queries, bool all);
    void orient_batch(std::vector<tray *> &batch);
    void select_astats(tray &t);
    void run(std::vector<tray *> &batch);
};

famfinder::famfinder() : pimpl(new impl()) {}
famfinder::famfinder(const famfinder &o) = default;
famfinder &famfinder::operator=(const famfinder &o) = default;
famfinder::~famfinder() = default;

int famfinder::turn_check(const cseq &query, bool all) { return pimpl->turn_check(query, all); }

// --turn (behaviour of src/famfinder.cpp:312-378): a query may arrive reversed and / or
// complemented.  Orientation o = (bit 0: reversed) | (bit 1: complemented); each candidate
// orientation gets one top-1 k-mer search and the strictly best-scoring one wins, the lowest o on
// ties and o = 0 when nothing scores above zero.  "revcomp" only tries o = 0 and o = 3, "all" tries
// the four.  All orientation variants of a whole batch go to the GPU as ONE top-1 search.
namespace {
cseq oriented(const cseq &q, int o) {
    cseq v(q);
    if (o & 1) v.reverse();
    if (o & 2) v.complement();
    return v;
}
const char *const orientation_names[4] = {"none", "reversed", "complemented", "reversed and complemented"};
}  // namespace

std::vector<int> famfinder::impl::best_orientations(const std::vector<const cseq *> &queries, bool all) {
    const int tried[4] = {0, 3, 1, 2};
    const int n_tried = all ? 4 : 2;
    std::vector<cseq> variants;
    variants.reserve(queries.size() * (size_t)n_tried);
    for (const cseq *q : queries)
        for (int x = 0; x < n_tried; x++) variants.push_back(oriented(*q, tried[x]));
    std::vector<const cseq *> vp;
    for (const cseq &v : variants) vp.push_back(&v);
    std::vector<search::result_vector> top1;
    index->find_batch(vp, top1, 1);
    std::vector<int> best(queries.size(), 0);
    for (size_t i = 0; i < queries.size(); i++) {
        float by_orientation[4] = {0, 0, 0, 0};
        for (int x = 0; x < n_tried; x++) {
            const search::result_vector &r = top1[i * (size_t)n_tried + x];
            if (!r.empty()) by_orientation[tried[x]] = r[0].score;
        }
        float top = 0;
        for (int o = 0; o < 4; o++)
            if (by_orientation[o] > top) {
                top = by_orientation[o];
                best[i] = o;
            }
    }
    return best;
}

int famfinder::impl::turn_check(const cseq &query, bool all) {
    return best_orientations(std::vector<const cseq *>{&query}, all)[0];
}

// turns the input sequences of a batch in place and records what was done (fn::turn)
void famfinder::impl::orient_batch(std::vector<tray *> &batch) {
    if (ff_opts.turn_which == TURN_NONE) {
        for (tray *t : batch) t->input_sequence->set_attr(fn::turn, "turn-check disabled");
        return;
    }
    std::vector<const cseq *> qs;
    for (tray *t : batch) qs.push_back(t->input_sequence);
    const std::vector<int> o = best_orientations(qs, ff_opts.turn_which == TURN_ALL);
    for (size_t i = 0; i < batch.size(); i++) {
        cseq &c = *batch[i]->input_sequence;
        c.set_attr(fn::turn, orientation_names[o[i]]);
        if (o[i]) c = oriented(c, o[i]);
    }
}

// src/famfinder.cpp:381-436 without the auto-filter (needs ARB fields)
void famfinder::impl::select_astats(tray &t) {
    alignment_stats *astats = nullptr;
    if (!ff_opts.posvar_filter.empty()) {
        for (alignment_stats &as : arb->getAlignmentStats()) {
            if (as.getName() == ff_opts.posvar_filter || as.getName() == ff_opts.posvar_filter + ":ALL" ||
                as.getName() == ff_opts.posvar_filter + ":all")
                astats = new alignment_stats(as);  // trays own (and delete) their astats
        }
    }
    if (astats == nullptr) astats = alignment_stats::shared_default();
    t.astats = astats;
}

// One pass of the filter cascade of famfinder::impl::match over ranked candidates
// (src/famfinder.cpp:497-612). Returns true when the loop condition
// says "enough".
namespace {
struct match_state {
    size_t have = 0, have_full = 0, have_cover_left = 0, have_cover_right = 0;
};
bool match_pass(search::result_vector &results, const cseq &query, match_state &st, const reference_store *store) {
    const ff_options &o = ff_opts;
    const size_t range_begin = 0, range_end = 0;
    st = match_state();
    auto remove = [&](const search::result_item &r) {
        const cseq &s = *r.sequence;
        // (size, first and last column: out of the store's table for its own sequences)
        reference_store::ref_meta m;
        if (store && store->owns(r.sequence)) m = store->meta(store->id_of(r.sequence));
        else m = reference_store::ref_meta{(uint32_t)s.size(), s.size() ? s.begin()->getPosition() : 0u,
                                           s.size() ? s.getById(s.size() - 1).getPosition() : 0u};
        const bool is_full = m.size >= o.fs_full_len;
        const bool is_left = m.size && m.first_pos <= range_begin;
        const bool is_right = m.size && m.last_pos >= range_end;
        if (m.size < o.fs_min_len) return true;
        if (o.fs_leave_query_out && query.name_ref() == s.name_ref()) return true;
        if (o.fs_msc_max <= 2 && o.fs_msc_max < 1 && identity_cover_query(query, s) > o.fs_msc_max) return true;
        const bool min_reached = st.have >= o.fs_min, max_reached = st.have >= o.fs_max;
        const bool score_good = r.score < o.fs_msc;  // sic (src/famfinder.cpp:565-567)
        const bool adds_to_full = o.fs_req_full && st.have_full < o.fs_req_full && is_full;
        const bool adds_to_range = (o.fs_cover_gene && st.have_cover_right < o.fs_cover_gene && is_right) ||
                                   (o.fs_cover_gene && st.have_cover_left < o.fs_cover_gene && is_left);
        if (min_reached && (max_reached || !score_good) && !adds_to_full && !adds_to_range) return true;
        ++st.have;
        if (o.fs_req_full && is_full) ++st.have_full;
        if (o.fs_cover_gene && is_right) ++st.have_cover_right;
        if (o.fs_cover_gene && is_left) ++st.have_cover_left;
        return false;
    };
    // (41 candidates out of a 100 000-sequence store: their records are asked for ahead of the pass)
    if (store)
        for (const auto &r : results)
            if (store->owns(r.sequence)) __builtin_prefetch(&store->meta(store->id_of(r.sequence)));
    results.erase(std::remove_if(results.begin(), results.end(), remove), results.end());
    return !(st.have < o.fs_max || st.have_full < o.fs_req_full || st.have_cover_left < o.fs_cover_gene ||
             st.have_cover_right < o.fs_cover_gene);
}
}  // namespace

// The text of align_family_slv from the list famfinder left in its place (cseq.h lazy_text): items are
// reference id << 32 | whole-number score, owner the store.  The text is sized first and then written through a
// pointer (forty relatives were eighty checked appends).
void render_family_list(const void *owner, const uint64_t *items, size_t n, std::string &out) {
    reference_store *st = const_cast<reference_store *>(static_cast<const reference_store *>(owner));
    size_t total = 0;
    for (size_t x = 0; x < n; x++) {
        const uint32_t v = (uint32_t)items[x];
        const size_t digits = v < 10 ? 1 : v < 100 ? 2 : v < 1000 ? 3 : v < 10000 ? 4 : v < 100000 ? 5 : v < 1000000 ? 6 : v < 10000000 ? 7 : 8;
        total += st->family_label((uint32_t)(items[x] >> 32)).size() + 1 + digits + 4;
    }
    out.resize(total);
    char *w = out.data();
    for (size_t x = 0; x < n; x++) {
        const std::string &label = st->family_label((uint32_t)(items[x] >> 32));
        memcpy(w, label.data(), label.size());
        w += label.size();
        *w++ = ':';
        w = std::to_chars(w, w + 10, (uint32_t)items[x]).ptr;
        memcpy(w, ".00 ", 4);
        w += 4;
    }
}

// src/famfinder.cpp:439-494 for a batch; the k-mer search of every escalation
// round is ONE launch over all queries still looking for relatives.
void famfinder::impl::run(std::vector<tray *> &batch) {
    const ff_options &o = ff_opts;
    std::vector<tray *> todo;
    // (the device takes queries of up to SINA_HIP_MAX_QUERY_LEN bases; a longer one fails alone, softly,
    // like a sequence without relatives -- not the whole batch)
    std::vector<tray *> searchable;
    for (tray *t : batch) {
        if (t->input_sequence->size() > SINA_HIP_MAX_QUERY_LEN) {
            t->log << "unable to align: sequence longer than " << SINA_HIP_MAX_QUERY_LEN << " bases;";
            t->input_sequence->set_attr(fn::turn, "turn-check disabled");
        } else {
            searchable.push_back(t);
        }
    }
    orient_batch(searchable);
    for (tray *t : searchable) {
        t->alignment_reference = object_cache<search::result_vector>::take();
        // (what the scores below are: raw k-mer counts of this engine -- the aligner's containment pre-filter asks)
        t->family_scores_kmer_k = o.engine == ENGINE_SINA_KMER ? (o.fs_no_fast ? -(int)o.fs_kmer_len : (int)o.fs_kmer_len) : 0;
        t->query_kmer_count = -1;
        todo.push_back(t);
    }
    size_t max_results = (size_t)o.fs_max + 1;
    const unsigned isize = index->size();
    while (!todo.empty()) {
        std::vector<const cseq *> qs;
        for (tray *t : todo) qs.push_back(t->input_sequence);
        std::vector<search::result_vector> found(todo.size());
        for (size_t i = 0; i < todo.size(); i++) found[i].swap(*todo[i]->alignment_reference);  // (their heap blocks, recycled)
        std::vector<uint32_t> nk;
        {
            scoped_phase ph_find("ff.find_batch");
            index->find_batch(qs, found, (unsigned)std::min<size_t>(max_results, isize),
                              o.engine == ENGINE_SINA_KMER ? &nk : nullptr);
        }
        for (size_t i = 0; i < nk.size(); i++) todo[i]->query_kmer_count = (int)nk[i];
        std::vector<char> done(todo.size(), 0);
        scoped_phase ph_match("ff.match_pass");
        parallel_for(todo.size(), [&](size_t i) {
            search::result_vector &res = *todo[i]->alignment_reference;
            res.swap(found[i]);
            if (res.empty()) {
                done[i] = 1;
                return;
            }
            match_state st;
            const bool enough = match_pass(res, *todo[i]->input_sequence, st, arb.get());
            done[i] = (enough || max_results >= isize) ? 1 : 0;
        });
        std::vector<tray *> next;
        for (size_t i = 0; i < todo.size(); i++)
            if (!done[i]) next.push_back(todo[i]);
        todo.swap(next);
        max_results *= 10;
    }
    scoped_phase ph_post("ff.post");
    parallel_for(batch.size(), [&](size_t i) {
        tray &t = *batch[i];
        if (t.alignment_reference == nullptr) {  // (not searched, see above)
            select_astats(t);
            return;
        }
        auto &vc = *t.alignment_reference;
        cseq &c = *t.input_sequence;
        uint64_t tk = host_tsc();
        // "<acc>.<start>:<score> " per relative (famfinder.cpp:462-470), ":%.2f " for the score.  Where every
        // relative is of the store and has a whole-number score (k-mer counts are) the attribute is kept as the
        // list of (id, score) and the text made when it is read (cseq.h lazy_text, render_family_list below).
        bool plain = true;
        for (auto &r : vc) {
            const float sc = r.score;
            if (!(arb->owns(r.sequence) && sc >= 0.f && sc < 16777216.f && sc == (float)(uint32_t)sc)) {
                plain = false;
                break;
            }
        }
        if (plain) {
            std::vector<uint64_t> &items = c.set_lazy_attr(fn::family, arb.get(), &render_family_list);
            items.resize(vc.size());
            for (size_t x = 0; x < vc.size(); x++) items[x] = ((uint64_t)arb->id_of(vc[x].sequence) << 32) | (uint32_t)vc[x].score;
        } else {
            std::string &fam = c.string_slot(fn::family);  // (written in place, into the sequence's recycled block)
            char buf[64];
            fam.reserve(vc.size() * 24);
            for (auto &r : vc) {
                snprintf(buf, sizeof(buf), ":%.2f ", (double)r.score);
                if (arb->owns(r.sequence)) {
                    fam += arb->family_label(arb->id_of(r.sequence));  // (the "<acc>.<start>" part, cached per reference)
                } else {
                    arb->loadKey(*r.sequence, fn::acc);
                    arb->loadKey(*r.sequence, fn::start);
                    fam += r.sequence->get_attr<std::string>(fn::acc) + "." + r.sequence->get_attr<std::string>(fn::start, "0");
                }
                fam += buf;
            }
        }
        tk = host_tick("ff.post: family string", tk);
        if (o.fs_req_gaps != 0) {  // :472-480
            auto too_few_gaps = [&](search::result_item &it) {
                if (arb->owns(it.sequence)) {
                    const reference_store::ref_meta &m = arb->meta(arb->id_of(it.sequence));
                    return 0 == m.size || m.last_pos - m.size + 1 < o.fs_req_gaps;
                }
                return 0 == it.sequence->size() ||
                       it.sequence->rbegin()->getPosition() - it.sequence->size() + 1 < o.fs_req_gaps;
            };
            vc.erase(std::remove_if(vc.begin(), vc.end(), too_few_gaps), vc.end());
        }
        select_astats(t);
        host_tick("ff.post: gaps filter + astats", tk);
        if (vc.size() < o.fs_req) {  // :486-491
            t.log << "unable to align: too few relatives (" << vc.size() << ");";
            object_cache<search::result_vector>::give(t.alignment_reference);
            t.alignment_reference = nullptr;
        }
    });
}

tray famfinder::operator()(const tray &t) {
    tray r(t);
    std::vector<tray *> b{&r};
    pimpl->run(b);
    return r;
}
void famfinder::operator()(std::vector<tray> &batch) {
    std::vector<tray *> b;
    for (auto &t : batch) b.push_back(&t);
    pimpl->run(b);
}

// ================================================================ host DAG build

// src/mseq.cpp:47-118 + src/graph.h:332-357,466-488, flat arrays.
// ---- --fs-no-graph: pseq + scoring_scheme_profile (src/pseq.{h,cpp}, src/scoring_schemes.h:37-100)
namespace {
struct base_shares {  // base_profile: shares of A, G, C, T/U, opened gaps, extended gaps in a column
    float v[6];
};
base_shares shares_of_base(unsigned mask) {  // base_profile(const base_iupac&), pseq.h:65-86
    base_shares b{};
    const int order = __builtin_popcount(mask & 0xfu);
    if (order > 0) {
        const float val = 1.f / (float)order;
        for (int i = 0; i < 4; i++)
            if (mask & (1u << i)) b.v[i] = val;
    }
    return b;
}
// base_profile::comp, pseq.h:100-113: sixteen products in i-outer, j-inner order, then the gap terms
float profile_comp(const base_shares &a, const base_shares &b, float match, float mismatch, float gap, float gap_ext) {
    float res = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            if (i == j) res += match * a.v[i] * b.v[j];
            else res += mismatch * a.v[i] * b.v[j];
        }
    return res + gap * a.v[4] + gap_ext * a.v[5];
}
}  // namespace

void profile_self_scores(float match, float mismatch, float gap, float gap_ext, float *out16) {
    out16[0] = 0.f;
    for (unsigned m = 1; m < 16; m++) {
        const base_shares b = shares_of_base(m);
        out16[m] = profile_comp(b, b, match, mismatch, gap, gap_ext);
    }
}

void build_family_profile(const std::vector<const cseq *> &fam, float match, float mismatch, float gap, float gap_ext,
                          host_graph *g) {
    const size_t F = fam.size();
    g->pos.clear(); g->mask.clear(); g->weight.clear(); g->score16.clear();
    g->pred_off.assign(1, 0); g->pred.clear(); g->succ_minpos.clear();
    g->width = F ? fam[0]->getWidth() : 0;
    std::vector<uint32_t> at(F, 0);   // next unread base of every member
    std::vector<char> in_gap(F, 1);   // (a member's leading gap counts as extended)
    base_shares code[16];
    for (unsigned m = 0; m < 16; m++) code[m] = shares_of_base(m);
    uint32_t column = 0;
    while (column < g->width) {  // column 0 first, occupied or not; then from occupied column to occupied column
        uint32_t next = g->width;
        int cnt[4] = {0, 0, 0, 0}, opened = 0, extended = 0;
        for (size_t j = 0; j < F; j++) {
            const auto &b = fam[j]->getAlignedBases();
            if (at[j] < b.size() && b[at[j]].getPosition() == column) {
                const unsigned mask = (b[at[j]].raw >> 24) & 0xfu;
                const int order = __builtin_popcount(mask);
                if (order > 0) {
                    const int points = 12 / order;
                    for (int i = 0; i < 4; i++)
                        if (mask & (1u << i)) cnt[i] += points;
                    in_gap[j] = 0;
                }
                ++at[j];
            } else if (in_gap[j]) {
                ++extended;
            } else {
                in_gap[j] = 1;
                ++opened;
            }
            if (at[j] < b.size()) next = std::min(next, (uint32_t)b[at[j]].getPosition());
        }
        base_shares col;
        {
            const int open = opened * 12, ext = extended * 12;
            const int sum = cnt[0] + cnt[1] + cnt[2] + cnt[3] + open + ext;
            for (int i = 0; i < 4; i++) col.v[i] = (float)cnt[i] / sum;
            col.v[4] = (float)open / sum;
            col.v[5] = (float)ext / sum;
        }
        const uint32_t node = (uint32_t)g->pos.size();
        g->pos.push_back(column);
        g->mask.push_back(0);
        g->weight.push_back(0.f);
        g->score16.push_back(std::numeric_limits<float>::infinity());  // (mask 0: no query base has it)
        for (unsigned m = 1; m < 16; m++) g->score16.push_back(profile_comp(col, code[m], match, mismatch, gap, gap_ext));
        if (node > 0) g->pred.push_back(node - 1);
        g->pred_off.push_back((uint32_t)g->pred.size());
        column = next;
    }
    // successor minima (for --insertion=forbid): the next node's column, 1000000 behind the last (mesh.h:480-484)
    const size_t N = g->pos.size();
    g->succ_minpos.resize(N);
    for (size_t m = 0; m < N; m++) g->succ_minpos[m] = m + 1 < N ? g->pos[m + 1] : 1000000u;
}

void build_family_graph(const std::vector<const cseq *> &fam, float fs_weight, host_graph *g) {
    const size_t F = fam.size();
    g->pos.clear(); g->mask.clear(); g->weight.clear();
    g->pred_off.assign(1, 0); g->pred.clear(); g->succ_minpos.clear();
    g->width = F ? fam[0]->getWidth() : 0;
    for (const cseq *c : fam)
        if (c->getWidth() != g->width)
            throw std::runtime_error("Aligned sequences to be stored in mseq of differ in length!");
    std::vector<uint32_t> cur(F, 0);
    std::vector<int32_t> last(F, -1);
    std::vector<float> count;
    struct ed { uint32_t b, a; };
    std::vector<ed> col_edges;
    int32_t node_of_mask[32];
    uint32_t next_col = 0xFFFFFFFFu;
    for (size_t j = 0; j < F; j++)
        if (fam[j]->size()) next_col = std::min(next_col, fam[j]->getById(0).getPosition());
    while (next_col != 0xFFFFFFFFu) {
        const uint32_t col = next_col;
        next_col = 0xFFFFFFFFu;
        for (int &x : node_of_mask) x = -1;
        const uint32_t first_node = (uint32_t)g->pos.size();
        col_edges.clear();
        for (size_t j = 0; j < F; j++) {
            const auto &b = fam[j]->getAlignedBases();
            if (cur[j] < b.size() && b[cur[j]].getPosition() == col) {
                const uint8_t m = b[cur[j]].getBase().mask() & 31;
                int32_t node = node_of_mask[m];
                if (node < 0) {
                    node = (int32_t)g->pos.size();
                    node_of_mask[m] = node;
                    g->pos.push_back(col);
                    g->mask.push_back(m);
                    count.push_back(1.f);
                    g->succ_minpos.push_back(1000000u);
                } else {
                    count[node] += 1.f;
                }
                if (last[j] >= 0) {
                    col_edges.push_back({(uint32_t)node, (uint32_t)last[j]});
                    if (col < g->succ_minpos[last[j]]) g->succ_minpos[last[j]] = col;
                }
                last[j] = node;
                ++cur[j];
            }
            if (cur[j] < b.size()) next_col = std::min(next_col, b[cur[j]].getPosition());
        }
        // reduce_edges: per node ascending unique predecessor ids
        std::sort(col_edges.begin(), col_edges.end(),
                  [](const ed &x, const ed &y) { return x.b != y.b ? x.b < y.b : x.a < y.a; });
        size_t e = 0;
        for (uint32_t node = first_node; node < g->pos.size(); node++) {
            uint32_t prev = 0xFFFFFFFFu;
            while (e < col_edges.size() && col_edges[e].b == node) {
                if (col_edges[e].a != prev) g->pred.push_back(prev = col_edges[e].a);
                ++e;
            }
            g->pred_off.push_back((uint32_t)g->pred.size());
        }
    }
    g->weight.resize(count.size());
    for (size_t i = 0; i < count.size(); i++)  // mseq.cpp:113: double reciprocal + float product
        g->weight[i] = (float)(1.0 / (double)(fs_weight + 1) + (double)(fs_weight * (count[i] / (float)(unsigned)F)));
}

// ================================================================ aligner

aligner::options *aligner::opts = nullptr;

static aligner::options al_defaults() {  // src/align.cpp:231-274
    aligner::options o;
    o.realign = false;
    o.overhang = OVERHANG_ATTACH;
    o.lowercase = LOWERCASE_NONE;
    o.insertion = INSERTION_SHIFT;
    o.calc_idty = false;
    o.fs_no_graph = false;
    o.fs_weight = 1;
    o.match_score = 2;
    o.mismatch_score = -1;
    o.gap_penalty = 5.0f;
    o.gap_ext_penalty = 2.0f;
    o.debug_graph = o.write_used_rels = o.use_subst_matrix = false;
    o.device_graph = true;  // family DAGs are built on the GPU (sina_hip_align_families)
    return o;
}
static aligner::options &al_opts() {
    if (!aligner::opts) aligner::opts = new aligner::options(al_defaults());
    return *aligner::opts;
}
void aligner::reset_options() { al_opts() = al_defaults(); }

void aligner::set_option(const std::string &name, const std::string &value) {
    options &o = al_opts();
    const std::string v = lower(value);
    if (name == "realign") o.realign = to_bool(value);
    else if (name == "overhang") {
        if (v == "attach") o.overhang = OVERHANG_ATTACH;
        else if (v == "remove") o.overhang = OVERHANG_REMOVE;
        else if (v == "edge") o.overhang = OVERHANG_EDGE;
        else throw std::logic_error("invalid value for --overhang: " + value);
    } else if (name == "lowercase") {
        if (v == "none") o.lowercase = LOWERCASE_NONE;
        else if (v == "original") o.lowercase = LOWERCASE_ORIGINAL;
        else if (v == "unaligned") o.lowercase = LOWERCASE_UNALIGNED;
        else throw std::logic_error("invalid value for --lowercase: " + value);
    } else if (name == "insertion") {
        if (v == "shift") o.insertion = INSERTION_SHIFT;
        else if (v == "forbid") o.insertion = INSERTION_FORBID;
        else if (v == "remove") o.insertion = INSERTION_REMOVE;
        else throw std::logic_error("invalid value for --insertion: " + value);
    } else if (name == "fs-weight") o.fs_weight = std::stof(value);
    else if (name == "match-score") o.match_score = std::stof(value);
    else if (name == "mismatch-score") o.mismatch_score = std::stof(value);
    else if (name == "pen-gap") o.gap_penalty = std::stof(value);
    else if (name == "pen-gapext") o.gap_ext_penalty = std::stof(value);
    else if (name == "write-used-rels") o.write_used_rels = to_bool(value);
    else if (name == "calc-idty") o.calc_idty = to_bool(value);
    else if (name == "fs-no-graph") o.fs_no_graph = to_bool(value);
    else if (name == "use-subst-matrix" || name == "debug-graph") {
        if (to_bool(value)) throw std::logic_error("aligner: --" + name + " is outside the accelerated path");
    } else if (name == "device-graph") o.device_graph = to_bool(value);
    else if (name == "db") o.database = value;
    else throw std::logic_error("aligner: unknown option " + name);
}
void aligner::validate_options() {}

aligner::aligner() { al_opts(); }
aligner::~aligner() = default;
aligner::aligner(const aligner &) = default;
aligner &aligner::operator=(const aligner &) = default;

static const std::string &make_datetime() {  // src/align.cpp:287-299 (formatted once per second and thread)
    thread_local time_t last = (time_t)-1;
    thread_local std::string text;
    const time_t t = time(nullptr);
    if (t != last) {
        struct tm tmv;
        char buf[50];
        gmtime_r(&t, &tmv);
        strftime(buf, 50, "%F %T", &tmv);
        text = buf;
        last = t;
    }
    return text;
}

namespace {
// case-insensitive substring search on base strings (boost::algorithm::icontains /
// ifind_first as used at src/align.cpp:329-333,364-369)
std::string upper_copy(const std::string &s) {
    std::string r(s);
    for (auto &c : r) c = (char)toupper((unsigned char)c);
    return r;
}

// First occurrence of `needle` in `hay` (std::string::find's answer).  Exact relatives are the common case of an
// amplicon run -- every member of a family may hold the query -- and std::string::find, a memchr for the first
// character and a memcmp at each hit, stops at every fourth position of a four-letter text: 2.5 us per member of
// 1500 bases, 100 us per query (tools/hoststub, --window 250).  Here: the needle's first four bytes against 32
// positions at a time (four shifted loads; a candidate every 256 positions), the scalar twin compares its first
// eight bytes as one word.
static size_t find_bases_scalar(const char *p, size_t from, size_t last, const std::string &needle) {
    const size_t n = needle.size();
    uint64_t first;
    memcpy(&first, needle.data(), 8);
    for (size_t i = from; i <= last; i++) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        if (w == first && memcmp(p + i + 8, needle.data() + 8, n - 8) == 0) return i;
    }
    return std::string::npos;
}
__attribute__((target("avx2"))) static size_t find_bases_avx2(const char *p, size_t last, const std::string &needle) {
    const size_t n = needle.size();
    const char *nd = needle.data();
    const __m256i c0 = _mm256_set1_epi8(nd[0]), c1 = _mm256_set1_epi8(nd[1]), c2 = _mm256_set1_epi8(nd[2]), c3 = _mm256_set1_epi8(nd[3]);
    size_t i = 0;
    // (a block reads bytes i .. i + 34: inside the text while i + 31 <= last, the needle being 8 or longer)
    for (; i + 31 <= last; i += 32) {
        const __m256i e = _mm256_and_si256(
            _mm256_and_si256(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i)), c0),
                             _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 1)), c1)),
            _mm256_and_si256(_mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 2)), c2),
                             _mm256_cmpeq_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + i + 3)), c3)));
        uint32_t m = (uint32_t)_mm256_movemask_epi8(e);
        while (m) {
            const unsigned b = (unsigned)__builtin_ctz(m);
            if (memcmp(p + i + b + 4, nd + 4, n - 4) == 0) return i + b;
            m &= m - 1;
        }
    }
    return find_bases_scalar(p, i, last, needle);
}
}  // namespace
size_t find_bases(const std::string &hay, const std::string &needle, int force_scalar) {
    const size_t n = needle.size(), h = hay.size();
    if (n < 8 || h < n) return hay.find(needle);
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return (avx2 && !force_scalar) ? find_bases_avx2(hay.data(), h - n, needle) : find_bases_scalar(hay.data(), 0, h - n, needle);
}
namespace {

struct dp_job {
    tray *t = nullptr;
    cseq *c = nullptr;  // working copy (becomes aligned_sequence)
    // the family in mseq input order IS the tray's alignment_reference as the preparation left it (a list of its
    // own per job was 9216 heap blocks per batch)
    const search::result_vector *fam = nullptr;
    size_t family_size() const { return fam->size(); }
    const cseq *member(size_t y) const { return (*fam)[y].sequence; }
    std::vector<const cseq *> family() const {
        std::vector<const cseq *> v;
        v.reserve(fam->size());
        for (const auto &r : *fam) v.push_back(r.sequence);
        return v;
    }
};
}  // namespace

tray aligner::operator()(tray t) {
    std::vector<tray> b{t};
    (*this)(b);
    return b[0];
}

// src/align.cpp:307-460 for a batch of trays.
void aligner::operator()(std::vector<tray> &batch) {
    const options &o = al_opts();
    std::vector<dp_job> jobs(batch.size());
    std::vector<char> need_dp(batch.size(), 0);

    std::unique_ptr<scoped_phase> ph(new scoped_phase("al.prepare(partition)"));
    std::shared_ptr<reference_store> prep_store;
    {
        const std::string db = o.database.empty() ? ff_opts.database : o.database;
        if (!db.empty()) {
            try {
                prep_store = reference_store::get(db);
            } catch (const std::exception &) {
            }
            if (prep_store && !batch.empty() && prep_store->size() > 0) prep_store->upper_bases(0);  // (fills the cache outside the loop)
        }
    }
    // |K(q)|: windows of k unambiguous bases ending before the last base, first base A in "fast" mode
    // (src/kmer.h:69-83,122-124,188-201) -- for a family whose scores famfinder stamped as
    // raw k-mer counts (tray::family_scores_kmer_k: k, negative = no-fast)
    auto query_kmer_count = [](const cseq &c, int stamp) -> unsigned {
        const unsigned k = (unsigned)(stamp < 0 ? -stamp : stamp);
        const bool nofast = stamp < 0;
        const auto &b = c.getAlignedBases();
        unsigned run = 0, n = 0;
        for (size_t e = 0; e + 1 < b.size(); e++) {
            const unsigned m = (b[e].raw >> 24) & 0xfu;
            run = (m & (m - 1)) == 0 && m != 0 ? run + 1 : 0;  // exactly one base bit
            if (run >= k && (nofast || ((b[e + 1 - k].raw >> 24) & 0xfu) == 1u)) n++;
        }
        return n;
    };
    parallel_for(batch.size(), [&](size_t i) {
        tray &t = batch[i];
        if (t.input_sequence == nullptr || t.alignment_reference == nullptr || t.astats == nullptr) return;  // :310-318
        if (t.input_sequence->size() > SINA_HIP_MAX_QUERY_LEN) {  // (device limit: soft failure of this tray)
            t.log << "unable to align: sequence of " << t.input_sequence->size() << " bases (device limit "
                  << SINA_HIP_MAX_QUERY_LEN << ");";
            return;
        }
        uint64_t tk = host_tsc();
        // (the working copy -- src/align.cpp:320-326 -- is made where its bases are known: below for a copied
        // alignment, when the DP is back for the rest; it never holds the query's own bases)
        search::result_vector &vc = *t.alignment_reference;
        // (the query's upper-case base string: only built if a family member passes the k-mer-count
        // test below and has to be searched for it -- exact relatives only)
        std::string ubases_store;
        bool have_ubases = false;
        const size_t n_bases = t.input_sequence->size();
        auto ubases_of_query = [&]() -> const std::string & {
            if (!have_ubases) {
                ubases_store = upper_copy(t.input_sequence->getBases());
                have_ubases = true;
            }
            return ubases_store;
        };
        // upper-case bases of a family member: cached per store (40 members x every query)
        auto ref_ubases = [&](const cseq *r, std::string &tmp) -> const std::string & {
            if (prep_store && prep_store->owns(r)) return prep_store->upper_bases(prep_store->id_of(r));
            tmp = upper_copy(r->getBases());
            return tmp;
        };
        // A family member can only contain the query's bases if it has every one of the query's
        // k-mers, i.e. if its k-mer score (what famfinder ranked it by) is the query's k-mer count --
        // which spares the string search for all but exact relatives.
        const float all_kmers = t.family_scores_kmer_k == 0 ? -1.f
                                : (float)(t.query_kmer_count >= 0 ? (unsigned)t.query_kmer_count
                                                                  : query_kmer_count(*t.input_sequence, t.family_scores_kmer_k));
        auto lacks_query = [&](search::result_item &item) {
            if (item.score < all_kmers) return true;
            std::string tmp;
            return find_bases(ref_ubases(item.sequence, tmp), ubases_of_query()) == std::string::npos;
        };
        tk = host_tick("prepare: kmer count", tk);
        auto holders = std::partition(vc.begin(), vc.end(), lacks_query);
        tk = host_tick("prepare: partition", tk);
        if (holders != vc.end()) {
            if (o.realign) {  // :337-348
                t.log << "sequences ";
                for (auto it = holders; it != vc.end(); ++it)
                    t.log << it->sequence->get_attr<std::string>(fn::acc) << " ";
                t.log << "containing exact candidate removed from family;";
                vc.erase(holders, vc.end());
                if (vc.empty()) {
                    t.log << "that's ALL of them. skipping sequence;";
                    return;
                }
            } else {  // :349-388 steal the alignment
                auto is_query_itself = [&](search::result_item &item) {
                    std::string tmp;
                    return ref_ubases(item.sequence, tmp) == ubases_of_query();
                };
                auto exact = std::find_if(holders, vc.end(), is_query_itself);
                cseq &c = *object_cache<cseq, cache_aligned_seq>::take();
                c.copy_meta(*t.input_sequence);
                if (exact != vc.end()) {
                    c.setAlignedBases(exact->sequence->getAlignedBases());
                    t.log << "copied alignment from identical template sequence "
                          << exact->sequence->get_attr<std::string>(fn::acc) << ":"
                          << exact->sequence->get_attr<std::string>(fn::start, "0") << "; ";
                } else {
                    const auto &refal = holders->sequence->getAlignedBases();
                    std::string tmp;
                    const size_t at = find_bases(ref_ubases(holders->sequence, tmp), ubases_of_query());
                    c.setAlignedBases(refal.data() + at, n_bases);
                    t.log << "copied alignment from (longer) template sequence "
                          << holders->sequence->get_attr<std::string>(fn::acc) << ":"
                          << holders->sequence->get_attr<std::string>(fn::start, "0") << "; ";
                }
                c.setWidth(holders->sequence->getWidth());
                c.set_attr(fn::date, make_datetime());
                c.set_attr(fn::qual, 100);
                if (o.calc_idty) c.set_attr(fn::idty, 100.f);
                c.set_attr(fn::head, 0);
                c.set_attr(fn::tail, 0);
                c.set_attr(fn::filter, "");
                t.aligned_sequence = &c;
                return;
            }
        }
        jobs[i].t = &t;
        jobs[i].fam = &vc;
        need_dp[i] = 1;
        host_tick("prepare: family list", tk);
    });

    ph.reset();
    // group DP jobs by scoring scheme: default-constructed astats (width 0) => simple
    // scheme, otherwise weighted with that tray's weights (src/align.cpp:404-416)
    // ... and by where the family's DAG is built: on the device (families of up to 128 members: the
    // DAG-build kernel's LDS tables) or, for the rare larger family (--fs-max beyond 128), by the host
    // twin of that kernel (build_family_graph) and handed over as a graph (sina_hip_align_graphs)
    constexpr size_t kDeviceFamilyMax = 128;
    std::map<std::pair<std::vector<float>, bool>, std::vector<size_t>> groups;
    // (--fs-no-graph: the family as a profile, built by the host; scoring_scheme_profile takes no positional
    // weights, src/align.cpp:428-433)
    for (size_t i = 0; i < batch.size(); i++)
        if (need_dp[i]) {
            if (o.fs_no_graph) groups[{std::vector<float>(), false}].push_back(i);
            else groups[{batch[i].astats->getWeights(), o.device_graph && jobs[i].family_size() <= kDeviceFamilyMax}].push_back(i);
        }

    std::shared_ptr<reference_store> store;
    if (!groups.empty()) {
        const std::string db = o.database.empty() ? ff_opts.database : o.database;
        store = reference_store::get(db);
    }
    for (auto &grp : groups) {
        const std::vector<float> &weights = grp.first.first;
        const bool graph_on_device = grp.first.second;
        const std::vector<size_t> &idx = grp.second;
        const size_t nq = idx.size();
        sina_hip_align_params p;
        sina_hip_align_params_default(&p);
        p.match_score = o.match_score;
        p.mismatch_score = o.mismatch_score;
        p.gap_penalty = o.gap_penalty;
        p.gap_ext_penalty = o.gap_ext_penalty;
        p.fs_weight = o.fs_weight;
        p.overhang = (int)o.overhang;
        p.lowercase = (int)o.lowercase;
        p.insertion = (int)o.insertion;
        p.weights = weights.empty() ? nullptr : weights.data();
        p.n_weights = (uint32_t)weights.size();
        p.assemble = 1;  // (the device finishes what it can: sina_hip_align_out::assembled)

        ph.reset(), ph.reset(new scoped_phase("al.pack_queries"));  // (the old phase ends first: the new one names the pool jobs)
        std::vector<uint64_t> qoff(nq + 1, 0);
        for (size_t x = 0; x < nq; x++) qoff[x + 1] = qoff[x] + jobs[idx[x]].t->input_sequence->size();
        thread_local batch_scratch<uint8_t> qmask_buf;
        uint8_t *const qmask = qmask_buf.get(qoff.back() + 1);
        parallel_for(nq, [&](size_t x) {  // (the DP looks at the four base bits only: case does not matter)
            const cseq &qs = *jobs[idx[x]].t->input_sequence;
            const size_t nb = qs.size();
            uint8_t *dst = qmask + qoff[x];
            if (const uint8_t *dense = qs.denseMasks()) memcpy(dst, dense, nb);
            else masks_of_packed(dst, qs.packed(), nb);
        });
        // Repeated queries -- the same bases in the same case against the same ordered family: amplicon runs are
        // full of them -- are aligned ONCE (one DAG, one DP, one walk); every tray then finishes from the device
        // results of its first occurrence, with its own name, log and attributes (group_equal_items above).
        std::vector<uint32_t> rep;
        const size_t dnq = group_equal_items(
            nq,
            [&](size_t x) {
                const dp_job &jb = jobs[idx[x]];
                uint64_t hf = 0xcbf29ce484222325ull ^ jb.family_size();
                for (size_t y = 0; y < jb.family_size(); y++) {
                    hf = (hf ^ (uint64_t)reinterpret_cast<uintptr_t>(jb.member(y))) * 0x100000001b3ull;
                    hf ^= hf >> 29;
                }
                return hash_ends(qmask + qoff[x], qoff[x + 1] - qoff[x], hf);
            },
            [&](size_t a, size_t b) {
                const dp_job &ja = jobs[idx[a]], &jb = jobs[idx[b]];
                if (qoff[a + 1] - qoff[a] != qoff[b + 1] - qoff[b] || ja.family_size() != jb.family_size()) return false;
                for (size_t y = 0; y < ja.family_size(); y++)
                    if (ja.member(y) != jb.member(y)) return false;
                return memcmp(qmask + qoff[a], qmask + qoff[b], qoff[a + 1] - qoff[a]) == 0;
            },
            rep);
        std::vector<uint32_t> slot_of(nq);       // device slot of group member x
        std::vector<size_t> uidx;                // job of device slot u
        std::vector<uint64_t> dqoff_store;
        const uint8_t *dqmask = qmask;
        thread_local batch_scratch<uint8_t> dqmask_buf;
        if (dnq == nq) {
            for (size_t x = 0; x < nq; x++) slot_of[x] = (uint32_t)x;
        } else {
            uidx.reserve(dnq);
            dqoff_store.assign(dnq + 1, 0);
            std::vector<uint32_t> first_x;
            first_x.reserve(dnq);
            for (size_t x = 0; x < nq; x++) {
                if (rep[x] == x) {
                    slot_of[x] = (uint32_t)uidx.size();
                    dqoff_store[uidx.size() + 1] = dqoff_store[uidx.size()] + (qoff[x + 1] - qoff[x]);
                    uidx.push_back(idx[x]);
                    first_x.push_back((uint32_t)x);
                } else {
                    slot_of[x] = slot_of[rep[x]];
                }
            }
            uint8_t *const um = dqmask_buf.get(dqoff_store.back() + 1);
            parallel_for(dnq, [&](size_t u) { memcpy(um + dqoff_store[u], qmask + qoff[first_x[u]], dqoff_store[u + 1] - dqoff_store[u]); });
            dqmask = um;
        }
        const std::vector<uint64_t> &dqoff = dnq == nq ? qoff : dqoff_store;
        std::vector<sina_hip_align_out> out(dnq);
        // (the aligned columns are read where the device copied them, in the context's pinned staging buffer:
        // sina_hip_staged_out_pos -- the context stays leased until the alignments below are finished)
        uint32_t *const out_pos = nullptr;
        auto dev = store->worker_device(reference_store::dev_align);
        sina_hip_ctx *ctx = dev.get();
        uint32_t width = 0;

        {   // ---- the device's share, over the DISTINCT queries of the group (the names below shadow the group's)
        const std::vector<size_t> &group_idx = idx;
        const std::vector<size_t> &idx = dnq == group_idx.size() ? group_idx : uidx;
        const size_t nq = dnq;
        const std::vector<uint64_t> &qoff = dqoff;
        const uint8_t *const qmask = dqmask;
        if (graph_on_device) {
            std::vector<uint64_t> foff(nq + 1, 0);
            for (size_t x = 0; x < nq; x++) foff[x + 1] = foff[x] + jobs[idx[x]].family_size();
            std::vector<uint32_t> fids(foff.back() ? foff.back() : 1);
            parallel_for(nq, [&](size_t x) {
                const dp_job &jb = jobs[idx[x]];
                for (size_t y = 0; y < jb.family_size(); y++) fids[foff[x] + y] = store->id_of(jb.member(y));
            });
            width = store->getAlignmentWidth();
            ph.reset(), ph.reset(new scoped_phase("al.align_families(C-ABI)"));  // (the old phase ends first: the new one names the pool jobs)
            hip_check(sina_hip_align_families(ctx, fids.data(), foff.data(), (uint32_t)nq, qmask, qoff.data(),
                                              &p, out.data(), out_pos),
                      "align_families");
        } else {
            std::vector<host_graph> gs(nq);
            ph.reset(), ph.reset(new scoped_phase("al.host_graph_build"));  // (the old phase ends first: the new one names the pool jobs)
            parallel_for(nq, [&](size_t x) {
                if (o.fs_no_graph)
                    build_family_profile(jobs[idx[x]].family(), -o.match_score, -o.mismatch_score, o.gap_penalty,
                                         o.gap_ext_penalty, &gs[x]);
                else
                    build_family_graph(jobs[idx[x]].family(), o.fs_weight, &gs[x]);
            });
            ph.reset(), ph.reset(new scoped_phase("al.host_graph_concat"));  // (the old phase ends first: the new one names the pool jobs)
            sina_hip_graph_batch gb;
            std::vector<uint64_t> node_off(nq + 1, 0), edge_off(nq + 1, 0);
            for (size_t x = 0; x < nq; x++) {
                node_off[x + 1] = node_off[x] + gs[x].pos.size();
                edge_off[x + 1] = edge_off[x] + gs[x].pred.size();
            }
            std::vector<uint32_t> npos(node_off.back()), pred(edge_off.back() ? edge_off.back() : 1),
                poff(node_off.back() + nq), smin(node_off.back());
            std::vector<uint8_t> nmask(node_off.back());
            std::vector<float> nw(node_off.back());
            std::vector<float> nscore(o.fs_no_graph ? 16 * (size_t)node_off.back() : 0);
            float self16[16];
            if (o.fs_no_graph) profile_self_scores(-o.match_score, -o.mismatch_score, o.gap_penalty, o.gap_ext_penalty, self16);
            parallel_for(nq, [&](size_t x) {
                const host_graph &g = gs[x];
                std::copy(g.pos.begin(), g.pos.end(), npos.begin() + node_off[x]);
                std::copy(g.mask.begin(), g.mask.end(), nmask.begin() + node_off[x]);
                std::copy(g.weight.begin(), g.weight.end(), nw.begin() + node_off[x]);
                std::copy(g.succ_minpos.begin(), g.succ_minpos.end(), smin.begin() + node_off[x]);
                std::copy(g.pred_off.begin(), g.pred_off.end(), poff.begin() + node_off[x] + x);
                std::copy(g.pred.begin(), g.pred.end(), pred.begin() + edge_off[x]);
                if (o.fs_no_graph) std::copy(g.score16.begin(), g.score16.end(), nscore.begin() + 16 * node_off[x]);
            });
            width = gs[0].width;
            gb.nq = (uint32_t)nq;
            gb.node_off = node_off.data();
            gb.edge_off = edge_off.data();
            gb.node_pos = npos.data();
            gb.node_mask = nmask.data();
            gb.node_weight = nw.data();
            gb.pred_off = poff.data();
            gb.pred = pred.data();
            gb.succ_minpos = smin.data();
            gb.width = width;
            gb.node_score16 = o.fs_no_graph ? nscore.data() : nullptr;
            gb.self_score16 = o.fs_no_graph ? self16 : nullptr;
            ph.reset(), ph.reset(new scoped_phase("al.align_graphs(C-ABI)"));  // (the old phase ends first: the new one names the pool jobs)
            hip_check(sina_hip_align_graphs(ctx, &gb, qmask, qoff.data(), &p, out.data(), out_pos),
                      "align_graphs");
        }
        }   // ---- (end of the device's share)

        const uint32_t *const staged_pos = sina_hip_staged_out_pos(ctx);
        // cseq container steps of backtrack() (src/mesh.h:603-736) + do_align attrs (:507-509)
        ph.reset(), ph.reset(new scoped_phase("al.finish(NAST,log)"));  // (the old phase ends first: the new one names the pool jobs)
        parallel_for(nq, [&](size_t x) {
            dp_job &jb = jobs[idx[x]];
            tray &t = *jb.t;
            const sina_hip_align_out &r = out[slot_of[x]];
            if (r.status != 0) throw std::runtime_error("device alignment failed for " + t.input_sequence->getName());
            uint64_t tk = host_tsc();
            // the working copy: name and attributes of the input plus do_align's own (src/align.cpp:507-509,455-457),
            // made in one pass (cseq.h copy_meta_with); its bases follow
            const float score_q = r.raw / r.sum_weight;
            const std::string &date_text = make_datetime();
            const std::string &filter_name = t.astats->getName();
            struct keys_t {
                const std::string *head = cseq::attr_key(fn::head), *tail = cseq::attr_key(fn::tail), *filter = cseq::attr_key(fn::filter),
                                  *qual = cseq::attr_key(fn::qual), *date = cseq::attr_key(fn::date);
            };
            static const keys_t keys;  // ("align_cutoff_head_slv" < "align_cutoff_tail_slv" < "align_filter_slv" < "align_quality_slv" < "aligned_slv")
            const cseq::attr_init extra[5] = {
                cseq::attr_init::of(keys.head, (int)r.cutoff_head),
                cseq::attr_init::of(keys.tail, (int)r.cutoff_tail),
                cseq::attr_init::of(keys.filter, std::string_view(filter_name)),
                cseq::attr_init::of(keys.qual, (int)std::min(100.f, std::max(0.f, 100.f * score_q))),
                cseq::attr_init::of(keys.date, std::string_view(date_text)),
            };
            cseq &c = *object_cache<cseq, cache_aligned_seq>::take();
            jb.c = &c;
            c.copy_meta_with(*t.input_sequence, extra, 5);
            tk = host_tick("finish: working copy + attributes", tk);
            const uint32_t L = (uint32_t)t.input_sequence->size();
            const uint32_t *pos = staged_pos + dqoff[slot_of[x]];
            if (r.assembled) {
                // the device did the container steps (append rule, setWidth, reverse) and a NAST fix-up
                // in which every insertion fitted its gap: the finished bases, and the fix-up's log line
                c.clearSequence();
                base_vector &fin = c.mutableAlignedBases();  // (a recycled sequence: no allocation)
                // (one pass, past the cache: base lists are not zeroed by resize() -- cseq.h, base_block_allocator -- and
                // nobody reads the finished list before a sink takes it)
                fin.resize(r.n_out);
                stream_copy(static_cast<void *>(fin.data()), pos, sizeof(aligned_base) * (size_t)r.n_out);
                c.setWidth(width);
                tk = host_tick("finish: assemble", tk);
                if (o.insertion == INSERTION_REMOVE) t.log << "insertion=remove not implemented, using shift; ";
                if (r.nast_total > 0)
                    t.log << "total inserted bases=" << r.nast_total << ";"
                          << "longest insertion=" << r.nast_longest << ";"
                          << "total inserted bases before shifting=" << r.nast_last_run << ";";
            } else {
            // query bases: the input's, upper-cased unless --lowercase=original (src/align.cpp:324-326)
            const uint32_t *qraw = t.input_sequence->packed();
            const uint32_t keep_case = o.lowercase == LOWERCASE_ORIGINAL ? 0xFFFFFFFFu : ~((uint32_t)16 << 24);
            const uint32_t lower_bit = o.lowercase == LOWERCASE_UNALIGNED ? (uint32_t)16 << 24 : 0u;
            auto qbase = [&](uint32_t i, bool overhang_base) -> uint32_t {  // base bits of query base i, in place
                return ((qraw[i] & keep_case) & 0xFF000000u) | (overhang_base ? lower_bit : 0u);
            };
            const bool keep_over = (o.overhang != OVERHANG_REMOVE);
            const uint32_t tail = keep_over ? (uint32_t)r.cutoff_tail : 0;
            const uint32_t head = keep_over ? (uint32_t)r.cutoff_head : 0;
            const uint32_t n_out = tail + (uint32_t)r.aligned_bases + head;
            // The container steps of backtrack() (src/mesh.h:603-726) in one pass: every emitted base is
            // appended under the container rule (a column left of the current width is moved up to it,
            // src/cseq.cpp:79-95), then the sequence is reversed (order and columns, :283-289).  The
            // columns after the rule never decrease, so the result is written back to front directly.
            c.clearSequence();
            base_vector &fin = c.mutableAlignedBases();  // (a recycled sequence: no allocation)
            fin.resize(n_out);
            uint32_t reach = 0;  // alignment_width while appending (clearSequence: 0)
            bool direct = true;
            {
                uint32_t n = 0;
                auto put = [&](uint32_t column, uint32_t base_bits) {
                    if (column >= reach) reach = column;
                    else column = reach;
                    if (column >= width) direct = false;  // (setWidth would have to repack: the general way below)
                    fin[n_out - 1 - n] = aligned_base::from_raw(((width - 1 - column) & 0xFFFFFFu) | base_bits);
                    n++;
                };
                for (uint32_t k = 0; k < tail; k++) put(pos[n], qbase(L - 1 - k, true));
                for (int k = 0; k < r.aligned_bases; k++) put(pos[n], qbase((uint32_t)r.end_s - (uint32_t)k, false));
                for (uint32_t k = 0; k < head; k++) put(pos[n], qbase((uint32_t)r.cutoff_head - 1 - k, true));
            }
            if (direct) {
                c.setWidth(width);
            } else {  // the same through the container's own operations
                c.clearSequence();
                uint32_t n = 0;
                for (uint32_t k = 0; k < tail; k++, n++) c.append(aligned_base::from_raw((pos[n] & 0xFFFFFFu) | qbase(L - 1 - k, true)));
                for (int k = 0; k < r.aligned_bases; k++, n++)
                    c.append(aligned_base::from_raw((pos[n] & 0xFFFFFFu) | qbase((uint32_t)r.end_s - (uint32_t)k, false)));
                for (uint32_t k = 0; k < head; k++, n++)
                    c.append(aligned_base::from_raw((pos[n] & 0xFFFFFFu) | qbase((uint32_t)r.cutoff_head - 1 - k, true)));
                c.setWidth(width);
                c.reverse();
            }
            tk = host_tick("finish: assemble", tk);
            c.fix_duplicate_positions(t.log, o.lowercase == LOWERCASE_UNALIGNED, o.insertion == INSERTION_REMOVE);
            tk = host_tick("finish: NAST fix-up", tk);
            }
            if (c.getWidth() > width) t.log << "warning: result sequence too wide!";
            const float rval = r.raw, sum_weight = r.sum_weight;
            const float score = rval / sum_weight;
            // (the line itself is rendered when the log is read: tray::score_note -- or at once, for a caller that
            // reads the stream itself)
            {
                const tray::score_note note{rval, sum_weight, score, L, (int32_t)r.aligned_bases, (uint32_t)t.log.view().size(), true};
                if (defer_score_line) {
                    t.pending_score = note;
                } else {
                    char line[192];
                    t.log.write(line, (std::streamsize)note.render(line, sizeof line));
                }
            }
            tk = host_tick("finish: score log text", tk);
            if (o.write_used_rels) {
                std::string s;
                for (size_t y = 0; y < jb.family_size(); y++) s += jb.member(y)->getName() + " ";
                c.set_attr(fn::used_rels, s);
            }
            t.aligned_sequence = &c;
            host_tick("finish: attributes", tk);
        });

        // --calc-idty (src/align.cpp:443-453): best overlap identity of the aligned query with a member
        // of its family -- one comparison launch for the group (sina_hip_compare)
        if (o.calc_idty) {
            scoped_phase ph_idty("al.calc_idty(C-ABI)");
            std::vector<uint64_t> qoff(nq + 1, 0), coff(nq + 1, 0);
            for (size_t x = 0; x < nq; x++) {
                qoff[x + 1] = qoff[x] + jobs[idx[x]].c->size();
                coff[x + 1] = coff[x] + jobs[idx[x]].family_size();
            }
            std::vector<uint32_t> qab(qoff.back() ? qoff.back() : 1), cids(coff.back() ? coff.back() : 1);
            for (size_t x = 0; x < nq; x++) {
                const cseq &c = *jobs[idx[x]].c;
                memcpy(qab.data() + qoff[x], c.packed(), 4 * (size_t)c.size());
                for (size_t y = 0; y < jobs[idx[x]].family_size(); y++)
                    cids[coff[x] + y] = store->id_of(jobs[idx[x]].member(y));
            }
            std::vector<sina_hip_match_counts> counts(coff.back() ? coff.back() : 1);
            auto dev = store->worker_device(reference_store::dev_compare);
            hip_check(sina_hip_compare(dev.get(), qab.data(), qoff.data(), (uint32_t)nq, cids.data(), coff.data(),
                                       SINA_CMP_IUPAC_OPTIMISTIC, 0, counts.data()),
                      "sina_hip_compare");
            const cseq_comparator calc_id(CMP_IUPAC_OPTIMISTIC, CMP_DIST_NONE, CMP_COVER_OVERLAP, false);
            for (size_t x = 0; x < nq; x++) {
                float idty = 0;
                for (uint64_t y = coff[x]; y < coff[x + 1]; y++) idty = std::max(idty, calc_id.score(counts[y]));
                jobs[idx[x]].c->set_attr(fn::idty, 100.f * idty);
            }
        }
    }
}

// ================================================================ cseq_comparator (src/cseq_comparator.cpp)

// The six counters cseq_comparator derives its score from (behaviour of src/cseq_comparator.cpp:
// 56-111,165-206), stated by column membership like compare_kernel (search.hip) states them: a
// filtered (lower-case, with filter_lowercase) base is as good as absent; a base outside the column
// range of the other sequence's remaining bases is "overhang"; inside it, it has a partner in the
// same column (match / mismatch by the IUPAC rule) or it has none.  A side without any remaining
// base gives all-zero counters.  Host version for single pairs and the CPU-side tests; batches go
// through sina_hip_compare.
void cseq_comparator::counts(const cseq &A, const cseq &B, CMP_IUPAC_TYPE iupac, bool filter_lc,
                             sina_hip_match_counts *m) {
    memset(m, 0, sizeof(*m));
    auto remaining = [&](const cseq &s) {
        std::vector<aligned_base> v;
        v.reserve(s.size());
        for (const aligned_base &x : s.getAlignedBases())
            if (!(filter_lc && x.getBase().isLowerCase())) v.push_back(x);
        return v;
    };
    const std::vector<aligned_base> a = remaining(A), b = remaining(B);
    if (a.empty() || b.empty()) return;
    auto same = [&](const aligned_base &x, const aligned_base &y) {
        const base_iupac p = x.getBase(), q = y.getBase();
        return iupac == CMP_IUPAC_OPTIMISTIC ? p.comp(q) : (iupac == CMP_IUPAC_PESSIMISTIC ? p.comp_pessimistic(q) : p.comp_exact(q));
    };
    const uint32_t a_lo = a.front().getPosition(), a_hi = a.back().getPosition();
    const uint32_t b_lo = b.front().getPosition(), b_hi = b.back().getPosition();
    size_t i = 0, j = 0;  // columns ascend on both sides: one merge
    while (i < a.size() || j < b.size()) {
        const uint32_t pa = i < a.size() ? a[i].getPosition() : 0xFFFFFFFFu;
        const uint32_t pb = j < b.size() ? b[j].getPosition() : 0xFFFFFFFFu;
        if (pa == pb) {
            if (same(a[i], b[j])) m->match++;
            else m->mismatch++;
            i++;
            j++;
        } else if (pa < pb) {
            if (pa < b_lo || pa > b_hi) m->only_a_overhang++;
            else m->only_a++;
            i++;
        } else {
            if (pb < a_lo || pb > a_hi) m->only_b_overhang++;
            else m->only_b++;
            j++;
        }
    }
}

// match fraction over the cover rule's denominator, optionally Jukes-Cantor corrected (behaviour of
// src/cseq_comparator.cpp:240-296, :42-44)
float cseq_comparator::score(const sina_hip_match_counts &m) const {
    const int paired = m.match + m.mismatch;
    const int a_alone = m.only_a + m.only_a_overhang, b_alone = m.only_b + m.only_b_overhang;
    int denom = 0;
    switch (cover_rule) {
    case CMP_COVER_ABS: denom = 1; break;
    case CMP_COVER_QUERY: denom = paired + a_alone; break;
    case CMP_COVER_TARGET: denom = paired + b_alone; break;
    case CMP_COVER_OVERLAP: denom = paired + m.only_a + m.only_b; break;
    case CMP_COVER_ALL: denom = paired + m.only_a + m.only_b + m.only_a_overhang + m.only_b_overhang; break;
    case CMP_COVER_AVERAGE: denom = paired + (m.only_a + m.only_b + m.only_a_overhang + m.only_b_overhang) / 2; break;
    case CMP_COVER_MIN: denom = paired + std::min(a_alone, b_alone); break;
    case CMP_COVER_MAX: denom = paired + std::max(a_alone, b_alone); break;
    case CMP_COVER_NOGAP: denom = paired; break;
    default: throw std::logic_error("unknown cover rule");
    }
    const float identity = (float)m.match / denom;
    if (dist_rule != CMP_DIST_JC) return identity;
    return (float)(-3.0 / 4 * log(1.0 - 4.0 / 3 * identity));
}
float cseq_comparator::operator()(const cseq &query, const cseq &target) const {
    sina_hip_match_counts m;
    counts(query, target, iupac_rule, filter_lc_rule, &m);
    return score(m);
}

// ================================================================ search_filter (src/search_filter.cpp)

const char *search_filter::fn_nearest = "nearest_slv";

struct search_filter::options {
    std::string pt_database;
    bool search_all;
    bool fs_no_fast;
    int fs_kmer_len;
    int kmer_candidates;
    float min_sim;
    bool ignore_super;
    int max_result;
    std::string lca_fields;
    std::vector<std::string> v_lca_fields;
    float lca_quorum;
    std::string copy_fields;
    std::vector<std::string> v_copy_fields;
    cseq_comparator comparator;
};
search_filter::options *search_filter::opts = nullptr;

static search_filter::options sf_defaults() {  // src/search_filter.cpp:91-126, cseq_comparator.cpp:434-464
    search_filter::options o;
    o.search_all = false;
    o.fs_no_fast = false;
    o.fs_kmer_len = 10;
    o.kmer_candidates = 1000;
    o.min_sim = .7f;
    o.ignore_super = false;
    o.max_result = 10;
    o.lca_quorum = .7f;
    return o;
}
static search_filter::options &sf_opts() {
    if (!search_filter::opts) search_filter::opts = new search_filter::options(sf_defaults());
    return *search_filter::opts;
}
void search_filter::reset_options() { sf_opts() = sf_defaults(); }
std::string search_filter_database() { return sf_opts().pt_database; }

void search_filter::set_option(const std::string &name, const std::string &value) {
    options &o = sf_opts();
    const std::string v = lower(value);
    if (name == "search-db" || (name == "db" && o.pt_database.empty())) o.pt_database = value;
    else if (name == "db") {}
    else if (name == "search-min-sim") o.min_sim = std::stof(value);
    else if (name == "search-max-result") o.max_result = std::stoi(value);
    else if (name == "lca-fields") o.lca_fields = value;
    else if (name == "lca-quorum") o.lca_quorum = std::stof(value);
    else if (name == "search-all") o.search_all = to_bool(value);
    else if (name == "search-no-fast") o.fs_no_fast = to_bool(value);
    else if (name == "search-kmer-candidates") o.kmer_candidates = std::stoi(value);
    else if (name == "search-kmer-len") o.fs_kmer_len = std::stoi(value);
    else if (name == "search-ignore-super") o.ignore_super = to_bool(value);
    else if (name == "search-copy-fields") o.copy_fields = value;
    else if (name == "search-iupac") {  // validate(), cseq_comparator.cpp:301-318: istarts_with(name, value)
        auto starts = [&](const char *full) { return !v.empty() && std::string(full).compare(0, v.size(), v) == 0; };
        if (starts("optimistic")) o.comparator.iupac_rule = CMP_IUPAC_OPTIMISTIC;
        else if (starts("pessimistic")) o.comparator.iupac_rule = CMP_IUPAC_PESSIMISTIC;
        else if (starts("exact")) o.comparator.iupac_rule = CMP_IUPAC_EXACT;
        else throw std::logic_error("iupac matching must be either optimistic or pessimistic");
    } else if (name == "search-correction") {
        if (v == "none") o.comparator.dist_rule = CMP_DIST_NONE;
        else if (v == "jc") o.comparator.dist_rule = CMP_DIST_JC;
        else throw std::logic_error("distance correction must be either none or jc");
    } else if (name == "search-cover") {
        static const char *names[] = {"abs", "query", "target", "overlap", "all", "average", "min", "max", "nogap"};
        int found = -1;
        for (int i = 0; i < 9; i++)
            if (v == names[i]) found = i;
        if (found < 0)
            throw std::logic_error("coverage type must be one of abs, query, target, overlap,"
                                   "average, nogap, min or max");
        o.comparator.cover_rule = (CMP_COVER_TYPE)found;
    } else if (name == "search-filter-lowercase") o.comparator.filter_lc_rule = to_bool(value);
    else if (name == "search-engine") {
        if (v != "internal" && v != "sina_kmer" && v != "kmer")
            throw std::logic_error("search: only the internal k-mer engine is accelerated");
    } else if (name == "search-kmer-mm") {
        if (std::stoi(value) != 0) throw std::logic_error("search: --search-kmer-mm is a PT-server option");
    } else if (name == "search-kmer-norel") {
        if (to_bool(value)) throw std::logic_error("search: --search-kmer-norel is a PT-server option");
    } else throw std::logic_error("search: unknown option " + name);
}

void search_filter::validate_options() {  // src/search_filter.cpp:130-168
    options &o = sf_opts();
    if (o.pt_database.empty()) throw std::logic_error("need search-db to search");
    if (o.comparator.cover_rule == CMP_COVER_ABS && o.comparator.dist_rule != CMP_DIST_NONE)
        throw std::logic_error("only fractional identity can be distance corrected");  // cseq_comparator.cpp:476-479
    auto split = [](const std::string &s, std::vector<std::string> &out) {  // boost::split on ":,"
        out.clear();
        std::string cur;
        for (char ch : s) {
            if (ch == ':' || ch == ',') {
                out.push_back(cur);
                cur.clear();
            } else {
                cur += ch;
            }
        }
        out.push_back(cur);
        if (out.back().empty()) out.pop_back();
    };
    split(o.lca_fields, o.v_lca_fields);
    split(o.copy_fields, o.v_copy_fields);
}

struct search_filter::priv_data {
    std::shared_ptr<reference_store> arb;
    kmer_search *index = nullptr;
};

search_filter::search_filter() : data(new priv_data) {
    options &o = sf_opts();
    if (o.pt_database.empty()) throw std::logic_error("need search-db to search");
    data->arb = reference_store::get(o.pt_database);
    if (!o.search_all) data->index = kmer_search::get_kmer_search(o.pt_database, o.fs_kmer_len, o.fs_no_fast);
}
search_filter::search_filter(const search_filter &) = default;
search_filter &search_filter::operator=(const search_filter &) = default;
search_filter::~search_filter() = default;

tray search_filter::operator()(tray t) {
    std::vector<tray> b{t};
    (*this)(b);
    return b[0];
}

namespace {
// boost::algorithm::contains(ref aligned bases, query aligned bases, a.comp(b)), search_filter.cpp:264-268
bool contains_query(const cseq &ref, const cseq &q) {
    const auto &h = ref.getAlignedBases();
    const auto &n = q.getAlignedBases();
    if (n.empty()) return true;
    if (n.size() > h.size()) return false;
    for (size_t i = 0; i + n.size() <= h.size(); i++) {
        size_t j = 0;
        while (j < n.size() && h[i + j].getBase().comp(n[j].getBase())) j++;
        if (j == n.size()) return true;
    }
    return false;
}

// LCA classification (behaviour of src/search_filter.cpp:374-409): the taxonomy paths of the search
// results, in result order, vote level by level.  The first path still in the vote names the group
// of the current level; the first path that cannot follow it (it has ended, or names another group)
// is dropped from the vote -- as is the leading path itself once it has ended -- and every drop
// uses up one of the `n_results * (1 - quorum)` allowed outliers.  The vote ends when the outliers
// are used up or no path is left; what all remaining paths agreed on so far is the classification.
std::string lca_vote(const std::vector<std::vector<std::string>> &paths, size_t n_results, float quorum) {
    int outliers_left = n_results * (1 - quorum) + .5;
    std::vector<size_t> voting(paths.size());
    for (size_t i = 0; i < voting.size(); i++) voting[i] = i;
    std::string agreed;
    for (size_t level = 0; outliers_left >= 0 && !voting.empty();) {
        const std::vector<std::string> &lead = paths[voting[0]];
        size_t dissenter = voting.size();
        if (level >= lead.size()) {
            dissenter = 0;
        } else {
            for (size_t k = 1; k < voting.size() && dissenter == voting.size(); k++) {
                const std::vector<std::string> &p = paths[voting[k]];
                if (level >= p.size() || p[level] != lead[level]) dissenter = k;
            }
        }
        if (dissenter < voting.size()) {
            voting.erase(voting.begin() + (std::ptrdiff_t)dissenter);
            outliers_left--;
        } else {
            agreed += lead[level] + ";";
            level++;
        }
    }
    const bool nothing = agreed.empty() || agreed == ";" ||
                         (agreed.size() > 1 && agreed.compare(agreed.size() - 2, 2, ";;") == 0);
    return nothing ? "Unclassified;" : agreed;
}
}  // namespace

// src/search_filter.cpp:244-412 for a batch of trays: the k-mer search and the comparisons of the
// whole batch are one GPU call each (sina_hip_kmer_topk, sina_hip_compare).
void search_filter::operator()(std::vector<tray> &batch) {
    const options &o = sf_opts();
    reference_store &st = *data->arb;
    std::vector<size_t> idx;  // trays that are searched
    for (size_t i = 0; i < batch.size(); i++) {
        tray &t = batch[i];
        if (t.aligned_sequence == nullptr) {
            t.log << "search: no sequence?!;";
            continue;
        }
        if (t.aligned_sequence->size() < 20) {
            t.log << "search:sequence too short (<20 bases);";
            continue;
        }
        t.search_result = new search::result_vector();
        idx.push_back(i);
    }
    if (idx.empty()) return;
    const size_t nq = idx.size();
    const unsigned n_refs = st.size();

    // ---- candidates
    std::vector<search::result_vector> cand(nq);
    if (!o.search_all) {
        scoped_phase ph("sf.find_batch");
        std::vector<const cseq *> qs(nq);
        for (size_t x = 0; x < nq; x++) qs[x] = batch[idx[x]].aligned_sequence;
        data->index->find_batch(qs, cand, (unsigned)o.kmer_candidates);
        if (o.ignore_super) {  // sic: partition() moves the containing ones to the front and the REST is erased
            parallel_for(nq, [&](size_t x) {
                const cseq &c = *batch[idx[x]].aligned_sequence;
                search::result_vector kept;
                for (auto &r : cand[x])
                    if (contains_query(*r.sequence, c)) kept.push_back(r);
                cand[x].swap(kept);
            });
        }
    }

    // ---- scores: one comparison launch per slice of the batch
    {
        scoped_phase ph("sf.compare(C-ABI)");
        auto dev = st.worker_device(reference_store::dev_compare);
        const uint64_t max_pairs = 8u << 20;
        size_t x0 = 0;
        while (x0 < nq) {
            size_t x1 = x0;
            uint64_t pairs = 0;
            while (x1 < nq) {
                const uint64_t k = o.search_all ? n_refs : cand[x1].size();
                if (x1 > x0 && pairs + k > max_pairs) break;
                pairs += k;
                x1++;
            }
            std::vector<uint64_t> qoff(x1 - x0 + 1, 0), coff(x1 - x0 + 1, 0);
            for (size_t x = x0; x < x1; x++) {
                qoff[x - x0 + 1] = qoff[x - x0] + batch[idx[x]].aligned_sequence->size();
                coff[x - x0 + 1] = coff[x - x0] + (o.search_all ? n_refs : cand[x].size());
            }
            std::vector<uint32_t> qab(qoff.back() ? qoff.back() : 1), cids(coff.back() ? coff.back() : 1);
            for (size_t x = x0; x < x1; x++) {
                const cseq &c = *batch[idx[x]].aligned_sequence;
                memcpy(qab.data() + qoff[x - x0], c.packed(), 4 * (size_t)c.size());
                uint32_t *dst = cids.data() + coff[x - x0];
                if (o.search_all) {
                    for (unsigned r = 0; r < n_refs; r++) dst[r] = r;
                } else {
                    for (size_t r = 0; r < cand[x].size(); r++) dst[r] = st.id_of(cand[x][r].sequence);
                }
            }
            std::vector<sina_hip_match_counts> counts(coff.back() ? coff.back() : 1);
            if (coff.back())
                hip_check(sina_hip_compare(dev.get(), qab.data(), qoff.data(), (uint32_t)(x1 - x0), cids.data(),
                                           coff.data(), (int)o.comparator.iupac_rule,
                                           o.comparator.filter_lc_rule ? 1 : 0, counts.data()),
                          "sina_hip_compare");
            for (size_t x = x0; x < x1; x++) {
                const sina_hip_match_counts *m = counts.data() + coff[x - x0];
                if (o.search_all) {
                    cand[x].clear();
                    cand[x].reserve(n_refs);
                    for (unsigned r = 0; r < n_refs; r++) cand[x].emplace_back(o.comparator.score(m[r]), &st.getCseq(r));
                } else {
                    for (size_t r = 0; r < cand[x].size(); r++) cand[x][r].score = o.comparator.score(m[r]);
                }
            }
            x0 = x1;
        }
    }

    // ---- ranking + attributes, per tray
    scoped_phase ph("sf.rank+lca");
    parallel_for(nq, [&](size_t x) {
        tray &t = batch[idx[x]];
        cseq *c = t.aligned_sequence;
        auto &vc = *t.search_result;
        if (o.search_all) {
            // every reference was compared: the `max_result` best that are not super-strings of the
            // query (--search-ignore-super), if above --search-min-sim (behaviour of src/
            // search_filter.cpp:271-296).  The sequence of libstdc++ calls is the contract here:
            // the window is re-sorted from the first kept candidate to the PREVIOUS window end, then
            // widened again, so std::partition also sees entries std::partial_sort left unordered.
            search::result_vector &all = cand[x];
            const auto stop = all.end();
            auto first_kept = all.begin();
            auto window_end = first_kept + (std::ptrdiff_t)std::min<size_t>((size_t)o.max_result, all.size());
            for (;;) {
                std::partial_sort(first_kept, window_end, stop, std::greater<search::result_item>());
                if (o.ignore_super) {
                    window_end = first_kept + (std::ptrdiff_t)std::min<size_t>((size_t)o.max_result, (size_t)(stop - first_kept));
                    first_kept = std::partition(first_kept, window_end, [&](search::result_item &item) {
                        return contains_query(*item.sequence, *c);
                    });
                }
                const bool window_full = first_kept + o.max_result <= window_end;
                if (window_end == stop || window_full) break;
            }
            for (auto it = first_kept; it != window_end && it->score > o.min_sim; ++it) vc.push_back(*it);
        } else {  // :297-331
            vc.swap(cand[x]);
            auto it = vc.begin();
            auto middle = vc.begin() + std::min<size_t>((size_t)o.max_result, vc.size());
            auto end = vc.end();
            std::partial_sort(it, middle, end, std::greater<search::result_item>());
            while (it != middle && it->score > o.min_sim) ++it;
            vc.erase(it, vc.end());
        }

        std::string nearest;
        std::map<std::string, std::vector<std::vector<std::string>>> group_names_map;
        for (auto &i : vc) {
            const cseq &r = *i.sequence;
            st.loadKey(r, "acc");
            st.loadKey(r, "version");
            st.loadKey(r, "start");
            st.loadKey(r, "stop");
            for (const std::string &s : o.v_lca_fields) {
                st.loadKey(r, s);
                std::string tax_path = r.get_attr<std::string>(s);
                if (tax_path == "Unclassified;") continue;
                std::vector<std::string> group_names;  // boost::split(..., is_any_of(";"))
                std::string cur;
                for (char ch : tax_path) {
                    if (ch == ';') {
                        group_names.push_back(cur);
                        cur.clear();
                    } else {
                        cur += ch;
                    }
                }
                group_names.push_back(cur);
                if (group_names.back().empty() || group_names.back() == " ") group_names.pop_back();
                group_names_map[s].push_back(group_names);
            }
            char buf[64];
            snprintf(buf, sizeof buf, "~%.3f ", (double)i.score);  // fmt "{}.{}.{}.{}~{:.3f} "
            nearest += r.get_attr<std::string>("acc") + "." + r.get_attr<std::string>("version") + "." +
                       r.get_attr<std::string>("start") + "." + r.get_attr<std::string>("stop") + buf;
            const std::string acc = r.get_attr<std::string>("acc");
            for (const std::string &s : o.v_copy_fields) {
                st.loadKey(r, s);
                c->set_attr<std::string>(std::string("copy_") + acc + std::string("_") + s, r.get_attr<std::string>(s));
            }
        }
        c->set_attr<std::string>(fn_nearest, nearest);
        for (const std::string &s : o.v_lca_fields)
            c->set_attr<std::string>(std::string("lca_") + s, lca_vote(group_names_map[s], vc.size(), o.lca_quorum));
    });
}

}  // namespace sina
