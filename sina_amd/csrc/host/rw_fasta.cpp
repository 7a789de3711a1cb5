// FASTA source / sink stages and the --show-dist accuracy metrics.
//
// Behaviour of the reference's src/rw_fasta.cpp (reader :229-315, writer :332-541) and Log::printer
// (src/log.cpp:279-325 show_dist, :364-430 operator()), written from that behaviour on this
// repository's own I/O layer:
//   * input goes through a block-buffered LINE CUTTER (4 MB blocks, memchr for the line ends) over a
//     byte source -- a plain file descriptor or a gzip stream (".gz" by extension; zlib is looked up at
//     run time) -- and a record is parsed as spans of that buffer: header, "; key = value" comments,
//     sequence lines; name / full_name split at the first blank, sequences with characters outside the
//     IUPAC alphabet are skipped and counted, --fasta-idx / --fasta-block slice the file by byte offset;
//   * output composes ONE text buffer per sequence (header + meta in one of three styles, the csv side
//     file's row, the sequence wrapped to --line-length) and hands it to a byte sink in one call; dots
//     vs dashes, DNA vs RNA, the --min-idty filter and --add-relatives as in the reference.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "stages.h"

namespace sina {

// ---------------------------------------------------------------- options

struct rw_fasta::options {
    FASTA_META_TYPE fastameta = FASTA_META_NONE;
    int line_length = 0;
    float min_idty = 0.f;
    long fasta_block = 0, fasta_idx = 0;
    bool out_dots = false, out_dna = false;
};
rw_fasta::options *rw_fasta::opts = nullptr;
static rw_fasta::options &fa_opts() {
    if (!rw_fasta::opts) rw_fasta::opts = new rw_fasta::options();
    return *rw_fasta::opts;
}
void rw_fasta::reset_options() { fa_opts() = options(); }
void rw_fasta::set_option(const std::string &name, const std::string &value) {
    options &o = fa_opts();
    std::string v(value);
    for (auto &c : v) c = (char)tolower((unsigned char)c);
    auto to_bool = [&]() { return v == "1" || v == "true" || v == "yes" || v == "on" || v.empty(); };
    if (name == "meta-fmt") {
        if (v == "none") o.fastameta = FASTA_META_NONE;
        else if (v == "header") o.fastameta = FASTA_META_HEADER;
        else if (v == "comment") o.fastameta = FASTA_META_COMMENT;
        else if (v == "csv") o.fastameta = FASTA_META_CSV;
        else throw std::logic_error("must be one of 'none', 'header', 'comment' or 'cvs'");
    } else if (name == "line-length") o.line_length = std::stoi(value);
    else if (name == "min-idty") o.min_idty = std::stof(value);
    else if (name == "fasta-write-dna") o.out_dna = to_bool();
    else if (name == "fasta-write-dots") o.out_dots = to_bool();
    else if (name == "fasta-idx") o.fasta_idx = std::stol(value);
    else if (name == "fasta-block") o.fasta_block = std::stol(value);
    else throw std::logic_error("fasta: unknown option " + name);
}

// ---------------------------------------------------------------- byte sources and sinks

// (aligned FASTA is 50 KB a sequence: whole blocks per system call, both ways)
static constexpr size_t kBlockBytes = 4u << 20;

namespace {

// zlib, looked up at run time: libz.so.1 is part of every base image, its development link is not,
// and nothing else of the library needs it (reference: boost::iostreams gzip filters when the file
// name ends in ".gz", src/rw_fasta.cpp:200-202,358-360)
struct zlib_api {
    void *(*open)(const char *, const char *) = nullptr;
    int (*read)(void *, void *, unsigned) = nullptr;
    int (*write)(void *, const void *, unsigned) = nullptr;
    int (*close)(void *) = nullptr;
    int (*buffer)(void *, unsigned) = nullptr;
    int (*flush)(void *, int) = nullptr;
    const char *(*error)(void *, int *) = nullptr;
    static const zlib_api &get() {
        static const zlib_api api = [] {
            zlib_api a;
            void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("libz.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) throw std::runtime_error("gzip'ed FASTA needs zlib (libz.so.1 not found)");
            a.open = reinterpret_cast<void *(*)(const char *, const char *)>(dlsym(h, "gzopen"));
            a.read = reinterpret_cast<int (*)(void *, void *, unsigned)>(dlsym(h, "gzread"));
            a.write = reinterpret_cast<int (*)(void *, const void *, unsigned)>(dlsym(h, "gzwrite"));
            a.close = reinterpret_cast<int (*)(void *)>(dlsym(h, "gzclose"));
            a.buffer = reinterpret_cast<int (*)(void *, unsigned)>(dlsym(h, "gzbuffer"));
            a.flush = reinterpret_cast<int (*)(void *, int)>(dlsym(h, "gzflush"));
            a.error = reinterpret_cast<const char *(*)(void *, int *)>(dlsym(h, "gzerror"));
            if (!a.open || !a.read || !a.write || !a.close) throw std::runtime_error("zlib without the gz* functions");
            return a;
        }();
        return api;
    }
    std::string why(void *f) const {
        int code = 0;
        const char *msg = (error && f) ? error(f, &code) : nullptr;
        return msg ? msg : "unknown zlib error";
    }
};

bool has_gz_extension(const std::string &path) { return path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0; }

struct byte_source {
    virtual ~byte_source() = default;
    virtual size_t pull(char *dst, size_t n) = 0;  // 0 at the end of the input; throws on a damaged one
};
struct fd_source : byte_source {
    int fd;
    explicit fd_source(const std::string &path) : fd(::open(path.c_str(), O_RDONLY | O_CLOEXEC)) {}
    ~fd_source() override {
        if (fd >= 0) ::close(fd);
    }
    size_t pull(char *dst, size_t n) override {
        for (;;) {
            const ssize_t got = ::read(fd, dst, n);
            if (got >= 0) return (size_t)got;
            if (errno != EINTR) throw std::runtime_error("read error on FASTA input");
        }
    }
};
struct gz_source : byte_source {
    const zlib_api &z = zlib_api::get();
    void *f;
    explicit gz_source(const std::string &path) : f(z.open(path.c_str(), "rb")) {
        if (f && z.buffer) z.buffer(f, 1u << 20);
    }
    ~gz_source() override {
        if (f) z.close(f);
    }
    size_t pull(char *dst, size_t n) override {
        const int got = z.read(f, dst, (unsigned)std::min<size_t>(n, 1u << 30));
        // (a corrupt or truncated archive must not look like a short file)
        // (zlib hands out what it could decode and then reports 0 bytes with an error pending)
        int code = 0;
        if (got <= 0 && z.error) z.error(f, &code);
        if (got < 0 || code < 0) throw std::runtime_error("gzip'ed FASTA input is damaged: " + z.why(f));
        return (size_t)got;
    }
};

struct byte_sink {
    virtual ~byte_sink() = default;
    virtual void push(const char *src, size_t n) = 0;
    virtual void flush() = 0;
};
struct fd_sink : byte_sink {  // collects up to a block, then one write(2)
    int fd;
    std::string pending;
    explicit fd_sink(const std::string &path) : fd(::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666)) {
        pending.reserve(kBlockBytes + (64u << 10));
    }
    ~fd_sink() override {
        if (fd >= 0) {
            try {
                flush();
            } catch (...) {
            }
            ::close(fd);
        }
    }
    void push(const char *src, size_t n) override {
        pending.append(src, n);
        if (pending.size() >= kBlockBytes) flush();
    }
    void flush() override {
        size_t done = 0;
        while (done < pending.size()) {
            const ssize_t put = ::write(fd, pending.data() + done, pending.size() - done);
            if (put < 0) {
                if (errno == EINTR) continue;
                throw std::runtime_error("write error on FASTA output");
            }
            done += (size_t)put;
        }
        pending.clear();
    }
};
struct gz_sink : byte_sink {
    const zlib_api &z = zlib_api::get();
    void *f;
    explicit gz_sink(const std::string &path) : f(z.open(path.c_str(), "wb")) {
        if (f && z.buffer) z.buffer(f, 1u << 20);
    }
    ~gz_sink() override {
        if (f) z.close(f);  // (errors at this point have nobody to go to: flush() is where they surface)
    }
    bool dirty = false;
    void push(const char *src, size_t n) override {
        while (n > 0) {
            const unsigned part = (unsigned)std::min<size_t>(n, 1u << 30);
            if (z.write(f, src, part) != (int)part) throw std::runtime_error("gzip'ed FASTA output failed: " + z.why(f));
            src += part;
            n -= part;
            dirty = true;
        }
    }
    // finishes the compressed stream written so far (a later write starts a new gzip member, which every
    // reader concatenates): what a deferred write error would otherwise only show at close
    void flush() override {
        if (!dirty || !z.flush) return;
        if (z.flush(f, 4 /* Z_FINISH */) != 0) throw std::runtime_error("gzip'ed FASTA output failed: " + z.why(f));
        dirty = false;
    }
};

// Cuts the input into lines.  A line is handed out as a span of the block buffer, NUL-terminated in
// place (the '\n' is overwritten), together with whether it ended in a line feed at all -- the last
// line of a file may not.
class line_cutter {
public:
    explicit line_cutter(std::unique_ptr<byte_source> s, uint64_t start_offset = 0)
        : src(std::move(s)), buf(kBlockBytes + 1), base(start_offset) {}
    // first byte of the next line, -1 at the end of the input
    int next_byte() {
        if (head == tail && !refill()) return -1;
        return (unsigned char)buf[head];
    }
    // the next line; false at the end of the input
    bool cut(char *&line, size_t &len, bool &had_newline) {
        size_t from = head;  // (where the search for the line end goes on after a refill)
        for (;;) {
            if (char *nl = static_cast<char *>(memchr(buf.data() + from, '\n', tail - from))) {
                line = buf.data() + head;
                len = (size_t)(nl - line);
                *nl = 0;
                head += len + 1;
                had_newline = true;
                return true;
            }
            from = tail;
            const size_t shift = head;
            if (!refill()) break;
            from -= shift - head;  // (refill moved the unread bytes to the front)
        }
        if (head == tail) return false;
        line = buf.data() + head;
        len = tail - head;
        buf[tail] = 0;  // (one spare byte behind the block)
        head = tail;
        had_newline = false;
        return true;
    }
    // file offset of the next unread byte
    uint64_t offset() const { return base + head; }
    bool exhausted() const { return drained && head == tail; }

private:
    bool refill() {  // more bytes behind the unread ones; false if the source has none
        if (drained) return false;
        if (head > 0) {
            memmove(buf.data(), buf.data() + head, tail - head);
            base += head;
            tail -= head;
            head = 0;
        }
        if (tail + 1 >= buf.size()) buf.resize(buf.size() * 2);  // a line longer than the block
        const size_t got = src->pull(buf.data() + tail, buf.size() - 1 - tail);
        if (got == 0) {
            drained = true;
            return false;
        }
        tail += got;
        return true;
    }
    std::unique_ptr<byte_source> src;
    std::vector<char> buf;
    size_t head = 0, tail = 0;
    uint64_t base;
    bool drained = false;
};

std::string_view strip(std::string_view s) {  // without blanks at either end
    while (!s.empty() && isspace((unsigned char)s.front())) s.remove_prefix(1);
    while (!s.empty() && isspace((unsigned char)s.back())) s.remove_suffix(1);
    return s;
}

}  // namespace

// ---------------------------------------------------------------- reader

struct rw_fasta::reader::priv_data {
    std::unique_ptr<line_cutter> lines;
    std::string filename;
    int lineno = 0, seqno = 0, skipped = 0;
    uint64_t block_end = 0;  // --fasta-block: records that start behind this offset belong to the next slice

    // One record.  0: none left; 1: `c` filled; -1: dropped for a character outside the alphabet.
    int record(cseq &c);
};

rw_fasta::reader::reader(const std::string &infile) : data(new priv_data) {
    const options &o = fa_opts();
    data->filename = infile;
    uint64_t start = 0;
    std::unique_ptr<byte_source> src;
    if (has_gz_extension(infile)) {
        if (o.fasta_block > 0) throw std::logic_error("Cannot use --fasta-idx with gzip'ed input");
        auto gz = std::make_unique<gz_source>(infile);
        if (!gz->f) throw std::runtime_error("Unable to open file \"" + infile + "\" for reading.");
        src = std::move(gz);
    } else {
        auto fd = std::make_unique<fd_source>(infile);
        if (fd->fd < 0) throw std::runtime_error("Unable to open file \"" + infile + "\" for reading.");
        if (o.fasta_block > 0) {
            start = (uint64_t)o.fasta_block * (uint64_t)o.fasta_idx;
            if (::lseek(fd->fd, (off_t)start, SEEK_SET) < 0) start = 0;
            data->block_end = (uint64_t)o.fasta_block * (uint64_t)(o.fasta_idx + 1);
        }
        src = std::move(fd);
    }
    data->lines = std::make_unique<line_cutter>(std::move(src), start);
}
rw_fasta::reader::reader(const reader &) = default;
rw_fasta::reader &rw_fasta::reader::operator=(const reader &) = default;
rw_fasta::reader::~reader() = default;
int rw_fasta::reader::skipped() const { return data->skipped; }

int rw_fasta::reader::priv_data::record(cseq &c) {
    line_cutter &in = *lines;
    char *text;
    size_t len;
    bool whole;
    // a slice of the file ends with the first record that starts behind its last byte
    if (block_end != 0 && !in.exhausted() && in.offset() > block_end) return 0;
    // whatever precedes the header line (the tail of a record cut by --fasta-block, stray text) is passed over;
    // like every line that is not sequence data, a header only counts if its line is complete
    for (int b; (b = in.next_byte()) != '>';) {
        if (b < 0 || !in.cut(text, len, whole) || !whole) return 0;
        lineno++;
    }
    if (!in.cut(text, len, whole) || !whole) return 0;
    lineno++;
    {
        std::string_view header(text, len);
        if (!header.empty() && header.back() == '\r') header.remove_suffix(1);
        header.remove_prefix(1);  // '>'
        const size_t blank = header.find_first_of(" \t");
        c.setName(std::string(header.substr(0, blank)));
        if (blank != std::string_view::npos) c.set_attr<std::string>(fn::fullname, std::string(header.substr(blank + 1)));
    }
    // "; key = value" lines behind the header become attributes, other ';' lines are comments
    while (in.next_byte() == ';') {
        if (!in.cut(text, len, whole) || !whole) break;
        lineno++;
        const std::string_view note(text + 1, len - 1);
        const size_t eq = note.find('=');
        if (eq != std::string_view::npos)
            c.set_attr(std::string(strip(note.substr(0, eq))), std::string(strip(note.substr(eq + 1))));
    }
    // everything up to the next header line (or the end of the input) is sequence
    bool bad = false;
    for (int b; (b = in.next_byte()) != '>' && b >= 0;) {
        in.cut(text, len, whole);
        lineno++;
        if (bad) continue;
        try {
            c.append(text);
        } catch (base_iupac::bad_character_exception &) {
            bad = true;  // (the reference reports name, file and line here; the rest of the record is passed over)
        }
    }
    return bad ? -1 : 1;
}

bool rw_fasta::reader::operator()(tray &t) {
    for (;;) {
        t.seqno = ++data->seqno;  // (dropped sequences keep their number)
        std::unique_ptr<cseq> c(new cseq());
        const int got = data->record(*c);
        if (got > 0) {
            t.input_sequence = c.release();
            return true;
        }
        t.input_sequence = nullptr;
        if (got == 0) return false;
        data->skipped++;
    }
}

// ---------------------------------------------------------------- writer

// boost::apply_visitor(lexical_cast_visitor<string>(), variant<string, char, int, float>)
static std::string attr_to_string(const cseq::variant &v) {
    if (const auto *s = std::get_if<std::string>(&v)) return *s;
    if (const auto *c = std::get_if<char>(&v)) return std::string(1, *c);
    if (const auto *i = std::get_if<int>(&v)) return std::to_string(*i);
    char buf[64];
    snprintf(buf, sizeof buf, "%.9g", (double)std::get<float>(v));  // lexical_cast<string>(float): 9 digits
    return buf;
}

namespace {

// one csv field behind `row`: as it is, or -- if it holds a quote, a comma or a line break -- in
// quotes with its own quotes doubled
void csv_field(std::string &row, std::string_view field) {
    bool plain = true;
    for (const char ch : field) plain = plain && ch != '"' && ch != ',' && ch != '\r' && ch != '\n';
    if (plain) {
        row.append(field);
        return;
    }
    row.push_back('"');
    for (const char ch : field) {
        if (ch == '"') row.push_back('"');
        row.push_back(ch);
    }
    row.push_back('"');
}

// How a sequence's attributes appear in the output (--meta-fmt).  A style adds to the record's header
// line, to the lines between header and sequence, and -- csv -- to the side file.
using attr_map = std::remove_cv_t<std::remove_reference_t<decltype(std::declval<const cseq &>().get_attrs())>>;
struct meta_style {
    virtual ~meta_style() = default;
    // (the alignment family is too long for any of them; the full name already follows the name)
    static bool in_fasta(const std::string &key) { return key != fn::family && key != fn::fullname; }
    virtual void on_header_line(std::string &, const attr_map &) const {}
    virtual void below_header(std::string &, const attr_map &) const {}
    virtual void side_file(std::string &, const cseq &, const attr_map &, bool /*first*/) const {}
};
struct meta_in_header : meta_style {  // >name full name [key=value] [key=value]
    void on_header_line(std::string &rec, const attr_map &attrs) const override {
        for (const auto &kv : attrs) {
            if (!in_fasta(kv.key())) continue;
            const std::string val = attr_to_string(kv.second);
            if (val.empty()) continue;
            rec.append(" [").append(kv.key()).append("=").append(val).append("]");
        }
    }
};
struct meta_in_comments : meta_style {  // ; key=value lines
    void below_header(std::string &rec, const attr_map &attrs) const override {
        for (const auto &kv : attrs)
            if (in_fasta(kv.key())) rec.append("; ").append(kv.key()).append("=").append(attr_to_string(kv.second)).append("\n");
    }
};
struct meta_in_csv : meta_style {  // <output>.csv: a title row with the first sequence's keys, a row per sequence
    void side_file(std::string &rows, const cseq &c, const attr_map &attrs, bool first) const override {
        if (first) {
            rows.append("name");
            for (const auto &kv : attrs)
                if (kv.key() != fn::family) csv_field(rows.append(","), kv.key());
            rows.append("\r\n");
        }
        rows.append(c.getName());
        for (const auto &kv : attrs)
            if (kv.key() != fn::family) csv_field(rows.append(","), attr_to_string(kv.second));
        rows.append("\r\n");
    }
};

}  // namespace

struct rw_fasta::writer::priv_data {
    std::unique_ptr<byte_sink> out, out_csv;
    std::unique_ptr<meta_style> style;
    std::string rec, rows;  // the record / csv rows being composed
    int count = 0, excluded = 0;
    std::unordered_set<std::string> relatives_written;
    unsigned long copy_relatives = 0;
    // records composed ahead of time (precompose): the text of a batch's aligned sequences, by sequence object
    std::vector<std::string> pre_text;
    std::unordered_map<const cseq *, size_t> pre_of;
    void compose(const cseq &c, std::string &rec) const;  // the FASTA record of one sequence (no shared state)
    void write(const cseq &c);
};

rw_fasta::writer::writer(const std::string &outfile, unsigned int copy_relatives) : data(new priv_data) {
    data->copy_relatives = copy_relatives;
    const std::string cannot = "Unable to open file \"" + outfile;
    if (has_gz_extension(outfile)) {
        auto gz = std::make_unique<gz_sink>(outfile);
        if (!gz->f) throw std::runtime_error(cannot + "\" for writing.");
        data->out = std::move(gz);
    } else {
        auto fd = std::make_unique<fd_sink>(outfile);
        if (fd->fd < 0) throw std::runtime_error(cannot + "\" for writing.");
        data->out = std::move(fd);
    }
    switch (fa_opts().fastameta) {
    case FASTA_META_NONE: data->style = std::make_unique<meta_style>(); break;
    case FASTA_META_HEADER: data->style = std::make_unique<meta_in_header>(); break;
    case FASTA_META_COMMENT: data->style = std::make_unique<meta_in_comments>(); break;
    case FASTA_META_CSV: {
        data->style = std::make_unique<meta_in_csv>();
        // beside the output, its extension replaced
        const size_t dot = outfile.find_last_of('.'), slash = outfile.find_last_of('/');
        const bool has_ext = dot != std::string::npos && (slash == std::string::npos || dot > slash);
        auto csv = std::make_unique<fd_sink>((has_ext ? outfile.substr(0, dot) : outfile) + ".csv");
        if (csv->fd < 0) throw std::runtime_error(cannot + ".csv\" for writing.");
        data->out_csv = std::move(csv);
        break;
    }
    default: throw std::runtime_error("Unknown meta-fmt output option");
    }
}
rw_fasta::writer::writer(const writer &) = default;
rw_fasta::writer &rw_fasta::writer::operator=(const writer &) = default;
rw_fasta::writer::~writer() = default;
int rw_fasta::writer::written() const { return data->count; }
int rw_fasta::writer::excluded() const { return data->excluded; }
void rw_fasta::writer::flush() {
    data->out->flush();
    if (data->out_csv) data->out_csv->flush();
}

tray rw_fasta::writer::operator()(tray t) {  // :394-435
    if (t.input_sequence == nullptr) throw std::runtime_error("Received broken tray in rw_fasta");
    if (t.aligned_sequence == nullptr) {  // "Not writing sequence N (>name): not aligned"
        ++data->excluded;
        return t;
    }
    if (fa_opts().min_idty > 0) {
        const float idty = t.aligned_sequence->get_attr<float>(fn::idty, 0);
        if (fa_opts().min_idty > idty) {  // "below identity threshold"
            ++data->excluded;
            return t;
        }
    }
    data->write(*t.aligned_sequence);
    if (data->copy_relatives != 0u) {
        auto *relatives = t.search_result != nullptr ? t.search_result : t.alignment_reference;
        if (relatives != nullptr) {
            int i = (int)data->copy_relatives;
            for (auto &item : *relatives) {
                if (data->relatives_written.insert(item.sequence->getName()).second) data->write(*item.sequence);
                if (--i == 0) break;
            }
        }
    }
    return t;
}

// One sequence = one buffer: ">name[ full name][ meta]\n[meta lines]sequence lines", pushed to the sink
// in a single call (and its csv row to the side file).
// The sequence line itself: a 50 000-column alignment row holds ~1500 bases -- the row is one memset to the
// gap character and a scatter of the bases, then the leading and trailing gaps become dots (getAligned()
// appends run by run through a std::string; same bytes).
static void aligned_line(const cseq &c, bool nodots, bool dna, std::string &out) {
    const size_t at = out.size();
    const uint32_t width = c.getWidth();
    const auto &bases = c.getAlignedBases();
    bool plain = true;  // columns strictly ascending and inside the alignment (what every finished sequence is)
    uint32_t prev = 0;
    for (size_t i = 0; i < bases.size() && plain; i++) {
        const uint32_t pos = bases[i].getPosition();
        plain = pos < width && (i == 0 || pos > prev);
        prev = pos;
    }
    if (!plain) {  // (the general way, whatever it makes of such a sequence)
        out += c.getAligned(nodots, dna);
        return;
    }
    out.append(width, '-');
    char *row = out.data() + at;
    for (const auto &b : bases) row[b.getPosition()] = (char)(dna ? b.getBase().iupac_dna() : b.getBase().iupac_rna());
    if (!nodots) {
        if (bases.empty()) {
            memset(row, '.', width);
        } else {
            memset(row, '.', bases.front().getPosition());
            const uint32_t last = bases.back().getPosition();
            memset(row + last + 1, '.', width - last - 1);
        }
    }
}

void rw_fasta::writer::priv_data::compose(const cseq &c, std::string &rec) const {
    const options &o = fa_opts();
    const auto &attrs = c.get_attrs();
    rec.clear();
    rec.append(">").append(c.getName());
    const std::string full_name = c.get_attr<std::string>(fn::fullname, "");
    if (!full_name.empty()) rec.append(" ").append(full_name);
    style->on_header_line(rec, attrs);
    rec.push_back('\n');
    style->below_header(rec, attrs);
    if (o.line_length <= 0) {  // the whole row on one line
        rec.reserve(rec.size() + c.getWidth() + 2);
        aligned_line(c, !o.out_dots, o.out_dna, rec);
        rec.push_back('\n');
        return;
    }
    std::string seq;
    aligned_line(c, !o.out_dots, o.out_dna, seq);
    const size_t per_line = (size_t)o.line_length;
    rec.reserve(rec.size() + seq.size() + seq.size() / per_line + 2);
    size_t at = 0;
    while (at < seq.size()) {  // (an empty sequence gets no line at all when lines are wrapped)
        const size_t n = std::min(per_line, seq.size() - at);
        rec.append(seq, at, n);
        rec.push_back('\n');
        at += n;
    }
}

void rw_fasta::writer::priv_data::write(const cseq &c) {
    if (out_csv) {
        rows.clear();
        style->side_file(rows, c, c.get_attrs(), count == 0);
        out_csv->push(rows.data(), rows.size());
    }
    const auto pre = pre_of.find(&c);
    if (pre != pre_of.end()) {
        const std::string &text = pre_text[pre->second];
        out->push(text.data(), text.size());
    } else {
        compose(c, rec);
        out->push(rec.data(), rec.size());
    }
    count++;
}

// The records of a batch's aligned sequences, composed on the loop pool ahead of the (serial, ordered) calls of
// operator(): what is left for those is the hand-over to the sink.  Same bytes either way; the text is dropped
// with the next precompose().  (Sequences operator() then decides not to write just go unused.)
void rw_fasta::writer::precompose(const std::vector<tray> &batch) {
    priv_data &d = *data;
    d.pre_of.clear();
    std::vector<const cseq *> todo;
    for (const tray &t : batch)
        if (t.aligned_sequence != nullptr && d.pre_of.emplace(t.aligned_sequence, todo.size()).second) todo.push_back(t.aligned_sequence);
    if (d.pre_text.size() < todo.size()) d.pre_text.resize(todo.size());
    parallel_for(todo.size(), [&](size_t i) { d.compose(*todo[i], d.pre_text[i]); });
}

// ---------------------------------------------------------------- Log::printer (--show-dist)

struct log_printer::priv_data {
    int sequence_num = 0;
    // data for computing alignment quality based on a reference (src/log.cpp:262-266)
    double total_sps = 0, total_cpm = 0, total_idty = 0, total_bps = 0;
    bool show_dist = false;
};
log_printer::log_printer(bool show_dist) : data(new priv_data) { data->show_dist = show_dist; }
log_printer::log_printer(const log_printer &) = default;
log_printer &log_printer::operator=(const log_printer &) = default;
log_printer::~log_printer() = default;

// --show-dist (behaviour of src/log.cpp:279-325, without a comparison database: "orig" is the input
// sequence as it was read, which for an aligned input file is the alignment to compare with).
// Three identities, each the match fraction over the first sequence's bases:
//   orig_idty          exact-IUPAC identity of the input alignment with the new one (the SP score),
//   orig_closest_idty  identity of the input alignment with its closest relative,
//   closest_idty       identity of the new alignment with that same relative,
// and cpm, the loss against the closest relative: orig_closest_idty - closest_idty.
static void show_dist_report(cseq &orig, cseq &aligned, search::result_vector &ref, log_printer::report &out) {
    auto report = [&](const char *what, float v) {
        char line[96];
        snprintf(line, sizeof line, "%s: %.6f\n", what, (double)v);
        out.text += line;
    };
    auto identity = [](const cseq &a, const cseq &b, CMP_IUPAC_TYPE rule) {
        return cseq_comparator(rule, CMP_DIST_NONE, CMP_COVER_QUERY, false)(a, b);
    };
    if (orig.getWidth() != aligned.getWidth()) {
        out.text += "Cannot show dist - " + orig.getName() + " and " + aligned.getName() + " have lengths " +
                    std::to_string(orig.getWidth()) + " and " + std::to_string(aligned.getWidth()) + "\n";
        return;
    }
    const float sps = identity(orig, aligned, CMP_IUPAC_EXACT);
    report("orig_idty", sps);
    out.has_sps = true;
    out.sps = sps;
    if (ref.empty()) {
        out.text += "reference / search result empty?\n";
        return;
    }
    // the closest relative of the input alignment: the last of the relatives sorted by identity
    // (std::sort on a copy, as the reference does: which of several equally close ones that is, is
    // the sort's business)
    search::result_vector by_identity(ref);
    for (search::result_item &r : by_identity) r.score = identity(orig, *r.sequence, CMP_IUPAC_OPTIMISTIC);
    std::sort(by_identity.begin(), by_identity.end());
    const search::result_item &closest = by_identity.back();
    const float before = closest.score, after = identity(aligned, *closest.sequence, CMP_IUPAC_OPTIMISTIC);
    report("orig_closest_idty", before);
    report("closest_idty", after);
    report("cpm", before - after);
    out.has_closest = true;
    out.idty = before;
    out.cpm = before - after;
}

// The report of one tray (src/log.cpp:364-430), as text + the figures the totals take: everything that does not
// depend on the trays before it, so that a batch's reports can be rendered side by side.
void log_printer::render(tray &t, report &out) const {
    if (t.input_sequence == nullptr) throw std::runtime_error("Received broken tray in log printer");
    out = report();
    std::string &log = out.text;
    log += "sequence_number: " + std::to_string(t.seqno) + "\n";
    log += "sequence_identifier: " + t.input_sequence->getName() + "\n";
    if (t.aligned_sequence == nullptr) {
        log += std::string(fn::align_log) + ": " + t.log_text() + "\n";
        log += std::string(fn::fullname) + ": " + t.input_sequence->get_attr<std::string>(fn::fullname) + "\n";
        log += "alignment failed!\n";
        return;
    }
    out.aligned = true;
    cseq &aligned = *t.aligned_sequence;
    // (helix pairing comes from the ARB database: no pairs, bp score 0)
    aligned.set_attr("align_bp_score_slv", 0);
    aligned.set_attr(fn::align_log, t.log_text());
    aligned.set_attr("nuc", (int)aligned.size());
    if (aligned.size() != 0u) {
        aligned.set_attr("align_startpos_slv", (int)aligned.begin()->getPosition());
        aligned.set_attr("align_stoppos_slv", (int)((--aligned.end())->getPosition()));
    } else {
        aligned.set_attr("align_startpos_slv", 0);
        aligned.set_attr("align_stoppos_slv", 0);
    }
    for (auto &ap : aligned.get_attrs()) {
        log += ap.key();
        log += ": ";
        log += attr_to_string(ap.second);
        log += "\n";
    }
    if (data->show_dist) {
        search::result_vector ref;
        if (t.search_result != nullptr) ref = *t.search_result;
        else if (t.alignment_reference != nullptr) ref = *t.alignment_reference;
        show_dist_report(*t.input_sequence, aligned, ref, out);
    }
}

// ... and what does: the count and the running totals, in tray order (the sums are doubles: their order shows)
void log_printer::commit(const report &r, std::ostream &log) {
    log << r.text;
    if (!r.aligned) return;
    ++data->sequence_num;
    if (r.has_sps) data->total_sps += r.sps;
    if (r.has_closest) {
        data->total_idty += r.idty;
        data->total_cpm += r.cpm;
    }
}

tray log_printer::operator()(tray t, std::ostream &log) {
    report r;
    render(t, r);
    commit(r, log);
    return t;
}

log_printer::summary log_printer::totals() const {  // ~priv_data, src/log.cpp:350-358
    summary s;
    s.sequences = data->sequence_num;
    const double n = data->sequence_num;
    s.avg_sps = data->total_sps / n;
    s.avg_cpm = data->total_cpm / n;
    s.avg_idty = data->total_idty / n;
    s.avg_bps = data->total_bps / n;
    return s;
}

}  // namespace sina
