// FASTA source / sink stages and the --show-dist accuracy metrics.
//
// Restates reference src/rw_fasta.cpp (reader::operator() :229-315, writer :332-541) and
// Log::printer (src/log.cpp:279-325 show_dist, :364-430 operator()) on std::fstream: name / full_name
// split, ";key=value" comment attributes, skipping of sequences with characters outside the IUPAC
// alphabet, --fasta-idx/--fasta-block slicing; on output the three text meta formats and the csv
// side file, dots vs dashes, DNA vs RNA, line wrapping, the --min-idty filter, --add-relatives.
// gzip in/out (boost::iostreams filters in the reference) is not provided.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <vector>
#include <sstream>
#include <stdexcept>
#include <unordered_set>

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

// ---------------------------------------------------------------- reader

// (aligned FASTA is 50 KB a sequence: with the streams' default 8 KB buffers every sequence costs
// half a dozen system calls each way)
static constexpr size_t kStreamBuffer = 4u << 20;

// gzip'ed files (reference: boost::iostreams gzip filters when the extension is ".gz",
// src/rw_fasta.cpp:200-202,358-360).  zlib is looked up at run time -- libz.so.1 is part of every
// base image, its development link is not, and nothing else of the library needs it.
namespace {
struct zlib_api {
    void *(*open)(const char *, const char *) = nullptr;
    int (*read)(void *, void *, unsigned) = nullptr;
    int (*write)(void *, const void *, unsigned) = nullptr;
    int (*close)(void *) = nullptr;
    int (*buffer)(void *, unsigned) = nullptr;
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
            if (!a.open || !a.read || !a.write || !a.close) throw std::runtime_error("zlib without the gz* functions");
            return a;
        }();
        return api;
    }
};
class gz_streambuf : public std::streambuf {
public:
    gz_streambuf(const std::string &path, bool writing) : z(zlib_api::get()), wr(writing), buf(kStreamBuffer) {
        f = z.open(path.c_str(), writing ? "wb" : "rb");
        if (f && z.buffer) z.buffer(f, 1u << 20);
        if (writing) setp(buf.data(), buf.data() + buf.size());
    }
    ~gz_streambuf() override {
        if (f) {
            if (wr) sync();
            z.close(f);
        }
    }
    bool is_open() const { return f != nullptr; }

protected:
    int_type underflow() override {
        if (wr || !f) return traits_type::eof();
        const int n = z.read(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) return traits_type::eof();
        setg(buf.data(), buf.data(), buf.data() + n);
        return traits_type::to_int_type(*gptr());
    }
    int_type overflow(int_type ch) override {
        if (!wr || !f || sync() != 0) return traits_type::eof();
        if (!traits_type::eq_int_type(ch, traits_type::eof())) {
            *pptr() = traits_type::to_char_type(ch);
            pbump(1);
        }
        return traits_type::not_eof(ch);
    }
    int sync() override {
        if (!wr || !f) return 0;
        const std::ptrdiff_t n = pptr() - pbase();
        if (n > 0 && z.write(f, pbase(), (unsigned)n) != (int)n) return -1;
        setp(buf.data(), buf.data() + buf.size());
        return 0;
    }

private:
    const zlib_api &z;
    void *f = nullptr;
    bool wr;
    std::vector<char> buf;
};
bool has_gz_extension(const std::string &path) { return path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0; }
}  // namespace

struct rw_fasta::reader::priv_data {
    std::vector<char> buf = std::vector<char>(kStreamBuffer);
    std::ifstream file;
    std::unique_ptr<gz_streambuf> gz;
    std::unique_ptr<std::istream> gz_in;
    std::istream *inp = nullptr;  // the plain file, or the gzip filter over it
    std::string filename;
    int lineno = 0, seqno = 0, skipped = 0;
};

rw_fasta::reader::reader(const std::string &infile) : data(new priv_data) {
    data->filename = infile;
    if (has_gz_extension(infile)) {
        if (fa_opts().fasta_block > 0) throw std::logic_error("Cannot use --fasta-idx with gzip'ed input");
        data->gz.reset(new gz_streambuf(infile, false));
        if (!data->gz->is_open()) throw std::runtime_error("Unable to open file \"" + infile + "\" for reading.");
        data->gz_in.reset(new std::istream(data->gz.get()));
        data->inp = data->gz_in.get();
        return;
    }
    data->file.rdbuf()->pubsetbuf(data->buf.data(), (std::streamsize)data->buf.size());  // (before open)
    data->file.open(infile, std::ios_base::binary);
    if (!data->file.is_open()) throw std::runtime_error("Unable to open file \"" + infile + "\" for reading.");
    data->inp = &data->file;
    if (fa_opts().fasta_block > 0) data->file.seekg(fa_opts().fasta_block * fa_opts().fasta_idx);
}
rw_fasta::reader::reader(const reader &) = default;
rw_fasta::reader &rw_fasta::reader::operator=(const reader &) = default;
rw_fasta::reader::~reader() = default;
int rw_fasta::reader::skipped() const { return data->skipped; }

static std::string trim(const std::string &s) {  // boost::trim
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

bool rw_fasta::reader::operator()(tray &t) {  // :229-315
    const options &o = fa_opts();
    std::istream &in = *data->inp;
    for (;;) {
        t.seqno = ++data->seqno;
        t.input_sequence = new cseq();
        cseq &c = *t.input_sequence;
        auto give_up = [&]() {
            delete t.input_sequence;
            t.input_sequence = nullptr;
            return false;
        };
        if (in.fail()) return give_up();
        // if fasta blocking enabled, check if we've passed block boundary in last sequence
        if (o.fasta_block > 0 && in.tellg() > o.fasta_block * (o.fasta_idx + 1)) return give_up();

        std::string line;
        // skip lines not beginning with '>'
        while (in.peek() != '>' && std::getline(in, line).good()) data->lineno++;

        // parse title
        data->lineno++;
        if (std::getline(in, line).good()) {
            if (!line.empty() && line[line.size() - 1] == '\r') line.resize(line.size() - 1);
            // set name to text between first '>' and first ' '
            unsigned int blank = (unsigned int)line.find_first_of(" \t");
            if (blank == 0) blank = (unsigned int)line.size();
            c.setName(line.substr(1, blank - 1));
            if (blank < line.size()) c.set_attr<std::string>(fn::fullname, line.substr(blank + 1));
        } else {  // didn't get a title
            return give_up();
        }

        // handle comments: "; key = value" becomes an attribute, others are ignored
        while (in.peek() == ';' && std::getline(in, line).good()) {
            data->lineno++;
            const size_t equalsign = line.find_first_of('=');
            if (equalsign != std::string::npos)
                c.set_attr(trim(line.substr(1, equalsign - 1)), trim(line.substr(equalsign + 1)));
        }

        try {
            // all lines until eof or next /^>/ are data
            while (in.peek() != '>' && in.good()) {
                std::getline(in, line);
                data->lineno++;
                c.append(line);
            }
        } catch (base_iupac::bad_character_exception &) {
            // "Skipping sequence N (>name) at file:line (contains character 'x')"
            while (in.peek() != '>' && std::getline(in, line).good()) data->lineno++;
            delete t.input_sequence;
            t.input_sequence = nullptr;
            data->skipped++;
            continue;  // (the reference recurses here)
        }
        return true;
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

static std::string escape_string(const std::string &in) {  // :378-392
    if (in.find_first_of("\",\r\n") == std::string::npos) return in;
    std::stringstream tmp;
    tmp << "\"";
    size_t j = 0;
    for (size_t i = in.find('"'); i != std::string::npos; j = i + 1, i = in.find('"', i + 1))
        tmp << in.substr(j, i - j) << "\"\"";
    tmp << in.substr(j) << "\"";
    return tmp.str();
}

struct rw_fasta::writer::priv_data {
    std::vector<char> buf = std::vector<char>(kStreamBuffer);
    std::ofstream file, out_csv;
    std::unique_ptr<gz_streambuf> gz;
    std::unique_ptr<std::ostream> gz_out;
    std::ostream *outp = nullptr;  // the plain file, or the gzip filter over it
    int count = 0, excluded = 0;
    std::unordered_set<std::string> relatives_written;
    unsigned long copy_relatives = 0;
    void write(const cseq &c);
};

rw_fasta::writer::writer(const std::string &outfile, unsigned int copy_relatives) : data(new priv_data) {
    data->copy_relatives = copy_relatives;
    if (has_gz_extension(outfile)) {
        data->gz.reset(new gz_streambuf(outfile, true));
        if (!data->gz->is_open()) throw std::runtime_error("Unable to open file \"" + outfile + "\" for writing.");
        data->gz_out.reset(new std::ostream(data->gz.get()));
        data->outp = data->gz_out.get();
    } else {
        data->file.rdbuf()->pubsetbuf(data->buf.data(), (std::streamsize)data->buf.size());  // (before open)
        data->file.open(outfile, std::ios_base::binary);
        if (!data->file.is_open()) throw std::runtime_error("Unable to open file \"" + outfile + "\" for writing.");
        data->outp = &data->file;
    }
    if (fa_opts().fastameta == FASTA_META_CSV) {
        const size_t dot = outfile.find_last_of('.'), slash = outfile.find_last_of('/');
        const std::string stem =
            (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? outfile.substr(0, dot) : outfile;
        data->out_csv.open(stem + ".csv");
        if (data->out_csv.fail()) throw std::runtime_error("Unable to open file \"" + outfile + ".csv\" for writing.");
    }
}
rw_fasta::writer::writer(const writer &) = default;
rw_fasta::writer &rw_fasta::writer::operator=(const writer &) = default;
rw_fasta::writer::~writer() = default;
int rw_fasta::writer::written() const { return data->count; }
int rw_fasta::writer::excluded() const { return data->excluded; }
void rw_fasta::writer::flush() {
    data->outp->flush();
    if (data->out_csv.is_open()) data->out_csv.flush();
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

void rw_fasta::writer::priv_data::write(const cseq &c) {  // :437-541
    const options &o = fa_opts();
    const auto &attrs = c.get_attrs();
    std::ostream &out = *outp;
    out << ">" << c.getName();
    const std::string fname = c.get_attr<std::string>(fn::fullname, "");
    if (!fname.empty()) out << " " << fname;
    switch (o.fastameta) {
    case FASTA_META_NONE: out << "\n"; break;
    case FASTA_META_HEADER:
        for (auto &ap : attrs) {
            if (ap.first == fn::family) continue;    // alignment family is too much
            if (ap.first == fn::fullname) continue;  // already written as description in header
            const std::string val = attr_to_string(ap.second);
            if (!val.empty()) out << " [" << ap.first << "=" << val << "]";
        }
        out << "\n";
        break;
    case FASTA_META_COMMENT:
        out << "\n";
        for (auto &ap : attrs) {
            if (ap.first == fn::family) continue;
            if (ap.first == fn::fullname) continue;
            out << "; " << ap.first << "=" << attr_to_string(ap.second) << "\n";
        }
        break;
    case FASTA_META_CSV:
        out << "\n";
        if (count == 0) {  // print header
            out_csv << "name";
            for (auto &ap : attrs) {
                if (ap.first == fn::family) continue;
                out_csv << "," << escape_string(ap.first);
            }
            out_csv << "\r\n";
        }
        out_csv << c.getName();
        for (auto &ap : attrs) {
            if (ap.first == fn::family) continue;
            out_csv << "," << escape_string(attr_to_string(ap.second));
        }
        out_csv << "\r\n";
        break;
    default: throw std::runtime_error("Unknown meta-fmt output option");
    }
    const std::string seq = c.getAligned(!o.out_dots, o.out_dna);
    const int len = (int)seq.size();
    if (o.line_length > 0) {
        for (int i = 0; i < len; i += o.line_length) {
            out.write(seq.data() + i, std::min(o.line_length, len - i));
            out.put('\n');
        }
    } else {
        out.write(seq.data(), len);
        out.put('\n');
    }
    count++;
}

// ---------------------------------------------------------------- Log::printer (--show-dist)

struct log_printer::priv_data {
    int sequence_num = 0;
    // data for computing alignment quality based on a reference (src/log.cpp:262-266)
    double total_sps = 0, total_cpm = 0, total_idty = 0, total_bps = 0;
    bool show_dist = false;
    void show(cseq &orig, cseq &aligned, search::result_vector &ref, std::ostream &log);
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
void log_printer::priv_data::show(cseq &orig, cseq &aligned, search::result_vector &ref, std::ostream &log) {
    auto report = [&](const char *what, float v) {
        char line[96];
        snprintf(line, sizeof line, "%s: %.6f\n", what, (double)v);
        log << line;
    };
    auto identity = [](const cseq &a, const cseq &b, CMP_IUPAC_TYPE rule) {
        return cseq_comparator(rule, CMP_DIST_NONE, CMP_COVER_QUERY, false)(a, b);
    };
    if (orig.getWidth() != aligned.getWidth()) {
        log << "Cannot show dist - " << orig.getName() << " and " << aligned.getName() << " have lengths "
            << orig.getWidth() << " and " << aligned.getWidth() << "\n";
        return;
    }
    const float sps = identity(orig, aligned, CMP_IUPAC_EXACT);
    report("orig_idty", sps);
    total_sps += sps;
    if (ref.empty()) {
        log << "reference / search result empty?\n";
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
    total_idty += before;
    report("orig_closest_idty", before);
    report("closest_idty", after);
    report("cpm", before - after);
    total_cpm += before - after;
}

tray log_printer::operator()(tray t, std::ostream &log) {  // src/log.cpp:364-430
    if (t.input_sequence == nullptr) throw std::runtime_error("Received broken tray in log printer");
    log << "sequence_number: " << t.seqno << "\n";
    log << "sequence_identifier: " << t.input_sequence->getName() << "\n";
    if (t.aligned_sequence == nullptr) {
        log << fn::align_log << ": " << t.log.str() << "\n";
        log << fn::fullname << ": " << t.input_sequence->get_attr<std::string>(fn::fullname) << "\n";
        log << "alignment failed!\n";
        return t;
    }
    ++data->sequence_num;
    cseq &aligned = *t.aligned_sequence;
    // (helix pairing comes from the ARB database: no pairs, bp score 0)
    aligned.set_attr("align_bp_score_slv", 0);
    aligned.set_attr(fn::align_log, t.log.str());
    aligned.set_attr("nuc", (int)aligned.size());
    if (aligned.size() != 0u) {
        aligned.set_attr("align_startpos_slv", (int)aligned.begin()->getPosition());
        aligned.set_attr("align_stoppos_slv", (int)((--aligned.end())->getPosition()));
    } else {
        aligned.set_attr("align_startpos_slv", 0);
        aligned.set_attr("align_stoppos_slv", 0);
    }
    for (auto &ap : aligned.get_attrs()) log << ap.first << ": " << attr_to_string(ap.second) << "\n";
    search::result_vector ref;
    if (t.search_result != nullptr) ref = *t.search_result;
    else if (t.alignment_reference != nullptr) ref = *t.alignment_reference;
    if (data->show_dist) data->show(*t.input_sequence, aligned, ref, log);
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
