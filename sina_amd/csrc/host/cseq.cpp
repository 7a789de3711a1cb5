// Host-side cseq container: behaviour of src/cseq.cpp / src/aligned_base.cpp,
// implemented from the specs in SURVEY.md (A.4 "container rule", A.5 NAST
// fix-up).  Index-based, flat; no code shared with the reference.
#include "cseq.h"

#include <algorithm>

namespace sina {

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
    bases.clear();
    alignment_width = 0;
}

cseq_base &cseq_base::append(const char *str) {
    for (; *str; ++str) {
        const char c = *str;
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') continue;
        if (c != '-' && c != '.') bases.emplace_back(alignment_width, (unsigned char)c);
        alignment_width++;
    }
    return *this;
}

// Container rule: a base may share the column of its predecessor
// (insertions are resolved later) but never precede it.
cseq_base &cseq_base::append(const aligned_base &ab) {
    if (ab.getPosition() >= alignment_width) {
        bases.push_back(ab);
        alignment_width = ab.getPosition();
    } else {
        bases.emplace_back(alignment_width, ab.getBase());
    }
    return *this;
}

void cseq_base::setWidth(vidx_type newWidth) {
    if (bases.empty() || newWidth >= bases.back().getPosition() + 1) {
        alignment_width = newWidth;
        return;
    }
    if (newWidth < size()) throw std::runtime_error("Attempted to shrink alignment width below base count");
    // pack the right-most bases against the new right edge
    const unsigned int n = size();
    unsigned int skip = 0;
    while (skip < n && bases[n - skip - 1].getPosition() + skip >= newWidth) skip++;
    for (unsigned int i = skip; i > 0; --i) bases[n - i].setPosition(newWidth - i);
    alignment_width = newWidth;
}

void cseq_base::reverse() {
    std::reverse(bases.begin(), bases.end());
    for (auto &b : bases) b.setPosition(alignment_width - 1 - b.getPosition());
}
void cseq_base::complement() {
    for (auto &b : bases) b.complement();
}
void cseq_base::upperCaseAll() {
    for (auto &b : bases) b.setUpperCase();
}

std::string cseq_base::getAligned(bool nodots, bool dna) const {
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
    s.reserve(bases.size());
    for (const auto &b : bases) s.push_back((char)b.getBase().iupac_rna());
    return s;
}

char cseq_base::operator[](vidx_type i) const {
    auto it = std::lower_bound(bases.begin(), bases.end(), aligned_base(i, '.'));
    if (it != bases.end() && it->getPosition() == i) return (char)it->getBase().iupac_rna();
    return '-';
}

std::ostream &operator<<(std::ostream &out, const cseq_base &c) { return out << c.getName(); }

// NAST insertion fix-up (reference src/cseq.cpp:456-594).
//
// After backtracking, several bases may sit in one column (insertions relative
// to the reference).  Walk the bases; `anchor` is the last base whose column is
// final.  A run of bases sharing the anchor's column is spread over the free
// columns up to the next differently placed base (right-aligned).  If the free
// range is too small, the run swallows neighbouring bases, always towards the
// nearer free column, until it fits.
void cseq_base::fix_duplicate_positions(std::ostream &log, bool lowercase, bool remove) {
    idx_type total_inserts = 0, longest_insert = 0, orig_inserts = 0;
    if (remove) log << "insertion=remove not implemented, using shift; ";

    const long n = (long)bases.size();
    auto col = [&](long i) { return bases[(size_t)i].getPosition(); };
    long anchor = 0;
    for (long cur = 0; cur < n; ++cur) {
        if (col(anchor) == col(cur)) {
            if (cur + 1 != n) continue;  // still inside a run
            ++cur;                       // run reaches the end of the sequence
        }
        idx_type run = (idx_type)(cur - anchor - 1);
        if (run == 0) {
            anchor = cur;
            continue;
        }
        idx_type free_begin = col(anchor) + 1;                          // first free column
        idx_type free_end = (cur == n) ? alignment_width : col(cur);    // first taken column
        long first = anchor + 1, last = cur - 1;                        // the run [first, last]
        orig_inserts = run;
        if (free_end - free_begin < run) {
            log << "shifting bases to fit in " << run << " bases at pos " << free_begin << " to " << free_end
                << ";";
            while (free_end - free_begin < run) {
                // nearest free column to the left of the range (-1: none)
                int left_gap;
                long l = first;
                if (l == 0) {
                    left_gap = free_begin > 0 ? (int)(free_begin - 1) : -1;
                } else if (col(l - 1) + 1 < free_begin) {
                    left_gap = (int)(free_begin - 1);
                } else {
                    --l;
                    while (l != 0 && col(l - 1) + 1 >= col(l)) --l;
                    left_gap = (int)(col(l) - 1u);
                }
                // nearest free column to the right of the range (-1: none)
                int right_gap;
                long r = last;
                if (r + 1 == n) {
                    right_gap = free_end < alignment_width ? (int)free_end : -1;
                } else if (col(r + 1) > free_end) {
                    right_gap = (int)free_end;
                } else {
                    ++r;
                    while (r + 1 != n && col(r) + 1 >= col(r + 1)) ++r;
                    right_gap = (int)(col(r) + 1);
                }
                const bool go_left = right_gap == -1 ||
                                     (left_gap != -1 && (idx_type)(free_begin - (idx_type)left_gap) <=
                                                            (idx_type)((idx_type)right_gap - (free_end - 1)));
                if (go_left) {
                    if (left_gap == -1)
                        throw std::runtime_error("ERROR: no space to left and right?? sequence longe than alignment?!");
                    run += (idx_type)(first - l);
                    free_begin = (idx_type)left_gap;
                    first = l;
                } else {
                    run += (idx_type)(r - last);
                    free_end = (idx_type)right_gap + 1;
                    last = r;
                }
            }
        } else {
            free_begin = free_end - run;  // right-align inside the free range
        }
        for (long i = first; i <= last; ++i) {
            bases[(size_t)i].setPosition(free_begin++);
            if (lowercase) bases[(size_t)i].setLowerCase();
        }
        total_inserts += run;
        longest_insert = std::max(longest_insert, run);
        cur = last + 1;
        anchor = cur;
    }
    if (total_inserts > 0) {
        log << "total inserted bases=" << total_inserts << ";"
            << "longest insertion=" << longest_insert << ";"
            << "total inserted bases before shifting=" << orig_inserts << ";";
    }
}

}  // namespace sina
