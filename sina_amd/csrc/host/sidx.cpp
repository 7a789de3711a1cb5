// The reference's on-disk k-mer index cache (".sidx") <-> the CSR index the GPU uses.
//
// File format, as kmer_search::impl::store / try_load write and read it
// (reference src/kmer_search.cpp:66-88,279-351) with vlimap::write / read (src/idset.h:386-410):
//   idx_header  { u64 magic "SINAKIDX"; u16 vers = 0; [2 pad]; u32 n_sequences;
//                 u16 flags = k | nofast << 8; [6 pad] }                      24 bytes
//   names       n_sequences lines, '\n' terminated
//   emptymap    one vlimap of the k-mer numbers that have a list
//   lists       one vlimap per such k-mer, ascending k-mer number
//   vlimap      { u32 inc (1, or 0xFFFFFFFF = -1 for an inverted list); u32 last; u32 bytesize;
//                 u32 size } + bytesize bytes: the ascending ids as deltas, 7 bits per byte, low
//                 group first, bit 7 = "more follows" (vlimap_abs::push_back, idset.h:279-286)
// A list longer than n_sequences / 2 is stored INVERTED: the ids NOT containing the k-mer
// (kmer_search.cpp:263-265, vlimap::invert idset.h:367-384).  invert() does not swap _size, so the
// `size` field of an inverted list still holds the number of sequences that DO contain the k-mer;
// `last` is the last id of the stored (inverted) list.
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "stages.h"

namespace sina {

namespace {
constexpr uint64_t kMagic = 0x5844494b414e4953ull;  // SINAKIDX

struct vl_list {  // one vlimap as it sits in the file
    uint32_t inc = 1, last = 0, size = 0;
    std::vector<uint8_t> data;
    void push_delta(uint32_t n) {
        while (n > 127) {
            data.push_back((uint8_t)(n | 0x80));
            n >>= 7;
        }
        data.push_back((uint8_t)n);
    }
    void write(FILE *f) const {
        const uint32_t head[4] = {inc, last, (uint32_t)data.size(), size};
        if (fwrite(head, 4, 4, f) != 4 || (data.size() && fwrite(data.data(), 1, data.size(), f) != data.size()))
            throw std::runtime_error("sidx: write failed");
    }
    bool read(FILE *f) {
        uint32_t head[4];
        if (fread(head, 4, 4, f) != 4) return false;
        inc = head[0];
        last = head[1];
        size = head[3];
        data.resize(head[2]);
        return data.empty() || fread(data.data(), 1, data.size(), f) == data.size();
    }
    // the stored ids (deltas summed up), appended to out
    void decode(std::vector<uint32_t> &out) const {
        uint32_t cur = 0;
        for (size_t i = 0; i < data.size();) {
            uint32_t val = data[i] & 0x7f, shift = 7;
            while (data[i] & 0x80) {
                ++i;
                if (i >= data.size()) throw std::runtime_error("sidx: truncated list");
                val |= (uint32_t)(data[i] & 0x7f) << shift;
                shift += 7;
            }
            ++i;
            cur += val;
            out.push_back(cur);
        }
    }
};
}  // namespace

void sidx_store(const std::string &path, unsigned k, bool nofast, const std::vector<std::string> &names,
                const std::vector<uint32_t> &offsets, const std::vector<uint32_t> &ids) {
    const uint32_t n_sequences = (uint32_t)names.size();
    const size_t n_kmers = (size_t)1 << (2 * k);
    if (offsets.size() != n_kmers + 1) throw std::logic_error("sidx_store: offsets do not fit k");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("sidx: cannot write " + path);
    try {
        unsigned char header[24];
        memset(header, 0, sizeof header);
        const uint16_t vers = 0, flags = (uint16_t)((k & 0xff) | ((nofast ? 1 : 0) << 8));
        memcpy(header, &kMagic, 8);
        memcpy(header + 8, &vers, 2);
        memcpy(header + 12, &n_sequences, 4);
        memcpy(header + 16, &flags, 2);
        if (fwrite(header, 1, 24, f) != 24) throw std::runtime_error("sidx: write failed");
        for (const auto &n : names) {
            fputs(n.c_str(), f);
            fputc('\n', f);
        }
        vl_list emptymap;
        uint32_t prev = 0;
        for (size_t km = 0; km < n_kmers; km++) {
            if (offsets[km + 1] == offsets[km]) continue;
            emptymap.push_delta((uint32_t)km - prev);
            prev = (uint32_t)km;
            emptymap.size++;
        }
        emptymap.last = prev;
        emptymap.write(f);
        vl_list l;
        for (size_t km = 0; km < n_kmers; km++) {
            const uint32_t b = offsets[km], e = offsets[km + 1];
            if (b == e) continue;
            l.data.clear();
            l.size = e - b;  // (also for inverted lists: invert() keeps _size)
            uint32_t last = 0;
            if (l.size > n_sequences / 2) {  // stored inverted: the ids in between
                l.inc = 0xFFFFFFFFu;
                uint32_t next_id = 0;
                for (uint32_t x = b; x <= e; x++) {
                    const uint32_t stop = (x < e) ? ids[x] : n_sequences;
                    for (; next_id < stop; next_id++) {
                        l.push_delta(next_id - last);
                        last = next_id;
                    }
                    next_id = stop + 1;
                }
            } else {
                l.inc = 1;
                for (uint32_t x = b; x < e; x++) {
                    l.push_delta(ids[x] - last);
                    last = ids[x];
                }
            }
            l.last = last;
            l.write(f);
        }
    } catch (...) {
        fclose(f);
        throw;
    }
    fclose(f);
}

bool sidx_load(const std::string &path, unsigned k, bool nofast, std::vector<std::string> *names,
               std::vector<uint32_t> *offsets, std::vector<uint32_t> *ids, std::string *why) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        if (why) *why = "cannot open";
        return false;
    }
    auto fail = [&](const char *msg) {
        if (why) *why = msg;
        fclose(f);
        return false;
    };
    unsigned char header[24];
    if (fread(header, 1, 24, f) != 24) return fail("short header");
    uint64_t magic;
    uint16_t vers, flags;
    uint32_t n_sequences;
    memcpy(&magic, header, 8);
    memcpy(&vers, header + 8, 2);
    memcpy(&n_sequences, header + 12, 4);
    memcpy(&flags, header + 16, 2);
    if (magic != kMagic) return fail("wrong magic");                        // (the reference exits here)
    if (vers != 0) return fail("created by different version");
    if ((flags & 0xff) != k) return fail("built for a different k");
    if (((flags >> 8) & 1) != (nofast ? 1u : 0u)) return fail("built for a different fast/no-fast setting");
    names->clear();
    std::string line;
    for (uint32_t i = 0; i < n_sequences; i++) {
        line.clear();
        int ch;
        while ((ch = fgetc(f)) != EOF && ch != '\n') line += (char)ch;
        if (ch == EOF) return fail("truncated names");
        names->push_back(line);
    }
    try {
        vl_list emptymap;
        if (!emptymap.read(f)) return fail("truncated k-mer map");
        std::vector<uint32_t> kmers;
        emptymap.decode(kmers);
        const size_t n_kmers = (size_t)1 << (2 * k);
        offsets->assign(n_kmers + 1, 0);
        ids->clear();
        std::vector<uint32_t> stored;
        vl_list l;
        size_t next_km = 0;
        for (uint32_t km : kmers) {
            if (km >= n_kmers) return fail("k-mer number out of range");
            for (; next_km <= km; next_km++) (*offsets)[next_km] = (uint32_t)ids->size();
            if (!l.read(f)) return fail("truncated list");
            stored.clear();
            l.decode(stored);
            if (l.inc == 1) {
                ids->insert(ids->end(), stored.begin(), stored.end());
            } else {  // inverted: every id that is NOT stored
                size_t x = 0;
                for (uint32_t id = 0; id < n_sequences; id++) {
                    if (x < stored.size() && stored[x] == id) x++;
                    else ids->push_back(id);
                }
            }
        }
        for (; next_km <= n_kmers; next_km++) (*offsets)[next_km] = (uint32_t)ids->size();
    } catch (const std::exception &e) {
        return fail("corrupt list");
    }
    fclose(f);
    return true;
}

}  // namespace sina
