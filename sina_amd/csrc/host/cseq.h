// Host-side sequence container with the interface SINA's stages use
// (reference: src/aligned_base.h, src/cseq.h, src/cseq.cpp).  Written from the
// behavioural spec in SURVEY.md (sections 8a, A.4-A.6); the packed layout of an
// aligned base is the reference's own (aligned_compact, src/aligned_base.h:287-319)
// because that is what crosses the C ABI.
#pragma once

#include <cstdint>
#include <map>
#include <new>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <variant>
#include <vector>

namespace sina {

// The three places where a query changes clothes between iupac byte masks (what the device takes) and packed
// aligned_base words (column | mask << 24, what a cseq holds): tray construction, famfinder and aligner packing.
// Plain loops over restrict pointers, cloned for AVX2: a uint8_t store may alias anything, and inside the
// stages' lambdas the compiler kept these scalar -- 2-3 us per 1500-base query each, 7 us of a query's 14.
__attribute__((target_clones("avx2", "default"))) inline void masks_of_packed(uint8_t *__restrict dst, const uint32_t *__restrict src, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (uint8_t)(src[i] >> 24);
}
__attribute__((target_clones("avx2", "default"))) inline void packed_of_masks(uint32_t *__restrict dst, const uint8_t *__restrict src, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (uint32_t)i | ((uint32_t)src[i] << 24);
}


// A copy whose destination nobody reads soon (a finished alignment on its way to a sink): non-temporal stores, so that
// the destination's lines are not first read into the cache to be overwritten -- a third of the copy's memory traffic --
// and do not push anything out of it.  Plain memcpy where AVX2 is missing or the block is small.
void stream_copy(void *dst, const void *src, size_t bytes);

enum base_types { BASE_A = 0, BASE_G = 1, BASE_C = 2, BASE_TU = 3, BASE_MAX = 4, BASE_LC = 4 };

// IUPAC base as a bit mask: A 1, G 2, C 4, T/U 8, lower-case 16 (src/aligned_base.h:47-52).
class base_iupac {
public:
    using value_type = unsigned char;
    class bad_character_exception : public std::exception {
    public:
        explicit bad_character_exception(value_type c) noexcept : character(c) {}
        const char *what() const noexcept override { return "Character not IUPAC encoded base or gap"; }
        value_type character;
    };
    base_iupac() = default;
    base_iupac(unsigned char c) : _data(from_char(c)) {}
    static base_iupac from_mask(value_type m) {
        base_iupac b;
        b._data = m;
        return b;
    }
    static value_type from_char(unsigned char c);  // throws bad_character_exception
    operator unsigned char() const { return iupac_rna(); }
    unsigned char iupac_rna() const;
    unsigned char iupac_dna() const;
    value_type mask() const { return _data; }
    base_types getBaseType() const { return static_cast<base_types>(__builtin_ctz(_data & 0xf)); }
    base_iupac &complement();
    base_iupac &setLowerCase() { _data |= 16; return *this; }
    base_iupac &setUpperCase() { _data &= ~16; return *this; }
    bool isLowerCase() const { return (_data & 16) != 0; }
    int ambig_order() const { return __builtin_popcount(_data & 0xf); }
    bool is_ambig() const { return ambig_order() > 1; }
    bool comp(const base_iupac &rhs) const { return (0xf & _data & rhs._data) != 0; }
    bool comp_pessimistic(const base_iupac &rhs) const { return !is_ambig() && (0xf & _data) == (0xf & rhs._data); }
    bool comp_exact(const base_iupac &rhs) const { return (0xf & _data) == (0xf & rhs._data); }

private:
    value_type _data{0};
};

// 24-bit column + 8-bit base in one uint32 -- identical bits to the reference's
// aligned_compact<base_iupac>, so vectors of these can be handed to the C ABI.
class aligned_base {
public:
    using idx_type = uint32_t;
    using base_type = base_iupac;
    aligned_base() = default;  // (trivial: `aligned_base()` / `{}` is column 0 without a base, a default-initialised one is not set at all)
    explicit aligned_base(idx_type pos, unsigned char c = '-') : raw((pos & 0xFFFFFFu) | ((uint32_t)base_iupac::from_char(c) << 24)) {}
    aligned_base(idx_type pos, base_iupac b) : raw((pos & 0xFFFFFFu) | ((uint32_t)b.mask() << 24)) {}
    static aligned_base from_raw(uint32_t r) {
        aligned_base a;
        a.raw = r;
        return a;
    }
    base_iupac getBase() const { return base_iupac::from_mask((unsigned char)(raw >> 24)); }
    void setBase(const base_iupac &b) { raw = (raw & 0xFFFFFFu) | ((uint32_t)b.mask() << 24); }
    idx_type getPosition() const { return raw & 0xFFFFFFu; }
    void setPosition(idx_type pos) { raw = (pos & 0xFFFFFFu) | (raw & 0xFF000000u); }
    void setLowerCase() { raw |= (uint32_t)16 << 24; }
    void setUpperCase() { raw &= ~((uint32_t)16 << 24); }
    void complement() {
        base_iupac b = getBase();
        b.complement();
        setBase(b);
    }
    bool operator<(const aligned_base &rhs) const { return getPosition() < rhs.getPosition(); }
    bool operator==(const aligned_base &rhs) const { return raw == rhs.raw; }
    bool operator!=(const aligned_base &rhs) const { return raw != rhs.raw; }
    uint32_t raw;
};
static_assert(sizeof(aligned_base) == 4, "aligned_base must stay 4 bytes");

// The heap blocks of base lists.  A pipeline moves a 6 KB block per query from stage to stage and thread to thread;
// out of the general allocator that was, per query, a page fault or two (fresh top-of-heap memory, 4 KB pages) and a
// zero fill by resize() before the real contents arrived.  Base lists therefore come out of a process-wide pool of
// blocks in power-of-two size classes (4 KB .. 64 KB; larger ones from malloc), carved from 2 MB regions that ask for
// transparent huge pages, kept on per-thread free lists with a shared depot behind them; and resize() leaves new
// elements as they are (every writer fills what it sizes).
void *base_block_alloc(size_t bytes);
void base_block_free(void *p, size_t bytes);
template <class T> struct base_block_allocator {
    using value_type = T;
    base_block_allocator() = default;
    template <class U> base_block_allocator(const base_block_allocator<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(base_block_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { base_block_free(p, n * sizeof(T)); }
    template <class U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }  // (default-, not value-initialised)
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) {
        ::new (static_cast<void *>(p)) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
    template <class U> bool operator==(const base_block_allocator<U> &) const { return true; }
    template <class U> bool operator!=(const base_block_allocator<U> &) const { return false; }
};
using base_vector = std::vector<aligned_base, base_block_allocator<aligned_base>>;

class cseq_base {
public:
    using idx_type = unsigned int;
    using vidx_type = aligned_base::idx_type;
    using value_type = aligned_base;
    using iterator = base_vector::iterator;
    using const_iterator = base_vector::const_iterator;
    using const_reverse_iterator = base_vector::const_reverse_iterator;

    cseq_base() = default;
    cseq_base(const char *_name, const char *_data = nullptr);

    void clearSequence();
    cseq_base &append(const char *str);
    cseq_base &append(const std::string &str) { return append(str.c_str()); }
    cseq_base &append(const aligned_base &ab);

    // ---- an UNALIGNED sequence kept as its iupac masks alone ("dense": base i sits in column i, the width is the
    // number of bases -- what a reader hands the pipeline before anything has aligned it).  One byte per base, the
    // byte the device takes, instead of a packed word: building the words (6 KB per 16S query), and famfinder and
    // the aligner each reading them back to get the bytes again, were three of the five times a query's 6 KB went
    // through a core.  The packed words are made when somebody asks for them (getAlignedBases, packed, iterators: a
    // const call that writes -- a sequence is touched by one thread at a time on its way through the stages, the
    // store's references are never dense); anything that changes the sequence makes them first and drops the bytes.
    void setDenseMasks(const uint8_t *masks, size_t n) {
        bases.clear();
        dmask.reserve((n + 1023) & ~(size_t)1023);  // (in steps of 1 KB: a recycled sequence's block fits its next occupant)
        dmask.assign(masks, masks + n);
        unpacked = true;
        alignment_width = (vidx_type)n;
    }
    // the mask bytes if the sequence still is what setDenseMasks made it, else nullptr
    const uint8_t *denseMasks() const { return dmask.empty() ? nullptr : dmask.data(); }

    vidx_type size() const { return (vidx_type)(unpacked ? dmask.size() : bases.size()); }
    const base_vector &getAlignedBases() const { unpack(); return bases; }
    void setAlignedBases(const base_vector &vab) { drop_masks(); bases = vab; }
    void setAlignedBases(base_vector &&vab) { drop_masks(); bases = std::move(vab); }
    void setAlignedBases(const aligned_base *first, size_t n) { drop_masks(); bases.assign(first, first + n); }
    base_vector takeAlignedBases() { touch(); return std::move(bases); }  // leaves the sequence empty
    // the base list itself, for writing it in place (a recycled sequence keeps its heap block: resize, fill)
    base_vector &mutableAlignedBases() { touch(); return bases; }
    vidx_type getWidth() const { return alignment_width; }
    void setWidth(vidx_type newWidth);
    void fix_duplicate_positions(std::ostream &log, bool lowercase, bool remove);
    static void sort() {}
    void reverse();
    void complement();
    void upperCaseAll();
    std::string getAligned(bool nodots = false, bool dna = false) const;
    std::string getBases() const;
    std::string getName() const { return name; }
    const std::string &name_ref() const { return name; }
    void setName(std::string n) { name = std::move(n); }
    void setName(std::string_view n) { name.assign(n); }

    iterator begin() { touch(); return bases.begin(); }
    const_iterator begin() const { unpack(); return bases.begin(); }
    iterator end() { touch(); return bases.end(); }
    const_iterator end() const { unpack(); return bases.end(); }
    const_reverse_iterator rbegin() const { unpack(); return bases.rbegin(); }
    const_reverse_iterator rend() const { unpack(); return bases.rend(); }
    const aligned_base &getById(idx_type i) const { unpack(); return bases[i]; }
    char operator[](vidx_type i) const;

    bool operator==(const cseq_base &rhs) const { return name == rhs.name && getAlignedBases() == rhs.getAlignedBases(); }
    bool operator!=(const cseq_base &rhs) const { return !(*this == rhs); }
    bool operator<(const cseq_base &rhs) const { return name < rhs.name; }

    // raw view for the C ABI
    const uint32_t *packed() const { unpack(); return reinterpret_cast<const uint32_t *>(bases.data()); }

private:
    void unpack() const {  // the packed words of a dense sequence, made on first use (the bytes stay valid)
        if (!unpacked) return;
        bases.resize(dmask.size());
        packed_of_masks(reinterpret_cast<uint32_t *>(bases.data()), dmask.data(), dmask.size());
        unpacked = false;
    }
    void drop_masks() {
        dmask.clear();
        unpacked = false;
    }
    void touch() {  // before any change: the words exist, the bytes no longer describe the sequence
        unpack();
        dmask.clear();
    }
    std::string name;
    mutable base_vector bases;
    std::vector<uint8_t> dmask;     // iupac masks of a dense sequence (empty: not dense, or changed since)
    mutable bool unpacked{false};   // dense and `bases` not made yet
    unsigned int alignment_width{0};
};

std::ostream &operator<<(std::ostream &out, const cseq_base &c);

// attribute map of annotated_cseq (src/cseq.h:233-272); boost::variant -> std::variant.
// The reference keeps a std::map<string, variant>.  Here: a short vector of {key, value} sorted by key
// (same iteration order), the keys INTERNED -- a sequence has half a dozen attributes out of a few dozen
// distinct names, a map costs two heap blocks per attribute (node + key longer than 15 characters) and
// the pipeline creates and destroys a dozen of them per query.
class annotated_cseq : public cseq_base {
public:
    using variant = std::variant<std::string, char, int, float>;
    struct attr {
        const std::string *name;  // interned: lives as long as the process
        variant second;
        const std::string &key() const { return *name; }
    };
    using attr_list = std::vector<attr>;
    // A string attribute as the LIST its text is made from, the text made when somebody reads it.  famfinder's
    // align_family_slv is forty "<acc>.<start>:<score> " pieces per query (src/famfinder.cpp:458-470) that a run
    // writes out only if the field is asked for: composing it for every query was the largest single item of the
    // host's time per query.  `items` and `owner` mean what `render` takes them to mean; reading the attribute any
    // way (get_attr, string_attr, has_attr, get_attrs) renders it into its slot first.
    struct lazy_text {
        const std::string *name = nullptr;  // interned key of the attribute it stands for (nullptr: none pending)
        std::vector<uint64_t> items;
        const void *owner = nullptr;
        void (*render)(const void *owner, const uint64_t *items, size_t n, std::string &out) = nullptr;
    };

    annotated_cseq(const char *_name, const char *_data = nullptr) : cseq_base(_name, _data) {}
    annotated_cseq() = default;
    // name, width and attributes of `o` WITHOUT its bases (the aligner's working copy: its bases are
    // written once, when the alignment is known -- copying the query's first would be 6 KB for nothing)
    struct meta_only {};
    annotated_cseq(const annotated_cseq &o, meta_only) {
        setName(o.getName());
        setWidth(o.getWidth());
        assign_attrs(o.attributes);
        copy_lazy(o);
    }
    // An attribute to be: interned key (attr_key) and an int, float or string value
    struct attr_init {
        const std::string *name;
        enum kind_t { k_int, k_float, k_string } kind;
        int i;
        float f;
        std::string_view s;
        static attr_init of(const std::string *n, int v) { return attr_init{n, k_int, v, 0.f, {}}; }
        static attr_init of(const std::string *n, float v) { return attr_init{n, k_float, 0, v, {}}; }
        static attr_init of(const std::string *n, std::string_view v) { return attr_init{n, k_string, 0, 0.f, v}; }
    };
    static const std::string *attr_key(std::string_view key) { return interned(key); }
    // copy_meta(o) plus the attributes `extra` (ascending by key; one of o's with the same key is replaced), the
    // list built in one pass in key order: the aligner's working copy with its half dozen result attributes --
    // set one by one they were half a dozen searches and insertions in the middle of a list of strings
    void copy_meta_with(const annotated_cseq &o, const attr_init *extra, size_t n_extra) {
        clearSequence();
        setName(std::string_view(o.name_ref()));
        setWidth(o.getWidth());
        retire_strings();
        attributes.clear();
        attributes.reserve(o.attributes.size() + n_extra);
        size_t e = 0;
        auto put_extra = [&](const attr_init &x) {
            if (x.kind == attr_init::k_int) attributes.push_back(attr{x.name, variant(x.i)});
            else if (x.kind == attr_init::k_float) attributes.push_back(attr{x.name, variant(x.f)});
            else {
                std::string mine = fresh_string();
                mine.assign(x.s);
                attributes.push_back(attr{x.name, variant(std::move(mine))});
            }
        };
        for (const attr &a : o.attributes) {
            while (e < n_extra && *extra[e].name < *a.name) put_extra(extra[e++]);
            if (e < n_extra && (extra[e].name == a.name || *extra[e].name == *a.name)) {
                put_extra(extra[e++]);
                continue;
            }
            if (const std::string *str = std::get_if<std::string>(&a.second)) {
                std::string mine = fresh_string();
                mine.assign(*str);
                attributes.push_back(attr{a.name, variant(std::move(mine))});
            } else {
                attributes.push_back(a);
            }
        }
        while (e < n_extra) put_extra(extra[e++]);
        copy_lazy(o);
        for (size_t x = 0; x < n_extra; x++)
            if (lazy.name && lazy.name == extra[x].name) lazy.name = nullptr;  // (replaced)
    }
    void copy_meta(const annotated_cseq &o) {  // the same into an existing (recycled) object
        clearSequence();
        setName(std::string_view(o.name_ref()));
        setWidth(o.getWidth());
        assign_attrs(o.attributes);
        copy_lazy(o);
    }
    // Back to the default-constructed state, keeping the heap blocks -- the base list's and those of the
    // attribute strings (family list, date ...: half a dozen blocks per sequence that one pool thread
    // allocated and another one frees; with a dozen threads doing so the allocator's arena locks were 8 us
    // per query of the V4 shape).  The strings wait in `spare` for the object's next life.
    void clear_all() {
        clearSequence();
        setName(std::string());
        retire_strings();
        attributes.clear();
        lazy.name = nullptr;
        lazy.items.clear();
    }

    // `key` becomes a string attribute whose text `render` makes from the returned list (to be filled by the caller)
    // when it is first read; one such attribute per sequence (a second one settles the first)
    std::vector<uint64_t> &set_lazy_attr(std::string_view key, const void *owner,
                                         void (*render)(const void *, const uint64_t *, size_t, std::string &)) {
        settle();
        string_slot(key);  // (the slot exists, in key order, empty until read)
        lazy.name = interned(key);
        lazy.owner = owner;
        lazy.render = render;
        lazy.items.clear();
        return lazy.items;
    }
    // the list behind `key` if its text has not been made yet (a sink that keeps the list instead of the text)
    const lazy_text *lazy_attr(std::string_view key) const { return lazy.name && *lazy.name == key ? &lazy : nullptr; }

    template <typename T> void set_attr(std::string_view key, T val) {
        if constexpr (std::is_same<T, std::string>::value) string_slot(key).assign(val);
        else slot(key) = variant(std::move(val));
    }
    void set_attr(std::string_view key, const std::string &val) { string_slot(key).assign(val); }
    void set_attr(std::string_view key, const char *val) { string_slot(key).assign(val); }
    void set_attr(std::string_view key, std::string_view val) { string_slot(key).assign(val); }
    // the string value of `key`, EMPTY, to be written in place (made a string attribute if it is none yet);
    // its heap block is a recycled one where the object has any
    std::string &string_slot(std::string_view key) {
        variant &v = slot(key);
        if (std::string *s = std::get_if<std::string>(&v)) {
            if (s->capacity() <= 15 && !spare.empty()) {  // (a fresh slot holds an empty, block-less string)
                *s = std::move(spare.back());
                spare.pop_back();
            }
            s->clear();
            return *s;
        }
        v = variant(fresh_string());
        return std::get<std::string>(v);
    }
    bool has_attr(std::string_view key) const { return find(key) != nullptr; }
    // the value of a string attribute in place (nullptr: absent or not a string) -- get_attr<std::string> copies
    const std::string *string_attr(std::string_view key) const {
        const variant *v = find(key);
        return v ? std::get_if<std::string>(v) : nullptr;
    }
    template <typename T> T get_attr(std::string_view attr) const { return get_attr<T>(attr, T()); }
    template <typename T> T get_attr(std::string_view attr, T dflt) const {
        const variant *v = find(attr);
        if (v == nullptr) return dflt;
        return std::visit([&](const auto &x) { return convert<T>(x); }, *v);
    }
    const attr_list &get_attrs() const { settle(); return attributes; }
    // the same without making a pending lazy text (its slot reads as an empty string): for a reader that looks for
    // other keys
    const attr_list &attrs_unsettled() const { return attributes; }

private:
    static const std::string *interned(std::string_view key);
    void settle() const {  // the pending text, made (const: the attribute's VALUE does not change, its form does)
        if (!lazy.name) return;
        for (attr &a : attributes)
            if (a.name == lazy.name) {
                if (std::string *s = std::get_if<std::string>(&a.second)) lazy.render(lazy.owner, lazy.items.data(), lazy.items.size(), *s);
                break;
            }
        lazy.name = nullptr;
    }
    void copy_lazy(const annotated_cseq &o) {
        lazy.name = o.lazy.name;
        lazy.owner = o.lazy.owner;
        lazy.render = o.lazy.render;
        if (o.lazy.name) lazy.items.assign(o.lazy.items.begin(), o.lazy.items.end());
        else lazy.items.clear();
    }
    const variant *find(std::string_view key) const {
        if (lazy.name && *lazy.name == key) settle();
        for (const attr &a : attributes)
            if (*a.name == key) return &a.second;
        return nullptr;
    }
    variant &slot(std::string_view key);  // the value of `key`, inserted in key order if new (cseq.cpp); settles a pending text of that key
    std::string fresh_string() {
        if (spare.empty()) return std::string();
        std::string s = std::move(spare.back());
        spare.pop_back();
        return s;
    }
    void retire_strings() {
        for (attr &a : attributes)
            if (std::string *s = std::get_if<std::string>(&a.second))
                if (s->capacity() > 15 && spare.size() < 16) {
                    s->clear();
                    spare.push_back(std::move(*s));
                }
    }
    void assign_attrs(const attr_list &from) {  // attributes = from, into recycled string blocks
        retire_strings();
        attributes.clear();
        attributes.reserve(from.size());
        for (const attr &a : from) {
            if (const std::string *s = std::get_if<std::string>(&a.second)) {
                std::string mine = fresh_string();
                mine.assign(*s);
                attributes.push_back(attr{a.name, variant(std::move(mine))});
            } else {
                attributes.push_back(a);
            }
        }
    }
    template <typename T, typename S> static T convert(const S &s) {
        if constexpr (std::is_same<T, S>::value) {
            return s;
        } else {  // boost::lexical_cast semantics: stream out, stream in, default on failure
            std::stringstream ss;
            ss << s;
            T t{};
            if constexpr (std::is_same<T, std::string>::value) {
                return ss.str();
            } else {
                ss >> t;
                if (ss.fail()) return T();
                return t;
            }
        }
    }
    mutable attr_list attributes;
    mutable lazy_text lazy;
    std::vector<std::string> spare;  // emptied strings of earlier lives, with their heap blocks
};

typedef annotated_cseq cseq;

// field names used by the stages (src/query_arb.cpp:83-107)
namespace fn {
extern const char *turn, *acc, *start, *cutoff_head, *cutoff_tail, *date, *qual, *head, *tail, *idty, *family,
    *filter, *align_log, *used_rels, *fullname;
}

}  // namespace sina
