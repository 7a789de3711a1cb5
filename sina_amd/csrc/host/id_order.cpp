#include "id_order.h"

#include <unordered_map>

namespace sina {

std::size_t arb_name_hash(const std::string &s) {
    // hash_range(first, last): seed = 0; for each element hash_combine(seed, hash_value(element));
    // hash_value(char) is the character converted to std::size_t (plain char: sign-extended);
    // hash_combine on a 64-bit seed mixes with the MurmurHash2 64-bit constants
    const uint64_t m = (uint64_t(0xc6a4a793u) << 32) + 0x5bd1e995u;
    uint64_t seed = 0;
    for (char ch : s) {
        uint64_t k = (uint64_t)(std::size_t)ch;
        k *= m;
        k ^= k >> 47;
        k *= m;
        seed ^= k;
        seed *= m;
        seed += 0xe6546b64u;  // ("completely arbitrary number, to prevent 0's from hashing to 0")
    }
    return (std::size_t)seed;
}

namespace {
struct name_hasher {
    std::size_t operator()(const std::string &s) const { return arb_name_hash(s); }
};
}  // namespace

std::vector<uint32_t> arb_name_order(const std::vector<std::string> &names, std::size_t *bucket_count) {
    std::unordered_map<std::string, uint32_t, name_hasher> by_name;  // (the reference's container, query_arb.cpp:160)
    for (uint32_t i = 0; i < names.size(); i++) {
        auto it = by_name.find(names[i]);
        if (it == by_name.end()) by_name[names[i]] = i;  // (operator[] as the reference: query_arb.cpp:473)
    }
    std::vector<uint32_t> order;
    order.reserve(by_name.size());
    for (const auto &kv : by_name) order.push_back(kv.second);
    if (bucket_count) *bucket_count = by_name.bucket_count();
    return order;
}

}  // namespace sina
