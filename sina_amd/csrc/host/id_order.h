// Reference id order of an ARB database (second half).
//
// The reference numbers its reference sequences by walking
//   std::unordered_map<std::string, GBDATA*, boost::hash<std::string>>   (src/query_arb.cpp:160)
// which it filled name by name in database order (src/query_arb.cpp:470-474); the walk
// (query_arb::getSequenceNames, src/query_arb.cpp:732-739) is what kmer_search numbers the sequences by
// (src/kmer_search.cpp:248) and what a .sidx file lists.  Ids decide ties between equal k-mer scores, so
// tie-exact parity with a stock SINA binary -- and reading an index cache it wrote -- needs that order.
//
// Two third-party pieces decide it, neither under /root/reference:
//   * boost::hash<std::string> -- Boost >= 1.62 (configure.ac:238) and < 1.81: hash_range over the characters
//     with the 64-bit hash_combine (boost/container_hash/hash.hpp; Boost 1.81 replaced both).  Restated
//     below from the published algorithm.
//   * libstdc++'s std::unordered_map (node order and prime rehash policy).  Not restated: the host library
//     is built with the same libstdc++ a SINA build on this system would use, so it simply uses the same
//     container with the restated hash.
// Parity unpinned: the reference holds no vector for this order and Boost is not in the image.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace sina {

// boost::hash<std::string>()(s), Boost 1.62 ... 1.80, 64-bit std::size_t
std::size_t arb_name_hash(const std::string &s);

// order[i] = position in `names` (database order) of the sequence the reference would give id i.
// Names are taken as distinct (a repeated name keeps its first position, as operator[] does).
// bucket_count (optional) receives the table's final bucket count.
std::vector<uint32_t> arb_name_order(const std::vector<std::string> &names, std::size_t *bucket_count = nullptr);

}  // namespace sina
