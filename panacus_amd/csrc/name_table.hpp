// name_table.hpp -- node2id on the device: segment names that are not numbers.
//
// The reference keeps `HashMap<Vec<u8>, ItemId>` (src/graph_broker/graph.rs:308-375) and asks it once per path step
// (graph.rs:231).  Here: an open-addressing table in HBM keyed by the name BYTES themselves -- up to 16, zero padded, so the key
// is the name and a hit needs no second look at the text -- one 32-byte entry per slot (key and id in one sector: a probe is one
// memory access).  Built by one thread per S line, verified by a second pass (every name must find ITS OWN id: a name that
// occurs twice finds the other one -- the reference panics there, graph.rs:336), read by the tokeniser and the L-line parser.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pnx {

struct NameEntry {
    unsigned long long k0, k1;  // the name, bytes 0..7 and 8..15, zero padded
    uint32_t id;                // 0 = empty
    uint32_t pad[3];
};
static_assert(sizeof(NameEntry) == 32, "one entry = half a 64-byte sector");

struct NameTab {
    NameEntry *e = nullptr;
    uint64_t mask = 0;  // slots - 1
};

__device__ static inline uint64_t name_hash(unsigned long long k0, unsigned long long k1) {
    uint64_t h = k0 * 0x9E3779B97F4A7C15ull;
    h ^= (k1 + 0x7F4A7C159E3779B9ull) * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    return h ^ (h >> 29);
}

// id of the name (0: the graph has no such segment)
__device__ static inline uint32_t name_lookup(const NameTab &t, unsigned long long k0, unsigned long long k1) {
    uint64_t slot = name_hash(k0, k1) & t.mask;
    for (;;) {
        const uint4 a = reinterpret_cast<const uint4 *>(t.e + slot)[0];
        const uint32_t id = reinterpret_cast<const uint32_t *>(t.e + slot)[4];
        if (id == 0u) return 0u;
        if ((((unsigned long long)a.y << 32) | a.x) == k0 && (((unsigned long long)a.w << 32) | a.z) == k1) return id;
        slot = (slot + 1) & t.mask;
    }
}

// the (at most 16) bytes [p, p + len) of the text as a key
__device__ static inline void name_key(const uint8_t *__restrict__ p, uint32_t len, unsigned long long &k0, unsigned long long &k1) {
    k0 = k1 = 0;
    for (uint32_t i = 0; i < len && i < 8u; ++i) k0 |= (unsigned long long)p[i] << (8 * i);
    for (uint32_t i = 8; i < len && i < 16u; ++i) k1 |= (unsigned long long)p[i] << (8 * (i - 8));
}

}  // namespace pnx
