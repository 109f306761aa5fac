// step_chunks.hpp -- the steps of a graph cut into chunks of RUN_CHUNK consecutive steps of one path.
// chunk_off[p] = number of chunks of the paths before p (a prefix sum over ceil(len / RUN_CHUNK), fixed
// per graph, built by ensure_chunk_off).  Kernels that stream ALL steps once -- the step preparation at
// upload (kernels_cover.hip) and the run index (kernels_runs.hip) -- take one wave per chunk and find the
// chunk's path by a search in chunk_off, so the host never builds a work list.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pnx {

constexpr uint32_t RUN_CHUNK = 4096;  // steps per chunk (one wave walks one chunk)
// a path whose runs (stretches of consecutive steps inside one tile) average fewer steps than this is not worth a run
// index: it is sorted by id at preparation (kernels_cover.hip), or takes the atomic scatter route
constexpr uint32_t RUN_MIN_AVG_LEN = 16;

struct RunChunk {
    uint64_t start;   // first step (absolute index into items)
    uint64_t pstart;  // first step of the path
    uint32_t len;     // steps in this chunk
    uint32_t path;
};

__device__ static inline RunChunk chunk_of(uint64_t c, const uint64_t *__restrict__ chunk_off,
                                           const uint64_t *__restrict__ path_off, uint32_t n_paths) {
    uint32_t lo = 0, hi = n_paths;  // last p with chunk_off[p] <= c (empty paths own no chunk)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_off[mid] <= c) lo = mid; else hi = mid;
    }
    RunChunk ch;
    ch.path = lo;
    ch.pstart = path_off[lo];
    ch.start = ch.pstart + (c - chunk_off[lo]) * RUN_CHUNK;
    const uint64_t left = path_off[lo + 1] - ch.start;
    ch.len = (uint32_t)(left < RUN_CHUNK ? left : RUN_CHUNK);
    return ch;
}

}  // namespace pnx
