// pansyn.hip -- pansyn-v1 synthetic pangenome generator, device side.
//
// Produces, directly in HBM, the same CSR (u32 step ids + u64 path offsets + node lengths)
// that the CPU generator of the test infrastructure produces for (seed, n_nodes, n_paths);
// the definition is integer-only so host and device agree bit for bit (DESIGN.md,
// "pansyn-v1").  This is the input of BASELINE.json configs 2-4 (10M nodes x 256..512
// paths = 1..2 G steps), which is too large to ship through PCIe for every benchmark run.
#include <vector>

#include "pnx_context.hpp"

namespace pnx {

__host__ __device__ static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ static inline uint64_t ps_key(uint64_t seed, uint64_t stream) {
    return splitmix64(seed ^ (0xA0761D6478BD642Full * (stream + 1)));
}
constexpr uint64_t ONE53 = 1ull << 53;
constexpr uint32_t GEN_PER_THREAD = 8;
constexpr uint32_t GEN_CHUNK = 256 * GEN_PER_THREAD;

// per-node containment threshold (on the 53-bit uniform) and length
// (node_lo: the shard holds the nodes node_lo + 1 .. node_lo + n_nodes of the graph as its items 1 .. n_nodes; a node of pansyn
// does not depend on how many nodes the graph has, so the shards of a graph are generated independently)
__global__ void k_pansyn_nodes(uint64_t seed, uint32_t n_nodes, uint32_t n_paths, uint64_t node_lo,
                               uint64_t *__restrict__ thr, uint32_t *__restrict__ lens) {
    const uint64_t il = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (il > n_nodes) return;
    if (il == 0) {
        thr[0] = 0;
        if (lens) lens[0] = 0;
        return;
    }
    const uint64_t i = node_lo + il;
    uint64_t t = splitmix64(ps_key(seed, 3) + i) >> 11;
    uint64_t th;
    if (t < ONE53 / 100 * 20) th = ONE53;                 // core: every path
    else if (t < ONE53 / 100 * 65) th = ONE53 / n_paths;  // near-singleton: ~1 path
    else th = splitmix64(ps_key(seed, 4) + i) >> 11;      // shell: uniform frequency
    thr[il] = th;
    if (lens) {
        uint64_t t1 = splitmix64(ps_key(seed, 1) + i) >> 11;
        uint32_t len = 1;
        if (t1 >= ONE53 / 100 * 55) {
            uint64_t e = splitmix64(ps_key(seed, 2) + i);
            uint64_t k = e ? (uint64_t)__builtin_clzll(e) : 63;
            if (k > 63) k = 63;
            uint64_t l = 1 + ((180 * ((k << 16) + (e & 0xFFFF))) >> 16);
            len = (uint32_t)(l > 50000 ? 50000 : l);
        }
        lens[il] = len;
    }
}

__device__ static inline uint32_t block_sum_256(uint32_t v, uint32_t *sh) {
    // wave reduce then 4 partials
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t s = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return s;
}

__device__ static inline void node_entries(uint64_t kp, const uint64_t *__restrict__ thr,
                                           uint32_t n_nodes, uint64_t first, uint32_t cnt[GEN_PER_THREAD]) {
    // kp: the path's key PLUS the shard's first node (the hash of (path, node) takes the node's number in the whole graph)
#pragma unroll
    for (uint32_t e = 0; e < GEN_PER_THREAD; ++e) {
        uint64_t i = first + e;
        uint32_t c = 0;
        if (i <= n_nodes) {
            uint64_t h = splitmix64(kp + i);
            if ((h >> 11) < thr[i]) c = 1 + ((h & 63) == 0);
        }
        cnt[e] = c;
    }
}

__global__ __launch_bounds__(256) void k_pansyn_count(uint64_t k5, const uint64_t *__restrict__ thr,
                                                      uint32_t n_nodes, uint32_t n_chunks, uint64_t node_lo,
                                                      uint32_t *__restrict__ counts) {
    __shared__ uint32_t sh[4];
    const uint32_t p = blockIdx.x / n_chunks, c = blockIdx.x % n_chunks;
    const uint64_t kp = splitmix64(k5 + p) + node_lo;
    uint32_t cnt[GEN_PER_THREAD];
    node_entries(kp, thr, n_nodes, (uint64_t)c * GEN_CHUNK + threadIdx.x * GEN_PER_THREAD + 1, cnt);
    uint32_t s = 0;
#pragma unroll
    for (uint32_t e = 0; e < GEN_PER_THREAD; ++e) s += cnt[e];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) counts[blockIdx.x] = s;
}

// one workgroup per path: exclusive scan of its chunk counts
__global__ __launch_bounds__(256) void k_pansyn_scan(const uint32_t *__restrict__ counts, uint32_t n_chunks,
                                                     uint64_t *__restrict__ chunk_base,
                                                     uint64_t *__restrict__ path_len) {
    __shared__ uint64_t sh[256];
    const uint32_t p = blockIdx.x;
    uint64_t run = 0;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += 256) {
        uint32_t c = c0 + threadIdx.x;
        uint64_t v = c < n_chunks ? counts[(uint64_t)p * n_chunks + c] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (c < n_chunks) chunk_base[(uint64_t)p * n_chunks + c] = run + sh[threadIdx.x] - v;
        run += sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) path_len[p] = run;
}

__global__ __launch_bounds__(256) void k_pansyn_fill(uint64_t k5, const uint64_t *__restrict__ thr,
                                                     uint32_t n_nodes, uint32_t n_chunks, uint64_t node_lo,
                                                     const uint64_t *__restrict__ chunk_base,
                                                     const uint64_t *__restrict__ path_off,
                                                     uint32_t *__restrict__ items) {
    __shared__ uint32_t sh[256];
    const uint32_t p = blockIdx.x / n_chunks, c = blockIdx.x % n_chunks;
    const uint64_t kp = splitmix64(k5 + p) + node_lo;
    const uint64_t first = (uint64_t)c * GEN_CHUNK + threadIdx.x * GEN_PER_THREAD + 1;
    uint32_t cnt[GEN_PER_THREAD];
    node_entries(kp, thr, n_nodes, first, cnt);
    uint32_t s = 0;
#pragma unroll
    for (uint32_t e = 0; e < GEN_PER_THREAD; ++e) s += cnt[e];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t pos = chunk_base[blockIdx.x] + (sh[threadIdx.x] - s);  // position within the path
    const uint64_t s0 = path_off[p], s1 = path_off[p + 1];
    const bool desc = (p % 16) == 15;  // every 16th path runs through the ids downwards
#pragma unroll
    for (uint32_t e = 0; e < GEN_PER_THREAD; ++e) {
        for (uint32_t r = 0; r < cnt[e]; ++r) {
            items[desc ? (s1 - 1 - pos) : (s0 + pos)] = (uint32_t)(first + e);
            ++pos;
        }
    }
}

// pansyn-v1r (the definition: DESIGN.md, "pansyn-v1r"): the paths of pansyn-v1 rearranged in blocks of 64
// steps -- reversed in place (1 %), replaced by a copy of an earlier block of the same path (0.1 %), moved elsewhere in the id
// space (0.05 %); the first and the last block of a path stay.  One wave per block, from the generated steps `src` into
// `dst`.  What paths that are not sorted by id look like to the coverage pass: steps outside the band their position says.
__global__ __launch_bounds__(256) void k_pansyn_rearrange(uint64_t k8, uint32_t n_nodes, uint32_t n_paths, const uint64_t *__restrict__ path_off,
                                                          const uint64_t *__restrict__ blk_off, uint64_t n_blk_total,
                                                          const uint32_t *__restrict__ src, uint32_t *__restrict__ dst) {
    const uint64_t w = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // global block number
    const uint32_t lane = threadIdx.x & 63u;
    if (w >= n_blk_total) return;
    // which path: the last p with blk_off[p] <= w
    uint32_t lo = 0, hi = n_paths;
    while (hi - lo > 1) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (blk_off[mid] <= w) lo = mid;
        else hi = mid;
    }
    const uint32_t p = lo;
    const uint64_t B = w - blk_off[p];
    const uint64_t s0 = path_off[p], len = path_off[p + 1] - s0, n_blocks = (len + 63) / 64;
    const uint64_t b0 = B * 64, bl = len - b0 < 64 ? len - b0 : 64;
    if (lane >= bl) return;
    uint64_t from = b0 + lane;
    uint64_t off = 0;
    if (B >= 1 && B + 1 < n_blocks) {
        const uint64_t h = splitmix64(splitmix64(k8 + p) + B), r = h % 10000;
        if (r < 100) from = b0 + bl - 1 - lane;
        else if (r < 110) from = ((h >> 20) % B) * 64 + lane;
        else if (r < 115 && n_nodes > 1) off = 1 + (h >> 24) % ((uint64_t)n_nodes - 1);
    }
    uint32_t id = src[s0 + from];
    if (off) id = (uint32_t)(((uint64_t)id - 1 + off) % n_nodes) + 1u;
    dst[s0 + b0 + lane] = id;
}

int pansyn_rearrange_device(pnx_ctx *ctx, uint64_t seed) {
    const uint32_t P = ctx->n_paths;
    const uint64_t S = ctx->n_steps;
    if (!P || !S) return PNX_OK;
    std::vector<uint64_t> blk((size_t)P + 1, 0);
    for (uint32_t p = 0; p < P; ++p) blk[p + 1] = blk[p] + (ctx->h_path_off[p + 1] - ctx->h_path_off[p] + 63) / 64;
    const uint64_t n_blk = blk[P];
    DevBuf d_blk, d_src;
    int rc;
    if ((rc = ensure(ctx, d_blk, ((size_t)P + 1) * 8)) || (rc = ensure(ctx, d_src, S * 4 + 64))) {
        release(d_blk);
        release(d_src);
        return rc;
    }
    hipError_t e = hipMemcpyAsync(d_blk.p, blk.data(), ((size_t)P + 1) * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_src.p, ctx->d_items.p, S * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess && n_blk) {
        const uint64_t grid = (n_blk * 64 + 255) / 256;
        hipLaunchKernelGGL(k_pansyn_rearrange, dim3((unsigned)grid), dim3(256), 0, ctx->stream, ps_key(seed, 8), ctx->n_items, P,
                           (const uint64_t *)ctx->d_path_off.p, (const uint64_t *)d_blk.p, n_blk, (const uint32_t *)d_src.p, (uint32_t *)ctx->d_items.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    release(d_blk);
    release(d_src);
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pansyn (rearranged): %s", hipGetErrorString(e));
    return PNX_OK;
}

int pansyn_generate_device(pnx_ctx *ctx, uint64_t seed, uint32_t n_nodes, uint32_t n_paths, int with_weights, uint64_t node_lo) {
    int rc;
    const uint32_t n_chunks = (n_nodes + GEN_CHUNK - 1) / GEN_CHUNK;
    const uint64_t nb = (uint64_t)n_paths * n_chunks;
    if (nb > 0x7FFFFFFFull) return ctx->fail(PNX_ELIMIT, "pansyn: n_paths * ceil(n_nodes/2048) must be < 2^31");
    DevBuf d_thr, d_counts, d_base, d_len;
    auto cleanup = [&]() {
        release(d_thr);
        release(d_counts);
        release(d_base);
        release(d_len);
    };
    if ((rc = ensure(ctx, d_thr, ((size_t)n_nodes + 1) * 8)) || (rc = ensure(ctx, d_counts, nb * 4)) ||
        (rc = ensure(ctx, d_base, nb * 8)) || (rc = ensure(ctx, d_len, (size_t)n_paths * 8))) {
        cleanup();
        return rc;
    }
    uint32_t *d_lens = nullptr;
    if (with_weights) {
        if ((rc = ensure(ctx, ctx->d_weights, ((size_t)n_nodes + 1) * 4))) {
            cleanup();
            return rc;
        }
        d_lens = (uint32_t *)ctx->d_weights.p;
    }
    const uint64_t k5 = ps_key(seed, 5);
    hipLaunchKernelGGL(k_pansyn_nodes, dim3((n_nodes + 1 + 255) / 256), dim3(256), 0, ctx->stream, seed, n_nodes,
                       n_paths, node_lo, (uint64_t *)d_thr.p, d_lens);
    hipLaunchKernelGGL(k_pansyn_count, dim3((unsigned)nb), dim3(256), 0, ctx->stream, k5, (const uint64_t *)d_thr.p,
                       n_nodes, n_chunks, node_lo, (uint32_t *)d_counts.p);
    hipLaunchKernelGGL(k_pansyn_scan, dim3(n_paths), dim3(256), 0, ctx->stream, (const uint32_t *)d_counts.p,
                       n_chunks, (uint64_t *)d_base.p, (uint64_t *)d_len.p);
    std::vector<uint64_t> len(n_paths);
    hipError_t e = hipMemcpyAsync(len.data(), d_len.p, (size_t)n_paths * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return ctx->fail(PNX_EHIP, "pansyn: %s", hipGetErrorString(e));
    }
    ctx->h_path_off.assign((size_t)n_paths + 1, 0);
    for (uint32_t p = 0; p < n_paths; ++p) ctx->h_path_off[p + 1] = ctx->h_path_off[p] + len[p];
    const uint64_t S = ctx->h_path_off[n_paths];
    if ((rc = ensure(ctx, ctx->d_items, S * 4 + 64)) || (rc = ensure(ctx, ctx->d_path_off, ((size_t)n_paths + 1) * 8))) {
        cleanup();
        return rc;
    }
    e = hipMemcpyAsync(ctx->d_path_off.p, ctx->h_path_off.data(), ((size_t)n_paths + 1) * 8, hipMemcpyHostToDevice,
                       ctx->stream);
    if (e != hipSuccess) {
        cleanup();
        return ctx->fail(PNX_EHIP, "pansyn: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k_pansyn_fill, dim3((unsigned)nb), dim3(256), 0, ctx->stream, k5, (const uint64_t *)d_thr.p,
                       n_nodes, n_chunks, node_lo, (const uint64_t *)d_base.p, (const uint64_t *)ctx->d_path_off.p,
                       (uint32_t *)ctx->d_items.p);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(PNX_EHIP, "pansyn: %s", hipGetErrorString(e));
    ctx->n_items = n_nodes;
    ctx->n_paths = n_paths;
    ctx->n_steps = S;
    ctx->weighted = ctx->have_weights = with_weights != 0;
    return PNX_OK;
}

}  // namespace pnx
