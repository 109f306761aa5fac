// random_sector_rate.hip -- how many random 64-byte sector reads per second does the MI355X memory
// system sustain?  (The tile index K0 is made of exactly such reads; this bounds it.)
// Variants: bytes per probe (64 / 128 / 256, aligned), independent probes in flight per thread
// (1 / 2 / 4), footprint (4 / 16 GB), lanes per probe (1 = every lane its own sector with four
// 16-byte loads, 4 = a quad shares one sector, 16 bytes per lane).
//   hipcc --offload-arch=gfx950 -O3 -o random_sector_rate random_sector_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ static inline uint64_t mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// every lane: ILP independent probes of SECT bytes; LOCAL > 0 keeps the lanes of a wave within
// LOCAL bytes of each other (K0's lanes probe neighbouring tiles of one path)
template <int SECT, int ILP>
__global__ void k_lane_probes(const uint4 *__restrict__ buf, uint64_t n_sect, uint64_t local_sect, uint32_t rounds,
                              uint32_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    uint64_t h = mix(tid);
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[ILP][SECT / 16];
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            h = mix(h + acc);  // depends on the previous round's data: a dependent chain like a search
            uint64_t s;
            if (local_sect) {
                const uint64_t base = mix((tid >> 6) * 977 + r * 131 + i) % (n_sect - local_sect);
                s = base + h % local_sect;
            } else {
                s = h % n_sect;
            }
            const uint4 *p = buf + s * (SECT / 16);
#pragma unroll
            for (int q = 0; q < SECT / 16; ++q) v[i][q] = p[q];
        }
#pragma unroll
        for (int i = 0; i < ILP; ++i)
#pragma unroll
            for (int q = 0; q < SECT / 16; ++q) acc += v[i][q].x ^ v[i][q].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// a quad of lanes shares each 64-byte sector (16 bytes per lane); every quad has ILP probes in flight
template <int ILP>
__global__ void k_quad_probes(const uint4 *__restrict__ buf, uint64_t n_sect, uint32_t rounds, uint32_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t quad = tid >> 2;
    const uint32_t ql = threadIdx.x & 3;
    uint32_t acc = 0;
    uint64_t h = mix(quad);
    for (uint32_t r = 0; r < rounds; ++r) {
        uint4 v[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            // all four lanes of a quad must agree on the sector: fold the quad's data
            uint32_t a = acc;
            a += __shfl_xor(a, 1);
            a += __shfl_xor(a, 2);
            h = mix(h + a);
            v[i] = buf[(h % n_sect) * 4 + ql];
        }
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc += v[i].x ^ v[i].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    uint32_t *sink;
    hipMalloc(&sink, 4);
    for (uint64_t gb : {4ull, 16ull}) {
        const uint64_t bytes = gb << 30;
        uint4 *buf;
        if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        hipMemset(buf, 1, bytes);
        const uint32_t rounds = 4;
        const unsigned threads = 5u << 20;  // 5 M threads, like the fine index kernel at 1 k paths
        const dim3 grid(threads / 256), block(256);
        auto report = [&](const char *name, int sect, int ilp, uint64_t local, double ms, double probes) {
            printf("{\"footprint_gb\": %llu, \"variant\": \"%s\", \"bytes_per_probe\": %d, \"ilp\": %d, \"wave_local_bytes\": %llu, "
                   "\"ms\": %.4f, \"G_probes_per_s\": %.2f, \"GB_per_s\": %.1f}\n",
                   (unsigned long long)gb, name, sect, ilp, (unsigned long long)local, ms, probes / ms / 1e6, probes * sect / ms / 1e6);
            fflush(stdout);
        };
#define LANE(S, I, LOC)                                                                                                   \
    {                                                                                                                     \
        const uint64_t ns = bytes / S, ls = (LOC) / S;                                                                    \
        double ms = time_ms([&] { hipLaunchKernelGGL((k_lane_probes<S, I>), grid, block, 0, 0, buf, ns, ls, rounds, sink); }, 5); \
        report("lane", S, I, LOC, ms, (double)threads * rounds * I);                                                      \
    }
        LANE(64, 1, 0) LANE(64, 2, 0) LANE(64, 4, 0) LANE(128, 1, 0) LANE(128, 2, 0) LANE(256, 1, 0)
        LANE(64, 1, 262144) LANE(64, 2, 262144) LANE(128, 1, 262144) LANE(256, 1, 262144)
#define QUAD(I)                                                                                                            \
    {                                                                                                                      \
        const uint64_t ns = bytes / 64;                                                                                    \
        double ms = time_ms([&] { hipLaunchKernelGGL((k_quad_probes<I>), grid, block, 0, 0, buf, ns, rounds, sink); }, 5); \
        report("quad", 64, I, 0, ms, (double)threads / 4 * rounds * I);                                                    \
    }
        QUAD(1) QUAD(4) QUAD(8)
        hipFree(buf);
    }
    return 0;
}
