// h2d_rate.hip -- how fast do the bytes of a mapped file reach HBM?  (GFA text -> device tokeniser)
//   a. hipMemcpy straight from the mapping (pageable: the runtime stages it)
//   b. T host threads copy pieces of the mapping into a ring of pinned buffers, each followed by hipMemcpyAsync
//   c. hipHostRegister of the mapping, then one hipMemcpyAsync
// build: hipcc --offload-arch=gfx950 -O2 -pthread -o h2d_rate h2d_rate.hip ; run: ./h2d_rate FILE [threads]
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const char *file = argv[1];
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    int fd = open(file, O_RDONLY);
    struct stat st;
    fstat(fd, &st);
    const size_t N = st.st_size;
    const char *map = (const char *)mmap(nullptr, N, PROT_READ, MAP_PRIVATE, fd, 0);
    volatile char sink = 0;
    for (size_t i = 0; i < N; i += 4096) sink += map[i];  // page cache + page tables warm
    double t0 = now();
    CK(hipSetDevice(0));
    char *d;
    CK(hipMalloc(&d, N));
    printf("{\"bytes\": %zu, \"init_ms\": %.1f", N, (now() - t0) * 1e3);
    t0 = now();
    CK(hipMemcpy(d, map, N, hipMemcpyHostToDevice));
    printf(", \"pageable_GBps\": %.2f", N / (now() - t0) / 1e9);
    t0 = now();
    CK(hipMemcpy(d, map, N, hipMemcpyHostToDevice));
    printf(", \"pageable_again_GBps\": %.2f", N / (now() - t0) / 1e9);
    for (size_t piece : {(size_t)8 << 20, (size_t)32 << 20}) {
        for (int threads : {T / 2, T}) {
            // ring: per thread 2 pinned buffers of `piece` bytes
            t0 = now();
            std::vector<char *> pin(threads * 2);
            for (auto &p : pin) CK(hipHostMalloc(&p, piece, hipHostMallocDefault));
            const double t_alloc = now() - t0;
            std::atomic<size_t> next{0};
            t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < threads; ++k)
                th.emplace_back([&, k] {
                    CK(hipSetDevice(0));
                    hipStream_t s;
                    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                    hipEvent_t ev[2];
                    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
                    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
                    bool used[2] = {false, false};
                    int b = 0;
                    for (;;) {
                        const size_t off = next.fetch_add(piece);
                        if (off >= N) break;
                        const size_t len = std::min(piece, N - off);
                        if (used[b]) CK(hipEventSynchronize(ev[b]));
                        memcpy(pin[2 * k + b], map + off, len);
                        CK(hipMemcpyAsync(d + off, pin[2 * k + b], len, hipMemcpyHostToDevice, s));
                        CK(hipEventRecord(ev[b], s));
                        used[b] = true;
                        b ^= 1;
                    }
                    CK(hipStreamSynchronize(s));
                });
            for (auto &t : th) t.join();
            printf(", \"pinned_ring_%zuMB_x%d_GBps\": %.2f, \"pinned_alloc_%zuMB_x%d_ms\": %.1f", piece >> 20, threads, N / (now() - t0) / 1e9,
                   piece >> 20, threads, t_alloc * 1e3);
            for (auto &p : pin) CK(hipHostFree(p));
        }
    }
    t0 = now();
    hipError_t e = hipHostRegister((void *)map, N, hipHostRegisterDefault);
    const double t_reg = now() - t0;
    if (e == hipSuccess) {
        t0 = now();
        CK(hipMemcpy(d, map, N, hipMemcpyHostToDevice));
        printf(", \"register_ms\": %.1f, \"registered_GBps\": %.2f", t_reg * 1e3, N / (now() - t0) / 1e9);
        hipHostUnregister((void *)map);
    } else {
        printf(", \"register\": \"%s\"", hipGetErrorString(e));
    }
    printf("}\n");
    return 0;
}
