// Microbenchmark: HBM read rate of the weight-streaming access patterns (no compute).
//   mode 0: linear -- every wave reads 1 KiB contiguous per instruction, workgroups walk the buffer.
//   mode 1: tile   -- a workgroup owns ROWS weight rows of K2 bytes and reads 128 B of every row per
//                     "stage" (8 rows x 128 B per wave instruction), U stages issued back to back.
//   mode 2: tile, 512 B of every row per stage (2 rows x 512 B per wave instruction).
// hipcc --offload-arch=gfx950 -O3 -o stream_pattern stream_pattern.hip && ./stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(512) void linear_k(const char *buf, size_t bytes, int *sink)
{
    const size_t chunk = 512 * 16;   // bytes per workgroup per load wave-front
    v4i acc = {0, 0, 0, 0};
    const size_t nchunks = bytes / chunk;
    for (size_t c = blockIdx.x; c + (size_t)(U - 1) * gridDim.x < nchunks; c += (size_t)U * gridDim.x) {
        v4i v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const v4i *>(buf + (c + (size_t)u * gridDim.x) * chunk + threadIdx.x * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

// tile pattern: ROWS rows per workgroup tile, SEG bytes of each row per stage
template <int ROWS, int SEG, int U, bool ROT = false>
__global__ __launch_bounds__(512) void tile_k(const char *buf, int n_rows, int K2, int *sink)
{
    constexpr int LPR = SEG / 16;                 // lanes per row
    constexpr int RPI = 512 / LPR;                // rows per workgroup-wide instruction
    constexpr int PIECES = ROWS / RPI;            // instructions per stage per thread
    v4i acc = {0, 0, 0, 0};
    const int n_tiles = n_rows / ROWS;
    const int stages = K2 / SEG;
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const char *base = buf + (size_t)t * ROWS * K2;
        const int rot = ROT ? (t * 5) % stages : 0;         // per-tile starting stage (summation order is free)
        for (int s0 = 0; s0 + U <= stages; s0 += U) {
            const int s = (s0 + rot) % stages;
            v4i v[U][PIECES];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < PIECES; ++p) {
                    const int row = p * RPI + threadIdx.x / LPR;
                    v[u][p] = *reinterpret_cast<const v4i *>(base + (size_t)row * K2 + (size_t)((s + u) % stages) * SEG + (threadIdx.x % LPR) * 16);
                }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < PIECES; ++p) acc ^= v[u][p];
        }
    }
    sink[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

// one decode-size launch: 8 x 11008 rows of 2048 B = 180 MB, a different buffer every launch (4 x 180 MB > the 256 MB
// Infinity Cache), events around 8 launches: what a pure weight stream of that size costs per launch, launch included
static void decode_size()
{
    const int K2 = 2048, n_rows = 8 * 11008;
    const size_t bytes = (size_t)n_rows * K2;
    char *buf[4]; int *sink;
    for (auto &b : buf) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
    hipMalloc(&sink, 2048 * 512 * 4);
    printf("== decode size: %d rows x %d B = %.1f MB per launch, 4 buffers rotated\n", n_rows, K2, bytes / 1e6);
    auto run = [&](const char *name, auto launch) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 8; ++i) launch(buf[i & 3]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 8; ++i) launch(buf[i & 3]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us per launch  %6.2f TB/s\n", name, ms / 8 * 1e3, bytes / (ms / 8) / 1e9);
    };
    run("linear U=2, 256 workgroups", [&](char *b) { hipLaunchKernelGGL((linear_k<2>), dim3(256), dim3(512), 0, 0, b, bytes, sink); });
    run("tile 128 rows x 128 B, U=1, 256 workgroups", [&](char *b) { hipLaunchKernelGGL((tile_k<128, 128, 1>), dim3(256), dim3(512), 0, 0, b, n_rows, K2, sink); });
    run("tile 128 rows x 128 B, U=2, 256 workgroups", [&](char *b) { hipLaunchKernelGGL((tile_k<128, 128, 2>), dim3(256), dim3(512), 0, 0, b, n_rows, K2, sink); });
    run("tile 128 rows x 128 B, U=4, 256 workgroups", [&](char *b) { hipLaunchKernelGGL((tile_k<128, 128, 4>), dim3(256), dim3(512), 0, 0, b, n_rows, K2, sink); });
    run("tile 128 rows x 128 B, U=2, 512 workgroups", [&](char *b) { hipLaunchKernelGGL((tile_k<128, 128, 2>), dim3(512), dim3(512), 0, 0, b, n_rows, K2, sink); });
    run("tile 128 rows x 128 B, U=2, 1024 workgroups", [&](char *b) { hipLaunchKernelGGL((tile_k<128, 128, 2>), dim3(1024), dim3(512), 0, 0, b, n_rows, K2, sink); });
    for (auto &b : buf) hipFree(b);
    hipFree(sink);
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'd') { decode_size(); return 0; }
    for (int K2 : {2048, 3584}) {
    const int n_rows = (K2 == 2048 ? 32 : 64) * 18432;
    const size_t bytes = (size_t)n_rows * K2;
    char *buf; int *sink;
    hipMalloc(&buf, bytes); hipMalloc(&sink, 2048 * 512 * 4);
    hipMemset(buf, 1, bytes);
    printf("== row stride %d B, %.2f GB\n", K2, bytes / 1e9);
    auto rep = [&](const char *name, double ms) { printf("%-44s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9); };
    const int wg = 256;
    rep("linear U=2", time_ms([&] { hipLaunchKernelGGL((linear_k<2>), dim3(wg), dim3(512), 0, 0, buf, bytes, sink); }));
    rep("tile 384 rows x 128 B, U=1", time_ms([&] { hipLaunchKernelGGL((tile_k<384, 128, 1>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 384 rows x 128 B, U=2", time_ms([&] { hipLaunchKernelGGL((tile_k<384, 128, 2>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 192 rows x 128 B, U=1", time_ms([&] { hipLaunchKernelGGL((tile_k<192, 128, 1>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 192 rows x 128 B, U=2", time_ms([&] { hipLaunchKernelGGL((tile_k<192, 128, 2>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 384 rows x 128 B, U=1, rotated", time_ms([&] { hipLaunchKernelGGL((tile_k<384, 128, 1, true>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 192 rows x 128 B, U=1, rotated", time_ms([&] { hipLaunchKernelGGL((tile_k<192, 128, 1, true>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 192 rows x 128 B, U=2, rotated", time_ms([&] { hipLaunchKernelGGL((tile_k<192, 128, 2, true>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    rep("tile 64 rows x 128 B, U=2, rotated", time_ms([&] { hipLaunchKernelGGL((tile_k<64, 128, 2, true>), dim3(wg), dim3(512), 0, 0, buf, n_rows, K2, sink); }));
    hipFree(buf); hipFree(sink);
    }
    return 0;
}
