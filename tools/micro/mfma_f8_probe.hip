// Probe of the gfx950 block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4 / 16x16x128) for the fp8-activation path:
//   1. operand slot pairing: lane l of the A operand, byte j  pairs with  lane l' = same half, byte j of the B operand
//      (the true k of a slot never matters as long as both operands use the same slot);
//   2. e4m3 subnormals: the raw nibble byte 0000qqqq read as e4m3 is q * 2^-9 (exponent field 0 or 1) -- does the
//      matrix core honour it, and does an E8M0 scale of 2^9 give back the integer?
//   3. issue rate and sustained throughput of the scaled forms against the INT8 form, on random data.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f8_probe mfma_f8_probe.hip && ./mfma_f8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

static float e4m3_to_float(uint8_t b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 0) v = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) v = NAN;
    else v = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

// ---- 1/2: one MFMA, operands given per (lane, byte), D returned per (lane, reg)
__global__ void one32(const uint8_t *a, const uint8_t *b, float *d, int sa, int sb)
{
    const int l = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = ((const int *)a)[l * 8 + i]; bv[i] = ((const int *)b)[l * 8 + i]; }
    v16f acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}
__global__ void one16(const uint8_t *a, const uint8_t *b, float *d, int sa, int sb)
{
    const int l = threadIdx.x;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = ((const int *)a)[l * 8 + i]; bv[i] = ((const int *)b)[l * 8 + i]; }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

static int check32(const char *name, const std::vector<uint8_t> &A, const std::vector<uint8_t> &B, int sa, int sb, double scale)
{
    uint8_t *da, *db; float *dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 64 * 16 * 4);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one32, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
    std::vector<float> D(64 * 16);
    hipMemcpy(D.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost);
    // slot-identity hypothesis: A operand rows = output index "lane & 31 of A" ... the kernel uses the weights as A:
    // D[i][j] with i from operand A's lane & 31, j from operand B's lane & 31, C layout col = lane & 31 (B side),
    // row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A side)
    int bad = 0; double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double ref = 0;
        for (int h = 0; h < 2; ++h) for (int j = 0; j < 32; ++j)
            ref += (double)e4m3_to_float(A[(h * 32 + row) * 32 + j]) * (double)e4m3_to_float(B[(h * 32 + col) * 32 + j]);
        ref *= scale;
        const double err = fabs(ref - D[l * 16 + r]);
        if (err > maxerr) maxerr = err;
        if (err > 1e-6 * (fabs(ref) + 1e-9)) ++bad;
    }
    printf("%-64s mismatches %4d / 1024   max|err| %.3e   D[0]=%g\n", name, bad, maxerr, D[0]);
    hipFree(da); hipFree(db); hipFree(dd);
    return bad;
}

static int check16(const char *name, const std::vector<uint8_t> &A, const std::vector<uint8_t> &B, int sa, int sb, double scale)
{
    uint8_t *da, *db; float *dd;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dd, 64 * 4 * 4);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one16, dim3(1), dim3(64), 0, 0, da, db, dd, sa, sb);
    std::vector<float> D(64 * 4);
    hipMemcpy(D.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost);
    int bad = 0; double maxerr = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) * 4 + r;
        double ref = 0;
        for (int h = 0; h < 4; ++h) for (int j = 0; j < 32; ++j)
            ref += (double)e4m3_to_float(A[(h * 16 + row) * 32 + j]) * (double)e4m3_to_float(B[(h * 16 + col) * 32 + j]);
        ref *= scale;
        const double err = fabs(ref - D[l * 4 + r]);
        if (err > maxerr) maxerr = err;
        if (err > 1e-6 * (fabs(ref) + 1e-9)) ++bad;
    }
    printf("%-64s mismatches %4d / 256    max|err| %.3e   D[0]=%g\n", name, bad, maxerr, D[0]);
    hipFree(da); hipFree(db); hipFree(dd);
    return bad;
}

// ---- 3: rates
template <int CBSZ, int BLGP, int NACC>
__global__ __launch_bounds__(512) void rate_f8_32(unsigned long long *cyc, float *sink, int iters, int seed)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (int)((threadIdx.x * 97u + i * 7u + seed * 13u) * 2654435761u) & 0x0F0F0F0F;            // nibble bytes
        b[i] = (int)((threadIdx.x * 2654435761u + i * 40503u + seed) * 2246822519u) & 0x7E7E7E7E;     // finite e4m3
    }
    v16f acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
            acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[n], CBSZ, BLGP, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512) void rate_f8_16(unsigned long long *cyc, float *sink, int iters, int seed)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (int)((threadIdx.x * 97u + i * 7u + seed * 13u) * 2654435761u) & 0x0F0F0F0F;
        b[i] = (int)((threadIdx.x * 2654435761u + i * 40503u + seed) * 2246822519u) & 0x7E7E7E7E;
    }
    v4f acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
            acc[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[n], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512) void rate_i8_32(unsigned long long *cyc, float *sink, int iters, int seed)
{
    v4i a, b;
    for (int i = 0; i < 4; ++i) {
        a[i] = (int)((threadIdx.x * 97u + i * 7u + seed * 13u) * 2654435761u) & 0x0F0F0F0F;
        b[i] = (int)((threadIdx.x * 2654435761u + i * 40503u + seed) * 2246822519u);
    }
    v16i acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename F>
static void run(const char *name, F launch, int blocks, int threads, int iters, int nacc, double ops_per_mfma)
{
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8);
    hipMalloc(&sink, sizeof(float) * blocks * threads);
    launch(cyc, sink, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(cyc, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = (double)iters * nacc;
    const int waves = threads / 64;
    const double total_ops = nm * ops_per_mfma * waves * blocks;
    printf("%-52s %.1f cyc/MFMA/wave  wall %.3f ms  %.0f TOPS  clk %.2f GHz\n", name, (double)h[0] / nm, ms,
           total_ops / (ms * 1e-3) / 1e12, (double)h[0] / (ms * 1e-3) / 1e9);
    hipFree(cyc); hipFree(sink);
}

int main()
{
    srand(7);
    // 1. slot pairing, small exact integers (e4m3 codes of 0,1,2,3: 0x00 0x38 0x40 0x44), signs on B
    const uint8_t code[4] = {0x00, 0x38, 0x40, 0x44};
    std::vector<uint8_t> A(2048), B(2048);
    for (int i = 0; i < 2048; ++i) { A[i] = code[rand() & 3]; B[i] = code[rand() & 3] | ((rand() & 1) << 7); }
    check32("32x32x64 fp8: slot (half,byte) pairs with same slot", A, B, 0x7F7F7F7F, 0x7F7F7F7F, 1.0);
    check16("16x16x128 fp8: slot (quarter,byte) pairs with same slot", A, B, 0x7F7F7F7F, 0x7F7F7F7F, 1.0);
    // 2. raw nibble bytes as e4m3 (subnormal / first binade), B = small integers
    for (int i = 0; i < 2048; ++i) A[i] = rand() & 15;
    check32("32x32x64: A = nibble bytes (q * 2^-9), unit scales", A, B, 0x7F7F7F7F, 0x7F7F7F7F, 1.0);
    check32("32x32x64: A = nibble bytes, scale_a = 2^9 (E8M0 136)", A, B, 0x88888888, 0x7F7F7F7F, 512.0);
    check16("16x16x128: A = nibble bytes, scale_a = 2^9", A, B, 0x88888888, 0x7F7F7F7F, 512.0);
    // full-range finite e4m3 on B with nibble A (fp32 accumulate: compare loosely)
    for (int i = 0; i < 2048; ++i) { B[i] = rand() & 0xFF; if ((B[i] & 0x7F) == 0x7F) B[i] ^= 1; }
    check32("32x32x64: nibble A x random finite e4m3 B, scale 2^9", A, B, 0x88888888, 0x7F7F7F7F, 512.0);

    // 3. rates
    const int it = 20000;
    run("i8 32x32x32, nibble A x random B, 12 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_i8_32<12>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 65536.0);
    run("i8 32x32x32, 9 acc, 2 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_i8_32<9>), dim3(256), dim3(512), 0, 0, c, s, n, 1); }, 256, 512, it, 9, 65536.0);
    run("f8f6f4 32x32x64 fp8 x fp8, 12 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<0, 0, 12>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 131072.0);
    run("f8f6f4 32x32x64 fp8 x fp8, 6 acc, 2 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<0, 0, 6>), dim3(256), dim3(512), 0, 0, c, s, n, 1); }, 256, 512, it, 6, 131072.0);
    run("f8f6f4 32x32x64 fp4(A) x fp8(B), 12 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<4, 0, 12>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 131072.0);
    run("f8f6f4 32x32x64 fp4 x fp4, 12 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<4, 4, 12>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 131072.0);
    run("f8f6f4 32x32x64 fp6 x fp6, 12 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<2, 2, 12>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 131072.0);
    run("f8f6f4 16x16x128 fp8 x fp8, 16 acc, 1 w/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_16<16>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 16, 65536.0);
    run("f8f6f4 32x32x64 fp8 x fp8, 12 acc, 1 CU", [&](auto c, auto s, int n) { hipLaunchKernelGGL((rate_f8_32<0, 0, 12>), dim3(1), dim3(256), 0, 0, c, s, n, 1); }, 1, 256, it, 12, 131072.0);
    return 0;
}
