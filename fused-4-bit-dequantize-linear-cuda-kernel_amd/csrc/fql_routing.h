// Token routing either side of the grouped GEMM, on the device and in single launches (SURVEY section 8f N1 and
// the expert-parallel step of 8e): what the reference does with argsort / bincount / cumsum / index ops / a
// broadcast multiply and a sum (benchmark/moe_grouped_gemm/routing.py:117-149, 172-189).
//
//   route_plan_kernel    stable counting sort of the (token, slot) pairs by expert id: per-expert counts, their
//                        exclusive prefix (= tokens_per_expert / input_offsets of the grouped GEMM), for every
//                        sorted position the token row it reads (the gather index of fql_moe_gather_fwd_f32) and
//                        for every slot its sorted position (the un-sort index of the combine).
//   combine_kernel       out[t][:] = sum_k w[t][k] * y[pos[t][k]][:]  (k ascending), 16-byte loads and stores.
//   regroup_index_kernel expert-parallel receive side: rows arrive (source rank, local expert)-major, the GEMM
//                        wants (local expert, source rank)-major: gather index + its inverse + the expert table.
#pragma once
#include "fql_common.h"

#define ROUTE_THREADS 256
#define ROUTE_MAX_EXPERTS 128          // LDS: ROUTE_THREADS x experts counters

// One workgroup.  Thread i owns the contiguous slots [i*chunk, (i+1)*chunk): sequential inside a thread and
// threads in slot order, so equal keys keep their order (stable, like argsort(stable=True)).
__global__ __launch_bounds__(ROUTE_THREADS) void route_plan_kernel(
    const int32_t *__restrict__ expert_of_slot, int n_slots, int top_k, int E,
    int32_t *__restrict__ counts, int32_t *__restrict__ offsets, int32_t *__restrict__ token_of_sorted,
    int32_t *__restrict__ pos_of_slot)
{
    extern __shared__ int s_cnt[];                 // [ROUTE_THREADS][E], then per-expert bases at the end
    int *s_base = s_cnt + ROUTE_THREADS * E;       // [E]
    const int tid = threadIdx.x;
    const int chunk = (n_slots + ROUTE_THREADS - 1) / ROUTE_THREADS;
    const int lo = tid * chunk, hi = (lo + chunk < n_slots) ? lo + chunk : n_slots;
    for (int e = 0; e < E; ++e) s_cnt[tid * E + e] = 0;
    for (int i = lo; i < hi; ++i) {
        int e = expert_of_slot[i];
        e = e < 0 ? 0 : (e >= E ? E - 1 : e);      // ids outside [0, E) are clamped
        s_cnt[tid * E + e] += 1;
    }
    __syncthreads();
    // per expert: exclusive scan over the threads (column e), total -> counts
    for (int e = tid; e < E; e += ROUTE_THREADS) {
        int run = 0;
        for (int t = 0; t < ROUTE_THREADS; ++t) {
            const int c = s_cnt[t * E + e];
            s_cnt[t * E + e] = run;
            run += c;
        }
        counts[e] = run;
        s_base[e] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int e = 0; e < E; ++e) {
            const int c = s_base[e];
            s_base[e] = run;
            offsets[e] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int i = lo; i < hi; ++i) {
        int e = expert_of_slot[i];
        e = e < 0 ? 0 : (e >= E ? E - 1 : e);
        const int pos = s_base[e] + s_cnt[tid * E + e]++;
        token_of_sorted[pos] = i / top_k;
        pos_of_slot[i] = pos;
    }
}

// grid: (ceil(N / (4 * 256)), T); one thread = 4 consecutive columns of one token
__global__ __launch_bounds__(256) void combine_kernel(
    const float *__restrict__ y, const int32_t *__restrict__ pos_of_slot, const float *__restrict__ w,
    float *__restrict__ out, int T, int top_k, int N, int R)
{
#pragma clang fp contract(off)                     // (y * w), THEN add, as the reference does: never an FMA
    const int t = blockIdx.y;
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    const bool vec = ((N & 3) == 0) && (((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15) == 0);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < top_k; ++k) {
        int p = pos_of_slot[t * top_k + k];
        p = p < 0 ? 0 : (p >= R ? R - 1 : p);
        const float wk = w != nullptr ? w[t * top_k + k] : 1.0f;     // (x * 1 == x: the gather-add form keeps the bits of pre-weighted rows)
        const float *row = y + (size_t)p * N + n;
        if (vec) {
            const v4f v = *reinterpret_cast<const v4f *>(row);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = acc[c] + v[c] * wk;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n + c < N) acc[c] = acc[c] + row[c] * wk;
        }
    }
    float *orow = out + (size_t)t * N + n;
    if (vec) *reinterpret_cast<v4f *>(orow) = v4f{acc[0], acc[1], acc[2], acc[3]};
    else {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (n + c < N) orow[c] = acc[c];
    }
}

// One workgroup.  cnt[s][e] = rows source rank s sent for local expert e (received order: s major, e minor).
// gather[d] = received row feeding expert-major position d; scatter[r] = expert-major position of received row r.
__global__ __launch_bounds__(256) void regroup_index_kernel(
    const int32_t *__restrict__ cnt, int G, int EL, int32_t *__restrict__ tpe, int32_t *__restrict__ offs,
    int32_t *__restrict__ gather, int32_t *__restrict__ scatter)
{
    extern __shared__ int s_tab[];                 // src_off[G*EL], exp_off[EL*G]
    int *src_off = s_tab, *exp_off = s_tab + G * EL;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < G * EL; ++i) { src_off[i] = run; run += cnt[i]; }
        run = 0;
        for (int e = 0; e < EL; ++e) {
            offs[e] = run;
            int tot = 0;
            for (int s = 0; s < G; ++s) { exp_off[e * G + s] = run; run += cnt[s * EL + e]; tot += cnt[s * EL + e]; }
            tpe[e] = tot;
        }
    }
    __syncthreads();
    for (int seg = 0; seg < G * EL; ++seg) {
        const int s = seg / EL, e = seg - s * EL;
        const int c = cnt[seg], so = src_off[seg], eo = exp_off[e * G + s];
        for (int i = tid; i < c; i += 256) {
            gather[eo + i] = so + i;
            scatter[so + i] = eo + i;
        }
    }
}
