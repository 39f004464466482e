// Pairs in KEY order for the wave-per-parent kernel (mvin_gather_attn_l2_prj_ordered_fwd): `order` = a permutation of 0 .. B-1 in which
// equal keys (the pairs' item ids) are neighbours.  Pairs that share an item gather the same ~100 rows; processed back to back by one
// wave their loads are L1 / L2 hits, processed wherever the batch happens to hold them every one of them goes past the L2 (each XCD's
// L2 is 4 MB against 82 MB of projected tables): at BASELINE C3 the launch's fabric requests fall from 58.4 M to 27.8 M (7.5 -> 3.6 GB)
// while its own time stays -- and the scoring step as a whole is bound by exactly those bytes.
//
// Not a sort: a partition into kOrdBuckets buckets by the key's low bits (two or three items per bucket at last-fm's 48 091 items;
// inside a bucket any order).  A global counting sort by item would put 46 000 atomics on the hottest item's counter (Zipf); here every
// workgroup counts its chunk in LDS (64 KB of counters), reserves its share of every bucket it touched with ONE global atomic per
// (workgroup, bucket), and scatters by (bucket start + the workgroup's share + the pair's rank inside the workgroup).  The ranks come
// from LDS atomics: the order inside a bucket differs from run to run, the results it feeds do not depend on it.
#include "mvin_kernels.h"

namespace mvin {

constexpr int kOrdBuckets = 16384;
constexpr int kOrdGrid = 64;
constexpr int kOrdThreads = 1024;

__device__ __forceinline__ unsigned ord_key(const int64_t* k64, const int32_t* k32, int64_t i) {
    return k64 ? reinterpret_cast<const unsigned*>(k64)[2 * i] : (unsigned)k32[i];
}

__global__ __launch_bounds__(kOrdThreads) void order_hist_kernel(const int64_t* __restrict__ k64, const int32_t* __restrict__ k32, int64_t B,
                                                                 int32_t* __restrict__ total, int32_t* __restrict__ base, int32_t* __restrict__ rank) {
    extern __shared__ int sCnt[];                         // [kOrdBuckets]
    for (int b = threadIdx.x; b < kOrdBuckets; b += kOrdThreads) sCnt[b] = 0;
    __syncthreads();
    const int64_t per = (B + gridDim.x - 1) / gridDim.x, i0 = (int64_t)blockIdx.x * per, i1 = min(B, i0 + per);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kOrdThreads) rank[i] = atomicAdd(&sCnt[ord_key(k64, k32, i) & (kOrdBuckets - 1)], 1);
    __syncthreads();
    for (int b = threadIdx.x; b < kOrdBuckets; b += kOrdThreads) {
        const int c = sCnt[b];
        if (c) base[(size_t)blockIdx.x * kOrdBuckets + b] = atomicAdd(&total[b], c);
    }
}

// exclusive scan of total[kOrdBuckets] in place (one workgroup: 16 counters per thread)
__global__ __launch_bounds__(kOrdThreads) void order_scan_kernel(int32_t* __restrict__ total) {
    __shared__ int sW[16];
    constexpr int PER = kOrdBuckets / kOrdThreads;
    int v[PER], s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = total[threadIdx.x * PER + j];
        s += v[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) sW[wave] = incl;
    __syncthreads();
    int pre = incl - s;
    for (int w = 0; w < wave; ++w) pre += sW[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        total[threadIdx.x * PER + j] = pre;
        pre += v[j];
    }
}

__global__ __launch_bounds__(kOrdThreads) void order_scatter_kernel(const int64_t* __restrict__ k64, const int32_t* __restrict__ k32, int64_t B,
                                                                    const int32_t* __restrict__ start, const int32_t* __restrict__ base,
                                                                    const int32_t* __restrict__ rank, int32_t* __restrict__ order) {
    const int64_t per = (B + gridDim.x - 1) / gridDim.x, i0 = (int64_t)blockIdx.x * per, i1 = min(B, i0 + per);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kOrdThreads) {
        const int b = (int)(ord_key(k64, k32, i) & (kOrdBuckets - 1));
        order[start[b] + base[(size_t)blockIdx.x * kOrdBuckets + b] + rank[i]] = (int32_t)i;
    }
}

size_t order_ws_elems(int64_t B) { return (size_t)B + (size_t)kOrdBuckets + (size_t)kOrdGrid * kOrdBuckets; }      // rank | total / start | base

hipError_t launch_order_by_key(const int64_t* k64, const int32_t* k32, int64_t B, int32_t* ws, int32_t* order, hipStream_t st) {
    int32_t* rank = ws;
    int32_t* total = ws + B;
    int32_t* base = total + kOrdBuckets;
    hipError_t e = hipMemsetAsync(total, 0, (size_t)kOrdBuckets * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    static thread_local bool attr = false;
    if (!attr) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(order_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kOrdBuckets * 4);
        if (e != hipSuccess) return e;
        attr = true;
    }
    order_hist_kernel<<<kOrdGrid, kOrdThreads, kOrdBuckets * sizeof(int), st>>>(k64, k32, B, total, base, rank);
    order_scan_kernel<<<1, kOrdThreads, 0, st>>>(total);
    order_scatter_kernel<<<kOrdGrid, kOrdThreads, 0, st>>>(k64, k32, B, total, base, rank, order);
    return hipGetLastError();
}

}  // namespace mvin
