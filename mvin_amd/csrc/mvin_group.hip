// The batch in user order, for the grouped form of key addressing (mvin_key_addressing_grouped_fwd): a counting
// sort by user id entirely on the device (no host sync, graph-capturable) --
//   count[u]      = pairs of user u in the batch                          (int atomics)
//   seg_user[s]   = the s-th user that occurs, in increasing id order ;  seg_ptr[s] = first position of its pairs
//   pair_index[p] = original index of the pair at position p              (order inside a segment: arbitrary;
//                                                                          per-pair results do not depend on it)
// Replaces torch.sort + seven elementwise / scan / scatter launches (0.27 ms per 524 288 pairs) by three small kernels
// (count, one-pass scan, scatter), or by ONE launch of one workgroup with the counters in LDS when the batch is small
// enough for a single workgroup to walk (group_small_kernel).
#include <cstdlib>

#include "mvin_kernels.h"

namespace mvin {

// Device-resident ids are not validated per launch: an id outside [0, n_user) is clamped into the table (every pair keeps
// a segment, so no output row is left unwritten), exactly as the per-pair kernels clamp it (key_addr_lists).
__device__ __forceinline__ int64_t clamp_user(int64_t u, int n_user) { return u < 0 ? 0 : (u >= n_user ? n_user - 1 : u); }

// rank[i] = how many pairs of the same user were counted before pair i: the scatter pass then needs no second round of atomics
__global__ void group_count_kernel(const int64_t* __restrict__ u64, const int32_t* __restrict__ u32, int64_t B, int n_user,
                                   int32_t* __restrict__ count, int32_t* __restrict__ rank) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        rank[i] = atomicAdd(count + clamp_user(u64 ? u64[i] : (int64_t)u32[i], n_user), 1);
    }
}

// Exclusive scan of a per-thread pair (c, s) over the 1024 threads of a workgroup: wave scan by lane shifts, then the
// 16 wave totals.  Returns the exclusive prefixes; totC / totS = the workgroup totals.
__device__ __forceinline__ void block_scan2(int c, int s, int* sC, int* sS, int& ec, int& es, int& totC, int& totS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ic = c, is = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int tc = __shfl_up(ic, o, kWave), ts = __shfl_up(is, o, kWave);
        if (lane >= o) {
            ic += tc;
            is += ts;
        }
    }
    if (lane == 63) {
        sC[wave] = ic;
        sS[wave] = is;
    }
    __syncthreads();
    int wc = 0, ws = 0;
    totC = 0;
    totS = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int a = sC[w], b = sS[w];
        if (w < wave) {
            wc += a;
            ws += b;
        }
        totC += a;
        totS += b;
    }
    ec = wc + ic - c;
    es = ws + is - s;
}

// one workgroup: count[] goes through LDS in tiles of kScanTile counters (coalesced loads; ONE tile for last-fm's 23 553
// users), thread t owns the contiguous chunk [t*per, (t+1)*per) of the tile -- its sums go through a single workgroup
// scan, then it walks the chunk again writing the segment table; offs[] leaves LDS coalesced.  (It was a loop of
// n_user/1024 workgroup scans with three barriers each: 37 us for 23 553 users; walking the chunks straight from global
// memory is a chain of uncoalesced load latencies: ~20 us there, 85 us for amazon-book's 70 585.)
constexpr int kScanTile = 16384;      // two LDS arrays per tile: the counters / offsets and the segment numbers (128 KB)
__global__ __launch_bounds__(1024) void group_scan_kernel(const int32_t* __restrict__ count, int n_user, int64_t B,
                                                          int32_t* __restrict__ offs, int32_t* __restrict__ seg_user,
                                                          int32_t* __restrict__ seg_ptr, int32_t* __restrict__ nseg) {
    extern __shared__ int sCnt[];
    int* sSeg = sCnt + min(n_user, kScanTile);           // segment number of every user of the tile that occurs, -1 otherwise
    __shared__ int sC[16], sS[16];
    const int tid = threadIdx.x;
    int carryC = 0, carryS = 0;
    for (int t0 = 0; t0 < n_user; t0 += kScanTile) {
        const int n = min(kScanTile, n_user - t0);
        __syncthreads();                                      // previous tile written out; sC / sS free again
        // (eight loads in flight per thread: one load -> LDS store per trip was a chain of n / 1024 memory latencies, 23 in a row at last-fm)
        for (int u0 = tid; u0 < n; u0 += 8 * 1024) {
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = u0 + j * 1024 < n ? count[t0 + u0 + j * 1024] : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (u0 + j * 1024 < n) sCnt[u0 + j * 1024] = v[j];
        }
        __syncthreads();
        const int per = ((n + 1023) / 1024) | 1;              // odd: lane t reads word t*per + j -- no LDS bank conflicts
        const int u0 = min(tid * per, n), u1 = min(u0 + per, n);
        int c = 0, s = 0;
        for (int u = u0; u < u1; ++u) {
            const int v = sCnt[u];
            c += v;
            s += v > 0 ? 1 : 0;
        }
        int ec, es, totC, totS;
        block_scan2(c, s, sC, sS, ec, es, totC, totS);
        ec += carryC;
        es += carryS;
        for (int u = u0; u < u1; ++u) {
            const int v = sCnt[u];
            sCnt[u] = ec;
            sSeg[u] = v > 0 ? es : -1;
            es += v > 0 ? 1 : 0;
            ec += v;
        }
        carryC += totC;
        carryS += totS;
        __syncthreads();
        // the segment table leaves LDS with consecutive lanes on consecutive users (written from the chunk walk above, a lane's 23 segments
        // apart from its neighbour's, it was 47 k single-word write transactions from one workgroup: 2/3 of the kernel's 27 us)
        for (int u = tid; u < n; u += 1024) {
            const int o = sCnt[u], sg = sSeg[u];
            offs[t0 + u] = o;
            if (sg >= 0) {
                seg_user[sg] = t0 + u;
                seg_ptr[sg] = o;
            }
        }
    }
    if (tid == 0) {
        nseg[0] = carryS;
        seg_ptr[carryS] = carryC;                             // = B
    }
}

// The whole counting sort in ONE launch of ONE workgroup, the per-user counters in LDS (n_user * 4 bytes: 94 KB for
// last-fm): histogram by LDS atomics -> chunked scan -> scatter by LDS atomics.  Taken for batches a single workgroup
// walks in a few microseconds (a rank's share of a split batch, the reference's own batch sizes): the three-kernel form
// costs four launches and a memset whatever the batch (67 us at 65 536 pairs).
__global__ __launch_bounds__(1024) void group_small_kernel(const int64_t* __restrict__ u64, const int32_t* __restrict__ u32, int64_t B,
                                                           int n_user, int32_t* __restrict__ seg_user, int32_t* __restrict__ seg_ptr,
                                                           int32_t* __restrict__ nseg, int32_t* __restrict__ pair_index) {
    extern __shared__ int sCount[];
    __shared__ int sC[16], sS[16];
    const int tid = threadIdx.x;
    for (int u = tid; u < n_user; u += 1024) sCount[u] = 0;
    __syncthreads();
    // ids in batches of UN independent loads per thread: one load -> one atomic at a time the walk was a chain of
    // global-memory latencies (88 us for 65 536 pairs)
    constexpr int UN = 16;
    for (int64_t base = 0; base < B; base += 1024 * UN) {
        int u[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * 1024 + tid;
            u[j] = i < B ? (int)clamp_user(u64 ? u64[i] : (int64_t)u32[i], n_user) : -1;
        }
#pragma unroll
        for (int j = 0; j < UN; ++j)
            if (u[j] >= 0) atomicAdd(&sCount[u[j]], 1);
    }
    __syncthreads();
    const int per = ((n_user + 1023) / 1024) | 1;             // odd: no LDS bank conflicts between the lanes' chunks
    const int u0 = min(tid * per, n_user), u1 = min(u0 + per, n_user);
    int c = 0, s = 0;
    for (int u = u0; u < u1; ++u) {
        const int v = sCount[u];
        c += v;
        s += v > 0 ? 1 : 0;
    }
    int ec, es, totC, totS;
    block_scan2(c, s, sC, sS, ec, es, totC, totS);
    for (int u = u0; u < u1; ++u) {
        const int v = sCount[u];
        sCount[u] = ec;                                       // becomes the running write position of user u
        if (v > 0) {
            seg_user[es] = u;
            seg_ptr[es] = ec;
            ++es;
        }
        ec += v;
    }
    if (tid == 0) {
        nseg[0] = totS;
        seg_ptr[totS] = totC;
    }
    __syncthreads();
    for (int64_t base = 0; base < B; base += 1024 * UN) {
        int u[UN], pos[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int64_t i = base + j * 1024 + tid;
            u[j] = i < B ? (int)clamp_user(u64 ? u64[i] : (int64_t)u32[i], n_user) : -1;
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) pos[j] = u[j] >= 0 ? atomicAdd(&sCount[u[j]], 1) : 0;
#pragma unroll
        for (int j = 0; j < UN; ++j)
            if (u[j] >= 0) pair_index[pos[j]] = (int32_t)(base + j * 1024 + tid);
    }
}

__global__ void group_scatter_kernel(const int64_t* __restrict__ u64, const int32_t* __restrict__ u32, int64_t B, int n_user,
                                     const int32_t* __restrict__ offs, const int32_t* __restrict__ rank,
                                     int32_t* __restrict__ pair_index) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B; i += (int64_t)gridDim.x * blockDim.x) {
        pair_index[offs[clamp_user(u64 ? u64[i] : (int64_t)u32[i], n_user)] + rank[i]] = (int32_t)i;
    }
}

// largest batch the one-workgroup form takes and its LDS limit.  LDS atomics retire about one lane per clock, so one
// workgroup sorts ~1 pair per ns: measured against the three-kernel form (scripts/bench_group.py, last-fm's 23 553 users)
// 14 / 16 / 30 / 84 us at 512 / 4 096 / 16 384 / 65 536 pairs against 41 / 45 / 52 / 69
constexpr int64_t kGroupSmallMaxB = 32768;
constexpr size_t kGroupSmallMaxLds = 150 * 1024;

hipError_t launch_group_pairs(const int64_t* u64, const int32_t* u32, int64_t B, int n_user, int32_t* count, int32_t* offs,
                              int32_t* rank, int32_t* seg_user, int32_t* seg_ptr, int32_t* nseg, int32_t* pair_index, hipStream_t st) {
    static const int force = getenv("MVIN_GROUP_SMALL") ? atoi(getenv("MVIN_GROUP_SMALL")) : -1;   // A/B switch: 0 / 1
    const size_t lds = (size_t)n_user * sizeof(int32_t);
    const bool small = lds <= kGroupSmallMaxLds && (force < 0 ? B <= kGroupSmallMaxB : force == 1);
    if (small) {
        if (lds > 48 * 1024) {      // per launch: the attribute belongs to the current device's copy of the function
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(group_small_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGroupSmallMaxLds);
            if (e != hipSuccess) return e;
        }
        group_small_kernel<<<1, 1024, lds, st>>>(u64, u32, B, n_user, seg_user, seg_ptr, nseg, pair_index);
        return hipGetLastError();
    }
    hipError_t e = hipMemsetAsync(count, 0, (size_t)n_user * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const int blocks = (int)((B + 255) / 256 < 2048 ? (B + 255) / 256 : 2048);
    group_count_kernel<<<blocks, 256, 0, st>>>(u64, u32, B, n_user, count, rank);
    {
        const size_t scan_lds = (size_t)2 * (n_user < kScanTile ? n_user : kScanTile) * sizeof(int32_t);
        if (scan_lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(group_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(2 * kScanTile * sizeof(int32_t)));
            if (e != hipSuccess) return e;
        }
        group_scan_kernel<<<1, 1024, scan_lds, st>>>(count, n_user, B, offs, seg_user, seg_ptr, nseg);
    }
    group_scatter_kernel<<<blocks, 256, 0, st>>>(u64, u32, B, n_user, offs, rank, pair_index);
    return hipGetLastError();
}

}  // namespace mvin
