// Inputs of the scoring path built on the GPU (scope row f-1): the fixed-fan-out sampled
// adjacency (contruct_random_adj, data_loader_user_set.py:375-388) and the per-user ripple sets
// (_get_user_triplet_set, :407-441) from a CSR of the undirected KG (construct_kg, :324-343).
//
// The reference draws from numpy's / random's global unseeded generators, so only its sampling
// RULES can be reproduced, not its draws.  Here every draw is a pure function of
// (seed, stream, a, b, c) through a splitmix64 finaliser, which makes the kernels
// deterministic and lets oracle/prep_ref.py restate them bit-exactly in Python integers.
//   without replacement = Floyd's algorithm (uniform over k-subsets)
//   with replacement    = independent uniform draws
#include "mvin_kernels.h"

namespace mvin {

__device__ __forceinline__ uint32_t rnd32(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b, uint64_t c) {
    uint64_t z = seed ^ (stream * 0xD1B54A32D192ED03ull) ^ (a * 0x9E3779B97F4A7C15ull) ^
                 (b * 0xC2B2AE3D27D4EB4Full) ^ (c * 0x165667B19E3779F9ull);
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

// uniform integer in [0, n), n < 2^32 (multiply-high; bias < n / 2^32)
__device__ __forceinline__ uint32_t rnd_below(uint32_t n, uint64_t seed, uint64_t stream, uint64_t a, uint64_t b,
                                              uint64_t c) {
    return (uint32_t)(((uint64_t)rnd32(seed, stream, a, b, c) * n) >> 32);
}

// ---------------------------------------------------------------------------------------
// contruct_random_adj: K neighbors per entity; without replacement when deg >= K (:383),
// with replacement otherwise (:384); entities absent from the KG keep the all-zero row.
// One thread per entity; the chosen edge positions are staged in the output row itself.
// ---------------------------------------------------------------------------------------
__global__ void sample_adjacency_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ dst,
                                        const int32_t* __restrict__ rel, int n_entity, int K, uint64_t seed,
                                        int32_t* __restrict__ adj_e, int32_t* __restrict__ adj_r) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n_entity) return;
    const int64_t lo = indptr[x];
    const uint32_t deg = (uint32_t)(indptr[x + 1] - lo);
    int32_t* pe = adj_e + (int64_t)x * K;
    int32_t* pr = adj_r + (int64_t)x * K;
    if (deg == 0) {
        for (int i = 0; i < K; ++i) {
            pe[i] = 0;
            pr[i] = 0;
        }
        return;
    }
    if (deg >= (uint32_t)K) {  // Floyd: for j = deg-K .. deg-1: t = U[0..j]; take t unless taken, else j
        for (int i = 0; i < K; ++i) {
            const uint32_t j = deg - K + i;
            uint32_t t = rnd_below(j + 1, seed, 1, (uint64_t)x, i, 0);
            bool taken = false;
            for (int q = 0; q < i; ++q) taken |= ((uint32_t)pe[q] == t);
            pe[i] = (int32_t)(taken ? j : t);
        }
    } else {
        for (int i = 0; i < K; ++i) pe[i] = (int32_t)rnd_below(deg, seed, 1, (uint64_t)x, i, 0);
    }
    for (int i = 0; i < K; ++i) {
        const int64_t pos = lo + pe[i];
        pr[i] = rel[pos];
        pe[i] = dst[pos];
    }
}

// ---------------------------------------------------------------------------------------
// _get_user_triplet_set: one wave per user, hops in sequence.
//   seeds(hop 0) = the user's positive train items (in interaction order); seeds(hop h) = the
//   tails of hop h-1's memories (:415-418).  Every seed contributes min(deg, n_neighbor) of its
//   (tail, relation) edges, sampled without replacement (:421); from that candidate list
//   n_memory triples are drawn, with replacement iff there are fewer than n_memory (:433-437);
//   an empty candidate list copies the previous hop (:429-430).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void ripple_sets_kernel(RippleBuildArgs a) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Nm = a.Nm;
    int* sV = smem_i + wave * 2 * Nm;   // virtual candidate indices of this hop
    int* sT = sV + Nm;                  // tails of the previous hop

    for (int64_t u = (int64_t)blockIdx.x * 4 + wave; u < a.n_user; u += (int64_t)gridDim.x * 4) {
        int32_t* out = a.out + u * (int64_t)a.P * 3 * Nm;
        for (int h = 0; h < a.P; ++h) {
            const int32_t* seeds;
            int n_seeds;
            if (h == 0) {
                seeds = a.hist_items + a.hist_ptr[u];
                n_seeds = (int)(a.hist_ptr[u + 1] - a.hist_ptr[u]);
            } else {
                seeds = sT;
                n_seeds = Nm;
            }
            // candidate count (every lane computes the same total)
            uint32_t C = 0;
            for (int s = 0; s < n_seeds; ++s) {
                const int e = seeds[s];
                const uint32_t deg = (uint32_t)(a.indptr[e + 1] - a.indptr[e]);
                C += deg < (uint32_t)a.n_neighbor ? deg : (uint32_t)a.n_neighbor;
            }
            int32_t* oh = out + (int64_t)h * 3 * Nm;
            if (C == 0) {
                if (h == 0) {  // no usable history: the reference has no entry; all hops stay zero
                    for (int i = lane; i < a.P * 3 * Nm; i += 64) out[i] = 0;
                    break;
                }
                for (int m = lane; m < Nm; m += 64)  // :429-430 copy the previous hop (tails unchanged)
                    for (int x = 0; x < 3; ++x) oh[x * Nm + m] = oh[x * Nm + m - 3 * Nm];
                continue;
            }
            if (lane == 0) {
                if (C >= (uint32_t)Nm) {  // without replacement: Floyd over [0, C)
                    for (int i = 0; i < Nm; ++i) {
                        const uint32_t j = C - Nm + i;
                        uint32_t t = rnd_below(j + 1, a.seed, 2, (uint64_t)u, h, i);
                        bool taken = false;
                        for (int q = 0; q < i; ++q) taken |= ((uint32_t)sV[q] == t);
                        sV[i] = (int)(taken ? j : t);
                    }
                } else {
                    for (int i = 0; i < Nm; ++i) sV[i] = (int)rnd_below(C, a.seed, 2, (uint64_t)u, h, i);
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int m = lane; m < Nm; m += 64) {
                const uint32_t v = (uint32_t)sV[m];
                // locate the seed position s and the offset w inside its sub-sample
                uint32_t base = 0;
                int s = 0, e = 0;
                uint32_t deg = 0, cnt = 0;
                for (; s < n_seeds; ++s) {
                    e = seeds[s];
                    deg = (uint32_t)(a.indptr[e + 1] - a.indptr[e]);
                    cnt = deg < (uint32_t)a.n_neighbor ? deg : (uint32_t)a.n_neighbor;
                    if (v < base + cnt) break;
                    base += cnt;
                }
                const uint32_t w = v - base;
                uint32_t pick = w;                      // deg <= n_neighbor: all edges are candidates
                if (deg > (uint32_t)a.n_neighbor) {     // w-th element of a Floyd n_neighbor-subset of [0, deg)
                    uint32_t chosen[32];
                    for (int i = 0; i <= (int)w; ++i) {
                        const uint32_t j = deg - a.n_neighbor + i;
                        uint32_t t = rnd_below(j + 1, a.seed, 3, ((uint64_t)u << 8) | (uint64_t)h, s, i);
                        bool taken = false;
                        for (int q = 0; q < i; ++q) taken |= (chosen[q] == t);
                        chosen[i] = taken ? j : t;
                    }
                    pick = chosen[w];
                }
                const int64_t pos = a.indptr[e] + pick;
                oh[0 * Nm + m] = e;
                oh[1 * Nm + m] = a.rel[pos];
                oh[2 * Nm + m] = a.dst[pos];
            }
            __builtin_amdgcn_wave_barrier();
            // next hop's seeds: this hop's tails (read back by the lanes that wrote them)
            for (int m = lane; m < Nm; m += 64) sT[m] = oh[2 * Nm + m];
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------
// Duplicate-slot encoding of a sampled adjacency (consumed by mvin_fused_packed.hip).
// contruct_random_adj draws WITH replacement when deg < K (data_loader_user_set.py:383-384), so a row repeats
// (neighbour, relation) slots; sum_k p_k E[y_k] over the K slots = sum over the DISTINCT slots with the softmax weights of
// equal slots added up (equal slots have equal logits: weight = multiplicity * exp(t[r] - max) / Z).  Row x becomes: distinct
// slots first, ordered by the distinct-slot count of the neighbour's own row (descending, ties in first-occurrence order:
// the children of a tree node that share a gather round then have lists of similar length), padding (= slot 0) last;
//   enc_e = neighbour | cnt[neighbour] << 24   (n_entity <= 2^24; the length of the neighbour's own list, known one
//                                               fetch early)
//   enc_r = relation | multiplicity << 16 | cnt[x] << 24     (multiplicity 0: padding)
// One wave per row, two passes (the order needs every row's count): MODE 0 writes cnt, MODE 1 the encoding.
// oracle/prep_ref.py:encode_adjacency restates it; integer work, bit-exact.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void encode_adjacency_kernel(const int32_t* __restrict__ adj_e,
                                                                  const int32_t* __restrict__ adj_r, int n_entity, int K,
                                                                  int32_t* __restrict__ cnt, int32_t* __restrict__ enc_e,
                                                                  int32_t* __restrict__ enc_r) {
    __shared__ int sE[4][128], sR[4][128], sF[4][128], sPad[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 4 + wave;
    if (x >= n_entity) return;
    int e[2], r[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = lane + 64 * i;
        e[i] = s < K ? adj_e[(int64_t)x * K + s] : -1;
        r[i] = s < K ? (adj_r ? adj_r[(int64_t)x * K + s] : 0) : -1;
        if (s < K) {
            // ids are clamped BEFORE they are compared and packed, as every plain-adjacency kernel clamps them where it
            // indexes a table: an id outside its field would otherwise overwrite the count / multiplicity bytes
            // (the library's rule: an unsigned min -- ids beyond the field, negative ones included, become its last value)
            e[i] = (int)min((unsigned)e[i], (unsigned)(n_entity - 1));
            r[i] = (int)min((unsigned)r[i], 0xFFFFu);
            sE[wave][s] = e[i];
            sR[wave][s] = r[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    bool first[2];
    int mult[2];
    int c = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = lane + 64 * i;
        first[i] = s < K;
        mult[i] = 0;
        for (int t = 0; t < K; ++t) {
            const bool eq = sE[wave][t] == e[i] && sR[wave][t] == r[i];
            mult[i] += eq ? 1 : 0;
            if (eq && t < s) first[i] = false;
        }
        c += __popcll(__ballot(first[i]));
    }
    if constexpr (MODE == 0) {
        if (lane == 0) cnt[x] = c;
        return;
    } else {
        int key[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = lane + 64 * i;
            key[i] = first[i] ? cnt[e[i] < 0 ? 0 : e[i]] : -1;
            if (s < K) sF[wave][s] = key[i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int s = lane + 64 * i;
            if (first[i]) {
                int rank = 0;
                for (int t = 0; t < K; ++t) {
                    const int kt = sF[wave][t];
                    rank += (kt > key[i] || (kt == key[i] && t < s)) ? 1 : 0;
                }
                enc_e[(int64_t)x * K + rank] = (int32_t)((unsigned)e[i] | ((unsigned)key[i] << 24));
                enc_r[(int64_t)x * K + rank] = (int32_t)((unsigned)r[i] | ((unsigned)mult[i] << 16) | ((unsigned)c << 24));
                if (rank == 0) {
                    sPad[wave][0] = (int32_t)((unsigned)e[i] | ((unsigned)key[i] << 24));
                    sPad[wave][1] = r[i];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int s = c + lane; s < K; s += 64) {
            enc_e[(int64_t)x * K + s] = sPad[wave][0];
            enc_r[(int64_t)x * K + s] = (int32_t)((unsigned)sPad[wave][1] | ((unsigned)c << 24));
        }
    }
}

hipError_t launch_encode_adjacency(const int32_t* adj_e, const int32_t* adj_r, int n_entity, int K, int32_t* cnt,
                                   int32_t* enc_e, int32_t* enc_r, hipStream_t st) {
    const int grid = (n_entity + 3) / 4;
    encode_adjacency_kernel<0><<<grid, kBlock, 0, st>>>(adj_e, adj_r, n_entity, K, cnt, nullptr, nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    encode_adjacency_kernel<1><<<grid, kBlock, 0, st>>>(adj_e, adj_r, n_entity, K, cnt, enc_e, enc_r);
    return hipGetLastError();
}

hipError_t launch_sample_adjacency(const int64_t* indptr, const int32_t* dst, const int32_t* rel, int n_entity,
                                   int K, uint64_t seed, int32_t* adj_e, int32_t* adj_r, hipStream_t st) {
    sample_adjacency_kernel<<<(n_entity + 255) / 256, 256, 0, st>>>(indptr, dst, rel, n_entity, K, seed, adj_e, adj_r);
    return hipGetLastError();
}

hipError_t launch_ripple_build(const RippleBuildArgs& a, hipStream_t st) {
    const int64_t nblk = (a.n_user + 3) / 4;
    const size_t lds = (size_t)4 * 2 * a.Nm * sizeof(int);
    ripple_sets_kernel<<<(int)(nblk < 4096 ? nblk : 4096), kBlock, lds, st>>>(a);
    return hipGetLastError();
}

}  // namespace mvin
