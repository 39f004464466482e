// MVIN._key_addressing attention reads (model.py:161-240) for per-pair ripple sets (the feed of
// train.py:117-120: every pair brings its own [Nm] head / relation / tail id lists per hop), as a
// STREAMING pipeline over LDS-DMA: one wave per workgroup, 16 workgroups per CU, every table row goes
// global -> LDS with global_load_lds_dwordx4 (no VGPR staging) in stages of two 1 KB pieces, one stage ahead of
// the one being reduced.  The register-resident kernel (mvin_keyaddr.hip) holds the 2*Nm rows of one hop in
// VGPRs (240 registers, 2 waves per SIMD) and its id loads, row loads, V loads and arithmetic run strictly one
// after another; here a wave needs ~100 registers and 8 KB of LDS, so 16 waves per CU keep the memory system
// busy while each of them alternates between landing and reducing small stages.
//   hop logits    s_m = h_m . V[b, r_m, :]   with V[b,r,:] = E[item_b] . R_KGE[r]  ((R h).v == h.(v R))
//   h-set logits  s_m = h0_m . w_h           (user term and bias cancel in the softmax, :171-189)
//   o = sum_m softmax(s)_m t_m  (hops, :223-229)   /   sum_m softmax(s)_m h0_m  (h-set, :189-195)
//
// What shaped it (measured with scripts/micro/dma_probe.hip on MI355X):
//  * an s_mov to M0 (the LDS base of an LDS-DMA instruction) stalls until every DMA piece the wave has in flight
//    is done: one M0 write per piece runs one piece at a time (~660 cycles each on random 256-byte rows), which is
//    what the first version of this kernel did (7.97 ms per 524 288 C3 pairs against 6.64 ms for the register
//    kernel).  So M0 is written ONCE per wave and everything the wave ever lands -- two stage slots, the id lists
//    and the pair's V block -- lives inside the 8 KB window the 13-bit instruction offset reaches around it;
//    the instruction offset also moves the global address, so every source pointer is pre-biased by -offset.
//  * all vector memory reads of the steady state are DMA pieces, issued from inline asm: vmcnt then counts only
//    them, they complete in issue order, and "stage s has landed" is `s_waitcnt vmcnt(<pieces issued after it>)`
//    with a compile-time count.  (For the builtin, hipcc of ROCm 7.2 puts a vmcnt(0) in front of every later LDS
//    read, which drains the stage in flight.)
//  * uniformly random 256-byte rows of a 27 MB table come out of the Infinity Cache at <= 8.8 TB/s for the whole
//    chip however they are requested (4 waves per CU with 4 pieces in flight each already get 8.3): that, not
//    the load path, bounds this kernel on the synthetic ripple sets of bench.py.
#include <type_traits>
#include <utility>

#include "mvin_kernels.h"

namespace mvin {

template <int OFF>
__device__ __forceinline__ void dma16_at(const char* g) {       // 16 B per lane -> M0 + OFF + lane * 16
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(g - OFF), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void dma4_at(const char* g) {        // 4 B per lane -> M0 + OFF + lane * 4
    asm volatile("global_load_lds_dword %0, off offset:%1" ::"v"(g - OFF), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int V>
using ic = std::integral_constant<int, V>;

__device__ __forceinline__ float dot4s(float4 a, float4 b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// window layout, byte offsets from the wave's LDS base (M0 = base + 4096)
constexpr int kWinSlot = 2048;        // two stage slots of two pieces: [0, 4096)
constexpr int kWinIds = 4096;         // three id lists, 256 B each
constexpr int kWinV = 5120;           // the pair's V block, at most 3 pieces
constexpr int kWinBytes = 8192;

// NMP: the memory count padded to a power of two (16 / 32 / 64); rows past Nm re-read row Nm-1 and get weight 0
template <int D, bool BF, int NMP>
__global__ __launch_bounds__(64) void key_addr_stream_kernel(KeyAddrArgs a) {
    constexpr int RB = D * (BF ? 2 : 4);      // row bytes
    constexpr int LPR = RB / 16;              // lanes per row
    constexpr int L2 = LPR == 4 ? 2 : LPR == 8 ? 3 : LPR == 16 ? 4 : 5;
    constexpr int RPI = 64 / LPR;             // rows per DMA piece
    constexpr int NIT = NMP / RPI;            // pieces per id list
    constexpr int SP = NIT >= 2 ? 2 : 1;      // pieces per stage
    constexpr int NST = NIT / SP;             // stages per list
    constexpr int EPL = BF ? 8 : 4;           // table elements per lane
    static_assert(LPR >= 4 && LPR <= 32 && NIT >= 1, "shape outside the streaming kernel");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int lane = threadIdx.x;
    const int g = lane / LPR, c = lane % LPR;
    const int Nm = a.Nm, P = a.P;
    const int slot0 = a.w ? 1 : 0;
    const int64_t NW = gridDim.x;
    const int lm = lane < Nm ? lane : Nm - 1;
    const bool do_set = a.w != nullptr;
    const int vbytes = a.nR * D * 4;
    float4 wv0 = make_float4(0.f, 0.f, 0.f, 0.f), wv1 = wv0;
    if (do_set) {
        wv0 = reinterpret_cast<const float4*>(a.w)[BF ? 2 * c : c];
        if constexpr (BF) wv1 = reinterpret_cast<const float4*>(a.w)[2 * c + 1];
    }
    // force the weights into registers now: the only register-returning vector loads of the kernel
    asm volatile("" : "+v"(wv0.x), "+v"(wv0.y), "+v"(wv0.z), "+v"(wv0.w));
    asm volatile("" : "+v"(wv1.x), "+v"(wv1.y), "+v"(wv1.z), "+v"(wv1.w));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds0 + 4096) : "memory");     // the ONLY M0 write of the wave

    const char* tab = reinterpret_cast<const char*>(a.E);
    const unsigned emax = (unsigned)(a.n_entity > 0 ? a.n_entity - 1 : 0x7fffffff);      // last row of E
    const int* sIds = reinterpret_cast<const int*>(smem + kWinIds);
    const float* sV = reinterpret_cast<const float*>(smem + kWinV);

    auto issue_ids = [&](int64_t b, int hop) {
        const KeyAddrLists l = key_addr_lists(a, b, hop);      // users feed: one uniform load of users[b] (drains the DMA queue:
        dma4_at<kWinIds - 4096>(reinterpret_cast<const char*>(l.h + lm));            // only the last tail stage is in flight here)
        dma4_at<kWinIds - 4096 + 256>(reinterpret_cast<const char*>(l.r + lm));
        dma4_at<kWinIds - 4096 + 512>(reinterpret_cast<const char*>(l.t + lm));
    };
    auto issue_v = [&](int64_t b) {      // always three pieces (a constant for the wait counts); tail lanes re-read the end
        const char* src = reinterpret_cast<const char*>(a.V + b * a.nR * (int64_t)D);
        auto piece = [&](int i) -> const char* {
            const int off = i * 1024 + lane * 16;
            return src + (off < vbytes ? off : vbytes - 16);
        };
        dma16_at<kWinV - 4096>(piece(0));
        dma16_at<kWinV - 4096 + 1024>(piece(1));
        dma16_at<kWinV - 4096 + 2048>(piece(2));
    };
    // stage ST (pieces ST*SP ..) of id list WHICH (0 = heads, 2 = tails) -> slot SLOT
    auto issue_stage = [&](auto which_c, auto st_c, auto slot_c) {
        constexpr int WHICH = decltype(which_c)::value, ST = decltype(st_c)::value, SLOT = decltype(slot_c)::value;
        wait_lds_reads();      // the slot's previous rows have been read
        const char* src[SP];
#pragma unroll
        for (int k = 0; k < SP; ++k) {
            int m = (ST * SP + k) * RPI + g;
            m = m < Nm ? m : Nm - 1;
            src[k] = tab + (size_t)min((unsigned)sIds[WHICH * 64 + m], emax) * RB + c * 16;     // clamped into the table
        }
        dma16_at<SLOT * kWinSlot - 4096>(src[0]);
        if constexpr (SP == 2) dma16_at<SLOT * kWinSlot - 4096 + 1024>(src[1]);
    };
    // this lane's 16 bytes of piece k of a slot, widened to fp32
    auto chunk = [&](int slot, int k, float4& lo, float4& hi) {
        const char* p = smem + slot * kWinSlot + k * 1024 + lane * 16;
        if constexpr (BF) {
            const uint4 r = *reinterpret_cast<const uint4*>(p);
            lo = bf16x4_to_f32(make_uint2(r.x, r.y));
            hi = bf16x4_to_f32(make_uint2(r.z, r.w));
        } else {
            lo = *reinterpret_cast<const float4*>(p);
            hi = lo;
        }
    };
    // reductions over the row groups of a wave (values are already equal inside a group)
    auto groups_max = [&](float v) {
#pragma unroll
        for (int o = LPR; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
        if (LPR <= 16) v = xor16_max(v);
        if (LPR <= 32) v = xor32_max(v);
        return v;
    };
    auto groups_sum = [&](float v) {
#pragma unroll
        for (int o = LPR; o < 16; o <<= 1) v += __shfl_xor(v, o, kWave);
        if (LPR <= 16) v = xor16_sum(v);
        if (LPR <= 32) v = xor32_sum(v);
        return v;
    };
    auto store_o = [&](float* o, float4 a0, float4 a1, float inv) {
        if (g == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(a0.x * inv, a0.y * inv, a0.z * inv, a0.w * inv);
            if constexpr (BF) *reinterpret_cast<float4*>(o + 4) = make_float4(a1.x * inv, a1.y * inv, a1.z * inv, a1.w * inv);
        }
    };

    int64_t b = blockIdx.x;
    if (b >= a.B) return;
    int hop = 0;
    issue_ids(b, 0);
    issue_v(b);
    for (;;) {
        wait_dma<0>();       // this task's id lists (and, for a new pair, its V block) have landed
        // ================= heads: logits of the Nm memories; the h-set read in one online-softmax pass =================
        const bool set_now = do_set && hop == 0;
        float ph[NIT];                      // this lane's row group's logits, one per piece
        float ms = -INFINITY, zs = 0.f;     // h-set: running max, running sum ...
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;     // ... and un-normalised weighted rows
        issue_stage(ic<0>{}, ic<0>{}, ic<0>{});
        static_for<NST>([&](auto st_c) {
            constexpr int ST = decltype(st_c)::value, SLOT = ST & 1;
            if constexpr (ST + 1 < NST) issue_stage(ic<0>{}, ic<ST + 1>{}, ic<(SLOT ^ 1)>{});
            else issue_stage(ic<2>{}, ic<0>{}, ic<(SLOT ^ 1)>{});      // first tail stage takes the other slot
            wait_dma<SP>();
            float4 lo[SP], hi[SP];
            float ps[SP];
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                const int m = (ST * SP + k) * RPI + g;
                const int r = sIds[64 + (m < Nm ? m : Nm - 1)];
                chunk(SLOT, k, lo[k], hi[k]);
                const float* vr = sV + r * D + EPL * c;
                float p = dot4s(lo[k], *reinterpret_cast<const float4*>(vr));
                if constexpr (BF) p += dot4s(hi[k], *reinterpret_cast<const float4*>(vr + 4));
                p = group_sum(p, L2);
                ph[ST * SP + k] = m < Nm ? p : -INFINITY;
                ps[k] = -INFINITY;
                if (set_now) {
                    float q = dot4s(lo[k], wv0);
                    if constexpr (BF) q += dot4s(hi[k], wv1);
                    q = group_sum(q, L2);
                    ps[k] = m < Nm ? q : -INFINITY;
                }
            }
            if (set_now) {                  // o_hset = sum_m softmax(s)_m h0_m (model.py:189-195), rows seen once
                float mx = ps[0];
                if constexpr (SP == 2) mx = fmaxf(mx, ps[1]);
                mx = fmaxf(groups_max(mx), ms);                 // finite from the first stage on (row 0 is valid)
                const float sc = expf(ms - mx);                 // exp(-inf) = 0 at the first stage
                zs *= sc;
                s0 = make_float4(s0.x * sc, s0.y * sc, s0.z * sc, s0.w * sc);
                if constexpr (BF) s1 = make_float4(s1.x * sc, s1.y * sc, s1.z * sc, s1.w * sc);
#pragma unroll
                for (int k = 0; k < SP; ++k) {
                    const float e = ps[k] != -INFINITY ? expf(ps[k] - mx) : 0.f;
                    zs += e;
                    s0 = f4_fma(e, lo[k], s0);
                    if constexpr (BF) s1 = f4_fma(e, hi[k], s1);
                }
                ms = mx;
            }
        });
        if (set_now) {
            s0 = group_xor_sum(s0, LPR);
            if constexpr (BF) s1 = group_xor_sum(s1, LPR);
            store_o(a.out + b * a.ldo + EPL * c, s0, s1, 1.f / groups_sum(zs));
        }
        // softmax over the Nm memories (tf.nn.softmax, model.py:223); every lane holds its row group's weights
        float mx = ph[0];
#pragma unroll
        for (int j = 1; j < NIT; ++j) mx = fmaxf(mx, ph[j]);
        mx = groups_max(mx);
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < NIT; ++j) {
            ph[j] = ph[j] != -INFINITY ? expf(ph[j] - mx) : 0.f;
            z += ph[j];
        }
        z = groups_sum(z);
        // what comes next for this wave
        int nhop = hop + 1;
        int64_t nb = b;
        if (nhop == P) {
            nhop = 0;
            nb += NW;
        }
        const bool more = nb < a.B;
        if (more && nhop == 0) {     // V of the next pair: every head stage of this one is done
            wait_lds_reads();
            issue_v(nb);
        }
        // ================= tails: o = sum_m p_m t_m (model.py:229) =================
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        static_for<NST>([&](auto st_c) {
            constexpr int ST = decltype(st_c)::value, SLOT = (NST + ST) & 1;      // stages alternate slots through the task
            if constexpr (ST + 1 < NST) {
                issue_stage(ic<2>{}, ic<ST + 1>{}, ic<(SLOT ^ 1)>{});
                wait_dma<SP>();
            } else {
                // last tail stage: the id lists are no longer needed -> fetch the next task's behind it
                wait_lds_reads();
                if (more) {
                    issue_ids(nb, nhop);
                    wait_dma<3>();
                } else {
                    wait_dma<0>();
                }
            }
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                float4 lo, hi;
                chunk(SLOT, k, lo, hi);
                const float wgt = ph[ST * SP + k];
                a0 = f4_fma(wgt, lo, a0);
                if constexpr (BF) a1 = f4_fma(wgt, hi, a1);
            }
        });
        a0 = group_xor_sum(a0, LPR);
        if constexpr (BF) a1 = group_xor_sum(a1, LPR);
        store_o(a.out + b * a.ldo + (int64_t)(slot0 + hop) * D + EPL * c, a0, a1, 1.f / z);
        if (!more) break;
        b = nb;
        hop = nhop;
    }
    wait_dma<0>();      // nothing of this wave may still be writing LDS when the allocation is handed on
}

static int stream_nmp(int Nm) { return Nm <= 16 ? 16 : Nm <= 32 ? 32 : 64; }

// P >= 1 hops, Nm <= 64, rows of 64..512 bytes, V block of at most three pieces
bool key_addr_stream_supported(const KeyAddrArgs& a, int table_bf16) {
    static const char* env = getenv("MVIN_KA_STREAM");
    if (env && env[0] == '0') return false;
    if (a.P < 1 || a.Nm < 1 || a.Nm > 64 || !a.V) return false;
    if (!(a.D == 16 || a.D == 32 || a.D == 64 || a.D == 128)) return false;
    const int rb = a.D * (table_bf16 ? 2 : 4);
    if (rb < 64 || rb > 512) return false;
    if (stream_nmp(a.Nm) < 64 / (rb / 16)) return false;          // at least one full piece per list
    const size_t vbytes = (size_t)a.nR * a.D * 4;
    return vbytes >= 16 && vbytes <= 3072;
}

template <int D, bool BF>
static hipError_t launch_stream_nm(const KeyAddrArgs& a, hipStream_t st) {
    const int nmp = stream_nmp(a.Nm);
    const int64_t cap = 256 * 16;
    const int grid = (int)(a.B < cap ? a.B : cap);
    constexpr int RB = D * (BF ? 2 : 4), RPI = 64 / (RB / 16);
    if (nmp == 16) {
        if constexpr (16 >= RPI) {
            key_addr_stream_kernel<D, BF, 16><<<grid, 64, kWinBytes, st>>>(a);
            return hipGetLastError();
        }
    } else if (nmp == 32) {
        if constexpr (32 >= RPI) {
            key_addr_stream_kernel<D, BF, 32><<<grid, 64, kWinBytes, st>>>(a);
            return hipGetLastError();
        }
    } else {
        key_addr_stream_kernel<D, BF, 64><<<grid, 64, kWinBytes, st>>>(a);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

hipError_t launch_key_addr_stream(const KeyAddrArgs& a, int table_bf16, hipStream_t st) {
    if (table_bf16) {
        switch (a.D) {
            case 32: return launch_stream_nm<32, true>(a, st);
            case 64: return launch_stream_nm<64, true>(a, st);
            case 128: return launch_stream_nm<128, true>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.D) {
        case 16: return launch_stream_nm<16, false>(a, st);
        case 32: return launch_stream_nm<32, false>(a, st);
        case 64: return launch_stream_nm<64, false>(a, st);
        case 128: return launch_stream_nm<128, false>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mvin
