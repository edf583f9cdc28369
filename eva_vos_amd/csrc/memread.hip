// memread.hip - space-time memory read of STCN on gfx950:
//   affinity S = (2 mk.qk - |mk|^2 - |qk|^2)/sqrt(64)          (reference prop_net.py:80-90)
//   per query column: top-50 over the T*H*W memory rows, softmax over the 50 (prop_net.py:53-60)
//   readout = sum_j w_j * mv[idx_j]                             (prop_net.py:108-115)
// The reference materialises the dense [T*HW x HW] affinity and multiplies by the dense matrix; here S
// only ever lives in registers.  Exact two-pass selection:
//   pass 1 (affinity_pass<false>): S tiles on v_mfma_f32_16x16x4_f32 (A = memory keys as 16-B fragments
//     straight from global/L2, double-buffered in registers; B = the wave's 16 query keys, resident;
//     -|mk|^2/2 is the initial accumulator).  Every lane keeps running maxima over disjoint row groups
//     of its query column.  The 50th largest of a query's G >= 50 group maxima is a lower bound of its
//     true 50th largest score (they are 50 distinct elements), and a tight one (expected rank ~ 63 for
//     G = 128 groups): threshold_kernel computes it with a wave-wide radix select.
//   pass 2 (affinity_pass<true>): the same MFMA walk; a register-level pre-filter against the now GLOBAL
//     threshold makes candidates rare (~1.3 x 50 per query over the whole bank), they are appended to
//     per-(query, chunk) LDS lists with LDS atomics; an overflowing list is cut back to its best 50 by
//     the radix select (only adversarial orderings get there).
//   merge_readout: one wave per query merges the chunk lists, softmaxes the 50 with wavefront
//     reductions and gathers 50 value rows (2 KB each, NHWC bank) per object.
//   The column-constant -|qk|^2 term cancels in exp(v - v_max) and is dropped.
#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TOPK = 50;
static constexpr int CAP = 128;       // per-(query, chunk) candidate list capacity (>= TOPK + 64)
static constexpr int TROWS = 128;     // memory rows per tile of the attention read
static constexpr int HROWS = 64;      // rows per step of the top-k passes
static constexpr int SLD = 132;       // attention-read S slab row stride (floats)
static constexpr int MAXCHUNK = 16;   // row chunks (grid.y) of the top-k passes
static constexpr int NGRP = 16;       // running maxima per lane-group: 4 lane groups x 4 row blocks

__device__ __forceinline__ unsigned f2key(float f) {          // order-preserving float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ int lanes_below(unsigned long long m, int lane) {
    return __popcll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Keep the best TOPK of the n (<= 128) entries of a wave-owned list (values lv, payload li): bitwise
// radix select on order-preserving keys (ballot + popcount per bit), ties by list position.
// Returns the TOPK-th best value; the list is compacted to exactly TOPK entries.
__device__ __forceinline__ float wave_select128(float *lv, int *li, int n, int lane) {
    lds_fence();
    const float f0 = lv[lane], f1 = lv[lane + 64];
    const int i0 = li[lane], i1 = li[lane + 64];
    const unsigned k0 = lane < n ? f2key(f0) : 0u, k1 = lane + 64 < n ? f2key(f1) : 0u;   // key 0 < every real key
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int c = __popcll(__ballot(k0 >= cand)) + __popcll(__ballot(k1 >= cand));
        if (c >= TOPK) prefix = cand;
    }
    const unsigned long long g0 = __ballot(k0 > prefix), g1 = __ballot(k1 > prefix);
    const unsigned long long e0 = __ballot(k0 == prefix), e1 = __ballot(k1 == prefix);
    const int need = TOPK - (__popcll(g0) + __popcll(g1));              // >= 1 ties to keep
    const bool keep0 = (k0 > prefix) || (k0 == prefix && lanes_below(e0, lane) < need);
    const bool keep1 = (k1 > prefix) || (k1 == prefix && __popcll(e0) + lanes_below(e1, lane) < need);
    const unsigned long long b0 = __ballot(keep0), b1 = __ballot(keep1);
    lds_fence();
    if (keep0) { const int p = lanes_below(b0, lane); lv[p] = f0; li[p] = i0; }
    if (keep1) { const int p = __popcll(b0) + lanes_below(b1, lane); lv[p] = f1; li[p] = i1; }
    lds_fence();
    return key2f(prefix);
}

// A fragments of one 64-row step: a[rb][kb] = mk[row0 + 16 rb + (lane&15)][16 kb + 4 (lane>>4) .. +3]
// (k-permuted like the conv kernel) and the accumulator init -|mk|^2/2 for this lane's C rows.
struct HalfFrag { f32x4 a[4][4]; f32x4 c[4]; };
__device__ __forceinline__ void load_half(const float *__restrict__ mk, const float *__restrict__ msq, int N,
                                          int row0, int lane, HalfFrag &h) {
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        int r = row0 + rb * 16 + col;
        r = r < N ? r : N - 1;
        const float *ap = mk + (long)r * 64 + 4 * g;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) h.a[rb][kb] = *reinterpret_cast<const f32x4 *>(ap + 16 * kb);
        // C layout 16x16: col = lane&15, row = 4*(lane>>4) + reg.  msq is padded by >= 64 readable floats.
        h.c[rb] = *reinterpret_cast<const f32x4 *>(msq + row0 + rb * 16 + 4 * g);
    }
}
// acc[rb][j] = (mk[row] . qk[col] - msq[row]/2) / 4 with row = row0 + 16 rb + 4 (lane>>4) + j, col = lane&15
__device__ __forceinline__ void mfma_half(const HalfFrag &h, const f32x4 (&bq)[4], f32x4 (&acc)[4]) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = h.c[rb] * -0.5f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h.a[rb][kb].x, bq[kb].x, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h.a[rb][kb].y, bq[kb].y, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h.a[rb][kb].z, bq[kb].z, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(h.a[rb][kb].w, bq[kb].w, acc[rb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) acc[rb] = acc[rb] * 0.25f;
}

__device__ __forceinline__ void load_bq(const float *__restrict__ qk, int Q, int q0, int lane, f32x4 (&bq)[4]) {
    int q = q0 + (lane & 15);
    q = q < Q ? q : Q - 1;
    const float *bp = qk + (long)q * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bq[kb] = *reinterpret_cast<const f32x4 *>(bp + 16 * kb);
}

// COLLECT == false: pass 1, writes gmax[(chunk*NGRP + 4*(lane>>4) + rb)][q] (running maxima).
// COLLECT == true : pass 2, filters against tau[q] and writes the chunk's winners cand[chunk][q][TOPK].
template <int WAVES, bool COLLECT>
__global__ __launch_bounds__(64 * WAVES) void affinity_pass_kernel(
    const float *__restrict__ mk, const float *__restrict__ msq, const float *__restrict__ qk, int N, int Q,
    int steps_per_chunk, float *__restrict__ gmax, const float *__restrict__ tau_in, float *__restrict__ cand_v,
    int32_t *__restrict__ cand_i) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int PER_WAVE = 2 * 16 * CAP + 32;                         // floats of LDS per wave (pass 2)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *LV = smem + wave * PER_WAVE;                                 // [16][CAP]
    int *LI = reinterpret_cast<int *>(LV + 16 * CAP);                   // [16][CAP]
    int *CNT = LI + 16 * CAP;                                           // [16]
    float *TAU = reinterpret_cast<float *>(CNT + 16);                   // [16]

    const int q0 = (blockIdx.x * WAVES + wave) * 16;
    if (q0 >= Q) return;                                                // wave-uniform; no block barriers below
    const int chunk = blockIdx.y;
    const int nsteps = (N + HROWS - 1) / HROWS;
    const int h0 = chunk * steps_per_chunk;
    const int h1 = min(nsteps, h0 + steps_per_chunk);
    const int g = lane >> 4, col = lane & 15;
    const int qcol = min(q0 + col, Q - 1);

    f32x4 bq[4];
    load_bq(qk, Q, q0, lane, bq);
    float tcol = -__builtin_inff();
    f32x4 gm = {tcol, tcol, tcol, tcol};                                // pass 1: maxima per row block rb
    if (COLLECT) {
        tcol = tau_in[qcol];
        if (lane < 16) { CNT[lane] = 0; TAU[lane] = tcol; }
        lds_fence();
    }

    HalfFrag cur, nxt;
    if (h0 < h1) load_half(mk, msq, N, h0 * HROWS, lane, cur);
    for (int h = h0; h < h1; ++h) {
        const int row0 = h * HROWS;
        // prefetch the next step's fragments under this step's MFMAs (clamped re-load on the last step)
        load_half(mk, msq, N, min(h + 1, h1 - 1) * HROWS, lane, nxt);
        f32x4 acc[4];
        mfma_half(cur, bq, acc);
        const bool tail = row0 + HROWS > N;
        if (!COLLECT) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = (tail && row0 + rb * 16 + 4 * g + j >= N) ? -__builtin_inff() : acc[rb][j];
                    gm[rb] = fmaxf(gm[rb], v);
                }
        } else {
            bool hit = false;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    hit |= (acc[rb][j] > tcol) && !(tail && row0 + rb * 16 + 4 * g + j >= N);
            if (__ballot(hit)) {                                        // rare once tau is global
                // make room: any list that could overflow in this step is cut back to its best TOPK
                const int cn = CNT[col];
                unsigned long long full = __ballot(cn > CAP - HROWS);
                unsigned fq = (unsigned)((full | (full >> 16) | (full >> 32) | (full >> 48)) & 0xffffull);
                while (fq) {
                    const int j = __builtin_ctz(fq);
                    fq &= fq - 1;
                    const int c = __builtin_amdgcn_readfirstlane(CNT[j]);
                    const float t = wave_select128(LV + j * CAP, LI + j * CAP, c, lane);
                    if (lane == 0) { CNT[j] = TOPK; TAU[j] = fmaxf(TAU[j], t); }
                    lds_fence();
                }
                tcol = TAU[col];
                if (hit) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int row = row0 + rb * 16 + 4 * g + j;
                            if (acc[rb][j] > tcol && row < N) {
                                const int p = atomicAdd(&CNT[col], 1);
                                LV[col * CAP + p] = acc[rb][j];
                                LI[col * CAP + p] = row;
                            }
                        }
                }
                lds_fence();
            }
        }
        cur = nxt;
    }
    if (!COLLECT) {
        if (q0 + col < Q) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) gmax[((long)chunk * NGRP + 4 * g + rb) * Q + q0 + col] = gm[rb];
        }
        return;
    }
    // chunk winners -> global: cand[chunk][q][TOPK] (missing entries = -inf)
    lds_fence();
    for (int j = 0; j < 16; ++j) {
        float *lv = LV + j * CAP;
        int *li = LI + j * CAP;
        int cnt = __builtin_amdgcn_readfirstlane(CNT[j]);
        if (cnt > TOPK) { wave_select128(lv, li, cnt, lane); cnt = TOPK; }
        lds_fence();
        const int q = q0 + j;
        if (q < Q && lane < TOPK) {
            const long o = ((long)chunk * Q + q) * TOPK + lane;
            cand_v[o] = lane < cnt ? lv[lane] : -__builtin_inff();
            cand_i[o] = lane < cnt ? li[lane] : 0;
        }
    }
}

// tau[q] = TOPK-th largest of the G = NC*NGRP group maxima of query q (-inf when fewer than TOPK are finite)
__global__ __launch_bounds__(256) void threshold_kernel(const float *__restrict__ gmax, int G, int Q,
                                                        float *__restrict__ tau) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    unsigned k[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int gi = lane + 64 * e;
        const float v = gi < G ? gmax[(long)gi * Q + q] : -__builtin_inff();
        k[e] = v > -__builtin_inff() ? f2key(v) : 0u;
    }
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        const int c = __popcll(__ballot(k[0] >= cand)) + __popcll(__ballot(k[1] >= cand)) +
                      __popcll(__ballot(k[2] >= cand)) + __popcll(__ballot(k[3] >= cand));
        if (c >= TOPK) prefix = cand;
    }
    // prefix == 0: fewer than TOPK finite maxima -> no usable bound.  The bound must stay BELOW the
    // TOPK-th best (the filter keeps v > tau): step one key down.
    if (lane == 0) tau[q] = prefix == 0u ? -__builtin_inff() : key2f(prefix - 1u);
}

// one wave per query: merge NC*50 chunk winners, softmax, sparse readout
__global__ __launch_bounds__(256) void merge_readout_kernel(const float *__restrict__ cand_v,
                                                            const int32_t *__restrict__ cand_i, int NC, int Q,
                                                            const float *__restrict__ mv, long mv_os, int k,
                                                            float *__restrict__ readout, long ro_os,
                                                            int32_t *__restrict__ topk_idx, float *__restrict__ topk_w) {
    __shared__ float s_v[4][MAXCHUNK * TOPK];
    __shared__ int s_i[4][MAXCHUNK * TOPK];
    __shared__ float s_w[4][64];
    __shared__ int s_x[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    if (q >= Q) return;
    float *sv = s_v[wave];
    int *si = s_i[wave];
    const int n = NC * TOPK;
    for (int e = lane; e < n; e += 64) {
        const int c = e / TOPK, j = e - c * TOPK;
        const long o = ((long)c * Q + q) * TOPK + j;
        sv[e] = cand_v[o];
        si[e] = cand_i[o];
    }
    lds_fence();
    // bitwise radix select of the TOPK-th largest key over n entries
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = prefix | (1u << bit);
        int c = 0;
        for (int e = lane; e < n; e += 64) c += f2key(sv[e]) >= cand ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (c >= TOPK) prefix = cand;
    }
    // compact winners (ties: first in list order) into s_w / s_x
    int base = 0, ngt = 0;
    for (int e = lane; e < n; e += 64) ngt += f2key(sv[e]) > prefix ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) ngt += __shfl_xor(ngt, o);
    int need = TOPK - ngt;
    for (int e0 = 0; e0 < n; e0 += 64) {
        const int e = e0 + lane;
        const unsigned key = e < n ? f2key(sv[e]) : 0u;
        const unsigned long long eq = __ballot(e < n && key == prefix);
        const bool keep = e < n && (key > prefix || (key == prefix && lanes_below(eq, lane) < need));
        const unsigned long long kb = __ballot(keep);
        if (keep) {
            const int p = base + lanes_below(kb, lane);
            s_w[wave][p] = sv[e];
            s_x[wave][p] = si[e];
        }
        base += __popcll(kb);
        const int used = __popcll(eq) < need ? __popcll(eq) : need;
        need -= used;
    }
    lds_fence();
    // softmax over the 50 (exp(v - max) / sum), wavefront reductions
    const float v = lane < TOPK ? s_w[wave][lane] : -__builtin_inff();
    float mx = v;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float ex = lane < TOPK ? expf(v - mx) : 0.f;
    float sum = ex;
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float wgt = ex / sum;
    lds_fence();
    if (lane < TOPK) {
        s_w[wave][lane] = wgt;
        if (topk_idx) topk_idx[(long)q * TOPK + lane] = s_x[wave][lane];
        if (topk_w) topk_w[(long)q * TOPK + lane] = wgt;
    }
    lds_fence();
    // gather: lane covers channels [4*lane, +4) and [256 + 4*lane, +4) of each 512-float value row
    for (int o = 0; o < k; ++o) {
        const float *mvo = mv + (long)o * mv_os;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
        for (int j = 0; j < TOPK; ++j) {
            const float wj = s_w[wave][j];
            const float *row = mvo + (long)s_x[wave][j] * 512 + 4 * lane;
            const f32x4 r0 = *reinterpret_cast<const f32x4 *>(row);
            const f32x4 r1 = *reinterpret_cast<const f32x4 *>(row + 256);
            a0 += r0 * wj;
            a1 += r1 * wj;
        }
        float *dst = readout + (long)o * ro_os + (long)q * 512 + 4 * lane;
        *reinterpret_cast<f32x4 *>(dst) = a0;
        *reinterpret_cast<f32x4 *>(dst + 256) = a1;
    }
}

int memread_num_chunks(int N) {
    const int steps = (N + HROWS - 1) / HROWS;
    return steps < MAXCHUNK ? steps : MAXCHUNK;
}

void memory_read_launch(const float *mk, const float *msq, const float *qk, int N, int Q, const float *mv,
                        long mv_os, int k, float *readout, long ro_os, int32_t *topk_idx, float *topk_w,
                        MemReadScratch scr, hipStream_t s) {
    constexpr int WAVES = 4;
    const int steps = (N + HROWS - 1) / HROWS;
    const int qblocks = (Q + 16 * WAVES - 1) / (16 * WAVES);
    // chunks: >= 4 so that G = 16*NC >= 64 group maxima exist (tight, valid threshold), and enough
    // workgroups to cover the 256 CUs a few times over
    int NC = (768 + qblocks - 1) / qblocks;
    if (NC < 4) NC = 4;
    if (NC > MAXCHUNK) NC = MAXCHUNK;
    if (NC > steps) NC = steps;
    const int spc = (steps + NC - 1) / NC;
    const int NCeff = (steps + spc - 1) / spc;
    const size_t lds2 = (size_t)WAVES * (2 * 16 * CAP + 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&affinity_pass_kernel<WAVES, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        attr_set = true;
    }
    const dim3 grid(qblocks, NCeff);
    float *gmax = scr.gmax, *tau = scr.tau;
    hipLaunchKernelGGL((affinity_pass_kernel<WAVES, false>), grid, dim3(64 * WAVES), 0, s, mk, msq, qk, N, Q, spc, gmax,
                       (const float *)nullptr, (float *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(threshold_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, gmax, NCeff * NGRP, Q, tau);
    hipLaunchKernelGGL((affinity_pass_kernel<WAVES, true>), grid, dim3(64 * WAVES), lds2, s, mk, msq, qk, N, Q, spc,
                       (float *)nullptr, tau, scr.cand_v, scr.cand_i);
    hipLaunchKernelGGL(merge_readout_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, scr.cand_v, scr.cand_i, NCeff, Q,
                       mv, mv_os, k, readout, ro_os, topk_idx, topk_w);
}

void merge_only_launch(const float *cand_v, const int32_t *cand_i, int NC, int Q, const float *mv, long mv_os, int k,
                       float *readout, long ro_os, hipStream_t s) {
    hipLaunchKernelGGL(merge_readout_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, cand_v, cand_i, NC, Q, mv, mv_os, k,
                       readout, ro_os, (int32_t *)nullptr, (float *)nullptr);
}

// one wave per query, no LDS: 50 rows x 2 KB gathered with 16-byte loads (same pattern as merge_readout's gather)
__global__ __launch_bounds__(256) void gather_sum_kernel(const float *__restrict__ table, const int32_t *__restrict__ idx,
                                                         const float *__restrict__ w, int Q, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
    for (int j = 0; j < TOPK; ++j) {
        const float wj = w[(long)q * TOPK + j];
        const float *row = table + (long)idx[(long)q * TOPK + j] * 512 + 4 * lane;
        a0 += *reinterpret_cast<const f32x4 *>(row) * wj;
        a1 += *reinterpret_cast<const f32x4 *>(row + 256) * wj;
    }
    float *dst = out + (long)q * 512 + 4 * lane;
    *reinterpret_cast<f32x4 *>(dst) = a0;
    *reinterpret_cast<f32x4 *>(dst + 256) = a1;
}
// same victim with scalar FMAs only (inline asm v_fmac_f32: the compiler cannot form v_pk_fma_f32)
__global__ __launch_bounds__(256) void gather_sum_scalar_kernel(const float *__restrict__ table,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ w, int Q,
                                                                float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
    for (int j = 0; j < TOPK; ++j) {
        const float wj = w[(long)q * TOPK + j];
        const float *row = table + (long)idx[(long)q * TOPK + j] * 512 + 4 * lane;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(row), r1 = *reinterpret_cast<const f32x4 *>(row + 256);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c]) : "v"(r0[c]), "v"(wj));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[4 + c]) : "v"(r1[c]), "v"(wj));
        }
    }
    float *dst = out + (long)q * 512 + 4 * lane;
    *reinterpret_cast<f32x4 *>(dst) = f32x4{a[0], a[1], a[2], a[3]};
    *reinterpret_cast<f32x4 *>(dst + 256) = f32x4{a[4], a[5], a[6], a[7]};
}
void gather_sum_launch(const float *table, const int32_t *idx, const float *w, int Q, float *out, hipStream_t s) {
    hipLaunchKernelGGL(gather_sum_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, table, idx, w, Q, out);
}
void gather_sum_scalar_launch(const float *table, const int32_t *idx, const float *w, int Q, float *out, hipStream_t s) {
    hipLaunchKernelGGL(gather_sum_scalar_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, table, idx, w, Q, out);
}

// ------------------------------------------------------------------------------------------------
// Fusion attention read (prop_net.py:117-138,198-211): W = softmax over memory rows of the T=1
// affinity; amap[kk][ch][q] = sum_m pooled[kk][ch][m] W[m][q]; then bilinear x16.
__global__ void area_pool16_kernel(const float *__restrict__ pos, const float *__restrict__ neg, int kk, int h,
                                   int w, float *__restrict__ pooled) {
    // pooled [kk][2][h*w]; one thread per output cell, 16x16 block mean (F.interpolate mode='area')
    const long i = blockIdx.x * 256L + threadIdx.x;
    const int hw = h * w;
    if (i >= (long)kk * 2 * hw) return;
    const int cell = (int)(i % hw);
    const int ch = (int)((i / hw) % 2);
    const int r = (int)(i / (2L * hw));
    const int cy = cell / w, cx = cell - cy * w;
    const int W = 16 * w;
    const float *src = (ch == 0 ? pos : neg) + (long)r * 256 * hw + (long)cy * 16 * W + cx * 16;
    float acc = 0.f;
    for (int y = 0; y < 16; ++y) {
        const f32x4 *p = reinterpret_cast<const f32x4 *>(src + (long)y * W);
        const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
    }
    pooled[i] = acc * (1.f / 256.f);
}

#define STCN_ATT_MAXCH 18   // (k+1)*2 with k <= 8
// cmax[q] = max over the G group maxima of pass 1 = exact column maximum of S
__global__ void colmax_kernel(const float *__restrict__ gmax, int G, int Q, float *__restrict__ cmax) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    float m = -__builtin_inff();
    for (int g = 0; g < G; ++g) m = fmaxf(m, gmax[(long)g * Q + q]);
    cmax[q] = m;
}

// pass 2 of the attention read: per (query block, row chunk) partial sums of e = exp(S - cmax[q]):
//   part[chunk][q][0] = sum_m e,  part[chunk][q][1 + c] = sum_m e * pooled[c][m]
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void attention_pass_kernel(
    const float *__restrict__ mk, const float *__restrict__ msq, const float *__restrict__ qk, int N, int Q,
    int steps_per_chunk, const float *__restrict__ cmax, const float *__restrict__ pooled, int nch,
    float *__restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q0 = (blockIdx.x * WAVES + wave) * 16;
    if (q0 >= Q) return;
    const int chunk = blockIdx.y;
    const int nsteps = (N + HROWS - 1) / HROWS;
    const int h0 = chunk * steps_per_chunk;
    const int h1 = min(nsteps, h0 + steps_per_chunk);
    const int g = lane >> 4, col = lane & 15;
    const int qcol = min(q0 + col, Q - 1);
    f32x4 bq[4];
    load_bq(qk, Q, q0, lane, bq);
    const float cm = cmax[qcol];
    float l = 0.f, a[STCN_ATT_MAXCH];
#pragma unroll
    for (int c = 0; c < STCN_ATT_MAXCH; ++c) a[c] = 0.f;
    for (int h = h0; h < h1; ++h) {
        const int row0 = h * HROWS;
        HalfFrag hf;
        load_half(mk, msq, N, row0, lane, hf);
        f32x4 acc[4];
        mfma_half(hf, bq, acc);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int r = row0 + rb * 16 + 4 * g;
            f32x4 e;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = r + j < N ? expf(acc[rb][j] - cm) : 0.f;
                l += e[j];
            }
#pragma unroll
            for (int c = 0; c < STCN_ATT_MAXCH; ++c)
                if (c < nch) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = r + j < N ? r + j : N - 1;
                        a[c] += e[j] * pooled[(long)c * N + rr];
                    }
                }
        }
    }
    // sum the 4 lane groups that share a query column (fixed order -> deterministic)
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
#pragma unroll
    for (int c = 0; c < STCN_ATT_MAXCH; ++c)
        if (c < nch) { a[c] += __shfl_xor(a[c], 16); a[c] += __shfl_xor(a[c], 32); }
    if (g == 0 && q0 + col < Q) {
        float *dst = part + ((long)chunk * Q + q0 + col) * (1 + STCN_ATT_MAXCH);
        dst[0] = l;
#pragma unroll
        for (int c = 0; c < STCN_ATT_MAXCH; ++c)
            if (c < nch) dst[1 + c] = a[c];
    }
}

__global__ void attention_finalize_kernel(const float *__restrict__ part, int NC, int Q, int nch,
                                          float *__restrict__ amap) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= Q) return;
    float l = 0.f;
    for (int ch = 0; ch < NC; ++ch) l += part[((long)ch * Q + q) * (1 + STCN_ATT_MAXCH)];
    for (int c = 0; c < nch; ++c) {
        float acc = 0.f;
        for (int ch = 0; ch < NC; ++ch) acc += part[((long)ch * Q + q) * (1 + STCN_ATT_MAXCH) + 1 + c];
        amap[(long)c * Q + q] = acc / l;
    }
}

__global__ void bilinear_up16_kernel(const float *__restrict__ amap, int nch, int h, int w, float *__restrict__ out) {
    const int H = 16 * h, W = 16 * w;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)nch * H * W) return;
    const int c = (int)(i / ((long)H * W));
    const long r = i - (long)c * H * W;
    const int oy = (int)(r / W), ox = (int)(r - (long)oy * W);
    float sy = ((float)oy + 0.5f) * 0.0625f - 0.5f, sx = ((float)ox + 0.5f) * 0.0625f - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > h - 1 ? h - 1 : y0;
    x0 = x0 > w - 1 ? w - 1 : x0;
    const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float *a = amap + (long)c * h * w;
    out[i] = (1.f - fy) * ((1.f - fx) * a[y0 * w + x0] + fx * a[y0 * w + x1]) +
             fy * ((1.f - fx) * a[y1 * w + x0] + fx * a[y1 * w + x1]);
}

void attention_read_launch(const float *mk, const float *msq, const float *qk, const float *pos, const float *neg,
                           int kk, int h, int w, float *pooled, float *amap, float *attn, AttnScratch scr,
                           hipStream_t s) {
    const int hw = h * w, nch = kk * 2;
    hipLaunchKernelGGL(area_pool16_kernel, dim3((unsigned)(((long)nch * hw + 255) / 256)), dim3(256), 0, s, pos, neg,
                       kk, h, w, pooled);
    constexpr int WAVES = 4;
    const int steps = (hw + HROWS - 1) / HROWS;
    const int qblocks = (hw + 16 * WAVES - 1) / (16 * WAVES);
    int NC = (512 + qblocks - 1) / qblocks;
    if (NC > MAXCHUNK) NC = MAXCHUNK;
    if (NC > steps) NC = steps;
    const int spc = (steps + NC - 1) / NC;
    const int NCeff = (steps + spc - 1) / spc;
    const dim3 grid(qblocks, NCeff);
    hipLaunchKernelGGL((affinity_pass_kernel<WAVES, false>), grid, dim3(64 * WAVES), 0, s, mk, msq, qk, hw, hw, spc,
                       scr.gmax, (const float *)nullptr, (float *)nullptr, (int32_t *)nullptr);
    hipLaunchKernelGGL(colmax_kernel, dim3((hw + 255) / 256), dim3(256), 0, s, scr.gmax, NCeff * NGRP, hw, scr.cmax);
    hipLaunchKernelGGL((attention_pass_kernel<WAVES>), grid, dim3(64 * WAVES), 0, s, mk, msq, qk, hw, hw, spc, scr.cmax,
                       pooled, nch, scr.part);
    hipLaunchKernelGGL(attention_finalize_kernel, dim3((hw + 255) / 256), dim3(256), 0, s, scr.part, NCeff, hw, nch,
                       amap);
    const long tot = (long)nch * 256 * hw;
    hipLaunchKernelGGL(bilinear_up16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, amap, nch, h, w,
                       attn);
}

}  // namespace stcn
