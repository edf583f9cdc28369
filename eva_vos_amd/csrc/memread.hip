// memread.hip - space-time memory read of STCN on gfx950:
//   affinity S = (2 mk.qk - |mk|^2 - |qk|^2)/sqrt(64)          (reference prop_net.py:80-90)
//   per query column: top-50 over the T*H*W memory rows, softmax over the 50 (prop_net.py:53-60)
//   readout = sum_j w_j * mv[idx_j]                             (prop_net.py:108-115)
// The reference materialises the dense [T*HW x HW] affinity in memory and multiplies the dense matrix;
// here S only ever lives in registers / LDS:
//   kernel A (affinity_topk): grid = (query blocks of 64) x (memory chunks).  Each WAVE owns 16 query
//     columns and walks its chunk in 128-row tiles: S tile = 16x16x4 fp32 MFMAs (A = memory keys
//     straight from global/L2 as 16-B fragments, B = the wave's 16 query keys in registers, the
//     -|mk|^2/2 row constant is the initial accumulator), the tile is transposed through a
//     wave-private LDS slab and filtered against the wave-uniform running threshold tau[q] (the current
//     50th best); survivors are appended by ballot/prefix (no atomics).  When a list would overflow, a
//     wave-wide bitwise radix select keeps the best 50 and raises tau.  No workgroup barrier in the
//     loop - the four waves only share the L1 lines of the key tile.
//   kernel B (merge_readout): one wave per query merges the chunk winners (same select), softmaxes
//     the 50 with wavefront reductions and gathers 50 value rows (2 KB each, NHWC bank) per object.
//   The column-constant -|qk|^2 term cancels in exp(v - v_max) and is dropped.
#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TOPK = 50;
static constexpr int CAP = 128;       // per-query candidate list capacity (>= TOPK + 64)
static constexpr int TROWS = 128;     // memory rows per tile
static constexpr int SLD = 132;       // S slab row stride (floats): 16-B aligned, conflict-free b128 writes
static constexpr int MAXCHUNK = 16;

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

// Keep the best TOPK of the n (<= 128) entries of a wave-owned list (values lv, payload li).
// Returns the new threshold (the TOPK-th best value); the list is compacted to exactly TOPK entries.
__device__ __forceinline__ float wave_select128(float *lv, int *li, int n, int lane) {
    lds_fence();
    const bool v0 = lane < n, v1 = lane + 64 < n;
    const float f0 = lv[lane], f1 = lv[lane + 64];
    const int i0 = li[lane], i1 = li[lane + 64];
    const unsigned k0 = v0 ? f2key(f0) : 0u, k1 = v1 ? f2key(f1) : 0u;   // key 0 < every real key
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

// One wave: S[128 rows x 16 queries] = (mk[rows] . qk[cols] - msq[rows]/2) / 4 into its LDS slab
// Sw[q][row] (row stride SLD).  bq[kb] holds the wave's B fragments: B[k = 16kb + 4g + j][col = lane&15].
__device__ __forceinline__ void s_tile_16q(const float *__restrict__ mk, const float *__restrict__ msq, int N,
                                           int row0, const f32x4 (&bq)[4], float *Sw, int lane) {
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x4 acc[4];
        f32x4 a[4][4];
        // issue all fragment loads of 4 row blocks (16 rows each) first
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            int r = row0 + (half * 4 + rb) * 16 + col;      // A: row = lane&15, k = 16kb + 4g + j
            r = r < N ? r : N - 1;
            const float *ap = mk + (long)r * 64 + 4 * g;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) a[rb][kb] = *reinterpret_cast<const f32x4 *>(ap + 16 * kb);
            // C layout 16x16: col = lane&15, row = 4*(lane>>4) + reg
            int rr = row0 + (half * 4 + rb) * 16 + 4 * g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r2 = rr + j < N ? rr + j : N - 1;
                acc[rb][j] = -0.5f * msq[r2];
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][kb].x, bq[kb].x, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][kb].y, bq[kb].y, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][kb].z, bq[kb].z, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb][kb].w, bq[kb].w, acc[rb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 v = acc[rb];
            v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f;
            *reinterpret_cast<f32x4 *>(&Sw[col * SLD + (half * 4 + rb) * 16 + 4 * g]) = v;
        }
    }
}

__device__ __forceinline__ void load_bq(const float *__restrict__ qk, int Q, int q0, int lane, f32x4 (&bq)[4]) {
    int q = q0 + (lane & 15);
    q = q < Q ? q : Q - 1;
    const float *bp = qk + (long)q * 64 + 4 * (lane >> 4);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bq[kb] = *reinterpret_cast<const f32x4 *>(bp + 16 * kb);
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void affinity_topk_kernel(
    const float *__restrict__ mk, const float *__restrict__ msq, const float *__restrict__ qk, int N, int Q,
    int tiles_per_chunk, float *__restrict__ cand_v, int32_t *__restrict__ cand_i) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *Sw = smem + wave * (16 * SLD);                               // [16][SLD]
    float *LV = smem + WAVES * 16 * SLD + wave * (16 * CAP);            // [16][CAP]
    int *LI = reinterpret_cast<int *>(smem + WAVES * 16 * SLD + WAVES * 16 * CAP) + wave * (16 * CAP);

    const int q0 = (blockIdx.x * WAVES + wave) * 16;
    if (q0 >= Q) return;                                                // wave-uniform; no block barriers below
    const int chunk = blockIdx.y;
    const int tile0 = chunk * tiles_per_chunk;
    const int ntiles_total = (N + TROWS - 1) / TROWS;
    const int tile1 = min(ntiles_total, tile0 + tiles_per_chunk);

    f32x4 bq[4];
    load_bq(qk, Q, q0, lane, bq);

    // per-query list length / threshold live in LDS (wave-private, read back wave-uniformly)
    int *CNT = reinterpret_cast<int *>(smem + WAVES * 16 * (SLD + 2 * CAP)) + wave * 32;
    float *TAU = reinterpret_cast<float *>(CNT + 16);
    if (lane < 16) { CNT[lane] = 0; TAU[lane] = -__builtin_inff(); }

    for (int tile = tile0; tile < tile1; ++tile) {
        const int row0 = tile * TROWS;
        lds_fence();
        s_tile_16q(mk, msq, N, row0, bq, Sw, lane);
        lds_fence();
        for (int j = 0; j < 16; ++j) {
            float *lv = LV + j * CAP;
            int *li = LI + j * CAP;
            int cnt = __builtin_amdgcn_readfirstlane(CNT[j]);
            float tau = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(TAU[j])));
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                const int row = row0 + step * 64 + lane;
                const float v = Sw[j * SLD + step * 64 + lane];
                bool pass = row < N && v > tau;
                unsigned long long bal = __ballot(pass);
                if (bal) {
                    if (cnt + __popcll(bal) > CAP) {
                        tau = wave_select128(lv, li, cnt, lane);
                        cnt = TOPK;
                        pass = pass && v > tau;
                        bal = __ballot(pass);
                    }
                    if (pass) {
                        const int p = cnt + lanes_below(bal, lane);
                        lv[p] = v;
                        li[p] = row;
                    }
                    cnt += __popcll(bal);
                }
            }
            if (lane == 0) { CNT[j] = cnt; TAU[j] = tau; }
        }
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
    const int tiles = (N + TROWS - 1) / TROWS;
    return tiles < MAXCHUNK ? tiles : MAXCHUNK;
}

void memory_read_launch(const float *mk, const float *msq, const float *qk, int N, int Q, const float *mv,
                        long mv_os, int k, float *readout, long ro_os, int32_t *topk_idx, float *topk_w,
                        MemReadScratch scr, hipStream_t s) {
    const int tiles = (N + TROWS - 1) / TROWS;
    const int NC = memread_num_chunks(N);
    const int tpc = (tiles + NC - 1) / NC;
    const int NCeff = (tiles + tpc - 1) / tpc;
    constexpr int WAVES = 4;
    const size_t lds = (size_t)WAVES * (16 * (SLD + 2 * CAP) + 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&affinity_topk_kernel<WAVES>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const dim3 grid((Q + 16 * WAVES - 1) / (16 * WAVES), NCeff);
    hipLaunchKernelGGL((affinity_topk_kernel<WAVES>), grid, dim3(64 * WAVES), lds, s, mk, msq, qk, N, Q, tpc,
                       scr.cand_v, scr.cand_i);
    hipLaunchKernelGGL(merge_readout_kernel, dim3((Q + 3) / 4), dim3(256), 0, s, scr.cand_v, scr.cand_i, NCeff, Q,
                       mv, mv_os, k, readout, ro_os, topk_idx, topk_w);
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
__global__ __launch_bounds__(64) void attention_softmax_kernel(const float *__restrict__ mk,
                                                               const float *__restrict__ msq,
                                                               const float *__restrict__ qk, int HW,
                                                               const float *__restrict__ pooled, int nch,
                                                               float *__restrict__ amap) {
    // one wave per 16 query columns; lanes stride over the memory rows of each S tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Sw = smem;
    const int lane = threadIdx.x;
    const int q0 = blockIdx.x * 16;
    f32x4 bq[4];
    load_bq(qk, HW, q0, lane, bq);
    const int ntiles = (HW + TROWS - 1) / TROWS;
    // state in LDS: m[16][64], l[16][64], acc[16][nch][64]  (lane-private columns)
    float *Sm = smem + 16 * SLD;
    float *Sl = Sm + 16 * 64;
    float *Sa = Sl + 16 * 64;
    for (int j = 0; j < 16; ++j) {
        Sm[j * 64 + lane] = -__builtin_inff();
        Sl[j * 64 + lane] = 0.f;
        for (int c = 0; c < nch; ++c) Sa[(j * nch + c) * 64 + lane] = 0.f;
    }
    for (int tile = 0; tile < ntiles; ++tile) {
        const int row0 = tile * TROWS;
        lds_fence();
        s_tile_16q(mk, msq, HW, row0, bq, Sw, lane);
        lds_fence();
        for (int step = 0; step < 2; ++step) {
            const int row = row0 + step * 64 + lane;
            if (row < HW) {
                float pv[STCN_ATT_MAXCH];
#pragma unroll
                for (int c = 0; c < STCN_ATT_MAXCH; ++c) pv[c] = c < nch ? pooled[(long)c * HW + row] : 0.f;
                for (int j = 0; j < 16; ++j) {
                    const float sv = Sw[j * SLD + step * 64 + lane];
                    const float mo = Sm[j * 64 + lane];
                    const float mn = fmaxf(mo, sv);
                    const float sc = expf(mo - mn);          // exp(-inf) = 0 on the first visit
                    const float e = expf(sv - mn);
                    Sm[j * 64 + lane] = mn;
                    Sl[j * 64 + lane] = Sl[j * 64 + lane] * sc + e;
#pragma unroll
                    for (int c = 0; c < STCN_ATT_MAXCH; ++c)
                        if (c < nch) {
                            float *ap = &Sa[(j * nch + c) * 64 + lane];
                            *ap = *ap * sc + e * pv[c];
                        }
                }
            }
        }
    }
    lds_fence();
    for (int j = 0; j < 16; ++j) {
        const float ml = Sm[j * 64 + lane];
        float M = ml;
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
        const float sc = expf(ml - M);                       // lanes that saw no row: exp(-inf) = 0
        float l = Sl[j * 64 + lane] * sc;
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        for (int c = 0; c < nch; ++c) {
            float a = Sa[(j * nch + c) * 64 + lane] * sc;
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0 && q0 + j < HW) amap[(long)c * HW + q0 + j] = a / l;
        }
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
                           int kk, int h, int w, float *pooled, float *amap, float *attn, hipStream_t s) {
    const int hw = h * w, nch = kk * 2;
    hipLaunchKernelGGL(area_pool16_kernel, dim3((unsigned)(((long)nch * hw + 255) / 256)), dim3(256), 0, s, pos, neg,
                       kk, h, w, pooled);
    const size_t lds = (size_t)(16 * SLD + 2 * 16 * 64 + 16 * nch * 64) * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&attention_softmax_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attention_softmax_kernel, dim3((hw + 15) / 16), dim3(64), lds, s, mk, msq, qk, hw, pooled, nch,
                       amap);
    const long tot = (long)nch * 256 * hw;
    hipLaunchKernelGGL(bilinear_up16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, amap, nch, h, w,
                       attn);
}

}  // namespace stcn
