// winograd4.hip - fp32 Winograd F(4x4, 3x3) for the stride-1 3x3 convolutions (64+ input channels) on gfx950.
//
// winograd.hip's F(2x2,3x3) spends 16 multiplies per 2x2 outputs (2.25x fewer than the direct conv); F(4x4,3x3) spends 36 per
// 4x4 outputs: 4x fewer than direct, 1.78x fewer than F(2x2), and its transformed input V is 2.25x the conv input instead of 4x.
// The price is conditioning: the transforms of F(4x4) amplify fp32 rounding (measured on 256 -> 256 layers, against fp64: direct
// 2e-7, F(2x2) 6e-7, F(4x4) with the textbook points {0, +-1, +-2} 9e-6 of the output range).  Two decisions follow:
//   * interpolation points {0, +-3/4, +-3/2} (every transform constant is still an exact binary fraction): 2.8e-6, 3.4x better
//     than the textbook set - inside the 2e-5 the conv tests hold every kernel to;
//   * rounds 3-4 ran it on the decoder side only (outputs become probabilities and memory VALUES) and kept the key encoder, whose
//     output decides top-50 MEMBERSHIP in the memory read, on F(2x2) / direct kernels on principle; round 5 measured it: with the
//     trunk's 3x3 convs on F(4x4) no more pixels differ from the CPU oracle (profiles/r05_key_trunk_f4.txt), so they run it too;
//     key_proj itself stays on F(2x2).  Reference layers: mivos/model/propagation/modules.py:15-35,127-163, prop_net.py:13-30.
//
//     V = B^T d B   (6x6 input tile d, per channel)        U = G g G^T   (3x3 filter g, host, once per model)
//     M[xi][nu] = sum_cin U[xi][nu] V[xi][nu]               Y = A^T M A   (4x4 outputs)
//
// Structure = winograd.hip's (transform kernel, then an LDS-free batched GEMM with the output transform in its epilogue), with
// what 36 positions change: the accumulators of ALL positions of a workgroup tile have to meet for the output transform, and
// 36 x (64 x 64) fp32 would be 2304 registers per lane of a CU that has 2048.  So a workgroup owns 32 tiles x 32 output channels:
// 12 waves x 3 positions x ONE 32x32 accumulator block; per k-block of 8 channels a wave loads 3 A and 3 B fragments (1 KB each,
// contiguous: V / U are laid out [position][C/8][row][8]) for 12 MFMAs - twice the L2 bytes per MFMA of the F(2x2) kernel, the
// unavoidable cost of the small tile - two k-blocks ahead, pinned with sched_barrier.  Epilogue: 36 x 32 x 32 floats = 144 KB
// of LDS; a thread gathers the 36 positions of (tile, 4 channels) with 16-byte reads and produces two of the four output
// columns: A^T M A, bias / residual / ReLU, 16-byte buffer stores.
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// DIAGNOSTIC build only (make EXTRA=-DSTCN_W4_CLOCK=1|2, tools/w4_clock.sh): wave 0 of every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) around the main loop; quotient x 100 MHz = the clock the chip held INSIDE the loop (MI355X_MICROARCH.md, DVFS
// give-back (6)).  =2 additionally replaces every MFMA by 16 v_fma_f32 on its accumulator registers: the same 64 issue cycles, the same
// loads, no matrix arithmetic - separates "clock held down by the MFMAs' power" from "loop bound by operand feed".  The shipped library
// executes no stamp.
#ifdef STCN_W4_CLOCK
__device__ unsigned long long g_w4_clock[2 * 8192];
#if STCN_W4_CLOCK == 2
#define W4_MFMA(a_, b_, c_) w4_fake_mfma(a_, b_, c_)
#endif
#endif
#ifndef W4_MFMA
#define W4_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, c_, 0, 0, 0)
#endif

static constexpr int W4T = 32;       // tiles per workgroup
static constexpr int W4N = 32;       // output channels per workgroup
static constexpr int W4W = 12;       // waves per workgroup (3 positions each)

#if defined(STCN_W4_CLOCK) && STCN_W4_CLOCK == 2
__device__ __forceinline__ f32x16 w4_fake_mfma(float a, float b, f32x16 c) {
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[e]) : "v"(a), "v"(b));
    return c;
}
#endif

// 1-D input transform with the points {0, 3/4, -3/4, 3/2, -3/2, inf}: o = B^T d
__device__ __forceinline__ void bt6(const f32x4 &d0, const f32x4 &d1, const f32x4 &d2, const f32x4 &d3, const f32x4 &d4, const f32x4 &d5,
                                    f32x4 &o0, f32x4 &o1, f32x4 &o2, f32x4 &o3, f32x4 &o4, f32x4 &o5) {
    o0 = d0 * 1.265625f - d2 * 2.8125f + d4;                       // 81/64, 45/16
    const f32x4 e1 = d4 - d2 * 2.25f, f1 = d3 * 0.75f - d1 * 1.6875f;       // 9/4; 3/4, 27/16
    o1 = e1 + f1;
    o2 = e1 - f1;
    const f32x4 e2 = d4 - d2 * 0.5625f, f2 = d3 * 1.5f - d1 * 0.84375f;     // 9/16; 3/2, 27/32
    o3 = e2 + f2;
    o4 = e2 - f2;
    o5 = d1 * 1.265625f - d3 * 2.8125f + d5;
}

// ------------------------------------------------------------------------------------------------ input transform
// thread = (tile, 16-byte channel group c16 of a 32-channel block), as wino_input_kernel: 8 consecutive lanes read one whole
// 128-byte line of a pixel; stores are 256-byte runs of V per position.
__global__ __launch_bounds__(256) void wino4_input_kernel(const float *__restrict__ x, unsigned x_bytes, long x_bs, int H, int W, int C,
                                                          int relu_in, int TH, int TW, int Mt, int Mt_pad, int cb_per_chunk,
                                                          int tile_lo, int tile_hi, float *__restrict__ V) {
    // XCD-contiguous block order: neighbouring tile rows share two pixel rows; dealt round-robin over the XCDs (the dispatcher's
    // order) those rows were fetched into two L2s - FETCH_SIZE 2.1x the input
    const int nbx = gridDim.x, q8 = nbx >> 3, r8 = nbx & 7, xcd = blockIdx.x & 7;
    const int bx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const long i = bx * 256L + threadIdx.x;
    const int c16 = (int)(i & 7);
    const long tile = tile_lo + (i >> 3);                    // [tile_lo, tile_hi): the launch's slice of the tiles (whole V when not chunked)
    if (tile >= tile_hi) return;
    const int KB = C / 8, NCB = C / 32;
    const int cb0 = blockIdx.y * cb_per_chunk, cb1 = min(NCB, cb0 + cb_per_chunk);
    const bool live = tile < Mt;
    const int tpi = TH * TW;
    const int b = live ? (int)(tile / tpi) : 0;
    const int r = live ? (int)(tile - (long)b * tpi) : 0;
    const int ty = r / TW, tx = r - ty * TW;
    const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
    unsigned long long ok = 0;                               // bit (yy * 6 + xx): the pixel lies inside the image
#pragma unroll
    for (int yy = 0; yy < 6; ++yy)
#pragma unroll
        for (int xx = 0; xx < 6; ++xx)
            if (live && (unsigned)(y0 + yy) < (unsigned)H && (unsigned)(x0 + xx) < (unsigned)W) ok |= 1ull << (yy * 6 + xx);
    // buffer loads: a pixel outside the image reads through an out-of-range offset and comes back as zero (the conv's padding) -
    // no branch per pixel (36 of them cost 64 spilled SGPRs in the first version)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, x_bytes, 0x00020000);
    const int xoff = (int)((b * x_bs + ((long)y0 * W + x0) * C + 4 * c16) * 4);      // may be negative at the border: only used when valid
    const long ps = ((long)KB * Mt_pad) << 3;                // floats between positions
    // a branch on relu_in around each load makes hipcc wait vmcnt(0) per load: 36 serialized round trips (measured: 125 us
    // for a 79 us job).  max(v, lo) with lo = 0 or -inf is the same arithmetic without the branch.
    const float lo = relu_in ? 0.f : -__builtin_inff();
    for (int cb = cb0; cb < cb1; ++cb) {
        f32x4 t[6][6];                                       // t = B^T d (columns transformed), row by row of the input
#pragma unroll
        for (int xx = 0; xx < 6; ++xx) {
            f32x4 d[6];
#pragma unroll
            for (int yy = 0; yy < 6; ++yy) {
                const unsigned vo = ((ok >> (yy * 6 + xx)) & 1ull) ? (unsigned)(xoff + ((yy * W + xx) * C + 32 * cb) * 4) : 0x80000000u;
                f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0));
                v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);      // ReLU or identity, branchless
                d[yy] = v;
            }
            bt6(d[0], d[1], d[2], d[3], d[4], d[5], t[0][xx], t[1][xx], t[2][xx], t[3][xx], t[4][xx], t[5][xx]);
        }
        const int kb = cb * 4 + (c16 >> 1);
        float *base = V + (((long)kb * Mt_pad + tile) << 3) + 4 * (c16 & 1);
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {                     // (B^T d) B: position = 6 xi + nu
            f32x4 v[6];
            bt6(t[xi][0], t[xi][1], t[xi][2], t[xi][3], t[xi][4], t[xi][5], v[0], v[1], v[2], v[3], v[4], v[5]);
            float *dst = base + (long)(xi * 6) * ps;
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) *reinterpret_cast<f32x4 *>(dst + nu * ps) = v[nu];
        }
    }
}

// ------------------------------------------------------------------------------------------------ batched GEMM + output transform
struct Wino4G {
    const float *V, *U;
    unsigned v_bytes, u_bytes;
    int Mt, Mt_pad, KB, N;
    int TH, TW, OH, OW, B, M;
    const float *bias, *res;
    long res_bs; int res_bmod;
    float *y; long y_bs;
    int relu_out;
    FastDiv fd_tpi, fd_tw, fd_tiles_n;
    // tail split (MB = 1 only): workgroups [0, full_wg) compute whole tiles; the remaining tiles are cut into `pieces` ranges of
    // kb_per_piece k-blocks whose output-domain partial sums (the output transform is linear) go to `partial`
    // [(tile - full_wg) * pieces + piece][32 tiles][16 pixels][32 channels] and are summed by wino4_reduce_kernel
    int full_wg, pieces, kb_per_piece;
    float *partial;
    int tm0;                                   // first tile block of this launch (chunked launches, MB = 2)
    // 2-D blocks per XCD (whole-tile workgroups): an XCD's contiguous range of xb_m x xb_n workgroups covers xb_m tile blocks x xb_n
    // channel blocks instead of a strip of rows - xb_m + xb_n operand streams through its L2 instead of rows + tiles_n (0: strips)
    int xb_m, xb_n, xb_cols;
};

// MB = 32-tile blocks per workgroup.  MB = 1: three fragment sets, loads two k-blocks ahead.  MB = 2 (64 tiles x 32 channels: 25 %
// fewer L2 bytes per MFMA, half the U traffic): 96 accumulator registers leave room for two HALF fragment sets - the loop walks
// half-steps (k-block, 32-tile block) and loads one half-step ahead (12 MFMAs x 3 waves sharing the SIMD: ~2300 cycles).
template <int MB>
__global__ __launch_bounds__(64 * W4W) void wino4_gemm_kernel(const Wino4G p, const int tiles_n) {
    constexpr int PPW = 3, WT = W4T * MB;
    extern __shared__ __attribute__((aligned(16))) float smem[];          // epilogue: [36][32 tiles][32 channels]
    auto xcd_contiguous = [](int bid, int nb) {
        const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    };
    int swz, piece = -1, kb0 = 0, kb1 = p.KB;
    if (MB == 1 && p.pieces > 1 && (int)blockIdx.x >= p.full_wg) {           // a K piece of one of the last tiles
        const int j = xcd_contiguous((int)blockIdx.x - p.full_wg, (int)gridDim.x - p.full_wg);
        const int rt = j / p.pieces;
        piece = j - rt * p.pieces;
        swz = p.full_wg + rt;
        kb0 = piece * p.kb_per_piece;
        kb1 = min(p.KB, kb0 + p.kb_per_piece);
    } else {
        swz = xcd_contiguous(blockIdx.x, MB == 1 && p.pieces > 1 ? p.full_wg : (int)gridDim.x);
    }
    int tm_l = fastdiv(swz, p.fd_tiles_n), tn = swz - tm_l * tiles_n;
    if (p.xb_m > 0 && piece < 0) {                                            // swz = block * (xb_m xb_n) + row-major position inside the block
        const int per = p.xb_m * p.xb_n, blk = swz / per, in = swz - blk * per;
        const int bm_i = blk / p.xb_cols, bn_i = blk - bm_i * p.xb_cols, im = in / p.xb_n;
        tm_l = bm_i * p.xb_m + im;
        tn = bn_i * p.xb_n + (in - im * p.xb_n);
    }
    const int tm = tm_l + p.tm0;
    const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);                  // wave-uniform: the position offsets below stay in SGPRs
    const int pos0 = PPW * wave;

    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.V), 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.U), 0, p.u_bytes, 0x00020000);
    // one lane offset; position / block / k-block offsets are wave-uniform and ride in the scalar offset of the buffer loads
    const unsigned vl = (unsigned)(l31 * 32 + h * 16);
    unsigned sa0[PPW], sb0[PPW];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi) {
        sa0[pi] = (unsigned)((((long)(pos0 + pi) * p.KB + kb0) * p.Mt_pad + tm * WT) * 32);
        sb0[pi] = (unsigned)((((long)(pos0 + pi) * p.KB + kb0) * p.N + tn * W4N) * 32);
    }
    const unsigned sa = (unsigned)p.Mt_pad * 32u, sb = (unsigned)p.N * 32u;   // bytes per k-block

    f32x16 acc[PPW][MB];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
        for (int bi = 0; bi < MB; ++bi)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[pi][bi][e] = 0.f;

    const int nk = kb1 - kb0;
    auto load1 = [&](int k, int pi, f32x4 (&fa)[MB], f32x4 &fb) {
        const unsigned oa = sa0[pi] + (unsigned)k * sa, ob = sb0[pi] + (unsigned)k * sb;
#pragma unroll
        for (int bi = 0; bi < MB; ++bi) fa[bi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vl, oa + bi * 1024u, 0));
        fb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, vl, ob, 0));
    };
    auto compute1 = [&](int pi, const f32x4 (&fa)[MB], const f32x4 &fb) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int bi = 0; bi < MB; ++bi) acc[pi][bi] = W4_MFMA(fa[bi][j], fb[j], acc[pi][bi]);
    };
#ifdef STCN_W4_CLOCK
    unsigned long long ck_t0 = 0, ck_r0 = 0;
    if (wave == 0) {
        __builtin_amdgcn_sched_barrier(0);
        ck_t0 = __builtin_amdgcn_s_memtime();
        ck_r0 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if (MB == 1) {
        // three fragment sets: the loads of k-block k+2 are issued before the MFMAs of k-block k (as wino_gemm_kernel)
        f32x4 fa[3][PPW][MB], fb[3][PPW];
        auto load = [&](int k, int set) {
#pragma unroll
            for (int pi = 0; pi < PPW; ++pi) load1(k, pi, fa[set][pi], fb[set][pi]);
        };
        auto compute = [&](int set) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
                    for (int bi = 0; bi < MB; ++bi)
                        acc[pi][bi] = W4_MFMA(fa[set][pi][bi][j], fb[set][pi][j], acc[pi][bi]);
        };
        load(0, 0);
        load(min(1, nk - 1), 1);
        int k = 0;
        for (; k + 2 < nk; k += 3) {
            load(k + 2, 2);
            __builtin_amdgcn_sched_barrier(0);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            load(min(k + 3, nk - 1), 0);
            __builtin_amdgcn_sched_barrier(0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
            load(min(k + 4, nk - 1), 1);
            __builtin_amdgcn_sched_barrier(0);
            compute(2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k < nk) compute(0);
        if (k + 1 < nk) compute(1);
    } else {
        // half-steps (k, bi): A fragments of ONE 32-tile block per position (set = bi), B fragments per k-block (set = k & 1);
        // the loads of half-step s + 1 are issued before the 12 MFMAs of half-step s.  Two k-blocks per loop body (nk is a
        // multiple of 4).
        f32x4 fa[2][PPW], fb[2][PPW];
        auto loadA = [&](int k, int bi, f32x4 (&d)[PPW]) {
#pragma unroll
            for (int pi = 0; pi < PPW; ++pi)
                d[pi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, vl, sa0[pi] + (unsigned)k * sa + bi * 1024u, 0));
        };
        auto loadB = [&](int k, f32x4 (&d)[PPW]) {
#pragma unroll
            for (int pi = 0; pi < PPW; ++pi)
                d[pi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, vl, sb0[pi] + (unsigned)k * sb, 0));
        };
        auto comp = [&](int bi, const f32x4 (&a)[PPW], const f32x4 (&bb)[PPW]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pi = 0; pi < PPW; ++pi) acc[pi][bi] = W4_MFMA(a[pi][j], bb[pi][j], acc[pi][bi]);
        };
        loadB(0, fb[0]);
        loadA(0, 0, fa[0]);
        for (int k = 0; k < nk; k += 2) {
            const int k2 = min(k + 2, nk - 2);                              // past the end: re-fetch the last pair (unused)
            loadA(k, 1, fa[1]);
            __builtin_amdgcn_sched_barrier(0);
            comp(0, fa[0], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            loadB(k + 1, fb[1]);
            loadA(k + 1, 0, fa[0]);
            __builtin_amdgcn_sched_barrier(0);
            comp(1, fa[1], fb[0]);
            __builtin_amdgcn_sched_barrier(0);
            loadA(k + 1, 1, fa[1]);
            __builtin_amdgcn_sched_barrier(0);
            comp(0, fa[0], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
            loadB(k2, fb[0]);
            loadA(k2, 0, fa[0]);
            __builtin_amdgcn_sched_barrier(0);
            comp(1, fa[1], fb[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifdef STCN_W4_CLOCK
    if (wave == 0) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ck_t1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0 && blockIdx.x < 8192) { g_w4_clock[2 * blockIdx.x] = ck_t1 - ck_t0; g_w4_clock[2 * blockIdx.x + 1] = ck_r1 - ck_r0; }
    }
#endif
    // ---- epilogue, 32 tiles at a time: all 36 positions of 32 x 32 (tile, channel) pairs meet in LDS, Y = A^T M A.
    // thread = (tile, 4 consecutive channels, half of the output columns): 16-byte LDS reads, residual loads and stores; the
    // first 8 waves work (32 tiles x 8 channel quads x 2 column pairs).  The residual block (+ bias) is requested before the
    // barrier - after the accumulators went to LDS, their registers are free - and is the start value of the output sums.
    const int chq = t & 7, tl = (t >> 3) & 31;
    const int jh = __builtin_amdgcn_readfirstlane(t >> 8);                    // wave-uniform: output columns 2 jh, 2 jh + 1
    const int tpi = p.TH * p.TW, ohw = p.OH * p.OW;
    const int n = tn * W4N + 4 * chq;
    const bool kpiece = MB == 1 && piece >= 0;
    const float lo = p.relu_out ? 0.f : -__builtin_inff();
    // residual / output through buffer resources: 32-bit byte offsets (extents < 4 GiB: wino4_workspace_floats), a masked store is
    // an out-of-range offset
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res : p.y), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, -1, 0x00020000);
    const unsigned y_bs = (unsigned)(p.y_bs ? p.y_bs : (long)ohw * p.N);
#pragma unroll
    for (int half = 0; half < MB; ++half) {
        int toff = 32 * half + tl;
        asm volatile("" : "+v"(toff));                                        // the index math of the second half stays behind the first
        const long gt = (long)tm * WT + toff;
        const bool work = jh < 2 && gt < p.Mt;
        const int gtc = (int)min(gt, (long)p.Mt - 1);
        const int b = fastdiv(gtc, p.fd_tpi);
        const int rr = gtc - b * tpi;
        const int ty = fastdiv(rr, p.fd_tw), tx = rr - ty * p.TW;
        unsigned prow[4], pcol[2];                                            // byte offsets inside the image; ragged tiles: clamped, stores masked
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) prow[i2] = (unsigned)(min(4 * ty + i2, p.OH - 1) * p.OW * p.N) * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) pcol[j] = (unsigned)(min(4 * tx + 2 * jh + j, p.OW - 1) * p.N + n) * 4u;
        if (half) __syncthreads();
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                smem[((pos0 + pi) * W4T + (r & 3) + 8 * (r >> 2) + 4 * h) * W4N + l31] = acc[pi][half][r];
        f32x4 yv[2][4], bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && !kpiece) {
            int nb = n;
            asm volatile("" : "+v"(nb));                                      // per half: four registers not live across the other half
            bv = *reinterpret_cast<const f32x4 *>(p.bias + nb);
        }
        if (p.res && !kpiece && jh < 2) {
            const unsigned rb = (unsigned)(p.res_bmod ? b % p.res_bmod : b) * (unsigned)p.res_bs * 4u;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2)
                    yv[j][i2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, rb + prow[i2] + pcol[j], 0, 0));
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) yv[j][i2] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (!work) continue;
        // rows of M are the vertical index xi: pos = 6 xi + nu.  Per row the transform along nu (6 -> this thread's 2 columns), then
        // along xi (6 -> 4 rows) as running sums (one z row live at a time): A^T = [1 1 1 1 1 0; 0 a -a b -b 0; 0 a^2 a^2 b^2 b^2 0; 0 a^3 -a^3 b^3 -b^3 1],
        // a = 3/4, b = 3/2
        auto zrow = [&](int xi, auto JH, f32x4 (&zr)[2]) {
            f32x4 m[6];
            int ro4 = (xi * 6 * W4T + tl) * (W4N / 4) + chq;                  // in 16-byte units: the alignment survives the pin
            asm volatile("" : "+v"(ro4));                                     // one address register per row, not 20 hoisted ones
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) m[nu] = reinterpret_cast<const f32x4 *>(smem)[ro4 + nu * W4T * (W4N / 4)];
            if (decltype(JH)::value == 0) {
                zr[0] = m[0] + (m[1] + m[2]) + (m[3] + m[4]);
                zr[1] = (m[1] - m[2]) * 0.75f + (m[3] - m[4]) * 1.5f;
            } else {
                zr[0] = (m[1] + m[2]) * 0.5625f + (m[3] + m[4]) * 2.25f;
                zr[1] = (m[1] - m[2]) * 0.421875f + (m[3] - m[4]) * 3.375f + m[5];     // 27/64, 27/8
            }
        };
        auto columns = [&](auto JH) {
            f32x4 za[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) yv[j][i2] += bv;
            zrow(0, JH, za);
#pragma unroll
            for (int j = 0; j < 2; ++j) yv[j][0] += za[j];
            constexpr float c1[4] = {0.75f, -0.75f, 1.5f, -1.5f}, c2[4] = {0.5625f, 0.5625f, 2.25f, 2.25f},
                            c3[4] = {0.421875f, -0.421875f, 3.375f, -3.375f};            // 27/64, 27/8
#pragma unroll
            for (int xi = 1; xi < 5; ++xi) {
                __builtin_amdgcn_sched_barrier(0);                            // keep the LDS reads of later rows behind (registers)
                zrow(xi, JH, za);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    yv[j][0] += za[j];
                    yv[j][1] += za[j] * c1[xi - 1];
                    yv[j][2] += za[j] * c2[xi - 1];
                    yv[j][3] += za[j] * c3[xi - 1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            zrow(5, JH, za);
#pragma unroll
            for (int j = 0; j < 2; ++j) yv[j][3] += za[j];
        };
        if (jh == 0) columns(std::integral_constant<int, 0>{});
        else columns(std::integral_constant<int, 1>{});
        if (kpiece) {
            // K piece: the outputs of (tile, channels) of THIS k range, no bias / residual / ReLU, into the tile-local slab
            float *dst = p.partial + (((long)(swz - p.full_wg) * p.pieces + piece) * W4T + tl) * (16 * W4N) + 4 * chq;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) *reinterpret_cast<f32x4 *>(dst + (i2 * 4 + 2 * jh + j) * W4N) = yv[j][i2];
            continue;
        }
        const unsigned yb = (unsigned)b * y_bs * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
#pragma unroll
                for (int c = 0; c < 4; ++c) yv[j][i2][c] = fmaxf(yv[j][i2][c], lo);
                const bool ok = 4 * tx + 2 * jh + j < p.OW && 4 * ty + i2 < p.OH;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, yv[j][i2]), ry,
                                                       ok ? yb + prow[i2] + pcol[j] : 0xFFFFFFFFu, 0, 0);
            }
    }
}

// tail split: y(tile) = sum over the K pieces of the tile-local partial outputs + bias / residual / ReLU.  One thread per output
// value: 64 workgroups per split tile (32 tiles x 16 pixels x 32 channels), the up to 8 partial values requested together.
__global__ __launch_bounds__(256) void wino4_reduce_kernel(const Wino4G p, const int tiles_n) {
    const int rt = blockIdx.x >> 6, swz = p.full_wg + rt;
    const int tm = fastdiv(swz, p.fd_tiles_n), tn = swz - tm * tiles_n;
    const int n_l = threadIdx.x & 31;
    const int e = ((blockIdx.x & 63) << 3) + (threadIdx.x >> 5);              // tile-local tile * 16 + pixel
    const int tl = e >> 4, px = e & 15;
    const int n = tn * W4N + n_l;
    const long gt = (long)tm * W4T + tl;
    if (gt >= p.Mt) return;
    const int tpi = p.TH * p.TW, ohw = p.OH * p.OW;
    const int b = fastdiv((int)gt, p.fd_tpi);
    const int rr = (int)(gt - (long)b * tpi);
    const int ty = fastdiv(rr, p.fd_tw), tx = rr - ty * p.TW;
    const int oh = 4 * ty + (px >> 2), ow = 4 * tx + (px & 3);
    if (oh >= p.OH || ow >= p.OW) return;
    const float *src = p.partial + ((long)rt * p.pieces * W4T * 16 + e) * W4N + n_l;
    float pv[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) pv[s2] = src[(long)min(s2, p.pieces - 1) * (W4T * 16 * W4N)];      // pieces <= 8; extra reads repeat the last
    const long po = ((long)oh * p.OW + ow) * p.N + n;
    const float rv = p.res ? p.res[(long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs + po] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) v += s2 < p.pieces ? pv[s2] : 0.f;
    v += (p.bias ? p.bias[n] : 0.f) + rv;
    p.y[(p.y_bs ? (long)b * p.y_bs : (long)b * ohw * p.N) + po] = fmaxf(v, p.relu_out ? 0.f : -__builtin_inff());
}

// ------------------------------------------------------------------------------------------------ host side
static int wino4_mode() {          // 0 off, 1 on for flagged layers with enough workgroups (default), 2 whenever the shape allows
    static const int m = [] { const char *e = getenv("STCN_WINO4"); return e ? atoi(e) : 1; }();
    return m;
}

static int wino4_cus() {
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    return cus;
}

// Small launches (round 4): a layer whose 32-tile workgroups do not even fill HALF a round of CUs - the value encoder's fuser and
// its frame parts at batch 1 (1620 pixels x 512 channels = 64 workgroups), the 1/16-scale decoder layers of a single frame - used
// to fall back to F(2x2) with split-K (threshold STCN_WINO4_MIN_WG).  The K pieces of the tail split generalise: EVERY tile is cut
// into up to 8 pieces of >= 8 k-blocks so that about one round of CUs is busy, the output-domain partial sums meet in
// wino4_reduce_kernel.  Returns the pieces per tile (1: leave the launch alone).
static int wino4_small_pieces(int grid, int KB) {
    static const bool on = [] { const char *e = getenv("STCN_WINO4_SMALL"); return !e || atoi(e) != 0; }();
    const int cus = wino4_cus();
    if (!on || grid * 2 > cus) return 1;
    constexpr int min_kb = 8;                                    // k-blocks per piece, at least (4 / 8 / 16 measured equal on the solo leg, round 4)
    int sp = cus / grid;
    sp = sp > 8 ? 8 : sp;
    while (sp > 1 && KB / sp < min_kb) --sp;
    const int per = (KB + sp - 1) / sp;
    return (KB + per - 1) / per;
}

// floats of V workspace the F(4x4) path needs for this conv (0: not eligible).  min_wg: fewest workgroups worth launching
size_t wino4_workspace_floats(const ConvP &p, int min_wg) {
    if (!wino4_mode() || !p.wino4_u || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.x1) return 0;
    if (p.Cin % 32 || p.Cin < W4_MIN_CIN || p.N % W4N || p.bs0 == 0) return 0;
    const long Mt = (long)p.B * ((p.OH + 3) / 4) * ((p.OW + 3) / 4);
    const long Mt_pad = (Mt + 2 * W4T - 1) / (2 * W4T) * (2 * W4T);          // whole 64-tile workgroup tiles
    if (36L * p.Cin * Mt_pad * 4 >= (1L << 32)) return 0;                   // 32-bit buffer offsets
    if ((long)p.B * (p.y_bs ? p.y_bs : (long)p.OH * p.OW * p.N) * 4 >= (1L << 32)) return 0;
    if (p.res && (long)(p.res_bmod ? p.res_bmod : p.B) * p.res_bs * 4 >= (1L << 32)) return 0;
    long wgs = (Mt_pad / W4T) * (p.N / W4N);
    if (wgs * 2 <= wino4_cus()) wgs *= wino4_small_pieces((int)wgs, p.Cin / 8);   // a small launch is cut into K pieces (wino4_plan)
    if (wino4_mode() < 2 && wgs < min_wg) return 0;                          // still too few workgroups: F(2x2) with its split-K is better
    return (size_t)36 * p.Cin * Mt_pad;
}

// launch plan of the F(4x4) GEMM: 64- or 32-tile workgroups, and (32-tile only) the tail split
struct W4Plan { int Mt, Mt_pad, tiles_m, tiles_n, mb, grid, full_wg, pieces, per, chunks, tm_per_chunk; };
static W4Plan wino4_plan(const ConvP &p, size_t slab_floats) {
    static const int mb_env = [] { const char *e = getenv("STCN_WINO4_MB"); return e ? atoi(e) : 0; }();
    static const bool tail_on = [] { const char *e = getenv("STCN_WINO4_TAIL"); return !e || atoi(e) != 0; }();
    const int cus = wino4_cus();
    W4Plan pl{};
    const int TH = (p.OH + 3) / 4, TW = (p.OW + 3) / 4, KB = p.Cin / 8;
    pl.Mt = p.B * TH * TW;
    pl.Mt_pad = (pl.Mt + 2 * W4T - 1) / (2 * W4T) * (2 * W4T);
    pl.tiles_n = p.N / W4N;
    // 64-tile workgroups (25 % fewer L2 bytes per MFMA: +6 % at 1/4 and 1/8 scale) unless they would leave CUs idle
    pl.mb = mb_env == 1 || mb_env == 2 ? mb_env : ((pl.Mt_pad / (2 * W4T)) * pl.tiles_n >= 200 ? 2 : 1);
    pl.tiles_m = pl.Mt_pad / (W4T * pl.mb);
    pl.grid = pl.tiles_m * pl.tiles_n;
    pl.full_wg = pl.grid; pl.pieces = 1; pl.per = KB;
    // Tail split (32-tile workgroups): one workgroup per CU is resident (144 KB of LDS), so a grid of 288 workgroups - the
    // 1/16-scale 512-channel layers over a 5-frame group - costs two rounds for 1.125 rounds of work.  The whole rounds run as they
    // are; the tiles of the ragged last round are cut into K pieces that fill the chip once more (32 tiles x 8 pieces of 8 k-blocks:
    // 1.125 instead of 2 tile times) and meet in wino4_reduce_kernel.
    if (tail_on && pl.mb == 1 && p.partial) {
        const int full = pl.grid / cus * cus, rem = pl.grid - full;
        if (full >= cus && rem > 0 && rem <= cus / 2) {
            int sp = cus / rem;
            sp = sp > 8 ? 8 : sp;
            while (sp > 1 && KB / sp < 8) --sp;                               // at least 8 k-blocks per piece
            const int per = (KB + sp - 1) / sp;
            sp = (KB + per - 1) / per;
            if (sp > 1 && (size_t)rem * sp * W4T * 16 * W4N <= slab_floats) {
                pl.full_wg = full; pl.pieces = sp; pl.per = per;
                pl.grid = full + rem * sp;
            }
        }
    }
    // Small launches: every tile in K pieces (see wino4_small_pieces)
    if (tail_on && pl.mb == 1 && p.partial && pl.pieces == 1 && pl.grid * 2 <= cus) {
        const int sp = wino4_small_pieces(pl.grid, KB);
        if (sp > 1 && (size_t)pl.grid * sp * W4T * 16 * W4N <= slab_floats) {
            pl.full_wg = 0; pl.pieces = sp; pl.per = (KB + sp - 1) / sp;
            pl.grid *= sp;
        }
    }
    // Chunked launches (64-tile workgroups): V of the 1/4-scale decoder layers over a 5-frame group is 299 MB - written by the
    // transform, it has left the 256 MB memory-side cache before the GEMM reads it, and the GEMM's loads (one half-step ahead)
    // then see HBM latency: 97-99 us per round of workgroups instead of 89 (measured at 75 / 151 / 302 / 604 MB of V).  Transform
    // and GEMM alternate over slices of whole rounds whose V stays under ~160 MB.
    pl.chunks = 1; pl.tm_per_chunk = pl.tiles_m;
    const long chunk_bytes = (long)p.kn.wino4_chunk_mb << 20;                  // tests run shapes under tiny chunks
    const long vbytes = 36L * p.Cin * pl.Mt_pad * 4;
    if (pl.mb == 2 && chunk_bytes > 0 && vbytes > chunk_bytes * 3 / 2) {
        const int rounds = (pl.grid + cus - 1) / cus;
        const int n = (int)((vbytes + chunk_bytes - 1) / chunk_bytes);
        const int rpc = (rounds + n - 1) / n;
        int tmpc = rpc * cus / pl.tiles_n;
        tmpc = tmpc < 1 ? 1 : tmpc;
        if (tmpc < pl.tiles_m && (pl.tiles_m + tmpc - 1) / tmpc <= 16) { pl.tm_per_chunk = tmpc; pl.chunks = (pl.tiles_m + tmpc - 1) / tmpc; }
    }
    return pl;
}
int wino4_chunks(const ConvP &p, size_t slab_floats) { return wino4_plan(p, slab_floats).chunks; }
bool wino4_tail_split(const ConvP &p, size_t slab_floats) { return wino4_plan(p, slab_floats).pieces > 1; }

void wino4_launch(const ConvP &p, float *V, size_t slab_floats, hipStream_t s, hipEvent_t *const *ev_in, hipEvent_t *const *ev_gemm,
                  hipEvent_t *ev_red) {
    const W4Plan pl = wino4_plan(p, slab_floats);
    const int TH = (p.OH + 3) / 4, TW = (p.OW + 3) / 4;
    const int Mt = pl.Mt, Mt_pad = pl.Mt_pad, KB = p.Cin / 8;
    Wino4G g{};
    g.V = V; g.U = p.wino4_u;
    g.v_bytes = (unsigned)((size_t)36 * p.Cin * Mt_pad * 4);
    g.u_bytes = (unsigned)((size_t)36 * p.Cin * p.N * 4);
    g.Mt = Mt; g.Mt_pad = Mt_pad; g.KB = KB; g.N = p.N;
    g.TH = TH; g.TW = TW; g.OH = p.OH; g.OW = p.OW; g.B = p.B; g.M = p.M;
    g.bias = p.bias; g.res = p.res; g.res_bs = p.res_bs; g.res_bmod = p.res_bmod; g.y = p.y; g.y_bs = p.y_bs; g.relu_out = p.relu_out;
    const int tiles_n = pl.tiles_n, mb = pl.mb, tiles_m = pl.tiles_m;
    g.fd_tpi = fastdiv_make((unsigned)(TH * TW)); g.fd_tw = fastdiv_make((unsigned)TW); g.fd_tiles_n = fastdiv_make((unsigned)tiles_n);
    g.full_wg = pl.full_wg; g.pieces = pl.pieces; g.kb_per_piece = pl.per; g.partial = p.partial;
    g.xb_m = g.xb_n = g.xb_cols = 0;
    {   // 2-D XCD blocks for the whole-tile workgroups of an unchunked launch: 8 equal blocks that tile the (rows x tiles_n) grid
        const int rows = pl.full_wg / tiles_n;                                // whole rows of workgroups in the unsplit part
        if (pl.chunks == 1 && pl.full_wg % 8 == 0 && rows * tiles_n == pl.full_wg && tiles_n >= 8) {
            const int per = pl.full_wg / 8;
            int best = 0, best_sum = per % tiles_n == 0 ? per / tiles_n + tiles_n : rows + tiles_n;      // strips: rows per XCD + tiles_n streams
            for (int bn = 2; bn <= tiles_n; ++bn) {
                if (per % bn || tiles_n % bn) continue;
                const int bm = per / bn;
                if (bm < 1 || rows % bm || (rows / bm) * (tiles_n / bn) != 8) continue;
                if (bm + bn < best_sum) { best_sum = bm + bn; best = bn; }
            }
            if (best && best != tiles_n) { g.xb_n = best; g.xb_m = per / best; g.xb_cols = tiles_n / best; }
        }
    }
    const size_t lds = (size_t)36 * W4T * W4N * sizeof(float);
    for (int c = 0; c < pl.chunks; ++c) {
        const int tm_lo = c * pl.tm_per_chunk, tm_hi = tm_lo + pl.tm_per_chunk < tiles_m ? tm_lo + pl.tm_per_chunk : tiles_m;
        const int tile_lo = tm_lo * W4T * mb, tile_hi = pl.chunks == 1 ? Mt_pad : tm_hi * W4T * mb;
        hipEvent_t *ei = ev_in ? ev_in[c] : nullptr, *eg = ev_gemm ? ev_gemm[c] : nullptr;
        {
            const unsigned gx = (unsigned)((8L * (tile_hi - tile_lo) + 255) / 256);
            const int NCB = p.Cin / 32;
            int chunks = (int)((2048 + gx - 1) / gx);
            chunks = chunks < 1 ? 1 : (chunks > NCB ? NCB : chunks);
            const int per = (NCB + chunks - 1) / chunks;
            chunks = (NCB + per - 1) / per;
            if (ei)
                hipExtLaunchKernelGGL(wino4_input_kernel, dim3(gx, chunks), dim3(256), 0, s, ei[0], ei[1], 0, p.x0, p.x0_bytes, p.bs0, p.H, p.W,
                                      p.Cin, p.relu_in, TH, TW, Mt, Mt_pad, per, tile_lo, tile_hi, V);
            else
                hipLaunchKernelGGL(wino4_input_kernel, dim3(gx, chunks), dim3(256), 0, s, p.x0, p.x0_bytes, p.bs0, p.H, p.W, p.Cin, p.relu_in, TH,
                                   TW, Mt, Mt_pad, per, tile_lo, tile_hi, V);
        }
        g.tm0 = tm_lo;
        if (mb == 2) {
            const int grid = (tm_hi - tm_lo) * tiles_n;
            allow_big_lds(reinterpret_cast<const void *>(&wino4_gemm_kernel<2>), lds);
            if (eg)
                hipExtLaunchKernelGGL(wino4_gemm_kernel<2>, dim3(grid), dim3(64 * W4W), lds, s, eg[0], eg[1], 0, g, tiles_n);
            else
                hipLaunchKernelGGL(wino4_gemm_kernel<2>, dim3(grid), dim3(64 * W4W), lds, s, g, tiles_n);
        } else {
            allow_big_lds(reinterpret_cast<const void *>(&wino4_gemm_kernel<1>), lds);
            if (eg)
                hipExtLaunchKernelGGL(wino4_gemm_kernel<1>, dim3(pl.grid), dim3(64 * W4W), lds, s, eg[0], eg[1], 0, g, tiles_n);
            else
                hipLaunchKernelGGL(wino4_gemm_kernel<1>, dim3(pl.grid), dim3(64 * W4W), lds, s, g, tiles_n);
            if (g.pieces > 1) {
                if (ev_red)
                    hipExtLaunchKernelGGL(wino4_reduce_kernel, dim3((tiles_m * tiles_n - g.full_wg) * 64), dim3(256), 0, s, ev_red[0], ev_red[1], 0, g, tiles_n);
                else
                    hipLaunchKernelGGL(wino4_reduce_kernel, dim3((tiles_m * tiles_n - g.full_wg) * 64), dim3(256), 0, s, g, tiles_n);
            }
        }
    }
}

// U [36][Cin/8][N][8] from the BN-folded direct weights w [N][Kp] (k = (ky*3 + kx) * Cin + c), on the host in double;
// G of the points {0, 3/4, -3/4, 3/2, -3/2, inf}
void wino4_transform_weights(const float *w, int N, int Cin, int Kp, float *U) {
    static const double G[6][3] = {{64.0 / 81, 0, 0},
                                   {-128.0 / 243, -32.0 / 81, -8.0 / 27},
                                   {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                                   {32.0 / 243, 16.0 / 81, 8.0 / 27},
                                   {32.0 / 243, -16.0 / 81, 8.0 / 27},
                                   {0, 0, 1}};
    const int KB = Cin / 8;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Cin; ++c) {
            double g[3][3], tmp[6][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(size_t)n * Kp + (size_t)(ky * 3 + kx) * Cin + c];
            for (int i = 0; i < 6; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    U[((((size_t)(i * 6 + j) * KB + c / 8) * N + n) << 3) + (c & 7)] = (float)u;
                }
        }
}

#ifdef STCN_W4_CLOCK
// diagnostic build only: the per-workgroup stamp pairs of the LAST wino4_gemm_kernel launches (not part of the C ABI header)
extern "C" int stcn_debug_w4_clock(unsigned long long *host_out, int n_pairs) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_w4_clock), (size_t)2 * (n_pairs < 8192 ? n_pairs : 8192) * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace stcn
