// winograd.hip - fp32 Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the STCN path on gfx950.
//
// 199 of the 290 GFLOP of a propagated frame are stride-1 3x3 convs (decoder ResBlocks, skip convs, key_comp, the 3x3 of the
// ResNet blocks: reference mivos/model/propagation/{prop_net.py:13-30, modules.py:15-52, mod_resnet.py}).  F(2x2,3x3) computes a
// 2x2 output tile from a 4x4 input tile with 16 multiplies per (cin, cout) instead of 36: 2.25x fewer MFMA FLOP, still exact-fp32
// arithmetic (the transforms only add / subtract / halve).
//     V = B^T d B   (4x4 input tile d, per channel)          U = G g G^T   (3x3 filter g, per (cout, cin); host, once per model)
//     M[xi][nu] = sum_cin U[xi][nu] V[xi][nu]                Y = A^T M A   (2x2 outputs)
// Structure (what the measurements of the direct kernel dictated: the fp32 MFMA shares issue with VALU, so no transform
// arithmetic may sit in the MFMA loop, and the 4x-sized transformed output must never travel through HBM):
//   * wino_input_kernel : X -> V, HBM-bound.  V is laid out [16 positions][C/8 k-blocks][tiles][8 floats] so that the MFMA A
//     fragment of (position, k-block) for 32 consecutive tiles is ONE contiguous KB: 64 lanes x 16 B, fully coalesced.
//     U has the same layout over output channels.
//   * wino_gemm_kernel  : one workgroup = 64 tiles x 64 output channels x ALL 16 positions, 8 waves, each wave 2 positions x
//     (2 x 2) blocks of v_mfma_f32_32x32x2_f32.  Positions are private to a wave, so no operand is shared between waves:
//     the main loop has NO LDS and NO barrier - every wave streams its own fragments global -> registers with buffer loads
//     (loop-invariant per-lane offsets + one scalar k-block offset: no address VALU), one k-block ahead.
//     8 x 16-byte loads per 32 MFMAs = 16 B per CU-cycle from L2, the same as the 64x64 direct tile.
//   * epilogue: the 16 positions of a tile meet in LDS (128 KB, 32 output channels at a time), Y = A^T M A, + bias / residual /
//     ReLU; a work item owns 4 channels of one output row of a tile: 16-byte LDS reads, loads and stores.  Split-K over input channels (few tiles) writes transformed partial sums into the slabs
//     of conv_reduce_kernel (the output transform is linear).
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int WT = 64;        // tiles per workgroup
static constexpr int WN = 64;        // output channels per workgroup

// ------------------------------------------------------------------------------------------------ input transform
// thread = (tile, 16-byte channel group c16 of a 32-channel block): 8 consecutive lanes read one whole 128-byte line of a
// pixel, 8 consecutive tiles (64 lanes) share their overlapping pixels through L1; a store instruction of the wave writes,
// per position, 4 k-blocks x 8 tiles x 32 B = four 256-byte runs of V.  Loops over the 32-channel blocks of its chunk
// (blockIdx.y).  (First version: thread = (tile, half k-block), 32-byte reads at a 2-pixel stride - every 128-byte line
// came from L2 four times: 15 % of all kernel time.)
__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, long x_bs, int H, int W, int C, int relu_in,
                                                         int TH, int TW, int Mt, int Mt_pad, int cb_per_chunk,
                                                         float *__restrict__ V) {
    // XCD-contiguous block order (neighbouring tile rows share pixel rows: keep them in one L2)
    const int nbx = gridDim.x, q8 = nbx >> 3, r8 = nbx & 7, xcd = blockIdx.x & 7;
    const int bx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const long i = bx * 256L + threadIdx.x;
    const int c16 = (int)(i & 7);
    const long tile = i >> 3;
    if (tile >= Mt_pad) return;
    const int KB = C / 8, NCB = C / 32;
    const int cb0 = blockIdx.y * cb_per_chunk, cb1 = min(NCB, cb0 + cb_per_chunk);
    const bool live = tile < Mt;
    const int tpi = TH * TW;
    const int b = live ? (int)(tile / tpi) : 0;
    const int r = live ? (int)(tile - (long)b * tpi) : 0;
    const int ty = r / TW, tx = r - ty * TW;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    bool ok[4][4];
#pragma unroll
    for (int yy = 0; yy < 4; ++yy)
#pragma unroll
        for (int xx = 0; xx < 4; ++xx)
            ok[yy][xx] = live && (unsigned)(y0 + yy) < (unsigned)H && (unsigned)(x0 + xx) < (unsigned)W;
    const float *xb = x + (long)b * x_bs + ((long)y0 * W + x0) * C + 4 * c16;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const long ps = ((long)KB * Mt_pad) << 3;                // floats between positions
    const float lo = relu_in ? 0.f : -__builtin_inff();      // ReLU or identity without a branch per load (a branch there makes hipcc drain vmcnt per load)
    for (int cb = cb0; cb < cb1; ++cb) {
        f32x4 d[4][4];
#pragma unroll
        for (int yy = 0; yy < 4; ++yy)
#pragma unroll
            for (int xx = 0; xx < 4; ++xx) {
                f32x4 v = zero;
                if (ok[yy][xx]) v = *reinterpret_cast<const f32x4 *>(xb + ((long)yy * W + xx) * C + 32 * cb);
                v.x = fmaxf(v.x, lo); v.y = fmaxf(v.y, lo); v.z = fmaxf(v.z, lo); v.w = fmaxf(v.w, lo);      // ReLU or identity, branchless
                d[yy][xx] = v;
            }
        f32x4 t[4][4];                                       // B^T d
#pragma unroll
        for (int xx = 0; xx < 4; ++xx) {
            t[0][xx] = d[0][xx] - d[2][xx];
            t[1][xx] = d[1][xx] + d[2][xx];
            t[2][xx] = d[2][xx] - d[1][xx];
            t[3][xx] = d[1][xx] - d[3][xx];
        }
        const int kb = cb * 4 + (c16 >> 1);
        float *base = V + (((long)kb * Mt_pad + tile) << 3) + 4 * (c16 & 1);
#pragma unroll
        for (int yy = 0; yy < 4; ++yy) {                     // (B^T d) B
            const f32x4 v0 = t[yy][0] - t[yy][2], v1 = t[yy][1] + t[yy][2], v2 = t[yy][2] - t[yy][1], v3 = t[yy][1] - t[yy][3];
            float *dst = base + (long)(yy * 4) * ps;
            *reinterpret_cast<f32x4 *>(dst) = v0;
            *reinterpret_cast<f32x4 *>(dst + ps) = v1;
            *reinterpret_cast<f32x4 *>(dst + 2 * ps) = v2;
            *reinterpret_cast<f32x4 *>(dst + 3 * ps) = v3;
        }
    }
}

// ------------------------------------------------------------------------------------------------ batched GEMM + output transform
struct WinoG {
    const float *V, *U;
    unsigned v_bytes, u_bytes;
    int Mt, Mt_pad, KB, N;
    int TH, TW, OH, OW, B, M;
    const float *bias, *res;
    long res_bs; int res_bmod;
    float *y; long y_bs;
    int relu_out;
    int splitk, kb_per_split;
    float *partial;
    FastDiv fd_tpi, fd_tw, fd_ntile, fd_tiles_n;
};

// PPW positions per wave: 2 -> 8 waves (512 threads), 1 -> 16 waves (1024 threads, 4 per SIMD: more independent instruction
// streams to keep the matrix pipe fed while a wave waits for its fragments)
template <int PPW>
__global__ __launch_bounds__(1024 / PPW) void wino_gemm_kernel(const WinoG p, const int tiles_n, const int ntile) {
    constexpr int NT = 1024 / PPW;                                          // threads per workgroup
    extern __shared__ __attribute__((aligned(16))) float smem[];          // epilogue: [16][64][32]
    const int nblk = gridDim.x;
    auto xcd_contiguous = [](int bid, int nb) {
        const int q8 = nb >> 3, r8 = nb & 7, xcd = bid & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    };
    const int swz = xcd_contiguous(blockIdx.x, nblk);
    const int split = fastdiv(swz, p.fd_ntile), tile = swz - split * ntile;
    const int tm = fastdiv(tile, p.fd_tiles_n), tn = tile - tm * tiles_n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int pos0 = PPW * wave;
    const int kb0 = split * p.kb_per_split, kb1 = min(p.KB, kb0 + p.kb_per_split);
    const int nk = kb1 - kb0;

    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.V), 0, p.v_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.U), 0, p.u_bytes, 0x00020000);
    unsigned va[PPW][2], vb[PPW][2];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
            va[pi][bi] = (unsigned)((((pos0 + pi) * p.KB + kb0) * (long)p.Mt_pad + tm * WT + 32 * bi + l31) * 32 + h * 16);
            vb[pi][bi] = (unsigned)((((pos0 + pi) * p.KB + kb0) * (long)p.N + tn * WN + 32 * bi + l31) * 32 + h * 16);
        }
    const unsigned sa = (unsigned)p.Mt_pad * 32u, sb = (unsigned)p.N * 32u;   // bytes per k-block

    f32x16 acc[PPW][2][2];
#pragma unroll
    for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[pi][bi][bj][e] = 0.f;

    auto load = [&](int k, f32x4 (&fa)[PPW][2], f32x4 (&fb)[PPW][2]) {
        const unsigned oa = (unsigned)k * sa, ob = (unsigned)k * sb;
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi) {
                fa[pi][bi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, va[pi][bi], oa, 0));
                fb[pi][bi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, vb[pi][bi], ob, 0));
            }
    };
    auto compute = [&](const f32x4 (&fa)[PPW][2], const f32x4 (&fb)[PPW][2]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int bj = 0; bj < 2; ++bj)
                        acc[pi][bi][bj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[pi][bi][j], fb[pi][bj][j], acc[pi][bi][bj], 0, 0, 0);
    };
    if (nk > 0) {
        // three fragment sets: the loads of k-block k+2 are issued before the MFMAs of k-block k, so a fragment has two
        // blocks (2 x 32 MFMAs x 2 waves per SIMD ~ 3.5 us) to arrive - V streams from HBM at the big layers.
        // sched_barrier pins "all 8 loads, THEN the 32 MFMAs": left alone hipcc sinks most loads to the end of the MFMA
        // block, right in front of their first use
        f32x4 fa0[PPW][2], fb0[PPW][2], fa1[PPW][2], fb1[PPW][2], fa2[PPW][2], fb2[PPW][2];
        load(0, fa0, fb0);
        load(min(1, nk - 1), fa1, fb1);
        int k = 0;
        for (; k + 2 < nk; k += 3) {
            load(k + 2, fa2, fb2);
            __builtin_amdgcn_sched_barrier(0);
            compute(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            load(min(k + 3, nk - 1), fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            compute(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            load(min(k + 4, nk - 1), fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
            compute(fa2, fb2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (k < nk) compute(fa0, fb0);
        if (k + 1 < nk) compute(fa1, fb1);
    }

    // ---- epilogue: 32 output channels at a time through LDS, Y = A^T M A.
    // work item = (tile, 4 consecutive channels, output row): 16-byte LDS reads, residual loads and stores (the scalar
    // one-channel-per-thread form spent more VALU cycles on addresses and 4-byte memory instructions than the transform needs);
    // 64 x 8 x 2 items per channel half: one pass of the 16-wave instance, two of the 8-wave one.  Arithmetic order = the
    // scalar form's: results are bit-identical to it.
    constexpr int NQ = 1024 / NT;
    const int chq = t & 7, tl = (t >> 3) & 63;
    const int tpi = p.TH * p.TW, ohw = p.OH * p.OW;
    const bool part = p.splitk > 1;
    // residual / output through buffer resources: 32-bit byte offsets (extents < 4 GiB: wino_workspace_floats); a masked store
    // is an out-of-range offset
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res : p.y), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(part ? p.partial : p.y, 0, -1, 0x00020000);
    const long gt = (long)tm * WT + tl;
    const bool work = gt < p.Mt;
    const int gtc = (int)min(gt, (long)p.Mt - 1);
    const int b = fastdiv(gtc, p.fd_tpi);
    const int rr = gtc - b * tpi;
    const int ty = fastdiv(rr, p.fd_tw), tx = rr - ty * p.TW;
    const unsigned ybase = part ? (unsigned)(((long)split * p.M + (long)b * ohw) * p.N) * 4u
                                : (unsigned)b * (unsigned)(p.y_bs ? p.y_bs : (long)ohw * p.N) * 4u;
    const unsigned rbase = (unsigned)(p.res_bmod ? b % p.res_bmod : b) * (unsigned)p.res_bs * 4u;
    const float lo = (p.relu_out && !part) ? 0.f : -__builtin_inff();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        __syncthreads();
#pragma unroll
        for (int pi = 0; pi < PPW; ++pi)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[((pos0 + pi) * WT + 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + l31] = acc[pi][bi][c][r];
        const int n = tn * WN + 32 * c + 4 * chq;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && !part) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
        // the residual block of all passes is requested before the barrier (offsets clamped inside the image for the ragged last
        // tile row / column)
        unsigned po[NQ][2];
        f32x4 rv[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = NQ == 1 ? (t >> 9) : q;
#pragma unroll
            for (int e = 0; e < 2; ++e)
                po[q][e] = (unsigned)((min(2 * ty + row, p.OH - 1) * p.OW + min(2 * tx + e, p.OW - 1)) * p.N + n) * 4u;
        }
        if (p.res && !part) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    rv[q][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, rbase + po[q][e], 0, 0));
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) rv[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (!work) continue;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int row = NQ == 1 ? __builtin_amdgcn_readfirstlane(t >> 9) : q;
            // rows of M are the vertical index xi: pos = 4 xi + nu.  Output row 0: u = m[0] + m[1] + m[2]; row 1: u = m[1] - m[2] - m[3]
            const f32x4 *sm4 = reinterpret_cast<const f32x4 *>(smem) + tl * 8 + chq;
            f32x4 u[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const f32x4 ma = sm4[((row * 4 + nu) * WT) * 8], mb = sm4[(((row + 1) * 4 + nu) * WT) * 8],
                            mc = sm4[(((row + 2) * 4 + nu) * WT) * 8];
                u[nu] = row ? ma - mb - mc : ma + mb + mc;
            }
            const f32x4 yv[2] = {u[0] + u[1] + u[2], u[1] - u[2] - u[3]};
            const bool rok = 2 * ty + row < p.OH;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                f32x4 v = yv[e] + bv + rv[q][e];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) v[c4] = fmaxf(v[c4], lo);
                const bool ok = rok && 2 * tx + e < p.OW;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), ry,
                                                       ok ? ybase + po[q][e] : 0xFFFFFFFFu, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static bool wino_enabled() {
    static const bool on = [] { const char *e = getenv("STCN_WINOGRAD"); return !e || atoi(e) != 0; }();
    return on;
}

// floats of V workspace the Winograd path needs for this conv (0: not eligible)
size_t wino_workspace_floats(const ConvP &p) {
    if (!wino_enabled() || !p.wino_u || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.x1) return 0;
    if (p.Cin % 32 || p.Cin < p.kn.wino_min_cin || p.N % WN || p.bs0 == 0) return 0;
    const long Mt = (long)p.B * ((p.OH + 1) / 2) * ((p.OW + 1) / 2);
    const long Mt_pad = (Mt + WT - 1) / WT * WT;
    // 64-channel layers (K = 8 k-blocks; opt-in: STCN_WINO_MIN_CIN=64): the transforms and the epilogue outweigh the MFMA saving
    // unless the launch is large - 64 -> 64 at 120x216 over a 5-frame group (32400 tiles): 94 -> 76 us back to back, but 81 -> 78 us
    // inside the engine's launch sequence (rocprofv3 trace): not worth 133 MB of V per layer; one frame (6480 tiles): 26 -> 32 us
    if (p.Cin < 128 && Mt < 16384) return 0;
    if (16L * p.Cin * Mt_pad * 4 >= (1L << 32)) return 0;                  // 32-bit buffer offsets
    if ((long)p.B * (p.y_bs ? p.y_bs : (long)p.OH * p.OW * p.N) * 4 >= (1L << 32)) return 0;
    if (p.res && (long)(p.res_bmod ? p.res_bmod : p.B) * p.res_bs * 4 >= (1L << 32)) return 0;
    return (size_t)16 * p.Cin * Mt_pad;
}

// split-K of the Winograd GEMM: with few (tile, channel) workgroups the input channels are cut so that about one round of
// CUs is busy; the pieces write transformed partial sums into the slabs of conv_reduce_kernel
int wino_plan_splitk(const ConvP &p, size_t slab_floats) {
    const int TH = (p.OH + 1) / 2, TW = (p.OW + 1) / 2;
    const int Mt_pad = (p.B * TH * TW + WT - 1) / WT * WT, KB = p.Cin / 8;
    const int ntile = (Mt_pad / WT) * (p.N / WN);
    constexpr int split_below = 160;             // (64 / 128 / 160 measured in round 3: +1.3 / +1.3 / 0 % kernel time)
    int sk = 1;
    if (ntile < split_below) {
        sk = (256 + ntile - 1) / ntile;
        const int smax = KB / 8 < 1 ? 1 : KB / 8;        // at least 8 k-blocks per piece
        sk = sk > smax ? smax : sk;
        while (sk > 1 && (size_t)sk * p.M * p.N > slab_floats) --sk;
    }
    const int per = (KB + sk - 1) / sk;
    return (KB + per - 1) / per;
}

// Winograd launch of a conv that wino_workspace_floats() accepted; V >= that many floats.  ev: optional {start, stop} pairs for
// the transform, GEMM and reduce dispatches.  Returns the split-K factor used.
void wino_launch(const ConvP &p, float *V, size_t slab_floats, hipStream_t s, hipEvent_t *ev_in, hipEvent_t *ev_gemm, hipEvent_t *ev_red) {
    const int TH = (p.OH + 1) / 2, TW = (p.OW + 1) / 2;
    const int Mt = p.B * TH * TW, Mt_pad = (Mt + WT - 1) / WT * WT, KB = p.Cin / 8;
    // ---- input transform: 8 threads per tile, the 32-channel blocks cut into chunks so that the grid fills the chip
    {
        const unsigned gx = (unsigned)((8L * Mt_pad + 255) / 256);
        const int NCB = p.Cin / 32;
        int chunks = (int)((2048 + gx - 1) / gx);
        chunks = chunks < 1 ? 1 : (chunks > NCB ? NCB : chunks);
        const int per = (NCB + chunks - 1) / chunks;
        chunks = (NCB + per - 1) / per;
        if (ev_in)
            hipExtLaunchKernelGGL(wino_input_kernel, dim3(gx, chunks), dim3(256), 0, s, ev_in[0], ev_in[1], 0, p.x0, p.bs0, p.H, p.W, p.Cin,
                                  p.relu_in, TH, TW, Mt, Mt_pad, per, V);
        else
            hipLaunchKernelGGL(wino_input_kernel, dim3(gx, chunks), dim3(256), 0, s, p.x0, p.bs0, p.H, p.W, p.Cin, p.relu_in, TH, TW, Mt,
                               Mt_pad, per, V);
    }
    // ---- GEMM
    WinoG g{};
    g.V = V; g.U = p.wino_u;
    g.v_bytes = (unsigned)((size_t)16 * p.Cin * Mt_pad * 4);
    g.u_bytes = (unsigned)((size_t)16 * p.Cin * p.N * 4);
    g.Mt = Mt; g.Mt_pad = Mt_pad; g.KB = KB; g.N = p.N;
    g.TH = TH; g.TW = TW; g.OH = p.OH; g.OW = p.OW; g.B = p.B; g.M = p.M;
    g.bias = p.bias; g.res = p.res; g.res_bs = p.res_bs; g.res_bmod = p.res_bmod; g.y = p.y; g.y_bs = p.y_bs; g.relu_out = p.relu_out;
    const int tiles_m = Mt_pad / WT, tiles_n = p.N / WN, ntile = tiles_m * tiles_n;
    g.fd_tpi = fastdiv_make((unsigned)(TH * TW)); g.fd_tw = fastdiv_make((unsigned)TW);
    g.fd_ntile = fastdiv_make((unsigned)ntile); g.fd_tiles_n = fastdiv_make((unsigned)tiles_n);
    const int sk = wino_plan_splitk(p, slab_floats);
    g.kb_per_split = (KB + sk - 1) / sk;
    g.splitk = sk; g.partial = p.partial;
    const size_t lds = (size_t)16 * WT * 32 * sizeof(float);
    // 16 waves x 1 position (4 waves per SIMD) feed the matrix pipe a little better on the short-K layers (+1.5-3 % up to 512
    // input channels); with 1024+ channels the 8-wave form with its deeper per-wave prefetch is as good or better
    const int ppw_env = p.kn.wino_ppw;                    // tests run every shape under both instances
    const int ppw = ppw_env == 1 || ppw_env == 2 ? ppw_env : (p.Cin <= 512 ? 1 : 2);
    if (ppw == 1) {
        allow_big_lds(reinterpret_cast<const void *>(&wino_gemm_kernel<1>), lds);
        if (ev_gemm)
            hipExtLaunchKernelGGL(wino_gemm_kernel<1>, dim3(ntile * sk), dim3(1024), lds, s, ev_gemm[0], ev_gemm[1], 0, g, tiles_n, ntile);
        else
            hipLaunchKernelGGL(wino_gemm_kernel<1>, dim3(ntile * sk), dim3(1024), lds, s, g, tiles_n, ntile);
    } else {
        allow_big_lds(reinterpret_cast<const void *>(&wino_gemm_kernel<2>), lds);
        if (ev_gemm)
            hipExtLaunchKernelGGL(wino_gemm_kernel<2>, dim3(ntile * sk), dim3(512), lds, s, ev_gemm[0], ev_gemm[1], 0, g, tiles_n, ntile);
        else
            hipLaunchKernelGGL(wino_gemm_kernel<2>, dim3(ntile * sk), dim3(512), lds, s, g, tiles_n, ntile);
    }
    if (sk > 1) {
        ConvP q = p;
        q.splitk = sk;
        conv_reduce_launch(q, s, ev_red);
    }
}

// U [16][Cin/8][N][8] from the BN-folded direct weights w [N][Kp] (k = (ky*3 + kx) * Cin + c), on the host in double
void wino_transform_weights(const float *w, int N, int Cin, int Kp, float *U) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int KB = Cin / 8;
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < Cin; ++c) {
            double g[3][3], tmp[4][3];
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[(size_t)n * Kp + (size_t)(ky * 3 + kx) * Cin + c];
            for (int i = 0; i < 4; ++i)
                for (int kx = 0; kx < 3; ++kx) tmp[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                    U[((((size_t)(i * 4 + j) * KB + c / 8) * N + n) << 3) + (c & 7)] = (float)u;
                }
        }
}

}  // namespace stcn
