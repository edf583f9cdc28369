// conv_gemm.hip - fp32 implicit-GEMM convolution for gfx950 (CDNA4) on v_mfma_f32_32x32x2_f32.
//
// Every nn.Conv2d of the STCN path (reference mivos/model/propagation/{modules,mod_resnet,prop_net}.py,
// mivos/model/fusion_net.py) lowers to  Y[M,N] = im2col(X)[M,K] * W^T[K,N]  with
//   M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin, activations NHWC (channels contiguous).
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * exact-fp32 MFMA (32x32x2, 64 cycles, a single dependent accumulator chain already reaches the
//     issue rate) -> one 32x32 accumulator tile per wave, 4 waves per workgroup (one per SIMD).
//     Small wave tiles keep the grid large: the path's GEMMs are small-M (1620..25920 rows).
//   * A (im2col gather) and B (weights [N][Kp]) tiles are staged global -> registers -> LDS as 16-byte
//     chunks along K (channels are contiguous in NHWC, so the gather is 16 B per lane, coalesced
//     in 128-B runs); LDS rows are padded to 36 floats so ds_read_b128 fragments are conflict-free
//     (row stride 144 B = 9 slots, 9 is a unit mod 16).
//   * inside a K block of 8 the k order is permuted (lane half h takes k = 4h..4h+3) so a fragment is
//     ONE ds_read_b128 per operand per 4 MFMAs; A and B use the same permutation, the sum is unchanged.
//   * register-staged double buffering: the global loads of tile t+1 are issued before the MFMAs of
//     tile t and written to the other LDS buffer after them; one barrier per K tile.
//   * fused prologue/epilogue: ReLU on the A operand (pre-activation ResBlocks), channel concat of
//     two sources (second one batch-broadcast), bias (+folded BN), residual add, ReLU.
//   * split-K over K tiles with an fp32 slab workspace + a vectorised reduce/epilogue kernel; the
//     host picks the split so that tiles*split fills the 256 CUs in whole rounds.
//   * XCD-aware block remap: each XCD (own L2) gets a contiguous range of (m-tile, n-tile) pairs.
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDT = 36;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static constexpr unsigned OOB = 0x80000000u;    // byte offset beyond any tensor: buffer loads return 0

// tile index -> (m-tile, n-tile).  Default: n fastest (the n-tiles of one m-tile run side by side and share its im2col
// rows in L2).  With more than `pn` n-tiles the tiles are walked in panels of `pn` n-tiles (4 of them):
// the weight slices in flight shrink to one panel, at the price of fetching the activations once per panel - measured
// on up_16_8.skip_conv over a 5-frame group with 64x64 tiles: FETCH_SIZE 652 -> 348 MB (panels of 4), 250 MB (of 2);
// over the whole path -10 % (the heavy shapes run on 128x128 tiles, which already fetch 3.8x less), same speed.
struct TileDiv { FastDiv ntile, tiles_n, per_panel; int tiles_m; };      // invariants of one launch (host: conv_launch)
__device__ __forceinline__ void tile_to_mn(int tile, int tiles_n, int ntile, int pn, const TileDiv &td, int &tm, int &tn) {
    if (pn > 0 && tiles_n % pn == 0 && tiles_n > pn) {
        const int per_panel = td.tiles_m * pn;
        const int panel = fastdiv(tile, td.per_panel), rem = tile - panel * per_panel;
        tm = pn == 4 ? rem >> 2 : rem / pn;
        tn = panel * pn + (rem - tm * pn);
    } else {
        tm = fastdiv(tile, td.tiles_n);
        tn = tile - tm * tiles_n;
    }
    (void)ntile;
}

// WM x WN waves per workgroup, each owning RM x RN accumulator blocks of 32x32: workgroup tile (32 WM RM) x (32 WN RN).
// PW: pointwise instance (1x1, stride 1, one dense source): no tap walk, no validity masks, no row decode - the set-up and
// per-K-tile scalar code of the general instance is as long as the MFMA work of an 8-K-tile ResNet 1x1 conv.
// STCN_PW_WAVES: waves per SIMD the one-accumulator-block instances (64x64 and 128x32 tiles, pointwise or not) are compiled for -
// at most 512 / n registers.  hipcc's own choice was 136 registers = 3 workgroups per CU; 4 (118 registers, no spills) puts a
// fourth workgroup on the CU behind which the short-K layers (8 K tiles between a prologue and a residual epilogue) hide their
// latencies: conv_gemm kernels of the solo R1 leg 118.4 -> 112.9 ms, whole leg 378.3 -> 374.1 ms (tools/gpu_ab_trace.sh, 3 alternating
// repetitions, gpurun_out/r4a/ab_pw_waves.txt).  -DSTCN_PW_WAVES=1 restores the compiler's choice.
#ifndef STCN_PW_WAVES
#define STCN_PW_WAVES 4
#endif
template <int WM, int WN, int RM, int RN, bool SMALLC, bool RELU, bool PW = false>
__global__ __launch_bounds__(256, (RM == 1 && RN == 1) ? STCN_PW_WAVES : 1) void conv_gemm_kernel(const ConvP p, const int tiles_n,
                                                        const int ntile, const int kt_per_split, const TileDiv td) {
    constexpr int PA = WM * RM, PB = WN * RN;          // 32-row pieces of the A / B tiles (= staging chunks per thread)
    constexpr int BM = 32 * PA, BN = 32 * PB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][LDT]
    float *Bs = smem + 2 * BM * LDT;   // [2][BN][LDT]

    // ---- block -> (split, m-tile, n-tile), XCD-contiguous (bijective remap)
    auto xcd_contiguous = [](int bid, int nblk) {       // block id -> position in a per-XCD contiguous order
        const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    };
    int split, tile, kper = kt_per_split;
    bool piece = false;                                // tail balancing: this workgroup computes a K piece of a tile
    if (p.rem_split > 1) {                             // blocks [0, rem_full): whole tiles; then the pieces
        const int bid = blockIdx.x;
        if (bid < p.rem_full) {
            tile = xcd_contiguous(bid, p.rem_full);
            split = 0;
        } else {
            const int j = xcd_contiguous(bid - p.rem_full, (int)gridDim.x - p.rem_full);
            const int rt = j / p.rem_split;
            split = j - rt * p.rem_split;
            tile = p.rem_full + rt;
            kper = p.rem_per;
            piece = true;
        }
    } else {
        const int swz = xcd_contiguous(blockIdx.x, gridDim.x);
        split = fastdiv(swz, td.ntile);
        tile = swz - split * ntile;
    }
    int tm, tn;
    tile_to_mn(tile, tiles_n, ntile, p.panel, td, tm, tn);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int kc = t & 7, r0 = t >> 3;

    // Buffer resources: out-of-range byte offsets read as zero in hardware, so padding taps, ragged
    // rows/columns and the K tail need no data fix-up (the fp32 MFMA shares the SIMD's FMA lanes with
    // VALU work - measured: MFMA-busy + VALU + LDS cycles add up - so every VALU op here costs MFMA time).
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x0), 0, p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x1 ? p.x1 : p.x0), 0, p.x1 ? p.x1_bytes : p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);

    // ---- per-thread im2col row state (rows r0 + 32*i of the A tile)
    int ih0[PA], iw0[PA], roff0[PA], roff1[PA];      // roffX: byte offset of (b, ih0, iw0, channel kc*4) in source X
    unsigned vmask[PA];                               // fast path: bit (kh*KW+kw) = tap inside the image
    bool rvalid[PA];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int m = tm * BM + r0 + 32 * i;
        rvalid[i] = m < p.M;
        const int mm = rvalid[i] ? m : 0;
        if (PW) {                                      // the im2col row IS activation row m; invalid rows read out of range (zero)
            ih0[i] = 0; iw0[i] = 0; vmask[i] = 0;
            roff0[i] = rvalid[i] ? (mm * p.c0 + kc * 4) * 4 : (int)OOB;
            roff1[i] = roff0[i];
            continue;
        }
        if (p.pointwise) {                             // 1x1 / stride 1 / dense: no (b, oh, ow) decode, two integer divisions saved per row
            ih0[i] = 0; iw0[i] = 0;
            roff0[i] = (mm * p.c0 + (SMALLC ? 0 : kc * 4)) * 4;
            roff1[i] = roff0[i];
            vmask[i] = rvalid[i] ? 1u : 0u;
            continue;
        }
        const int b = fastdiv(mm, p.fd_ohw), pix = mm - b * ohw;
        const int oh = fastdiv(pix, p.fd_ow), ow = pix - oh * p.OW;
        ih0[i] = oh * p.stride - p.pad;
        iw0[i] = ow * p.stride - p.pad;
        roff0[i] = (b * (int)p.bs0 + (ih0[i] * p.W + iw0[i]) * p.c0 + (SMALLC ? 0 : kc * 4)) * 4;
        roff1[i] = (b * (int)p.bs1 + (ih0[i] * p.W + iw0[i]) * p.c1 + (SMALLC ? 0 : kc * 4)) * 4;
        unsigned vm = 0;
        if (!SMALLC) {
            int bit = 0;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw, ++bit)
                    if (rvalid[i] && (unsigned)(ih0[i] + kh) < (unsigned)p.H && (unsigned)(iw0[i] + kw) < (unsigned)p.W)
                        vm |= 1u << bit;
        }
        vmask[i] = vm;
    }
    unsigned woff[PB];                                // byte offset of weight row n, chunk kc (OOB when n >= N)
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int n = tn * BN + r0 + 32 * i;
        woff[i] = n < p.N ? (unsigned)((n * p.Kp + kc * 4) * 4) : OOB;
    }

    const int nkt = p.Kp / BK;
    const int kt0 = split * kper;
    const int kt1 = min(nkt, kt0 + kper);

    // Filter-tap walk.  Fast path (Cin and c0 multiples of 32, <= 32 taps): a K tile lies inside one tap
    // and one source, so tap / source / channel base are workgroup-uniform and live in scalar registers.
    // Generic path (stems, Cin = 4/8/12): per-thread decode of its 4-channel chunk.
    // The fast path walks K as (channel block, kh, kw) with the TAP fastest, not in the weights' memory order
    // (kh, kw, channel): the 9 taps of a 32-channel block touch the same three image rows within 9 consecutive K
    // tiles, so the im2col re-reads hit L1 / L2.  In (kh, kw, channel) order the three kh phases of a 3x3 conv are
    // a third of the kernel apart and every input row came from beyond L2 three times (FETCH_SIZE 2.6x the input).
    // Only the visiting order changes: K tile position q -> weight K tile (tap * Cin/32 + channel block).
    const int ncb = p.Cin / BK, ntap = p.KH * p.KW;
    int u_kh, u_kw, u_cb;
    if (PW) {
        u_cb = kt0 * BK; u_kh = 0; u_kw = 0;
    } else if (SMALLC) {
        const int k0 = kt0 * BK;
        const int tap = k0 / p.Cin;
        u_cb = k0 - tap * p.Cin;
        u_kh = tap / p.KW;
        u_kw = tap - u_kh * p.KW;
    } else {
        const int cbi = kt0 / ntap, tap = kt0 - cbi * ntap;
        u_cb = cbi * BK;
        u_kh = tap / p.KW;
        u_kw = tap - u_kh * p.KW;
    }

    // Software pipeline (one K tile = 16 MFMAs = 1024 matrix-pipe cycles per wave):
    //   iteration t runs C(t) = LDS->MFMA and, between those MFMAs,
    //     G(t+2): global -> staging registers (buffer loads, issued right after the first MFMAs),
    //     S(t+1): staging registers -> the other LDS buffer (after the later MFMAs).
    //   Two staging register sets (tile parity) give every load ~1.5 iterations to land before its
    //   write-back; sched_barrier pins the placement (hipcc otherwise sinks the loads to the loop end).
    f32x4 ra[2][PA], rb[2][PB];
    // per-tile uniform (fast path) / per-thread (generic) tap state produced by g_tap
    int g_kh = 0, g_kw = 0, g_coff = 0, g_wkt = 0;     // g_wkt: K tile of the weight array for this position
    unsigned g_bit = 0;
    bool g_src1 = false, g_kvalid = true;
    auto g_tap = [&](int kt) {
        if (PW) {
            g_coff = u_cb * 4;
            g_wkt = min(kt, nkt - 1);
            u_cb += BK;
        } else if (SMALLC) {
            const int k = kt * BK + kc * 4;
            g_wkt = kt < nkt ? kt : nkt - 1;
            g_kvalid = k < p.K;
            const int tap = fastdiv(k, p.fd_cin);          // two run-time divisions per thread and K tile cost ~50 VALU ops of the
            const int c = k - tap * p.Cin;                 // 16-MFMA tile; with the launch's magic numbers: 4
            g_kh = fastdiv(tap, p.fd_kw);
            g_kw = tap - g_kh * p.KW;
            g_coff = ((g_kh * p.W + g_kw) * p.c0 + c) * 4;
        } else {
            g_src1 = u_cb >= p.c0;
            g_bit = (unsigned)(u_kh * p.KW + u_kw);
            const int cs = g_src1 ? p.c1 : p.c0;
            g_coff = ((u_kh * p.W + u_kw) * cs + (g_src1 ? u_cb - p.c0 : u_cb)) * 4;
            g_wkt = min((int)g_bit * ncb + u_cb / BK, nkt - 1);          // past the K range: stay inside [N][Kp]
            if (++u_kw == p.KW) { u_kw = 0; if (++u_kh == p.KH) { u_kh = 0; u_cb += BK; } }
        }
    };
    auto g_a = [&](int i, auto setc) {
        constexpr int ST = decltype(setc)::value;
        unsigned voff;
        if (PW) {
            voff = (unsigned)roff0[i] + (unsigned)g_coff;         // out of range stays out of range (offsets < 2 GiB)
            ra[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 0, 0));
        } else if (SMALLC) {
            const bool ok = g_kvalid && rvalid[i] && (unsigned)(ih0[i] + g_kh) < (unsigned)p.H &&
                            (unsigned)(iw0[i] + g_kw) < (unsigned)p.W;
            voff = ok ? (unsigned)(roff0[i] + g_coff) : OOB;
            ra[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, 0, 0));
        } else {
            const bool ok = (vmask[i] >> g_bit) & 1u;
            voff = ok ? (unsigned)((g_src1 ? roff1[i] : roff0[i]) + g_coff) : OOB;
            ra[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(g_src1 ? rs1 : rs0, voff, 0, 0));
        }
    };
    auto g_b = [&](int i, auto setc) {
        constexpr int ST = decltype(setc)::value;
        rb[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[i], g_wkt * (BK * 4), 0));
    };
    float *const a_st = As + r0 * LDT + kc * 4;
    float *const b_st = Bs + r0 * LDT + kc * 4;
    auto s_a = [&](int i, int buf, auto setc) {
        constexpr int ST = decltype(setc)::value;
        f32x4 v = ra[ST][i];
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4 *>(a_st + (buf * BM + 32 * i) * LDT) = v;
    };
    auto s_b = [&](int i, int buf, auto setc) {
        constexpr int ST = decltype(setc)::value;
        *reinterpret_cast<f32x4 *>(b_st + (buf * BN + 32 * i) * LDT) = rb[ST][i];
    };
    auto gload_all = [&](int kt, auto setc) {
        g_tap(kt);
#pragma unroll
        for (int i = 0; i < PA; ++i) g_a(i, setc);
#pragma unroll
        for (int i = 0; i < PB; ++i) g_b(i, setc);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    f32x16 acc[RM][RN];
#pragma unroll
    for (int a = 0; a < RM; ++a)
#pragma unroll
        for (int b = 0; b < RN; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int nk = kt1 - kt0;
    if (nk > 0) {
        gload_all(kt0, I0{});
        gload_all(kt0 + 1, I1{});                  // tile 1 -> set 1 (past the K range: zeros / last tile);
                                                   // issued before the first write-back so both latencies overlap
#pragma unroll
        for (int i = 0; i < PA; ++i) s_a(i, 0, I0{});
#pragma unroll
        for (int i = 0; i < PB; ++i) s_b(i, 0, I0{});
    }
    __syncthreads();

    // Pointwise instance: the residual of the wave's accumulator block is requested HERE, before the K loop, not in the epilogue -
    // the ResNet conv3 layers (64 -> 256, 128 -> 512, 256 -> 1024: 2 - 8 K tiles) are chains of dependent round trips otherwise
    // (operands -> MFMAs -> residual -> store); with the residual in flight under the K loop they are one round trip shorter.
    float rpre[16];
    const bool res_pre = PW && p.res && p.affine_out && p.splitk == 1 && !piece;
    if (PW) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rpre[r] = 0.f;
        if (res_pre) {
            const __amdgpu_buffer_rsrc_t rr0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res), 0, (unsigned)((long)p.M * p.N * 4), 0x00020000);
            const int n0 = tn * BN + wn * 32 + (lane & 31), mb0 = tm * BM + wm * 32 + 4 * (lane >> 5);
            const unsigned v00 = n0 < p.N ? (unsigned)(((long)mb0 * p.N + n0) * 4) : OOB, n40 = (unsigned)p.N * 4u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rpre[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr0, v00 + (unsigned)((r & 3) + 8 * (r >> 2)) * n40, 0, 0));
        }
    }
    const int arow = (wm * RM * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * RN * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    constexpr int NPIECE = PA + PB;
    constexpr int NS = 16 * RM * RN;                  // MFMAs (= staging slots) per K tile
    // iteration `it`: C(it) from LDS buffer it&1; G(it+2) -> register set it&1;
    //                 S(it+1) from the other set -> LDS buffer (it+1)&1
    auto iteration = [&](int it, auto gsc) {
        constexpr int GS = decltype(gsc)::value;
        using SS = std::integral_constant<int, GS ^ 1>;
        constexpr int buf = GS;
        const int kt2 = kt0 + it + 2;
        const float *a_s = As + buf * BM * LDT + arow;
        const float *b_s = Bs + buf * BN * LDT + brow;
        f32x4 fa[2][RM], fb[2][RN];
#pragma unroll
        for (int a = 0; a < RM; ++a) fa[0][a] = *reinterpret_cast<const f32x4 *>(a_s + a * 32 * LDT);
#pragma unroll
        for (int b = 0; b < RN; ++b) fb[0][b] = *reinterpret_cast<const f32x4 *>(b_s + b * 32 * LDT);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int cur = kb & 1;
            if (kb < 3) {                              // prefetch the next fragments
#pragma unroll
                for (int a = 0; a < RM; ++a) fa[cur ^ 1][a] = *reinterpret_cast<const f32x4 *>(a_s + a * 32 * LDT + (kb + 1) * 8);
#pragma unroll
                for (int b = 0; b < RN; ++b) fb[cur ^ 1][b] = *reinterpret_cast<const f32x4 *>(b_s + b * 32 * LDT + (kb + 1) * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < RM; ++a)
#pragma unroll
                    for (int b = 0; b < RN; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][a][j], fb[cur][b][j], acc[a][b], 0, 0, 0);
                        // staging pieces between the MFMAs (unconditional: a branch per piece makes hipcc drain
                        // vmcnt(0) at every block boundary; on the tail iterations they fetch zeros / the last
                        // weight tile and write the unused LDS buffer, which is harmless)
                        const int slot = ((4 * kb + j) * RM + a) * RN + b;
                        if (slot == 0) g_tap(kt2);
                        else if (slot <= PA) g_a(slot - 1, gsc);
                        else if (slot <= NPIECE) g_b(slot - 1 - PA, gsc);
                        else if (slot >= NS - 1 - NPIECE && slot < NS - 1 - PB) s_a(slot - (NS - 1 - NPIECE), buf ^ 1, SS{});
                        else if (slot >= NS - 1 - PB && slot < NS - 1) s_b(slot - (NS - 1 - PB), buf ^ 1, SS{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        __syncthreads();
    };
    int it = 0;
    for (; it + 1 < nk; it += 2) {
        iteration(it, I0{});
        iteration(it + 1, I1{});
    }
    if (it < nk) iteration(it, I0{});

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ba = 0; ba < RM; ++ba)
#pragma unroll
        for (int bb = 0; bb < RN; ++bb) {
            const int n = tn * BN + (wn * RN + bb) * 32 + (lane & 31);
            if (n >= p.N && !piece) continue;
            const int mbase = tm * BM + (wm * RM + ba) * 32 + 4 * (lane >> 5);
            const f32x16 &c = acc[ba][bb];
            if (piece) {                               // tile-local partial sums, all rows / columns
                float *dst = p.partial + ((long)(tile - p.rem_full) * p.rem_split + split) * (BM * BN);
                const int col = (wn * RN + bb) * 32 + (lane & 31), row0 = (wm * RM + ba) * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(row0 + (r & 3) + 8 * (r >> 2)) * BN + col] = c[r];
                continue;
            }
            if (p.splitk > 1) {
                float *dst = p.partial + (long)split * p.M * p.N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < p.M) dst[(long)m * p.N + n] = c[r];
                }
                continue;
            }
            const float bv = p.bias ? p.bias[n] : 0.f;
            if (p.affine_out) {
                // dense [M][N] output (and residual): buffer stores / loads, one address add per row; rows beyond M are dropped
                // by the hardware range check of the vector offset (first version: ~8 address / predicate VALU ops per row,
                // which the fp32 MFMA of the co-resident workgroups cannot hide)
                const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (unsigned)((long)p.M * p.N * 4), 0x00020000);
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res : p.y), 0, (unsigned)((long)p.M * p.N * 4), 0x00020000);
                const unsigned v0 = (unsigned)(((long)mbase * p.N + n) * 4), n4 = (unsigned)p.N * 4u;
                // rows >= M lie at >= M*N*4 bytes, beyond the descriptor range (no wrap: (M + 128) * N * 4 < 4 GiB, run_conv).
                // The 16 residual values are requested TOGETHER, before the first use: with the load behind `if (p.res)` inside the
                // loop hipcc waited vmcnt(0) per element - 16 serialized round trips per accumulator block, longer than the
                // 8-K-tile MFMA body of the ResNet 1x1 convs that carry the residual.
                float rv[16];
                if (PW && res_pre) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = rpre[r];
                } else if (p.res) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, v0 + (unsigned)((r & 3) + 8 * (r >> 2)) * n4, 0, 0));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = 0.f;
                }
                const float lo = p.relu_out ? 0.f : -__builtin_inff();
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned vo = v0 + (unsigned)((r & 3) + 8 * (r >> 2)) * n4;
                    const float v = fmaxf(c[r] + bv + rv[r], lo);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, vo, 0, 0);
                }
                continue;
            }
            // general layout (batch strides / broadcast residual): addresses first, then all residual loads, then the stores
            const bool needb = p.res != nullptr || p.y_bs != 0;
            long yo[16], ro[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);          // rows beyond M: clamped here, not stored below
                yo[r] = (long)m * p.N + n;
                ro[r] = 0;
                if (needb) {
                    const int b = p.B == 1 ? 0 : fastdiv(m, p.fd_ohw);
                    const long po = (long)(m - b * ohw) * p.N + n;
                    ro[r] = (long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs + po;
                    if (p.y_bs) yo[r] = (long)b * p.y_bs + po;
                }
            }
            float rv[16];
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = p.res[ro[r]];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            }
            // every value first (ONE wait for the residual loads), then the masked stores: with the arithmetic inside the masked
            // blocks hipcc put a vmcnt(0) in front of each, which also waits for the previous block's STORE
            const float lo = p.relu_out ? 0.f : -__builtin_inff();
            float ov[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { ov[r] = fmaxf(c[r] + bv + rv[r], lo); asm volatile("" : "+v"(ov[r])); }   // pinned: hipcc sinks it into the masked blocks otherwise
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (mbase + (r & 3) + 8 * (r >> 2) < p.M) p.y[yo[r]] = ov[r];
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pointwise CHAIN kernel: the ResNet-50 1x1 convs (conv1 / conv3 of every bottleneck, 2 - 32 K tiles) with dense [M][N] output.
// The one-tile-per-workgroup instance above pays, per 64x64 tile, a prologue in which nothing computes (tile decode, two K tiles
// of global loads before the first MFMA) and an epilogue - around as little as 2 K tiles of work (64 -> 256 of res2).  Here a
// workgroup walks `nt` tiles (strided over the grid, see below) as ONE software pipeline: the flattened
// sequence of (tile, K tile) pairs is loaded two steps ahead exactly as above, so the first K tiles of tile j + 1 are already in
// LDS / in flight while tile j finishes, its epilogue (residual requested a tile ahead, bias, ReLU, 16 row stores) runs between two
// pipeline steps, and the grid is sized to ONE resident set of workgroups (4 per CU): no ragged last round, no tail split, no reduce.
// Same staging layout, fragment reads, MFMA order and accumulation order as conv_gemm_kernel<2,2,1,1,false,RELU,true>: results are
// bit-identical to it.
template <bool RELU>
__global__ __launch_bounds__(256, STCN_PW_WAVES) void pw_chain_kernel(const ConvP p, const int tiles_n, const int ntile, const int nt,
                                                                     const TileDiv td) {
    constexpr int BM = 64, BN = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][LDT]
    float *Bs = smem + 2 * BM * LDT;   // [2][BN][LDT]
    const int nb = gridDim.x, q8 = nb >> 3, r8 = nb & 7, xcd = blockIdx.x & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);      // XCD-contiguous
    // tile walk of this workgroup: tile0, tile0 + tstep, ... (< tile1).  Consecutive (p.kn.pw_chain == 1): nt neighbouring tiles - the
    // workgroup re-reads its own activation rows, which the residual / output streams of the 127 other workgroups of the XCD have
    // pushed out of the 4 MB L2 by then (FETCH_SIZE of the class +22 %).  Strided (default): in step j the grid as a whole covers the
    // contiguous band [j G, (j + 1) G) of tiles, exactly as the resident set of a one-tile launch does: the sibling n-tiles of an
    // activation block run side by side on one XCD and share it in L2.
    const bool strided = p.kn.pw_chain != 1;
    const int tile0 = strided ? swz : swz * nt, tstep = strided ? nb : 1;
    const int tile1 = strided ? ntile : min(ntile, tile0 + nt);
    if (tile0 >= tile1) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kc = t & 7, r0 = t >> 3;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x0), 0, p.x0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, p.w_bytes, 0x00020000);
    const unsigned out_bytes = (unsigned)((long)p.M * p.N * 4);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.res ? p.res : p.y), 0, out_bytes, 0x00020000);
    const int nkt = p.Kp / BK;

    // ---- load stream: (l_tile, l_kt) walks the flattened sequence; past the last tile every offset is out of range (zeros)
    unsigned aoff[2], woff[2];
    int l_tile = tile0, l_kt = 0;
    auto set_load_tile = [&](int tile) {
        int tm, tn;
        tile_to_mn(min(tile, ntile - 1), tiles_n, ntile, p.panel, td, tm, tn);
        const bool live = tile < tile1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = tm * BM + r0 + 32 * i, n = tn * BN + r0 + 32 * i;
            aoff[i] = live && m < p.M ? (unsigned)((m * p.c0 + kc * 4) * 4) : OOB;
            woff[i] = live && n < p.N ? (unsigned)((n * p.Kp + kc * 4) * 4) : OOB;
        }
    };
    set_load_tile(tile0);
    f32x4 ra[2][2], rb[2][2];
    int g_k = 0;                                       // byte offset of the K tile being loaded (set by g_step)
    auto g_step = [&]() {                              // advance the load stream by one K tile
        g_k = l_kt * (BK * 4);
        if (++l_kt == nkt) { l_kt = 0; l_tile += tstep; }
    };
    auto g_next_tile = [&]() { if (l_kt == 0) set_load_tile(l_tile); };     // after the loads of the last K tile of a tile were issued
    auto g_a = [&](int i, auto setc) {
        constexpr int ST = decltype(setc)::value;
        ra[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs0, aoff[i] + (unsigned)g_k, 0, 0));
    };
    auto g_b = [&](int i, auto setc) {
        constexpr int ST = decltype(setc)::value;
        rb[ST][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, woff[i] + (unsigned)g_k, 0, 0));
    };
    float *const a_st = As + r0 * LDT + kc * 4;
    float *const b_st = Bs + r0 * LDT + kc * 4;
    auto s_a = [&](int i, int buf, auto setc) {
        constexpr int ST = decltype(setc)::value;
        f32x4 v = ra[ST][i];
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4 *>(a_st + (buf * BM + 32 * i) * LDT) = v;
    };
    auto s_b = [&](int i, int buf, auto setc) {
        constexpr int ST = decltype(setc)::value;
        *reinterpret_cast<f32x4 *>(b_st + (buf * BN + 32 * i) * LDT) = rb[ST][i];
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto gload_all = [&](auto setc) {
        g_step();
        g_a(0, setc); g_a(1, setc); g_b(0, setc); g_b(1, setc);
        g_next_tile();
    };

    // ---- compute stream
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    int c_tile = tile0, k_left = nkt;
    float rpre[16];
    unsigned v0 = OOB;                                 // byte offset of (first row, column) of this lane's accumulator block; OOB: column >= N
    const unsigned n4 = (unsigned)p.N * 4u;
    float bv = 0.f;
    auto tile_begin = [&](int tile) {                  // output coordinates of the tile, its bias and its residual block (in flight under the K loop)
        int tm, tn;
        tile_to_mn(tile, tiles_n, ntile, p.panel, td, tm, tn);
        const int n0 = tn * BN + wn * 32 + (lane & 31), mb0 = tm * BM + wm * 32 + 4 * (lane >> 5);
        v0 = n0 < p.N ? (unsigned)(((long)mb0 * p.N + n0) * 4) : OOB;
        bv = p.bias && n0 < p.N ? p.bias[n0] : 0.f;
        if (p.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rpre[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, v0 + (unsigned)((r & 3) + 8 * (r >> 2)) * n4, 0, 0));
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) rpre[r] = 0.f;
        }
    };
    const float lo = p.relu_out ? 0.f : -__builtin_inff();
    auto tile_end = [&]() {                            // epilogue of the finished tile; rows >= M lie beyond the descriptor range (dropped)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(acc[r] + bv + rpre[r], lo);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, v0 + (unsigned)((r & 3) + 8 * (r >> 2)) * n4, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        c_tile += tstep;
        if (c_tile < tile1) tile_begin(c_tile);
    };

    gload_all(I0{});
    gload_all(I1{});
    s_a(0, 0, I0{}); s_a(1, 0, I0{}); s_b(0, 0, I0{}); s_b(1, 0, I0{});
    __syncthreads();
    tile_begin(tile0);
    const int arow = (wm * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    // step `it`: C(it) from LDS buffer it & 1; G(it + 2) -> register set it & 1; S(it + 1) from the other set -> LDS buffer (it + 1) & 1
    auto step = [&](auto gsc) {
        constexpr int GS = decltype(gsc)::value;
        using SS = std::integral_constant<int, GS ^ 1>;
        constexpr int buf = GS;
        const float *a_s = As + buf * BM * LDT + arow;
        const float *b_s = Bs + buf * BN * LDT + brow;
        f32x4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const f32x4 *>(a_s);
        fb[0] = *reinterpret_cast<const f32x4 *>(b_s);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int cur = kb & 1;
            if (kb < 3) {
                fa[cur ^ 1] = *reinterpret_cast<const f32x4 *>(a_s + (kb + 1) * 8);
                fb[cur ^ 1] = *reinterpret_cast<const f32x4 *>(b_s + (kb + 1) * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][j], fb[cur][j], acc, 0, 0, 0);
                const int slot = 4 * kb + j;
                if (slot == 0) g_step();
                else if (slot <= 2) g_a(slot - 1, gsc);
                else if (slot <= 4) g_b(slot - 3, gsc);
                else if (slot == 5) g_next_tile();
                else if (slot >= 11 && slot < 13) s_a(slot - 11, buf ^ 1, SS{});
                else if (slot >= 13 && slot < 15) s_b(slot - 13, buf ^ 1, SS{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (--k_left == 0) { k_left = nkt; tile_end(); }
    };
    const int total = ((tile1 - tile0 + tstep - 1) / tstep) * nkt;
    int it = 0;
    for (; it + 1 < total; it += 2) {
        step(I0{});
        step(I1{});
    }
    if (it < total) step(I0{});
}

// tiles per workgroup of the chain kernel for this conv, or 0 when it does not apply (conv_plan decides; conv_launch follows p.chain)
static int pw_chain_tiles(const ConvP &p, int force_splitk) {
    if (!p.kn.pw_chain || force_splitk > 0 || !p.pointwise || !p.affine_out || p.N <= 32 || (p.Cin % 32) != 0) return 0;
    // lanes whose column is >= N carry the 0x80000000 sentinel as their offset and add row * N * 4 to it: that only stays out of range
    // (a dropped store) while the output itself is shorter than 2 GiB - or when no such lane exists
    if ((p.N % 64) != 0 && ((long)p.M + 64) * p.N * 4 >= (1L << 31)) return 0;
    const long ntile = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    static const int cus = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    const long resident = (long)cus * STCN_PW_WAVES;                  // workgroups the chip holds at once (one wave per SIMD each)
    if (ntile < resident + resident / 2) return 0;                    // up to 1.5 resident sets: the one-tile instance (with its tail balancing) is as good
    const long nt = (ntile + resident - 1) / resident;
    return (int)(nt > 16 ? 16 : nt);
}

// y = sum_s partial[s] + bias (+res) (relu); 4 columns per thread
__global__ __launch_bounds__(256) void conv_reduce_kernel(const ConvP p) {
    const long total4 = (long)p.M * p.N / 4;
    const int ohw = p.OH * p.OW;
    const long slab = (long)p.M * p.N;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
        f32x4 v = *reinterpret_cast<const f32x4 *>(p.partial + e);
        for (int s = 1; s < p.splitk; ++s) {
            const f32x4 u = *reinterpret_cast<const f32x4 *>(p.partial + s * slab + e);
            v += u;
        }
        if (p.bias) v += *reinterpret_cast<const f32x4 *>(p.bias + n);
        long yo = e;
        if (p.res || p.y_bs) {
            const int b = p.B == 1 ? 0 : m / ohw;
            const long po = (long)(m - b * ohw) * p.N + n;
            if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + (long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs + po);
            if (p.y_bs) yo = (long)b * p.y_bs + po;
        }
        if (p.relu_out) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4 *>(p.y + yo) = v;
    }
}

// tail balancing: y(tile) = sum over the K pieces of the tile-local partials + epilogue; one block per 32 tile rows
__global__ __launch_bounds__(256) void conv_reduce_tiles_kernel(const ConvP p, const int tiles_n, const int ntile, const int BM, const int BN,
                                                                const TileDiv td) {
    const int rows_per_blk = 1024 / BN;                          // 256 threads x 4 columns
    const int blks_per_tile = BM / rows_per_blk;
    const int rt = blockIdx.x / blks_per_tile, rb = blockIdx.x - rt * blks_per_tile;
    const int tile = p.rem_full + rt;
    int tm, tn;
    tile_to_mn(tile, tiles_n, ntile, p.panel, td, tm, tn);
    const int e4 = threadIdx.x * 4;
    const int row = rb * rows_per_blk + e4 / BN, col = e4 - (e4 / BN) * BN;
    const int m = tm * BM + row, n = tn * BN + col;
    if (m >= p.M || n >= p.N) return;
    const float *src = p.partial + (long)rt * p.rem_split * (BM * BN) + row * BN + col;
    f32x4 v = *reinterpret_cast<const f32x4 *>(src);
    for (int s = 1; s < p.rem_split; ++s) v += *reinterpret_cast<const f32x4 *>(src + (long)s * (BM * BN));
    if (p.bias) v += *reinterpret_cast<const f32x4 *>(p.bias + n);
    long yo = (long)m * p.N + n;
    if (p.res || p.y_bs) {
        const int ohw = p.OH * p.OW;
        const int b = p.B == 1 ? 0 : m / ohw;
        const long po = (long)(m - b * ohw) * p.N + n;
        if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + (long)(p.res_bmod ? b % p.res_bmod : b) * p.res_bs + po);
        if (p.y_bs) yo = (long)b * p.y_bs + po;
    }
    if (p.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4 *>(p.y + yo) = v;
}

static inline bool narrow_variant(const ConvP &p) { return p.N <= 32; }
static inline bool smallc_variant(const ConvP &p) {
    return (p.Cin % 32) != 0 || (p.x1 && (p.c0 % 32) != 0) || p.KH * p.KW > 32;
}
// ---- launch plan ---------------------------------------------------------------------------------------------
// Cost model in K-tile times of one 64x64 workgroup (~0.45 us): the chip runs "rounds" of 256 workgroups (one per CU;
// co-resident workgroups of a CU share its MFMA pipes, so only the count per CU matters), a workgroup costs its K tiles
// plus ~3 for prologue / epilogue.  Candidates:
//   * tile 64x64 (128x32 for Cout <= 32) or, for deep-K GEMMs, 128x128 (each wave 2x2 accumulator blocks: half the
//     LDS reads and staging per MFMA, +8 % on a GEMM that fills the chip evenly; +4 % on the 5-frame decoder batches at
//     1/8 and 1/16 scale).  STCN_CONV_BIG: 0 never, 1 the model chooses (default), 2 always for deep-K GEMMs;
//   * plain split-K s (slabs + conv_reduce_kernel) when the tiles do not fill the chip;
//   * tail balancing when they fill it more than once: the whole rounds run unsplit and only the tiles of the ragged
//     last round are cut into K pieces (1620 tiles = 6 rounds + 84 tiles x 3 pieces instead of 7 rounds).
// STCN_CONV_TAIL=0 switches the tail balancing off.
struct Plan { double cost; int big, splitk, rem_full, rem_split, rem_per; };

static Plan plan_variant(const ConvP &p, bool big, int force_splitk, size_t ws_floats) {
    static const bool tail_on = [] { const char *e = getenv("STCN_CONV_TAIL"); return !e || atoi(e) != 0; }();
    const int BM = narrow_variant(p) ? 128 : (big ? 128 : 64), BN = narrow_variant(p) ? 32 : (big ? 128 : 64);
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nkt = p.Kp / BK;
    const double mn = (double)p.M * p.N;
    const double tile_cost = (double)(BM * BN) / (64 * 64) * (big ? 0.92 : 1.0);
    Plan best{1e300, big ? 1 : 0, 1, 0, 0, 0};
    const int smax = nkt / 4 < 1 ? 1 : (nkt / 4 > 32 ? 32 : nkt / 4);
    for (int s = 1; s <= smax; ++s) {
        if (force_splitk > 0 && s != force_splitk && !(force_splitk > smax && s == smax)) continue;
        const int per = (nkt + s - 1) / s;
        if ((long)per * (s - 1) >= nkt) continue;  // an empty split
        if (s > 1 && (size_t)s * p.M * p.N > ws_floats) continue;
        const long rounds = (tiles * s + 255) / 256;
        // a workgroup alone on its CU has nobody to hide its load / barrier latencies behind (measured: +20 %)
        const double lone = tiles * s <= 256 ? 1.2 : (tiles * s <= 512 ? 1.04 : 1.0);
        double cost = (double)rounds * (per + 3.0) * tile_cost * lone;
        if (s > 1) cost += 4.6 + (s + 1) * 2.3e-6 * mn;
        if (cost < best.cost) best = Plan{cost, big ? 1 : 0, s, 0, 0, 0};
    }
    const long full_rounds = tiles / 256, n_rem = tiles - full_rounds * 256;
    if (tail_on && force_splitk <= 0 && full_rounds >= 1 && n_rem > 0 && nkt >= 8) {
        for (int sr = 2; sr <= 8 && sr <= nkt / 4; ++sr) {
            const int per = (nkt + sr - 1) / sr;
            if ((long)per * (sr - 1) >= nkt) continue;
            const long pieces = n_rem * sr;
            if ((size_t)pieces * BM * BN > ws_floats) continue;
            const double cost = ((double)full_rounds * (nkt + 3.0) + (double)((pieces + 255) / 256) * (per + 3.0)) * tile_cost +
                                4.6 + (sr + 1) * 2.3e-6 * (double)n_rem * BM * BN;
            if (cost < best.cost) best = Plan{cost, big ? 1 : 0, 1, (int)(full_rounds * 256), sr, per};
        }
    }
    return best;
}

void conv_plan(ConvP &p, int force_splitk, size_t ws_floats) {
    static const int big_mode = [] { const char *e = getenv("STCN_CONV_BIG"); return e ? atoi(e) : 1; }();
    Plan pl = plan_variant(p, false, force_splitk, ws_floats);
    constexpr int big_mink = 2304;               // smallest padded K for the 128x128 instance (1x1 convs measured no gain from it)
    const bool big_ok = big_mode != 0 && !narrow_variant(p) && !smallc_variant(p) && p.N >= 128 && p.Kp >= big_mink;
    if (big_ok) {
        const Plan pb = plan_variant(p, true, force_splitk, ws_floats);
        if (pb.cost < pl.cost || big_mode >= 2) pl = pb;
    }
    p.panel = 4;                                 // n-tiles per panel of the tile walk (tile_to_mn)
    p.tile_big = pl.big; p.splitk = pl.splitk;
    p.rem_full = pl.rem_full; p.rem_split = pl.rem_split; p.rem_per = pl.rem_per;
    p.chain = pw_chain_tiles(p, force_splitk);
    if (p.chain) { p.tile_big = 0; p.splitk = 1; p.rem_full = p.rem_split = p.rem_per = 0; }      // whole tiles, walked nt at a time
    static const bool dbg = getenv("STCN_CONV_PLAN_DEBUG") != nullptr;
    if (dbg)
        fprintf(stderr, "conv_plan M=%d N=%d K=%d: %s splitk=%d tail=(%d full, %d pieces of %d) cost %.0f\n", p.M, p.N, p.Kp,
                pl.big ? "128x128" : (narrow_variant(p) ? "128x32" : "64x64"), pl.splitk, pl.rem_full, pl.rem_split, pl.rem_per, pl.cost);
}

// dynamic LDS above 64 KB has to be opted into once per (device, kernel function): a process may hold models on several
// devices and drive them from several host threads (lanes)
void allow_big_lds(const void *kernel, size_t lds) {
    if (lds <= 64 * 1024) return;
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.insert({dev, kernel}).second)
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// which instance conv_launch takes for this (planned) conv - reported to the tests through stcn_last_conv_path()
const char *conv_variant_name(const ConvP &p) {
    const bool narrow = narrow_variant(p), big = p.tile_big != 0, smallc = smallc_variant(p);
    if (p.chain) return "direct_pointwise_chain";
    if (p.pointwise && !big && !narrow && !smallc) return "direct_pointwise";
    if (big) return "direct_big";
    if (narrow) return smallc ? "direct_narrow_smallc" : "direct_narrow";
    return smallc ? "direct_smallc" : "direct";
}

void conv_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev_gemm, hipEvent_t *ev_red) {
    const bool narrow = narrow_variant(p), big = p.tile_big != 0;
    const int BM = narrow ? 128 : (big ? 128 : 64), BN = narrow ? 32 : (big ? 128 : 64);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ntile = tiles_m * tiles_n;
    const int nkt = p.Kp / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    const bool tail = p.rem_split > 1;
    const dim3 grid(tail ? p.rem_full + (ntile - p.rem_full) * p.rem_split : ntile * p.splitk);
    const bool smallc = smallc_variant(p);
    const int pn = p.panel;
    const TileDiv td{fastdiv_make((unsigned)ntile), fastdiv_make((unsigned)tiles_n), fastdiv_make((unsigned)(tiles_m * (pn > 0 ? pn : 1))), tiles_m};
    hipEvent_t e0 = ev_gemm ? ev_gemm[0] : nullptr, e1 = ev_gemm ? ev_gemm[1] : nullptr;
    if (p.chain) {                                        // pointwise chain: nt consecutive tiles per workgroup, one resident set of workgroups
        const dim3 cgrid((unsigned)((ntile + p.chain - 1) / p.chain));
        if (p.relu_in) {
            if (e0) hipExtLaunchKernelGGL(pw_chain_kernel<true>, cgrid, dim3(256), lds, s, e0, e1, 0, p, tiles_n, ntile, p.chain, td);
            else hipLaunchKernelGGL(pw_chain_kernel<true>, cgrid, dim3(256), lds, s, p, tiles_n, ntile, p.chain, td);
        } else {
            if (e0) hipExtLaunchKernelGGL(pw_chain_kernel<false>, cgrid, dim3(256), lds, s, e0, e1, 0, p, tiles_n, ntile, p.chain, td);
            else hipLaunchKernelGGL(pw_chain_kernel<false>, cgrid, dim3(256), lds, s, p, tiles_n, ntile, p.chain, td);
        }
        return;
    }
    {
#define STCN_LAUNCH(WM_, WN_, RM_, RN_, SC_, RL_, ...)                                                                  \
    do {                                                                                                                 \
        auto kfn = conv_gemm_kernel<WM_, WN_, RM_, RN_, SC_, RL_, ##__VA_ARGS__>;                                        \
        allow_big_lds(reinterpret_cast<const void *>(kfn), lds);                                                         \
        if (e0) hipExtLaunchKernelGGL(kfn, grid, dim3(256), lds, s, e0, e1, 0, p, tiles_n, ntile, per, td);              \
        else hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, p, tiles_n, ntile, per, td);                               \
    } while (0)
    const bool pw = p.pointwise && !big && !narrow && !smallc;
    const int key = (pw ? 16 : 0) | (big ? 8 : 0) | (narrow ? 4 : 0) | (smallc ? 2 : 0) | (p.relu_in ? 1 : 0);
    switch (key) {
        case 16: STCN_LAUNCH(2, 2, 1, 1, false, false, true); break;
        case 17: STCN_LAUNCH(2, 2, 1, 1, false, true, true); break;
        case 0: STCN_LAUNCH(2, 2, 1, 1, false, false); break;
        case 1: STCN_LAUNCH(2, 2, 1, 1, false, true); break;
        case 2: STCN_LAUNCH(2, 2, 1, 1, true, false); break;
        case 3: STCN_LAUNCH(2, 2, 1, 1, true, true); break;
        case 4: STCN_LAUNCH(4, 1, 1, 1, false, false); break;
        case 5: STCN_LAUNCH(4, 1, 1, 1, false, true); break;
        case 6: STCN_LAUNCH(4, 1, 1, 1, true, false); break;
        case 7: STCN_LAUNCH(4, 1, 1, 1, true, true); break;
        case 8: STCN_LAUNCH(2, 2, 2, 2, false, false); break;
        default: STCN_LAUNCH(2, 2, 2, 2, false, true); break;
    }
    }
#undef STCN_LAUNCH
    if (tail) {
        const unsigned blocks = (unsigned)((ntile - p.rem_full) * (BM * BN / 1024));
        if (ev_red)
            hipExtLaunchKernelGGL(conv_reduce_tiles_kernel, dim3(blocks), dim3(256), 0, s, ev_red[0], ev_red[1], 0, p, tiles_n, ntile, BM, BN, td);
        else
            hipLaunchKernelGGL(conv_reduce_tiles_kernel, dim3(blocks), dim3(256), 0, s, p, tiles_n, ntile, BM, BN, td);
    } else if (p.splitk > 1) {
        conv_reduce_launch(p, s, ev_red);
    }
}

void conv_reduce_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev_red) {
    const long total4 = (long)p.M * p.N / 4;
    long blocks = (total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (ev_red)
        hipExtLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ev_red[0], ev_red[1], 0, p);
    else
        hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// Cout == 1 convolution: LPP lanes cooperate on one output pixel (4 channels per lane per step),
// weights [KH*KW*C] are read through L1/L2 (tiny), the input through L2 (3x3 neighbourhood reuse).
template <int LPP>
__global__ __launch_bounds__(256) void conv_n1_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      float bias, float *__restrict__ y, int B, int H, int W,
                                                      int C, int KH, int relu_in) {
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP, pl = lane / LPP;
    const long gw = (blockIdx.x * 256L + threadIdx.x) >> 6;
    const long npix = (long)B * H * W;
    const long pixel = gw * PPW + pl;
    const bool valid = pixel < npix;
    const long pp = valid ? pixel : 0;
    const int b = (int)(pp / ((long)H * W));
    const int rem = (int)(pp - (long)b * H * W);
    const int oh = rem / W, ow = rem - oh * W;
    const int pad = KH / 2;
    float acc = 0.f;
    for (int kh = 0; kh < KH; ++kh) {
        const int ih = oh + kh - pad;
        for (int kw = 0; kw < KH; ++kw) {
            const int iw = ow + kw - pad;
            const bool inb = valid && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            const float *xp = x + (((long)b * H + ih) * W + iw) * C;
            const float *wp = w + (kh * KH + kw) * C;
            for (int c = sub * 4; c < C; c += LPP * 4) {
                if (inb) {
                    f32x4 v = *reinterpret_cast<const f32x4 *>(xp + c);
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(wp + c);
                    if (relu_in) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    acc += v.x * u.x + v.y * u.y + v.z * u.z + v.w * u.w;
                }
            }
        }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (valid && sub == 0) y[pixel] = acc + bias;
}

// Cout == 1, 3x3 (decoder.pred: C = 256; FusionNet.final_conv: C = 32).  LPP = C / 4 lanes hold the channels of a pixel (4 each); a
// group of LPP lanes computes 8 consecutive pixels of a row: it loads its 3 x 10 input pixels ONCE (16 B per lane each) and keeps
// its 36 weights in registers; a wave holds 64 / LPP such groups side by side (64 pixels of the row at C = 32), the 4 waves of a
// workgroup take 4 consecutive rows of the same column strip, so two of a wave's three input rows are its neighbours' too (L1).
// The first versions (one wave / one 8-lane group per pixel, 9 taps fetched per output) pulled 4.3x the input through L2; a
// `if (relu_in)` branch behind every load then made hipcc drain vmcnt per load (30 serialized round trips per strip): ReLU is
// max(v, lo) with lo = 0 or -inf now.
template <int C>
__global__ __launch_bounds__(256) void conv_n1_strip_kernel(const float *__restrict__ x, const float *__restrict__ w, float bias,
                                                            float *__restrict__ y, int B, int H, int W, int relu_in) {
    constexpr int PX = 8, LPP = C / 4, NS = 64 / LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sl = lane % LPP, sub = lane / LPP;
    const int strips = (W + PX * NS - 1) / (PX * NS), rows4 = (H + 3) / 4;
    // XCD-contiguous: a workgroup's 6 input rows x 10 columns overlap its neighbours' (halo 1.9x); dealt round-robin the neighbours sit
    // on other XCDs and every L2 fetches its own copy of the halo (decoder.pred: 301 MB fetched for a 133 MB input)
    const int bid = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x);
    const int sx = bid % strips;
    const int ry = (bid / strips) % rows4, b = bid / (strips * rows4);
    const int oy = ry * 4 + wave, ox0 = (sx * NS + sub) * PX;
    if (oy >= H) return;
    const float lo = relu_in ? 0.f : -__builtin_inff();
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4 *>(w + t * C + 4 * sl);
    float acc[PX];
#pragma unroll
    for (int i = 0; i < PX; ++i) acc[i] = 0.f;
    const float *xb = x + (long)b * H * W * C + 4 * sl;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy + kh - 1;
        if ((unsigned)iy >= (unsigned)H) continue;                     // wave-uniform
        f32x4 v[PX + 2];
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            const int ix = ox0 + j - 1;
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)ix < (unsigned)W) u = *reinterpret_cast<const f32x4 *>(xb + ((long)iy * W + ix) * C);
            u.x = fmaxf(u.x, lo); u.y = fmaxf(u.y, lo); u.z = fmaxf(u.z, lo); u.w = fmaxf(u.w, lo);
            v[j] = u;
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const f32x4 g = wt[kh * 3 + kw];
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const f32x4 u = v[i + kw];
                acc[i] += u.x * g.x + u.y * g.y + u.z * g.z + u.w * g.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PX; ++i)
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
    if (sl < PX && ox0 + sl < W) {
        float r = acc[0];
#pragma unroll
        for (int i = 1; i < PX; ++i) r = sl == i ? acc[i] : r;
        y[((long)b * H + oy) * W + ox0 + sl] = r + bias;
    }
}

void conv_n1_launch(const float *x, const float *w, float bias, float *y, int B, int H, int W, int C,
                    int KH, int relu_in, hipStream_t s) {
    const long npix = (long)B * H * W;
    if (C == 256 && KH == 3) {
        const unsigned blocks = (unsigned)((long)B * ((H + 3) / 4) * ((W + 7) / 8));
        hipLaunchKernelGGL(conv_n1_strip_kernel<256>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, B, H, W, relu_in);
        return;
    }
    if (C == 32 && KH == 3) {
        const unsigned blocks = (unsigned)((long)B * ((H + 3) / 4) * ((W + 63) / 64));
        hipLaunchKernelGGL(conv_n1_strip_kernel<32>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, B, H, W, relu_in);
        return;
    }
    if (C >= 256) {
        const long waves = npix;
        hipLaunchKernelGGL((conv_n1_kernel<64>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, w, bias, y,
                           B, H, W, C, KH, relu_in);
    } else {
        const long waves = (npix + 7) / 8;
        hipLaunchKernelGGL((conv_n1_kernel<8>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, w, bias, y,
                           B, H, W, C, KH, relu_in);
    }
}

}  // namespace stcn
