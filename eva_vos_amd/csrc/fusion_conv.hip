// fusion_conv.hip - the 3x3 convolutions of FusionNet (reference mivos/model/fusion_net.py:12-30,38-48) on gfx950.
//
// Rounds >= 2 of an annotation session fuse every frame between two interacted frames (inference_core.py:184-207): five
// 3x3 convs with 32 output channels at FULL resolution (480x864: M = 414 720 rows, N = 32, K = 288 / 108) per fused frame and
// object - 33 GFLOP, a third of what the decoder of that frame costs.  As implicit GEMMs of the generic kernel (128x32 tiles, K
// tile 32: 9 barriers and 9 im2col re-stagings of the same pixels per tile) they ran far below the matrix rate.  Shape-specific
// structure instead:
//   * the WHOLE weight matrix lives in registers: the B operand of v_mfma_f32_32x32x2_f32 is one VGPR per MFMA, K = 288 means
//     144 VGPRs per lane hold every B fragment of the layer (loaded once per workgroup, 36 x 16 B per lane);
//   * the input patch of the workgroup (PH output rows x 32 columns + halo) is staged ONCE into LDS in full 128-byte pixel
//     lines (coalesced), pixel stride padded to CINP + 4 floats so that the A fragments (lane = pixel, lane half = 4-channel
//     group, one ds_read_b128 per 4 MFMAs, k-permuted exactly like the weights) are bank-conflict free; the nine taps are
//     nine shifted reads of the same patch: no im2col copy, one barrier per workgroup;
//   * a wave owns PH / 4 output rows of 32 pixels: 144 MFMAs per row back to back, nothing but ds_reads between them;
//   * persistent workgroups, the next patch's global loads in flight under the MFMAs of the current one;
//   * bias / residual / ReLU in the epilogue, 128-byte stores.
// CIN = 12 (conv1: 9 channels padded to 12 in HBM) runs as 16 channels per tap in LDS / registers (zero filled).
#include <hip/hip_ext.h>

#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CIN, int PH, bool RELU_OUT, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void fusion_conv_kernel(const float *__restrict__ x, const float *__restrict__ w, int Kp,
                                                             const float *__restrict__ bias, const float *__restrict__ res,
                                                             float *__restrict__ y, int H, int W, int tiles_x, int n_patches) {
    constexpr int CINP = (CIN + 7) / 8 * 8, KB = CINP / 8, LP = CINP + 4;      // channels per tap in LDS, k-blocks, pixel stride
    constexpr int PW = 32, RB = PH / 4;                                       // patch columns, output rows per wave
    constexpr int CH = CIN / 4;                                               // 16-byte chunks per pixel in HBM
    extern __shared__ __attribute__((aligned(16))) float patch[];            // [2][(PH + 2)][(PW + 2)][LP]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;

    // PERSISTENT workgroups (two per CU): a workgroup walks the patches blockIdx.x, + gridDim.x, ...  The first version (one
    // workgroup per 8-row patch) staged, then computed, and lost the tail of its 3.2 rounds: 64 TFLOP/s, slower than the generic
    // kernel's 82.  Here the global loads of the NEXT patch are issued before the MFMAs of this one (registers), the weights are
    // loaded once per workgroup, and 4-row patches (3240 at 480x864) leave a tail of 6.3 -> 7 rounds instead of 3.2 -> 4:
    // 95 TFLOP/s on the 32 -> 32 layers (the MFMA loop alone, without any global access: 108 - the clock the chip holds).
    // staging: thread t moves the 16-byte part (t % CHP) of the pixels t / CHP + (256 / CHP) i of the patch.  Everything per
    // chunk (pixel row / column, validity, addresses) is recomputed per patch from two registers - hoisted out of the patch
    // loop it cost 30 VGPRs and spilled (the weights alone take 144).
    constexpr int CHP = CH > 4 ? 8 : 4, PPI = 256 / CHP;                       // lanes per pixel (power of two), pixels per pass
    constexpr int NPIX = (PH + 2) * (PW + 2), NIT = (NPIX + PPI - 1) / PPI;
    f32x4 st[NIT];
    int p0 = t / CHP;
    const int part = t % CHP;
    auto gload = [&](int pidx) {                                              // out-of-image pixels are zero (the conv's padding)
        const int by = pidx / tiles_x, bx = pidx - by * tiles_x;
        const int x0 = bx * PW, y0 = by * PH;
        asm volatile("" : "+v"(p0));                                          // keeps the per-chunk index math inside the loop
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int p = p0 + PPI * i;
            const int py = p / (PW + 2), px = p - py * (PW + 2);
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (p < NPIX && part < CH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                v = *reinterpret_cast<const f32x4 *>(x + ((long)gy * W + gx) * CIN + 4 * part);
            st[i] = v;
        }
    };
    constexpr int PBUF = NPIX * LP;                                           // floats per patch buffer
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int p = p0 + PPI * i;
            if (p < NPIX && part < CH) *reinterpret_cast<f32x4 *>(patch + buf * PBUF + p * LP + 4 * part) = st[i];
        }
    };
    int pidx = blockIdx.x;
    gload(pidx);
    // ---- the layer's weights -> registers: wr[tap][kb] = W[n = l31][tap * CIN + 8 kb + 4 h .. + 3] (zero beyond CIN)
    f32x4 wr[9][KB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int c0 = 8 * kb + 4 * h;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c0 < CIN) v = *reinterpret_cast<const f32x4 *>(w + (long)l31 * Kp + tap * CIN + c0);
            wr[tap][kb] = v;
        }
    const float bv = bias[l31];
    if (CINP != CIN) {                                                        // the zero channels CIN .. CINP - 1 of every pixel (once)
        for (int p = t; p < 2 * NPIX; p += 256) *reinterpret_cast<f32x4 *>(patch + p * LP + CIN) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    sstore(0);
    const float *pa0 = patch + ((wave * RB) * (PW + 2) + l31) * LP + 4 * h;
    // Two patch buffers, ONE barrier per patch: the next patch's global loads are issued at the top, written to the other
    // buffer in the MIDDLE of this patch's MFMA stream (half a patch of matrix time after the loads: they have landed), and the
    // barrier at the top of the next iteration finds both the stores and every reader of the old buffer done.  (Single buffer:
    // barrier - compute - barrier - store, and the two co-resident workgroups of a CU, sharing the pipes fairly, arrive at those
    // phases together.)
    int buf = 0;
    for (; pidx < n_patches; pidx += gridDim.x, buf ^= 1) {
        __syncthreads();                                                      // patch `buf` is in LDS, nobody reads `buf ^ 1` any more
        const float *pa = pa0 + buf * PBUF;
        const int nxt = pidx + gridDim.x;
        if (nxt < n_patches) gload(nxt);                                              // lands while the MFMAs below run
        __builtin_amdgcn_sched_barrier(0);
        // ---- RB output rows of 32 pixels per wave; acc rows = pixels, columns = output channels
        f32x16 acc[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        // 9 taps x KB k-blocks = NG fragment groups of 4 MFMAs per row; the fragments of group g + 2 are requested before the
        // MFMAs of group g (three register sets, pinned: left alone hipcc issues each ds_read right in front of its first use
        // and every group pays the LDS latency: 0.61 of the matrix rate)
        constexpr int NG = 9 * KB;
        f32x4 fa[3][RB];
        auto fread = [&](int g2, f32x4 (&dst)[RB]) {
            const int tap = g2 / KB, kb = g2 - tap * KB, dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
            for (int b = 0; b < RB; ++b) dst[b] = *reinterpret_cast<const f32x4 *>(pa + ((b + dy) * (PW + 2) + dx) * LP + 8 * kb);
        };
        fread(0, fa[0]);
        fread(1, fa[1]);
#pragma unroll
        for (int g2 = 0; g2 < NG; ++g2) {
            if (g2 + 2 < NG) fread(g2 + 2, fa[(g2 + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (g2 == NG / 2 && nxt < n_patches) {
                sstore(buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int tap = g2 / KB, kb = g2 - tap * KB;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < RB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g2 % 3][b][j], wr[tap][kb][j], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: C/D layout column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 h
        const int by = pidx / tiles_x, bx = pidx - by * tiles_x;
        const int x0 = bx * PW, y0 = by * PH;
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const int oy = y0 + wave * RB + b;
            if (oy >= H) continue;
            const long rowo = ((long)oy * W + x0) * 32 + l31;
            float rv[16];                                                     // residual: all 16 loads in flight before the first use
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = min((r & 3) + 8 * (r >> 2) + 4 * h, W - 1 - x0);
                rv[r] = HAS_RES ? res[rowo + px * 32] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                rv[r] += acc[b][r] + bv;
                if (RELU_OUT) rv[r] = fmaxf(rv[r], 0.f);
                asm volatile("" : "+v"(rv[r]));                                 // pinned in front of the masked stores
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (x0 + px < W) y[rowo + px * 32] = rv[r];
            }
        }
    }
}

// conv1 (9 -> 12 channels in HBM, 16 per tap here: a quarter of its MFMAs multiply zeros) measured 53 us on this kernel, 49 us
// on the generic one: off by default (STCN_FUSION_CONV12=1 switches it on; the instance stays tested)
static bool fusion_conv12() {
    const char *e = getenv("STCN_FUSION_CONV12");
    return e && atoi(e) != 0;
}

bool fusion_conv_eligible(const ConvP &p) {
    static const bool on = [] { const char *e = getenv("STCN_FUSION_CONV"); return !e || atoi(e) != 0; }();
    return on && p.N == 32 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.B == 1 && !p.x1 && !p.relu_in && (p.Cin == 32 || (p.Cin == 12 && fusion_conv12())) &&
           p.K == 9 * p.Cin && (p.y_bs == 0) && (!p.res || !p.res_bmod) && p.bias;
}

void fusion_conv_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev) {
    constexpr int PH = 4;
    const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + PH - 1) / PH, n_patches = tiles_x * tiles_y;
    static const int resident = [] {                  // two workgroups per CU (206 VGPRs: two waves per SIMD)
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return 2 * (cus > 0 ? cus : 256);
    }();
    const dim3 grid((unsigned)(n_patches < resident ? n_patches : resident));
    const int cinp = (p.Cin + 7) / 8 * 8;
    const size_t lds = (size_t)2 * (PH + 2) * 34 * (cinp + 4) * sizeof(float);
#define STCN_FC(CIN_, RL_, RS_)                                                                                                   \
    do {                                                                                                                          \
        auto kfn = fusion_conv_kernel<CIN_, PH, RL_, RS_>;                                                                        \
        if (ev) hipExtLaunchKernelGGL(kfn, grid, dim3(256), lds, s, ev[0], ev[1], 0, p.x0, p.w, p.Kp, p.bias, p.res, p.y, p.H, p.W, tiles_x, n_patches); \
        else hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, p.x0, p.w, p.Kp, p.bias, p.res, p.y, p.H, p.W, tiles_x, n_patches); \
    } while (0)
    const int key = (p.Cin == 12 ? 4 : 0) | (p.relu_out ? 2 : 0) | (p.res ? 1 : 0);
    switch (key) {
        case 0: STCN_FC(32, false, false); break;
        case 1: STCN_FC(32, false, true); break;
        case 2: STCN_FC(32, true, false); break;
        case 3: STCN_FC(32, true, true); break;
        case 4: STCN_FC(12, false, false); break;
        case 5: STCN_FC(12, false, true); break;
        case 6: STCN_FC(12, true, false); break;
        default: STCN_FC(12, true, true); break;
    }
#undef STCN_FC
}

}  // namespace stcn
