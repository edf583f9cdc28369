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
                                                             float *__restrict__ y, int H, int W, int tiles_x) {
    constexpr int CINP = (CIN + 7) / 8 * 8, KB = CINP / 8, LP = CINP + 4;      // channels per tap in LDS, k-blocks, pixel stride
    constexpr int PW = 32, RB = PH / 4;                                       // patch columns, output rows per wave
    constexpr int CH = CIN / 4;                                               // 16-byte chunks per pixel in HBM
    extern __shared__ __attribute__((aligned(16))) float patch[];            // [(PH + 2)][(PW + 2)][LP]
    const int nblk = gridDim.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7;            // XCD-contiguous: an XCD's L2 sees a band of rows
    const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int by = bid / tiles_x, bx = bid - by * tiles_x;
    const int x0 = bx * PW, y0 = by * PH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;

    // ---- stage the patch: chunk c -> (pixel, 16-byte part); out-of-image pixels are zero (the conv's padding)
    constexpr int NPIX = (PH + 2) * (PW + 2), NCHUNK = NPIX * CH, NIT = (NCHUNK + 255) / 256;
    f32x4 st[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = t + 256 * i;
        const int p = c / CH, part = c - p * CH;
        const int py = p / (PW + 2), px = p - py * (PW + 2);
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < NCHUNK && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
            v = *reinterpret_cast<const f32x4 *>(x + ((long)gy * W + gx) * CIN + 4 * part);
        st[i] = v;
    }
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
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int c = t + 256 * i;
        const int p = c / CH, part = c - p * CH;
        if (c < NCHUNK) *reinterpret_cast<f32x4 *>(patch + p * LP + 4 * part) = st[i];
    }
    if (CINP != CIN) {                                                        // the zero channels CIN .. CINP - 1 of every pixel
        for (int p = t; p < NPIX; p += 256) *reinterpret_cast<f32x4 *>(patch + p * LP + CIN) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // ---- RB output rows of 32 pixels per wave; acc rows = pixels, columns = output channels
    f32x16 acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const float *pa = patch + ((wave * RB) * (PW + 2) + l31) * LP + 4 * h;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 a[RB];
#pragma unroll
            for (int b = 0; b < RB; ++b) a[b] = *reinterpret_cast<const f32x4 *>(pa + ((b + dy) * (PW + 2) + dx) * LP + 8 * kb);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][j], wr[tap][kb][j], acc[b], 0, 0, 0);
        }
    }

    // ---- epilogue: C/D layout column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 h
    const float bv = bias[l31];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        const int oy = y0 + wave * RB + b;
        if (oy >= H) continue;
        const long rowo = ((long)oy * W + x0) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
            if (x0 + px >= W) continue;
            float v = acc[b][r] + bv;
            if (HAS_RES) v += res[rowo + px * 32];
            if (RELU_OUT) v = fmaxf(v, 0.f);
            y[rowo + px * 32] = v;
        }
    }
}

bool fusion_conv_eligible(const ConvP &p) {
    static const bool on = [] { const char *e = getenv("STCN_FUSION_CONV"); return !e || atoi(e) != 0; }();
    return on && p.N == 32 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.B == 1 && !p.x1 && !p.relu_in && (p.Cin == 32 || p.Cin == 12) &&
           p.K == 9 * p.Cin && (p.y_bs == 0) && (!p.res || !p.res_bmod) && p.bias;
}

void fusion_conv_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev) {
    constexpr int PH = 8;
    const int tiles_x = (p.W + 31) / 32, tiles_y = (p.H + PH - 1) / PH;
    const dim3 grid((unsigned)(tiles_x * tiles_y));
    const int cinp = (p.Cin + 7) / 8 * 8;
    const size_t lds = (size_t)(PH + 2) * 34 * (cinp + 4) * sizeof(float);
#define STCN_FC(CIN_, RL_, RS_)                                                                                                   \
    do {                                                                                                                          \
        auto kfn = fusion_conv_kernel<CIN_, PH, RL_, RS_>;                                                                        \
        if (ev) hipExtLaunchKernelGGL(kfn, grid, dim3(256), lds, s, ev[0], ev[1], 0, p.x0, p.w, p.Kp, p.bias, p.res, p.y, p.H, p.W, tiles_x); \
        else hipLaunchKernelGGL(kfn, grid, dim3(256), lds, s, p.x0, p.w, p.Kp, p.bias, p.res, p.y, p.H, p.W, tiles_x);            \
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
