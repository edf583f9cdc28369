// metrics.hip - J (region IoU) and F (boundary measure) counts on the GPU, integer-exact.
// Reference: interactions/metrics.py:24-34 (get_j_and_f), :38-97 (_seg2bmap at equal size), :100-160 (f_measure:
// disk(ceil(0.008*|shape|)) dilation of each boundary map, matches = boundary & dilated other boundary).
// HBM-bound stencil work: one pass builds both 1-pixel boundary maps, a second pass visits the pixels and, ONLY
// at boundary pixels (a few thousand per frame), scans the disk window of the other map; six integer counters per
// frame are accumulated with integer atomics (exact, order-independent).
#include <algorithm>

#include "kernels.h"

namespace stcn {

void jf_counts_launch(const uint8_t *gt, const uint8_t *pred, int T, int H, int W, int radius, uint8_t *bmap, int *counts, hipStream_t s);

// Counting: a thread visits PPT pixels (a wave 64 consecutive pixels per step: coalesced byte loads), keeps its counters in registers
// and the wave adds them up ONCE at the end - one integer atomic per wave and counter when the wave's 64 x PPT pixels lie inside one
// frame (all but the ~T waves that straddle a frame boundary, which fall back to one atomic per pixel).  Round 3 issued one atomic per
// wave and counter per 64 PIXELS: 1.7 M atomics on 6 T addresses made the two kernels 4.4 ms per 66-frame 480p clip - 15 % of an
// annotation round of the eval driver; a per-pixel atomicAdd before that 49 ms.
static constexpr int PPT = 16;

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// bit0 = gt boundary, bit1 = pred boundary.  j_only: intersection / union only (bmap untouched)
__global__ __launch_bounds__(256) void jf_boundary_kernel(const uint8_t *__restrict__ gt, const uint8_t *__restrict__ pr, int T, int H, int W,
                                                          uint8_t *__restrict__ bmap, int *__restrict__ counts, int j_only) {
    const long hw = (long)H * W, n = T * hw;
    const int lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64L * PPT);          // first pixel of this wave
    if (base >= n) return;
    const long last = min(base + 64L * PPT, n) - 1;
    const int t0 = (int)(base / hw);
    const bool one_frame = (int)(last / hw) == t0;                                       // wave-uniform
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int j = 0; j < PPT; ++j) {
        const long i = base + j * 64L + lane;
        if (i >= n) break;
        const int t = one_frame ? t0 : (int)(i / hw);
        const int rem = (int)(i - t * hw);
        const int g = gt[i] != 0, p = pr[i] != 0;
        int bg = 0, bp = 0;
        if (!j_only) {
            const int y = rem / W, x = rem - y * W;
            auto bnd = [&](const uint8_t *seg, int s) -> int {
                const uint8_t *q = seg + i;
                if (y < H - 1 && x < W - 1) {
                    const int e = q[1] != 0, so = q[W] != 0, se = q[W + 1] != 0;
                    return (s ^ e) | (s ^ so) | (s ^ se);
                }
                if (y == H - 1 && x < W - 1) return s ^ (q[1] != 0);
                if (x == W - 1 && y < H - 1) return s ^ (q[W] != 0);
                return 0;
            };
            bg = bnd(gt, g);
            bp = bnd(pr, p);
            bmap[i] = (uint8_t)(bg | (bp << 1));
        }
        if (one_frame) { c0 += g & p; c1 += g | p; c2 += bg; c3 += bp; }
        else {                                                                           // a wave across a frame boundary: per pixel
            int *c = counts + t * 6;
            if (g & p) atomicAdd(&c[0], 1);
            if (g | p) atomicAdd(&c[1], 1);
            if (bg) atomicAdd(&c[2], 1);
            if (bp) atomicAdd(&c[3], 1);
        }
    }
    if (one_frame) {
        c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2); c3 = wave_sum(c3);
        int *c = counts + t0 * 6;          // inter, union, n_gt_b, n_fg_b, gt_match, fg_match
        if (lane == 0) {
            if (c0) atomicAdd(&c[0], c0);
            if (c1) atomicAdd(&c[1], c1);
            if (c2) atomicAdd(&c[2], c2);
            if (c3) atomicAdd(&c[3], c3);
        }
    }
}

__global__ __launch_bounds__(256) void jf_match_kernel(const uint8_t *__restrict__ bmap, int T, int H, int W, int r, int *__restrict__ counts) {
    const long hw = (long)H * W, n = T * hw;
    const int lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64L * PPT);
    if (base >= n) return;
    const long last = min(base + 64L * PPT, n) - 1;
    const int t0 = (int)(base / hw);
    const bool one_frame = (int)(last / hw) == t0;
    int c4 = 0, c5 = 0;
    for (int j = 0; j < PPT; ++j) {
        const long i = base + j * 64L + lane;
        if (i >= n) break;
        const int me = bmap[i];
        if (!me) continue;                                    // only boundary pixels (a few thousand per frame) scan the disk
        const int t = one_frame ? t0 : (int)(i / hw);
        const int rem = (int)(i - t * hw);
        const int y = rem / W, x = rem - y * W;
        const uint8_t *b = bmap + (long)t * hw;
        const int want = ((me & 1) ? 2 : 0) | ((me & 2) ? 1 : 0);
        int other = 0;                                        // bits of the OTHER map found inside the disk
        // rows from the centre outwards (0, -1, +1, -2, ...): the two boundaries usually run close to each other, the scan ends early
        for (int k = 0; k <= 2 * r && (other & want) != want; ++k) {
            const int dy = (k & 1) ? -((k + 1) >> 1) : (k >> 1);
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            int hx = 0;                                       // half width of the disk at this row: largest dx with dx^2 + dy^2 <= r^2
            while ((hx + 1) * (hx + 1) + dy * dy <= r * r) ++hx;
            const int x0 = max(x - hx, 0), x1 = min(x + hx, W - 1);
            const uint8_t *row = b + (long)yy * W;
            for (int xx = x0; xx <= x1; ++xx) other |= row[xx];
        }
        const int m4 = (me & 1) && (other & 2), m5 = (me & 2) && (other & 1);
        if (one_frame) { c4 += m4; c5 += m5; }
        else {
            if (m4) atomicAdd(&counts[t * 6 + 4], 1);         // gt boundary pixel inside dilated pred boundary
            if (m5) atomicAdd(&counts[t * 6 + 5], 1);         // pred boundary pixel inside dilated gt boundary
        }
    }
    if (one_frame) {
        c4 = wave_sum(c4); c5 = wave_sum(c5);
        if (lane == 0) {
            if (c4) atomicAdd(&counts[t0 * 6 + 4], c4);
            if (c5) atomicAdd(&counts[t0 * 6 + 5], c5);
        }
    }
}

// ---- one annotation round on the device (round 6): compose -> counts -> quality + selection ------------------------------------------
// gen[t] = the engine's mask of frame t (cropped out of the padded [T][nh][nw] tensor, non-zero = object), or the ground truth where the frame
// is annotated (interactions/eval.py:57-60: annotated frames count with their GT mask).  gen is what util/fq_dataset.py:64-84 saves as a state.
__global__ __launch_bounds__(256) void round_compose_kernel(const uint8_t *__restrict__ masks, int nh, int nw, int lh, int lw,
                                                            const uint8_t *__restrict__ gt, const uint8_t *__restrict__ annotated, int T, int H, int W,
                                                            uint8_t *__restrict__ gen) {
    const long hw = (long)H * W, n = T * hw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i / hw);
        const int rem = (int)(i - t * hw);
        const int y = rem / W, x = rem - y * W;
        gen[i] = annotated[t] ? (uint8_t)(gt[i] != 0) : (uint8_t)(masks[((long)t * nh + y + lh) * nw + x + lw] != 0);
    }
}

// quality[t] in fp64 with the operations, and their order, of the host path (eva_vos_amd/metrics.py::_scores_from_counts, itself the
// reference's interactions/metrics.py:141-158 / eval.py:62-79): J = inter / union (0 when the union is empty), F = 2 p r / (p + r) with the
// reference's special cases, J&F = 0.5 (J + F); frames whose ground truth is empty get the NO_OBJECT token.  IEEE division / multiplication /
// addition are correctly rounded on the device as on the host, so the values - and therefore the arg-min (first index of the minimum, as
// numpy.argmin) - are bit-identical to the host path's.  One workgroup; T <= a few hundred.
__global__ __launch_bounds__(256) void round_quality_kernel(const int *__restrict__ counts, const uint8_t *__restrict__ noobj, int T, int j_only,
                                                            double no_object, double *__restrict__ quality, int *__restrict__ select) {
    __shared__ double sv[256];
    __shared__ int si[256];
    double best = __builtin_inf();
    int besti = 0x7fffffff;
    for (int t = threadIdx.x; t < T; t += 256) {
        const int *c = counts + t * 6;
        const double j = c[1] == 0 ? 0.0 : __ddiv_rn((double)c[0], (double)c[1]);
        double q = j;
        if (!j_only) {
            const int n_gt = c[2], n_fg = c[3];
            double p, r;
            if (n_fg == 0 && n_gt > 0) { p = 1.0; r = 0.0; }
            else if (n_fg > 0 && n_gt == 0) { p = 0.0; r = 1.0; }
            else if (n_fg == 0 && n_gt == 0) { p = 1.0; r = 1.0; }
            else { p = __ddiv_rn((double)c[5], (double)n_fg); r = __ddiv_rn((double)c[4], (double)n_gt); }
            const double s = __dadd_rn(p, r);
            const double f = s == 0.0 ? 0.0 : __ddiv_rn(__dmul_rn(__dmul_rn(2.0, p), r), s);
            q = __dmul_rn(0.5, __dadd_rn(j, f));
        }
        if (noobj[t]) q = no_object;
        quality[t] = q;
        if (q < best) { best = q; besti = t; }                               // ascending t per thread: the first minimum of its subsequence
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v = sv[threadIdx.x + o]; const int i2 = si[threadIdx.x + o];
            if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && i2 < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) select[0] = si[0];
}

// masks / gt / annotated / gen / counts point at the FIRST of the Tn frames to (re)compose and count; the quality and the arg-min cover all
// T_all frames of the clip, whose counts start Tn0 = (counts - counts_all) / 6 frames earlier: the caller passes counts of frame t0 and T_all,
// noobj and quality are whole-clip arrays
void round_score_launch(const uint8_t *masks, int nh, int nw, int lh, int lw, const uint8_t *gt, const uint8_t *annotated, const uint8_t *noobj,
                        int Tn, int H, int W, int radius, double no_object, uint8_t *gen, uint8_t *bmap, int *counts, int T_all, double *quality,
                        int *select, hipStream_t s, int t0) {
    const long n = (long)Tn * H * W;
    const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(round_compose_kernel, dim3(blocks), dim3(256), 0, s, masks, nh, nw, lh, lw, gt, annotated, Tn, H, W, gen);
    jf_counts_launch(gt, gen, Tn, H, W, radius, bmap, counts, s);
    hipLaunchKernelGGL(round_quality_kernel, dim3(1), dim3(256), 0, s, counts - (long)t0 * 6, noobj, T_all, radius < 0 ? 1 : 0, no_object, quality, select);
}

void jf_counts_launch(const uint8_t *gt, const uint8_t *pred, int T, int H, int W, int radius, uint8_t *bmap,
                      int *counts, hipStream_t s) {
    const long n = (long)T * H * W;
    const unsigned blocks = (unsigned)((n + 256L * PPT - 1) / (256L * PPT));
    (void)hipMemsetAsync(counts, 0, (size_t)T * 6 * sizeof(int), s);
    // radius < 0: J only (intersection / union; bmap may be null)
    hipLaunchKernelGGL(jf_boundary_kernel, dim3(blocks), dim3(256), 0, s, gt, pred, T, H, W, bmap, counts, radius < 0 ? 1 : 0);
    if (radius >= 0) hipLaunchKernelGGL(jf_match_kernel, dim3(blocks), dim3(256), 0, s, bmap, T, H, W, radius, counts);
}

}  // namespace stcn
