// metrics.hip - J (region IoU) and F (boundary measure) counts on the GPU, integer-exact.
// Reference: interactions/metrics.py:24-34 (get_j_and_f), :38-97 (_seg2bmap at equal size), :100-160 (f_measure:
// disk(ceil(0.008*|shape|)) dilation of each boundary map, matches = boundary & dilated other boundary).
// HBM-bound stencil work: one pass builds both 1-pixel boundary maps, a second pass visits the pixels and, ONLY
// at boundary pixels (a few thousand per frame), scans the disk window of the other map; six integer counters per
// frame are accumulated with integer atomics (exact, order-independent).
#include "kernels.h"

namespace stcn {

// bit0 = gt boundary, bit1 = pred boundary
__global__ void jf_boundary_kernel(const uint8_t *__restrict__ gt, const uint8_t *__restrict__ pr, int T, int H, int W,
                                   uint8_t *__restrict__ bmap, int *__restrict__ counts) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    const long hw = (long)H * W;
    if (i >= T * hw) return;
    const int t = (int)(i / hw);
    const int rem = (int)(i - t * hw);
    const int y = rem / W, x = rem - y * W;
    auto bnd = [&](const uint8_t *seg) -> int {
        const uint8_t *p = seg + (long)t * hw;
        const int s = p[rem] != 0;
        if (y < H - 1 && x < W - 1) {
            const int e = p[rem + 1] != 0, so = p[rem + W] != 0, se = p[rem + W + 1] != 0;
            return (s ^ e) | (s ^ so) | (s ^ se);
        }
        if (y == H - 1 && x < W - 1) return s ^ (p[rem + 1] != 0);
        if (x == W - 1 && y < H - 1) return s ^ (p[rem + W] != 0);
        return 0;
    };
    const int g = gt[i] != 0, p = pr[i] != 0;
    const int bg = bnd(gt), bp = bnd(pr);
    bmap[i] = (uint8_t)(bg | (bp << 1));
    int *c = counts + t * 6;          // inter, union, n_gt_b, n_fg_b, gt_match, fg_match
    if (g & p) atomicAdd(&c[0], 1);
    if (g | p) atomicAdd(&c[1], 1);
    if (bg) atomicAdd(&c[2], 1);
    if (bp) atomicAdd(&c[3], 1);
}

__global__ void jf_match_kernel(const uint8_t *__restrict__ bmap, int T, int H, int W, int r, int *__restrict__ counts) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    const long hw = (long)H * W;
    if (i >= T * hw) return;
    const int me = bmap[i];
    if (!me) return;
    const int t = (int)(i / hw);
    const int rem = (int)(i - t * hw);
    const int y = rem / W, x = rem - y * W;
    const uint8_t *b = bmap + (long)t * hw;
    int other = 0;                    // bits of the OTHER maps found inside the disk
    const int want = ((me & 1) ? 2 : 0) | ((me & 2) ? 1 : 0);
    for (int dy = -r; dy <= r && (other & want) != want; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int dx = -r; dx <= r; ++dx) {
            if (dx * dx + dy * dy > r * r) continue;
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)W) continue;
            other |= b[(long)yy * W + xx];
        }
    }
    int *c = counts + t * 6;
    if ((me & 1) && (other & 2)) atomicAdd(&c[4], 1);     // gt boundary pixel inside dilated pred boundary
    if ((me & 2) && (other & 1)) atomicAdd(&c[5], 1);     // pred boundary pixel inside dilated gt boundary
}

void jf_counts_launch(const uint8_t *gt, const uint8_t *pred, int T, int H, int W, int radius, uint8_t *bmap,
                      int *counts, hipStream_t s) {
    const long n = (long)T * H * W;
    (void)hipMemsetAsync(counts, 0, (size_t)T * 6 * sizeof(int), s);
    hipLaunchKernelGGL(jf_boundary_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gt, pred, T, H, W, bmap,
                       counts);
    hipLaunchKernelGGL(jf_match_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bmap, T, H, W, radius, counts);
}

}  // namespace stcn
