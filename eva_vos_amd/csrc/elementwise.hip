// elementwise.hip - HBM-bound helper kernels of the STCN path (NHWC fp32, 16-byte accesses).
// Reference semantics cited per kernel (paths under /root/reference/mivos).
#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline unsigned nblocks(long n, int per = 256, long cap = 1 << 20) {
    long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// image NCHW [3,H,W] -> NHWC4 [nh,nw,4] with symmetric zero pad (tensor_util.py:62-80)
__global__ void pack_image_kernel(const float *__restrict__ img, float *__restrict__ out, int H, int W,
                                  int nh, int nw, int lw, int lh) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)nh * nw) return;
    const int y = (int)(i / nw), x = (int)(i - (long)y * nw);
    const int sy = y - lh, sx = x - lw;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
        const long o = (long)sy * W + sx, pl = (long)H * W;
        v.x = img[o]; v.y = img[o + pl]; v.z = img[o + 2 * pl];
    }
    *reinterpret_cast<f32x4 *>(out + i * 4) = v;
}
void pack_image_launch(const float *img, float *out, int H, int W, int nh, int nw, int lw, int lh,
                       hipStream_t s) {
    hipLaunchKernelGGL(pack_image_kernel, dim3(nblocks((long)nh * nw)), dim3(256), 0, s, img, out, H, W, nh,
                       nw, lw, lh);
}

// value-encoder input (prop_net.py:157-169, modules.py:119): [k,npix,8] = rgb, mask_i, others_i, 0,0,0
__global__ void pack_value_input_kernel(const float *__restrict__ img4, const float *__restrict__ masks,
                                        long mstride, int k, int npix, float *__restrict__ out) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)k * npix) return;
    const int b = (int)(i / npix), pix = (int)(i - (long)b * npix);
    const f32x4 im = *reinterpret_cast<const f32x4 *>(img4 + (long)pix * 4);
    const float me = masks[b * mstride + pix];
    // others = sum of the other masks (exactly zero for k == 1, as torch.zeros_like)
    float others = 0.f;
    for (int j = 0; j < k; ++j)
        if (j != b) others += masks[j * mstride + pix];
    f32x4 a = {im.x, im.y, im.z, me}, c = {others, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(out + i * 8) = a;
    *reinterpret_cast<f32x4 *>(out + i * 8 + 4) = c;
}
void pack_value_input_launch(const float *img4, const float *masks, long mask_stride, int k, int npix,
                             float *out, hipStream_t s) {
    hipLaunchKernelGGL(pack_value_input_kernel, dim3(nblocks((long)k * npix)), dim3(256), 0, s, img4, masks,
                       mask_stride, k, npix, out);
}

// MaxPool2d(3, stride 2, pad 1) (modules.py:113,144): -inf padding
__global__ void maxpool_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2, C4 = C / 4;
    const long i = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) * 256L + threadIdx.x;      // neighbouring output rows share an input row
    if (i >= (long)B * OH * OW * C4) return;
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int b = (int)(r / OH);
    // taps outside the image are CLAMPED to the edge pixel, which is inside the window anyway - the same maximum without a
    // branch per tap (a `continue` in front of each load made hipcc wait for every load before issuing the next: 9 serialized
    // round trips per output).  With even H, W (all the engine passes) only the top / left taps can leave the image; the bottom /
    // right clamp keeps an odd size inside the buffer (OH = H / 2 rows are produced either way)
    const int iy0 = max(2 * oh - 1, 0), ix0 = max(2 * ow - 1, 0), iy2 = min(2 * oh + 1, H - 1), ix2 = min(2 * ow + 1, W - 1);
    const float *xb = x + (long)b * H * W * C + c4 * 4;
    f32x4 v[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int iy = dy == 0 ? iy0 : (dy == 2 ? iy2 : 2 * oh), ix = dx == 0 ? ix0 : (dx == 2 ? ix2 : 2 * ow);
            v[dy * 3 + dx] = *reinterpret_cast<const f32x4 *>(xb + ((long)iy * W + ix) * C);
        }
    f32x4 m = v[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) { m.x = fmaxf(m.x, v[t].x); m.y = fmaxf(m.y, v[t].y); m.z = fmaxf(m.z, v[t].z); m.w = fmaxf(m.w, v[t].w); }
    *reinterpret_cast<f32x4 *>(y + i * 4) = m;
}
// test aid: keeps a stream busy for `us` microseconds (wall_clock64: the constant 100 MHz counter)
__global__ void spin_kernel(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
void spin_launch(int us, hipStream_t s) { hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, (long)us * 100); }

void maxpool3x3s2_launch(const float *x, float *y, int B, int H, int W, int C, hipStream_t s) {
    hipLaunchKernelGGL(maxpool_kernel, dim3(nblocks((long)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, s, x, y,
                       B, H, W, C);
}

// bilinear source coordinate, align_corners=False (F.interpolate): src = max(0, (dst+0.5)*scale-0.5)
__device__ __forceinline__ void bil(int dst, float scale, int n, int &i0, int &i1, float &f) {
    float s = ((float)dst + 0.5f) * scale - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > n - 1) i0 = n - 1;
    i1 = i0 < n - 1 ? i0 + 1 : i0;
    f = s - (float)i0;
}

// UpsampleBlock (modules.py:159-163): u[b] = skip_conv(skip)[b * skip_bs] (skip_bs 0: batch-broadcast) + up2x(x[b])
__global__ void upsample2x_add_kernel(const float *__restrict__ x, const float *__restrict__ skip,
                                      float *__restrict__ u, int B, int h, int w, int C, long skip_bs, int skip_bmod) {
    const int OH = 2 * h, OW = 2 * w, C4 = C / 4;
    // XCD-contiguous: an output row pair shares its two low-resolution source rows; dealt round-robin every XCD pulled (nearly) all of x
    // into its own L2 (385 MB fetched for 300 algorithmic at 1/4 scale over a 5-frame group)
    const long i = xcd_contiguous_block((int)blockIdx.x, (int)gridDim.x) * 256L + threadIdx.x;
    if (i >= (long)B * OH * OW * C4) return;
    const int c4 = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int b = (int)(r / OH);
    int y0, y1, x0, x1; float fy, fx;
    bil(oy, 0.5f, h, y0, y1, fy);
    bil(ox, 0.5f, w, x0, x1, fx);
    const float *xb = x + (long)b * h * w * C + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4 *>(xb + ((long)y0 * w + x0) * C);
    const f32x4 v01 = *reinterpret_cast<const f32x4 *>(xb + ((long)y0 * w + x1) * C);
    const f32x4 v10 = *reinterpret_cast<const f32x4 *>(xb + ((long)y1 * w + x0) * C);
    const f32x4 v11 = *reinterpret_cast<const f32x4 *>(xb + ((long)y1 * w + x1) * C);
    const f32x4 sk = *reinterpret_cast<const f32x4 *>(skip + (long)(skip_bmod ? b % skip_bmod : b) * skip_bs + ((long)oy * OW + ox) * C + c4 * 4);
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    f32x4 o;
    o.x = sk.x + (w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x);
    o.y = sk.y + (w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y);
    o.z = sk.z + (w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z);
    o.w = sk.w + (w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w);
    *reinterpret_cast<f32x4 *>(u + i * 4) = o;
}
void upsample2x_add_launch(const float *x, const float *skip, float *u, int B, int h, int w, int C,
                           hipStream_t s, long skip_bs, int skip_bmod) {
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3(nblocks((long)B * 4 * h * w * (C / 4))), dim3(256), 0, s, x,
                       skip, u, B, h, w, C, skip_bs, skip_bmod);
}

// aggregate_wbg (aggregate.py:22-37): odds / sum(odds) after clamping.  MAXOBJ = 8 (every caller of the reference: k = 1, BASELINE config 3:
// k = 5) or STCN_MAX_OBJECTS: the probabilities of a pixel stay in registers either way, the arithmetic and its order are the same.
template <int MAXOBJ>
__device__ __forceinline__ void aggregate_store(const float *p, int k, float *agg, long stride, long pix) {
    float bg = 1.f;
#pragma unroll
    for (int o = 0; o < MAXOBJ; ++o)
        if (o < k) bg *= (1.f - p[o]);
    const float lo = 1e-7f, hi = 1.f - 1e-7f;
    float odds[MAXOBJ + 1];
    float q = fminf(fmaxf(bg, lo), hi);
    odds[0] = q / (1.f - q);
    float tot = odds[0];
#pragma unroll
    for (int o = 0; o < MAXOBJ; ++o) {
        if (o < k) {
            q = fminf(fmaxf(p[o], lo), hi);
            odds[o + 1] = q / (1.f - q);
            tot += odds[o + 1];
        }
    }
    agg[pix] = odds[0] / tot;
#pragma unroll
    for (int o = 0; o < MAXOBJ; ++o)
        if (o < k) agg[(o + 1) * stride + pix] = odds[o + 1] / tot;
}

// Decoder tail (prop_net.py:27-29,192) + aggregate: logit4 -> bilinear x4 -> sigmoid -> aggregate
template <int MAXOBJ>
__global__ void up4_sigmoid_aggregate_kernel(const float *__restrict__ logit4, int k, int h4, int w4,
                                             float *__restrict__ agg, long stride, long obj_stride, long logit_gs, long agg_gs) {
    const int H = 4 * h4, W = 4 * w4;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)H * W) return;
    logit4 += blockIdx.y * logit_gs;                     // frame blockIdx.y of a decode group
    agg += blockIdx.y * agg_gs;
    const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
    int y0, y1, x0, x1; float fy, fx;
    bil(oy, 0.25f, h4, y0, y1, fy);
    bil(ox, 0.25f, w4, x0, x1, fx);
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    float p[MAXOBJ];
#pragma unroll
    for (int o = 0; o < MAXOBJ; ++o) {
        p[o] = 0.f;
        if (o < k) {
            const float *l = logit4 + (long)o * obj_stride;
            const float v = w00 * l[y0 * w4 + x0] + w01 * l[y0 * w4 + x1] + w10 * l[y1 * w4 + x0] +
                            w11 * l[y1 * w4 + x1];
            p[o] = sigmoidf_(v);
        }
    }
    aggregate_store<MAXOBJ>(p, k, agg, stride, i);
}
void up4_sigmoid_aggregate_launch(const float *logit4, int k, int h4, int w4, float *agg, long agg_stride,
                                  hipStream_t s, long obj_stride, int G, long logit_gs, long agg_gs) {
    const dim3 grid(nblocks(16L * h4 * w4), G);
    const long os = obj_stride ? obj_stride : (long)h4 * w4;
    if (k <= 8)
        hipLaunchKernelGGL(up4_sigmoid_aggregate_kernel<8>, grid, dim3(256), 0, s, logit4, k, h4, w4, agg, agg_stride, os, logit_gs, agg_gs);
    else
        hipLaunchKernelGGL(up4_sigmoid_aggregate_kernel<STCN_MAX_OBJECTS>, grid, dim3(256), 0, s, logit4, k, h4, w4, agg, agg_stride, os, logit_gs, agg_gs);
}

// fusion tail (inference_core.py:203-207): sigmoid(fuse_net(...)) per object -> aggregate
template <int MAXOBJ>
__global__ void sigmoid_aggregate_kernel(const float *__restrict__ logit, int k, long npix,
                                         float *__restrict__ agg, long stride) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= npix) return;
    float p[MAXOBJ];
#pragma unroll
    for (int o = 0; o < MAXOBJ; ++o) p[o] = o < k ? sigmoidf_(logit[o * npix + i]) : 0.f;
    aggregate_store<MAXOBJ>(p, k, agg, stride, i);
}
void sigmoid_aggregate_launch(const float *logit, int k, long npix, float *agg, long agg_stride, hipStream_t s) {
    if (k <= 8)
        hipLaunchKernelGGL(sigmoid_aggregate_kernel<8>, dim3(nblocks(npix)), dim3(256), 0, s, logit, k, npix, agg, agg_stride);
    else
        hipLaunchKernelGGL(sigmoid_aggregate_kernel<STCN_MAX_OBJECTS>, dim3(nblocks(npix)), dim3(256), 0, s, logit, k, npix, agg, agg_stride);
}

// final masks (inference_core.py:247-248): argmax over the k+1 rows, first maximum wins
__global__ void argmax_kernel(const float *__restrict__ prob, int kk, int T, long npix, uint8_t *__restrict__ masks) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)T * npix) return;
    float best = prob[i];
    int bi = 0;
    for (int r = 1; r < kk; ++r) {
        const float v = prob[(long)r * T * npix + i];
        if (v > best) { best = v; bi = r; }
    }
    masks[i] = (uint8_t)bi;
}
void argmax_launch(const float *prob, int kk, int T, long npix, uint8_t *masks, hipStream_t s) {
    hipLaunchKernelGGL(argmax_kernel, dim3(nblocks((long)T * npix)), dim3(256), 0, s, prob, kk, T, npix, masks);
}

// |mk|^2 per memory row (prop_net.py:86), 16 lanes per 64-float row
__global__ void rowsumsq_kernel(const float *__restrict__ x, int n, int C, float *__restrict__ out, long x_bs, long out_bs) {
    x += blockIdx.y * x_bs;                              // batch element blockIdx.y (consecutive key-cache slots)
    out += blockIdx.y * out_bs;
    const long gt = blockIdx.x * 256L + threadIdx.x;
    const long row = gt >> 4;
    const int sub = (int)(gt & 15);
    float acc = 0.f;
    if (row < n)
        for (int c = sub * 4; c < C; c += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + row * C + c);
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (row < n && sub == 0) out[row] = acc;
}
void rowsumsq_launch(const float *x, int n, int C, float *out, hipStream_t s, int B, long x_bs, long out_bs) {
    hipLaunchKernelGGL(rowsumsq_kernel, dim3(nblocks((long)n * 16), B), dim3(256), 0, s, x, n, C, out, x_bs, out_bs);
}

__global__ void fill_kernel(float *p, float v, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
void fill_launch(float *p, float v, long n, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(nblocks(n, 256, 4096)), dim3(256), 0, s, p, v, n);
}

__global__ void copy_rows_kernel(const float *__restrict__ src, long ss, float *__restrict__ dst, long ds, int rows,
                                 long n) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= rows * n) return;
    const int r = (int)(i / n);
    const long c = i - r * n;
    dst[r * ds + c] = src[r * ss + c];
}
// two contiguous copies in one launch: a [na floats, multiple of 4] -> da, b [nb floats] -> db (bank insertion: key rows + |mk|^2)
__global__ void copy2_kernel(const float *__restrict__ a, float *__restrict__ da, long na4, const float *__restrict__ b, float *__restrict__ db,
                             long nb) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < na4) reinterpret_cast<f32x4 *>(da)[i] = reinterpret_cast<const f32x4 *>(a)[i];
    else if (i - na4 < nb) db[i - na4] = b[i - na4];
}
void copy2_launch(const float *a, float *da, long na, const float *b, float *db, long nb, hipStream_t s) {
    hipLaunchKernelGGL(copy2_kernel, dim3(nblocks(na / 4 + nb)), dim3(256), 0, s, a, da, na / 4, b, db, nb);
}

void copy_rows_launch(const float *src, long src_stride, float *dst, long dst_stride, int rows, long n,
                      hipStream_t s) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(nblocks(rows * n)), dim3(256), 0, s, src, src_stride, dst, dst_stride,
                       rows, n);
}

// interaction bookkeeping (inference_core.py:220-226)
__global__ void interact_mask_kernel(const float *__restrict__ mask, int mc, int H, int W, int nh, int nw, int lw,
                                     int lh, float *__restrict__ prob_idx, long prs, int kk,
                                     float *__restrict__ padded, float *__restrict__ pos, float *__restrict__ neg) {
    const long npix = (long)nh * nw;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= npix) return;
    const int y = (int)(i / nw), x = (int)(i - (long)y * nw);
    const int sy = y - lh, sx = x - lw;
    const bool in = (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
    for (int c = 0; c < mc; ++c) padded[c * npix + i] = in ? mask[((long)c * H + sy) * W + sx] : 0.f;
    for (int r = 0; r < kk; ++r) {
        const int c = mc == 1 ? 0 : r;                    // torch broadcasting of a 1-channel mask
        const float m = in ? mask[((long)c * H + sy) * W + sx] : 0.f;
        const float d = m - prob_idx[r * prs + i];
        pos[r * npix + i] = fminf(fmaxf(d, 0.f), 1.f);
        neg[r * npix + i] = fminf(fmaxf(-d, 0.f), 1.f);
        prob_idx[r * prs + i] = m;
    }
}
void interact_mask_launch(const float *mask, int mc, int H, int W, int nh, int nw, int lw, int lh,
                          float *prob_idx, long prob_row_stride, int kk, float *padded, float *pos, float *neg,
                          hipStream_t s) {
    hipLaunchKernelGGL(interact_mask_kernel, dim3(nblocks((long)nh * nw)), dim3(256), 0, s, mask, mc, H, W, nh, nw,
                       lw, lh, prob_idx, prob_row_stride, kk, padded, pos, neg);
}

// fusion input (fusion_net.py:35-38; inference_core.py:199-205), 9 channels padded to 12
__global__ void pack_fusion_input_kernel(const float *__restrict__ img4, const float *__restrict__ prev,
                                         const float *__restrict__ curr, const float *__restrict__ attn2, float nc,
                                         float nr, long npix, float *__restrict__ out) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= npix) return;
    const f32x4 im = *reinterpret_cast<const f32x4 *>(img4 + i * 4);
    const f32x4 a = {im.x, im.y, im.z, prev[i]};
    const f32x4 b = {curr[i], attn2[i], attn2[npix + i], nc};
    const f32x4 c = {nr, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(out + i * 12) = a;
    *reinterpret_cast<f32x4 *>(out + i * 12 + 4) = b;
    *reinterpret_cast<f32x4 *>(out + i * 12 + 8) = c;
}
void pack_fusion_input_launch(const float *img4, const float *prev, const float *curr, const float *attn2,
                              float nc, float nr, long npix, float *out, hipStream_t s) {
    hipLaunchKernelGGL(pack_fusion_input_kernel, dim3(nblocks(npix)), dim3(256), 0, s, img4, prev, curr, attn2, nc,
                       nr, npix, out);
}

// ---------------------------------------------------------------------------------------------
// CBAM (cbam.py:21-77): channel gate (avg+max pool -> shared MLP -> sigmoid), then spatial gate
// (channel max/mean -> 7x7 conv -> sigmoid); FeatureFusionBlock adds the result to x (modules.py:48-50).
// scratch layout per call: pooled [B][2][512] | gate [B][512] | sp [B][hw][2] | (unused)
#define CBAM_SLICES 16
// partial avg/max pooling: grid (8 channel groups of 64, B, CBAM_SLICES row slices); part [B][SLICES][2][512]
__global__ __launch_bounds__(256) void cbam_pool_kernel(const float *__restrict__ x, int hw, float *__restrict__ part) {
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, z = blockIdx.z;
    const int per = (hw + CBAM_SLICES - 1) / CBAM_SLICES;
    const int p0 = z * per, p1 = min(hw, p0 + per);
    const float *xb = x + (long)b * hw * 512 + c;
    float sum = 0.f, mx = -__builtin_inff();
    for (int p = p0 + sl; p < p1; p += 4) {
        const float v = xb[(long)p * 512];
        sum += v;
        mx = fmaxf(mx, v);
    }
    __shared__ float ssum[256], smax[256];
    ssum[threadIdx.x] = sum;
    smax[threadIdx.x] = mx;
    __syncthreads();
    if (sl == 0) {
        const int l = threadIdx.x;
        sum = ssum[l] + ssum[l + 64] + ssum[l + 128] + ssum[l + 192];
        mx = fmaxf(fmaxf(smax[l], smax[l + 64]), fmaxf(smax[l + 128], smax[l + 192]));
        float *dst = part + ((long)b * CBAM_SLICES + z) * 1024;
        dst[c] = sum;
        dst[512 + c] = mx;
    }
}

__global__ __launch_bounds__(512) void cbam_mlp_kernel(const float *__restrict__ part, int hw, CbamW cw,
                                                       float *__restrict__ gate) {
    const int b = blockIdx.x, t = threadIdx.x;
    __shared__ float hid[64];
    __shared__ float pooled[1024];            // [avg 512 | max 512]
    {
        float sum = 0.f, mx = -__builtin_inff();
        for (int z = 0; z < CBAM_SLICES; ++z) {          // fixed order: deterministic
            const float *src = part + ((long)b * CBAM_SLICES + z) * 1024;
            sum += src[t];
            mx = fmaxf(mx, src[512 + t]);
        }
        pooled[t] = sum / (float)hw;
        pooled[512 + t] = mx;
    }
    __syncthreads();
    // 64 hidden units (2 inputs x 32); 8 lanes cooperate on one 512-long dot product
    const int unit = t >> 3, sub = t & 7;
    const int which = unit >> 5, h = unit & 31;
    const float *v = pooled + which * 512;
    float acc = 0.f;
    for (int c = sub; c < 512; c += 8) acc += cw.w1[h * 512 + c] * v[c];
    for (int o = 4; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (sub == 0) hid[unit] = fmaxf(acc + cw.b1[h], 0.f);
    __syncthreads();
    float o1 = cw.b2[t], o2 = cw.b2[t];
    for (int j = 0; j < 32; ++j) {
        const float wv = cw.w2[t * 32 + j];
        o1 += wv * hid[j];
        o2 += wv * hid[32 + j];
    }
    gate[(long)b * 512 + t] = sigmoidf_(o1 + o2);
}

__global__ __launch_bounds__(256) void cbam_spool_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                         int hw, long total, float *__restrict__ sp) {
    const long gw = (blockIdx.x * 256L + threadIdx.x) >> 6;   // one wave per (b, pixel)
    const int lane = threadIdx.x & 63;
    if (gw >= total) return;
    const int b = (int)(gw / hw);
    const float *xp = x + gw * 512 + lane * 8;
    const float *g = gate + (long)b * 512 + lane * 8;
    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(xp), a1 = *reinterpret_cast<const f32x4 *>(xp + 4);
    const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g), g1 = *reinterpret_cast<const f32x4 *>(g + 4);
    const float v[8] = {a0.x * g0.x, a0.y * g0.y, a0.z * g0.z, a0.w * g0.w,
                        a1.x * g1.x, a1.y * g1.y, a1.z * g1.z, a1.w * g1.w};
    float mx = v[0], sum = v[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) { mx = fmaxf(mx, v[i]); sum += v[i]; }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, o));
        sum += __shfl_xor(sum, o);
    }
    if (lane == 0) { sp[gw * 2] = mx; sp[gw * 2 + 1] = sum * (1.f / 512.f); }
}

__global__ __launch_bounds__(256) void cbam_apply_kernel(const float *__restrict__ x, const float *__restrict__ gate,
                                                         const float *__restrict__ sp, CbamW cw, int h, int w,
                                                         long total, float *__restrict__ out) {
    const long gw = (blockIdx.x * 256L + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (gw >= total) return;
    const int hw = h * w;
    const int b = (int)(gw / hw), pix = (int)(gw - (long)b * hw);
    const int py = pix / w, px = pix - py * w;
    float acc = 0.f;
    if (lane < 49) {
        const int ky = lane / 7, kx = lane - ky * 7;
        const int iy = py + ky - 3, ix = px + kx - 3;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
            const float *q = sp + ((long)b * hw + iy * w + ix) * 2;
            acc = cw.wsp[lane] * q[0] + cw.wsp[49 + lane] * q[1];   // wsp [2][7][7]: channel 0 = max, 1 = mean
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const float sg = sigmoidf_(acc + cw.bsp);
    const float *xp = x + gw * 512 + lane * 8;
    const float *g = gate + (long)b * 512 + lane * 8;
    f32x4 a0 = *reinterpret_cast<const f32x4 *>(xp), a1 = *reinterpret_cast<const f32x4 *>(xp + 4);
    const f32x4 g0 = *reinterpret_cast<const f32x4 *>(g), g1 = *reinterpret_cast<const f32x4 *>(g + 4);
    a0.x += a0.x * g0.x * sg; a0.y += a0.y * g0.y * sg; a0.z += a0.z * g0.z * sg; a0.w += a0.w * g0.w * sg;
    a1.x += a1.x * g1.x * sg; a1.y += a1.y * g1.y * sg; a1.z += a1.z * g1.z * sg; a1.w += a1.w * g1.w * sg;
    *reinterpret_cast<f32x4 *>(out + gw * 512 + lane * 8) = a0;
    *reinterpret_cast<f32x4 *>(out + gw * 512 + lane * 8 + 4) = a1;
}

void cbam_launch(const float *x, float *out, int B, int h, int w, const CbamW &cw, float *scratch, hipStream_t s) {
    const int hw = h * w;
    float *part = scratch, *gate = scratch + (long)B * CBAM_SLICES * 1024, *sp = gate + (long)B * 512;
    hipLaunchKernelGGL(cbam_pool_kernel, dim3(8, B, CBAM_SLICES), dim3(256), 0, s, x, hw, part);
    hipLaunchKernelGGL(cbam_mlp_kernel, dim3(B), dim3(512), 0, s, part, hw, cw, gate);
    const long total = (long)B * hw;
    hipLaunchKernelGGL(cbam_spool_kernel, dim3(nblocks(total * 64)), dim3(256), 0, s, x, gate, hw, total, sp);
    hipLaunchKernelGGL(cbam_apply_kernel, dim3(nblocks(total * 64)), dim3(256), 0, s, x, gate, sp, cw, h, w, total,
                       out);
}

// ------------------------------------------------------------------------------------------------ matrix-rate probe
// What do the matrix pipes deliver under a pure fp32-MFMA load on THIS chip right now?  12 waves per CU-resident workgroup (3 per
// SIMD, as the F(4x4) GEMM), each wave `iters` x 12 v_mfma_f32_32x32x2_f32 on three accumulator chains with non-zero operands in
// registers: no memory traffic, no LDS.  bench.py reports its rate next to the datasheet peak (the clock the chip holds under
// matrix load is ~2.0 GHz, not the 2.4 GHz of the datasheet figure).
typedef float f32x16p __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(768) void mfma_probe_kernel(float *out, int iters, float seed) {
    f32x16p acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    // 12 operand pairs per lane with pseudo-random mantissas (all-equal or zero operands toggle fewer wires and raise the clock:
    // cdna_hip_programming.md rule 25); their signs flip every iteration so the sums stay bounded
    float a[12], b[12];
    unsigned st = (blockIdx.x * 768u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        st = st * 1664525u + 1013904223u; a[j] = seed + (float)((st >> 8) & 0xffff) / 65536.f;
        st = st * 1664525u + 1013904223u; b[j] = (float)((st >> 8) & 0xffff) / 32768.f - 1.f;
    }
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u)                                           // 48 MFMAs (3072 cycles) per 12 sign flips: VALU share < 2 %
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(3 * j + c + 5 * u) % 12], b[(3 * j + c + 7 * u) % 12], acc[c], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 12; ++j) a[j] = -a[j];
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[c][e];
    if (sum == 123456.789f) out[0] = sum;                                     // keeps the chains alive
}

// FLOP of one launch = grid x 12 waves x iters x 12 MFMAs x 4096
double mfma_probe_launch(float *out, int grid, int iters, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(768), 0, s, out, iters, 0.5f);
    return (double)grid * 12.0 * iters * 12.0 * 4096.0;
}

}  // namespace stcn
