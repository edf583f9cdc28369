// kernels.h - launch wrappers of the gfx950 kernels (host-callable, enqueue on a stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stcn_hip.h"   // STCN_MAX_OBJECTS

namespace stcn {

// ---------------------------------------------------------------- division by a launch invariant
// An integer division by a run-time value costs ~25 VALU instructions on gfx950 (float reciprocal + correction), also for
// wave-uniform operands; the conv kernels did 5-10 of them per workgroup in set-up / epilogue code, which the fp32 MFMA of the
// co-resident workgroups cannot hide.  q = (x * magic) >> (31 + s) with magic = floor(2^(31+s) / d) + 1, s = ceil(log2 d), is
// exact for 0 <= x < 2^31 (x * (magic * d - 2^(31+s)) <= x * d < 2^(31+s)): one v_mul_hi + one shift.
struct FastDiv {
    unsigned magic, shift, d;       // shift = s - 1 (after the mul_hi's 32); d == 1: identity
};
static inline FastDiv fastdiv_make(unsigned d) {
    FastDiv f{0, 0, d};
    if (d <= 1) return f;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.magic = (unsigned)(((1ull << (31 + s)) / d) + 1);
    f.shift = s - 1;
    return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int fastdiv(int x, const FastDiv f) {
    return f.d <= 1 ? x : (int)(__umulhi((unsigned)x, f.magic) >> f.shift);
}
// Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2): blocks b and b + 8 share one.  Kernels whose
// NEIGHBOURING blocks share input (stencil halos, bilinear taps, im2col rows) walk their index space in this order instead: every XCD
// gets one contiguous eighth, so shared lines are fetched into ONE L2 (bijective for any grid size; speed only, never correctness).
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk) {
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}
#endif

// ---------------------------------------------------------------- launch-level tunables
// Read from the environment ONCE per workspace (= once per engine, once per stage-hook call) by Knobs::from_env() and carried in
// every ConvP - never read per launch: a driver may change os.environ between legs while host threads of other lanes launch.
struct Knobs {
    int wino_min_cin = 128;     // STCN_WINO_MIN_CIN: fewest input channels for which a stride-1 3x3 conv takes the F(2x2) path
    int wino_ppw = 0;           // STCN_WINO_PPW: 1 / 2 pins the F(2x2) GEMM instance (positions per wave), 0 = by shape
    int wino4_chunk_mb = 160;   // STCN_WINO4_CHUNK_MB: V bytes per slice of a chunked F(4x4) launch (0: unchunked)
    int fusion_conv12 = 0;      // STCN_FUSION_CONV12: FusionNet conv1 on the direct FusionNet kernel
    int fusion_wino = 1;        // STCN_FUSION_WINO: FusionNet convs as Winograd F(2x2) inside the workgroup
    int pw_chain = 2;           // STCN_PW_CHAIN: large pointwise convs on the chain kernel (several tiles per workgroup, one pipeline): 0 off,
                                // 1 consecutive tiles per workgroup, 2 tiles strided over the grid (default)
    static Knobs from_env();
};

// ---------------------------------------------------------------- implicit-GEMM convolution
// Activations NHWC fp32.  Input = channel-concat of up to two sources (second may be batch-broadcast).
struct ConvP {
    const float *x0, *x1;   // sources; x1 == nullptr when unused
    int c0, c1;             // channels of each source (multiples of 4)
    long bs0, bs1;          // batch strides in elements (0 = broadcast over the batch)
    unsigned x0_bytes, x1_bytes, w_bytes;   // extents for the buffer descriptors (hardware bounds check)
    int B, H, W;            // input batch / spatial
    int OH, OW;
    int KH, KW, stride, pad;
    int Cin;                // c0 + c1
    int M, N;               // M = B*OH*OW rows, N = Cout
    int K, Kp;              // K = KH*KW*Cin, Kp = K rounded up to 32 (weights zero-padded)
    const float *w;         // [N][Kp], k ordered (kh, kw, cin)
    const float *bias;      // [N] or nullptr
    const float *res;       // residual [B][OH*OW][N] or nullptr
    long res_bs;            // batch stride of res (0 = broadcast)
    int res_bmod;           // > 0: the residual of batch element b is res + (b % res_bmod) * res_bs (a per-FRAME tensor under a
                            // batch laid out [object][frame]: decode groups of multi-object engines)
    float *y;               // [M][N]
    long y_bs;              // output batch stride in elements (0 = dense OH*OW*N)
    int relu_in, relu_out;
    int splitk;             // >= 1
    float *partial;         // [splitk][M][N] workspace when splitk > 1
    // tail balancing (splitk == 1 only, see conv_plan): tiles [0, rem_full) run whole; each of the remaining tiles is
    // cut into rem_split K pieces of rem_per K tiles whose tile-local partial sums go to `partial`
    // ([(tile - rem_full) * rem_split + piece][BM][BN]) and are summed by conv_reduce_tiles_kernel
    int rem_full, rem_split, rem_per;
    const float *wino_u;    // Winograd-transformed weights [16][Cin/8][N][8] (stride-1 3x3 convs with Cin >= 64, N % 64 == 0) or nullptr
    const float *wino4_u;   // Winograd F(4x4,3x3) weights [36][Cin/8][N][8] (decoder layers only: winograd4.hip) or nullptr
    FastDiv fd_ohw, fd_ow;  // divisions by OH*OW and OW
    FastDiv fd_cin, fd_kw;  // divisions by Cin and KW (the stem instance decodes (tap, channel) of its K index per thread and K tile)
    int pointwise;          // 1x1, stride 1, no padding, one dense source: im2col row m IS activation row m (no row decode)
    int affine_out;         // y (and res, if any) are dense [M][N]: element (m, n) at (m * N + n) * 4 bytes, < 4 GiB
    int tile_big;           // 1 = 128x128 workgroup tiles (fp32 kernel)
    int panel;              // > 0: tiles are walked in panels of this many n-tiles (fp32 kernel)
    int chain;              // > 0: pointwise chain kernel, this many consecutive tiles per workgroup (conv_plan)
    Knobs kn;               // the workspace's snapshot of the launch-level tunables
};
// fills the launch plan of p (tile variant, split-K or tail balancing); force_splitk > 0 pins a plain split-K;
// workspace_floats = capacity of p.partial
void conv_plan(ConvP &p, int force_splitk, size_t workspace_floats);
// ev_gemm / ev_red: optional {start, stop} event pairs attached to the GEMM / reduce dispatches themselves
// (hipExtLaunchKernelGGL: kernel begin/end timestamps, no extra barrier packets)
void conv_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev_gemm = nullptr, hipEvent_t *ev_red = nullptr);
const char *conv_variant_name(const ConvP &p);     // the conv_gemm_kernel instance a planned conv takes ("direct", "direct_pointwise", ...)
// split-K tail of a conv whose slabs p.partial [p.splitk][M][N] are filled: sum + bias / residual / ReLU -> y
void conv_reduce_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev_red = nullptr);
// Winograd F(2x2,3x3) path (winograd.hip): V workspace floats this conv needs, or 0 when it is not eligible
size_t wino_workspace_floats(const ConvP &p);
void wino_launch(const ConvP &p, float *V, size_t slab_floats, hipStream_t s, hipEvent_t *ev_in = nullptr, hipEvent_t *ev_gemm = nullptr,
                 hipEvent_t *ev_red = nullptr);
int wino_plan_splitk(const ConvP &p, size_t slab_floats);
void wino_transform_weights(const float *w, int N, int Cin, int Kp, float *U);
// fewest input channels of an F(4x4) layer.  64 (8 k-blocks, one K piece) since round 5: the 64-channel 3x3 convs of the value encoder's
// ResNet-18 layer1 over the objects of a multi-object engine (129 600 rows at k = 5) take 56 instead of 98 us each - config 3 219 -> 224 frames/s;
// at one object they have 52 workgroups and stay on the direct kernel (wino4_min_wg).  The key encoder's 64-channel res2 convs are flagged too since the
// key trunk moved to F(4x4) (engine.cpp wino4_layer; STCN_WINO4_KEY=0 at model creation takes the whole key trunk off F(4x4) again).
#ifndef W4_MIN_CIN
#define W4_MIN_CIN 64
#endif
// Winograd F(4x4,3x3) path (winograd4.hip, decoder layers): V workspace floats, or 0 when not eligible / fewer than min_wg workgroups
size_t wino4_workspace_floats(const ConvP &p, int min_wg);
// ev_in / ev_gemm: one {start, stop} pair per chunk (wino4_chunks: the transform and the GEMM alternate over slices of the tiles
// when V would not stay in the memory-side cache)
void wino4_launch(const ConvP &p, float *V, size_t slab_floats, hipStream_t s, hipEvent_t *const *ev_in = nullptr,
                  hipEvent_t *const *ev_gemm = nullptr, hipEvent_t *ev_red = nullptr);
int wino4_chunks(const ConvP &p, size_t slab_floats);
// does wino4_launch cut the last round of this conv's workgroups into K pieces (a reduce launch follows)?
bool wino4_tail_split(const ConvP &p, size_t slab_floats);
void wino4_transform_weights(const float *w, int N, int Cin, int Kp, float *U);

// FusionNet convs (fusion_conv.hip): 3x3, stride 1, Cout = 32, Cin = 32 or 12, one dense image: weights in registers, patch in LDS
bool fusion_conv_eligible(const ConvP &p);
bool fusion_conv_winograd(const ConvP &p);      // an eligible conv runs as Winograd F(2x2,3x3) inside the workgroup (32 -> 32 layers)
void fusion_conv_launch(const ConvP &p, hipStream_t s, hipEvent_t *ev_gemm = nullptr);

// Cout == 1 convolution (decoder.pred, FusionNet.final_conv): one dot product per output pixel.
// x [B,H,W,C] (C multiple of 4), w [KH*KW*C], y [B*H*W]; stride 1, "same" padding.
void conv_n1_launch(const float *x, const float *w, float bias, float *y, int B, int H, int W, int C,
                    int KH, int relu_in, hipStream_t s);

// ---------------------------------------------------------------- elementwise / pooling
void maxpool3x3s2_launch(const float *x, float *y, int B, int H, int W, int C, hipStream_t s);
// NCHW image [3,H,W] (unpadded) -> NHWC4 padded [nh,nw,4] with zero border (pad lw, lh)
void pack_image_launch(const float *img_chw, float *out, int H, int W, int nh, int nw, int lw, int lh,
                       hipStream_t s);
// value-encoder input [k,nh,nw,8] = (rgb from NHWC4 image, mask_i, sum_{j!=i} mask_j, 0,0,0)
void pack_value_input_launch(const float *img4, const float *masks, long mask_stride, int k, int npix,
                             float *out, hipStream_t s);
// u[b] = skip[b * skip_bs] (skip_bs 0: broadcast over b) + bilinear_up2x(x[b]); x [B,h,w,C] -> u [B,2h,2w,C]
// skip_bmod > 0: the skip of batch element b is skip + (b % skip_bmod) * skip_bs
void upsample2x_add_launch(const float *x, const float *skip, float *u, int B, int h, int w, int C,
                           hipStream_t s, long skip_bs = 0, int skip_bmod = 0);
// logit4 [k,h4*w4] -> bilinear x4 -> sigmoid -> aggregate_wbg -> agg [k+1][nh*nw] (row stride agg_stride)
// obj_stride: floats between the logit planes of consecutive objects (0 = h4*w4)
// G > 1: G frames in one launch - frame g reads logit4 + g * logit_gs and writes agg + g * agg_gs
void up4_sigmoid_aggregate_launch(const float *logit4, int k, int h4, int w4, float *agg,
                                  long agg_stride, hipStream_t s, long obj_stride = 0, int G = 1, long logit_gs = 0, long agg_gs = 0);
// logits [k,npix] -> sigmoid -> aggregate -> agg rows (fusion output)
void sigmoid_aggregate_launch(const float *logit, int k, long npix, float *agg, long agg_stride,
                              hipStream_t s);
// masks[t] = argmax over rows of prob [(k+1), T, npix] for all t (first max wins)
void argmax_launch(const float *prob, int kk, int T, long npix, uint8_t *masks, hipStream_t s);
// rows [n, C] -> msq[n] = sum_c x^2
// B > 1: B batch elements in one launch (element b at x + b * x_bs -> out + b * out_bs)
void rowsumsq_launch(const float *x, int n, int C, float *out, hipStream_t s, int B = 1, long x_bs = 0, long out_bs = 0);
void fill_launch(float *p, float v, long n, hipStream_t s);
void spin_launch(int us, hipStream_t s);        // test aid: one wave that keeps stream s busy for `us` microseconds
void copy_rows_launch(const float *src, long src_stride, float *dst, long dst_stride, int rows, long n,
                      hipStream_t s);
// a [na floats, na % 4 == 0] -> da and b [nb floats] -> db in one launch
void copy2_launch(const float *a, float *da, long na, const float *b, float *db, long nb, hipStream_t s);
// interaction mask handling (inference_core.py:220-226): pads mask [mc,H,W] into [mc,nh,nw] planes,
// writes pos/neg = clamp(+-(mask - prob[:,idx])) for all kk rows, then prob[:,idx] = mask (broadcast)
void interact_mask_launch(const float *mask, int mc, int H, int W, int nh, int nw, int lw, int lh,
                          float *prob_idx, long prob_row_stride, int kk, float *padded, float *pos,
                          float *neg, hipStream_t s);

// ---------------------------------------------------------------- CBAM (cbam.py:21-77)
struct CbamW { const float *w1, *b1, *w2, *b2, *wsp; float bsp; };  // 512->32->512 MLP, 7x7 [2] conv
// x [B,hw,512] -> out = x + CBAM(x); scratch >= B*(16*1024 + 512 + 2*hw) floats
void cbam_launch(const float *x, float *out, int B, int h, int w, const CbamW &cw, float *scratch,
                 hipStream_t s);

// ---------------------------------------------------------------- space-time memory read
// launch plan of the top-50 read: `steps` 64-row steps of the bank; pass 1 visits every ss-th step (ns of them) in nc1 chunks of
// spc1 sampled steps, pass 2 all steps in nc2 chunks of spc2
struct MemReadPlan { int steps, ss, ns, nc1, spc1, nc2, spc2; };
MemReadPlan memread_plan(int N, int Q);
// upper bound of (chunks x queries) of either pass for Q queries: sizes the scratch below
size_t memread_list_pairs(int Q);
// P = memread_list_pairs(Q): cand_v / cand_i [P][50], cand_n [P], gmax [64 P], tau [Q]
struct MemReadScratch { float *cand_v; int32_t *cand_i; int32_t *cand_n; float *gmax; float *tau; };
// dynamic LDS above 64 KB has to be opted into once per (device, kernel function)
void allow_big_lds(const void *kernel, size_t lds);
// mk [N,64], msq [N] (+ >= 64 readable floats of padding), qk [Q,64]; mv [k][N][512] with object stride mv_os; readout [k][Q][512] with
// object stride ro_os.  topk_idx/topk_w optional outputs [Q,50].
void memory_read_launch(const float *mk, const float *msq, const float *qk, int N, int Q,
                        const float *mv, long mv_os, int k, float *readout, long ro_os,
                        int32_t *topk_idx, float *topk_w, MemReadScratch scr, hipStream_t s);
// fusion attention read: mk,qk [hw,64]; pos,neg [kk][16h*16w planes] -> attn [kk][2][nh*nw]; pooled: scratch of attention_nchp(2 kk) * h * w floats
struct AttnScratch { float *gmax, *cmax, *part; };   // [256][hw], [hw], attention_part_floats(kk, hw) ([16][hw][19] up to 8 objects)
int attention_nchp(int nch);                          // channels 2 kk padded to the widths the pass kernel is instantiated for
size_t attention_part_floats(int kk, int hw);
// pos == nullptr: `pooled` already holds attention_pool_launch's output for this interaction
void attention_pool_launch(const float *pos, const float *neg, int kk, int h, int w, float *pooled, hipStream_t s);
void attention_read_launch(const float *mk, const float *msq, const float *qk, const float *pos,
                           const float *neg, int kk, int h, int w, float *pooled, float *amap,
                           float *attn, AttnScratch scr, hipStream_t s);
// fusion input [nh*nw,12] = (rgb, prev, curr, attn_pos, attn_neg, nc, nr, 0,0,0)
void pack_fusion_input_launch(const float *img4, const float *prev, const float *curr,
                              const float *attn2, float nc, float nr, long npix, float *out,
                              hipStream_t s);

// ---------------------------------------------------------------- J / F metric counts (interactions/metrics.py)
// gt, pred uint8 [T,H,W] (non-zero = object); bmap scratch [T*H*W]; counts [T][6] =
// (intersection, union, gt boundary px, pred boundary px, matched gt boundary px, matched pred boundary px)
void jf_counts_launch(const uint8_t *gt, const uint8_t *pred, int T, int H, int W, int radius, uint8_t *bmap,
                      int *counts, hipStream_t s);
// one annotation round scored on the device: gen = engine mask / GT on annotated frames, J or J&F counts, fp64 quality per frame, arg-min
// (pointers at the first of the Tn frames to recount; quality / arg-min over the T_all frames of the clip, t0 = index of that first frame)
void round_score_launch(const uint8_t *masks, int nh, int nw, int lh, int lw, const uint8_t *gt, const uint8_t *annotated, const uint8_t *noobj,
                        int Tn, int H, int W, int radius, double no_object, uint8_t *gen, uint8_t *bmap, int *counts, int T_all, double *quality,
                        int *select, hipStream_t s, int t0);

// debug/stress: launch ONLY the merge stage on prepared candidate lists (cand_v/cand_i [NC][Q][50])
void merge_only_launch(const float *cand_v, const int32_t *cand_i, int NC, int Q, const float *mv, long mv_os, int k,
                       float *readout, long ro_os, hipStream_t s);

// pure fp32-MFMA load (no memory traffic): launches `grid` workgroups of 12 waves x iters x 12 MFMAs, returns the FLOP of the launch
double mfma_probe_launch(float *out, int grid, int iters, hipStream_t s);

}  // namespace stcn
