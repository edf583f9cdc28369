// hooks.cpp - stage-level C-ABI entry points used by tests/ and bench.py (see include/stcn_hip.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

using namespace stcn;
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

namespace {
struct TmpWork {
    Work w;
    int rc;
    TmpWork(int nh, int nw, int k) { rc = w.init(nh, nw, k); }
    ~TmpWork() { (void)hipDeviceSynchronize(); w.release(); }
};
struct DevBuf {
    float *p = nullptr;
    int alloc(size_t floats) { HIPCHK(hipMalloc((void **)&p, floats * 4)); return STCN_OK; }
    ~DevBuf() { if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); } }
};
}  // namespace

extern "C" {

int stcn_test_conv(void *stream, const float *x, const float *wgt, const float *bias, const float *res, float *y, int B,
                   int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int flags, int splitk) {
    if (Cin % 4 || pad != KH / 2 || KH != KW) { set_error("stcn_test_conv: Cin%%4==0, square kernel, pad=K/2 required"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    Model m;
    ConvW cw;
    cw.cout = Cout; cw.cin = Cin; cw.cin_p = Cin; cw.kh = KH; cw.kw = KW;
    cw.K = KH * KW * Cin; cw.Kp = (cw.K + 31) / 32 * 32;
    DevBuf wpad, ws;
    RC(wpad.alloc((size_t)Cout * cw.Kp));
    HIPCHK(hipMemsetAsync(wpad.p, 0, (size_t)Cout * cw.Kp * 4, s));
    HIPCHK(hipMemcpy2DAsync(wpad.p, (size_t)cw.Kp * 4, wgt, (size_t)cw.K * 4, (size_t)cw.K * 4, Cout, hipMemcpyDeviceToDevice, s));
    cw.w = wpad.p; cw.bias = const_cast<float *>(bias);
    struct Guard { Model &m; ~Guard() { (void)hipDeviceSynchronize(); for (void *p : m.allocs) (void)hipFree(p); } } guard{m};
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    Work w;
    DevBuf wv;
    if (KH == 3 && stride == 1) {          // stride-1 3x3: the Winograd path, as in the engine
        HIPCHK(hipStreamSynchronize(s));
        std::vector<float> hw((size_t)Cout * cw.Kp);
        HIPCHK(hipMemcpy(hw.data(), wpad.p, hw.size() * 4, hipMemcpyDeviceToHost));
        RC(make_wino(m, cw, hw));
        RC(make_wino_fusion12(m, cw, hw));
        if (flags & 4) { RC(make_wino4(m, cw, hw)); m.wino4_min_wg = 0; }        // flags bit 2: as a decoder layer (F(4x4,3x3))
        // V of either Winograd form: 16 positions x tiles of 2x2 padded to 64, or 36 positions x tiles of 4x4 padded to 128 (for a
        // handful of tiles the padded F(4x4) workspace is the larger one - sized for F(2x2) alone such a case fell back silently)
        const size_t v2 = (size_t)16 * Cin * (((size_t)B * ((OH + 1) / 2) * ((OW + 1) / 2) + 63) / 64 * 64);
        const size_t v4 = (size_t)36 * Cin * (((size_t)B * ((OH + 3) / 4) * ((OW + 3) / 4) + 127) / 128 * 128);
        w.wino_v_floats = v2 > v4 ? v2 : v4;
        RC(wv.alloc(w.wino_v_floats));
        w.wino_v = wv.p;
    }
    m.conv["t"] = cw;
    w.splitk_floats = (size_t)16 * 1024 * 1024;
    RC(ws.alloc(w.splitk_floats));
    w.splitk = ws.p;
    if (Cout == 1) {
        if (stride != 1) { set_error("Cout==1 path is stride 1"); return STCN_E_INVALID; }
        float b0 = 0.f;
        if (bias) HIPCHK(hipMemcpy(&b0, bias, 4, hipMemcpyDeviceToHost));
        conv_n1_launch(x, wpad.p, b0, y, B, H, W, Cin, KH, flags & 1, s);
        set_conv_path("n1");
    } else {
        RC(run_conv(m, w, s, "t", x, Cin, (long)H * W * Cin, nullptr, 0, 0, B, H, W, stride, y, 0, res, (long)OH * OW * Cout,
                    flags & 1, (flags >> 1) & 1, splitk));
    }
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_bench_conv(void *stream, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int splitk,
                    int iters, float *avg_ms, double *flops_per_launch) {
    hipStream_t s = (hipStream_t)stream;
    (void)pad;
    Model m;
    ConvW cw;
    cw.cout = Cout; cw.cin = Cin; cw.cin_p = Cin; cw.kh = KH; cw.kw = KW;
    cw.K = KH * KW * Cin; cw.Kp = (cw.K + 31) / 32 * 32;
    const int OH = (H + 2 * (KH / 2) - KH) / stride + 1, OW = (W + 2 * (KW / 2) - KW) / stride + 1;
    DevBuf x, wt, b, y, ws;
    RC(x.alloc((size_t)B * H * W * Cin)); RC(wt.alloc((size_t)Cout * cw.Kp)); RC(b.alloc(Cout));
    RC(y.alloc((size_t)B * OH * OW * Cout));
    // non-trivial data (zero operands raise the clock: cdna_hip_programming.md rule 25)
    std::vector<float> h((size_t)B * H * W * Cin);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto &v : h) v = rnd();
    HIPCHK(hipMemcpy(x.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    h.assign((size_t)Cout * cw.Kp, 0.f);
    for (auto &v : h) v = rnd() * 0.05f;
    HIPCHK(hipMemcpy(wt.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(b.p, 0, Cout * 4, s));
    cw.w = wt.p; cw.bias = b.p;
    struct Guard { Model &m; ~Guard() { (void)hipDeviceSynchronize(); for (void *p : m.allocs) (void)hipFree(p); } } guard{m};
    Work w;
    DevBuf wv;
    if (KH == 3 && stride == 1) {
        RC(make_wino(m, cw, h));
        RC(make_wino_fusion12(m, cw, h));
        if (getenv("STCN_BENCH_CONV_F4")) RC(make_wino4(m, cw, h));              // time the layer as a decoder layer
        // V of either Winograd form: 16 positions x tiles of 2x2 padded to 64, or 36 positions x tiles of 4x4 padded to 128 (for a
        // handful of tiles the padded F(4x4) workspace is the larger one - sized for F(2x2) alone such a case fell back silently)
        const size_t v2 = (size_t)16 * Cin * (((size_t)B * ((OH + 1) / 2) * ((OW + 1) / 2) + 63) / 64 * 64);
        const size_t v4 = (size_t)36 * Cin * (((size_t)B * ((OH + 3) / 4) * ((OW + 3) / 4) + 127) / 128 * 128);
        w.wino_v_floats = v2 > v4 ? v2 : v4;
        RC(wv.alloc(w.wino_v_floats));
        w.wino_v = wv.p;
    }
    m.conv["t"] = cw;
    w.splitk_floats = (size_t)32 * 1024 * 1024;
    RC(ws.alloc(w.splitk_floats));
    w.splitk = ws.p;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    DevBuf resb;                                                  // STCN_BENCH_CONV_RES=1: with a residual operand (the ResNet conv3 layers)
    const long obs = (long)OH * OW * Cout;
    if (getenv("STCN_BENCH_CONV_RES")) { RC(resb.alloc((size_t)B * obs)); HIPCHK(hipMemsetAsync(resb.p, 0, (size_t)B * obs * 4, s)); }
    for (int i = 0; i < 3; ++i)
        RC(run_conv(m, w, s, "t", x.p, Cin, (long)H * W * Cin, nullptr, 0, 0, B, H, W, stride, y.p, 0, resb.p, resb.p ? obs : 0, 0, 1, splitk));
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        RC(run_conv(m, w, s, "t", x.p, Cin, (long)H * W * Cin, nullptr, 0, 0, B, H, W, stride, y.p, 0, resb.p, resb.p ? obs : 0, 0, 1, splitk));
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (avg_ms) *avg_ms = ms / iters;
    if (flops_per_launch) *flops_per_launch = 2.0 * B * OH * OW * (double)Cout * KH * KW * Cin;
    return STCN_OK;
}

static int pack_one(const float *img_chw, int nh, int nw, float *img4, hipStream_t s) {
    pack_image_launch(img_chw, img4, nh, nw, nh, nw, 0, 0, s);
    return STCN_OK;
}

int stcn_test_encode_key(const stcn_model *m, void *stream, const float *img, int nh, int nw, float *k16, float *f16_thin,
                         float *f16, float *f8, float *f4) {
    if (!m || !img || nh % 16 || nw % 16) { set_error("stcn_test_encode_key: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TmpWork t(nh, nw, 1);
    RC(t.rc);
    const Dims &d = t.w.d;
    DevBuf img4, tf16, tk16, tmsq;
    RC(img4.alloc((size_t)d.npix * 4)); RC(tf16.alloc((size_t)d.hw16 * 1024)); RC(tk16.alloc((size_t)d.hw16 * 64));
    RC(tmsq.alloc(d.hw16));
    RC(pack_one(img, nh, nw, img4.p, s));
    KeyOut ko{k16 ? k16 : tk16.p, tmsq.p, f16_thin, f16 ? f16 : tf16.p, nullptr, nullptr, f8, f4, nullptr, nullptr};
    RC(encode_key(m->m, t.w, s, img4.p, ko));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_test_encode_value(const stcn_model *m, void *stream, const float *img, const float *f16, const float *masks, int k,
                           int nh, int nw, float *out) {
    if (!m || !img || !f16 || !masks || !out || k < 1 || k > STCN_MAX_OBJECTS) { set_error("stcn_test_encode_value: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TmpWork t(nh, nw, k);
    RC(t.rc);
    DevBuf img4;
    RC(img4.alloc((size_t)t.w.d.npix * 4));
    RC(pack_one(img, nh, nw, img4.p, s));
    RC(encode_value(m->m, t.w, s, img4.p, f16, masks, t.w.d.npix, out, 0));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_test_memory_read(void *stream, const float *mk, const float *mv, const float *qk, int N, int Q, int k,
                          int32_t *topk_idx, float *topk_w, float *readout) {
    if (!mk || !mv || !qk || !readout || N < 50 || Q < 1 || k < 1) { set_error("stcn_test_memory_read: bad arguments (N >= 50)"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf msq, cv, ci, cn, gm, tau;
    const size_t pairs = memread_list_pairs(Q);
    RC(msq.alloc(N + 64)); RC(cv.alloc(pairs * 50)); RC(ci.alloc(pairs * 50)); RC(cn.alloc(pairs));
    RC(gm.alloc(pairs * 64)); RC(tau.alloc(Q));
    HIPCHK(hipMemsetAsync(msq.p, 0, (size_t)(N + 64) * 4, s));
    rowsumsq_launch(mk, N, 64, msq.p, s);
    memory_read_launch(mk, msq.p, qk, N, Q, mv, (long)N * 512, k, readout, (long)Q * 512, topk_idx, topk_w,
                       MemReadScratch{cv.p, reinterpret_cast<int32_t *>(ci.p), reinterpret_cast<int32_t *>(cn.p), gm.p, tau.p}, s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

// Timed memory reads on caller-provided device data: `iters` whole reads (pass 1, threshold, pass 2, merge + gather) between
// two HIP events on `stream`; scratch is allocated once, outside the timed region.  ms = average per read.
int stcn_bench_memory_read(void *stream, const float *mk, const float *mv, const float *qk, int N, int Q, int k, int iters,
                           float *readout, float *ms, int32_t *plan7) {
    if (!mk || !mv || !qk || !readout || !ms || N < 50 || Q < 1 || k < 1 || iters < 1) { set_error("stcn_bench_memory_read: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf msq, cv, ci, cn, gm, tau;
    const size_t pairs = memread_list_pairs(Q);
    RC(msq.alloc(N + 64)); RC(cv.alloc(pairs * 50)); RC(ci.alloc(pairs * 50)); RC(cn.alloc(pairs));
    RC(gm.alloc(pairs * 64)); RC(tau.alloc(Q));
    HIPCHK(hipMemsetAsync(msq.p, 0, (size_t)(N + 64) * 4, s));
    rowsumsq_launch(mk, N, 64, msq.p, s);
    const MemReadScratch scr{cv.p, reinterpret_cast<int32_t *>(ci.p), reinterpret_cast<int32_t *>(cn.p), gm.p, tau.p};
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it)
        memory_read_launch(mk, msq.p, qk, N, Q, mv, (long)N * 512, k, readout, (long)Q * 512, nullptr, nullptr, scr, s);
    HIPCHK(hipEventRecord(e0, s));
    for (int it = 0; it < iters; ++it)
        memory_read_launch(mk, msq.p, qk, N, Q, mv, (long)N * 512, k, readout, (long)Q * 512, nullptr, nullptr, scr, s);
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    *ms /= (float)iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (plan7) {
        const MemReadPlan pl = memread_plan(N, Q);
        const int v[7] = {pl.steps, pl.ss, pl.ns, pl.nc1, pl.spc1, pl.nc2, pl.spc2};
        for (int i = 0; i < 7; ++i) plan7[i] = v[i];
    }
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_memread_plan(int N, int Q, int32_t *plan7) {
    if (!plan7 || N < 1 || Q < 1) { set_error("stcn_memread_plan: bad arguments"); return STCN_E_INVALID; }
    const MemReadPlan pl = memread_plan(N, Q);
    const int v[7] = {pl.steps, pl.ss, pl.ns, pl.nc1, pl.spc1, pl.nc2, pl.spc2};
    for (int i = 0; i < 7; ++i) plan7[i] = v[i];
    return STCN_OK;
}

int stcn_test_decode(const stcn_model *m, void *stream, const float *readout, const float *f16_thin, const float *f8,
                     const float *f4, int k, int nh, int nw, float *logit4, float *agg) {
    if (!m || !readout || !f16_thin || !f8 || !f4 || !agg) { set_error("stcn_test_decode: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TmpWork t(nh, nw, k);
    RC(t.rc);
    const Dims &d = t.w.d;
    DevBuf s8, s4;
    RC(s8.alloc((size_t)d.hw8 * 512)); RC(s4.alloc((size_t)d.hw4 * 256));
    RC(run_conv(m->m, t.w, s, "decoder.up_16_8.skip_conv", f8, 512, 0, nullptr, 0, 0, 1, d.h8, d.w8, 1, s8.p, 0, nullptr, 0, 0, 0));
    RC(run_conv(m->m, t.w, s, "decoder.up_8_4.skip_conv", f4, 256, 0, nullptr, 0, 0, 1, d.h4, d.w4, 1, s4.p, 0, nullptr, 0, 0, 0));
    RC(decode(m->m, t.w, s, readout, f16_thin, s8.p, s4.p, agg, d.npix));
    if (logit4) HIPCHK(hipMemcpyAsync(logit4, t.w.logit4, (size_t)k * d.hw4 * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_test_attention(void *stream, const float *mk, const float *qk, const float *pos, const float *neg, int kk, int nh,
                        int nw, float *attn) {
    if (!mk || !qk || !pos || !neg || !attn || kk < 1 || kk > STCN_MAX_OBJECTS + 1) { set_error("stcn_test_attention: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    const int h = nh / 16, w = nw / 16;
    DevBuf msq, pooled, amap, gm, cm, part;
    RC(gm.alloc((size_t)256 * h * w)); RC(cm.alloc(h * w)); RC(part.alloc(attention_part_floats(kk, h * w)));
    RC(msq.alloc(h * w + 64)); RC(pooled.alloc((size_t)std::max(20, attention_nchp(2 * kk)) * h * w)); RC(amap.alloc((size_t)kk * 2 * h * w));
    rowsumsq_launch(mk, h * w, 64, msq.p, s);
    HIPCHK(hipMemsetAsync(msq.p, 0, (size_t)(h * w + 64) * 4, s));
    rowsumsq_launch(mk, h * w, 64, msq.p, s);
    attention_read_launch(mk, msq.p, qk, pos, neg, kk, h, w, pooled.p, amap.p, attn, AttnScratch{gm.p, cm.p, part.p}, s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_test_fusion(const stcn_model *m, void *stream, const float *img, const float *prev, const float *curr,
                     const float *attn, float nc, float nr, int nh, int nw, float *logit) {
    if (!m || !img || !prev || !curr || !attn || !logit) { set_error("stcn_test_fusion: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    TmpWork t(nh, nw, 1);
    RC(t.rc);
    DevBuf img4;
    RC(img4.alloc((size_t)t.w.d.npix * 4));
    RC(pack_one(img, nh, nw, img4.p, s));
    RC(fusion_logit(m->m, t.w, s, img4.p, prev, curr, attn, nc, nr, logit));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_metrics_jf_counts(void *stream, const uint8_t *gt_dev, const uint8_t *pred_dev, int T, int H, int W,
                           int32_t *counts_dev, uint8_t *scratch_dev) {
    if (!gt_dev || !pred_dev || !counts_dev || !scratch_dev || T < 1 || H < 2 || W < 2) {
        set_error("stcn_metrics_jf_counts: bad arguments");
        return STCN_E_INVALID;
    }
    // bound_pix = ceil(0.008 * ||(H, W)||)  (interactions/metrics.py:119-120)
    const int radius = (int)std::ceil(0.008 * std::sqrt((double)H * H + (double)W * W));
    jf_counts_launch(gt_dev, pred_dev, T, H, W, radius, scratch_dev, counts_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_metrics_j_counts(void *stream, const uint8_t *gt_dev, const uint8_t *pred_dev, int T, int H, int W, int32_t *counts_dev) {
    if (!gt_dev || !pred_dev || !counts_dev || T < 1 || H < 1 || W < 1) { set_error("stcn_metrics_j_counts: bad arguments"); return STCN_E_INVALID; }
    jf_counts_launch(gt_dev, pred_dev, T, H, W, -1, nullptr, counts_dev, (hipStream_t)stream);       // radius < 0: intersection / union only
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_metrics_round(void *stream, const uint8_t *masks_dev, int nh, int nw, int lh, int lw, const uint8_t *gt_dev, const uint8_t *annotated_dev,
                       const uint8_t *noobj_dev, int T, int H, int W, int t0, int t1, int j_only, double no_object, uint8_t *gen_dev, uint8_t *scratch_dev,
                       int32_t *counts_dev, double *quality_dev, int32_t *select_dev) {
    if (!masks_dev || !gt_dev || !annotated_dev || !noobj_dev || !gen_dev || !counts_dev || !quality_dev || !select_dev || (!j_only && !scratch_dev) ||
        T < 1 || H < 2 || W < 2 || lh < 0 || lw < 0 || lh + H > nh || lw + W > nw || t0 < 0 || t1 > T || t0 >= t1) {
        set_error("stcn_metrics_round: bad arguments");
        return STCN_E_INVALID;
    }
    const int radius = j_only ? -1 : (int)std::ceil(0.008 * std::sqrt((double)H * H + (double)W * W));      // interactions/metrics.py:119-120
    // frames [t0, t1): composed and counted now; the counts (and gen) of the other frames are the caller's from earlier rounds
    const size_t hw = (size_t)H * W;
    round_score_launch(masks_dev + (size_t)t0 * nh * nw, nh, nw, lh, lw, gt_dev + t0 * hw, annotated_dev + t0, noobj_dev, t1 - t0, H, W, radius, no_object,
                       gen_dev + t0 * hw, scratch_dev, counts_dev + (size_t)t0 * 6, T, quality_dev, select_dev, (hipStream_t)stream, t0);
    HIPCHK(hipGetLastError());
    return STCN_OK;
}

int stcn_bench_mfma_rate(void *stream, int ms_target, float *tflops, float *ms_out) {
    if (!tflops || ms_target < 1 || ms_target > 2000) { set_error("stcn_bench_mfma_rate: bad arguments"); return STCN_E_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 256;
    HIPCHK(hipGetDevice(&dev));
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    DevBuf out;
    RC(out.alloc(64));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    // one wave-iteration = 12 MFMAs x 64 cycles x 3 waves per SIMD: ~1.15 us at 2 GHz; warm-up launch, then the timed one
    const int iters = ((int)(ms_target * 1000.0 / 1.15) + 3) / 4 * 4;         // the kernel walks 4 iterations per loop trip
    (void)mfma_probe_launch(out.p, 2 * cus, (iters / 8 + 3) / 4 * 4, s);
    HIPCHK(hipEventRecord(e0, s));
    const double fl = mfma_probe_launch(out.p, 2 * cus, iters, s);              // two workgroups per CU in turn (one resident: 768 threads x 2 fit)
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(hipGetLastError());
    *tflops = (float)(fl / (ms * 1e-3) / 1e12);
    if (ms_out) *ms_out = ms;
    return STCN_OK;
}

}  // extern "C"
