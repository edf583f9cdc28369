/*
 * stcn_hip.h - C ABI of libstcn_hip.so, the MI355X (gfx950) STCN mask-propagation engine.
 *
 * The reference (thanosDelatolas/eva-vos) has no FFI: its boundary for this path is the Python
 * class mivos.inference_core.InferenceCore.  Every entry point below cites the reference
 * interface it replaces (paths under /root/reference).  The only caller is the ctypes shim
 * eva_vos_amd/inference_core.py (re-exported as mivos/inference_core.py); see INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes; no torch / C++ types cross the boundary.
 *   - every function returns 0 on success or a negative STCN_E_* code; stcn_last_error() gives a
 *     thread-local message.  No exception crosses the ABI.
 *   - "dev" pointers are HIP device pointers on the engine's device; the caller owns them and
 *     keeps them alive for the lifetime stated per function.
 *   - an engine is NOT thread-safe; one engine <-> one HIP stream <-> one video.  Distinct engines
 *     may be driven concurrently from distinct host threads.
 *   - all activations are fp32 (the reference computes in fp32 throughout).
 */
#ifndef STCN_HIP_H
#define STCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STCN_OK            0
#define STCN_E_INVALID    -1   /* bad argument / shape */
#define STCN_E_MISSING    -2   /* a required weight tensor is absent */
#define STCN_E_HIP        -3   /* HIP runtime error */
#define STCN_E_STATE      -4   /* call not valid in the current engine state */

/* Objects per engine (`num_objects` of InferenceCore.__init__, mivos/inference_core.py:34): the reference class has no limit (its own
 * pipelines pass 1: datasets/annotation_dataset.py:55-67); this library is built and tested for 1 .. STCN_MAX_OBJECTS. */
#define STCN_MAX_OBJECTS  32

typedef struct stcn_model  stcn_model;   /* folded + repacked weights, read-only, shareable */
typedef struct stcn_engine stcn_engine;  /* per-video state */

/* One named fp32 tensor of a state_dict (host memory, C-contiguous).
 * Replaces: torch `state_dict()` of PropagationNetwork / FusionNet handed to
 * InferenceCore.__init__ as live nn.Modules (mivos/inference_core.py:34-38;
 * eval_annotation_method.py:58-64). */
typedef struct {
    const char  *name;      /* e.g. "key_encoder.layer3.0.conv2.weight" */
    const float *data;      /* host pointer, fp32 */
    int32_t      ndim;      /* 1..4 */
    int64_t      shape[4];
} stcn_weight_desc;

const char *stcn_last_error(void);
const char *stcn_version(void);

/* Build a model on HIP device `device`: folds every eval-mode BatchNorm into its convolution,
 * repacks weights to the engine's [Cout][kh][kw][Cin] layout and uploads them.
 * `prop` must hold the 405-tensor PropagationNetwork dict (int64 num_batches_tracked entries may be
 * omitted); `fuse` the 12-tensor FusionNet dict or NULL/0 (then fusion rounds fail with STCN_E_STATE).
 * Replaces: prop_net.to(device) / fuse_net.to(device) (inference_core.py:36-38). */
int stcn_model_create(int device,
                      const stcn_weight_desc *prop, int n_prop,
                      const stcn_weight_desc *fuse, int n_fuse,
                      stcn_model **out);
int stcn_model_destroy(stcn_model *m);

/* Create the per-video engine.
 *   images_dev : fp32 [1,T,3,H,W] (NCHW, normalized, unpadded), read once during this call.
 *   prob_dev   : fp32 [k+1,T,1,nh,nw] owned by the caller (a torch tensor in the shim); the engine
 *                initialises it (bg row 1e-7, others 0) and writes into it on every interact().
 *   masks_dev  : uint8 [T,1,nh,nw] owned by the caller; per-frame argmax written by interact().
 *   nh, nw     : H, W rounded up to multiples of 16 (symmetric zero pad, tensor_util.py:62-80).
 *   stream     : hipStream_t (as void*) all engine work is enqueued on; NULL = default stream.
 * Replaces: InferenceCore.__init__ (inference_core.py:34-99) with mem_profile=0. */
int stcn_engine_create(const stcn_model *m, int T, int H, int W, int k, int mem_freq,
                       void *stream, const float *images_dev, float *prob_dev, uint8_t *masks_dev,
                       stcn_engine **out);
int stcn_engine_destroy(stcn_engine *e);

/* Engine tunables given explicitly instead of through the environment.  A NEGATIVE field means "not given": the engine then
 * takes the environment variable (read once, while the engine is created) or its default.  Results never depend on them.
 *   lookahead     STCN_LOOKAHEAD     (default 2)  0 = no side stream at all (drivers that keep several videos in flight per GPU)
 *   decode_batch  STCN_DECODE_BATCH  (default 8)  frames per memory-read + decoder pass, clipped to mem_freq and 16 / k
 *   key_batch     STCN_KEY_BATCH     (default 0 = max(4, decode group))  frames per key-encoder pass
 *   fuse_side     STCN_FUSE_SIDE     (default 1)  FusionNet of rounds >= 2 on the side stream
 * stcn_engine_create(...) = stcn_engine_create_ex(..., NULL, out).  A clone inherits its source's resolved values.
 * (No reference counterpart: InferenceCore has no tuning surface; eva_vos_amd.InferenceCore(..., engine_options={...}).) */
typedef struct { int32_t lookahead, decode_batch, key_batch, fuse_side; } stcn_engine_opts;
int stcn_engine_create_ex(const stcn_model *m, int T, int H, int W, int k, int mem_freq,
                          void *stream, const float *images_dev, float *prob_dev, uint8_t *masks_dev,
                          const stcn_engine_opts *opts, stcn_engine **out);
/* The values the engine actually runs with (after the environment / defaults were resolved and clipped: look-ahead is 0 when the
 * clip is longer than the key cache, the decode group never exceeds mem_freq or 16 / k objects, fuse_side needs a fusion network). */
int stcn_engine_get_opts(const stcn_engine *e, stcn_engine_opts *out);

/* Back to the state right after stcn_engine_create (same clip): forgets interactions, certain memory and the
 * key-feature cache, re-initialises prob / masks.  Lets a driver reuse one engine's device memory for the
 * next sample of the same shape instead of re-allocating ~4.6 GB.  (No reference counterpart: the reference
 * constructs a new InferenceCore per sample, interactions/eval.py:99.) */
int stcn_engine_reset(stcn_engine *e);

/* Deep copy of all engine state (certain memory, key cache, interaction set).  The clone writes to
 * the caller-provided prob/masks buffers, which the caller must have filled with a copy of the
 * source's.  Replaces: copy.deepcopy(processor) (interactions/policies.py:103-104). */
int stcn_engine_clone(const stcn_engine *src, float *prob_dev, uint8_t *masks_dev, void *stream,
                      stcn_engine **out);

/* Interact -> propagate both ways -> fuse -> per-frame argmax.
 *   mask_dev      : fp32 [mask_channels,1,H,W] (unpadded).  mask_channels is k (no bg row; only
 *                   valid for k==1, scribble==0) or k+1 (bg first; requires scribble!=0), exactly as
 *                   the reference accepts (inference_core.py:220-233).
 *   Enqueues on the engine stream and returns without synchronising; prob_dev / masks_dev are
 *   complete once the stream has drained.
 *   Errors: bad arguments (STCN_E_INVALID) and a failing up-front reservation of bank memory (STCN_E_HIP: out of memory) are
 *   detected before anything is touched - the engine stays usable, the call may be repeated.  A failure INSIDE the
 *   interaction (failed launch, out of memory while the bank grows) rolls the host bookkeeping back (set of
 *   interacted frames, certain-memory count) and puts the engine into a failed state: prob_dev / masks_dev hold
 *   a partially propagated round, further stcn_interact calls return STCN_E_STATE until stcn_engine_reset().
 * Replaces: InferenceCore.interact (inference_core.py:209-259) incl. do_pass (:126-191) and
 * fuse_one_frame (:193-207). */
int stcn_interact(stcn_engine *e, const float *mask_dev, int mask_channels, int idx, int scribble);

/* Counters of the last interact(): frames visited by the propagation loops, key-encoder misses,
 * value encodes, fused frames, final bank sizes of the forward / backward pass. */
typedef struct {
    int32_t frames, key_miss, value_enc, fused, bank_fwd, bank_bwd;
} stcn_stats;
int stcn_get_stats(const stcn_engine *e, stcn_stats *out);

/* Algorithmic work of the last interact() in FLOP (2 x MAC of every conv / GEMM issued). */
int stcn_get_flops(const stcn_engine *e, double *flops);

/* ---- stage-level hooks (used by tests/ and bench.py only; all pointers are device fp32) --------
 * Layouts are the engine's internal ones: activations NHWC, i.e. [rows = h*w][channels]. */

/* Generic convolution through the implicit-GEMM kernel (what every nn.Conv2d of the path lowers to:
 * modules.py / mod_resnet.py / prop_net.py convs).  x: [B,H,W,Cin] (Cin multiple of 4),
 * w: [Cout,KH,KW,Cin], bias: [Cout], res: [B,OH,OW,Cout] or NULL, y: [B,OH,OW,Cout].
 * flags: bit0 relu on input, bit1 relu on output.  splitk<=0 lets the engine choose. */
int stcn_test_conv(void *stream, const float *x, const float *w, const float *bias, const float *res,
                   float *y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                   int pad, int flags, int splitk);

/* Which kernel family the calling thread's last convolution (stcn_test_conv, or the last conv an interact() enqueued) ran as:
 * "direct splitk=1", "direct_pointwise splitk=1", "direct_narrow ...", "direct_smallc ...", "direct_big ...", "... +tail ...",
 * "wino2 ppw=1 splitk=2" (Winograd F(2x2,3x3)), "wino4 chunks=2 +tail" (F(4x4,3x3)), "fusion_wino", "fusion_direct", "n1".
 * Tests assert the path per shape: a silent fall-back to another instance would still pass a numerical comparison. */
const char *stcn_last_conv_path(void);
/* Test hook: stcn_test_conv_trace(1) starts (and clears) a log of the CALLING THREAD's convolutions, one "layer=path\n" line per conv
 * enqueued (stage hooks and stcn_interact alike: engine launches happen on the caller's thread); (0) stops it; _get returns the log.
 * Lets a sequence test assert that e.g. every decoder layer of a 853x480 clip really ran as "wino4 ...". */
int stcn_test_conv_trace(int on);
const char *stcn_test_conv_trace_get(void);

/* encode_key of one frame (prop_net.py:172-177).  img: [1,3,nh,nw] NCHW padded.  Outputs (NHWC):
 * k16 [hw16,64], f16_thin [hw16,512], f16 [hw16,1024], f8 [hw8,512], f4 [hw4,256]; any may be NULL. */
int stcn_test_encode_key(const stcn_model *m, void *stream, const float *img, int nh, int nw,
                         float *k16, float *f16_thin, float *f16, float *f8, float *f4);

/* encode_value (prop_net.py:153-170).  masks: [k,nh*nw] planes; f16 NHWC; out [k,hw16,512]. */
int stcn_test_encode_value(const stcn_model *m, void *stream, const float *img, const float *f16,
                           const float *masks, int k, int nh, int nw, float *out);

/* Space-time memory read (prop_net.py:80-115 with the top-50 softmax of :53-60).
 * mk [N,64], mv [k,N,512], qk [Q,64] -> topk_idx [Q,50] (int32), topk_w [Q,50], readout [k,Q,512]. */
int stcn_test_memory_read(void *stream, const float *mk, const float *mv, const float *qk,
                          int N, int Q, int k, int32_t *topk_idx, float *topk_w, float *readout);

/* Measurement hook of the same read: `iters` whole reads on caller-provided device data between two HIP events on
 * `stream` (scratch allocated outside the timed region); *ms = average per read; plan7 (may be NULL) receives the launch
 * plan {steps, pass-1 sample stride, sampled steps, pass-1 chunks, steps per chunk, pass-2 chunks, steps per chunk}. */
int stcn_bench_memory_read(void *stream, const float *mk, const float *mv, const float *qk, int N, int Q, int k,
                           int iters, float *readout, float *ms, int32_t *plan7);

/* Decoder + sigmoid + soft aggregation (prop_net.py:13-30,189-192; aggregate.py:22-37).
 * readout [k,hw16,512], f16_thin/f8/f4 NHWC -> logit4 [k,hw4] (may be NULL), agg [k+1,nh*nw]. */
int stcn_test_decode(const stcn_model *m, void *stream, const float *readout, const float *f16_thin,
                     const float *f8, const float *f4, int k, int nh, int nw, float *logit4, float *agg);

/* Attention read of fusion (prop_net.py:117-138,198-211).  mk,qk [hw16,64]; pos,neg [kk,nh*nw]
 * -> attn [kk,2,nh*nw]. */
int stcn_test_attention(void *stream, const float *mk, const float *qk, const float *pos,
                        const float *neg, int kk, int nh, int nw, float *attn);

/* FusionNet logit for one object (fusion_net.py:32-50).  img [1,3,nh,nw] NCHW; prev,curr [nh*nw];
 * attn [2,nh*nw] -> logit [nh*nw]. */
int stcn_test_fusion(const stcn_model *m, void *stream, const float *img, const float *prev,
                     const float *curr, const float *attn, float nc, float nr, int nh, int nw,
                     float *logit);

/* Launch plan of the top-50 memory read for N bank rows and Q queries (what stcn_test_memory_read / the engine will run):
 * plan7 = { 64-row steps, pass-1 sample stride, sampled steps, pass-1 chunks, steps per pass-1 chunk, pass-2 chunks,
 * steps per pass-2 chunk }.  Lets tests assert WHICH plan (sample stride 1/2/4/8) a comparison exercised. */
int stcn_memread_plan(int N, int Q, int32_t *plan7);

/* Engine buffers (workspaces, key cache, memory bank, packed clip) come from a per-device pool: a destroyed engine's buffers
 * wait there for the next engine of the same sizes (the reference's drivers build one InferenceCore per sample; a hipFree
 * per buffer would synchronise the device under the other videos in flight).  STCN_POOL_GB bounds the pool (default 64, 0 = off);
 * this call returns everything it holds to the driver. */
int stcn_pool_release(void);

/* Test hook (fault injection): the n-th kernel-launch status check made by the CALLING THREAD from now on reports a
 * failure (n = 0 disarms).  Used to show that a failing stcn_interact leaves the engine in a defined state.
 * n = -1: the next stcn_interact of the calling thread fails in its up-front memory reservation - before anything is touched:
 * that call returns STCN_E_HIP and the engine stays USABLE (no failed state). */
int stcn_test_fail_at(int n);

/* Test hook (stream ordering): every FusionNet group handed to the engine's side stream starts `us` microseconds late (0: off;
 * process-wide).  Orderings between the two streams that rest on events alone become observable: a missing wait is a wrong result. */
int stcn_test_side_delay_us(int us);

/* Time `iters` launches of the dominant kernel (implicit-GEMM conv) on `stream` with HIP events;
 * returns average milliseconds per launch.  Used by bench.py for the roofline object. */
int stcn_bench_conv(void *stream, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad, int splitk, int iters, float *avg_ms, double *flops_per_launch);

/* Matrix rate the chip sustains under a pure fp32-MFMA load (v_mfma_f32_32x32x2_f32 on register operands, no memory
 * traffic) for about ms_target milliseconds: TFLOP/s.  bench.py reports it beside the datasheet peak (the clock under
 * matrix load is lower than the datasheet's). */
int stcn_bench_mfma_rate(void *stream, int ms_target, float *tflops, float *ms_out);

/* Per-kernel-class time of the last interact() measured with HIP events on the engine stream
 * (enabled by stcn_engine_set_profiling(e,1); adds a few % overhead).  ms[] indexed by STCN_K_*. */
enum { STCN_K_CONV = 0, STCN_K_CONV_REDUCE, STCN_K_MEMREAD, STCN_K_ELEMWISE, STCN_K_CONV_N1,
       STCN_K_OTHER, STCN_K_WINO_INPUT /* Winograd input transform of the 3x3 convs */,
       STCN_K_FUSION_CONV /* the FusionNet conv GEMMs (rounds >= 2) */, STCN_K_ATTENTION /* fusion attention read */, STCN_K_COUNT };
int stcn_engine_set_profiling(stcn_engine *e, int on);
int stcn_get_kernel_ms(const stcn_engine *e, float *ms /*[STCN_K_COUNT]*/, int32_t *launches /*[STCN_K_COUNT]*/);
/* Algorithmic FLOP (2 x MAC) issued per kernel class by the last interact(). */
int stcn_get_kernel_flops(const stcn_engine *e, double *flops /*[STCN_K_COUNT]*/);
/* Algorithmic HBM bytes (every operand of every launch once; conv class only) of the last interact(). */
int stcn_get_kernel_bytes(const stcn_engine *e, double *bytes /*[STCN_K_COUNT]*/);
/* FLOP the matrix cores actually EXECUTED per class (= the algorithmic FLOP except for the convs that ran as Winograd
 * F(2x2,3x3): 2.25x fewer multiplies than the 2*M*N*K of stcn_get_kernel_flops). */
int stcn_get_kernel_exec_flops(const stcn_engine *e, double *flops /*[STCN_K_COUNT]*/);
/* Conv launches of the last interact() whose arithmetic intensity (algorithmic FLOP / algorithmic bytes) lies below the
 * machine balance 157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B - HBM-bound, e.g. the 1x1 channel expansions of the key encoder.
 * They are part of the STCN_K_CONV totals; out[6] = { FLOP, bytes, device ms (profiling on), launches of those launches,
 * algorithmic FLOP of the convs that ran as Winograd F(2x2,3x3), ... as Winograd F(4x4,3x3) }. */
int stcn_get_conv_regimes(stcn_engine *e, double *out /*[6]*/);

/* ---- caller-side metric -------------------------------------------------------
 * Integer counts behind J (region IoU) and F (boundary measure) for T frames, on the device.
 *   gt_dev, pred_dev : uint8 [T,H,W], non-zero = object (unpadded masks)
 *   counts_dev       : int32 [T,6] = intersection, union, gt boundary px, pred boundary px,
 *                      gt boundary px matched within the disk, pred boundary px matched within the disk
 *   scratch_dev      : uint8 [T*H*W]
 * The disk radius is ceil(0.008 * ||(H,W)||) as in the reference.  Enqueues on `stream`, no sync.
 * Replaces: get_j_and_f / f_measure / _seg2bmap (interactions/metrics.py:24-34,38-97,100-160) as called per
 * frame per round by eval_processor_metric (interactions/eval.py:50-79). */
int stcn_metrics_jf_counts(void *stream, const uint8_t *gt_dev, const uint8_t *pred_dev, int T, int H, int W,
                           int32_t *counts_dev, uint8_t *scratch_dev);
/* The region measure alone: counts_dev int32 [T,6] with columns 0 (intersection) and 1 (union) filled, the others zero; no scratch.
 * What the oracle annotation policy needs per round (interactions/mask.py:113-146 selects the frame with the worst J; eval_processor_metric
 * with metric='j', interactions/eval.py:27-81). */
int stcn_metrics_j_counts(void *stream, const uint8_t *gt_dev, const uint8_t *pred_dev, int T, int H, int W, int32_t *counts_dev);
/* One annotation round scored ON THE DEVICE (enqueue only): what the reference's loops do on the host after every interact() -
 * eval_processor_metric (interactions/eval.py:27-81: annotated frames count with their ground truth :57-60, per-frame J or J&F :62-79, the
 * NO_OBJECT token for frames without the object :67) followed by the oracle policy's arg-min (interactions/mask.py:130-133; policies.py:62-72).
 *   masks_dev     : the engine's uint8 [T][nh][nw] mask tensor (InferenceCore.masks), cropped at (lh, lw) to H x W
 *   gt_dev        : uint8 [T,H,W] ground truth, non-zero = object;  annotated_dev / noobj_dev : uint8 [T] flags
 *   gen_dev       : uint8 [T,H,W] OUT - the evaluated masks (engine mask, GT on annotated frames): the state util/fq_dataset.py:64-84 saves
 *   scratch_dev   : uint8 [T*H*W] (unused when j_only);  counts_dev : int32 [T,6] OUT as stcn_metrics_jf_counts
 *   quality_dev   : double [T] OUT - per-frame J (j_only) or J&F, no_object for flagged frames; fp64 with the host path's operations in the
 *                   host path's order, i.e. bit-identical to it;  select_dev : int32 [1] OUT - first index of the minimum (numpy.argmin)
 * Only select (4 bytes) has to cross PCIe per round; the quality rows of a session can be fetched together at its end.
 * [t0, t1): the frames whose masks the round just propagated can have changed (the spans on both sides of the new annotation, and the annotated
 *   frame itself): only they are composed and counted; gen / counts of the other frames are kept from the caller's earlier rounds (the first
 *   round passes 0, T).  quality / select always cover all T frames. */
int stcn_metrics_round(void *stream, const uint8_t *masks_dev, int nh, int nw, int lh, int lw, const uint8_t *gt_dev, const uint8_t *annotated_dev,
                       const uint8_t *noobj_dev, int T, int H, int W, int t0, int t1, int j_only, double no_object, uint8_t *gen_dev, uint8_t *scratch_dev,
                       int32_t *counts_dev, double *quality_dev, int32_t *select_dev);

#ifdef __cplusplus
}
#endif
#endif /* STCN_HIP_H */
