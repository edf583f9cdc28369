// engine.h - host-side model / workspace / per-video engine of libstcn_hip.so (internal).
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/stcn_hip.h"
#include "kernels.h"

namespace stcn {

void set_error(const char *fmt, ...);
struct Model;
struct ConvW;
// builds the Winograd weights of a stride-1-capable 3x3 conv from its repacked fp32 host weights (no-op when not eligible)
// engine buffers come from a per-device pool (engine.cpp: a hipFree per buffer synchronises the device and stalls the other lanes)
hipError_t pool_malloc(void **p, size_t bytes);
void pool_free(void *p);
void pool_release();
int make_wino(Model &m, ConvW &cw, const std::vector<float> &w_host);
// F(4x4,3x3) weights of a decoder-side 3x3 conv (no-op when not eligible)
int make_wino4(Model &m, ConvW &cw, const std::vector<float> &w_host);
int make_wino_fusion12(Model &m, ConvW &cw, const std::vector<float> &w_host);   // FusionNet conv1: U over 16 zero-padded channels
#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            set_error("%s:%d %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            return STCN_E_HIP;                                                             \
        }                                                                                  \
    } while (0)

struct ConvW {
    float *w = nullptr, *bias = nullptr;   // device: [Cout][Kp], [Cout]
    float *wino_u = nullptr;               // device: Winograd F(2x2,3x3) weights [16][Cin/8][Cout][8] (eligible 3x3 convs)
    float *wino4_u = nullptr;              // device: Winograd F(4x4,3x3) weights [36][Cin/8][Cout][8] (decoder layers)
    float bias0 = 0.f;                     // host copy of bias[0] (Cout == 1 convs)
    int cout = 0, cin = 0, cin_p = 0, kh = 0, kw = 0, K = 0, Kp = 0;
};

struct Model {
    int device = 0;
    std::map<std::string, ConvW> conv;
    CbamW cbam{};
    bool has_fuse = false;
    int wino4_min_wg = 100;                // fewest 32 x 32 workgroups for which a flagged layer takes the F(4x4) kernel (tests: 0)
    std::vector<void *> allocs;
    const ConvW &c(const std::string &name) const;
};

struct Dims {
    int nh, nw, h2, w2, h4, w4, h8, w8, h16, w16;
    long npix;
    int hw2, hw4, hw8, hw16;
    void set(int nh_, int nw_) {
        nh = nh_; nw = nw_;
        h2 = nh / 2; w2 = nw / 2; h4 = nh / 4; w4 = nw / 4; h8 = nh / 8; w8 = nw / 8; h16 = nh / 16; w16 = nw / 16;
        npix = (long)nh * nw; hw2 = h2 * w2; hw4 = h4 * w4; hw8 = h8 * w8; hw16 = h16 * w16;
    }
};

// per-kernel-class accounting (flops always; device time when profiling is on)
struct Prof {
    bool on = false;
    double flops[STCN_K_COUNT] = {0};
    double bytes[STCN_K_COUNT] = {0};     // algorithmic HBM bytes (each operand once)
    double exec_flops[STCN_K_COUNT] = {0};   // FLOP the matrix cores executed (Winograd convs: 2.25x fewer than algorithmic)
    int launches[STCN_K_COUNT] = {0};
    struct Ev { int cls; bool hbm; hipEvent_t a, b; };    // a, b contiguous: attach() hands out &a as hipEvent_t[2]
    std::deque<Ev> events;                      // deque: attach() returns pointers into it
    std::vector<hipEvent_t> pool;
    void reset();
    void begin(int cls, hipStream_t s);
    // register an event pair that the launch itself will fill (hipExtLaunchKernelGGL); null when off
    hipEvent_t *attach(int cls, bool hbm_bound_conv = false);
    // conv launches below the machine balance (HBM-bound), also counted in the STCN_K_CONV totals
    double hbm_conv_flops = 0, hbm_conv_bytes = 0, hbm_conv_ms = 0; int hbm_conv_launches = 0;
    double wino2_flops = 0, wino4_flops = 0;    // algorithmic FLOP of the convs that ran as Winograd F(2x2,3x3) / F(4x4,3x3)
    void end(hipStream_t s);
    int collect(float *ms);
    ~Prof();
};

// scratch for one in-flight frame computation (sized for nh x nw and k objects)
struct Work {
    Dims d{};
    int k = 0;
    size_t S = 0;                       // floats per big buffer
    float *A = nullptr, *B = nullptr, *C = nullptr, *D = nullptr;
    float *splitk = nullptr; size_t splitk_floats = 0;
    float *wino_v = nullptr; size_t wino_v_floats = 0;   // Winograd-transformed input of the conv in flight
    float *cbam = nullptr;
    float *readout = nullptr;           // [k][hw16][512]
    float *logit4 = nullptr;            // [k][hw4]
    float *flogit = nullptr;            // [k][npix]  fusion logits
    float *agg = nullptr;               // [k+1][npix] aggregated output of the current frame
    float *agg_alt = nullptr;           // second buffer: decode groups alternate while the side stream fuses the previous group
    float *pooled = nullptr, *amap = nullptr, *attn = nullptr;   // attention read
    float *cand_v = nullptr; int32_t *cand_i = nullptr, *cand_n = nullptr;   // memory-read chunk winners
    float *gmax = nullptr, *tau = nullptr;                       // memory-read group maxima / thresholds
    float *qk = nullptr;                // [group][hw16][64] queries of a decode group
    float *vin = nullptr;               // value-encoder packed input [k][npix][8]
    Prof *prof = nullptr;
    Knobs kn = Knobs::from_env();       // launch-level tunables, read once when the workspace is made
    int conv_cls = STCN_K_CONV;         // accounting class of the conv GEMMs launched through this workspace (fusion_logit switches it)
    std::vector<void *> allocs;
    int init(int nh, int nw, int k, int key_batch = 1, int group = 1);   // group: frames decoded per pass (k == 1)
    void release();
};

// ---- stages (enqueue only) -----------------------------------------------------------------
struct KeyOut { float *k16, *msq, *f16_thin, *f16, *s8, *s4, *f8_copy, *f4_copy, *dthin = nullptr, *cthin = nullptr; };
void inject_failure_after(int n);        // tests: the n-th launch_status() of this thread fails
int launch_status(const char *what);
const char *last_conv_path();            // kernel family of this thread's last run_conv ("wino4 chunks=2", "direct_pointwise splitk=1", ...)
void set_conv_path(const char *s);
void conv_trace(int on);                 // tests: log "name=path" of every conv this thread enqueues
const char *conv_trace_get();     // STCN_OK, or STCN_E_HIP with the failing launch class in the error string
int run_conv(const Model &m, Work &w, hipStream_t s, const char *name, const float *x0, int c0, long bs0,
             const float *x1, int c1, long bs1, int B, int H, int W, int stride, float *y, long y_bs,
             const float *res, long res_bs, int relu_in, int relu_out, int force_splitk = 0, int res_bmod = 0);
// B consecutive frames at once; outputs of frame b at o.<ptr> + b * out_bs
int encode_key(const Model &m, Work &w, hipStream_t s, const float *img4, const KeyOut &o, int B = 1, long out_bs = 0);
// vd / vc: cached frame-only halves of fuser.block1 (nullptr: compute the full two-source convs)
int encode_value(const Model &m, Work &w, hipStream_t s, const float *img4, const float *f16,
                 const float *masks, long mask_stride, float *out, long out_bs, const float *vd = nullptr,
                 const float *vc = nullptr);
int value_frame_parts(const Model &m, Work &w, hipStream_t s, const float *f16, float *vd, float *vc);
// dthin / cthin: cached frame-only halves of decoder.compress (nullptr: full two-source convs)
// G > 1: G frames at once, batch laid out [object][frame] - readout [k][G][hw16][512] (what one memory read of G * hw16
// queries writes), agg [G][k+1][npix], the per-frame inputs of frame g at <ptr> + g * slot_bs (consecutive key-cache slots)
int decode(const Model &m, Work &w, hipStream_t s, const float *readout, const float *f16_thin,
           const float *s8, const float *s4, float *agg, long agg_stride, const float *dthin = nullptr,
           const float *cthin = nullptr, int G = 1, long slot_bs = 0, long agg_gs = 0);      // agg_gs: floats between the agg blocks of consecutive frames (0: (k + 1) * agg_stride)
int fusion_logit(const Model &m, Work &w, hipStream_t s, const float *img4, const float *prev,
                 const float *curr, const float *attn2, float nc, float nr, float *logit);

}  // namespace stcn

struct stcn_model { stcn::Model m; };

struct stcn_engine {
    const stcn::Model *model = nullptr;
    hipStream_t stream = nullptr;
    int T = 0, H = 0, W = 0, k = 0, mem_freq = 5;
    stcn_engine_opts opts{-1, -1, -1, -1};   // tunables: as given to stcn_engine_create_ex, then resolved (environment / defaults) at create
    int lw = 0, uw = 0, lh = 0, uh = 0;
    stcn::Dims d{};
    float *images4 = nullptr;          // [T][nh][nw][4]; immutable after create, shared with clones
    std::shared_ptr<void> images_owner;  // frees images4 with the last engine that uses it
    float *prob = nullptr;             // caller-owned [k+1][T][npix]
    uint8_t *masks = nullptr;          // caller-owned [T][npix]
    // key-feature cache
    int n_slots = 0;
    std::vector<int> slot_of;          // frame -> slot or -1
    std::vector<char> vparts_ready;    // frame -> value-encoder frame parts (vd, vc) are in its slot
    int n_cached = 0;
    float *cache = nullptr; size_t slot_floats = 0;
    // memory bank: rows = slots * hw16
    int bank_cap = 0, n_certain = 0;
    float *bank_k = nullptr, *bank_msq = nullptr, *bank_v = nullptr;
    std::vector<void *> retired;       // bank buffers replaced by a larger generation, freed once retire_ev has fired
    hipEvent_t retire_ev = nullptr;
    std::set<int> interacted;
    float *mask_pad = nullptr, *pos = nullptr, *neg = nullptr;   // [k+1][npix] each
    stcn::Work work;
    // key-encoder look-ahead: frames ahead of the decode chain are encoded on a side stream
    hipStream_t side = nullptr;
    stcn::Work work_side;
    // round 6: the BACKWARD sweep of an interaction on its own stream and workspace, concurrently with the forward sweep (the two sweeps
    // of do_pass share nothing but the certain memory, which neither writes: inference_core.py:250-253 runs them one after the other).
    // Its temporary bank slots grow DOWNWARD from bank_lo, in front of the certain slots, the forward sweep's upward behind them: each
    // sweep reads one contiguous row range [its temporaries | certain] / [certain | its temporaries].  One video in flight only (same
    // switch as the key-encoder look-ahead); clips longer than the key cache (flush-all policy) keep the serial order.
    hipStream_t stream2 = nullptr;
    stcn::Work work2;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int bank_lo = 0;                     // bank slots in front of the certain slots (0: serial sweeps)
    std::vector<hipEvent_t> key_ready;   // per frame: recorded on `side` after its encode_key
    std::vector<char> key_pending;       // per frame: main stream has not yet waited on key_ready
    int lookahead = 0;
    // FusionNet of a decoded group on the side stream (rounds >= 2: the side stream has no keys to encode), two agg buffers
    bool fuse_side = false;
    hipEvent_t ev_dec[2] = {nullptr, nullptr}, ev_fuse[2] = {nullptr, nullptr};
    char fuse_pending[2] = {0, 0};       // ev_fuse[b] is recorded and the main stream has not waited for it yet
    int agg_buf = 0;
    int key_batch = 1;                   // frames per key-encoder pass (env STCN_KEY_BATCH)
    int group = 1;                       // frames per memory-read + decoder pass (env STCN_DECODE_BATCH, k == 1)
    stcn::Prof prof;
    stcn_stats stats{};
    std::string failed;                  // non-empty: a failed interaction left prob / masks half-written (see stcn_interact)
    std::vector<void *> allocs;
};
