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
#include "kernels.h"

namespace stcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDT = 36;

template <int WM, int WN, bool SMALLC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvP p, const int tiles_n,
                                                        const int ntile, const int kt_per_split) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                  // [2][BM][LDT]
    float *Bs = smem + 2 * BM * LDT;   // [2][BN][LDT]

    // ---- block -> (split, m-tile, n-tile), XCD-contiguous (bijective remap)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int split = swz / ntile;
    const int tile = swz - split * ntile;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int kc = t & 7, r0 = t >> 3;

    // ---- per-thread im2col row state (rows r0 + 32*i of the A tile); 32-bit element offsets
    int ih0[WM], iw0[WM], base0[WM], base1[WM];
    bool rvalid[WM];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int m = tm * BM + r0 + 32 * i;
        rvalid[i] = m < p.M;
        const int mm = rvalid[i] ? m : 0;
        const int b = mm / ohw, pix = mm - b * ohw;
        const int oh = pix / p.OW, ow = pix - oh * p.OW;
        ih0[i] = oh * p.stride - p.pad;
        iw0[i] = ow * p.stride - p.pad;
        base0[i] = b * (int)p.bs0 + (ih0[i] * p.W + iw0[i]) * p.c0;
        base1[i] = b * (int)p.bs1 + (ih0[i] * p.W + iw0[i]) * p.c1;
    }
    const float *wrow[WN];
    bool nvalid[WN];
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int n = tn * BN + r0 + 32 * i;
        nvalid[i] = n < p.N;
        wrow[i] = p.w + (long)(nvalid[i] ? n : 0) * p.Kp + kc * 4;
    }

    const int nkt = p.Kp / BK;
    const int kt0 = split * kt_per_split;
    const int kt1 = min(nkt, kt0 + kt_per_split);

    // Filter-tap walk.  Fast path (Cin and c0 multiples of 32): a K tile lies inside one tap and one
    // source, so (kh, kw, channel base) are workgroup-uniform and advance incrementally (scalar unit).
    // Small-Cin path (stems, Cin = 4/8/12): per-thread decode of its 4-channel chunk.
    int u_kh, u_kw, u_cb;
    {
        const int k0 = kt0 * BK;
        const int tap = k0 / p.Cin;
        u_cb = k0 - tap * p.Cin;
        u_kh = tap / p.KW;
        u_kw = tap - u_kh * p.KW;
    }

    // gload() only ISSUES the loads (raw data + validity bits); the zero/ReLU fix-up happens in
    // lstore(), after the MFMAs of the current tile, so the loads stay in flight under the MFMAs.
    f32x4 ra[WM], rb[WN];
    unsigned okmask = 0;
    auto gload = [&](int kt) {
        int kh, kw, c;
        bool kvalid = true;
        if (SMALLC) {
            const int k = kt * BK + kc * 4;
            kvalid = k < p.K;
            const int tap = k / p.Cin;
            c = k - tap * p.Cin;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        } else {
            kh = u_kh; kw = u_kw; c = u_cb + kc * 4;
            u_cb += BK;
            if (u_cb >= p.Cin) { u_cb = 0; if (++u_kw == p.KW) { u_kw = 0; ++u_kh; } }
        }
        const bool src1 = c >= p.c0;
        const float *sbase = src1 ? p.x1 : p.x0;
        const int cs = src1 ? p.c1 : p.c0;
        const int coff = (kh * p.W + kw) * cs + (src1 ? c - p.c0 : c);
        okmask = 0;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            // branch-free gather: out-of-image / out-of-range chunks read a safe address
            const bool ok = kvalid && rvalid[i] && (unsigned)(ih0[i] + kh) < (unsigned)p.H &&
                            (unsigned)(iw0[i] + kw) < (unsigned)p.W;
            const int off = ok ? (src1 ? base1[i] : base0[i]) + coff : 0;
            okmask |= (ok ? 1u : 0u) << i;
            ra[i] = *reinterpret_cast<const f32x4 *>(sbase + off);
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(wrow[i] + (long)kt * BK);
    };
    const float relu_lo = p.relu_in ? 0.f : -__builtin_inff();
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const bool ok = (okmask >> i) & 1u;
            f32x4 v = ra[i];
            v.x = ok ? fmaxf(v.x, relu_lo) : 0.f; v.y = ok ? fmaxf(v.y, relu_lo) : 0.f;
            v.z = ok ? fmaxf(v.z, relu_lo) : 0.f; v.w = ok ? fmaxf(v.w, relu_lo) : 0.f;
            *reinterpret_cast<f32x4 *>(&As[(buf * BM + r0 + 32 * i) * LDT + kc * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            f32x4 v = rb[i];
            if (!nvalid[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4 *>(&Bs[(buf * BN + r0 + 32 * i) * LDT + kc * 4]) = v;
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (kt0 < kt1) {
        gload(kt0);
        lstore(0);
    }
    __syncthreads();

    const int arow = (wm * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int brow = (wn * 32 + (lane & 31)) * LDT + (lane >> 5) * 4;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) gload(kt + 1);
        const float *a_s = As + buf * BM * LDT + arow;
        const float *b_s = Bs + buf * BN * LDT + brow;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(a_s + kb * 8);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(b_s + kb * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the staging write-back behind the MFMAs
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = tn * BN + wn * 32 + (lane & 31);
    if (n >= p.N) return;
    const int mbase = tm * BM + wm * 32 + 4 * (lane >> 5);
    if (p.splitk > 1) {
        float *dst = p.partial + (long)split * p.M * p.N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            if (m < p.M) dst[(long)m * p.N + n] = acc[r];
        }
        return;
    }
    const float bv = p.bias ? p.bias[n] : 0.f;
    const bool needb = p.res != nullptr || p.y_bs != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        if (m < p.M) {
            float v = acc[r] + bv;
            long yo = (long)m * p.N + n;
            if (needb) {
                const int b = m / ohw;
                const long po = (long)(m - b * ohw) * p.N + n;
                if (p.res) v += p.res[(long)b * p.res_bs + po];
                if (p.y_bs) yo = (long)b * p.y_bs + po;
            }
            if (p.relu_out) v = fmaxf(v, 0.f);
            p.y[yo] = v;
        }
    }
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
            const int b = m / ohw;
            const long po = (long)(m - b * ohw) * p.N + n;
            if (p.res) v += *reinterpret_cast<const f32x4 *>(p.res + (long)b * p.res_bs + po);
            if (p.y_bs) yo = (long)b * p.y_bs + po;
        }
        if (p.relu_out) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<f32x4 *>(p.y + yo) = v;
    }
}

static inline bool narrow_variant(const ConvP &p) { return p.N <= 32; }

int conv_choose_splitk(const ConvP &p) {
    const int BM = narrow_variant(p) ? 128 : 64, BN = narrow_variant(p) ? 32 : 64;
    const long tiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nkt = p.Kp / BK;
    const double mn = (double)p.M * p.N;
    double best = 1e300;
    int best_s = 1;
    const int smax = nkt / 4 < 1 ? 1 : (nkt / 4 > 32 ? 32 : nkt / 4);
    for (int s = 1; s <= smax; ++s) {
        const int per = (nkt + s - 1) / s;
        if ((long)per * (s - 1) >= nkt) continue;  // an empty split
        const long rounds = (tiles * s + 255) / 256;
        double cost = (double)rounds * (per + 3.0);   // in K-tile times of one workgroup (~0.43 us)
        if (s > 1) cost += 4.6 + (s + 1) * 2.3e-6 * mn;
        if (cost < best) { best = cost; best_s = s; }
    }
    return best_s;
}

size_t conv_workspace_floats(const ConvP &p) {
    return p.splitk > 1 ? (size_t)p.splitk * p.M * p.N : 0;
}

void conv_launch(const ConvP &p, hipStream_t s) {
    const bool narrow = narrow_variant(p);
    const int BM = narrow ? 128 : 64, BN = narrow ? 32 : 64;
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ntile = tiles_m * tiles_n;
    const int nkt = p.Kp / BK;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const size_t lds = (size_t)2 * (BM + BN) * LDT * sizeof(float);
    const dim3 grid(ntile * p.splitk);
    const bool smallc = (p.Cin % 32) != 0 || (p.x1 && (p.c0 % 32) != 0);
    if (narrow && smallc)
        hipLaunchKernelGGL((conv_gemm_kernel<4, 1, true>), grid, dim3(256), lds, s, p, tiles_n, ntile, per);
    else if (narrow)
        hipLaunchKernelGGL((conv_gemm_kernel<4, 1, false>), grid, dim3(256), lds, s, p, tiles_n, ntile, per);
    else if (smallc)
        hipLaunchKernelGGL((conv_gemm_kernel<2, 2, true>), grid, dim3(256), lds, s, p, tiles_n, ntile, per);
    else
        hipLaunchKernelGGL((conv_gemm_kernel<2, 2, false>), grid, dim3(256), lds, s, p, tiles_n, ntile, per);
    if (p.splitk > 1) {
        const long total4 = (long)p.M * p.N / 4;
        long blocks = (total4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    }
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

void conv_n1_launch(const float *x, const float *w, float bias, float *y, int B, int H, int W, int C,
                    int KH, int relu_in, hipStream_t s) {
    const long npix = (long)B * H * W;
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
