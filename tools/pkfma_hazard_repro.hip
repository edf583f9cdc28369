// pkfma_hazard_repro.hip - standalone check (no engine code) of an observation made while building libstcn_hip.so:
// a PACKED fp32 VALU instruction (v_pk_fma_f32) that is the first consumer of a freshly loaded register pair seemed to
// return a wrong value in part of a wave while ANOTHER wave on the same SIMD issued 16-bit MFMAs.
//
// Two tiny kernels:
//   victim<PACKED>  : out[q][c] = sum_j w[q][j] * table[idx[q][j]][c]  (50 rows of 2 KB per query, one wave per query),
//                     accumulating either with v_pk_fma_f32 or with v_fmac_f32 (inline asm, so the compiler cannot change it);
//   spinner<KIND>   : every wave issues MFMAs in a loop for a fixed number of iterations
//                     (KIND 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_32x32x2_f32, 2: plain v_fma_f32).
// The victim runs `iters` times on stream A while the spinner runs back to back on stream B (256 blocks x 256 threads = one
// wave per SIMD on every CU, so victim waves share SIMDs with spinner waves); each victim output is compared bit for bit
// with the victim's own solo result.  Prints a 2 x 4 matrix of "outputs that differ / iters".
//
// build:  hipcc -O3 --offload-arch=gfx950 tools/pkfma_hazard_repro.hip -o gpurun_out/pkfma_repro
// run  :  gpurun_out/pkfma_repro [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool PACKED>
__global__ __launch_bounds__(256) void victim(const float *__restrict__ table, const int *__restrict__ idx,
                                              const float *__restrict__ w, int Q, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    f32x2 a[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll 5
    for (int j = 0; j < 50; ++j) {
        const float wj = w[(long)q * 50 + j];
        const float *row = table + (long)idx[(long)q * 50 + j] * 512 + 4 * lane;
        const f32x4 r0 = *reinterpret_cast<const f32x4 *>(row), r1 = *reinterpret_cast<const f32x4 *>(row + 256);
        const f32x2 ww = {wj, wj};
        const f32x2 p[4] = {{r0.x, r0.y}, {r0.z, r0.w}, {r1.x, r1.y}, {r1.z, r1.w}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (PACKED) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[c]) : "v"(p[c]), "v"(ww));
            } else {
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c].x) : "v"(p[c].x), "v"(wj));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[c].y) : "v"(p[c].y), "v"(wj));
            }
        }
    }
    float *dst = out + (long)q * 512 + 4 * lane;
    *reinterpret_cast<f32x4 *>(dst) = f32x4{a[0].x, a[0].y, a[1].x, a[1].y};
    *reinterpret_cast<f32x4 *>(dst + 256) = f32x4{a[2].x, a[2].y, a[3].x, a[3].y};
}

template <int KIND>
__global__ __launch_bounds__(256) void spinner(float *sink, int loops) {
    const int lane = threadIdx.x & 63;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f16x8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (lane + i)); hb[i] = (_Float16)(0.002f * (lane - i)); }
    float fa = 0.001f * lane, fb = 0.5f, fc = 0.f;
    for (int l = 0; l < loops; ++l) {
        if (KIND == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(fc) : "v"(fa), "v"(fb));
        }
    }
    float s = fc;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) sink[0] = s;      // keeps the loop alive
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 80;
    const int Q = 1620, N = 1620;
    float *table, *w, *ref, *outs, *sink;
    int *idx;
    CK(hipMalloc(&table, (size_t)N * 512 * 4)); CK(hipMalloc(&w, (size_t)Q * 50 * 4)); CK(hipMalloc(&idx, (size_t)Q * 50 * 4));
    CK(hipMalloc(&ref, (size_t)Q * 512 * 4)); CK(hipMalloc(&outs, (size_t)iters * Q * 512 * 4)); CK(hipMalloc(&sink, 64));
    std::vector<float> h((size_t)N * 512), hw((size_t)Q * 50);
    std::vector<int> hx((size_t)Q * 50);
    unsigned st = 777u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto &v : h) v = rnd() * 10.f;
    for (size_t i = 0; i < hw.size(); ++i) { hw[i] = 0.02f + 0.001f * rnd(); st = st * 1664525u + 1013904223u; hx[i] = (int)((st >> 8) % N); }
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(idx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s (%s), %d CUs; %d victim launches per cell, victim grid %d x 256, spinner grid %d x 256\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, iters, (Q + 3) / 4, prop.multiProcessorCount);
    const char *vn[2] = {"v_pk_fma_f32", "v_fmac_f32  "};
    const char *sn[4] = {"nothing", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32", "v_fma_f32 (VALU)"};
    std::vector<float> hr((size_t)Q * 512), ho((size_t)Q * 512);
    for (int v = 0; v < 2; ++v)
        for (int s = 0; s < 4; ++s) {
            auto launch_victim = [&](float *out) {
                if (v == 0) hipLaunchKernelGGL(victim<true>, dim3((Q + 3) / 4), dim3(256), 0, sa, table, idx, w, Q, out);
                else hipLaunchKernelGGL(victim<false>, dim3((Q + 3) / 4), dim3(256), 0, sa, table, idx, w, Q, out);
            };
            launch_victim(ref);
            CK(hipStreamSynchronize(sa));
            for (int i = 0; i < iters; ++i) {
                const dim3 g(prop.multiProcessorCount), b(256);
                if (s == 1) hipLaunchKernelGGL(spinner<0>, g, b, 0, sb, sink, 4000);
                if (s == 2) hipLaunchKernelGGL(spinner<1>, g, b, 0, sb, sink, 4000);
                if (s == 3) hipLaunchKernelGGL(spinner<2>, g, b, 0, sb, sink, 4000);
                launch_victim(outs + (size_t)i * Q * 512);
            }
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            CK(hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost));
            int bad = 0; long first = -1;
            for (int i = 0; i < iters; ++i) {
                CK(hipMemcpy(ho.data(), outs + (size_t)i * Q * 512, ho.size() * 4, hipMemcpyDeviceToHost));
                if (memcmp(hr.data(), ho.data(), hr.size() * 4)) {
                    ++bad;
                    if (first < 0) for (size_t e = 0; e < hr.size(); ++e) if (memcmp(&hr[e], &ho[e], 4)) { first = (long)e; break; }
                }
            }
            printf("victim %s beside %-24s: %3d / %d outputs differ", vn[v], sn[s], bad, iters);
            if (first >= 0) printf("  (first: query %ld, channel %ld -> lane %ld)", first / 512, first % 512, (first % 256) / 4);
            printf("\n");
        }
    CK(hipGetLastError());
    return 0;
}
