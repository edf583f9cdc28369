// HBM streaming yardstick for the Winograd input transforms: read R bytes, write Wr bytes (16 B per lane, fully coalesced),
// nothing else.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_stream.hip -o gpurun_out/hbm_stream ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// each thread reads one 16-byte chunk and writes `ratio4` / 4 chunks (ratio 2.25 -> 9 writes per 4 reads)
__global__ __launch_bounds__(256) void stream_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long n_in, int w_per_r4) {
    const long i0 = (blockIdx.x * 256L + threadIdx.x) * 4;
    if (i0 + 3 >= n_in) return;
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = in[i0 + j * 0 + (long)j * 1];
    const long nthr = (long)gridDim.x * 256;
    const long t = blockIdx.x * 256L + threadIdx.x;
    for (int w = 0; w < w_per_r4; ++w) out[(long)w * nthr + t] = v[w & 3] * (float)(w + 1);
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = in[i];
}
int main() {
    const long in_bytes = 132710400L, n_in = in_bytes / 16;          // 5 x 120 x 216 x 256 floats
    const long out_bytes = in_bytes / 4 * 9;                          // 2.25x
    f32x4 *in, *out;
    hipMalloc(&in, in_bytes); hipMalloc(&out, out_bytes + (1 << 20));
    hipMemset(in, 1, in_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((n_in / 4 + 255) / 256);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(256), 0, 0, in, out, n_in, 9);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read 1 : write 2.25 (%.0f + %.0f MB): %.1f us per pass = %.2f TB/s\n", in_bytes / 1e6, out_bytes / 1e6, ms * 100, (in_bytes + out_bytes) / (ms / 10 * 1e-3) / 1e12);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, in, out, n_in);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("copy 1 : 1 (%.0f + %.0f MB): %.1f us per pass = %.2f TB/s\n", in_bytes / 1e6, in_bytes / 1e6, ms * 100, 2.0 * in_bytes / (ms / 10 * 1e-3) / 1e12);
    }
    return 0;
}
