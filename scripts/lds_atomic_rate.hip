// development microbenchmark: LDS read-modify-write rates per CU (one 8-wave workgroup per CU, every wave on its own 4 columns of a 32 x 272 image, as the fused
// Schur kernel's build phase) and the dependent-chain rate of v_mfma_f64_16x16x4_f64.   hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_rate.hip -o /tmp/lar && /tmp/lar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(512) void k_rmw(unsigned long long* out, int iters)
{
    extern __shared__ double s_u[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < 32 * 272; j += 512) s_u[j] = 0.0;
    __syncthreads();
    double* col = s_u + 4 * w * 272;
    // 48 active lanes: 8 "records" x 6 lanes, rows pseudo-random inside a 144-row band, two columns
    const int r8 = lane >> 3, c8 = lane & 7;
    unsigned s = 1234567u + 977u * threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int base = (int)((s >> 8) % 138u) + ((r8 * 17) % 7);
        const int row = base + ((c8 < 3) ? 2 * c8 : 2 * c8 - 6);
        const double v = 1.0 + it;
        if (c8 < 6) {
            if (MODE == 0) { atomicAdd(&col[row], v); atomicAdd(&col[272 + row], v); atomicAdd(&col[row + 1], v); atomicAdd(&col[272 + row + 1], v); }
            if (MODE == 1) { unsigned long long* c = (unsigned long long*)col; atomicAdd(&c[row], (unsigned long long)it); atomicAdd(&c[272 + row], (unsigned long long)it); atomicAdd(&c[row + 1], (unsigned long long)it); atomicAdd(&c[272 + row + 1], (unsigned long long)it); }
            if (MODE == 2) { unsigned* c = (unsigned*)col; atomicAdd(&c[row], (unsigned)it); atomicAdd(&c[544 + row], (unsigned)it); atomicAdd(&c[row + 1], (unsigned)it); atomicAdd(&c[544 + row + 1], (unsigned)it); }
            if (MODE == 3) { float* c = (float*)col; atomicAdd(&c[row], (float)v); atomicAdd(&c[544 + row], (float)v); atomicAdd(&c[row + 1], (float)v); atomicAdd(&c[544 + row + 1], (float)v); }
            if (MODE == 4) { col[row] += v; col[272 + row] += v; col[row + 1] += v; col[272 + row + 1] += v; }      // (racy on purpose: the cost of plain read + write)
            if (MODE == 5) { double2* c2 = (double2*)col; double2 a = c2[row]; a.x += v; a.y += v; c2[row] = a; double2 b = c2[136 + row]; b.x += v; b.y += v; c2[136 + row] = b; }   // 16-B RMW: two columns interleaved
        }
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[w] = t1 - t0;
    if (s_u[threadIdx.x] == 12345.678) out[100] = 1;
}
template <int CHAINS> __global__ __launch_bounds__(1024) void k_mfma(unsigned long long* out, double* sink, int iters)
{
    double4_t acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int k = 0; k < CHAINS; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    sink[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
int main()
{
    unsigned long long* d; double* sink; hipMalloc(&d, 1024); hipMalloc(&sink, 256 * 1024 * 8);
    const int iters = 2000; unsigned long long h[8];
    const char* names[] = {"ds_add_f64 x4", "ds_add_u64 x4", "ds_add_u32 x4", "ds_add_f32 x4", "plain rmw f64 x4", "plain rmw 16 B x2"};
    hipFuncSetAttribute((const void*)k_rmw<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    hipFuncSetAttribute((const void*)k_rmw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    hipFuncSetAttribute((const void*)k_rmw<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    hipFuncSetAttribute((const void*)k_rmw<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    hipFuncSetAttribute((const void*)k_rmw<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    hipFuncSetAttribute((const void*)k_rmw<5>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8);
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k_rmw<0>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            if (m == 1) hipLaunchKernelGGL(k_rmw<1>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            if (m == 2) hipLaunchKernelGGL(k_rmw<2>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            if (m == 3) hipLaunchKernelGGL(k_rmw<3>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            if (m == 4) hipLaunchKernelGGL(k_rmw<4>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            if (m == 5) hipLaunchKernelGGL(k_rmw<5>, dim3(256), dim3(512), 32 * 272 * 8, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-20s: %8.1f cycles per round of 8 records per wave (8 waves per CU) -> %6.2f CU-cycles per lane-op (192 lane-ops per round and wave)\n", names[m], (double)h[0] / iters, (double)h[0] / iters / (8 * 192.0) * 8);
    }
    for (int wps = 1; wps <= 4; wps *= 2)
    for (int c = 1; c <= 8; c *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            if (c == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 4000);
            if (c == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 4000);
            if (c == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 4000);
            if (c == 8) hipLaunchKernelGGL(k_mfma<8>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 4000);
            hipDeviceSynchronize();
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        if (c == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 40000);
        if (c == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 40000);
        if (c == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 40000);
        if (c == 8) hipLaunchKernelGGL(k_mfma<8>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 40000);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        const double nm = 256.0 * 4 * wps * 40000.0 * c;
        printf("v_mfma_f64_16x16x4, %d accumulators per wave, %d wave(s) per SIMD: wave 0 sees %6.1f ticks per own MFMA; launch %.3f ms = %.1f TFLOP/s = %.1f ns per MFMA and SIMD\n", c, wps, (double)h[0] / 40000 / c, ms, nm * 2048 / ms * 1e-9, ms * 1e6 / (nm / 1024));
    }
    return 0;
}
