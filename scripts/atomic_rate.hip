#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// N lanes-groups of 6 lanes add 6 doubles to one random 64-B line (like the warp kernel's per-pixel sums): request rate by atomic scope
template <int SCOPE>
__global__ void k(double* acc, long nlines, long n_req)
{
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x);
    const long req = g / 6; const int comp = (int)(g % 6);
    if (req >= n_req) return;
    unsigned long long h = (unsigned long long)req * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    double* p = acc + 8 * (long)(h % (unsigned long long)nlines) + comp;
    if (SCOPE == 0) atomicAdd(p, 1.0);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <int SCOPE> void run(const char* name, double* acc, long nlines, long n_req)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const long threads = n_req * 6;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<SCOPE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, acc, nlines, n_req);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep == 2) printf("%-28s %8.1f us  %6.2f G requests/s\n", name, ms * 1e3, n_req / (ms * 1e-3) / 1e9);
    }
}
int main()
{
    const long nlines = 2L * 1024 * 1024, n_req = 675000;    // 2 M pixels x 64 B, 675 k requests (the BASELINE workload's inliers)
    double* acc; hipMalloc(&acc, nlines * 64); hipMemset(acc, 0, nlines * 64);
    run<0>("atomicAdd (default)", acc, nlines, n_req);
    run<1>("agent scope", acc, nlines, n_req);
    run<2>("workgroup scope", acc, nlines, n_req);
    run<3>("wavefront scope", acc, nlines, n_req);
    const long big = 20000000;
    run<0>("atomicAdd, 20 M requests", acc, nlines, big);
    run<2>("workgroup, 20 M requests", acc, nlines, big);
    return 0;
}
