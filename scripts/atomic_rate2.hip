#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// Round 4: what a memory-side request costs by KIND, at the BASELINE workload's shape (675 k requests into 2 M lines of 64 B, drawn from `distinct` lines).
//   0: 6 x fp64 atomic add per request (the warp kernel's per-pixel sums)      1: 1 x int32 atomic add per request
//   2: 1 x fp64 atomic add per request                                          3: 6 x fp64 plain store per request
//   4: 1 x int32 plain store per request (the marker)                           5: 1 x int32 atomic add WITH return per request
//   6: 1 x fp64 load (8 B gather) per request                                   7: 128-B line gather (8 lanes x 16 B) per request
__device__ __forceinline__ long line_of(long req, long distinct, long nlines)
{
    unsigned long long h = (unsigned long long)(req % distinct) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return (long)(h % (unsigned long long)nlines);
}
template <int MODE>
__global__ void k(double* acc, long nlines, long n_req, long distinct, double* sink)
{
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x);
    constexpr int LPR = (MODE == 0 || MODE == 3) ? 6 : (MODE == 7 ? 8 : 1);     // lanes per request
    const long req = g / LPR; const int comp = (int)(g % LPR);
    if (req >= n_req) return;
    const long ln = line_of(req, distinct, nlines);
    double* p = acc + 8 * ln + comp;
    if (MODE == 0 || MODE == 2) atomicAdd(p, 1.0);
    else if (MODE == 1) atomicAdd(reinterpret_cast<int*>(p), 1);
    else if (MODE == 3) *p = 1.0;
    else if (MODE == 4) *reinterpret_cast<int*>(p) = 1;
    else if (MODE == 5) { const int old = atomicAdd(reinterpret_cast<int*>(p), 1); if (old == 0x7FFFFFFF) sink[0] = 1.0; }
    else if (MODE == 6) { const double v = *p; if (v == 1.2345e300) sink[0] = v; }
    else { const double2 v = reinterpret_cast<const double2*>(acc + 16 * (ln >> 1))[comp]; if (v.x == 1.2345e300) sink[0] = v.y; }
}
template <int MODE> void run(const char* name, double* acc, long nlines, long n_req, long distinct, double* sink)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    constexpr int LPR = (MODE == 0 || MODE == 3) ? 6 : (MODE == 7 ? 8 : 1);
    const long threads = n_req * LPR;
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, acc, nlines, n_req, distinct, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep >= 2 && ms < best) best = ms;
    }
    printf("%-44s n=%9ld distinct=%8ld  %8.1f us  %6.2f G requests/s\n", name, n_req, distinct, best * 1e3, n_req / (best * 1e-3) / 1e9);
}
__global__ void empty_k() {}
int main()
{
    const long nlines = 2L * 1024 * 1024;
    double* acc; hipMalloc(&acc, nlines * 64); hipMemset(acc, 0, nlines * 64);
    double* sink; hipMalloc(&sink, 64);
    {   // an empty launch between two events: what the numbers below include
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 4; ++rep) { hipEventRecord(a); hipLaunchKernelGGL(empty_k, dim3(4096), dim3(256), 0, 0); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (rep == 3) printf("empty launch, 4096 blocks: %.1f us\n", ms * 1e3); }
    }
    for (long n_req : {675000L, 6750000L}) {
        for (long distinct : {n_req, 143000L}) {
            run<0>("6 x fp64 atomic add (pixacc)", acc, nlines, n_req, distinct, sink);
            run<2>("1 x fp64 atomic add", acc, nlines, n_req, distinct, sink);
            run<1>("1 x int32 atomic add", acc, nlines, n_req, distinct, sink);
            run<5>("1 x int32 atomic add, returning", acc, nlines, n_req, distinct, sink);
            run<3>("6 x fp64 plain store", acc, nlines, n_req, distinct, sink);
            run<4>("1 x int32 plain store (marker)", acc, nlines, n_req, distinct, sink);
            run<6>("1 x fp64 load (8-B gather)", acc, nlines, n_req, distinct, sink);
            run<7>("128-B line gather (8 lanes x 16 B)", acc, nlines, n_req, distinct, sink);
        }
    }
    return 0;
}
