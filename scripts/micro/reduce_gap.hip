// Dev micro-benchmark: what a streaming read-and-reduce kernel of 32 MB costs when it follows (a) itself (clean caches),
// (b) a kernel that streamed the same 32 MB out, (c) a kernel that scattered it in 16-byte pieces -- and how the time
// depends on the bytes (fixed cost vs bandwidth).   hipcc --offload-arch=gfx950 -O2 -o reduce_gap reduce_gap.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write_stream(double2 *a, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) a[i] = make_double2((double)i, 1.0);
}

__global__ __launch_bounds__(256) void k_write_scatter(double2 *a, long n)       // n = power of two
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const long j = (i * 2654435761l + 12345l) & (n - 1);                      // odd multiplier: a permutation
        a[j] = make_double2((double)i, 1.0);
    }
}

template <int PER>
__global__ __launch_bounds__(256) void k_reduce(const double2 *__restrict__ a, long n, double *out)
{
    __shared__ double ls[4];
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) / 8 * (8 * PER) + (threadIdx.x & 7);
    double2 v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = (base + 8 * j < n) ? a[base + 8 * j] : make_double2(0.0, 0.0);
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += v[j].x + v[j].y * v[j].y;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) ls[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ls[0] + ls[1] + ls[2] + ls[3];
}

static float median(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long nmax = 1l << 24;                                                    // 16 M records of 16 bytes = 256 MB
    double2 *a;
    double *out_dev, *out_host;
    CK(hipMalloc((void **)&a, nmax * sizeof(double2)));
    CK(hipMalloc((void **)&out_dev, 1 << 20));
    CK(hipHostMalloc((void **)&out_host, 1 << 20, hipHostMallocDefault));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (long n : {1l << 19, 1l << 20, 1l << 21, 1l << 22, 1l << 24}) {
        const unsigned wb = (unsigned)((n + 255) / 256), rb = (unsigned)((n / 16 + 255) / 256);
        for (int prev = 0; prev < 4; ++prev) {
            for (int host = 0; host < 2; ++host) {
                std::vector<float> t;
                for (int rep = 0; rep < 15; ++rep) {
                    if (prev == 1) hipLaunchKernelGGL(k_write_stream, dim3(wb), dim3(256), 0, s, a, n);
                    if (prev == 2) hipLaunchKernelGGL(k_write_scatter, dim3(wb), dim3(256), 0, s, a, n);
                    if (prev == 0) hipLaunchKernelGGL((k_reduce<16>), dim3(rb), dim3(256), 0, s, a, n, out_dev);
                    if (prev == 3) CK(hipStreamSynchronize(s));                   // idle GPU before the measured kernel
                    CK(hipEventRecord(e0, s));
                    hipLaunchKernelGGL((k_reduce<16>), dim3(rb), dim3(256), 0, s, a, n, host ? out_host : out_dev);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms * 1000.f);
                }
                printf("%6.0f MB  after %-14s out to %-6s : %7.1f us (events)\n", n * 16.0 / 1e6,
                       prev == 0 ? "the same read" : prev == 1 ? "streamed write" : prev == 2 ? "scattered write" : "an idle stream", host ? "host" : "device",
                       median(t));
            }
        }
    }
    return 0;
}
