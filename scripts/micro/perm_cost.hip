// Dev micro-benchmark (round 3): what does a random permutation of N 16-byte result records cost on MI355X, as
//  (a) scattered 16-byte stores from a streaming kernel (what k_brick_query's row-order result stores are),
//  (b) a gather in a streaming reduce-like kernel (value[perm[i]], perm read coalesced), PER loads in flight per lane,
//  (c) the same gather of 8-byte values, (d) scattered 4-byte stores, (e) coalesced baseline of both.
// hipcc --offload-arch=gfx950 -O2 -o perm_cost perm_cost.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k_scatter(T *__restrict__ out, const int *__restrict__ perm, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        T v;
        __builtin_memset(&v, 0, sizeof(T));
        *reinterpret_cast<int *>(&v) = (int)i;
        out[perm[i]] = v;
    }
}

template <typename T, int PER>
__global__ __launch_bounds__(256) void k_gather(const T *__restrict__ in, const int *__restrict__ perm, long n, double *out)
{
    __shared__ double ls[4];
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) / 8 * (8 * PER) + (threadIdx.x & 7);
    int p[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) p[j] = (base + 8 * j < n) ? perm[base + 8 * j] : 0;
    T v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = in[p[j]];
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += *reinterpret_cast<const double *>(&v[j]);
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
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (long n : {1l << 20, 1l << 21, 1l << 23, 1l << 24}) {
        std::vector<int> ident(n), rnd(n), local(n);
        std::iota(ident.begin(), ident.end(), 0);
        rnd = ident;
        std::mt19937_64 g(1234);
        std::shuffle(rnd.begin(), rnd.end(), g);
        // "local": random inside windows of 4096 rows (what a spatially coherent order looks like to a brick)
        local = ident;
        for (long w = 0; w + 4096 <= n; w += 4096) std::shuffle(local.begin() + w, local.begin() + w + 4096, g);
        int *perm;
        double2 *a16;
        double *a8, *out;
        CK(hipMalloc((void **)&perm, n * sizeof(int)));
        CK(hipMalloc((void **)&a16, n * sizeof(double2)));
        CK(hipMalloc((void **)&a8, n * sizeof(double)));
        CK(hipMalloc((void **)&out, 1 << 20));
        CK(hipMemset(a16, 0, n * sizeof(double2)));
        CK(hipMemset(a8, 0, n * sizeof(double)));
        const unsigned wb = (unsigned)((n + 255) / 256);
        struct { const char *name; std::vector<int> *p; } orders[3] = {{"identity", &ident}, {"window4096", &local}, {"random", &rnd}};
        for (auto &o : orders) {
            CK(hipMemcpy(perm, o.p->data(), n * sizeof(int), hipMemcpyHostToDevice));
            for (int what = 0; what < 7; ++what) {
                std::vector<float> t;
                for (int rep = 0; rep < 11; ++rep) {
                    CK(hipEventRecord(e0, s));
                    switch (what) {
                    case 0: hipLaunchKernelGGL((k_scatter<double2>), dim3(wb), dim3(256), 0, s, a16, perm, n); break;
                    case 1: hipLaunchKernelGGL((k_scatter<double>), dim3(wb), dim3(256), 0, s, a8, perm, n); break;
                    case 2: hipLaunchKernelGGL((k_scatter<int>), dim3(wb), dim3(256), 0, s, (int *)a8, perm, n); break;
                    case 3: hipLaunchKernelGGL((k_gather<double2, 16>), dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, s, a16, perm, n, out); break;
                    case 4: hipLaunchKernelGGL((k_gather<double2, 8>), dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, a16, perm, n, out); break;
                    case 5: hipLaunchKernelGGL((k_gather<double2, 4>), dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a16, perm, n, out); break;
                    case 6: hipLaunchKernelGGL((k_gather<double, 16>), dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, s, a8, perm, n, out); break;
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t.push_back(ms * 1000.f);
                }
                static const char *names[7] = {"scatter 16 B", "scatter 8 B", "scatter 4 B", "gather 16 B x16", "gather 16 B x8", "gather 16 B x4", "gather 8 B x16"};
                printf("n %8ld  %-10s  %-16s : %7.1f us\n", n, o.name, names[what], median(t));
            }
        }
        CK(hipFree(perm)); CK(hipFree(a16)); CK(hipFree(a8)); CK(hipFree(out));
    }
    return 0;
}
