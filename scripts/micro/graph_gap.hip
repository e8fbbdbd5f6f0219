// Dev micro-benchmark: what one kernel node costs inside a replayed hipGraph as a function of its kernarg size
// and of what the kernel touches (nothing / a global atomic / pinned host memory).
// hipcc --offload-arch=gfx950 -O2 -o graph_gap graph_gap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int WORDS> struct Arg { unsigned long long w[WORDS]; };

template <int WORDS, int MODE>
__global__ void k_node(Arg<WORDS> a, unsigned int *dev, unsigned int *host)
{
    if (MODE == 1 && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(dev, 1u);
    if (MODE == 2 && threadIdx.x == 0 && blockIdx.x == 0) host[0] = (unsigned int)a.w[0];
    if (a.w[WORDS - 1] == 0x1234567ull) dev[1] = 1u;       // keeps the argument alive
}

template <int WORDS, int MODE>
static int run(const char *label, hipStream_t s, unsigned int *dev, unsigned int *host, int nodes, int blocks)
{
    Arg<WORDS> a = {};
    // eager
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_node<WORDS, MODE>), dim3(blocks), dim3(256), 0, s, a, dev, host);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 200;
    for (int r = 0; r < reps; ++r)
        for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL((k_node<WORDS, MODE>), dim3(blocks), dim3(256), 0, s, a, dev, host);
    CK(hipStreamSynchronize(s));
    double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * nodes);
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL((k_node<WORDS, MODE>), dim3(blocks), dim3(256), 0, s, a, dev, host);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * nodes);
    // graph, one launch + sync at a time (what a step does)
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
    double graph_sync = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf("%-28s kernarg %5zu B  blocks %5d  eager %6.2f us/kernel  graph %6.2f us/node  graph+sync %7.1f us per %d-node launch\n", label,
           sizeof(a) + 16, blocks, eager, graph, graph_sync, nodes);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return 0;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned int *dev, *host;
    CK(hipMalloc((void **)&dev, 256));
    CK(hipMemset(dev, 0, 256));
    CK(hipHostMalloc((void **)&host, 256, hipHostMallocDefault));
    const int nodes = 8;
    for (int blocks : {1, 512, 4096}) {
        if (run<4, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<32, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<46, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<62, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<66, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<88, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<136, 0>("empty", s, dev, host, nodes, blocks)) return 1;
        if (run<4, 1>("global atomic", s, dev, host, nodes, blocks)) return 1;
        if (run<88, 1>("global atomic", s, dev, host, nodes, blocks)) return 1;
        if (run<4, 2>("pinned host store", s, dev, host, nodes, blocks)) return 1;
        if (run<88, 2>("pinned host store", s, dev, host, nodes, blocks)) return 1;
    }
    return 0;
}
