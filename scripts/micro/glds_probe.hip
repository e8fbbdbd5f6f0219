// Dev probe (round 3): LDS-DMA destination addressing on gfx950 -- how many bits of M0 count, what masked lanes do.
// hipcc --offload-arch=gfx950 -O2 -o glds_probe glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(64) void probe(const float4 *src, const uint32_t *src4, float4 *out, uint32_t *out4, int noffs, const uint32_t *offs)
{
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x;
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    for (int i = lane; i < 150 * 1024 / 16; i += 64) lds[i] = make_float4(-1.f, -1.f, -1.f, -1.f);
    __syncthreads();
    for (int k = 0; k < noffs; ++k) {
        if (lane % 3 != 1) glds16(src + k * 64 + lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + offs[k])));          // lanes 1, 4, 7, ... masked
        glds4(src4 + k * 64 + lane, (uint32_t)__builtin_amdgcn_readfirstlane((int)(base + offs[k] + 2048)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < noffs; ++k) {
        out[k * 64 + lane] = lds[offs[k] / 16 + lane];
        out4[k * 64 + lane] = reinterpret_cast<uint32_t *>(lds)[(offs[k] + 2048) / 4 + lane];
    }
    if (lane == 0) out4[noffs * 64] = base;
}

int main()
{
    const int noffs = 6;
    uint32_t offs_h[noffs] = {0, 16 * 1024, 60 * 1024, 68 * 1024, 100 * 1024, 140 * 1024};
    std::vector<float4> src_h(noffs * 64);
    std::vector<uint32_t> src4_h(noffs * 64);
    for (int i = 0; i < noffs * 64; ++i) { src_h[i] = make_float4((float)i, 0.5f, 0.25f, 7.f); src4_h[i] = 1000u + i; }
    float4 *src, *out; uint32_t *src4, *out4, *offs;
    CK(hipMalloc((void **)&src, src_h.size() * 16)); CK(hipMalloc((void **)&out, src_h.size() * 16));
    CK(hipMalloc((void **)&src4, src4_h.size() * 4)); CK(hipMalloc((void **)&out4, src4_h.size() * 4 + 4)); CK(hipMalloc((void **)&offs, sizeof(offs_h)));
    CK(hipMemcpy(src, src_h.data(), src_h.size() * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(src4, src4_h.data(), src4_h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(offs, offs_h, sizeof(offs_h), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 150 * 1024, 0, src, src4, out, out4, noffs, offs);
    CK(hipDeviceSynchronize());
    std::vector<float4> out_h(noffs * 64); std::vector<uint32_t> out4_h(noffs * 64 + 1);
    CK(hipMemcpy(out_h.data(), out, out_h.size() * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out4_h.data(), out4, out4_h.size() * 4, hipMemcpyDeviceToHost));
    printf("dynamic LDS base offset %u\n", out4_h[noffs * 64]);
    for (int k = 0; k < noffs; ++k) {
        int ok16 = 0, masked_kept = 0, ok4 = 0;
        for (int l = 0; l < 64; ++l) {
            const float4 v = out_h[k * 64 + l];
            if (l % 3 != 1) ok16 += (v.x == (float)(k * 64 + l) && v.w == 7.f);
            else masked_kept += (v.x == -1.f);
            ok4 += out4_h[k * 64 + l] == 1000u + k * 64 + l;
        }
        printf("offset %6u: x4 active lanes ok %2d/43, masked lanes untouched %2d/21, dword ok %2d/64   (lane0 x=%g, lane2 x=%g)\n", offs_h[k], ok16, masked_kept, ok4,
               out_h[k * 64].x, out_h[k * 64 + 2].x);
    }
    return 0;
}
