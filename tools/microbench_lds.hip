// microbench_lds.hip -- measures the primitives the counting kernels are built from on gfx950:
// streaming HBM reads, random LDS reads / atomics (independent and as a dependent chain).
// Build: hipcc -O3 --offload-arch=gfx950 microbench_lds.hip -o microbench_lds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32; typedef uint64_t u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_hbm_read(const uint4* __restrict__ p, u64 n16, u32* out) {
    u32 acc = 0;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { uint4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

// MODE 0: ds_add_u32 no return, independent random addresses
// MODE 1: ds_add_rtn_u32, next address depends on the returned value (chain)
// MODE 2: ds_read_b64 (dependent chain) + ds_add_u32 on hit
// MODE 3: ds_read_b32 dependent chain only
template <int MODE, int SLOTS>
__global__ void k_lds(u32 iters, u32* out) {
    __shared__ u64 tab[SLOTS];
    __shared__ u32 cnt[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) { tab[i] = ((u64)(i * 2654435761u) << 32) | (u32)(i * 40503u + 17u); cnt[i] = i * 7919u; }
    __syncthreads();
    u32 x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32 acc = 0;
    for (u32 it = 0; it < iters; ++it) {
        if (MODE == 0) {
            x = x * 1664525u + 1013904223u;
            atomicAdd(&cnt[(x >> 8) & (SLOTS - 1)], 1u);
        } else if (MODE == 1) {
            u32 old = atomicAdd(&cnt[(x >> 8) & (SLOTS - 1)], 1u);
            x = (x ^ old) * 1664525u + 1013904223u;
        } else if (MODE == 2) {
            u32 h = (x >> 8) & (SLOTS - 1);
            u64 e = tab[h];
            if ((u32)e != 0xdeadbeefu) atomicAdd(&cnt[h], 1u);
            x = (x ^ (u32)(e >> 32)) * 1664525u + 1013904223u;
        } else {
            u32 v = cnt[(x >> 8) & (SLOTS - 1)];
            x = (x ^ v) * 1664525u + 1013904223u;
        }
    }
    acc = x;
    __syncthreads();
    if (acc == 0x12345678u) out[0] = cnt[threadIdx.x & (SLOTS - 1)];
}

template <int MODE, int SLOTS>
int run_lds(const char* name, int threads, int blocks_per_cu, u32* d_out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 iters = 20000;
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k_lds<MODE, SLOTS>), dim3(grid), dim3(threads), 0, 0, 100u, d_out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_lds<MODE, SLOTS>), dim3(grid), dim3(threads), 0, 0, iters, d_out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double ops = (double)grid * threads * iters;
    printf("%-34s thr=%4d blk/CU=%d slots=%5d : %8.3f ms  %7.2f Tops/s  %6.2f lane-ops/clk/CU(@2.4GHz)\n", name, threads, blocks_per_cu, SLOTS, ms,
           ops / ms / 1e9, ops / (ms * 1e-3) / 256 / 2.4e9);
    return 0;
}

int main() {
    u32* d_out; CK(hipMalloc(&d_out, 4096));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz LDS/block %zu\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
    // HBM read
    {
        u64 bytes = 8ull << 30;
        uint4* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
            hipLaunchKernelGGL(k_hbm_read, dim3(grid), dim3(256), 0, 0, p, bytes / 16, d_out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_hbm_read, dim3(grid), dim3(256), 0, 0, p, bytes / 16, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("hbm_read uint4 grid=%5d : %.3f ms/pass  %.2f TB/s\n", grid, ms / 5, bytes * 5.0 / ms / 1e9);
        }
        CK(hipFree(p));
    }
    for (int bpc : {1, 2}) {
        run_lds<0, 4096>("ds_add_u32 indep random", 1024, bpc, d_out);
        run_lds<1, 4096>("ds_add_rtn_u32 dependent chain", 1024, bpc, d_out);
        run_lds<2, 4096>("ds_read_b64 chain + ds_add", 1024, bpc, d_out);
        run_lds<3, 4096>("ds_read_b32 dependent chain", 1024, bpc, d_out);
    }
    run_lds<0, 4096>("ds_add_u32 indep random", 256, 1, d_out);
    run_lds<1, 4096>("ds_add_rtn_u32 dependent chain", 256, 1, d_out);
    run_lds<2, 4096>("ds_read_b64 chain + ds_add", 256, 1, d_out);
    run_lds<2, 4096>("ds_read_b64 chain + ds_add", 512, 1, d_out);
    run_lds<0, 512>("ds_add_u32 indep random", 1024, 2, d_out);
    run_lds<2, 512>("ds_read_b64 chain + ds_add", 1024, 2, d_out);
    run_lds<1, 512>("ds_add_rtn_u32 dependent chain", 1024, 2, d_out);
    return 0;
}
