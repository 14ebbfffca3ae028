#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
__global__ void who(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hwid;
    }
    // hold the CU for a while so that blocks spread over all CUs
    long long t0 = clock64(); while (clock64() - t0 < 200000) {}
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("name=%s CUs=%d clock=%d kHz memclk=%d kHz L2=%d B sharedPerBlock=%zu maxSharedPerMP=%zu regsPerBlock=%d wave=%d\n", p.name, p.multiProcessorCount, p.clockRate, p.memoryClockRate,
           p.l2CacheSize, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.warpSize);
    unsigned* d; hipMalloc(&d, 256 * 2 * 4);
    hipLaunchKernelGGL(who, dim3(256), dim3(512), 131072, 0, d);   // same footprint as the GEMM: 1 block per CU
    hipDeviceSynchronize();
    unsigned h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::set<unsigned long long> cus;
    int perx[8] = {0};
    for (int b = 0; b < 256; ++b) { unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; cus.insert(((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu); perx[xcc & 7]++; }
    printf("256 blocks (512 thr, 128 KiB LDS) ran on %zu distinct CUs; per XCC:", cus.size());
    for (int i = 0; i < 8; ++i) printf(" %d", perx[i]);
    printf("\n");
    return 0;
}
