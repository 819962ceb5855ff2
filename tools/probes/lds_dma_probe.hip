// Probe of buffer_load ... lds (LDS-DMA) semantics on gfx950: destination layout, OOB behaviour, imm offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned* p, unsigned* out, int nbytes, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2048];   // 8 KB
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = 0xDEADBEEF;
    __syncthreads();
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int off = lane * 16 + wave * 1024;                   // each lane its own 16 bytes
    if (mode == 1 && (lane & 3) == 1) off = 0x80000000;  // some lanes out of range
    if (mode == 2) off = (63 - lane) * 16 + wave * 1024; // reversed source order
    if (mode == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + wave * 256), 16, off, 0, 2048, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 16384); hipMalloc(&o, 8192);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, o, 16384, mode);
        std::vector<unsigned> r(2048);
        hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
        printf("mode %d:", mode);
        for (int i = 0; i < 24; ++i) printf(" %x", r[i]);
        printf(" | w1:");
        for (int i = 256; i < 264; ++i) printf(" %x", r[i]);
        printf(" | @512:");
        for (int i = 512; i < 520; ++i) printf(" %x", r[i]);
        printf(" | @768:");
        for (int i = 768; i < 776; ++i) printf(" %x", r[i]);
        printf("\n");
    }
    return 0;
}
