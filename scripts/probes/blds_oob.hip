// probe: does `buffer_load_dwordx4 ... lds` write ZEROS into LDS for lanes whose offset is out of range?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* __restrict__ g, float* out, unsigned nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x & 1) off = 0xfffffff0u;
    if ((threadIdx.x & 3) == 2) off = nbytes + 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *g, *o; float h[256];
    hipMalloc(&g, 1024); hipMalloc(&o, 1024);
    for (int i = 0; i < 256; ++i) h[i] = i + 1;
    hipMemcpy(g, h, 1024, hipMemcpyHostToDevice);
    k<<<1, 64>>>(g, o, 1024);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        float want = (l & 1) || ((l & 3) == 2) ? 0.0f : (float)(4 * l + j + 1);
        if (h[4*l+j] != want) ++bad;
    }
    printf("OOB lanes zero-filled: %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    return 0;
}
