// probe: lane <-> element mapping of ds_read_b64_tr_b16 (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    // each lane points at an 8-byte piece: piece index = lane (pieces laid out contiguously)
    const __attribute__((address_space(3))) s4* p = (const __attribute__((address_space(3))) s4*)(lds + threadIdx.x * 4);
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
    short *o; short h[256];
    hipMalloc(&o, 512);
    k<<<1, 64>>>(o);
    hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    // element value v = 4*src_lane + src_elem
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf("  (L%2d,e%d)", h[4*l+e] / 4, h[4*l+e] % 4);
        printf("\n");
    }
    return 0;
}
