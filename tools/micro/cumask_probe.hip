// Where do the blocks of a CU-masked stream run? (measurement infrastructure, round 5)
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/micro/cumask_probe.hip -o tools/micro/cumask_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void probe_kernel(uint32_t* out, int spin) {
    // HW_REG_XCC_ID = 20, HW_REG_HW_ID = 4 (gfx9: cu_id bits 11:8, sh_id 12, se_id 15:13)
    uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    uint32_t hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
extern "C" {
int cumask_stream_create(void** stream_out, const uint32_t* mask, int words) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return 0;
}
int cumask_probe(void* stream, uint32_t* out_dev, int blocks, int spin) {
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out_dev, spin);
    return (int)hipGetLastError();
}
}
