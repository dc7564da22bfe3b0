// Stand-alone probe: what cache-policy bits on global stores do to the streaming WRITE rate of an MI355X (gfx950).
// The f16x2 GEMMs' plane outputs (w_1: 268 MB per launch) leave at the plain-store rate (bench.py hbm_copy_probe.own_write_GBps, 4.4 TB/s);
// this asks whether nt / sc0 / sc1 stores are faster, at the GEMM's output size and at 2 GiB, as a pure stream and in the GEMM's
// shape of stores (every lane 16 B of a 256-B run, rows 4 KB apart).
//   hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ST(POL) asm volatile("global_store_dwordx4 %0, %1, off " POL : : "v"(p), "v"(v) : "memory")
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ void store16(u4* p, u4 v) {
    if constexpr (POL == 0) ST("");
    else if constexpr (POL == 1) ST("nt");
    else if constexpr (POL == 2) ST("sc1");
    else if constexpr (POL == 3) ST("sc0 sc1");
    else if constexpr (POL == 4) ST("sc0");
    else if constexpr (POL == 5) ST("nt sc1");
    else ST("nt sc0 sc1");
}

// SHAPE 0: grid-stride stream. SHAPE 1: tile order -- workgroup b owns 256 rows x 256 B of a [rows][4096 B] matrix (the plane tile of a
// 256 x 128 GEMM block): lane = 16 B of a 256-B run, 16 lanes a run, a wave 4 rows per store, 64 stores per wave.
template <int POL, int SHAPE>
__global__ __launch_bounds__(256) void store_kernel(u4* dst, size_t n16) {
    const u4 v = {(unsigned)threadIdx.x, blockIdx.x, 2u, 3u};
    if constexpr (SHAPE == 0) {
        const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
        for (size_t i = tid; i < n16; i += nt) store16<POL>(dst + i, v);
    } else {
        const size_t rows = n16 / 256;                 // 4096-B rows
        const size_t tiles_n = 16;                     // 4096 / 256
        const size_t tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
        if (tm * 256 >= rows) return;
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int s = 0; s < 16; ++s) {
            const size_t row = tm * 256 + wave * 64 + s * 4 + (lane >> 4);
            store16<POL>(dst + row * 256 + tn * 16 + (lane & 15), v);
        }
    }
}

template <int POL, int SHAPE>
static float run(u4* dst, size_t n16, int launches) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned blocks = SHAPE == 0 ? 8192u : (unsigned)(n16 / 256 / 256 * 16);
    float best = 1e30f;
    for (int l = 0; l <= launches; ++l) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((store_kernel<POL, SHAPE>), dim3(blocks), dim3(256), 0, 0, dst, n16);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (l && ms < best) best = ms;
    }
    return best;
}

int main() {
    const char* names[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc0", "nt sc1", "nt sc0 sc1"};
    u4* buf; const size_t big = (size_t)1 << 31;
    if (hipMalloc(&buf, big) != hipSuccess) return 1;
    for (size_t bytes : {(size_t)268435456, big}) {
        const size_t n16 = bytes / 16;
        float ms[2][7];
#define ROW(P) ms[0][P] = run<P, 0>(buf, n16, 10); ms[1][P] = run<P, 1>(buf, n16, 10);
        ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6)
        for (int s = 0; s < 2; ++s)
            for (int p = 0; p < 7; ++p)
                printf("{\"bytes\": %zu, \"shape\": \"%s\", \"policy\": \"%s\", \"ms\": %.4f, \"GBps\": %.0f}\n", bytes, s ? "gemm tile" : "stream", names[p],
                       ms[s][p], bytes / ms[s][p] / 1e6);
    }
    hipFree(buf);
    return 0;
}
