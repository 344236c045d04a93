// Probe (GPU box): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that the LDS-DMA GEMM staging relies on:
//   1. destination = M0 base + 16 * lane (lane-linear), whatever the per-lane source offsets are;
//   2. an out-of-range source offset (beyond num_records) WRITES ZEROS into the lane's LDS slot (it must not leave the old bytes).
// hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_probe.hip -o /tmp/lds_dma_probe && /tmp/lds_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* in, unsigned nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float s[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += 256) s[i] = -7.f;      // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // source: lane l of wave w fetches 16-byte chunk (63 - l) + 64 w (reversed), every third lane out of range
    unsigned off = (unsigned)((63 - lane) + 64 * wave) * 16u;
    if (lane % 3 == 1) off = 0xFFFFFFFFu;
    if (lane % 3 == 2 && wave == 1) off = nbytes + 16u * lane;          // just beyond the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s + wave * 256), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 256; i += 256) out[i] = s[i];
}
int main() {
    const int n = 4 * 256;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    float *din, *dout;
    hipMalloc(&din, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, (unsigned)(n * 4), dout);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const bool oob = (l % 3 == 1) || (l % 3 == 2 && w == 1);
                const float want = oob ? 0.f : h[((63 - l) + 64 * w) * 4 + e];
                const float got = o[w * 256 + l * 4 + e];
                if (got != want) { if (bad < 8) printf("wave %d lane %d e %d: got %g want %g\n", w, l, e, got, want); ++bad; }
            }
    printf("lds_dma_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK lane-linear destination, OOB lanes zero-filled", bad);
    return bad != 0;
}
