// Micro-benchmark: where do the workgroups of a 1024 x 256 grid land (XCD, shader engine, CU), and which SIMD gets wavefront w?
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/placement.hip -o /tmp/placement && /tmp/placement [grid] [lds bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k(unsigned* hw, unsigned* xcc, int lds_floats)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < lds_floats; i += 256) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        hw[blockIdx.x * 4 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        xcc[blockIdx.x * 4 + wave] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
    // stay resident for a while so that the whole grid is on the chip together
    const long long t0 = clock64();
    while (clock64() - t0 < 200000) __builtin_amdgcn_s_sleep(10);
    if (lds[(threadIdx.x * 7) % lds_floats] < 0) hw[0] = 0;
}

int main(int argc, char** argv)
{
    const int grid = argc > 1 ? atoi(argv[1]) : 1024;
    const int lds_bytes = argc > 2 ? atoi(argv[2]) : 22272;
    unsigned *hw, *xcc;
    CK(hipMalloc(&hw, grid * 4 * 4)); CK(hipMalloc(&xcc, grid * 4 * 4));
    for (int rep = 0; rep < 2; ++rep) {
        k<<<grid, 256, lds_bytes>>>(hw, xcc, lds_bytes / 4);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(grid * 4), x(grid * 4);
    CK(hipMemcpy(h.data(), hw, grid * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(x.data(), xcc, grid * 16, hipMemcpyDeviceToHost));
    std::map<unsigned, std::vector<int>> cu_blocks;
    int simd_pattern[4][4] = {};
    for (int b = 0; b < grid; ++b) {
        const unsigned v = h[b * 4];
        const unsigned cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xc = x[b * 4] & 15;
        cu_blocks[(xc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        for (int w = 0; w < 4; ++w) simd_pattern[w][(h[b * 4 + w] >> 4) & 3]++;
    }
    printf("grid %d, %d B of LDS: %zu distinct CUs\n", grid, lds_bytes, cu_blocks.size());
    printf("wavefront w of a workgroup on SIMD s (counts):\n");
    for (int w = 0; w < 4; ++w) printf("  w%d: %5d %5d %5d %5d\n", w, simd_pattern[w][0], simd_pattern[w][1], simd_pattern[w][2], simd_pattern[w][3]);
    int shown = 0;
    std::map<int, int> per_cu;
    for (auto& kv : cu_blocks) {
        per_cu[(int)kv.second.size()]++;
        if (shown < 12 || shown % 37 == 0) {
            printf("  xcc %u se %u sh %u cu %2u:", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) printf(" %4d(w0 on simd %u)", b, (h[b * 4] >> 4) & 3);
            printf("\n");
        }
        ++shown;
    }
    for (auto& kv : per_cu) printf("CUs with %d workgroups: %d\n", kv.first, kv.second);
    // co-resident workgroups: differences of block indices
    std::map<int, int> diffs;
    for (auto& kv : cu_blocks) { auto v = kv.second; std::sort(v.begin(), v.end()); for (size_t i = 1; i < v.size(); ++i) diffs[v[i] - v[i - 1]]++; }
    printf("differences between the block indices that share a CU:");
    for (auto& kv : diffs) printf(" %d x%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
