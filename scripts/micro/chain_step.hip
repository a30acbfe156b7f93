// Micro-benchmark: what does one step of the biquad feedback recurrence y = (u - a1*y1) - a2*y2 cost on one wavefront of a SIMD?
//   hipcc -O3 -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 scripts/micro/chain_step.hip -o /tmp/chain_step && /tmp/chain_step
// Variants: 0 registers only (the dependent instructions alone), 1 the product's 16-sample register blocks over an LDS row
// (wave_effects_body.hpp chain_biquad), 2 the same with the block buffers alternating (no register moves), 3 as 2 with 32-sample blocks,
// 4 / 5 four samples at a time with one request ahead, unrolled 8 / 16 times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kN = 256;      // steps per row
constexpr int kRow = 4 + kN;

__device__ __forceinline__ void step4(float4& v, float& y1, float& y2, float a1, float a2)
{
    v.x = (v.x - (a1 * y1)) - (a2 * y2);
    v.y = (v.y - (a1 * v.x)) - (a2 * y1);
    v.z = (v.z - (a1 * v.y)) - (a2 * v.x);
    v.w = (v.w - (a1 * v.z)) - (a2 * v.y);
    y2 = v.z;
    y1 = v.w;
}

template <int V>
__global__ __launch_bounds__(256) void k(float* out, long long* ticks, unsigned* simd, int rot, int repeats, int lanes, float a1, float a2)
{
    __shared__ float lds[4][kRow * 8 + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * (kRow * 8 + 4); i += 256) (&lds[0][0])[i] = 1e-3f * (i % 17);
    __syncthreads();
    // one chain wavefront per workgroup; the other three wait at the barrier behind the chain like the product's do
    if (wave != ((blockIdx.x + rot * (blockIdx.x >> 8)) & 3)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    if (lane == 0) simd[blockIdx.x] = (__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 4) & 3;
    float* row = &lds[wave][0] + (lane % 8) * kRow;
    float y1 = 0.1f, y2 = 0.2f;
    const long long t0 = clock64();
    if (lane < lanes) {
        for (int r = 0; r < repeats; ++r) {
            if (V == 0) {
                float4 c0 = make_float4(0.1f, 0.2f, 0.3f, 0.4f);
                for (int i = 0; i < kN; i += 16) {
                    float4 a = c0, b = c0, c = c0, d = c0;
                    step4(a, y1, y2, a1, a2); step4(b, y1, y2, a1, a2); step4(c, y1, y2, a1, a2); step4(d, y1, y2, a1, a2);
                    c0.x = d.w;
                }
            } else if (V == 1) {
                float4* r4 = reinterpret_cast<float4*>(row + 4);
                float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3];
                for (int i = 0; i + 16 <= kN; i += 16) {
                    const int q = i >> 2;
                    float4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
                    if (i + 32 <= kN) { n0 = r4[q + 4]; n1 = r4[q + 5]; n2 = r4[q + 6]; n3 = r4[q + 7]; }
                    step4(c0, y1, y2, a1, a2); step4(c1, y1, y2, a1, a2); step4(c2, y1, y2, a1, a2); step4(c3, y1, y2, a1, a2);
                    r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
                }
            } else if (V == 2) {
                float4* r4 = reinterpret_cast<float4*>(row + 4);
                float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3];
                for (int i = 0; i + 32 <= kN; i += 32) {
                    const int q = i >> 2;
                    float4 n0 = r4[q + 4], n1 = r4[q + 5], n2 = r4[q + 6], n3 = r4[q + 7];
                    step4(c0, y1, y2, a1, a2); step4(c1, y1, y2, a1, a2); step4(c2, y1, y2, a1, a2); step4(c3, y1, y2, a1, a2);
                    r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
                    if (i + 64 <= kN) { c0 = r4[q + 8]; c1 = r4[q + 9]; c2 = r4[q + 10]; c3 = r4[q + 11]; }
                    step4(n0, y1, y2, a1, a2); step4(n1, y1, y2, a1, a2); step4(n2, y1, y2, a1, a2); step4(n3, y1, y2, a1, a2);
                    r4[q + 4] = n0; r4[q + 5] = n1; r4[q + 6] = n2; r4[q + 7] = n3;
                }
            } else if (V == 6 || V == 7) {
                // variant 2 without its stores (6) or without its loads (7): what does each cost?
                float4* r4 = reinterpret_cast<float4*>(row + 4);
                float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3];
                for (int i = 0; i + 32 <= kN; i += 32) {
                    const int q = i >> 2;
                    float4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
                    if (V == 6) { n0 = r4[q + 4]; n1 = r4[q + 5]; n2 = r4[q + 6]; n3 = r4[q + 7]; }
                    step4(c0, y1, y2, a1, a2); step4(c1, y1, y2, a1, a2); step4(c2, y1, y2, a1, a2); step4(c3, y1, y2, a1, a2);
                    if (V == 7) { r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3; }
                    if (V == 6 && i + 64 <= kN) { c0 = r4[q + 8]; c1 = r4[q + 9]; c2 = r4[q + 10]; c3 = r4[q + 11]; }
                    step4(n0, y1, y2, a1, a2); step4(n1, y1, y2, a1, a2); step4(n2, y1, y2, a1, a2); step4(n3, y1, y2, a1, a2);
                    if (V == 7) { r4[q + 4] = n0; r4[q + 5] = n1; r4[q + 6] = n2; r4[q + 7] = n3; c0 = n3; c1 = n2; c2 = n1; c3 = n0; }
                }
            } else if (V == 4 || V == 5) {
                // four samples at a time, one request ahead (reverb.hip biquad_chain), unrolled 8 or 16 times
                float4* r4 = reinterpret_cast<float4*>(row + 4);
                float4 u = r4[0];
#pragma unroll(V == 4 ? 8 : 16)
                for (int q = 0; q < kN / 4; ++q) {
                    const float4 un = r4[q + 1];   // the last one reads the four floats behind the row
                    step4(u, y1, y2, a1, a2);
                    r4[q] = u;
                    u = un;
                }
            } else {
                float4* r4 = reinterpret_cast<float4*>(row + 4);
                float4 c[8], n[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = r4[j];
                for (int i = 0; i + 64 <= kN; i += 64) {
                    const int q = i >> 2;
#pragma unroll
                    for (int j = 0; j < 8; ++j) n[j] = r4[q + 8 + j];
#pragma unroll
                    for (int j = 0; j < 8; ++j) step4(c[j], y1, y2, a1, a2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) r4[q + j] = c[j];
                    if (i + 128 <= kN) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) c[j] = r4[q + 16 + j];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) step4(n[j], y1, y2, a1, a2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) r4[q + 8 + j] = n[j];
                }
            }
        }
    }
    const long long t1 = clock64();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
    if (lane < lanes) out[blockIdx.x * 64 + lane] = y1 + y2 + row[5];
}

template <int V> int run(const char* what, int grid, int lanes, int rot = 0)
{
    float* out; long long* ticks; unsigned* simd;
    CK(hipMalloc(&simd, grid * sizeof(unsigned)));
    CK(hipMalloc(&out, grid * 64 * sizeof(float)));
    CK(hipMalloc(&ticks, grid * sizeof(long long)));
    const int repeats = 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k<V><<<grid, 256>>>(out, ticks, simd, rot, repeats, lanes, -1.2f, 0.5f);
    CK(hipEventRecord(e0));
    k<V><<<grid, 256>>>(out, ticks, simd, rot, repeats, lanes, -1.2f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> t(grid);
    CK(hipMemcpy(t.data(), ticks, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : t) mean += v; mean /= grid;
    const double steps = double(repeats) * kN;
    std::vector<unsigned> sm(grid);
    CK(hipMemcpy(sm.data(), simd, grid * sizeof(unsigned), hipMemcpyDeviceToHost));
    int clash = 0;   // co-resident workgroups (block indices 256 apart) whose chain wavefronts share a SIMD
    for (int b = 0; b + 256 < grid; ++b)
        for (int c = b + 256; c < grid; c += 256) clash += sm[b] == sm[c];
    printf("%-36s grid %5d lanes %2d rot %d: %7.2f us, %6.2f ns/step, %6.1f ticks/step, %d SIMD clashes\n", what, grid, lanes, rot, ms * 1e3, ms * 1e6 / steps, mean / steps, clash);
    (void)hipFree(simd);
    (void)hipFree(out); (void)hipFree(ticks);
    return 0;
}

int main()
{
    for (int grid : {256, 1024})
        for (int lanes : {8, 16, 64}) {
            if (run<0>("registers only", grid, lanes, 0)) return 1;
            if (run<1>("16-sample blocks, moves (round 1)", grid, lanes, 0)) return 1;
            if (run<2>("16-sample blocks, alternating", grid, lanes, 0)) return 1;
            if (run<6>("  ... without the stores", grid, lanes, 0)) return 1;
            if (run<7>("  ... without the loads", grid, lanes, 0)) return 1;
            if (run<3>("32-sample blocks, alternating", grid, lanes, 0)) return 1;
            if (run<4>("4-sample blocks, unrolled 8", grid, lanes, 0)) return 1;
            if (run<5>("4-sample blocks, unrolled 16", grid, lanes, 0)) return 1;
        }
    return 0;
}
