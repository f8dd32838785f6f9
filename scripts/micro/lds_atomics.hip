// Microbenchmark (design input for the round-2 intersect kernel): LDS op cost per wave-instruction on gfx950.
// ds_write_b32 / ds_or_b32 / ds_or_b64 / ds_write_b8 / ds_read_u8, for 64/32/16/8 active lanes and two address
// patterns (conflict-free stride 1 dword; stride ~10 dwords + jitter like adjacent 16-byte lanes of a DV1 row).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip ; run: ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ITERS = 2000;
constexpr int UNROLL = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, int active, int pattern, unsigned long long *cyc) {
    __shared__ uint32_t lds[8192];
    const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0;
    __syncthreads();
    uint32_t addr = pattern == 0 ? (uint32_t)l : (uint32_t)(l * 10 + ((l * 7) & 3));
    addr = (addr + wv * 2048u) & 8191u;
    const bool on = l < active;
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t a = (addr + u * 640u + (it & 3)) & 8191u;
            if (on) {
                if (OP == 0) lds[a] = it;
                else if (OP == 1) atomicOr(&lds[a], 1u << (it & 31));
                else if (OP == 2) atomicOr(reinterpret_cast<unsigned long long *>(lds) + (a >> 1), 1ull << (it & 63));
                else if (OP == 3) reinterpret_cast<uint8_t *>(lds)[a * 4 + (it & 3)] = (uint8_t)it;
                else if (OP == 4) acc += reinterpret_cast<uint8_t *>(lds)[a * 4 + (it & 3)];
                else if (OP == 5) acc += lds[a];
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u || lds[tid] == 0xdeadbeefu) out[0] = acc;
}

template <int OP> static void run(const char *name, int wgs_per_cu) {
    uint32_t *d_out; unsigned long long *d_cyc;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&d_out, 64); hipMalloc(&d_cyc, grid * 8);
    for (int pattern = 0; pattern < 2; pattern++)
        for (int active : {64, 32, 16, 8}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<OP><<<grid, 256>>>(d_out, active, pattern, d_cyc);
            hipEventRecord(e0);
            k<OP><<<grid, 256>>>(d_out, active, pattern, d_cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // wave-instructions per CU: wgs_per_cu * 4 waves * ITERS * UNROLL
            const double winstr = (double)wgs_per_cu * 4 * ITERS * UNROLL;
            const double cycles = ms * 1e-3 * 2.4e9;
            printf("%-12s wgs/cu=%d pattern=%d active=%2d : %.3f ms  -> %.1f CU-cycles per wave-instr (%.2f per active lane)\n", name, wgs_per_cu,
                   pattern, active, ms, cycles / winstr, cycles / winstr / active);
        }
    hipFree(d_out); hipFree(d_cyc);
}

int main() {
    for (int w : {2, 5}) {
        run<0>("ds_write_b32", w);
        run<1>("ds_or_b32", w);
        run<2>("ds_or_b64", w);
        run<3>("ds_write_b8", w);
        run<4>("ds_read_u8", w);
        run<5>("ds_read_b32", w);
    }
    return 0;
}
