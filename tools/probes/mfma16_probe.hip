// Control experiment for the prepass kernel's clock: what shader clock does an MI355X grant a SUSTAINED dense 16-bit MFMA
// stream, and does it depend on what travels with the MFMAs?
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_probe.bin mfma16_probe.hip && ./mfma16_probe.bin [seconds per variant]
//
// Every variant runs 256 workgroups x 4 waves (one wave per SIMD, as mlp_lp_kernel does) in back-to-back launches of ~0.25 s for the given
// time; workgroup 0 stamps clock64 (shader cycles) and wall_clock64 (100 MHz) at entry and exit of every launch, exactly as
// dsp_debug_last_clocks does for the product kernels, and the host prints the granted clock of the last launches and the TFLOP/s over them.
//
//   A  v_mfma_f32_32x32x16_f16 only, operands in registers (4 independent accumulator tiles per wave): the clean 16-bit matrix stream
//   B  A + one ds_read_b128 per MFMA feeding the A operand (K0's LDS duty: one 1 KiB A fragment per MFMA and wave)
//   C  B + K0's epilogue rate (per 8 MFMAs: 2 v_cvt_pk_f16_f32 + 2 v_pk_max_f16 on live accumulators)
//   D  C + one global_load_lds_dwordx4 per 2 MFMAs (K0's LDS-DMA refill rate: 4 pieces per 8 k-steps = 16 MFMAs ... issued at twice that here)
//   E  v_mfma_f32_16x16x32_f16 only (the candidate instruction of a 64-points-per-wave restructure), 8 independent tiles
//   F  E + one ds_read_b128 per 4 MFMAs (that restructure's LDS duty: one A fragment shared by four 16-point column blocks)
//   G  v_mfma_f32_16x16x4_f32 only (the fp32 kernels' instruction): the reference clock of this chip under fp32 MFMA
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack_relu(float lo, float hi) {
    const f32x2 v = {lo, hi};
    unsigned p = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2)), r;
    asm volatile("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(p));
    return r;
}

// pseudo-random f16 pair in (-0.125, 0.125): real weights / activations toggle every bit of the operand buses, a constant does not --
// and the chip's clock under load follows the switching power, not the instruction mix (the first run of this probe, constant data: 2.39 GHz
// in EVERY variant, while the product's prepass kernel is granted 1.71 GHz)
__device__ __forceinline__ unsigned rnd_h2(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const f32x2 v = {((int)(x & 0xffffu) - 32768) * (0.125f / 32768.f), ((int)(x >> 16) - 32768) * (0.125f / 32768.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void k_probe(int iters, const char* gsrc, float* sink, unsigned long long* clk, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64(); clk[1] = wall_clock64(); }
    // LDS: 64 KiB of small f16 values (0.0009765625 = 2^-10) so that accumulators stay finite
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(smem)[i] = rnd ? rnd_h2(i * 2654435761u + blockIdx.x) : 0x14001400u;
    __syncthreads();
    u32x4 a = {0x14001400u, 0x14001400u, 0x14001400u, 0x14001400u};
    u32x4 b = {0x14001400u + lane, 0x14001400u, 0x14001400u, 0x14001400u};
    if (rnd) {
        for (int q = 0; q < 4; ++q) { a[q] = rnd_h2(threadIdx.x * 8 + q + 77u * blockIdx.x); b[q] = rnd_h2(threadIdx.x * 8 + 4 + q + 131u * blockIdx.x); }
    }
    const __attribute__((address_space(3))) char* lbase = (const __attribute__((address_space(3))) char*)(size_t)((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16 + wave * 16384);
    if constexpr (VAR <= 3) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        u32x4 an[4];
        unsigned keep = 0;
        for (int j = 0; j < 4; ++j) an[j] = a;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // 16 MFMAs per iteration
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4 cur = an[j];
                    if constexpr (VAR >= 1) {
                        // one A fragment per MFMA, read one group ahead (the compiler places the lgkmcnt wait at the use)
                        an[j] = *reinterpret_cast<const __attribute__((address_space(3))) volatile u32x4*>(lbase + ((u * 4 + j) * 1024));
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur), __builtin_bit_cast(h8, b), acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // keep the written order: four independent accumulators in rotation, reads one rotation ahead
                    if constexpr (VAR >= 3) {
                        if ((j & 1) == 0) {        // one DMA piece per 2 MFMAs
                            const char* g = gsrc + (size_t)((it * 16 + u * 4 + j) & 1023) * 1024 + lane * 16;
                            const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + 65536 + wave * 1024);
                            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory");
                        }
                    }
                }
                if constexpr (VAR >= 2) {
                    if (u & 1) {                   // 2 cvt_pk + 2 pk_max per 8 MFMAs
                        // (the two OLDEST accumulators: their MFMAs were issued three and two MFMAs ago and have retired -- K0's epilogue reads the
                        // previous output group's accumulators, never one in flight)
                        keep ^= pack_relu(acc[0][u], acc[0][u + 4]);
                        keep ^= pack_relu(acc[1][u + 8], acc[1][u + 12]);
                    }
                }
            }
            if constexpr (VAR >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // keep the accumulators bounded: fold them every 64 iterations
            if ((it & 63) == 63)
                for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] *= 0.f;
        }
        float s = __uint_as_float(keep & 0xffu);
        for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    } else if constexpr (VAR <= 5) {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        u32x4 an = a, an2 = a;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {          // 32 MFMAs (16 cycles each) per iteration
                u32x4 cur0 = an, cur1 = an2;
                if constexpr (VAR == 5) {       // the next pair of A fragments, one group of eight MFMAs ahead
                    an = *reinterpret_cast<const __attribute__((address_space(3))) volatile u32x4*>(lbase + (u * 2) * 1024);
                    an2 = *reinterpret_cast<const __attribute__((address_space(3))) volatile u32x4*>(lbase + (u * 2 + 1) * 1024);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, cur0), __builtin_bit_cast(h8, b), acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 4; j < 8; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, cur1), __builtin_bit_cast(h8, b), acc[j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((it & 63) == 63)
                for (int j = 0; j < 8; ++j) acc[j] *= 0.f;
        }
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    } else {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float af = rnd ? __uint_as_float(0x3c000000u | (rnd_h2(threadIdx.x) & 0x7fffffu)) * (lane & 1 ? -1.f : 1.f) : 0.001f;
        const float bf = rnd ? __uint_as_float(0x3c000000u | (rnd_h2(threadIdx.x + 999u) & 0x7fffffu)) * (lane & 2 ? -1.f : 1.f) : 0.002f + lane;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[j], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
            if ((it & 63) == 63)
                for (int j = 0; j < 8; ++j) acc[j] *= 0.f;
        }
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (s == 12345.678f) sink[threadIdx.x] = s;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[2] = clock64(); clk[3] = wall_clock64(); }
}

template <int VAR>
void run(const char* name, double flop_per_iter_per_wave, double cycles_per_iter, double seconds, const char* gsrc, float* sink, unsigned long long* clk, int rnd) {
    const size_t lds = 65536 + 4096 + 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    // ~0.25 s per launch at 2 GHz if the matrix pipe were the only limit
    const int iters = (int)(0.25 * 2.0e9 / cycles_per_iter);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> mhz, tfs;
    double elapsed = 0.0;
    int launches = 0;
    while (elapsed < seconds && launches < 200) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_probe<VAR>), dim3(n_cu), dim3(256), lds, 0, iters, gsrc, sink, clk, rnd);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[4];
        CHECK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
        mhz.push_back((double)(h[2] - h[0]) / ((double)(h[3] - h[1]) / 100e6) / 1e6);
        tfs.push_back(flop_per_iter_per_wave * iters * 4.0 * n_cu / (ms * 1e-3) / 1e12);
        elapsed += ms * 1e-3;
        ++launches;
    }
    const int tail = std::max(1, launches / 2);            // the second half of the launches: the sustained state
    double m = 0, t = 0;
    for (int i = launches - tail; i < launches; ++i) { m += mhz[i]; t += tfs[i]; }
    m /= tail; t /= tail;
    printf("| %s, %s data | %d | %.2f | %.0f | %.0f | %.0f | %.1f | %.3f |\n", name, rnd ? "random" : "constant", launches, elapsed, mhz.front(), mhz[launches / 2], m, t,
           t / (flop_per_iter_per_wave / cycles_per_iter * 4.0 * n_cu * m * 1e6 / 1e12));
    (void)0;
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const char* only = argc > 2 ? argv[2] : "ABCDEFG";
    char* gsrc; float* sink; unsigned long long* clk;
    CHECK(hipMalloc(&gsrc, 2 << 20));
    {   // the DMA source: random bytes too (a zero page moves no bits)
        std::vector<unsigned> hsrc((2 << 20) / 4);
        unsigned x = 12345u;
        for (auto& w : hsrc) { x = x * 1664525u + 1013904223u; w = (x >> 3) & 0x33ff33ffu; }
        CHECK(hipMemcpy(gsrc, hsrc.data(), 2 << 20, hipMemcpyHostToDevice));
    }
    CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&clk, 64));
    printf("| variant | launches | seconds | MHz first launch | MHz middle | MHz sustained (2nd half) | TFLOP/s sustained | matrix-pipe duty at that clock |\n|---|---|---|---|---|---|---|---|\n");
    const double f32 = 32.0 * 32 * 16 * 2, f16x = 16.0 * 16 * 32 * 2, f4 = 16.0 * 16 * 4 * 2;
    if (strchr(only, 'G')) for (int rnd = 0; rnd < 2; ++rnd) run<6>("G fp32 16x16x4 MFMA only", 32 * f4, 32 * 32.0, seconds, gsrc, sink, clk, rnd);     // 2048 FLOP at 64 FLOP / clk / SIMD = 32 cycles
    if (strchr(only, 'A')) for (int rnd = 0; rnd < 2; ++rnd) run<0>("A f16 32x32x16 MFMA only", 16 * f32, 16 * 32.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'B')) for (int rnd = 0; rnd < 2; ++rnd) run<1>("B + ds_read_b128 per MFMA", 16 * f32, 16 * 32.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'C')) for (int rnd = 0; rnd < 2; ++rnd) run<2>("C + cvt_pk / pk_max epilogue", 16 * f32, 16 * 32.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'D')) for (int rnd = 0; rnd < 2; ++rnd) run<3>("D + LDS-DMA piece per 2 MFMAs", 16 * f32, 16 * 32.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'E')) for (int rnd = 0; rnd < 2; ++rnd) run<4>("E f16 16x16x32 MFMA only", 32 * f16x, 32 * 16.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'F')) for (int rnd = 0; rnd < 2; ++rnd) run<5>("F + ds_read_b128 per 4 MFMAs", 32 * f16x, 32 * 16.0, seconds, gsrc, sink, clk, rnd);
    if (strchr(only, 'A')) for (int rnd = 0; rnd < 2; ++rnd) run<0>("A again (after the others)", 16 * f32, 16 * 32.0, seconds, gsrc, sink, clk, rnd);
    return 0;
}
