// Does straight-line code larger than the instruction cache cost MFMA throughput on gfx950?
// mlp_kernel's layer body is ~90 KB of unrolled code; the instruction cache is 64 KB per CU pair.
// Each kernel runs the same number of MFMAs, as a loop over a straight-line body of N MFMAs (8 bytes each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#define DSR(o) if (DS) { asm volatile("ds_read_b128 %0, %1 offset:" #o : "=v"(av) : "v"(lane * 16) : "memory"); }
#define DSW if (DS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); a += av.x * 1e-30f; }
#define B8 DSR(1024) MF(0) MF(1) MF(2) MF(3) DSW DSR(2048) MF(4) MF(5) MF(6) MF(7) DSW
#define R4(x) x x x x
#define M64(x) R4(R4(R4(x)))
#define M256(x) R4(M64(x))

template <int N, int DS>
__global__ __launch_bounds__(256) void k_body(float* out, int loops, long long* clk) {
    __shared__ f32x4 lds[64 * 20];
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < 20; ++i) lds[lane + 64 * i] = (f32x4){1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    float a = 1.0f + threadIdx.x * 1e-9f, b = 1.0f;
    f32x4 av = lds[lane];
    const long long t0 = clock64();
    for (int l = 0; l < loops; ++l) {
        // N / 8 blocks of eight MFMAs, expanded by the preprocessor so that the body really is straight-line code
        if (N == 2048) { M256(B8) }
        if (N == 5632) { M256(B8) M256(B8) M64(B8) M64(B8) M64(B8) }
        if (N == 7168) { M256(B8) M256(B8) M256(B8) M64(B8) M64(B8) }
        if (N == 11264) { M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M64(B8) M64(B8) }
        if (N == 22528) { M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) M256(B8) }
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int N, int DS>
void run(const char* name, float* out, long long* clk) {
    const long long total = 22528LL * 16;
    const int loops = (int)(total / N);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_body<N, DS>), dim3(256), dim3(256), 0, 0, out, loops, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        long long mx = 0; for (int i = 0; i < 256; ++i) mx = h[i] > mx ? h[i] : mx;
        if (rep == 2)
            printf("%-28s body %6d MFMA (%4d KB) x %4d loops: %.3f ms, %.2f shader-clk/MFMA (max WG), %.1f TFLOP/s\n", name, N, N * 8 / 1024,
                   loops, ms, (double)mx / ((double)loops * N), 2048.0 * loops * N * 1024 / (ms * 1e-3) / 1e12);
    }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
    run<2048, 0>("mfma only", out, clk);
    run<5632, 0>("mfma only", out, clk);
    run<7168, 0>("mfma only", out, clk);
    run<11264, 0>("mfma only", out, clk);
    run<22528, 0>("mfma only", out, clk);
    run<2048, 1>("mfma + ds_read/4", out, clk);
    run<5632, 1>("mfma + ds_read/4", out, clk);
    run<11264, 1>("mfma + ds_read/4", out, clk);
    return 0;
}
