// Hardware-semantics probes for gfx950 used while developing mlp_kernel.hip:
//  (1) lane maps of v_mfma_f32_16x16x4_f32 (A, B, D);  (2) LDS-DMA (global_load_lds_dwordx4) destination
//  addressing, in particular LDS byte addresses >= 64 KiB through M0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ void k_glds(const float* src, float* out, int n_slots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < n_slots * 4096; i += 256) f[i] = -1.f;   // 16 KiB slots
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    for (int s = 0; s < n_slots; ++s) {
        const char* g = (const char*)src + (size_t)s * 16384 + wave * 4096 + lane * 16;
        for (int i = 0; i < 4; ++i) glds16(g + i * 1024, base + s * 16384 + wave * 4096 + i * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < n_slots * 4096; i += 256) out[i] = f[i];
    if (threadIdx.x == 0) out[n_slots * 4096] = (float)base;
}

__global__ void k_glds_off(const float* src, float* out) {
    // does the instruction's immediate offset move the LDS destination too, or only the global source?
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    float* f = (float*)smem;
    for (int i = threadIdx.x; i < 4096; i += 64) f[i] = -1.f;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* g = (const char*)src + lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = f[i];
}

int main() {
    // (1) MFMA maps: A[i][k] = 100*i + k,  B[k][j] = (k==kk)*... use one-hot probes
    std::vector<float> ha(64), hb(64), hd(256);
    float *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    // hypothesis: A lane l -> A[i=l&15][k=l>>4]; B lane l -> B[k=l>>4][j=l&15]; D lane l reg r -> D[4*(l>>4)+r][l&15]
    std::vector<float> A(16 * 4), B(4 * 16);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (float)(1 + i) + 0.25f * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k + 1) * (j % 5 + 1)) + 0.5f * (j / 5);
    for (int l = 0; l < 64; ++l) { ha[l] = A[(l & 15) * 4 + (l >> 4)]; hb[l] = B[(l >> 4) * 16 + (l & 15)]; }
    hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd.data(), dd, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        double ref = 0; for (int k = 0; k < 4; ++k) ref += (double)A[row * 4 + k] * B[k * 16 + col];
        worst = fmax(worst, fabs(ref - hd[l * 4 + r]));
    }
    printf("MFMA16x16x4 lane-map hypothesis: max err %.3g  -> %s\n", worst, worst < 1e-3 ? "CONFIRMED" : "WRONG");
    // (2) LDS-DMA into 9 slots of 16 KiB (144 KiB): which land where?
    const int n_slots = 9;
    std::vector<float> hs(n_slots * 4096), ho(n_slots * 4096 + 1);
    for (size_t i = 0; i < hs.size(); ++i) hs[i] = (float)i;
    float *ds, *dout;
    hipMalloc(&ds, hs.size() * 4); hipMalloc(&dout, ho.size() * 4);
    hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k_glds, hipFuncAttributeMaxDynamicSharedMemorySize, n_slots * 16384);
    hipLaunchKernelGGL(k_glds, dim3(1), dim3(256), n_slots * 16384, 0, ds, dout, n_slots);
    hipError_t e = hipDeviceSynchronize();
    printf("glds kernel: %s\n", hipGetErrorString(e));
    hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
    printf("dynamic LDS base = %.0f\n", ho[n_slots * 4096]);
    for (int s = 0; s < n_slots; ++s) {
        int ok = 0, untouched = 0, other = 0; float first_other = 0;
        for (int i = 0; i < 4096; ++i) {
            const float v = ho[s * 4096 + i];
            if (v == (float)(s * 4096 + i)) ++ok; else if (v == -1.f) ++untouched; else { if (!other) first_other = v; ++other; }
        }
        printf(" slot %d @%6d: ok %4d untouched %4d other %4d (first other value %.0f)\n", s, s * 16384, ok, untouched, other, first_other);
    }
    {
        float* dout2; hipMalloc(&dout2, 4096 * 4);
        hipLaunchKernelGGL(k_glds_off, dim3(1), dim3(64), 16384, 0, ds, dout2);
        std::vector<float> h2(4096);
        hipMemcpy(h2.data(), dout2, 4096 * 4, hipMemcpyDeviceToHost);
        int first = -1; for (int i = 0; i < 4096; ++i) if (h2[i] != -1.f) { first = i; break; }
        printf("glds offset:2048 with M0=base: first written float index %d (LDS byte %d), value %.0f (source float index) -> %s\n",
               first, first * 4, first >= 0 ? h2[first] : -1.f,
               first == 0 ? "offset applies to GLOBAL only" : (first == 512 ? "offset applies to BOTH global and LDS" : "other"));
    }
    return 0;
}
