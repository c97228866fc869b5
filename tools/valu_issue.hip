// valu_issue.hip -- issue cost of the vector instructions the lattice kernels are made of, on gfx950.
//
// The microarchitecture guide lists v_fma_f32 (2 cycles per wave-instruction on a SIMD-32) but not the packed fp32 forms
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) nor the fp64 ones; bench.py's VALU roofline line needs them.  Each kernel runs
// a loop of independent instructions of ONE kind (eight accumulators, inline asm so that the compiler cannot fuse or drop
// them) in W waves per SIMD (one workgroup of 256 W threads per CU, 100 KiB of LDS so that no second workgroup joins) and
// stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around it.
//
//   cycles per wave-instruction and SIMD = median over waves of (cycles of the loop) / (instructions of one wave) / W
//
// build: hipcc --offload-arch=gfx950 -O2 -o tools/valu_issue tools/valu_issue.hip ; run: tools/valu_issue
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { OP_FMA32, OP_ADD32, OP_MUL32, OP_PKFMA32, OP_PKADD32, OP_PKMUL32, OP_MOV, OP_MOV_DPP, OP_RCP32, OP_SQRT32,
       OP_FMA64, OP_ADD64, OP_MUL64, OP_RCP64, OP_BPERM, NOPS };
static const char* names[NOPS] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_mov_b32",
                                  "v_mov_b32 dpp wave_shr:1", "v_rcp_f32", "v_sqrt_f32", "v_fma_f64", "v_add_f64", "v_mul_f64",
                                  "v_rcp_f64", "ds_bpermute_b32"};

constexpr int ITER = 2048, UNROLL = 8;   // 16384 instructions per wave

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float seed) {
    __shared__ char pad[100 * 1024];
    if (seed == 12345.f) pad[threadIdx.x] = 1;   // keeps the array
    float a[UNROLL];
    f32x2 p[UNROLL];
    double d[UNROLL];
    const float b = seed * 1.0001f, c = seed * 0.5f;
    const f32x2 pb = {b, b}, pc = {c, c};
    const double db = b, dc = c;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1}; d[i] = a[i]; }
    const int addr = ((threadIdx.x + 1) & 63) * 4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == OP_FMA32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == OP_ADD32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MUL32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            if (OP == OP_PKFMA32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
            if (OP == OP_PKADD32) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
            if (OP == OP_PKMUL32) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
            if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) % UNROLL]));
            if (OP == OP_RCP32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == OP_SQRT32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(db), "v"(dc));
            if (OP == OP_ADD64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (OP == OP_MUL64) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
            if (OP == OP_RCP64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            if (OP == OP_BPERM) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(addr));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 1.2345f) out[0] = 1;   // keeps the results
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 + 2 * w] = t1 - t0;
        out[3 + 2 * w] = r1 - r0;
    }
}

template <int OP>
void run(int W, unsigned long long* dev, std::vector<unsigned long long>& host) {
    const int blocks = 256, threads = 256 * W, waves = blocks * threads / 64;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, dev, 1.5f);   // warm-up (clocks)
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, dev, 1.5f);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host.data(), dev, (2 + 2 * (size_t)waves) * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; ++w) {
        cyc[w] = (double)host[2 + 2 * w] / ((double)ITER * UNROLL) / W;
        clk[w] = (double)host[2 + 2 * w] / (double)host[3 + 2 * w] * 0.1;   // GHz
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(clk.begin(), clk.end());
    std::printf("%-26s W=%d  cycles per wave-instruction and SIMD: median %.2f (min %.2f)  clock %.2f GHz\n", names[OP], W, cyc[waves / 2], cyc[0],
                clk[waves / 2]);
}

template <int OP>
void sweep(unsigned long long* dev, std::vector<unsigned long long>& host) {
    for (int W : {1, 2, 4}) run<OP>(W, dev, host);
}

int main() {
    unsigned long long* dev = nullptr;
    const size_t n = 2 + 2 * 256 * 16;
    (void)hipMalloc(&dev, n * 8);
    (void)hipMemset(dev, 0, n * 8);
    std::vector<unsigned long long> host(n);
    sweep<OP_FMA32>(dev, host); sweep<OP_ADD32>(dev, host); sweep<OP_MUL32>(dev, host);
    sweep<OP_PKFMA32>(dev, host); sweep<OP_PKADD32>(dev, host); sweep<OP_PKMUL32>(dev, host);
    sweep<OP_MOV>(dev, host); sweep<OP_MOV_DPP>(dev, host); sweep<OP_RCP32>(dev, host); sweep<OP_SQRT32>(dev, host);
    sweep<OP_FMA64>(dev, host); sweep<OP_ADD64>(dev, host); sweep<OP_MUL64>(dev, host); sweep<OP_RCP64>(dev, host);
    sweep<OP_BPERM>(dev, host);
    (void)hipFree(dev);
    return 0;
}
