// tools/patterns.hip -- memory-pattern microbenchmarks behind the layout / kernel-shape decisions in
// DESIGN.md (no LBM arithmetic: what does an 18-stream D2Q9-shaped access pattern get on MI355X?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/patterns tools/patterns.hip && ./tools/patterns [nx ny]
// Reported GB/s = (bytes read + bytes written) / time; fp32, 9 planes in + 9 planes out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int cxk(int k) { return (k == 1 || k == 5 || k == 8) ? 1 : ((k == 3 || k == 6 || k == 7) ? -1 : 0); }
__device__ __forceinline__ int cyk(int k) { return (k == 2 || k == 5 || k == 6) ? 1 : ((k == 4 || k == 7 || k == 8) ? -1 : 0); }
template <int NT> __device__ __forceinline__ v4f ld(const float* p) { if (NT) return __builtin_nontemporal_load((const v4f*)p); return *(const v4f*)p; }
template <int NT> __device__ __forceinline__ void st(float* p, v4f v) { if (NT) __builtin_nontemporal_store(v, (v4f*)p); else *(v4f*)p = v; }

template <int NT>
__global__ __launch_bounds__(256) void copy1(const float* __restrict__ a, float* __restrict__ b, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) st<NT>(b + 4 * i, ld<NT>(a + 4 * i));
}

// element (k, x, y) at k*plane + (y+1)*row + 4 + x.  MODE 0: pull (x-+1 shifted 16-B loads, aligned stores),
// 1: push (aligned loads, shifted stores), 2: pull with the x shifts staged through LDS, 3: no shifts at all.
template <int MODE, int NT, int XCD>
__global__ __launch_bounds__(256) void lat(const float* __restrict__ src, float* __restrict__ dst, long long plane, long long row,
                                           int nx, int nxb, int nblocks) {
    int b = blockIdx.x;
    if (XCD) { int per = nblocks >> 3; if (b < (per << 3)) b = (b & 7) * per + (b >> 3); }
    const int y = b / nxb, x = ((b % nxb) * 256 + threadIdx.x) * 4;
    const long long o = (long long)(y + 1) * row + 4 + x;
    v4f v[9];
    if (MODE == 0 || MODE == 3) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float* p = src + k * plane + o + (MODE == 3 ? 0 : (-cxk(k) + (long long)cyk(k) * row));
            if (MODE == 0 && cxk(k) != 0) { v4f t; __builtin_memcpy(&t, p, 16); v[k] = t; } else v[k] = ld<NT>(p);
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) st<NT>(dst + k * plane + o, v[k]);
    } else if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = ld<NT>(src + k * plane + o);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float* p = dst + k * plane + o + (cxk(k) - (long long)cyk(k) * row);
            if (cxk(k) != 0) __builtin_memcpy(p, &v[k], 16); else st<NT>(p, v[k]);
        }
    } else {
        __shared__ float tile[6][1024 + 8];
        int j = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float* p = src + k * plane + o + (long long)cyk(k) * row;
            v4f own = *(const v4f*)p;
            if (cxk(k) == 0) { v[k] = own; continue; }
            *(v4f*)&tile[j][4 + threadIdx.x * 4] = own;
            if (threadIdx.x == 0) tile[j][3] = p[-1];
            if (threadIdx.x == 255) tile[j][4 + 1024] = p[4];
            ++j;
        }
        __syncthreads();
        j = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (cxk(k) == 0) continue;
            const float* q = &tile[j][4 + threadIdx.x * 4 - cxk(k)];
            v[k] = v4f{q[0], q[1], q[2], q[3]};
            ++j;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) *(v4f*)(dst + k * plane + o) = v[k];
    }
}

template <typename F> float timeit(F f, int it) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < it; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / it;
}

int main(int argc, char** argv) {
    int nx = argc > 1 ? atoi(argv[1]) : 4096, ny = argc > 2 ? atoi(argv[2]) : 4096;
    int pitch = nx + 8; size_t bytes = (size_t)9 * pitch * (ny + 3) * 4;
    float *s, *d; CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMemset(s, 0, bytes)); CK(hipMemset(d, 0, bytes));
    double alg = 18.0 * nx * ny * 4; float ms; int nxb = nx / 1024, nb = nxb * ny; size_t n4 = bytes / 16;
    printf("lattice %d x %d fp32, %.2f GB per lattice\n", nx, ny, bytes / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((copy1<0>), dim3((n4 + 255) / 256), dim3(256), 0, 0, s, d, n4); }, 10); printf("%-52s %7.0f GB/s\n", "plain copy, 16 B/lane", 2.0 * bytes / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL((copy1<1>), dim3((n4 + 255) / 256), dim3(256), 0, 0, s, d, n4); }, 10); printf("%-52s %7.0f GB/s\n", "plain copy, 16 B/lane, non-temporal", 2.0 * bytes / ms / 1e6);
    for (int layout = 0; layout < 2; ++layout) {
        long long plane = layout ? pitch : (long long)pitch * (ny + 2), row = layout ? 9LL * pitch : pitch;
        const char* L = layout ? "[y][k][x]" : "[k][y][x]";
        char buf[128];
#define RUN(MODE, NT, XCD, what) ms = timeit([&] { hipLaunchKernelGGL((lat<MODE, NT, XCD>), dim3(nb), dim3(256), 0, 0, s, d, plane, row, nx, nxb, nb); }, 10); \
        snprintf(buf, sizeof buf, "%s %s%s%s", L, what, NT ? ", non-temporal" : "", XCD ? ", XCD bands" : ""); printf("%-52s %7.0f GB/s\n", buf, alg / ms / 1e6);
        RUN(3, 0, 0, "9+9 streams, no shift") RUN(3, 0, 1, "9+9 streams, no shift") RUN(0, 0, 0, "pull") RUN(0, 0, 1, "pull") RUN(0, 1, 1, "pull")
        RUN(1, 0, 1, "push") RUN(2, 0, 1, "pull, x shifts through LDS")
    }
    return 0;
}
