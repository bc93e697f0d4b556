// Streaming micro-benchmarks on one MI355X: what read / write / mixed streams reach on this pool.
//   hipcc -O3 --offload-arch=gfx950 -o stream stream.hip && ./stream [GiB=8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void k_read(const d2* __restrict__ a, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { d2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
template <bool NT> __global__ void k_write(d2* __restrict__ a, size_t n) {
    d2 v; v.x = 1.0; v.y = 2.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v;
    }
}
// one-pass grid (no grid stride): block b handles a contiguous chunk
template <bool NT> __global__ void k_write_flat(d2* __restrict__ a, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    d2 v; v.x = 1.0; v.y = 2.0;
    if (i < n) { if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v; }
}
template <bool NT> __global__ void k_copy(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d2 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
__global__ void k_read_flat(const d2* __restrict__ a, size_t n, double* out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) { const d2 v = a[i]; if (v.x + v.y == 1.2345) out[0] = v.x; }
}
__global__ void k_copy_flat(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ void k_axpy_flat(const d2* __restrict__ x, const d2* __restrict__ f, d2* __restrict__ o, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) o[i] = x[i] + f[i];
}
// block b handles a contiguous chunk of `per` elements per thread-row (a tile-like order: long runs per workgroup)
__global__ void k_axpy_chunk(const d2* __restrict__ x, const d2* __restrict__ f, d2* __restrict__ o, size_t n, size_t per) {
    const size_t base = blockIdx.x * per * blockDim.x;
    for (size_t t = 0; t < per; ++t) {
        const size_t i = base + t * blockDim.x + threadIdx.x;
        if (i < n) o[i] = x[i] + f[i];
    }
}
// out = x + f (2 reads : 1 write), like a class-coded sweep without the stencil
template <bool NT> __global__ void k_axpy(const d2* __restrict__ x, const d2* __restrict__ f, d2* __restrict__ o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        d2 a = x[i], b = NT ? __builtin_nontemporal_load(f + i) : f[i];
        d2 v = a + b;
        if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
}
// 8 B per lane variants
__global__ void k_axpy8(const double* __restrict__ x, const double* __restrict__ f, double* __restrict__ o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = x[i] + f[i];
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t n = (size_t)(gib * (1ull << 30)) / 16;       // d2 elements
    d2 *a, *b, *c; double* out;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, double bytes, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-44s %8.3f ms  %7.3f TB/s\n", name, ms, bytes / ms / 1e9);
        fflush(stdout);
    };
    const double B = (double)n * 16;
    for (int grid : {2048, 4096, 8192}) {
        printf("-- grid %d x 256 threads, %.1f GiB per array\n", grid, gib);
        time("read 16B/lane", B, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); });
        time("write 16B/lane", B, [&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, a, n); });
        time("write 16B/lane nt", B, [&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, a, n); });
        time("copy 16B/lane (r+w bytes)", 2 * B, [&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        time("copy 16B/lane nt (r+w bytes)", 2 * B, [&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n); });
        time("o = x + f 16B/lane (2r+1w bytes)", 3 * B, [&] { hipLaunchKernelGGL(k_axpy<false>, dim3(grid), dim3(256), 0, 0, a, b, c, n); });
        time("o = x + f 16B/lane nt f,o (2r+1w bytes)", 3 * B, [&] { hipLaunchKernelGGL(k_axpy<true>, dim3(grid), dim3(256), 0, 0, a, b, c, n); });
        time("o = x + f 8B/lane (2r+1w bytes)", 3 * B, [&] { hipLaunchKernelGGL(k_axpy8, dim3(grid), dim3(256), 0, 0, (double*)a, (double*)b, (double*)c, 2 * n); });
    }
    {
        const unsigned g = (unsigned)((n + 255) / 256);
        time("write flat grid", B, [&] { hipLaunchKernelGGL(k_write_flat<false>, dim3(g), dim3(256), 0, 0, a, n); });
        time("write flat grid nt", B, [&] { hipLaunchKernelGGL(k_write_flat<true>, dim3(g), dim3(256), 0, 0, a, n); });
        time("read flat grid", B, [&] { hipLaunchKernelGGL(k_read_flat, dim3(g), dim3(256), 0, 0, a, n, out); });
        time("copy flat grid (r+w bytes)", 2 * B, [&] { hipLaunchKernelGGL(k_copy_flat, dim3(g), dim3(256), 0, 0, a, b, n); });
        time("o = x + f flat grid (2r+1w bytes)", 3 * B, [&] { hipLaunchKernelGGL(k_axpy_flat, dim3(g), dim3(256), 0, 0, a, b, c, n); });
        for (size_t per : {8, 64, 512}) {
            char name[64]; snprintf(name, sizeof name, "o = x + f, %zu x 4 KiB per block (2r+1w)", per);
            const unsigned gc = (unsigned)((n + per * 256 - 1) / (per * 256));
            time(name, 3 * B, [&] { hipLaunchKernelGGL(k_axpy_chunk, dim3(gc), dim3(256), 0, 0, a, b, c, n, per); });
        }
    }
    return 0;
}
