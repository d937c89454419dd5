// gather_probe -- how the FETCH_SIZE counter of rocprofv3 relates to bytes actually requested, for the two access
// patterns that matter here: a coalesced stream and a gather of 64-byte records at random places (what the
// traversal kernel does).  The buffer (16 GiB) is far larger than every cache, and every record is read once.
//   gather_probe stream|gather64|gather128 [n_records]
// Prints the bytes the kernel requested; run under `rocprofv3 --pmc FETCH_SIZE` and compare.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// record i of the pass is record perm(i) of the buffer: a bijection on [0, n) for n a power of two (odd multiplier)
__device__ __forceinline__ unsigned long long perm(unsigned long long i, unsigned long long mask) {
    return (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & mask;
}

template <int REC> // bytes per record: 64 or 128, read by ONE lane as REC/16 uint4 loads; four records in flight per lane
__global__ void k_gather(const uint4 *buf, unsigned long long n, unsigned long long *out) {
    unsigned long long acc = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        uint4 v[4][REC / 16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint4 *p = buf + perm(i + j * stride, n - 1) * (REC / 16);
#pragma unroll
            for (int k = 0; k < REC / 16; k++) v[j][k] = p[k];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < REC / 16; k++) acc += v[j][k].x ^ v[j][k].y ^ v[j][k].z ^ v[j][k].w;
    }
    if (acc == 0x123456789ull) out[0] = acc; // keeps the loads alive
}

__global__ void k_stream(const uint4 *buf, unsigned long long n16, unsigned long long *out) {
    unsigned long long acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x123456789ull) out[0] = acc;
}

int main(int argc, char **argv) {
    const char *mode = argc > 1 ? argv[1] : "gather64";
    const size_t bytes = (size_t)16 << 30;
    uint4 *buf = nullptr;
    unsigned long long *out = nullptr;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&out, 8));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long requested = 0;
    CK(hipEventRecord(e0));
    if (!strcmp(mode, "stream")) {
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, buf, (unsigned long long)(bytes / 16), out);
        requested = bytes;
    } else if (!strcmp(mode, "gather128")) {
        const unsigned long long n = bytes / 128;
        hipLaunchKernelGGL(k_gather<128>, dim3(256 * 16), dim3(256), 0, 0, buf, n, out);
        requested = n * 128;
    } else {
        const unsigned long long n = bytes / 64;
        hipLaunchKernelGGL(k_gather<64>, dim3(256 * 16), dim3(256), 0, 0, buf, n, out);
        requested = n * 64;
    }
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("{\"mode\": \"%s\", \"requested_bytes\": %llu, \"ms\": %.3f, \"GBps\": %.1f}\n", mode, requested, ms, requested / (ms * 1e6));
    (void)hipFree(buf); (void)hipFree(out);
    return 0;
}
