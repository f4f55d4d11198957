// Does hipcc's SLP vectoriser change fp32 results?  out = {x/d, y/d} per thread, built twice (with and without
// -fno-slp-vectorize), compared bitwise with the host's IEEE division.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
__global__ void k_div(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[2 * i], y = a[2 * i + 1], d = b[i];
    out[2 * i] = x / d;
    out[2 * i + 1] = y / d;
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(2 * n), b(n), o(2 * n);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < 2 * n; ++i) { uint32_t u = rnd(); float f; u = (u & 0x807FFFFFu) | (((u >> 23) % 60 + 97) << 23); memcpy(&f, &u, 4); a[i] = f; }
    for (int i = 0; i < n; ++i) { uint32_t u = rnd(); float f; u = (u & 0x807FFFFFu) | (((u >> 23) % 60 + 97) << 23); memcpy(&f, &u, 4); b[i] = f; }
    float *da, *db, *dout;
    hipMalloc(&da, 8 * n); hipMalloc(&db, 4 * n); hipMalloc(&dout, 8 * n);
    hipMemcpy(da, a.data(), 8 * n, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_div, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(o.data(), dout, 8 * n, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        volatile float q0 = a[2 * i] / b[i], q1 = a[2 * i + 1] / b[i];
        float r0 = q0, r1 = q1;
        if (memcmp(&r0, &o[2 * i], 4) || memcmp(&r1, &o[2 * i + 1], 4)) ++bad;
    }
    printf("%ld of %d quotient pairs differ from IEEE division\n", bad, n);
    return 0;
}
