#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
// does the LDS float atomic add round like v_add_f32 (IEEE RNE, denormals kept)?
__global__ void k(const float* a, const float* b, float* out_lds, float* out_valu, float* out_old, int n) {
    __shared__ float s[256];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        s[threadIdx.x] = a[i];
        __builtin_amdgcn_wave_barrier();
        const float old = __hip_atomic_fetch_add(&s[threadIdx.x], b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        out_lds[i] = s[threadIdx.x];
        out_old[i] = old;
        out_valu[i] = __fadd_rn(a[i], b[i]);
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> a(n), b(n), r1(n), r2(n), r3(n);
    srand(1);
    auto rnd = [](int mode) { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); if (mode == 0) { u &= 0x7FFFFFFFu; }  float f; memcpy(&f, &u, 4); return f; };
    for (int i = 0; i < n; i++) {
        int m = i & 3;
        if (m == 0) { a[i] = rnd(1); b[i] = rnd(1); }                                  // any bit patterns (NaN, inf, denormals, signs)
        else if (m == 1) { a[i] = (float)rand() / RAND_MAX * 30.f; b[i] = (float)rand() / RAND_MAX * 8.f; }     // score-like
        else if (m == 2) { unsigned u = rand() & 0x7FFFFF; float f; memcpy(&f, &u, 4); a[i] = f; unsigned v = rand() & 0xFFFFFF; memcpy(&f, &v, 4); b[i] = f; }   // denormals
        else { a[i] = (float)rand() / RAND_MAX; unsigned u; memcpy(&u, &a[i], 4); u += (rand() % 64) << 23; memcpy(&a[i], &u, 4); b[i] = (float)rand() / RAND_MAX; }
    }
    float *da, *db, *d1, *d2, *d3;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, da, db, d1, d2, d3, n);
    hipMemcpy(r1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r3.data(), d3, n * 4, hipMemcpyDeviceToHost);
    long bad[4] = {0, 0, 0, 0}, badold = 0, badhost = 0; int shown = 0;
    for (int i = 0; i < n; i++) {
        const bool nan1 = r1[i] != r1[i], nan2 = r2[i] != r2[i];
        if (memcmp(&r1[i], &r2[i], 4) != 0 && !(nan1 && nan2)) { bad[i & 3]++; if (shown++ < 8) printf("mismatch mode %d: a=%a b=%a lds=%a valu=%a\n", i & 3, a[i], b[i], r1[i], r2[i]); }
        if (memcmp(&r3[i], &a[i], 4) != 0 && !(a[i] != a[i])) badold++;
        volatile float h = a[i] + b[i]; float hh = h;
        if (memcmp(&hh, &r2[i], 4) != 0 && !(hh != hh)) badhost++;
    }
    printf("{\"n\": %d, \"lds_vs_valu_mismatch\": {\"any_bits\": %ld, \"score_like\": %ld, \"denormal\": %ld, \"mixed_exponent\": %ld}, \"returned_old_mismatch\": %ld, \"valu_vs_host_mismatch\": %ld}\n", n, bad[0], bad[1], bad[2], bad[3], badold, badhost);
    return 0;
}
