// accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) on [-3.3, 3.3] rad
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* a, float* s, float* c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rev = a[i] * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> a(n), s(n), c(n);
    for (int i = 0; i < n; ++i) a[i] = -3.3f + 6.6f * (float)i / (float)(n - 1);
    float *da, *ds, *dc;
    hipMalloc(&da, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
        es = fmax(es, fabs((double)s[i] - sin((double)a[i])));
        ec = fmax(ec, fabs((double)c[i] - cos((double)a[i])));
    }
    printf("max abs err: sin %.3e cos %.3e\n", es, ec);
    return 0;
}
