// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) drop the AQL barrier bit on gfx950 (AMD_LOG_LEVEL=4 shows the header:
// 0xb02 barrier=1 -> 0xa02 barrier=0: yes), and do consecutive launches of one stream then overlap?  Every wavefront of a
// launch spins 3 us; each launch records its first start and its last end (100 MHz wall clock).
// build: hipcc -O3 --offload-arch=gfx950 -o anyorder_probe tools/micro/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(unsigned long long* first_start, unsigned long long* last_end, int slot, uint64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x == 0) {
        atomicMin(first_start + slot, (unsigned long long)t0);
        atomicMax(last_end + slot, (unsigned long long)wall_clock64());
    }
}

int main(int argc, char** argv)
{
    const int launches = 200;
    unsigned long long *fs, *le;
    (void)hipMalloc(&fs, launches * sizeof(*fs));
    (void)hipMalloc(&le, launches * sizeof(*le));
    hipStream_t s;
    (void)hipStreamCreate(&s);
    uint64_t ticks = 300;
    for (int blocks : {256, 1280}) {
        for (int flags = 0; flags <= 1; ++flags) {
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipMemsetAsync(fs, 0xff, launches * sizeof(*fs), s);
                (void)hipMemsetAsync(le, 0, launches * sizeof(*le), s);
                (void)hipStreamSynchronize(s);
                const auto h0 = std::chrono::steady_clock::now();
                for (int i = 0; i < launches; ++i) {
                    void* args[] = {&fs, &le, &i, &ticks};
                    (void)hipExtLaunchKernel((const void*)spin_kernel, dim3(blocks), dim3(256), args, 0, s, nullptr, nullptr, flags);
                }
                const auto h1 = std::chrono::steady_clock::now();
                (void)hipStreamSynchronize(s);
                const auto h2 = std::chrono::steady_clock::now();
                std::vector<unsigned long long> a(launches), b(launches);
                (void)hipMemcpy(a.data(), fs, launches * sizeof(*fs), hipMemcpyDeviceToHost);
                (void)hipMemcpy(b.data(), le, launches * sizeof(*le), hipMemcpyDeviceToHost);
                int overlapping = 0;
                std::vector<double> period;
                for (int i = 101; i < launches; ++i) {
                    if (a[i] < b[i - 1]) ++overlapping;
                    period.push_back((a[i] - a[i - 1]) * 0.01);
                }
                std::sort(period.begin(), period.end());
                printf("blocks %4d flags %d: host enqueue %.2f us/launch, total %.2f us/launch; launches 101-199: %d of 99 start before the "
                       "previous one's last wavefront has ended, start-to-start median %.2f us\n",
                       blocks, flags, std::chrono::duration<double, std::micro>(h1 - h0).count() / launches,
                       std::chrono::duration<double, std::micro>(h2 - h0).count() / launches, overlapping, period[period.size() / 2]);
            }
        }
    }
    return 0;
}
