// tools/queue_probe.hip — which of four freshly created HIP streams run concurrently?  (build: hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/queue_probe)
// usage: tools/queue_probe <foreign streams created and kept before the four>
// For every ordered pair (i, j): a kernel that spins for ~200 us goes to stream i, then a kernel that takes a timestamp goes to stream j.  If the
// timestamp lies before the end of the spin, the two streams ran at the same time (different hardware queues); otherwise j waited for i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_spin(unsigned long long ticks, unsigned long long* out) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } out[0] = t0; out[1] = wall_clock64(); }
__global__ void k_stamp(unsigned long long* out) { out[0] = wall_clock64(); }
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int foreign = argc > 1 ? atoi(argv[1]) : 0;
    std::vector<hipStream_t> f(foreign);
    for (auto& s : f) OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t s[4];
    for (auto& x : s) OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    unsigned long long* d = nullptr; OK(hipMalloc(&d, 64));
    for (int i = 0; i < 4; ++i) { k_stamp<<<1, 1, 0, s[i]>>>(d + 4); OK(hipStreamSynchronize(s[i])); }      // every stream has been used once
    printf("foreign %d: ", foreign);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        if (i == j) continue;
        k_spin<<<1, 1, 0, s[i]>>>(20000ull /* 100 MHz ticks: 200 us */, d);
        k_stamp<<<1, 1, 0, s[j]>>>(d + 2);
        OK(hipDeviceSynchronize());
        unsigned long long h[3]; OK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("%d%d%c ", i, j, h[2] < h[1] ? '|' : '-');      // | concurrent, - serialised
    }
    printf("\n");
    // a fifth stream beside the four: default priority, then the highest — does it get a hardware queue of its own?
    for (int high = 0; high < 2; ++high) {
        hipStream_t x;
        int lo = 0, hi = 0;
        OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (high) OK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, hi)); else OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        k_stamp<<<1, 1, 0, x>>>(d + 4); OK(hipStreamSynchronize(x));
        printf("  fifth stream (%s priority) against the four, both ways: ", high ? "highest" : "default");
        for (int i = 0; i < 4; ++i) {
            unsigned long long h[3];
            k_spin<<<1, 1, 0, s[i]>>>(20000ull, d); k_stamp<<<1, 1, 0, x>>>(d + 2); OK(hipDeviceSynchronize());
            OK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost)); printf("%dx%c ", i, h[2] < h[1] ? '|' : '-');
            k_spin<<<1, 1, 0, x>>>(20000ull, d); k_stamp<<<1, 1, 0, s[i]>>>(d + 2); OK(hipDeviceSynchronize());
            OK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost)); printf("x%d%c ", i, h[2] < h[1] ? '|' : '-');
        }
        printf("\n");
    }
    return 0;
}
