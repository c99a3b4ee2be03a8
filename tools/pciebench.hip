// Host->device copy rate of one MI355X box for pageable and page-locked host memory (hipcc --offload-arch=gfx950).
// Context for the PCIe-inclusive figure in DESIGN.md section 6.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double rate(void* d, const void* h, size_t n, hipStream_t s)
{
    for (int i = 0; i < 2; ++i) { (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 8;
    for (int i = 0; i < reps; ++i) (void)hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s);
    (void)hipStreamSynchronize(s);
    return (double)n * reps / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
}
int main()
{
    const size_t n = 256u << 20;
    void* d; CK(hipMalloc(&d, n));
    hipStream_t s; CK(hipStreamCreate(&s));
    void* pg = aligned_alloc(4096, n); memset(pg, 1, n);
    printf("pageable            %.1f GB/s\n", rate(d, pg, n, s));
    const unsigned flags[3] = {hipHostMallocDefault, hipHostMallocPortable, hipHostMallocNonCoherent};
    const char* names[3] = {"pinned default     ", "pinned portable    ", "pinned non-coherent"};
    for (int i = 0; i < 3; ++i) {
        void* p; CK(hipHostMalloc(&p, n, flags[i])); memset(p, 1, n);
        printf("%s %.1f GB/s\n", names[i], rate(d, p, n, s));
        CK(hipHostFree(p));
    }
    CK(hipHostRegister(pg, n, hipHostRegisterDefault));
    printf("registered pageable %.1f GB/s\n", rate(d, pg, n, s));
    return 0;
}
