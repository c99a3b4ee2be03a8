// ldsbench.hip -- what an LDS read ADDRESS PATTERN costs on gfx950 (round 6: the operand reads of the sparse matrix forms).
// A pattern is one byte address per lane and read (two reads per step, as the sparse forms issue them: the two 16-byte halves of a
// lane's 32 operand bytes); every wave of the launch repeats the step `iters` times, 8 blocks of 4 waves per CU, so the LDS pipe
// is the only thing that can saturate.  Reported: nanoseconds per read instruction and CU, and the ratio to the conflict-free
// pattern (lane l at 16 l).  The patterns themselves come from the caller (tools/ldsbench.py) -- the kernel only replays them.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/ldsbench.hip -o tools/libldsbench.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <int WIDTH>
__global__ void __launch_bounds__(256) lds_replay(const int* a0, const int* a1, int iters, uint32_t* sink)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 4096u; i += 256u) lds[i] = i * 2654435761u;      // 16 KB of something
    __syncthreads();
    // every wave the same pattern (a base that is a multiple of the 256-byte bank row changes nothing); addresses wrap at 16 KB
    const uint32_t x0 = ((uint32_t)a0[lane] + 1024u * wave) & 16376u, x1 = ((uint32_t)a1[lane] + 1024u * wave) & 16376u;
    int acc = 0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (WIDTH == 16) {
            i4 r0, r1, r2, r3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:16384\n\tds_read_b128 %3, %5 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1) : "memory");
            acc ^= r0.x ^ r1.y ^ r2.z ^ r3.w;
        } else {
            i2 r0, r1, r2, r3;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %4 offset:16384\n\tds_read_b64 %3, %5 offset:16384\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(x0), "v"(x1) : "memory");
            acc ^= r0.x ^ r1.y ^ r2.x ^ r3.y;
        }
    }
    if (acc == 0x12345678) sink[threadIdx.x] = (uint32_t)acc;
}

static int* d_a0 = nullptr;
static int* d_a1 = nullptr;
static uint32_t* d_sink = nullptr;

// returns milliseconds of `reps` launches (4 read instructions per iteration and wave); < 0 on error
extern "C" float ldsbench_run(const int* a0, const int* a1, int width, int iters, int blocks, int reps)
{
    if (!d_a0) {
        if (hipMalloc(&d_a0, 256) != hipSuccess || hipMalloc(&d_a1, 256) != hipSuccess || hipMalloc(&d_sink, 1024) != hipSuccess) return -1.0f;
    }
    if (hipMemcpy(d_a0, a0, 256, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_a1, a1, 256, hipMemcpyHostToDevice) != hipSuccess) return -1.0f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (width == 16) hipLaunchKernelGGL(lds_replay<16>, dim3(blocks), dim3(256), 32768 + 16, 0, d_a0, d_a1, iters, d_sink);
        else hipLaunchKernelGGL(lds_replay<8>, dim3(blocks), dim3(256), 32768 + 16, 0, d_a0, d_a1, iters, d_sink);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess) return -2.0f;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -3.0f;
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}
