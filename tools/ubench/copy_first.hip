// copy_first: what the FIRST device<->host copy of each size class costs in a fresh process, and the second (round 5: the 20 MHz cell scan's
// first batch spent 8 ms in one 30 KB copy).  hipcc --offload-arch=gfx950 -O2 copy_first.hip -o copy_first
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(uint8_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (uint8_t)i; }
int main(int argc, char **argv)
{
    const char *order = argc > 1 ? argv[1] : "small-first";
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    std::vector<size_t> sizes = {64, 4096, 16384, 30720, 65536, 262144, 1048576};
    if (order[0] == 'b') sizes = {262144, 30720, 64, 1048576, 4096};
    void *pinned; hipHostMalloc(&pinned, 4 << 20, hipHostMallocDefault);
    std::vector<uint8_t> pageable(4 << 20);
    for (int pass = 0; pass < 2; pass++)
        for (size_t n : sizes) {
            uint8_t *d; hipMalloc((void **)&d, n);
            k_fill<<<(n + 255) / 256, 256, 0, s>>>(d, n); hipStreamSynchronize(s);
            double t0 = now(); hipMemcpyAsync(pinned, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t1 = now();
            hipMemcpyAsync(pageable.data(), d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t2 = now();
            hipMemcpyAsync(d, pinned, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t3 = now();
            hipMemcpyAsync(d, pageable.data(), n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t4 = now();
            printf("pass %d  %8zu B   D2H pinned %8.1f us  D2H pageable %8.1f us   H2D pinned %8.1f us  H2D pageable %8.1f us\n", pass, n, 1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2), 1e6 * (t4 - t3));
            hipFree(d);
        }
    return 0;
}
