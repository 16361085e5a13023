// How long does hipMemcpyAsync take from / to freshly allocated pageable host memory (1.18 MB images, 40-KB result arrays),
// against the same bytes staged through ONE pinned buffer?  (bin/sfm_native's 20 - 37 ms stalls in some frames' uploads.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? atol(argv[1]) : 1179648;
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    void *dev; hipMalloc(&dev, n);
    void *pin; hipHostMalloc(&pin, n, hipHostMallocDefault);
    std::vector<std::vector<unsigned char>> keep;
    for (int mode = 0; mode < 3; ++mode) {
        printf("%s:", mode == 0 ? "pageable H2D, fresh vector each" : mode == 1 ? "pageable D2H, fresh vector each" : "staged through pinned (memcpy + H2D)");
        for (int it = 0; it < 16; ++it) {
            keep.emplace_back(n, (unsigned char)it);                  // fresh heap memory, kept alive like the frames' images
            std::vector<unsigned char> junk(300000 + 4096 * it, 1);   // heap churn between the frames
            unsigned char *h = keep.back().data();
            const double t0 = now();
            if (mode == 0) hipMemcpyAsync(dev, h, n, hipMemcpyHostToDevice, st);
            else if (mode == 1) hipMemcpyAsync(h, dev, n, hipMemcpyDeviceToHost, st);
            else { memcpy(pin, h, n); hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st); }
            hipStreamSynchronize(st);
            printf(" %.2f", (now() - t0) * 1e3);
        }
        printf(" ms\n");
    }
    return 0;
}
