// PCIe probe: what the host-buffer entry points can hope for.  hipcc -O2 -o /tmp/pcie_probe scripts/probes/pcie_probe.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 256u << 20;
    char *pageable = (char *)malloc(n), *pinned = nullptr, *dev = nullptr;
    memset(pageable, 1, n);
    hipHostMalloc((void **)&pinned, n);
    memset(pinned, 2, n);
    hipMalloc((void **)&dev, n);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(dev, pageable, n, hipMemcpyHostToDevice); double a = now() - t;
        t = now(); hipMemcpy(dev, pinned, n, hipMemcpyHostToDevice); double b = now() - t;
        t = now(); hipMemcpy(pageable, dev, n, hipMemcpyDeviceToHost); double c = now() - t;
        t = now(); hipMemcpy(pinned, dev, n, hipMemcpyDeviceToHost); double d = now() - t;
        printf("H2D pageable %.1f GB/s, pinned %.1f GB/s; D2H pageable %.1f GB/s, pinned %.1f GB/s\n", n / a / 1e9, n / b / 1e9, n / c / 1e9, n / d / 1e9);
    }
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpyAsync(dev, pageable, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double a = now() - t;
        t = now(); hipMemcpyAsync(pageable, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double c = now() - t;
        printf("hipMemcpyAsync on a stream + sync: H2D pageable %.1f GB/s, D2H pageable %.1f GB/s\n", n / a / 1e9, n / c / 1e9);
    }
    {   // a buffer that is new every time (a fresh Go slice / numpy array)
        for (int rep = 0; rep < 2; rep++) {
            char *fresh = (char *)malloc(n); memset(fresh, 3, n);
            double t = now(); hipMemcpy(dev, fresh, n, hipMemcpyHostToDevice); double a = now() - t;
            char *fresh2 = (char *)malloc(n);
            t = now(); hipMemcpy(fresh2, dev, n, hipMemcpyDeviceToHost); double c = now() - t;
            printf("fresh buffers: H2D %.1f GB/s, D2H (untouched destination) %.1f GB/s\n", n / a / 1e9, n / c / 1e9);
            free(fresh); free(fresh2);
        }
    }
    for (int T : {1, 2, 4, 8}) {
        double t = now();
        std::vector<std::thread> th;
        for (int i = 0; i < T; i++) th.emplace_back([&, i] { memcpy(pinned + n / T * i, pageable + n / T * i, n / T); });
        for (auto &x : th) x.join();
        printf("memcpy pageable -> pinned, %d threads: %.1f GB/s\n", T, n / (now() - t) / 1e9);
    }
    return 0;
}
