// zsw_timer.hpp — HIP-event timing of the dominant kernel, on the stream it is launched on.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>
#include <vector>

namespace zsw {

struct KernelTimer {
    bool enabled = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    size_t used = 0;

    void begin(hipStream_t s) {
        if (!enabled) return;
        if (used == pool.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            pool.emplace_back(a, b);
        }
        (void)hipEventRecord(pool[used].first, s);
    }
    void end(hipStream_t s) {
        if (!enabled || used >= pool.size()) return;
        (void)hipEventRecord(pool[used].second, s);
        ++used;
    }
    // Sums the recorded intervals (synchronises on each stop event) and resets.
    hipError_t collect(double* seconds, uint64_t* launches) {
        double total = 0;
        for (size_t i = 0; i < used; ++i) {
            hipError_t e = hipEventSynchronize(pool[i].second);
            if (e != hipSuccess) return e;
            float ms = 0;
            e = hipEventElapsedTime(&ms, pool[i].first, pool[i].second);
            if (e != hipSuccess) return e;
            total += ms * 1e-3;
        }
        *seconds = total;
        *launches = used;
        used = 0;
        return hipSuccess;
    }
    void destroy() {
        for (auto& p : pool) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        pool.clear();
        used = 0;
    }
};

}  // namespace zsw
