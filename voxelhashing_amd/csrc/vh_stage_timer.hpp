// vh_stage_timer.hpp -- per-stage device timing with HIP events recorded on
// the stream the kernels are launched on (the reference's TimingLog,
// DSC/TimingLog.h:21-46, without its cudaDeviceSynchronize per stage).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/vh_api.h"

#include <cstdint>
#include <utility>
#include <vector>

struct VhStageTimer {
    explicit VhStageTimer(int nStages) : totalMs(nStages, 0.0), count(nStages, 0), open(nStages, nullptr), pending(nStages) {}
    ~VhStageTimer()
    {
        for (auto& v : pending)
            for (auto& p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        for (hipEvent_t e : pool) (void)hipEventDestroy(e);
        for (hipEvent_t e : open)
            if (e) (void)hipEventDestroy(e);
    }
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        // timing only: device-scope release (a default event record makes the queue write back its caches and idles it
        // for ~6 us, which distorts the frame rate being measured)
        if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventCreate(&e);
        }
        return e;
    }
    void start(int stage, hipStream_t s)
    {
        hipEvent_t e = get();
        (void)hipEventRecord(e, s);
        open[stage] = e;
    }
    void stop(int stage, hipStream_t s)
    {
        hipEvent_t e = get();
        (void)hipEventRecord(e, s);
        pending[stage].push_back(std::make_pair(open[stage], e));
        open[stage] = nullptr;
    }
    // the NEXT kernel this thread launches (one of those that go through VH_LAUNCH_TIMED: the ray caster, computeNormals,
    // the fused integrate pass) is timed by its own dispatch time stamps: no record before or behind it
    void arm(int stage)
    {
        hipEvent_t a = get(), b = get();
        (void)vh_time_next_launch((void*)a, (void*)b);
        pending[stage].push_back(std::make_pair(a, b));
    }
    // waits for the stream and folds all finished pairs into the totals
    void resolve(hipStream_t s)
    {
        (void)hipStreamSynchronize(s);
        for (size_t st = 0; st < pending.size(); st++) {
            for (auto& p : pending[st]) {
                float ms = 0.0f;
                if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { totalMs[st] += ms; count[st]++; }
                pool.push_back(p.first);
                pool.push_back(p.second);
            }
            pending[st].clear();
        }
    }
    void clear()
    {
        for (size_t st = 0; st < totalMs.size(); st++) { totalMs[st] = 0.0; count[st] = 0; }
    }
    std::vector<double> totalMs;
    std::vector<uint64_t> count;
    std::vector<hipEvent_t> open;
    std::vector<std::vector<std::pair<hipEvent_t, hipEvent_t>>> pending;
    std::vector<hipEvent_t> pool;
};
