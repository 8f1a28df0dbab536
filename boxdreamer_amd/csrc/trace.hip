// Optional launch tracing for bench.py's roofline: HIP events recorded on the launch stream around
// every MFMA-kernel launch (GEMM, attention) while a trace is active.  This is the only state the
// library ever holds, it is off by default and it is process-global (one trace at a time).
#include <vector>

#include "bd_common.h"

namespace {
struct Slot { hipEvent_t e0, e1; int kind, M, N, K; };
std::vector<Slot> g_slots;
int g_count = -1;   // -1: tracing off
}  // namespace

// Event records cannot be timed once captured into a HIP graph (ROCm 7.2: hipEventElapsedTime -> hipErrorInvalidHandle,
// and hipEventRecordExternal nodes are refused during torch's capture), so launches issued while the stream is
// capturing are simply not traced; bench.py times graph replays and traces the same step un-graphed right after.
static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

int bd_trace_open(hipStream_t s, int kind, int M, int N, int K) {
    if (g_count < 0 || g_count >= (int)g_slots.size() || capturing(s)) return -1;
    Slot& sl = g_slots[g_count];
    sl.kind = kind; sl.M = M; sl.N = N; sl.K = K;
    (void)hipEventRecord(sl.e0, s);
    return g_count++;
}

void bd_trace_close(hipStream_t s, int slot) {
    if (slot >= 0) (void)hipEventRecord(g_slots[slot].e1, s);
}

extern "C" int bd_trace_begin(int capacity) {
    if (capacity <= 0) return BD_ERR_SHAPE;
    while ((int)g_slots.size() < capacity) {
        Slot sl{};
        hipError_t e = hipEventCreate(&sl.e0);
        if (e != hipSuccess) return (int)e;
        e = hipEventCreate(&sl.e1);
        if (e != hipSuccess) return (int)e;
        g_slots.push_back(sl);
    }
    g_count = 0;
    return BD_OK;
}

extern "C" int bd_trace_end(bd_trace_record* out, int capacity) {
    if (g_count < 0) return 0;
    const int n = g_count < capacity ? g_count : capacity;
    g_count = -1;
    for (int i = 0; i < n; ++i) {
        Slot& sl = g_slots[i];
        (void)hipEventSynchronize(sl.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, sl.e0, sl.e1);
        if (out) { out[i].kind = sl.kind; out[i].M = sl.M; out[i].N = sl.N; out[i].K = sl.K; out[i].ms = ms; }
    }
    return n;
}
