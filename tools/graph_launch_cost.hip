// What does the HOST pay per Krylov step on short vectors, and what would a captured graph save?  
// One "step" here has the shape of a projection-route expand! at 1e5 rows: NK small kernels (apply, finalize, scale, project, finalize,
// solve, unproject, finalize), one 8 KB device-to-host copy into pinned memory, one event record.  Measured, per step, host time to
// ENQUEUE and wall time per step with the host waiting on the event of the step before (depth-1 run-ahead, as the library does):
//   (a) stream calls one by one,   (b) the same sequence captured once and replayed with hipGraphLaunch,
//   (c) ONE kernel + copy + event (what a fused whole-step kernel would cost).
// Build: hipcc --offload-arch=gfx950 -O3 tools/graph_launch_cost.hip -o tools/bin/graph_launch_cost ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_small(double* p, int n, double a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * a + 1.0;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int n = 100000, NK = 8, STEPS = 2000;
    double *d = nullptr, *h = nullptr, *ws = nullptr;
    CK(hipMalloc(&d, n * sizeof(double)));
    CK(hipMalloc(&ws, 1024 * sizeof(double)));
    CK(hipHostMalloc(&h, 2 * 1024 * sizeof(double)));
    CK(hipMemset(d, 0, n * sizeof(double)));
    CK(hipMemset(ws, 0, 1024 * sizeof(double)));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev[2];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto enqueue_step = [&](int slot, int nk) -> int {
        for (int k = 0; k < nk; ++k) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, s, d, n, 1.0);
        CK(hipMemcpyAsync(h + slot * 1024, ws, 1024 * sizeof(double), hipMemcpyDeviceToHost, s));
        return 0;
    };
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {
        hipGraphExec_t ge[2] = {nullptr, nullptr};
        if (mode == 1) {
            for (int slot = 0; slot < 2; ++slot) {
                hipGraph_t g;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                if (enqueue_step(slot, NK)) return 1;
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge[slot], g, nullptr, nullptr, 0));
                CK(hipGraphDestroy(g));
            }
        }
        double t_enq = 0;
        for (int rep = 0; rep < 2; ++rep) {   // rep 0 = warm-up
            CK(hipStreamSynchronize(s));
            t_enq = 0;
            const double t0 = now_us();
            for (int it = 0; it < STEPS; ++it) {
                const int slot = it & 1;
                const double a = now_us();
                if (mode == 1) CK(hipGraphLaunch(ge[slot], s));
                else if (enqueue_step(slot, mode == 0 ? NK : 1)) return 1;
                CK(hipEventRecord(ev[slot], s));
                t_enq += now_us() - a;
                if (it > 0) CK(hipEventSynchronize(ev[slot ^ 1]));   // the step before: its scalars are consumed now
            }
            CK(hipStreamSynchronize(s));
            res[mode][0] = t_enq / STEPS;
            res[mode][1] = (now_us() - t0) / STEPS;
        }
        for (int slot = 0; slot < 2; ++slot) if (ge[slot]) CK(hipGraphExecDestroy(ge[slot]));
    }
    printf("{\"tool\": \"graph_launch_cost\", \"rows\": %d, \"kernels_per_step\": %d, \"steps\": %d, "
           "\"stream_calls\": {\"host_enqueue_us\": %.2f, \"wall_us_per_step\": %.2f}, "
           "\"graph_replay\": {\"host_enqueue_us\": %.2f, \"wall_us_per_step\": %.2f}, "
           "\"one_kernel_per_step\": {\"host_enqueue_us\": %.2f, \"wall_us_per_step\": %.2f}}\n",
           n, NK, STEPS, res[0][0], res[0][1], res[1][0], res[1][1], res[2][0], res[2][1]);
    return 0;
}
