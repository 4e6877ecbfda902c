// Development probe: does hipGraphLaunch return before the graph can run?  (Round 3: replays of per-stream graphs did
// not overlap - the FFT graph of replay r+1 started only when the host had got through replay r's launches.)
// Stream A runs a ~3 ms spinning kernel and records an event; stream B waits for the event and then gets either an eager
// kernel or a one-kernel graph.  Host time of each call is printed.
//   hipcc -O2 --offload-arch=gfx950 -o tools/bin/probe_graph_async tools/experiments/probe_graph_async.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void k_spin(long long clocks, int *out)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) {
    }
    if (out)
        *out = 1;
}
__global__ void k_small(int *out) { *out = 2; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    int *d;
    CK(hipMalloc(&d, 64));
    const long long spin = 300000;  // wall_clock64 runs at 100 MHz: 3 ms
    // graphs: one kernel; six kernels
    hipGraph_t g1, g6;
    hipGraphExec_t x1, x6;
    CK(hipStreamBeginCapture(B, hipStreamCaptureModeRelaxed));
    hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, B, d);
    CK(hipStreamEndCapture(B, &g1));
    CK(hipGraphInstantiate(&x1, g1, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(B, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 6; i++)
        hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, B, d + i);
    CK(hipStreamEndCapture(B, &g6));
    CK(hipGraphInstantiate(&x6, g6, nullptr, nullptr, 0));
    // warm
    CK(hipGraphLaunch(x1, B));
    CK(hipGraphLaunch(x6, B));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, A, 1000, d + 8);
    CK(hipDeviceSynchronize());

    for (int scenario = 0; scenario < 6; scenario++) {
        const char *name[] = {"eager kernel behind a wait for a pending event",
                              "graph (1 kernel) behind a wait for a pending event",
                              "graph (6 kernels) behind a wait for a pending event",
                              "graph (1 kernel) behind a pending kernel of the SAME stream",
                              "graph (1 kernel) on an idle stream, nothing pending",
                              "graph (1 kernel) launched twice in a row behind a pending wait"};
        double t0 = now_us();
        if (scenario != 4) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, scenario == 3 ? B : A, spin, d + 8);
            if (scenario != 3) {
                CK(hipEventRecord(ev, A));
                CK(hipStreamWaitEvent(B, ev, 0));
            }
        }
        const double t1 = now_us();
        if (scenario == 0)
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, B, d);
        else if (scenario == 2)
            CK(hipGraphLaunch(x6, B));
        else
            CK(hipGraphLaunch(x1, B));
        const double t2 = now_us();
        if (scenario == 5)
            CK(hipGraphLaunch(x1, B));
        const double t3 = now_us();
        CK(hipEventRecord(ev2, B));
        const double t4 = now_us();
        CK(hipDeviceSynchronize());
        const double t5 = now_us();
        printf("%-62s setup %7.1f us | launch %8.1f us | 2nd %8.1f us | record %6.1f us | drain %8.1f us\n", name[scenario], t1 - t0, t2 - t1,
               t3 - t2, t4 - t3, t5 - t4);
    }
    return 0;
}
