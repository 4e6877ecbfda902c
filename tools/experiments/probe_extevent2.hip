// Probe: which sequences of external event nodes does a stream capture accept?  usage: probe_extevent2 <case>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
__global__ void nop(int *x) { if (x) *x = 1; }
#define P(label, call) do { printf(" | %s %s", label, hipGetErrorString(call)); fflush(stdout); } while (0)
int main(int argc, char **argv)
{
    const int c = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    int *x; hipMalloc(&x, 4);
    hipEvent_t stop_b, plain_b, stop_a, fresh, fresh2;
    hipEventCreateWithFlags(&stop_b, hipEventDisableTiming);
    hipEventCreateWithFlags(&plain_b, hipEventDisableTiming);
    hipEventCreateWithFlags(&stop_a, hipEventDisableTiming);
    hipEventCreateWithFlags(&fresh, hipEventDisableTiming);
    hipEventCreateWithFlags(&fresh2, hipEventDisableTiming);
    hipExtLaunchKernelGGL(nop, dim3(1), dim3(1), 0, b, nullptr, stop_b, 0, x);
    hipLaunchKernelGGL(nop, dim3(1), dim3(1), 0, b, x); hipEventRecord(plain_b, b);
    hipExtLaunchKernelGGL(nop, dim3(1), dim3(1), 0, a, nullptr, stop_a, 0, x);
    hipDeviceSynchronize();
    hipGraph_t g = nullptr;
    printf("case %d:", c);
    P("begin", hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed));
    hipLaunchKernelGGL(nop, dim3(1), dim3(1), 0, a, x);
    switch (c) {
    case 0: P("wait plain_b", hipStreamWaitEvent(a, plain_b, hipEventWaitExternal)); break;
    case 1: P("wait stop_b", hipStreamWaitEvent(a, stop_b, hipEventWaitExternal)); break;
    case 2: P("wait fresh", hipStreamWaitEvent(a, fresh, hipEventWaitExternal)); P("wait fresh2", hipStreamWaitEvent(a, fresh2, hipEventWaitExternal)); break;
    case 3: P("wait stop_b", hipStreamWaitEvent(a, stop_b, hipEventWaitExternal)); P("wait stop_a", hipStreamWaitEvent(a, stop_a, hipEventWaitExternal)); break;
    case 4: P("wait stop_b", hipStreamWaitEvent(a, stop_b, hipEventWaitExternal)); hipLaunchKernelGGL(nop, dim3(1), dim3(1), 0, a, x); P("record fresh", hipEventRecordWithFlags(fresh, a, hipEventRecordExternal)); break;
    case 5: P("record stop_a", hipEventRecordWithFlags(stop_a, a, hipEventRecordExternal)); P("record fresh", hipEventRecordWithFlags(fresh, a, hipEventRecordExternal)); break;
    case 6: P("wait fresh", hipStreamWaitEvent(a, fresh, hipEventWaitExternal)); P("record fresh2", hipEventRecordWithFlags(fresh2, a, hipEventRecordExternal)); P("wait stop_b", hipStreamWaitEvent(a, stop_b, hipEventWaitExternal)); P("record stop_a", hipEventRecordWithFlags(stop_a, a, hipEventRecordExternal)); break;
    }
    hipLaunchKernelGGL(nop, dim3(1), dim3(1), 0, a, x);
    P("end", hipStreamEndCapture(a, &g));
    size_t n = 0;
    if (g) hipGraphGetNodes(g, nullptr, &n);
    printf(" | %zu nodes\n", n);
    return 0;
}
