// Probe (ROCm 7.2, gfx950): which uses of external event nodes does stream capture / the graph API survive?
// Every scenario runs in a forked child, so an abort inside the runtime is reported and the next one still runs.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
__global__ void k_write(int *p, int v) { if (threadIdx.x == 0) { for (volatile int i = 0; i < 100000; i++) {} *p = v; } }
__global__ void k_copy(const int *p, int *q) { if (threadIdx.x == 0) *q = *p; }

static int scenario(int which)
{
    hipStream_t s[4];
    for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    hipEvent_t ev, ev2;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    int *p, *q;
    CK(hipMalloc(&p, 4)); CK(hipMalloc(&q, 4)); CK(hipMemset(p, 0, 4)); CK(hipMemset(q, 0, 4));
    CK(hipDeviceSynchronize());
    hipGraph_t ga = nullptr, gb = nullptr;
    switch (which) {
    case 1:  // b alone captures; first node: external wait on an event recorded normally on a (a not capturing)
        CK(hipEventRecord(ev, s[0])); CK(hipDeviceSynchronize());
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        CK(hipStreamWaitEvent(s[1], ev, hipEventWaitExternal));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        break;
    case 2:  // the same, the event never recorded
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        CK(hipStreamWaitEvent(s[1], ev, hipEventWaitExternal));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        break;
    case 3:  // b alone; a kernel first, then the external wait, then a kernel
        CK(hipEventRecord(ev, s[0])); CK(hipDeviceSynchronize());
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s[1], p, 3);
        CK(hipStreamWaitEvent(s[1], ev, hipEventWaitExternal));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        break;
    case 4: {  // a alone: kernel, external record, kernel; then b alone: external wait, kernel (captured one after the other)
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s[0], p, 7);
        CK(hipEventRecordWithFlags(ev, s[0], hipEventRecordExternal));
        CK(hipStreamEndCapture(s[0], &ga));
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        CK(hipStreamWaitEvent(s[1], ev, hipEventWaitExternal));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        break;
    }
    case 5: {  // explicit graph API: empty root -> event wait node -> (kernel captured separately is not needed) record node
        CK(hipEventRecord(ev, s[0])); CK(hipDeviceSynchronize());
        CK(hipGraphCreate(&gb, 0));
        hipGraphNode_t w, r;
        CK(hipGraphAddEventWaitNode(&w, gb, nullptr, 0, ev));
        CK(hipGraphAddEventRecordNode(&r, gb, &w, 1, ev2));
        break;
    }
    case 6: {  // captured kernel chain, event nodes spliced in afterwards with the graph API
        CK(hipEventRecord(ev, s[0])); CK(hipDeviceSynchronize());
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s[1], p, 5);
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        size_t n = 0;
        CK(hipGraphGetNodes(gb, nullptr, &n));
        hipGraphNode_t nodes[8];
        CK(hipGraphGetNodes(gb, nodes, &n));
        size_t nr = 0;
        CK(hipGraphGetRootNodes(gb, nullptr, &nr));
        hipGraphNode_t root;
        nr = 1;
        CK(hipGraphGetRootNodes(gb, &root, &nr));
        hipGraphNode_t other = nodes[0] == root ? nodes[1] : nodes[0];
        hipGraphNode_t w, r;
        CK(hipGraphAddEventWaitNode(&w, gb, nullptr, 0, ev));      // wait in front of the first kernel
        CK(hipGraphAddDependencies(gb, &w, &root, 1));
        CK(hipGraphAddEventRecordNode(&r, gb, &root, 1, ev2));     // record between the two kernels
        CK(hipGraphAddDependencies(gb, &r, &other, 1));
        printf("    %zu captured nodes, 2 event nodes spliced in\n", n);
        break;
    }
    case 7: {  // a alone: kernel, external record; b alone: kernel, WAIT NODE ADDED BY HAND into the open capture, kernel
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(k_write, dim3(1), dim3(64), 0, s[0], p, 9);
        CK(hipEventRecordWithFlags(ev, s[0], hipEventRecordExternal));
        CK(hipStreamEndCapture(s[0], &ga));
        CK(hipStreamBeginCapture(s[1], hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        hipStreamCaptureStatus cs;
        unsigned long long id = 0;
        hipGraph_t g = nullptr;
        const hipGraphNode_t *deps = nullptr;
        size_t nd = 0;
        CK(hipStreamGetCaptureInfo_v2(s[1], &cs, &id, &g, &deps, &nd));
        hipGraphNode_t w;
        CK(hipGraphAddEventWaitNode(&w, g, deps, nd, ev));
        CK(hipStreamUpdateCaptureDependencies(s[1], &w, 1, hipStreamSetCaptureDependencies));
        hipLaunchKernelGGL(k_copy, dim3(1), dim3(64), 0, s[1], p, q);
        CK(hipStreamEndCapture(s[1], &gb));
        printf("    wait node spliced into the open capture behind %zu dependency node(s)\n", nd);
        break;
    }
    }
    hipGraphExec_t xa = nullptr, xb = nullptr;
    if (ga) CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
    CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; rep++) {
        if (xa) CK(hipGraphLaunch(xa, s[0]));
        CK(hipGraphLaunch(xb, s[1]));
        CK(hipDeviceSynchronize());
    }
    int h = -1;
    CK(hipMemcpy(&h, q, 4, hipMemcpyDeviceToHost));
    printf("    ok, q = %d\n", h);
    return 0;
}

int main()
{
    const char *names[] = {"", "capture: first node = external wait on a normally recorded event", "capture: external wait on a never recorded event",
                           "capture: kernel, external wait, kernel", "two captures one after the other: record in one, wait in the other",
                           "graph API: event wait + record nodes only", "captured kernel chain + event nodes spliced in with the graph API",
                           "open capture: hipGraphAddEventWaitNode + hipStreamUpdateCaptureDependencies behind a kernel"};
    for (int w = 1; w <= 7; w++) {
        printf("scenario %d: %s\n", w, names[w]);
        fflush(stdout);
        const pid_t pid = fork();
        if (pid == 0)
        { const int rc_ = scenario(w); fflush(stdout); _exit(rc_); }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st))
            printf("    CRASHED with signal %d\n", WTERMSIG(st));
        else if (WEXITSTATUS(st))
            printf("    failed\n");
        fflush(stdout);
    }
    return 0;
}
