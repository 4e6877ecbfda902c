// Probe: do external event record / wait nodes inside captured single-stream graphs order two graphs launched on two
// streams the way hipEventRecord / hipStreamWaitEvent would?  Graph A (stream a): wait(F) ; slow kernel writes x = r ;
// record(E).  Graph B (stream b): wait(E) ; kernel copies x to y[r] ; record(F).  Launched A, B, A, B, ...; y[r] must
// be r for every replay r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void producer(volatile int *x, const int *replay, int spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    *x = *replay;
}
__global__ void bump(int *replay) { *replay += 1; }
__global__ void consumer(const volatile int *x, int *y, const int *replay_b) { y[*replay_b] = *x; }
__global__ void bump_b(int *replay_b) { *replay_b += 1; }
int main()
{
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t E, F;
    CHECK(hipEventCreateWithFlags(&E, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&F, hipEventDisableTiming));
    int *x, *y, *ra, *rb;
    const int R = 200;
    CHECK(hipMalloc(&x, 4)); CHECK(hipMalloc(&y, 4 * R)); CHECK(hipMalloc(&ra, 4)); CHECK(hipMalloc(&rb, 4));
    CHECK(hipMemset(x, 0xff, 4)); CHECK(hipMemset(y, 0xff, 4 * R)); CHECK(hipMemset(ra, 0, 4)); CHECK(hipMemset(rb, 0, 4));
    CHECK(hipDeviceSynchronize());
    hipGraph_t ga, gb;
    CHECK(hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed));
    CHECK(hipStreamWaitEvent(a, F, hipEventWaitExternal));
    hipLaunchKernelGGL(producer, dim3(1), dim3(1), 0, a, x, ra, 2000);  // 20 us
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, a, ra);
    CHECK(hipEventRecordWithFlags(E, a, hipEventRecordExternal));
    CHECK(hipStreamEndCapture(a, &ga));
    CHECK(hipStreamBeginCapture(b, hipStreamCaptureModeRelaxed));
    CHECK(hipStreamWaitEvent(b, E, hipEventWaitExternal));
    hipLaunchKernelGGL(consumer, dim3(1), dim3(1), 0, b, x, y, rb);
    hipLaunchKernelGGL(bump_b, dim3(1), dim3(1), 0, b, rb);
    CHECK(hipEventRecordWithFlags(F, b, hipEventRecordExternal));
    CHECK(hipStreamEndCapture(b, &gb));
    hipGraphExec_t ea, eb;
    CHECK(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0));
    CHECK(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0));
    size_t na = 0, nb = 0;
    CHECK(hipGraphGetNodes(ga, nullptr, &na)); CHECK(hipGraphGetNodes(gb, nullptr, &nb));
    printf("graph A %zu nodes, graph B %zu nodes\n", na, nb);
    for (int r = 0; r < R; r++) {
        CHECK(hipGraphLaunch(ea, a));
        CHECK(hipGraphLaunch(eb, b));
    }
    CHECK(hipDeviceSynchronize());
    std::vector<int> h(R);
    CHECK(hipMemcpy(h.data(), y, 4 * R, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int r = 0; r < R; r++)
        if (h[r] != r) { if (bad < 8) printf("replay %d: consumer saw %d\n", r, h[r]); bad++; }
    printf("%d replays, %d out of order\n", R, bad);
    return bad ? 2 : 0;
}
