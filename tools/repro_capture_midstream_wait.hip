// Attempt at a standalone reproducer of the capture fault met three times in rounds 3 and 4: while a sub-step is being
// recorded (thread-local stream capture, the capturing stream forks to side streams by events), a SIDE stream that waits for an
// event recorded on another stream AFTER its own fork and then takes more work made the process die inside hipStreamEndCapture
// (torch 2.10 + ROCm 7.2: `_contract` moved onto sweep A's stream; the boundary sweep on the second side stream behind `k_bdry`,
// waited for by the first).  Variants here:  1 = the shape that works in the engine (side streams only wait when they are entered),
// 2 = a side stream waits mid-way for an event of the capturing stream, 3 = a side stream waits mid-way for another side stream.
//     hipcc --offload-arch=gfx950 -O2 tools/repro_capture_midstream_wait.hip -o /tmp/repro_cap && /tmp/repro_cap
// Prints one line per variant ("variant k: captured, N nodes, launched ok") or dies where the runtime does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void work(double* p, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0;
}

int main() {
  const int n = 1 << 16;
  double* buf;
  CK(hipMalloc(&buf, 6 * n * sizeof(double)));
  CK(hipMemset(buf, 0, 6 * n * sizeof(double)));
  hipStream_t cap, s1, s2;
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev[8];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const dim3 g((n + 255) / 256), b(256);
  for (int variant = 1; variant <= 3; ++variant) {
    CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(ev[0], cap));                                  // fork
    work<<<g, b, 0, cap>>>(buf, n, 1.0001);                           // "test network" on the capturing stream
    CK(hipStreamWaitEvent(s1, ev[0], 0));
    work<<<g, b, 0, s1>>>(buf + n, n, 1.0002);                        // forward pass on side stream 1
    CK(hipEventRecord(ev[1], s1));
    CK(hipStreamWaitEvent(s2, ev[1], 0));                             // side stream 2 is entered behind the forward pass
    work<<<g, b, 0, s2>>>(buf + 2 * n, n, 1.0003);                    // boundary residual
    CK(hipEventRecord(ev[2], s2));
    work<<<g, b, 0, s1>>>(buf + n, n, 1.0004);                        // sweeps A
    CK(hipEventRecord(ev[3], cap));                                   // behind the test network
    if (variant == 2) {
      CK(hipStreamWaitEvent(s1, ev[3], 0));                           // MID-WAY: side 1 waits for the capturing stream's later event
      work<<<g, b, 0, s1>>>(buf + 4 * n, n, 1.0006);                  // (the reduction on sweep A's stream)
    }
    if (variant == 3) {
      work<<<g, b, 0, s2>>>(buf + 5 * n, n, 1.0007);                  // more work on side 2 behind its recorded event ...
      CK(hipEventRecord(ev[5], s2));
      CK(hipStreamWaitEvent(s1, ev[5], 0));                           // ... and side 1 waits for it MID-WAY
      work<<<g, b, 0, s1>>>(buf + 4 * n, n, 1.0008);
    }
    CK(hipEventRecord(ev[4], s1));
    CK(hipStreamWaitEvent(cap, ev[1], 0));
    work<<<g, b, 0, cap>>>(buf + 3 * n, n, 1.0005);                   // sweep B behind test network and forward pass
    CK(hipStreamWaitEvent(cap, ev[4], 0));                            // joins
    CK(hipStreamWaitEvent(cap, ev[2], 0));
    work<<<g, b, 0, cap>>>(buf, n, 1.0009);                           // update
    hipGraph_t graph;
    CK(hipStreamEndCapture(cap, &graph));
    size_t nodes = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nodes));
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int k = 0; k < 20; ++k) CK(hipGraphLaunch(exec, cap));
    CK(hipStreamSynchronize(cap));
    printf("variant %d: captured, %zu nodes, launched ok\n", variant, nodes);
    fflush(stdout);
    // (kept alive on purpose: destroying graphs is the OTHER fault of this stack, tools/repro_graph_destroy.hip)
  }
  return 0;
}
