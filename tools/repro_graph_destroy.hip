// Attempt at a standalone reproducer of the runtime fault behind engine._KEPT_GRAPHS: destroying the
// executable of a multi-branch captured HIP graph made a LATER launch of another, live graph fault inside the runtime
// (hip::Graph::UpdateStreams under hipGraphLaunch; found in a sequence of 27 GPU tests, when Python's collector destroyed
// the graphs of a solver that had gone out of scope).  This program builds graphs of the engine's shape -- a main stream
// that forks to three side streams and joins them, a dozen kernel nodes, captured in thread-local mode -- and destroys some
// of them between launches of the others, in several orders.
//     hipcc --offload-arch=gfx950 -O2 tools/repro_graph_destroy.hip -o /tmp/repro && /tmp/repro [rounds]
// Prints "no fault in N rounds" or dies where the runtime does.  Result on ROCm 7.2 / MI355X: see DESIGN 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void work(double* p, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0;
}

struct G { hipGraph_t g; hipGraphExec_t e; };

G capture(hipStream_t cap, hipStream_t* side, hipEvent_t* ev, double* buf, int n, int variant) {
  G r;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(ev[0], cap));
  work<<<(n + 255) / 256, 256, 0, cap>>>(buf, n, 1.0001);                         // "test network"
  for (int s = 0; s < 3; ++s) CK(hipStreamWaitEvent(side[s], ev[0], 0));
  work<<<(n + 255) / 256, 256, 0, side[0]>>>(buf + n, n, 1.0002);                 // forward
  CK(hipEventRecord(ev[1], side[0]));
  CK(hipStreamWaitEvent(side[1], ev[1], 0));
  work<<<(n + 255) / 256, 256, 0, side[1]>>>(buf + 2 * n, n, 1.0003);             // boundary residual
  CK(hipEventRecord(ev[2], side[1]));
  work<<<(n + 255) / 256, 256, 0, side[0]>>>(buf + n, n, 1.0004);                 // sweeps A
  CK(hipEventRecord(ev[3], side[0]));
  CK(hipEventRecord(ev[4], cap));
  CK(hipStreamWaitEvent(cap, ev[1], 0));
  work<<<(n + 255) / 256, 256, 0, cap>>>(buf + 3 * n, n, 1.0005);                 // sweep B
  CK(hipStreamWaitEvent(side[2], ev[3], 0));
  CK(hipStreamWaitEvent(side[2], ev[4], 0));
  CK(hipStreamWaitEvent(side[2], ev[2], 0));
  for (int k = 0; k < 1 + variant; ++k) work<<<(n + 255) / 256, 256, 0, side[2]>>>(buf + 4 * n, n, 1.0006);   // reduction
  CK(hipEventRecord(ev[5], side[2]));
  CK(hipStreamWaitEvent(cap, ev[5], 0));
  work<<<(n + 255) / 256, 256, 0, cap>>>(buf, n, 1.0007);                         // Adam
  CK(hipStreamEndCapture(cap, &r.g));
  CK(hipGraphInstantiate(&r.e, r.g, nullptr, nullptr, 0));
  return r;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 200, n = 1 << 16;
  double* buf;
  CK(hipMalloc(&buf, sizeof(double) * 5 * n));
  CK(hipMemset(buf, 0, sizeof(double) * 5 * n));
  hipStream_t cap, run, side[3];
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&run, hipStreamNonBlocking));
  for (auto& s : side) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[6];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  std::vector<G> live;
  for (int r = 0; r < rounds; ++r) {
    // a "solver": three graphs (generator, discriminator, a third segment); replay them a few times
    std::vector<G> mine;
    for (int v = 0; v < 3; ++v) mine.push_back(capture(cap, side, ev, buf, n, v));
    for (int it = 0; it < 4; ++it)
      for (auto& g : mine) CK(hipGraphLaunch(g.e, run));
    // keep one solver alive, let the previous one "go out of scope" in different orders relative to launches of the live ones
    if (!live.empty()) {
      if (r % 3 == 0) CK(hipStreamSynchronize(run));
      for (size_t k = 0; k < live.size(); ++k) {
        if (r % 2) { CK(hipGraphExecDestroy(live[k].e)); CK(hipGraphDestroy(live[k].g)); }
        else { CK(hipGraphDestroy(live[k].g)); CK(hipGraphExecDestroy(live[k].e)); }
        CK(hipGraphLaunch(mine[k % mine.size()].e, run));      // a live graph launched right behind a destruction
      }
    }
    live = mine;
    if (r % 5 == 4) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  printf("no fault in %d rounds (%d graphs created, destroyed between launches of live ones)\n", rounds, 3 * rounds);
  return 0;
}
