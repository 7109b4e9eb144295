// hipExtStreamCreateWithCUMask on an MI355X (8 XCCs x 32 CUs): which CUs do the blocks of a masked stream reach -- launched eagerly,
// and as a kernel node of a graph captured from that stream?     hipcc --offload-arch=gfx950 -O3 -o probe_cumask probe_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_where(unsigned* out, long spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void report(const char* what, const std::vector<unsigned>& h, int blocks) {
  std::map<unsigned, int> perCu, perXcc;
  for (int b = 0; b < blocks; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    perCu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    perXcc[xcc]++;
  }
  printf("%-58s CUs used %3zu; blocks per XCC:", what, perCu.size());
  for (auto& kv : perXcc) printf(" %d", kv.second);
  printf("\n");
}
int main() {
  unsigned* d;
  const int blocks = 1024;
  CK(hipMalloc(&d, 2 * blocks * sizeof(unsigned)));
  std::vector<unsigned> h(2 * blocks);
  struct M { const char* name; unsigned w[8]; } masks[] = {
    {"all 256 bits", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
    {"bits 0..63", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
    {"bits 0..191", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0, 0}},
    {"bits 192..255", {0, 0, 0, 0, 0, 0, ~0u, ~0u}},
    {"every 4th bit (64 bits)", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}},
    {"three of every 4 bits (192 bits)", {0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu, 0xeeeeeeeeu}},
    {"bits 0..7 of every 32", {0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu}},
  };
  for (auto& m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
    if (e != hipSuccess) { printf("%-40s hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e)); continue; }
    CK(hipMemsetAsync(d, 0, 2 * blocks * sizeof(unsigned), s));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 2000L);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    char buf[128];
    snprintf(buf, sizeof buf, "eager, mask = %s", m.name);
    report(buf, h, blocks);
    // the same launch as a node of a graph captured from the masked stream, replayed on the masked stream and on the null stream
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(256), 0, s, d, 2000L);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    snprintf(buf, sizeof buf, "  graph replayed on the masked stream");
    report(buf, h, blocks);
    CK(hipGraphLaunch(ge, 0)); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    snprintf(buf, sizeof buf, "  graph replayed on the null stream");
    report(buf, h, blocks);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
  }
  return 0;
}
