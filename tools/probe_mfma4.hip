// Probe: lane maps of v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks per instruction) on gfx950, by one-hot inputs.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_mfma4 probe_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* a, const double* b, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  for (int pass = 0; pass < 2; ++pass) {
    printf("%s one-hot lane -> output lanes that see it (with the other operand all ones)\n", pass == 0 ? "A" : "B");
    for (int h = 0; h < 64; ++h) {
      for (int l = 0; l < 64; ++l) { ha[l] = pass == 0 ? (l == h) : 1.0; hb[l] = pass == 1 ? (l == h) : 1.0; }
      hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
      k<<<1, 64>>>(da, db, dd); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
      printf("%2d:", h);
      for (int l = 0; l < 64; ++l) if (hd[l] != 0.0) printf(" %d", l);
      printf("\n");
    }
  }
  return 0;
}
