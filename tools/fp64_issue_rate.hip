// fp64_issue_rate.hip -- how fast does gfx950 issue FP64 FMAs at TWO waves per SIMD (the accumulate
// kernel's occupancy), as a function of the number of independent dependency chains per lane?
// Answers whether a 60 % VALU utilisation at 254 instructions per correspondence is a property of the
// kernel's dependent chains or of the FP64 pipe.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue_rate tools/fp64_issue_rate.hip && /tmp/fp64_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void chains(double* out, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) x[c] = threadIdx.x * 1e-9 + c;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  if (s == 12345.6789) out[0] = s;
}

template <int CHAINS, int W>
void run(double* out, int blocks) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chains<CHAINS, W>), dim3(blocks), dim3(256), 0, 0, out, 16, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((chains<CHAINS, W>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)blocks * 256 * iters * 16 * CHAINS;
  std::printf("chains %2d, %d wave(s)/SIMD (%d blocks): %7.3f ms  %6.1f TFLOP/s FP64 (%.0f %% of 78.6)\n", CHAINS, W, blocks, ms,
              2 * fmas / (ms * 1e-3) / 1e12, 100 * 2 * fmas / (ms * 1e-3) / 78.6e12);
}

int main() {
  double* out;
  hipMalloc((void**)&out, 64);
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  run<1, 1>(out, cus); run<2, 1>(out, cus); run<4, 1>(out, cus); run<8, 1>(out, cus);
  run<1, 2>(out, 2 * cus); run<2, 2>(out, 2 * cus); run<4, 2>(out, 2 * cus); run<8, 2>(out, 2 * cus);
  run<1, 4>(out, 4 * cus); run<4, 4>(out, 4 * cus);
  return 0;
}
