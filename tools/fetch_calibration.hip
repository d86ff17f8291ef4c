// fetch_calibration.hip -- what does rocprofv3's FETCH_SIZE count on gfx950 for the access patterns of
// the accumulate kernel?  (MI355X_MICROARCH.md: FETCH_SIZE is exactly 1/2 of the bytes of a wide
// coalesced streaming read; "other access widths are uncalibrated: calibrate on a known byte count in
// your own access pattern".)
//
// Four kernels over a 1 GiB array (4x the 256 MiB Infinity Cache, so nothing is served on-die), each
// touching every byte's cache line exactly ONCE, so the true memory-side traffic is known up to the
// fetch granule:
//   stream16 : lane i reads 16 B at 16 i                         -- the guide's reference pattern
//   line16   : lane i reads 16 B at 128 i  (one piece per 128-B line; 1/8 of the array's lines... all lines of
//              the first 1/8 th)                                 -- is a sparse touch a 64-B or a 128-B fetch?
//   rec36seq : lane i reads 16 + 16 + 4 B of 48-byte record i    -- the accumulate kernel's SOURCE records
//   rec36rnd : the same from record perm[i], perm a random permutation -- its TARGET gathers
// Build + run on the GPU box (tools/run_fetch_calibration.sh): every kernel is launched once per
// rocprofv3 pass; wall time per kernel comes from HIP events in a separate, unprofiled run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void stream16(const v4f* __restrict__ a, size_t n16, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n16) return;
  const v4f v = a[i];
  if (v.x == 12345.678f) out[0] = v.y;  // never true: keeps the load
}

__global__ void line16(const char* __restrict__ a, size_t n_lines, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_lines) return;
  const v4f v = *(const v4f*)(a + i * 128);
  if (v.x == 12345.678f) out[0] = v.y;
}

__global__ void rec36(const char* __restrict__ a, const unsigned* __restrict__ perm, size_t n_rec, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rec) return;
  const char* r = a + (size_t)(perm ? perm[i] : (unsigned)i) * 48;
  const v4f p = *(const v4f*)r, q = *(const v4f*)(r + 16);
  const float z = *(const float*)(r + 32);
  if (p.x + q.x + z == 12345.678f) out[0] = p.y;
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)1 << 30;
  const size_t n16 = bytes / 16, n_lines = bytes / 128 / 8, n_rec = bytes / 48;
  char* a = nullptr;
  float* out = nullptr;
  unsigned* perm = nullptr;
  CHECK(hipMalloc((void**)&a, bytes));
  CHECK(hipMemset(a, 0, bytes));
  CHECK(hipMalloc((void**)&out, 64));
  {
    std::vector<unsigned> h(n_rec);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937_64 rng(7);
    for (size_t i = n_rec - 1; i > 0; --i) std::swap(h[i], h[rng() % (i + 1)]);
    CHECK(hipMalloc((void**)&perm, n_rec * sizeof(unsigned)));
    CHECK(hipMemcpy(perm, h.data(), n_rec * sizeof(unsigned), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto timed = [&](const char* name, double true_lo, double true_hi, auto launch) {
    launch();  // warm-up (page tables, code)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%-9s %8.3f ms   bytes touched: %.1f MB useful, %.1f MB in 64-B granules, %.1f MB in 128-B lines\n", name, ms, true_lo / 1e6,
                true_hi / 2e6 > true_lo / 1e6 ? true_hi / 2e6 : true_lo / 1e6, true_hi / 1e6);
  };
  const int bs = 256;
  timed("stream16", (double)bytes, (double)bytes, [&] { hipLaunchKernelGGL(stream16, dim3((n16 + bs - 1) / bs), dim3(bs), 0, 0, (const v4f*)a, n16, out); });
  timed("line16", 16.0 * n_lines, 128.0 * n_lines, [&] { hipLaunchKernelGGL(line16, dim3((n_lines + bs - 1) / bs), dim3(bs), 0, 0, a, n_lines, out); });
  timed("rec36seq", 36.0 * n_rec, (double)bytes, [&] { hipLaunchKernelGGL(rec36, dim3((n_rec + bs - 1) / bs), dim3(bs), 0, 0, a, (const unsigned*)nullptr, n_rec, out); });
  timed("rec36rnd", 36.0 * n_rec, (double)bytes, [&] { hipLaunchKernelGGL(rec36, dim3((n_rec + bs - 1) / bs), dim3(bs), 0, 0, a, perm, n_rec, out); });
  std::printf("array %zu bytes; stream16 %zu lanes; line16 %zu lines; rec36 %zu records (+ %zu bytes of permutation for rec36rnd)\n", bytes, n16,
              n_lines, n_rec, n_rec * sizeof(unsigned));
  return 0;
}
