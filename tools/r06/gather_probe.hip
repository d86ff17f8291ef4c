// gather_probe.hip -- how the L1 / texture path of a CU prices the accumulate kernel's target gathers.
// PMC of the 256-pair accumulate launch: TD busy 84 %, 2.58 L1 accesses per slot, ~87 % of them the gathers (each lane
// fetches its 4 target records as 16 + 16 + 4 byte pieces: 12 LDS-DMA instructions per step, every lane its own line).
// Question: does the path charge per ACCESS (then lanes that cooperate on one record -- consecutive lanes fetching the
// consecutive pieces of one record -- would cut the cost) or per byte?  Variants, same records, same indices:
//   0  product pattern: lane l, slot c fetches record idx[l][c] as 16 + 16 + 4 from 48-byte records         (12 instr / step)
//   1  the same without the 4-byte piece (wrong data; prices one piece)                                       ( 8 instr / step)
//   2  three consecutive lanes fetch one 36-byte record as 3 x 12 bytes (dwordx3) from a dense 36-byte array (13 instr / step)
//   3  four consecutive lanes fetch one 64-byte-aligned 64-byte record as 4 x 16 bytes                        (16 instr / step)
//   4  variant 0's instruction count with every lane on the SAME record (all hits, one line per instruction: the floor)
// Every variant then reads the staged bytes back from LDS and folds them into a checksum.  usage: gather_probe [points] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define GL __attribute__((address_space(1)))
#define LD __attribute__((address_space(3)))
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256, 2) void probe(const int* __restrict__ idx, const char* __restrict__ rec48, const char* __restrict__ rec36,
                                                const char* __restrict__ rec64, int n_points, int steps_total, double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LD char* stage = (LD char*)smem + wave * 16384;
  const int waves = gridDim.x * 4, w = blockIdx.x * 4 + wave;
  double acc = 0.0;
  for (int step = w; step < steps_total; step += waves) {
    const int base = (step * 64) % n_points;  // this wave-step's 64 source points
    if (V == 0 || V == 1 || V == 4) {
      const int4 j = *reinterpret_cast<const int4*>(idx + 4 * (size_t)(base + lane));
      const int jj[4] = {V == 4 ? 0 : j.x, V == 4 ? 0 : j.y, V == 4 ? 0 : j.z, V == 4 ? 0 : j.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const char* p = rec48 + 48 * (size_t)jj[c];
        __builtin_amdgcn_global_load_lds((const GL void*)p, (LD void*)(stage + c * 2304), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GL void*)(p + 16), (LD void*)(stage + c * 2304 + 1024), 16, 0, 0);
        if (V != 1) __builtin_amdgcn_global_load_lds((const GL void*)(p + 32), (LD void*)(stage + c * 2304 + 2048), 4, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const v2d a = *(const LD v2d*)(stage + c * 2304 + 16 * lane);
        const v4f b = *(const LD v4f*)(stage + c * 2304 + 1024 + 16 * lane);
        const float z = V == 1 ? 0.f : *(const LD float*)(stage + c * 2304 + 2048 + 4 * lane);
        acc += a.x + a.y + (double)(b.x + b.y + b.z + b.w + z);
      }
    } else if (V == 2) {
      // record R = 21 i + lane / 3 of the step's 256 (R = 4 * source + slot), piece lane % 3
      const int r = lane / 3, piece = lane - 3 * r;
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        const int R = min(21 * i + r, 255);
        const int j = idx[4 * (size_t)base + R];
        const char* p = rec36 + 36 * (size_t)j + 12 * piece;
        __builtin_amdgcn_global_load_lds((const GL void*)p, (LD void*)(stage + i * 768), 12, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int R = 4 * lane + c, i = R / 21, pos = R - 21 * i;
        const LD float* q = (const LD float*)(stage + i * 768 + pos * 36);
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += (double)q[k];
      }
    } else {
      const int r = lane >> 2, piece = lane & 3;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int R = 16 * i + r;
        const int j = idx[4 * (size_t)base + R];
        const char* p = rec64 + 64 * (size_t)j + 16 * piece;
        __builtin_amdgcn_global_load_lds((const GL void*)p, (LD void*)(stage + i * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int R = 4 * lane + c, i = R >> 4, pos = R & 15;
        const v2d a = *(const LD v2d*)(stage + i * 1024 + pos * 64);
        const v4f b = *(const LD v4f*)(stage + i * 1024 + pos * 64 + 16);
        const v4f d = *(const LD v4f*)(stage + i * 1024 + pos * 64 + 32);
        acc += a.x + a.y + (double)(b.x + b.y + b.z + b.w + d.x);
      }
    }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <int V>
static float run(const int* idx, const char* r48, const char* r36, const char* r64, int n, int steps, double* out, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((probe<V>), dim3(512), dim3(256), 65536, 0, idx, r48, r36, r64, n, steps, out);
  CK(hipEventRecord(e0, 0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((probe<V>), dim3(512), dim3(256), 65536, 0, idx, r48, r36, r64, n, steps, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3f * ms / reps;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 100000, reps = argc > 2 ? std::atoi(argv[2]) : 20;
  const int pairs = 64;  // wave-steps of a launch: pairs x n / 64
  const int steps = (int)((long long)pairs * n / 64);
  // neighbour indices like a K = 4 search between curve-ordered clouds: near the source's own position, shared with its neighbours
  std::mt19937 rng(7);
  std::vector<int> idx((size_t)4 * n);
  std::normal_distribution<float> jitter(0.f, 6.f);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 4; ++c) {
      int j = i + (int)std::lround(jitter(rng)) + 3 * c;
      idx[4 * (size_t)i + c] = j < 0 ? 0 : (j >= n ? n - 1 : j);
    }
  int* d_idx; char *r48, *r36, *r64; double* out;
  CK(hipMalloc(&d_idx, sizeof(int) * idx.size() + 4096));
  CK(hipMalloc(&r48, (size_t)48 * n + 4096)); CK(hipMalloc(&r36, (size_t)36 * n + 4096)); CK(hipMalloc(&r64, (size_t)64 * n + 4096));
  CK(hipMalloc(&out, 64));
  CK(hipMemcpy(d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
  CK(hipMemset(r48, 0, (size_t)48 * n)); CK(hipMemset(r36, 0, (size_t)36 * n)); CK(hipMemset(r64, 0, (size_t)64 * n));
  for (int v = 0; v < 5; ++v) {
    CK(hipFuncSetAttribute(v == 0 ? (const void*)probe<0> : v == 1 ? (const void*)probe<1> : v == 2 ? (const void*)probe<2> : v == 3 ? (const void*)probe<3> : (const void*)probe<4>,
                           hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  }
  const float t0 = run<0>(d_idx, r48, r36, r64, n, steps, out, reps), t1 = run<1>(d_idx, r48, r36, r64, n, steps, out, reps);
  const float t2 = run<2>(d_idx, r48, r36, r64, n, steps, out, reps), t3 = run<3>(d_idx, r48, r36, r64, n, steps, out, reps);
  const float t4 = run<4>(d_idx, r48, r36, r64, n, steps, out, reps);
  const double recs = (double)steps * 256;
  std::printf("{\"points\": %d, \"wave_steps\": %d, \"records_gathered\": %.0f, \"us_per_launch\": {\"v0_product_16_16_4_per_lane\": %.1f, \"v1_without_the_4_byte_piece\": %.1f, "
              "\"v2_three_lanes_per_36B_record_dwordx3\": %.1f, \"v3_four_lanes_per_64B_record\": %.1f, \"v4_all_lanes_one_record\": %.1f}, "
              "\"ns_per_1000_records\": {\"v0\": %.2f, \"v1\": %.2f, \"v2\": %.2f, \"v3\": %.2f, \"v4\": %.2f}}\n",
              n, steps, recs, t0, t1, t2, t3, t4, 1e6 * t0 / recs, 1e6 * t1 / recs, 1e6 * t2 / recs, 1e6 * t3 / recs, 1e6 * t4 / recs);
  return 0;
}
