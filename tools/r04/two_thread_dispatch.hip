// Minimal reproducer for the rocprofv3 crash of round 3 / 4: two host threads dispatch at the same time -- one
// queues pinned H2D copies + small kernels + event records on its stream (what sicp_stream_add_cloud does on the
// submitting thread), the other launches an instantiated graph of kernel nodes and waits for it (what the stream's
// worker thread does).  No libsicp involved.
//   hipcc --offload-arch=gfx950 -O2 two_thread_dispatch.hip -o two_thread_dispatch -lpthread
//   rocprofv3 --kernel-trace --stats -d /tmp/x -- ./two_thread_dispatch 5
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(3); } } while (0)
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv) {
  const double secs = argc > 1 ? std::atof(argv[1]) : 5.0;
  const int mode = argc > 2 ? std::atoi(argv[2]) : 3;  // bit 0: uploader thread, bit 1: graph thread, 4: the graph loop in the MAIN thread, 8: as 2 with plain launches instead of the graph
  CK(hipSetDevice(0));
  const int n = 100000;
  std::atomic<bool> stop{false};
  std::atomic<long long> ups{0}, graphs{0};
  auto uploader = [&] {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *h, *d; CK(hipHostMalloc((void**)&h, n * sizeof(float), hipHostMallocDefault)); CK(hipMalloc((void**)&d, n * sizeof(float)));
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    while (!stop) {
      CK(hipMemcpyAsync(d, h, n * sizeof(float), hipMemcpyHostToDevice, st));
      for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, st, d, n);
      CK(hipMemcpyAsync(h, d, n * sizeof(float), hipMemcpyDeviceToHost, st));
      CK(hipEventRecord(ev, st));
      if ((++ups & 15) == 0) CK(hipEventSynchronize(ev));
    }
    CK(hipStreamSynchronize(st));
    CK(hipEventDestroy(ev)); CK(hipFree(d)); CK(hipHostFree(h)); CK(hipStreamDestroy(st));
  };
  auto grapher = [&] {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* d; CK(hipMalloc((void**)&d, n * sizeof(float))); CK(hipMemsetAsync(d, 0, n * sizeof(float), st));
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    int nn = n; void* args[] = {(void*)&d, (void*)&nn};
    hipKernelNodeParams p = {}; p.func = (void*)touch; p.gridDim = dim3((n + 255) / 256); p.blockDim = dim3(256); p.kernelParams = args;
    hipGraphNode_t prev = nullptr, node = nullptr;
    for (int k = 0; k < 16; ++k) { CK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &p)); prev = node; }
    hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    while (!stop) {
      if (mode & 8) { for (int k = 0; k < 16; ++k) hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, st, d, n); }
      else CK(hipGraphLaunch(ex, st));
      CK(hipStreamSynchronize(st)); ++graphs;
    }
    CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipFree(d)); CK(hipStreamDestroy(st));
  };
  std::vector<std::thread> th;
  if (mode & 1) th.emplace_back(uploader);
  if (mode & (2 | 8)) th.emplace_back(grapher);
  if (mode & 4) {  // graph launches from the main thread: a timer thread ends the loop
    std::thread timer([&] { std::this_thread::sleep_for(std::chrono::duration<double>(secs)); stop = true; });
    grapher();
    timer.join();
  } else {
    std::this_thread::sleep_for(std::chrono::duration<double>(secs));
  }
  stop = true;
  for (auto& t : th) t.join();
  std::printf("ok: %lld upload rounds, %lld graph launches\n", (long long)ups, (long long)graphs);
  return 0;
}
