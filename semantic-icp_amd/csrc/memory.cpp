// memory.cpp -- device memory arena and the pool of recycled clouds (see engine.hpp for the types).
#include "engine.hpp"

namespace sicp {
namespace host {

DevArena& dev_arena() {
  static DevArena* a = new DevArena;  // never destroyed: it may outlive the HIP runtime at process exit
  return *a;
}

CloudPool& cloud_pool() {
  static CloudPool* pool = new CloudPool;
  return *pool;
}

std::shared_ptr<Cloud> acquire_cloud(int device) {
  CloudPool& pool = cloud_pool();
  const int slot = device % kPoolDevices;
  Cloud* c = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool.m);
    if (!pool.free_list[slot].empty()) { c = pool.free_list[slot].back(); pool.free_list[slot].pop_back(); }
  }
  if (!c) c = new Cloud();
  return std::shared_ptr<Cloud>(c, [slot, device](Cloud* dead) {
    // (a stream's cloud may be dropped without any handle having waited for its upload on the host: settle it
    // while the uploading stream still exists -- see settle_cloud)
    if (dead->pending && dead->ready_ev) (void)hipEventSynchronize(dead->ready_ev);
    dead->pending = false;
    dead->n = 0; dead->n_caller = 0; dead->is_set = false; dead->has_label = false; dead->layout = -1;
    dead->keep.clear(); dead->drop_i.clear(); dead->drop_xyz.clear();
    dead->feat_valid = false; dead->cov_general = false; dead->proj_valid = false; dead->feat_epoch = 0; dead->proj_cm_id = 0;
    CloudPool& pl = cloud_pool();
    {
      std::lock_guard<std::mutex> lock(pl.m);
      if (pl.free_list[slot].size() < kPoolCap) { pl.free_list[slot].push_back(dead); return; }
    }
    // the pool is full: free this one (hipFree synchronises the device -- only beyond the cap)
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    {
      DevArena::FreeScope once(device);  // (one wait for the device, not one per feature buffer)
      delete dead;
    }
    if (switched) (void)hipSetDevice(cur);
  });
}

unsigned long long next_epoch() {
  static std::atomic<unsigned long long> counter{0};
  return ++counter;
}

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

bool debug_enabled() {
  static const bool on = std::getenv("SICP_DEBUG") != nullptr;
  return on;
}

}  // namespace host
}  // namespace sicp
