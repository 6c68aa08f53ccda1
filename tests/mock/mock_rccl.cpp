// mock_rccl.cpp — TEST DOUBLE for librccl's five entry points the library binds (capi.hip, rccl_api): lets N processes
// that share ONE GPU run qtr_comm_init / qtr_gather_results[_v] end to end (the count exchange, the padded blocks, the
// rank-uniform early-outs) on a box where real RCCL cannot form a communicator (it refuses two ranks on one device).
// Transport: a POSIX shared-memory segment named after the unique id, one slot per rank, a generation barrier.
// Selected with QTR_RCCL_LIB=<this .so>.  Never part of the product.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

namespace {
constexpr size_t kSlot = 1 << 20;  // bytes per rank and collective
constexpr int kMaxWorld = 16;
struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  char pad[56];
  char slots[kMaxWorld][kSlot];
};
struct Comm {
  Shared* sh;
  int rank, world;
  char name[64];
};
struct IdByValue {
  char internal[128];
};
bool barrier(Comm* c) {
  const int gen = c->sh->generation.load(std::memory_order_acquire);
  if (c->sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    c->sh->arrived.store(0, std::memory_order_release);
    c->sh->generation.store(gen + 1, std::memory_order_release);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->generation.load(std::memory_order_acquire) == gen) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;  // a rank never came: report, do not hang
  }
  return true;
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) int ncclGetUniqueId(void* id) {
  memset(id, 0, 128);
  snprintf((char*)id, 128, "/qtr_mock_rccl_%d_%lld", (int)getpid(),
           (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return 0;
}
__attribute__((visibility("default"))) int ncclCommInitRank(void** comm, int world, IdByValue id, int rank) {
  if (world < 1 || world > kMaxWorld) return 4;
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  snprintf(c->name, sizeof(c->name), "%s", id.internal);
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return 2;
  c->sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->sh == MAP_FAILED) return 2;
  *comm = c;
  return barrier(c) ? 0 : 6;  // (a fresh segment is zero-filled: counters start at 0)
}
__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t count, int /*dtype: ncclChar*/,
                                                         void* comm, hipStream_t st) {
  Comm* c = (Comm*)comm;
  if (count > kSlot) return 4;
  if (hipStreamSynchronize(st) != hipSuccess) return 1;
  if (hipMemcpy(c->sh->slots[c->rank], send, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  if (!barrier(c)) return 6;
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy((char*)recv + (size_t)r * count, c->sh->slots[r], count, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return barrier(c) ? 0 : 6;  // (nobody overwrites a slot before everybody has read it)
}
__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, sizeof(Shared));
  delete c;
  return 0;
}
__attribute__((visibility("default"))) const char* ncclGetErrorString(int rc) {
  return rc == 6 ? "mock transport: a rank did not arrive within 60 s" : "mock transport error";
}
}
