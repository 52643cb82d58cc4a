// exchange.hip — the round's exchange step inside the C ABI: RCCL all-gather over xGMI on the caller's HIP stream.
//
// The reference exchanges committed trajectories on the /trajs topic (neptune_ros.cpp:434-480 publish, :379-430 receive);
// here a rank owns a block of agents and one all-gather per round rebuilds what every rank replans against: the interval
// hulls of everybody's committed trajectories (nep_batch_exchange_hulls, the default) or the records themselves
// (nep_batch_exchange_records).  RCCL is bound at run time (dlopen of librccl.so, whichever copy the process already
// holds — PyTorch ships one — or the system's): the library has no link-time dependency on it and a single-GPU user
// never touches it.  Types and prototypes are RCCL's own (<rccl/rccl.h>).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <mutex>
#include <string>

#include "nep_device.h"

namespace nep {
void set_last_error(const std::string& msg);
int64_t batch_hull_block_bytes(const struct ::nep_batch* h);
void batch_dims(const struct ::nep_batch* h, int* n_scenes, int* n_local, int* num_agents);
}

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
  bool load() {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (lib) return true;
    // RTLD_NOLOAD first: reuse the copy the process already mapped (torch.distributed's), so that one RCCL serves both
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (lib) break; }
    if (!lib) for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
    if (!lib) { const char* e = dlerror(); err = std::string("librccl.so not found: ") + (e ? e : ""); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather || !CommCount || !GetErrorString) { err = "librccl.so lacks an expected symbol"; dlclose(lib); lib = nullptr; return false; }
    return true;
  }
};
Rccl g_rccl;

int fail_nccl(const char* what, ncclResult_t r) {
  nep::set_last_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
  return NEP_E_HIP;
}

// [W][S][nl] records (rank-major, as the all-gather delivers them) -> [S][W * nl] (scene-major, id order)
__global__ void regroup_records_kernel(const double* __restrict__ src, double* __restrict__ dst, int W, int S, int nl, int words) {
  const long total = (long)W * S * nl * words;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int w = (int)(e % words); long r = e / words;
    const int a = (int)(r % nl); r /= nl;
    const int s = (int)(r % S); const int k = (int)(r / S);
    dst[(((long)s * W + k) * nl + a) * words + w] = src[e];
  }
}

}  // namespace

struct nep_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  // one staging buffer per kind of exchange ([0] records, [1] slots): the two may be in flight on different streams, and a
  // captured graph keeps the pointer it was captured with
  double* staging[2] = {nullptr, nullptr}; size_t staging_bytes[2] = {0, 0};
  // grows staging[k] to `bytes`.  Never while `stream` is capturing: hipMalloc / hipFree are not capturable and a graph captured
  // earlier holds the old pointer — the caller reserves first (nep_comm_reserve, or one eager call of the exchange).
  int ensure(int k, size_t bytes, hipStream_t stream) {
    if (staging_bytes[k] >= bytes) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      nep::set_last_error("exchange staging buffer must grow while the stream is capturing: call nep_comm_reserve (or run the exchange once eagerly) before the capture");
      return NEP_E_ARG;
    }
    if (staging[k]) { (void)hipDeviceSynchronize(); (void)hipFree(staging[k]); }      // (a kernel of an earlier exchange may still read it)
    staging[k] = nullptr; staging_bytes[k] = 0;
    if (hipMalloc((void**)&staging[k], bytes) != hipSuccess) { nep::set_last_error("hipMalloc(exchange staging)"); return NEP_E_HIP; }
    staging_bytes[k] = bytes;
    return 0;
  }
};

extern "C" {

int nep_comm_unique_id(uint8_t id_out[128]) {
  if (!id_out) { nep::set_last_error("null argument"); return NEP_E_ARG; }
  if (!g_rccl.load()) { nep::set_last_error(g_rccl.err); return NEP_E_HIP; }
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail_nccl("ncclGetUniqueId", r);
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId size");
  for (int i = 0; i < 128; i++) id_out[i] = (uint8_t)id.internal[i];
  return 0;
}

nep_comm_t* nep_comm_create(const uint8_t id[128], int32_t world, int32_t rank) {
  if (!id || world < 1 || rank < 0 || rank >= world) { nep::set_last_error("bad communicator arguments"); return nullptr; }
  if (!g_rccl.load()) { nep::set_last_error(g_rccl.err); return nullptr; }
  ncclUniqueId uid;
  for (int i = 0; i < 128; i++) uid.internal[i] = (char)id[i];
  nep_comm* c = new nep_comm();
  c->world = world; c->rank = rank;
  const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);     // on the calling thread's current HIP device
  if (r != ncclSuccess) { fail_nccl("ncclCommInitRank", r); delete c; return nullptr; }
  return c;
}

int nep_comm_nranks(nep_comm_t* c) {
  if (!c || !c->comm) { nep::set_last_error("null communicator"); return NEP_E_ARG; }
  int n = 0;
  const ncclResult_t r = g_rccl.CommCount(c->comm, &n);
  if (r != ncclSuccess) return fail_nccl("ncclCommCount", r);
  return n;
}

void nep_comm_destroy(nep_comm_t* c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  for (int k = 0; k < 2; k++) if (c->staging[k]) hipFree(c->staging[k]);
  delete c;
}

// Sizes the staging buffers of the two regrouping exchanges ahead of a graph capture: records_bytes / slots_bytes = the gathered
// size (world x one rank's piece) the largest nep_batch_exchange_records / nep_batch_exchange_slots call will need; 0 leaves a
// buffer as it is.  Synchronises the device when a buffer has to be replaced.
int nep_comm_reserve(nep_comm_t* c, int64_t records_bytes, int64_t slots_bytes) {
  if (!c || records_bytes < 0 || slots_bytes < 0) { nep::set_last_error("bad arguments"); return NEP_E_ARG; }
  if (records_bytes > 0) if (int e = c->ensure(0, (size_t)records_bytes, nullptr)) return e;
  if (slots_bytes > 0) if (int e = c->ensure(1, (size_t)slots_bytes, nullptr)) return e;
  return 0;
}

int nep_batch_exchange_hulls(nep_batch_t* h, nep_comm_t* c, const void* d_block, void* d_blocks, void* stream) {
  if (!h || !c || !d_block || !d_blocks) { nep::set_last_error("null argument"); return NEP_E_ARG; }
  const size_t bytes = (size_t)nep::batch_hull_block_bytes(h);
  const ncclResult_t r = g_rccl.AllGather(d_block, d_blocks, bytes, ncclChar, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail_nccl("ncclAllGather(hull blocks)", r);
  return 0;
}

int nep_batch_exchange_records(nep_batch_t* h, nep_comm_t* c, const nep_traj_rec* d_commit_local, nep_traj_rec* d_committed_all, void* stream) {
  if (!h || !c || !d_commit_local || !d_committed_all) { nep::set_last_error("null argument"); return NEP_E_ARG; }
  int S = 0, nl = 0, N = 0;
  nep::batch_dims(h, &S, &nl, &N);
  if (nl * c->world != N) { nep::set_last_error("world * n_local must equal num_agents"); return NEP_E_ARG; }
  const size_t piece = (size_t)S * nl * sizeof(nep_traj_rec);
  if (int e = c->ensure(0, piece * c->world, (hipStream_t)stream)) return e;
  const ncclResult_t r = g_rccl.AllGather(d_commit_local, c->staging[0], piece, ncclChar, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail_nccl("ncclAllGather(records)", r);
  const int words = (int)(sizeof(nep_traj_rec) / sizeof(double));
  const long total = (long)c->world * S * nl * words;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(regroup_records_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c->staging[0], (double*)d_committed_all, c->world, S, nl, words);
  if (hipGetLastError() != hipSuccess) { nep::set_last_error("regroup_records_kernel launch"); return NEP_E_HIP; }
  return 0;
}

// Any per-slot array [n_scenes][n_local][bytes_per_slot] of every rank -> [n_scenes][N][bytes_per_slot] on every rank (all-gather +
// regrouping; bytes_per_slot a multiple of 8): e.g. the entangle states at point A the safety pass's re-check reads of every agent.
int nep_batch_exchange_slots(nep_batch_t* h, nep_comm_t* c, const void* d_local, void* d_all, int64_t bytes_per_slot, void* stream) {
  if (!h || !c || !d_local || !d_all || bytes_per_slot <= 0 || bytes_per_slot % 8) { nep::set_last_error("bad arguments (bytes_per_slot must be a positive multiple of 8)"); return NEP_E_ARG; }
  int S = 0, nl = 0, N = 0;
  nep::batch_dims(h, &S, &nl, &N);
  if (nl * c->world != N) { nep::set_last_error("world * n_local must equal num_agents"); return NEP_E_ARG; }
  const size_t piece = (size_t)S * nl * (size_t)bytes_per_slot;
  if (int e = c->ensure(1, piece * c->world, (hipStream_t)stream)) return e;
  const ncclResult_t r = g_rccl.AllGather(d_local, c->staging[1], piece, ncclChar, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return fail_nccl("ncclAllGather(slots)", r);
  const int words = (int)(bytes_per_slot / 8);
  const long total = (long)c->world * S * nl * words;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(regroup_records_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, c->staging[1], (double*)d_all, c->world, S, nl, words);
  if (hipGetLastError() != hipSuccess) { nep::set_last_error("regroup kernel launch"); return NEP_E_HIP; }
  return 0;
}

// Test hook: the regrouping step of nep_batch_exchange_records for any world size, on device buffers
// (src: [W][S][nl] records as an all-gather delivers them, dst: [S][W * nl]).
int nep_debug_regroup_records(const nep_traj_rec* d_src, nep_traj_rec* d_dst, int32_t world, int32_t n_scenes, int32_t n_local, void* stream) {
  if (!d_src || !d_dst || world < 1 || n_scenes < 1 || n_local < 1) { nep::set_last_error("bad arguments"); return NEP_E_ARG; }
  const int words = (int)(sizeof(nep_traj_rec) / sizeof(double));
  const long total = (long)world * n_scenes * n_local * words;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(regroup_records_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)d_src, (double*)d_dst, world, n_scenes, n_local, words);
  if (hipGetLastError() != hipSuccess) { nep::set_last_error("regroup_records_kernel launch"); return NEP_E_HIP; }
  return 0;
}

}  // extern "C"
