// pmx_shard.cpp — the multi-GPU part of the C ABI (include/pmx.h "sharding across GPUs").
//
// The reference's population loop (likelihood/matrix.rs:79-98: rayon over subjects, serial over support points) has no
// cross-iteration state, so the path shards by SUBJECT with no data-path collective: one process per GPU, each with the
// population of its own subject range (pmx_shard_bounds + pmx_population_create_shard) and the full support grid.  The
// one optional exchange - a caller that wants every rank's prediction rows on every device - is an in-place all-gather
// of the row blocks over RCCL (xGMI inside a node).  librccl.so (570 MB) is opened on first use, never at load time.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pmx.h"

namespace pmx {
// thread-local error text of the C ABI (pmx_api.cpp)
int32_t set_error(int32_t code, const std::string& msg);
}  // namespace pmx

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

// the process's RCCL: the copy another component (PyTorch) already loaded, else ROCm's
const Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.handle || !g_rccl.error.empty()) return &g_rccl;
  const char* const names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if ((g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) != nullptr) break;
  if (!g_rccl.handle)
    for (const char* n : names)
      if ((g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
  if (!g_rccl.handle) {
    g_rccl.error = std::string("librccl.so could not be opened: ") + dlerror();
    return &g_rccl;
  }
  bool ok = true;
  auto sym = [&](const char* name) {
    void* p = dlsym(g_rccl.handle, name);
    if (!p) {
      ok = false;
      g_rccl.error = std::string("librccl.so has no symbol ") + name;
    }
    return p;
  };
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
  g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
  g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
  g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
  g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
  if (!ok) {
    dlclose(g_rccl.handle);
    g_rccl.handle = nullptr;
  }
  return &g_rccl;
}

int32_t nccl_fail(const Rccl* r, const char* what, ncclResult_t e) {
  return pmx::set_error(PMX_ERR_HIP, std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(e) : "RCCL error"));
}

}  // namespace

struct pmx_comm {
  ncclComm_t comm = nullptr;
  int32_t n_ranks = 0, rank = 0, device = 0;
};

extern "C" {

int32_t pmx_shard_bounds(const pmx_population_desc* d, int32_t n_shards, int64_t* bounds) {
  if (!d || !bounds || n_shards < 1) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_shard_bounds: null argument or n_shards < 1");
  const int64_t S = d->n_subjects;
  if (S < 0 || (S > 0 && (!d->subj_occ_off || !d->occ_ev_off)))
    return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_shard_bounds: null offset arrays");
  // Contiguous ranges with equal shares of the events (= subject-event-steps per support point): rank r starts at the
  // first subject whose preceding subjects hold at least r/n of all events.  Subjects without events weigh nothing; a
  // population without any event is split by subject count.
  auto events_before = [&](int64_t s) { return d->occ_ev_off[d->subj_occ_off[s]]; };  // events of subjects [0, s)
  const int64_t total = S > 0 ? events_before(S) : 0;
  bounds[0] = 0;
  for (int32_t r = 1; r < n_shards; ++r) {
    int64_t cut;
    if (total == 0) {
      cut = (S * r + n_shards - 1) / n_shards;
    } else {
      // smallest s with events_before(s) * n >= total * r   (a searchsorted on the running event count)
      int64_t lo = 0, hi = S;
      while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (static_cast<__int128>(events_before(mid)) * n_shards >= static_cast<__int128>(total) * r)
          hi = mid;
        else
          lo = mid + 1;
      }
      cut = lo;
    }
    if (cut < bounds[r - 1]) cut = bounds[r - 1];
    if (cut > S) cut = S;
    bounds[r] = cut;
  }
  bounds[n_shards] = S;
  return PMX_OK;
}

int32_t pmx_shard_rows(const pmx_population_desc* d, int32_t n_shards, const int64_t* bounds, int64_t* rows) {
  if (!d || !bounds || !rows || n_shards < 1) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_shard_rows: null argument");
  const int64_t S = d->n_subjects;
  for (int32_t r = 0; r <= n_shards; ++r)
    if (bounds[r] < 0 || bounds[r] > S || (r > 0 && bounds[r] < bounds[r - 1]))
      return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_shard_rows: bounds must be a non-decreasing cover of [0, n_subjects]");
  // prediction rows = observations in event order (SubjectPredictions::flat_predictions, subject.rs:145-148)
  int64_t row = 0;
  int32_t r = 0;
  while (r <= n_shards && bounds[r] == 0) rows[r++] = 0;
  for (int64_t s = 0; s < S; ++s) {
    const int64_t e0 = d->occ_ev_off[d->subj_occ_off[s]], e1 = d->occ_ev_off[d->subj_occ_off[s + 1]];
    for (int64_t e = e0; e < e1; ++e) row += d->ev_kind[e] == PMX_EV_OBSERVATION;
    while (r <= n_shards && bounds[r] == s + 1) rows[r++] = row;
  }
  return PMX_OK;
}

int32_t pmx_comm_unique_id(uint8_t* id) {
  if (!id) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_comm_unique_id: id is null");
  const Rccl* r = rccl();
  if (!r->handle) return pmx::set_error(PMX_ERR_NO_DEVICE, r->error);
  static_assert(PMX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "pmx.h and rccl.h disagree on the id size");
  ncclUniqueId u;
  const ncclResult_t e = r->GetUniqueId(&u);
  if (e != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId", e);
  std::memcpy(id, u.internal, PMX_COMM_ID_BYTES);
  return PMX_OK;
}

int32_t pmx_comm_create(const uint8_t* id, int32_t n_ranks, int32_t rank, int32_t device, pmx_comm** out) {
  if (!id || !out) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_comm_create: null argument");
  *out = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_comm_create: rank out of range");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return pmx::set_error(PMX_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n_dev) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_comm_create: device ordinal out of range");
  const Rccl* r = rccl();
  if (!r->handle) return pmx::set_error(PMX_ERR_NO_DEVICE, r->error);
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device) != hipSuccess) return pmx::set_error(PMX_ERR_HIP, "hipSetDevice failed");
  ncclUniqueId u;
  std::memcpy(u.internal, id, PMX_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t e = r->CommInitRank(&c, n_ranks, u, rank);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != ncclSuccess) return nccl_fail(r, "ncclCommInitRank", e);
  pmx_comm* pc = new pmx_comm;
  pc->comm = c;
  pc->n_ranks = n_ranks;
  pc->rank = rank;
  pc->device = device;
  *out = pc;
  return PMX_OK;
}

void pmx_comm_destroy(pmx_comm* comm) {
  if (!comm) return;
  const Rccl* r = rccl();
  if (r->handle && comm->comm) (void)r->CommDestroy(comm->comm);
  delete comm;
}

int32_t pmx_comm_size(const pmx_comm* comm) { return comm ? comm->n_ranks : -1; }
int32_t pmx_comm_rank(const pmx_comm* comm) { return comm ? comm->rank : -1; }

int32_t pmx_allgather_predictions(pmx_comm* comm, double* d_full, const int64_t* rows, int64_t ld, void* stream) {
  if (!comm || !d_full || !rows) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_allgather_predictions: null argument");
  if (ld < 1) return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_allgather_predictions: ld must be >= 1");
  const int32_t n = comm->n_ranks;
  bool equal = true;
  for (int32_t r = 0; r < n; ++r) {
    if (rows[r + 1] < rows[r] || rows[0] != 0)
      return pmx::set_error(PMX_ERR_INVALID_ARGUMENT, "pmx_allgather_predictions: rows must start at 0 and not decrease");
    if (rows[r + 1] - rows[r] != rows[1] - rows[0]) equal = false;
  }
  if (const char* e = std::getenv("PMX_DEBUG_ALLGATHER_BROADCAST"); e && e[0] == '1') equal = false;  // (tests: both paths on one rank)
  const Rccl* rc = rccl();
  if (!rc->handle) return pmx::set_error(PMX_ERR_NO_DEVICE, rc->error);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != comm->device && hipSetDevice(comm->device) != hipSuccess) return pmx::set_error(PMX_ERR_HIP, "hipSetDevice failed");
  ncclResult_t e = ncclSuccess;
  const char* what = "ncclAllGather";
  if (equal) {
    // every block the same size: ONE in-place all-gather (send buffer = this rank's block inside the receive buffer)
    const size_t count = static_cast<size_t>(rows[1] - rows[0]) * static_cast<size_t>(ld);
    if (count > 0) e = rc->AllGather(d_full + rows[comm->rank] * ld, d_full, count, ncclDouble, comm->comm, st);
  } else {
    // unequal blocks (events-balanced shards of a ragged population): every block is broadcast in place from its owner,
    // all n fused into one group so that they progress together - an all-gather-v without padding or a second copy
    what = "ncclBroadcast (grouped)";
    e = rc->GroupStart();
    for (int32_t r = 0; r < n && e == ncclSuccess; ++r) {
      const size_t count = static_cast<size_t>(rows[r + 1] - rows[r]) * static_cast<size_t>(ld);
      if (count == 0) continue;
      double* blk = d_full + rows[r] * ld;
      e = rc->Broadcast(blk, blk, count, ncclDouble, r, comm->comm, st);
    }
    const ncclResult_t ge = rc->GroupEnd();
    if (e == ncclSuccess) e = ge;
  }
  if (prev >= 0 && prev != comm->device) (void)hipSetDevice(prev);
  if (e != ncclSuccess) return nccl_fail(rc, what, e);
  return PMX_OK;
}

}  // extern "C"
