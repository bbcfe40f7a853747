// gather.hip.h — the one exchange step of the multi-GPU path behind the C ABI: an all-gather of the tracked
// poses over RCCL, for hosts that are not Python (bench.py reaches RCCL through torch.distributed: odometry_amd/dist.py is the
// same schedule). Tracking shards by sequence (rank r owns sequences r, r + N, ...): there is no data-path collective; per
// tracked frame a rank contributes one row of 14 four-byte words (int32 sequence id, int32 frame id, 3x4 float pose, row-major), `every` rows per
// collective.
//
// The schedule is agreed up front, never derived from a rank's own frame count: with 11 sequences over 8 ranks some ranks push
// twice as many rows as others, so EVERY rank issues exactly ceil(n_max_frames / every) ncclAllGather calls of a fixed
// (every, 14) block and pads with rows whose sequence id is -1 once it has run out of frames.
//
// RCCL is loaded with dlopen at the first use (librccl.so.1): the library has no link-time dependency on it and loads on hosts
// without it. Collectives go to a stream of their own and are never waited for while tracking; odo_gather_flush waits.
// Included by odometry_hip.hip.
#pragma once
#include <dlfcn.h>

namespace odo_rccl {
// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat == 7, ncclSuccess == 0)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef int (*CommCountFn)(Comm, int*);
typedef const char* (*GetErrorStringFn)(int);
struct Api {
  void* so;
  GetUniqueIdFn get_unique_id;
  CommInitRankFn comm_init_rank;
  AllGatherFn all_gather;
  CommDestroyFn comm_destroy;
  CommCountFn comm_count;
  GetErrorStringFn error_string;
};
static Api* api() {
  static Api a = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {getenv("ODO_RCCL_SO"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n) continue;
      a.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (a.so) break;
    }
    if (a.so) {
      a.get_unique_id = (GetUniqueIdFn)dlsym(a.so, "ncclGetUniqueId");
      a.comm_init_rank = (CommInitRankFn)dlsym(a.so, "ncclCommInitRank");
      a.all_gather = (AllGatherFn)dlsym(a.so, "ncclAllGather");
      a.comm_destroy = (CommDestroyFn)dlsym(a.so, "ncclCommDestroy");
      a.comm_count = (CommCountFn)dlsym(a.so, "ncclCommCount");
      a.error_string = (GetErrorStringFn)dlsym(a.so, "ncclGetErrorString");
      if (!a.get_unique_id || !a.comm_init_rank || !a.all_gather || !a.comm_destroy) { dlclose(a.so); a.so = nullptr; }
    }
  }
  return a.so ? &a : nullptr;
}
}  // namespace odo_rccl

constexpr int kGatherRow = ODO_GATHER_ROW;

struct odo_gather {
  int device, world, rank, every;
  int n_local, n_max, n_coll;   // rows this rank will push, the largest such number over the ranks, collectives in the schedule
  int pushed, issued;
  odo_rccl::Comm comm;
  hipStream_t stream;
  float* h_send;   // pinned: n_coll x every x 14
  float* d_send;
  float* d_recv;   // n_coll x world x every x 14
  float* h_recv;   // pinned
  bool flushed;
  std::vector<std::vector<float>>* rows;   // per rank: the valid rows, in arrival order (after flush)
};

#define RCCL_OK(expr)                                                                                  \
  do {                                                                                                 \
    const int rc_ = (expr);                                                                            \
    if (rc_ != 0) {                                                                                    \
      odo_rccl::Api* a_ = odo_rccl::api();                                                             \
      return fail("%s failed: %s", #expr, (a_ && a_->error_string) ? a_->error_string(rc_) : "RCCL error"); \
    }                                                                                                  \
  } while (0)

extern "C" int odo_gather_unique_id(unsigned char id[ODO_GATHER_ID_BYTES]) {
  if (!id) return fail("odo_gather_unique_id: NULL arg");
  odo_rccl::Api* a = odo_rccl::api();
  if (!a) return fail("odo_gather: librccl.so.1 could not be loaded (set ODO_RCCL_SO)");
  odo_rccl::UniqueId u;
  RCCL_OK(a->get_unique_id(&u));
  static_assert(sizeof(u) == ODO_GATHER_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id, u.internal, ODO_GATHER_ID_BYTES);
  return 0;
}

extern "C" int odo_gather_destroy(odo_gather* g) {
  if (!g) return 0;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->comm) { odo_rccl::Api* a = odo_rccl::api(); if (a) (void)a->comm_destroy(g->comm); }
  if (g->h_send) (void)hipHostFree(g->h_send);
  if (g->h_recv) (void)hipHostFree(g->h_recv);
  if (g->d_send) (void)hipFree(g->d_send);
  if (g->d_recv) (void)hipFree(g->d_recv);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g->rows;
  delete g;
  return 0;
}

extern "C" int odo_gather_create(int device, int world, int rank, const unsigned char id[ODO_GATHER_ID_BYTES], int every,
                                 int n_local_frames, int n_max_frames, odo_gather** out) {
  if (!out || !id) return fail("odo_gather_create: NULL arg");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail("odo_gather_create: rank %d of %d", rank, world);
  if (every < 1 || n_local_frames < 0 || n_max_frames < n_local_frames)
    return fail("odo_gather_create: every >= 1 and 0 <= n_local_frames <= n_max_frames required");
  odo_rccl::Api* a = odo_rccl::api();
  if (!a) return fail("odo_gather: librccl.so.1 could not be loaded (set ODO_RCCL_SO)");
  odo_gather* g = new (std::nothrow) odo_gather();
  if (!g) return fail("out of memory");
  memset(g, 0, sizeof(*g));
  g->device = device; g->world = world; g->rank = rank; g->every = every;
  g->n_local = n_local_frames; g->n_max = n_max_frames;
  g->n_coll = (n_max_frames + every - 1) / every;
  g->rows = new std::vector<std::vector<float>>((size_t)world);
  const size_t blk = (size_t)every * kGatherRow;
  const size_t n_send = (size_t)(g->n_coll > 0 ? g->n_coll : 1) * blk, n_recv = n_send * (size_t)world;
  bool ok = hipSetDevice(device) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&g->h_send, sizeof(float) * n_send, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&g->h_recv, sizeof(float) * n_recv, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&g->d_send, sizeof(float) * n_send) == hipSuccess;
  ok = ok && hipMalloc((void**)&g->d_recv, sizeof(float) * n_recv) == hipSuccess;
  if (!ok) { odo_gather_destroy(g); return fail("odo_gather_create: device allocation failed"); }
  for (size_t i = 0; i < n_send; i++) g->h_send[i] = __builtin_nanf("");
  for (int c = 0; c < (g->n_coll > 0 ? g->n_coll : 1); c++)
    for (int r = 0; r < every; r++) { const int neg = -1; memcpy(&g->h_send[((size_t)c * every + r) * kGatherRow], &neg, sizeof(int)); }   // padding rows: sequence id -1
  odo_rccl::UniqueId u;
  memcpy(u.internal, id, ODO_GATHER_ID_BYTES);
  const int rc = a->comm_init_rank(&g->comm, world, u, rank);
  if (rc != 0) {
    g->comm = nullptr;
    odo_gather_destroy(g);
    return fail("ncclCommInitRank failed: %s", a->error_string ? a->error_string(rc) : "RCCL error");
  }
  *out = g;
  return 0;
}

// Issues collective `c` of the schedule: this rank's block c (rows pushed so far, the rest padding) to every rank.
static int gather_issue(odo_gather* g) {
  odo_rccl::Api* a = odo_rccl::api();
  const int c = g->issued;
  const size_t blk = (size_t)g->every * kGatherRow;
  HIP_OK(hipSetDevice(g->device));
  HIP_OK(hipMemcpyAsync(g->d_send + (size_t)c * blk, g->h_send + (size_t)c * blk, sizeof(float) * blk, hipMemcpyHostToDevice, g->stream));
  RCCL_OK(a->all_gather(g->d_send + (size_t)c * blk, g->d_recv + (size_t)c * blk * g->world, blk, 7 /* ncclFloat */, g->comm, g->stream));
  HIP_OK(hipMemcpyAsync(g->h_recv + (size_t)c * blk * g->world, g->d_recv + (size_t)c * blk * g->world, sizeof(float) * blk * g->world,
                        hipMemcpyDeviceToHost, g->stream));
  g->issued++;
  return 0;
}

extern "C" int odo_gather_push(odo_gather* g, int seq_id, int frame_id, const float abs_pose_colmajor[16]) {
  if (!g || !abs_pose_colmajor) return fail("odo_gather_push: NULL arg");
  if (g->flushed) return fail("odo_gather_push: after odo_gather_flush");
  if (g->pushed >= g->n_local) return fail("odo_gather_push: more rows pushed than announced (n_local_frames = %d)", g->n_local);
  float* row = g->h_send + (size_t)g->pushed * kGatherRow;   // blocks are contiguous: row p lives in block p / every
  memcpy(&row[0], &seq_id, sizeof(int));     // ids travel as int32 bit patterns in the float row: exact for any id
  memcpy(&row[1], &frame_id, sizeof(int));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) row[2 + i * 4 + j] = abs_pose_colmajor[j * 4 + i];
  g->pushed++;
  if (g->pushed % g->every == 0) return gather_issue(g);
  return 0;
}

extern "C" int odo_gather_flush(odo_gather* g) {
  if (!g) return fail("odo_gather_flush: NULL arg");
  if (g->flushed) return 0;
  while (g->issued < g->n_coll)
    if (gather_issue(g)) return -1;   // a short last block, or blocks of nothing but padding: the schedule is the same on every rank
  HIP_OK(hipSetDevice(g->device));
  HIP_OK(hipStreamSynchronize(g->stream));
  const size_t blk = (size_t)g->every * kGatherRow;
  for (int c = 0; c < g->n_coll; c++)
    for (int r = 0; r < g->world; r++) {
      const float* src = g->h_recv + ((size_t)c * g->world + r) * blk;
      for (int k = 0; k < g->every; k++) {
        int sid;
        memcpy(&sid, &src[(size_t)k * kGatherRow], sizeof(int));
        if (sid >= 0)
          (*g->rows)[r].insert((*g->rows)[r].end(), src + (size_t)k * kGatherRow, src + (size_t)(k + 1) * kGatherRow);
      }
    }
  g->flushed = true;
  return 0;
}

extern "C" int odo_gather_rows(odo_gather* g, int rank, const float** rows, int* n_rows) {
  if (!g || rank < 0 || rank >= g->world) return fail("odo_gather_rows: bad arg");
  if (!g->flushed) return fail("odo_gather_rows: call odo_gather_flush first");
  if (rows) *rows = (*g->rows)[rank].data();
  if (n_rows) *n_rows = (int)((*g->rows)[rank].size() / kGatherRow);
  return 0;
}

extern "C" int odo_gather_issued(const odo_gather* g) { return g ? g->issued : 0; }
// Number of ranks in the RCCL communicator, asked of the communicator itself (ncclCommCount), not echoed from `world`.
extern "C" int odo_gather_ranks(const odo_gather* g) {
  if (!g || !g->comm) return fail("odo_gather_ranks: no communicator");
  odo_rccl::Api* a = odo_rccl::api();
  if (!a || !a->comm_count) return fail("odo_gather_ranks: ncclCommCount not available");
  int n = 0;
  RCCL_OK(a->comm_count(g->comm, &n));
  return n;
}
