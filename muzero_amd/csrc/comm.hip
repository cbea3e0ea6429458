// The one collective on the path: the broadcast of fresh weights from the rank that trained them to
// every rank's search engine (the reference shares one network object inside one process,
// train.py:155).  One flat device buffer, RCCL over xGMI; a root with 7 peers on a fully connected
// node sends on all 7 links at once, so the library's broadcast is used as it is.
//
// RCCL is bound at run time (dlopen), not at link time: the library that is already in the process
// (PyTorch's) is preferred, so that there is exactly one RCCL instance; libmuzero_amd.so itself loads
// and works on a machine without RCCL, and only these entry points fail there.
#include <dlfcn.h>

#include "common.h"

namespace {

struct UniqueId { char bytes[128]; };  // ncclUniqueId
typedef void* comm_t;
typedef int (*get_unique_id_fn)(UniqueId*);
typedef int (*comm_init_rank_fn)(comm_t*, int, UniqueId, int);
typedef int (*broadcast_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int, comm_t, hipStream_t);
typedef int (*comm_destroy_fn)(comm_t);
typedef const char* (*error_string_fn)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  broadcast_fn broadcast = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  error_string_fn error_string = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)  // a copy already loaded into the process first
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.handle)
      for (const char* n : names)
        if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (r.handle) {
      r.get_unique_id = reinterpret_cast<get_unique_id_fn>(dlsym(r.handle, "ncclGetUniqueId"));
      r.comm_init_rank = reinterpret_cast<comm_init_rank_fn>(dlsym(r.handle, "ncclCommInitRank"));
      r.broadcast = reinterpret_cast<broadcast_fn>(dlsym(r.handle, "ncclBroadcast"));
      r.comm_destroy = reinterpret_cast<comm_destroy_fn>(dlsym(r.handle, "ncclCommDestroy"));
      r.error_string = reinterpret_cast<error_string_fn>(dlsym(r.handle, "ncclGetErrorString"));
    }
  }
  const bool ok = r.handle && r.get_unique_id && r.comm_init_rank && r.broadcast && r.comm_destroy;
  return ok ? &r : nullptr;
}

int rccl_fail(const char* what, int code) {
  Rccl* r = rccl();
  return mz::fail(MZ_E_LAUNCH, "%s: RCCL error %d (%s)", what, code,
                  r && r->error_string ? r->error_string(code) : "?");
}

struct Comm {
  comm_t comm;
  int rank, world;
};

}  // namespace

extern "C" {

int mz_comm_unique_id(void* id128) {
  MZ_REQUIRE(id128, "mz_comm_unique_id: null buffer");
  Rccl* r = rccl();
  if (!r) return mz::fail(MZ_E_STATE, "mz_comm_unique_id: RCCL (librccl.so) could not be loaded");
  if (int rc = r->get_unique_id(static_cast<UniqueId*>(id128))) return rccl_fail("mz_comm_unique_id", rc);
  return MZ_OK;
}

int mz_comm_init(const void* id128, int32_t rank, int32_t world, void** comm_out) {
  MZ_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "mz_comm_init: bad arguments");
  Rccl* r = rccl();
  if (!r) return mz::fail(MZ_E_STATE, "mz_comm_init: RCCL (librccl.so) could not be loaded");
  UniqueId id = *static_cast<const UniqueId*>(id128);
  comm_t c = nullptr;
  if (int rc = r->comm_init_rank(&c, world, id, rank)) return rccl_fail("mz_comm_init", rc);
  *comm_out = new Comm{c, rank, world};
  return MZ_OK;
}

int mz_comm_bcast_weights(void* comm, void* buffer, int64_t bytes, int32_t root, mz_stream_t stream) {
  MZ_REQUIRE(comm && buffer && bytes > 0, "mz_comm_bcast_weights: bad arguments");
  Comm* c = static_cast<Comm*>(comm);
  MZ_REQUIRE(root >= 0 && root < c->world, "mz_comm_bcast_weights: root %d outside the %d ranks", root, c->world);
  Rccl* r = rccl();
  if (!r) return mz::fail(MZ_E_STATE, "mz_comm_bcast_weights: RCCL is not loaded");
  // ncclInt8 = 0: the buffer is opaque bytes (float32 parameters and int64 counters alike)
  if (int rc = r->broadcast(buffer, buffer, (size_t)bytes, 0, root, c->comm, mz::as_stream(stream)))
    return rccl_fail("mz_comm_bcast_weights", rc);
  return MZ_OK;
}

int mz_comm_destroy(void* comm) {
  if (!comm) return MZ_OK;
  Comm* c = static_cast<Comm*>(comm);
  Rccl* r = rccl();
  int rc = r ? r->comm_destroy(c->comm) : 0;
  delete c;
  return rc ? rccl_fail("mz_comm_destroy", rc) : MZ_OK;
}

}  // extern "C"
