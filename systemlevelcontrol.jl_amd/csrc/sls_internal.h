// sls_internal.h — what the translation units behind include/sls_mi355x.h share (context object, error plumbing).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/sls_mi355x.h"

struct sls_ctx {
  std::vector<int> devs;
  std::vector<int> ncu;
  std::string err;
  uint32_t flags = 0;
  std::vector<double> ridge_x, ridge_u;   // sls_set_ridge (empty = none)
  // Per device slot: streams and the big scratch workspace are created once and lent to plans (hipStreamCreate costs
  // ≈4 ms and a GB-sized hipMalloc ≈10 ms on this stack — more than a whole README solve).  One context is used by one
  // thread at a time (header), so a simple "in use" flag is enough; a second concurrent plan gets its own.
  struct Slot {
    std::vector<hipStream_t> streams;   // [0] main, [1..] aux
    hipStream_t refine_stream = nullptr; // main stream of the refinement plan of sls_h2_sf_solve (it lives beside the main plan)
    std::vector<hipStream_t> streams_lo; // aux streams of the lowest priority (launches of a handful of workgroups)
    int streams_in_use = 0;
    void* scratch = nullptr; size_t scratch_bytes = 0; bool scratch_in_use = false;
    void* arena = nullptr; size_t arena_bytes = 0; bool arena_in_use = false;      // plan tables (one-shot calls: same size every time)
    void* ltab = nullptr; size_t ltab_bytes = 0; bool ltab_in_use = false;         // device-built tables of the localized route (same idea)
    // pinned staging ring of the download (sls_plan_download): kDlLanes lanes, each its own stream and pinned chunk
    void* pinned = nullptr; size_t pinned_bytes = 0;
    std::vector<hipStream_t> dl_streams;
  };
  std::vector<Slot> slots;
};

namespace sls {
// record the message on the context (and globally, for sls_last_error(NULL)) and hand back `code`
int fail(sls_ctx* ctx, int code, const std::string& msg);
int hipfail(sls_ctx* ctx, hipError_t e, const char* what);
// a plan / loop object may outlive its context (host-language GC order): checked before touching it
bool ctx_is_live(const sls_ctx* ctx);
}  // namespace sls

#define HIPCHK(ctx, call)                                        \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return sls::hipfail(ctx, e__, #call); \
  } while (0)
