// Version / error strings of the C ABI.
#include "tl_common.h"

extern "C" {

int tl_version(void) { return 1; }

const char* tl_error_string(int code) {
  switch (code) {
    case TL_OK: return "ok";
    case TL_ERR_ARG: return "invalid argument";
    case TL_ERR_LAUNCH: return "HIP launch / runtime error";
    case TL_ERR_UNSUPPORTED: return "unsupported configuration";
  }
  return "unknown error";
}

}  // extern "C"

// TEMPORARY until tl_cluster.hip lands
extern "C" int64_t tl_cluster_ws_bytes(int64_t n) { return 0; }
extern "C" int tl_cluster_grid(const float*, int64_t, float, int32_t*, int32_t*, void*, tl_stream_t) { return TL_ERR_UNSUPPORTED; }
