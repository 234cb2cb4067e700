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

