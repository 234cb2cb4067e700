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
    case TL_ERR_ARENA: return "arena too small (tl_forward_args.needed_bytes)";
    case TL_ERR_REACH_ZERO: return "sparse conv output spatial shape reach zero!!!";
    case TL_ERR_BLK: return "block-local unit builder flagged skipped units in the previous forward of this context (internal assertion)";
    case TL_ERR_EXTENT: return "voxelize: tile extent exceeds spatial_shape, batch id out of range or voxel coordinate outside [0, 65536)";
  }
  return "unknown error";
}

}  // extern "C"

