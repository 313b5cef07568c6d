// Status strings and version of the libsv3d_hip.so C ABI (include/sv3d_hip.h).
#include "sv3d_hip.h"

extern "C" const char *sv3d_strerror(int status)
{
    switch (status) {
    case SV3D_OK: return "ok";
    case SV3D_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive size or bad enum)";
    case SV3D_ERR_UNSUPPORTED: return "size outside the supported range";
    case SV3D_ERR_WORKSPACE: return "workspace too small for this call (or input beyond the built-in array of a *_static call)";
    case SV3D_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    case SV3D_ERR_NOT_READY: return "first use of a *_static entry point on this GPU must happen outside stream capture";
    default: return "unknown sv3d status";
    }
}

extern "C" int sv3d_abi_version(void) { return 3; }
