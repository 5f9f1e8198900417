// Small runtime queries the host side needs and torch does not expose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qt_hip.h"

extern "C" int qt_stream_capture_id(void *stream, unsigned long long *id) {
    if (!id) return QT_ERR_BAD_ARG;
    *id = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    if (hipStreamGetCaptureInfo((hipStream_t)stream, &cs, &cid) != hipSuccess) {
        (void)hipGetLastError();
        return QT_ERR_NO_DEVICE;
    }
    if (cs == hipStreamCaptureStatusActive) *id = cid ? cid : ~0ull;
    return QT_OK;
}
